set -x
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fp8 or hipgraph or growth" 2>&1 | tail -4
for f in 1 0 1 0; do SRGD_Q_FUSED=$f python bench.py --steps 5 --warmup 5 --no_cpu_baseline --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('fused=$f', d['value'], d['roofline']['achieved'], d['kernel_time_share'])"; done
