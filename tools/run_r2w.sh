python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fp8 or twin" 2>&1 | tail -3
for p in fp8_mixed fp8; do
  python bench.py --precision $p --ddpm_steps 100 --class_cond_scale 2.0 --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', d['value'], d['ms_per_step'])"
done
