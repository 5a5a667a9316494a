mkdir -p gpurun_out
python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "fused or conv1x1 or config1" > gpurun_out/r2j_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2j_pytest.log
for v in 0 1 0 1; do
  echo "== SRGD_FINAL_FUSION=$v"
  SRGD_FINAL_FUSION=$v python bench.py --steps 5 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_time_share'])"
done
