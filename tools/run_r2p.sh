for v in 0 1; do
  echo "== kernel tests SRGD_LA128_TM32=$v"; SRGD_LA128_TM32=$v python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_attention" 2>&1 | tail -2
done
for v in 0 1 0 1; do
  echo "== SRGD_LA128_TM32=$v"
  SRGD_LA128_TM32=$v python bench.py --steps 5 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_time_share'])"
done
