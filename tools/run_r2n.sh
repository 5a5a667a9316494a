python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "groupnorm or mxfp8 or quant or gn" 2>&1 | tail -3
for lib in lib_old lib_new lib_old lib_new; do
  echo "== $lib bf16"
  SRGD_HIP_LIB=tools/ab/$lib.bin python bench.py --steps 5 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_time_share'])"
done
for lib in lib_old lib_new; do
  echo "== $lib fp8 config5"
  SRGD_HIP_LIB=tools/ab/$lib.bin python bench.py --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 --steps 2 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_time_share'])"
done
