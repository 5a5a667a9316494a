cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "quant or fused_twins or fp8_mode_uses" 2>&1 | tail -2
O=gpurun_out/r4_gn; mkdir -p $O
for H in 0 1 0 1; do
  SRGD_GN_HOIST=$H python bench.py --no_cpu_baseline --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_hoist$H.json 2>$O/err.txt || { tail $O/err.txt; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_fp8_hoist$H.json").read().strip().splitlines()[-1]); print("fp8 HOIST $H", round(d["value"],4), d["kernel_time_share"].get("quantize_mxfp8"), d["kernel_time_share"]["groupnorm_silu"])
PY
done
