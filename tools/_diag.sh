cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_wide; mkdir -p $O
for K in 0 768 0 768 512; do
  SRGD_CONV1X1_WIDE_MIN_K=$K python bench.py --no_cpu_baseline > $O/bench_wide$K.json 2>$O/err.txt || { tail $O/err.txt; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_wide$K.json").read().strip().splitlines()[-1]); print("MIN_K $K", round(d["value"],4), d["kernel_time_share"]["conv1x1_bf16"], d["hbm_kernels"]["conv1x1_bf16"]["avg_launch_us"])
PY
done
python -m pytest tests/test_kernels_gpu.py -x -q -k "pointwise or conv1x1 or streaming" 2>&1 | tail -2
