cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_gnin; mkdir -p $O
for N in 1 2 99 0 1 2; do
  if [ $N = 0 ]; then export SRGD_GN_FUSION=0; unset SRGD_GN_FUSION_NTILES; else unset SRGD_GN_FUSION; export SRGD_GN_FUSION_NTILES=$N; fi
  python bench.py --no_cpu_baseline > $O/bench_ntiles$N.json 2>$O/bench_ntiles$N.err || { tail $O/bench_ntiles$N.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_ntiles$N.json").read().strip().splitlines()[-1]); print("ntiles $N", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("kernel_time_share"))
PY
done
