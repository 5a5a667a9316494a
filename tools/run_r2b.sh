set -x
O=gpurun_out/r2b; mkdir -p $O
./tools/probe_mxfp8.bin > $O/probe.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_attention or fused" > $O/pytest_la.log 2>&1; echo "rc=$?" >> $O/pytest_la.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "lockstep or hipgraph or config2_geometry or dim128_config1 or growth" > $O/pytest_eng.log 2>&1; echo "rc=$?" >> $O/pytest_eng.log
python bench.py --steps 5 --warmup 5 --no_cpu_baseline > $O/bench_la2.json 2> $O/bench_la2.err
SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=1 python bench.py --steps 5 --warmup 5 --no_cpu_baseline > $O/bench_gnin1.json 2> $O/bench_gnin1.err
SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=2 python bench.py --steps 5 --warmup 5 --no_cpu_baseline > $O/bench_gnin2.json 2> $O/bench_gnin2.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o la2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 0 --no_cpu_baseline --no_profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O/prof | head; DB=$(find $O/prof -name "*.db" | head -1); python tools/rocprof_db_stats.py $DB $O/la2_kernel_stats.csv; rm -rf $O/prof
tail -3 $O/pytest_la.log $O/pytest_eng.log; cat $O/probe.log
for f in la2 gnin1 gnin2; do python - <<PY
import json
d=json.loads([l for l in open("$O/bench_$f.json") if l.startswith("{")][0])
print("$f", d["value"], d["kernel_time_share"], d["roofline"]["achieved"])
PY
done
head -30 $O/la2_kernel_stats.csv
