# One gpurun job (about 9 minutes of box time): full GPU suite on the final defaults, default bench with its cpu_baseline leg, then tools/run_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
(time timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
tail -4 $O/pytest_gpu.log
grep -q "rc=0" $O/pytest_gpu.log || exit 1
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err || exit 1
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('default bench', round(d['value'],4), 'roofline', round(d['roofline']['achieved']), round(d['roofline']['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'))
print(json.dumps(d['kernel_time_share']))
PY
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
bash tools/run_profiles.sh r6 "${PROFILE_MODES:-bf16 f16x3}" > $O/profiles.log 2>&1 || exit 1
tail -40 $O/profiles.log
