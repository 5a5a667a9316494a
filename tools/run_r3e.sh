# round 3, GPU job e: full GPU suite on the new defaults (GNIN for one-tile layers, tied-MFMA MX kernel), the default bench line with its
# cpu_baseline leg, the MX kernel's LDS diagnostic, then the profile collection of tools/run_profiles.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
(time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
tail -4 $O/pytest_gpu.log
grep -q "rc=0" $O/pytest_gpu.log || exit 1
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err || exit 1
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('default bench', round(d['value'],4), 'roofline', round(d['roofline']['achieved']), round(d['roofline']['frac'],3), 'cpu', d.get('cpu_baseline',{}).get('value'))
print(json.dumps(d['kernel_time_share'])); print(json.dumps(d.get('hbm_kernels')))
PY
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_nolds.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_nolds.json > $O/conv_fp8_nolds.txt 2>&1
timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8.json > $O/conv_fp8.txt 2>&1
paste <(cut -c1-60 $O/conv_fp8.txt) <(awk '{print $4}' $O/conv_fp8_nolds.txt)
bash tools/run_profiles.sh r3 || exit 1
