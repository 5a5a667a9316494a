# round 3, GPU job v: pointwise epilogue with the tail-operand loads hoisted (bf16 and MX kernels) vs HEAD
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q) > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -4 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline"
$B > $O/bench_new.json 2>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_head3.so $B > $O/bench_old.json 2>>$O/err.log &&
$B > $O/bench_new2.json 2>>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_head3.so $B > $O/bench_old2.json 2>>$O/err.log &&
$B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_off.json 2>>$O/err.log &&
SRGD_MX1X1=1 $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_mx128.json 2>>$O/err.log &&
SRGD_MX1X1=1 SRGD_MX1X1_BM=256 $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_mx256.json 2>>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_head3.so $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_old.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_bf16','conv3x3_mxfp8','conv1x1_bf16','conv1x1_mxfp8','groupnorm_silu')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
