# round 3, GPU job d: world-8 thread tests + MX-fp8 kernel tests first, then the A/B runs of job c (the suite itself passed 156/157 there)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_world8_partitioning_gpu.py tests/test_kernels_gpu.py -m gpu -x -q --durations=8) > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -4 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_pipe.json > $O/conv_fp8_pipe.txt 2>&1 &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3tied.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_tied.json > $O/conv_fp8_tied.txt 2>&1 &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3base.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_base.json > $O/conv_fp8_base.txt 2>&1 &&
SRGD_CONV3_STAGGER=0 timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_pipe_nostagger.json > $O/conv_fp8_pipe_nostagger.txt 2>&1
python - $O <<'PY'
import json,sys
O=sys.argv[1]
L=lambda n:{r['shape']:r for r in json.load(open(f"{O}/{n}.json"))}
try:
    a,t,b,c=L('conv_fp8_base'),L('conv_fp8_tied'),L('conv_fp8_pipe'),L('conv_fp8_pipe_nostagger')
    for k in b: print(f"{k:28s} r2 {a[k]['mxfp8_tflops']:7.1f}  tied {t[k]['mxfp8_tflops']:7.1f}  pipe {b[k]['mxfp8_tflops']:7.1f}  pipe/nostagger {c[k]['mxfp8_tflops']:7.1f}  bf16 {b[k]['bf16_tflops']:7.1f} nostag {c[k]['bf16_tflops']:7.1f}")
except Exception as e: print('fp8 table ERR', e)
PY
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline"
$B > $O/bench_m16_base.json 2>$O/err.log &&
SRGD_CONV3_STAGGER=0 $B > $O/bench_m16_nostagger.json 2>>$O/err.log &&
SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=1 $B > $O/bench_m16_gnin1.json 2>>$O/err.log &&
SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=2 $B > $O/bench_m16_gnin2.json 2>>$O/err.log &&
SRGD_GN_FUSION=1 $B > $O/bench_m16_gninall.json 2>>$O/err.log &&
SRGD_CONV3_M16=0 $B > $O/bench_m32_base.json 2>>$O/err.log &&
SRGD_CONV3_M16=0 SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=1 $B > $O/bench_m32_gnin1.json 2>>$O/err.log &&
SRGD_CONV3_M16=0 SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=2 $B > $O/bench_m32_gnin2.json 2>>$O/err.log &&
$B > $O/bench_m16_base2.json 2>>$O/err.log &&
$B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_config5.json 2>>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3base.so $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_config5_base.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), 'conv3', k.get('conv3x3_bf16'), k.get('conv3x3_mxfp8'), 'gn', k.get('groupnorm_silu'), 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt_bf16 -o k -- python3 $R/bench.py --steps 5 --warmup 0 --no_cpu_baseline --no_profile > $O/kt_bf16.log 2>&1
python3 $R/tools/rocprof_db_stats.py $(find $O/kt_bf16 -name "*.db" | head -1) $O/bf16_kernel_stats.csv > $O/bf16_kernel_stats.txt
rm -rf $O/kt_bf16
head -30 $O/bf16_kernel_stats.txt
