# round 3, GPU job l: pointwise kernels with the incremental issue stream (no runtime divisions / per-lane multiplies per K-step) vs the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "conv1x1 or mx or fp8 or shuffle or downsample or two_sources or production or kernel_chains or headline or config5") > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -4 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 3 > $O/conv1x1_bf16_new.txt 2>&1 &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3head.so timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 3 > $O/conv1x1_bf16_old.txt 2>&1 &&
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 4 > $O/conv1x1_mxfp8_new.txt 2>&1 &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3head.so timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 4 > $O/conv1x1_mxfp8_old.txt 2>&1
paste <(grep 1x1 $O/conv1x1_bf16_old.txt) <(grep 1x1 $O/conv1x1_bf16_new.txt | sed 's/.*{/{/') <(grep 1x1 $O/conv1x1_mxfp8_old.txt | sed 's/.*{/{/') <(grep 1x1 $O/conv1x1_mxfp8_new.txt | sed 's/.*{/{/')
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline"
$B > $O/bench_bf16_new.json 2>$O/err.log &&
SRGD_GN_FUSION=0 $B > $O/bench_bf16_new_nognin.json 2>>$O/err.log &&
SRGD_GN_FUSION_NTILES=2 $B > $O/bench_bf16_new_gnin2.json 2>>$O/err.log &&
$B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_new.json 2>>$O/err.log &&
SRGD_MX1X1=0 $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_new_nomx1x1.json 2>>$O/err.log &&
SRGD_MX1X1_MIN_CIN=768 $B --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_fp8_new_mx768.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_bf16','conv3x3_mxfp8','conv1x1_bf16','conv1x1_mxfp8','groupnorm_silu','linear_attention')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
