# round 3, GPU job t: MX pointwise kernel with 128-pixel tiles (two workgroups per CU) vs 256-pixel tiles vs conv1x1_bf16
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
(time timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv1x1") > $O/pytest_bm128.log 2>&1; echo "rc=$?" >> $O/pytest_bm128.log
tail -3 $O/pytest_bm128.log; grep -q "rc=0" $O/pytest_bm128.log || exit 1
(time SRGD_MX1X1_BM=256 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv1x1") > $O/pytest_bm256.log 2>&1; echo "rc=$?" >> $O/pytest_bm256.log
tail -3 $O/pytest_bm256.log; grep -q "rc=0" $O/pytest_bm256.log || exit 1
(time timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "mx or fp8") > $O/pytest_eng.log 2>&1; echo "rc=$?" >> $O/pytest_eng.log
tail -3 $O/pytest_eng.log; grep -q "rc=0" $O/pytest_eng.log || exit 1
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 3 > $O/conv1x1_bf16.txt 2>&1 &&
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 4 > $O/conv1x1_mx128.txt 2>&1 &&
SRGD_MX1X1_BM=256 timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 4 > $O/conv1x1_mx256.txt 2>&1
paste <(grep 1x1 $O/conv1x1_bf16.txt) <(grep 1x1 $O/conv1x1_mx128.txt | sed 's/.*{/{/') <(grep 1x1 $O/conv1x1_mx256.txt | sed 's/.*{/{/')
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline --ddpm_steps 100 --class_cond_scale 2.0 --precision fp8"
$B > $O/bench_fp8_off.json 2>$O/err.log &&
SRGD_MX1X1=1 $B > $O/bench_fp8_mx128.json 2>>$O/err.log &&
SRGD_MX1X1=1 SRGD_MX1X1_BM=256 $B > $O/bench_fp8_mx256.json 2>>$O/err.log &&
SRGD_MX1X1=1 SRGD_MX1X1_MIN_CIN=768 $B > $O/bench_fp8_mx128_768.json 2>>$O/err.log &&
$B > $O/bench_fp8_off2.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_mxfp8','conv1x1_bf16','conv1x1_mxfp8','groupnorm_silu','linear_attention')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
