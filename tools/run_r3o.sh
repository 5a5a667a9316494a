# round 3, GPU job o: MX 3x3 kernel with 8 waves per workgroup (64 x 64 wave tiles, accumulators in AGPRs, four waves per SIMD) vs the 4-wave shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "mx or fp8 or config5") > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -4 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_w8.json > $O/conv_fp8_w8.txt 2>&1 &&
SRGD_MXFP8_WAVES=4 timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_w4.json > $O/conv_fp8_w4.txt 2>&1
paste <(cut -c1-58 $O/conv_fp8_w4.txt) <(cut -c30-80 $O/conv_fp8_w8.txt) | grep -v amdgpu
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline --ddpm_steps 100 --class_cond_scale 2.0"
$B --precision fp8 > $O/bench_fp8_w8.json 2>$O/err.log &&
SRGD_MXFP8_WAVES=4 $B --precision fp8 > $O/bench_fp8_w4.json 2>>$O/err.log &&
$B --precision fp8_mixed > $O/bench_fp8mixed_w8.json 2>>$O/err.log &&
SRGD_MXFP8_WAVES=4 $B --precision fp8_mixed > $O/bench_fp8mixed_w4.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_bf16','conv3x3_mxfp8','conv1x1_bf16','groupnorm_silu','linear_attention')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
