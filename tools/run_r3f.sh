# round 3, GPU job f: MX-fp8 pointwise kernel (tests, per-shape rate next to conv1x1_bf16, configs[4] benches with SRGD_MX1X1 on / off),
# DMA-latency diagnostic of the MX 3x3 kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q --durations=8 -k "mx or fp8 or conv1x1 or config5") > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -6 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 3 > $O/conv1x1_bf16.txt 2>&1 &&
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --only 1x1 --impls 4 > $O/conv1x1_mxfp8.txt 2>&1
paste <(grep 1x1 $O/conv1x1_bf16.txt) <(grep 1x1 $O/conv1x1_mxfp8.txt | sed 's/.*{/{/')
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_latewait.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_latewait.json > $O/conv_fp8_latewait.txt 2>&1
timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8.json > $O/conv_fp8.txt 2>&1
paste <(cut -c1-58 $O/conv_fp8.txt) <(cut -c30-58 $O/conv_fp8_latewait.txt) | grep -v amdgpu
B="timeout -k 10 400 python bench.py --steps 5 --warmup 5 --no_cpu_baseline --ddpm_steps 100 --class_cond_scale 2.0"
$B --precision fp8 > $O/bench_fp8_mx1x1.json 2>$O/err.log &&
SRGD_MX1X1=0 $B --precision fp8 > $O/bench_fp8_nomx1x1.json 2>>$O/err.log &&
$B --precision fp8_mixed > $O/bench_fp8mixed_mx1x1.json 2>>$O/err.log &&
SRGD_MX1X1=0 $B --precision fp8_mixed > $O/bench_fp8mixed_nomx1x1.json 2>>$O/err.log &&
$B --precision bf16 > $O/bench_bf16_config5.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_bf16','conv3x3_mxfp8','conv1x1_bf16','conv1x1_mxfp8','groupnorm_silu','linear_attention','quantize_mxfp8')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
