# round 3, GPU job j: bf16 conv3x3 with the lean K loop (3 VALU per fragment address, scalar DMA offsets) vs the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
(time timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "conv3x3 or conv or groupnorm or hipgraph or headline or config2 or repeat") > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -4 $O/pytest_sel.log
grep -q "rc=0" $O/pytest_sel.log || exit 1
timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --shapes big --impls 0 > $O/conv_new.txt 2>&1 &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3head.so timeout -k 10 300 python tools/bench_conv.py --batch 125 --iters 10 --shapes big --impls 0 > $O/conv_old.txt 2>&1
paste <(grep 3x3 $O/conv_old.txt) <(grep 3x3 $O/conv_new.txt | sed 's/.*{/{/')
B="timeout -k 10 300 python bench.py --steps 5 --warmup 5 --no_cpu_baseline"
$B > $O/bench_new.json 2>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3head.so $B > $O/bench_old.json 2>>$O/err.log &&
$B > $O/bench_new2.json 2>>$O/err.log &&
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_r3head.so $B > $O/bench_old2.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_bf16','conv1x1_bf16','groupnorm_silu','linear_attention')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
