# round 3, GPU job b: full GPU suite, conv3x3 wave-stagger variants, GroupNorm-in-staging A/B on the spill-free 32x32x16 instance
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
(time python -m pytest tests -m gpu -x -q --durations=12) > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
python tools/bench_conv.py --batch 125 --iters 10 --shapes big --impls 0 > $O/conv_base.txt 2>&1
for v in ws1 ws2 ws4; do
  SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_$v.so python tools/bench_conv.py --batch 125 --iters 10 --shapes big --impls 0 > $O/conv_$v.txt 2>&1
done
python tools/bench_conv.py --batch 125 --iters 10 --shapes big --impls 0 > $O/conv_base2.txt 2>&1
B="python bench.py --steps 5 --warmup 5 --no_cpu_baseline"
$B > $O/bench_m16_base.json 2>/dev/null
SRGD_CONV3_M16=0 $B > $O/bench_m32_base.json 2>/dev/null
SRGD_CONV3_M16=0 SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=1 $B > $O/bench_m32_gnin1.json 2>/dev/null
SRGD_CONV3_M16=0 SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=2 $B > $O/bench_m32_gnin2.json 2>/dev/null
SRGD_CONV3_M16=0 SRGD_GN_FUSION=1 $B > $O/bench_m32_gninall.json 2>/dev/null
SRGD_GN_FUSION=1 SRGD_GN_FUSION_NTILES=1 $B > $O/bench_m16_gnin1.json 2>/dev/null
$B > $O/bench_m16_base2.json 2>/dev/null
tail -4 $O/pytest_gpu.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), 'conv3', k.get('conv3x3_bf16'), 'gn', k.get('groupnorm_silu'), 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
