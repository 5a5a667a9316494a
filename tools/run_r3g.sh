# round 3, GPU job g: what bounds conv3x3_mxfp8 (timing-only diagnostic builds), MX pointwise layers restricted to the K-heavy ones
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
for v in nobar mfmaonly; do
  SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_$v.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_$v.json > $O/conv_fp8_$v.txt 2>&1 || exit 1
done
timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8.json > $O/conv_fp8.txt 2>&1
paste <(cut -c1-58 $O/conv_fp8.txt) <(cut -c30-58 $O/conv_fp8_nobar.txt) <(cut -c30-58 $O/conv_fp8_mfmaonly.txt) | grep -v amdgpu
B="timeout -k 10 400 python bench.py --steps 5 --warmup 5 --no_cpu_baseline --ddpm_steps 100 --class_cond_scale 2.0 --precision fp8"
SRGD_MX1X1=0 $B > $O/bench_fp8_nomx1x1.json 2>$O/err.log &&
SRGD_MX1X1_MIN_CIN=768 $B > $O/bench_fp8_mx1x1_768.json 2>>$O/err.log &&
SRGD_MX1X1_MIN_CIN=512 $B > $O/bench_fp8_mx1x1_512.json 2>>$O/err.log &&
$B > $O/bench_fp8_mx1x1_all.json 2>>$O/err.log &&
SRGD_MX1X1=0 $B > $O/bench_fp8_nomx1x1_2.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_time_share']
    print(sys.argv[1].split('/')[-1], round(d['value'],4), {a:k.get(a) for a in ('conv3x3_mxfp8','conv1x1_bf16','conv1x1_mxfp8','groupnorm_silu','linear_attention')}, 'TF', round(d['roofline']['achieved']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
