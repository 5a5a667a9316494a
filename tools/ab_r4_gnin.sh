# Round-4 A/B of the GroupNorm-in-staging (GNIN) instances of conv3x3_bf16 on one box: per-shape micro-benchmark + whole bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_gnin; mkdir -p $O; cd $R
for V in r3base gnin_pk gnin_scalar_nolean default; do
  L=$R/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$R/srgd_amd/libsrgd_hip.so
  echo "== $V" >> $O/bench_conv_gnin.txt
  SRGD_HIP_LIB=$L python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @256" --impls 5 >> $O/bench_conv_gnin.txt 2>&1 || exit 1
  SRGD_HIP_LIB=$L python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @256" --impls 2 >> $O/bench_conv_gnin.txt 2>&1 || exit 1
  SRGD_HIP_LIB=$L python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @128" --impls 5 >> $O/bench_conv_gnin.txt 2>&1 || exit 1
done
cat $O/bench_conv_gnin.txt
python -m pytest tests/test_engine_gpu.py -x -q -k "groupnorm_fused or unet_eps" > $O/pytest_gnin.log 2>&1 || { tail -30 $O/pytest_gnin.log; exit 1; }
tail -3 $O/pytest_gnin.log
for V in r3base default r3base default; do
  L=$R/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$R/srgd_amd/libsrgd_hip.so
  SRGD_HIP_LIB=$L python bench.py --no_cpu_baseline > $O/bench_$V.json 2>$O/bench_$V.err || { tail $O/bench_$V.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1]); print("$V", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("kernel_time_share"))
PY
done
