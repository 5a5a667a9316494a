cd $GRAFT_REPO_ROOT
python tools/check_gnin.py 2>&1 | grep -v amdgpu.ids || exit 1
python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @256" --impls 5 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @256" --impls 2 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py --batch 125 --iters 20 --only "3x3 128->128 @128" --impls 5 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_engine_gpu.py -x -q -k "groupnorm_fused or unet_eps" 2>&1 | tail -3
O=gpurun_out/r4_gnin; mkdir -p $O
for V in r3base default r3base default; do
  L=$PWD/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$PWD/srgd_amd/libsrgd_hip.so
  SRGD_HIP_LIB=$L python bench.py --no_cpu_baseline > $O/bench_$V.json 2>$O/bench_$V.err || { tail $O/bench_$V.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1]); print("$V", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("kernel_time_share"))
PY
done
