#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# conv3x3_bf16 / conv3x3_mxfp8: full-line stores through LDS (shipped) against the direct 16-byte stores (variant c3direct), same box
set -e
out=gpurun_out/c3s; mkdir -p $out
V=$PWD/srgd_amd/variants/libsrgd_hip_c3direct.so
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv3x3 or conv2d or fast" > $out/pytest_kernels.log 2>&1 || { tail -30 $out/pytest_kernels.log; exit 1; }
tail -2 $out/pytest_kernels.log
SRGD_HIP_LIB=$V python tools/bench_conv.py --impls 1,0 --only 3x3 --iters 5 --stats 1 > $out/direct.log 2>&1
python tools/bench_conv.py --impls 1,0 --only 3x3 --iters 5 --stats 1 > $out/lines.log 2>&1
for f in direct lines; do grep -h "impl" $out/$f.log | sed "s/^/$f /" | sed 's/|diff.*//'; done
for i in 1 2; do
  SRGD_HIP_LIB=$V python bench.py --no_cpu_baseline --no_profile > $out/bench_direct_$i.json 2>$out/err.txt || { tail $out/err.txt; exit 1; }
  python bench.py --no_cpu_baseline --no_profile > $out/bench_lines_$i.json 2>$out/err.txt || { tail $out/err.txt; exit 1; }
done
SRGD_HIP_LIB=$V python bench.py --no_cpu_baseline --no_profile --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $out/bench_fp8_direct.json 2>$out/err.txt
python bench.py --no_cpu_baseline --no_profile --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $out/bench_fp8_lines.json 2>$out/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"],4), round(d["ms_per_step"],1))
PY
