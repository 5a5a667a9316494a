#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# conv3x3_split: full-line stores through LDS (shipped) against the direct 16-byte stores (variant c3sdirect), same box
set -e
out=gpurun_out/c3ss; mkdir -p $out
V=$PWD/srgd_amd/variants/libsrgd_hip_c3sdirect.so
python -m pytest tests/test_split_gpu.py -m gpu -x -q > $out/pytest_split.log 2>&1 || { tail -30 $out/pytest_split.log; exit 1; }
tail -2 $out/pytest_split.log
for i in 1 2; do
  SRGD_HIP_LIB=$V python bench.py --no_cpu_baseline --no_profile --precision f16x3 > $out/bench_direct_$i.json 2>$out/err.txt || { tail $out/err.txt; exit 1; }
  python bench.py --no_cpu_baseline --no_profile --precision f16x3 > $out/bench_lines_$i.json 2>$out/err.txt || { tail $out/err.txt; exit 1; }
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"],4), round(d["ms_per_step"],1))
PY
