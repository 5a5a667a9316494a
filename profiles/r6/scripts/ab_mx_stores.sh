#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# conv3x3_mxfp8: full-line stores through LDS (shipped) against the direct 16-byte stores (variant mxdirect), same box
set -e
out=gpurun_out/mxs; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mxfp8" > $out/pytest_kernels.log 2>&1 || { tail -30 $out/pytest_kernels.log; exit 1; }
tail -2 $out/pytest_kernels.log
python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fp8" > $out/pytest_engine.log 2>&1 || { tail -30 $out/pytest_engine.log; exit 1; }
tail -2 $out/pytest_engine.log
SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_mxdirect.so python tools/bench_conv_fp8.py --iters 5 --out $out/direct.json > $out/direct.log 2>&1
python tools/bench_conv_fp8.py --iters 5 --out $out/lines.json > $out/lines.log 2>&1
SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_mxdirect.so python tools/bench_conv_fp8.py --iters 5 --out $out/direct2.json > $out/direct2.log 2>&1
python tools/bench_conv_fp8.py --iters 5 --out $out/lines2.json > $out/lines2.log 2>&1
for f in direct lines direct2 lines2; do grep -h "mxfp8" $out/$f.log | sed "s/^/$f /" | cut -c1-75; done
