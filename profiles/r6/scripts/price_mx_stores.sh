#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# conv3x3_mxfp8: what do the epilogue's stores cost, and is it the requests or the HBM write stream?  (diagnostic variants, GPU box)
set -e
out=gpurun_out/mxw; mkdir -p $out
python tools/bench_conv_fp8.py --iters 5 --out $out/base.json > $out/base.log 2>&1
SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_mxd2.so python tools/bench_conv_fp8.py --iters 5 --out $out/nostores.json > $out/nostores.log 2>&1
SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_mxd4.so python tools/bench_conv_fp8.py --iters 5 --out $out/l2window.json > $out/l2window.log 2>&1
for f in base nostores l2window; do grep -h "mxfp8" $out/$f.log | sed "s/^/$f /" | cut -c1-75; done
