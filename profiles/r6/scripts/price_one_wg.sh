#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# 3x3 kernels with ONE workgroup per CU (LDS padded) against the shipped two: can one workgroup alone feed the matrix pipe?
set -e
out=gpurun_out/onewg; mkdir -p $out
V=$PWD/srgd_amd/variants/libsrgd_hip_onewg.so
python tools/bench_conv.py --impls 1,0 --only 3x3 --iters 5 --stats 1 > $out/bf16_two.log 2>&1
SRGD_HIP_LIB=$V python tools/bench_conv.py --impls 1,0 --only 3x3 --iters 5 --stats 1 > $out/bf16_one.log 2>&1
python tools/bench_conv_fp8.py --iters 5 --out $out/fp8_two.json > $out/fp8_two.log 2>&1
SRGD_HIP_LIB=$V python tools/bench_conv_fp8.py --iters 5 --out $out/fp8_one.json > $out/fp8_one.log 2>&1
for f in bf16_two bf16_one; do grep -h "impl" $out/$f.log | sed "s/^/$f /" | sed 's/|diff.*//' | sed 's/impl 1 *[0-9.]* TF  //'; done
for f in fp8_two fp8_one; do grep -h "mxfp8" $out/$f.log | sed "s/^/$f /" | cut -c1-62; done
