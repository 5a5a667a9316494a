#!/bin/bash
[ -n "$GRAFT_REPO_ROOT" ] || { echo "archived GPU-box script (see README.md next to it)"; exit 2; }
# Prices a persistent form of conv3x3_mxfp8 before building it (GPU box): diagnostic variants of the shipped kernel
# (tools/build_variant.py mxdN -DSRGD_MXFP8_DIAG=N; 1 = the prologue does not wait for the halo patch, 2 = no epilogue stores, 3 = both)
set -e
out=gpurun_out/mxp; mkdir -p $out
python tools/bench_conv_fp8.py --iters 5 --out $out/base.json > $out/base.log 2>&1
for d in 1 2 3; do
  SRGD_HIP_LIB=$PWD/srgd_amd/variants/libsrgd_hip_mxd$d.so python tools/bench_conv_fp8.py --iters 5 --out $out/diag$d.json > $out/diag$d.log 2>&1
done
grep -h "mxfp8" $out/base.log | sed 's/^/base  /'; for d in 1 2 3; do grep -h "mxfp8" $out/diag$d.log | sed "s/^/diag$d /"; done
