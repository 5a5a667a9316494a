cd $GRAFT_REPO_ROOT
for V in default mxnopro default mxnopro; do
  L=$PWD/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$PWD/srgd_amd/libsrgd_hip.so
  echo "== $V"
  SRGD_HIP_LIB=$L python tools/bench_conv_fp8.py --batch 125 --iters 20 2>&1 | grep -v amdgpu.ids | grep TF | cut -c1-60
done
