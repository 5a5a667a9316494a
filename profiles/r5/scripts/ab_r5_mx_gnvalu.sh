# Pricing GroupNorm1-apply + SiLU + MX quantisation inside conv3x3_mxfp8's staging: the production kernel against a build that
# issues the transform's vector work (on dummy registers) in every tap, on the one-n-tile layers; next to what the separate pass costs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_mx_gnvalu; mkdir -p $O; V=$PWD/srgd_amd/variants
for R in 1 2; do for S in "128->128 @256" "128+128->128 @256" "128->128 @128"; do
  python tools/bench_conv_fp8.py --batch 125 --iters 10 --only "3x3 $S" --out $O/x.json 2>&1 | grep "3x3" | sed "s/^/production  : /" >> $O/table.txt
  SRGD_HIP_LIB=$V/libsrgd_hip_gnvalu.so python tools/bench_conv_fp8.py --batch 125 --iters 10 --only "3x3 $S" --out $O/x.json 2>&1 | grep "3x3" | sed "s/^/+ transform  : /" >> $O/table.txt
done; done
cat $O/table.txt
