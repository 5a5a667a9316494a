R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
for v in mfmaonly d_nob d_nodma d_pure; do
  SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_$v.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_$v.json > $O/conv_fp8_$v.txt 2>&1 || exit 1
done
paste <(cut -c1-58 $O/conv_fp8_mfmaonly.txt) <(cut -c30-58 $O/conv_fp8_d_nob.txt) <(cut -c30-58 $O/conv_fp8_d_nodma.txt) <(cut -c30-58 $O/conv_fp8_d_pure.txt) | grep -v amdgpu
