R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
SRGD_MXFP8_WAVES=4 SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_d_noapatch.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_noapatch_w4.json > $O/conv_fp8_noapatch_w4.txt 2>&1
SRGD_MXFP8_WAVES=4 timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_w4.json > $O/conv_fp8_w4.txt 2>&1
SRGD_HIP_LIB=$R/srgd_amd/variants/libsrgd_hip_d_noapatch.so timeout -k 10 200 python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/conv_fp8_noapatch_w8.json > $O/conv_fp8_noapatch_w8.txt 2>&1
paste <(cut -c1-58 $O/conv_fp8_w4.txt) <(cut -c30-58 $O/conv_fp8_noapatch_w4.txt) <(cut -c30-58 $O/conv_fp8_noapatch_w8.txt) | grep -v amdgpu
