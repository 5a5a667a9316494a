# Round 5: register-direct epilogue of conv3x3_mxfp8 against the round-4 kernel (variants/libsrgd_hip_base.so), one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_mx; mkdir -p $O
V=$PWD/srgd_amd/variants
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "rc=$?" >> $O/pytest_kernels.log; tail -5 $O/pytest_kernels.log
grep -q "rc=0" $O/pytest_kernels.log || exit 1
for R in 1 2; do
  SRGD_HIP_LIB=$V/libsrgd_hip_base.so python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/base_$R.json > $O/base_$R.txt 2>&1 || { tail $O/base_$R.txt; exit 1; }
  python tools/bench_conv_fp8.py --batch 125 --iters 10 --out $O/new_$R.json > $O/new_$R.txt 2>&1 || { tail $O/new_$R.txt; exit 1; }
done
for R in 1 2; do paste -d'|' $O/base_$R.txt $O/new_$R.txt | grep -v amdgpu.ids | cut -c1-190; done
