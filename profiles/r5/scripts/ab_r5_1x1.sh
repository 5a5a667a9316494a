# Round 5: register-direct epilogue of the pointwise kernels against the round-4 library, one box: full GPU suite, per-shape bench, whole bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_1x1; mkdir -p $O
V=$PWD/srgd_amd/variants
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
grep -q "rc=0" $O/pytest_gpu.log || exit 1
for R in 1 2; do
  SRGD_HIP_LIB=$V/libsrgd_hip_base.so python tools/bench_conv.py --only 1x1 --batch 125 --iters 10 --impls 1,3 > $O/base_$R.txt 2>&1 || { tail $O/base_$R.txt; exit 1; }
  python tools/bench_conv.py --only 1x1 --batch 125 --iters 10 --impls 1,3 > $O/new_$R.txt 2>&1 || { tail $O/new_$R.txt; exit 1; }
done
for R in 1 2; do paste -d'|' <(cut -c1-24,44-100 $O/base_$R.txt) <(cut -c44-150 $O/new_$R.txt) | grep -v amdgpu.ids; done
bash tools/ab_bench.sh r5_1x1 base default
