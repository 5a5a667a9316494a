# Round 5: register-direct epilogue of conv3x3_bf16 against the round-4 kernel (srgd_amd/variants/libsrgd_hip_base.so built
# from a worktree of a9bfeba: `git worktree add /tmp/base a9bfeba; SRGD_CSRC=/tmp/base/srgd_amd/csrc python tools/build_variant.py base`;
# the round-4 library takes its stamps from the environment, SRGD_CONV3_STAMPS=1), one box: kernel tests, two alternating rounds of
# tools/bench_conv.py, phase stamps.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_regepi; mkdir -p $O
V=$PWD/srgd_amd/variants
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "rc=$?" >> $O/pytest_kernels.log; tail -5 $O/pytest_kernels.log
grep -q "rc=0" $O/pytest_kernels.log || exit 1
python tools/check_gnin.py > $O/check_gnin.txt 2>&1; echo "rc=$?" >> $O/check_gnin.txt; cat $O/check_gnin.txt
python tools/bench_conv.py --shapes big --batch 8 --iters 2 --impls 1,2 > $O/diff_vs_generic.txt 2>&1; cut -c1-40,100-200 $O/diff_vs_generic.txt
for R in 1 2; do
  SRGD_HIP_LIB=$V/libsrgd_hip_base.so python tools/bench_conv.py --shapes big --batch 125 --iters 10 --impls 2 > $O/base_$R.txt 2>&1 || { tail $O/base_$R.txt; exit 1; }
  python tools/bench_conv.py --shapes big --batch 125 --iters 10 --impls 2 > $O/new_$R.txt 2>&1 || { tail $O/new_$R.txt; exit 1; }
done
paste -d'|' $O/base_2.txt $O/new_2.txt | cut -c1-200
for S in "128->128 @256" "128+128->128 @256" "128->128 @128" "1024->1024 @32"; do
  SRGD_CONV3_STAMPS=1 SRGD_HIP_LIB=$V/libsrgd_hip_base.so python tools/bench_conv.py --only "3x3 $S" --batch 125 --iters 3 --impls 2 2>&1 | grep stamps | tail -2 >> $O/stamps_base.txt
  SRGD_HIP_LIB=$V/libsrgd_hip_stamps.so python tools/bench_conv.py --only "3x3 $S" --batch 125 --iters 3 --impls 2 2>&1 | grep stamps | tail -1 >> $O/stamps_new.txt
done
cat $O/stamps_base.txt $O/stamps_new.txt
