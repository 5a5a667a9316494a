cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_diag; mkdir -p $O
V=$PWD/srgd_amd/variants
for R in 1 2; do
  for L in base reads1 reads2; do
    SRGD_HIP_LIB=$V/libsrgd_hip_$L.so python tools/bench_conv.py --shapes big --batch 125 --iters 10 --impls 2 > $O/${L}_$R.txt 2>&1 || { tail $O/${L}_$R.txt; exit 1; }
  done
  python tools/bench_conv.py --shapes big --batch 125 --iters 10 --impls 2 > $O/new_$R.txt 2>&1 || { tail $O/new_$R.txt; exit 1; }
done
for R in 1 2; do paste -d'|' $O/base_$R.txt $O/new_$R.txt $O/reads1_$R.txt $O/reads2_$R.txt | sed 's/3x3 //g; s/{2: //g; s/}//g' | cut -c1-150; done
