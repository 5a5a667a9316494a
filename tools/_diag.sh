R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_la; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in r3base default; do
  L=$R/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$R/srgd_amd/libsrgd_hip.so
  export SRGD_HIP_LIB=$L
  rocprofv3 --kernel-trace --stats -d $O/kt_$V -o k -- python3 $R/bench.py --steps 5 --warmup 0 --ddpm_steps 4 --no_cpu_baseline --no_profile > $O/kt_$V.log 2>&1
  python3 $R/tools/rocprof_db_stats.py $(find $O/kt_$V -name "*.db" | head -1) $O/${V}_kernel_stats.csv > $O/${V}_kernel_stats.txt
  rm -rf $O/kt_$V
  echo "== $V"; grep "la1\|la2\|la_" $O/${V}_kernel_stats.csv
done
