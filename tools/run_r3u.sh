R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SRGD_MX1X1=1
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 $R/bench.py --steps 5 --warmup 0 --no_cpu_baseline --no_profile --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/kt.log 2>&1
python3 $R/tools/rocprof_db_stats.py $(find $O/kt -name "*.db" | head -1) $O/fp8_mx1x1_kernel_stats.csv > $O/fp8_mx1x1_kernel_stats.txt
rm -rf $O/kt
head -24 $O/fp8_mx1x1_kernel_stats.csv
