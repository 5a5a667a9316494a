# kernel-trace statistics of the default bench (bf16) after the C = 256 LinearAttention fusion
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 $R/bench.py --steps 5 --warmup 0 --no_cpu_baseline --no_profile > $O/kt.log 2>&1
python3 $R/tools/rocprof_db_stats.py $(find $O/kt -name "*.db" | head -1) $O/bf16_kernel_stats.csv > $O/bf16_kernel_stats.txt
rm -rf $O/kt
head -24 $O/bf16_kernel_stats.csv
