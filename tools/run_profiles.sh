# Profile collection on the GPU box (gpurun): kernel-trace stats and separate PMC passes, as MI355X_MICROARCH.md prescribes
# (FETCH_SIZE and WRITE_SIZE in passes of their own, never combined with sys/hip tracing).
R=$GRAFT_REPO_ROOT; TAG=${1:-r6}; O=$R/gpurun_out/${TAG}prof; mkdir -p $O
MODES=${2:-"bf16 f16x3 fp8 fp8_mixed"}
cd /tmp && export TMPDIR=/tmp
# one step lane: the per-kernel durations of two concurrent lanes overlap (their sum exceeds the wall time) and the launch mix changes
# (125 / 80 tiles per launch becomes 125 / 40 + 40 or 63 + 62 / 40 + 40); bench.py's profiled pass - the `roofline` numbers these
# summaries must agree with - runs with one lane for the same reason
export SRGD_STEP_LANES=1
for MODE in $MODES; do
  EXTRA=""; [ $MODE != bf16 ] && EXTRA="--precision $MODE --ddpm_steps 100 --class_cond_scale 2.0"
  [ $MODE = f16x3 ] && EXTRA="--precision f16x3"       # the parity mode at the headline configuration (50 steps, CFG 1.0)
  STEPS=5; [ $MODE = fp8 ] && STEPS=5
  # the bench line of THIS box and lane setting (HIP-event time per launch of the dominant kernel: must agree with the rocprof average below)
  python3 $R/bench.py --no_cpu_baseline $EXTRA > $O/bench_${MODE}_same_box.json 2>$O/bench_${MODE}_same_box.err
  rocprofv3 --kernel-trace --stats -d $O/kt_$MODE -o k -- python3 $R/bench.py --steps $STEPS --warmup 0 --no_cpu_baseline --no_profile $EXTRA > $O/kt_$MODE.log 2>&1
  python3 $R/tools/rocprof_db_stats.py $(find $O/kt_$MODE -name "*.db" | head -1) $O/${MODE}_kernel_stats.csv > $O/${MODE}_kernel_stats.txt
  rm -rf $O/kt_$MODE
  [ $MODE = fp8_mixed ] && continue        # kernel-trace statistics only: its kernels are the bf16 and fp8 ones
  PMCX="--steps 5 --warmup 0 --no_cpu_baseline --no_profile --ddpm_steps 2"; [ $MODE != bf16 ] && PMCX="$PMCX --precision $MODE"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf_$MODE -o f -- python3 $R/bench.py $PMCX > $O/pf_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw_$MODE -o w -- python3 $R/bench.py $PMCX > $O/pw_$MODE.log 2>&1
  python3 $R/tools/pmc_traffic.py $(find $O/pf_$MODE -name "*.db" | head -1) $(find $O/pw_$MODE -name "*.db" | head -1) $O/pmc_traffic_$MODE.json "bench.py $PMCX (125/80 tiles per launch)" > $O/pmc_traffic_$MODE.txt
  rm -rf $O/pf_$MODE $O/pw_$MODE
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/ps_$MODE -o s -- python3 $R/bench.py $PMCX > $O/ps_$MODE.log 2>&1
  python3 $R/tools/pmc_sq.py $(find $O/ps_$MODE -name "*.db" | head -1) $O/pmc_sq_$MODE.json > $O/pmc_sq_$MODE.txt
  rm -rf $O/ps_$MODE
done
SRGD_GRAPHS=0 rocprofv3 --kernel-trace --stats -d $O/kt_nograph -o k -- python3 $R/bench.py --steps 5 --warmup 0 --no_cpu_baseline --no_profile > $O/kt_nograph.log 2>&1
python3 $R/tools/rocprof_db_stats.py $(find $O/kt_nograph -name "*.db" | head -1) $O/bf16_nograph_kernel_stats.csv > $O/bf16_nograph_kernel_stats.txt
rm -rf $O/kt_nograph
grep -h copyBuffer $O/bf16_kernel_stats.csv $O/bf16_nograph_kernel_stats.csv
cd $R; for M in $MODES; do head -16 $O/${M}_kernel_stats.csv; done; cat $O/pmc_traffic_*.txt; cat $O/pmc_sq_*.txt | head -80
