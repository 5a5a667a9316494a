R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/ps -o s -- python3 $R/bench.py --precision fp32 --steps 1 --warmup 0 --images 1 --ddpm_steps 2 --no_cpu_baseline --no_profile > $O/ps.log 2>&1
python3 $R/tools/pmc_sq.py $(find $O/ps -name "*.db" | head -1) $O/pmc_sq_fp32.json > $O/pmc_sq_fp32.txt
rm -rf $O/ps
cat $O/pmc_sq_fp32.txt
