R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R
rm -f gpurun_out/parity_report.jsonl
SRGD_MX1X1=1 timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "config5 or config2 or fp8_unet_forward" > $O/pytest_mx_on.log 2>&1; echo "rc=$?" >> $O/pytest_mx_on.log
tail -5 $O/pytest_mx_on.log
cp gpurun_out/parity_report.jsonl $O/parity_mx_on.jsonl
grep -h "fp8" $O/parity_mx_on.jsonl | cut -c1-700
