# Round 5 (review item 2): does the weight stream of the deep 3x3 layers stay in the XCDs' L2?  TCC hit / miss and fabric read
# ARCHIVED (round 6): needs profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch applied - see README.md here.
grep -q SRGD_CONV3_NT_GROUP srgd_amd/csrc/conv3x3_bf16.hip || { echo "apply profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch first: the knobs this script drives are not in this tree"; exit 1; }
# (needs the knobs of profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch applied to srgd_amd/csrc/conv3x3_bf16.hip: they were removed after this measurement)
# requests of conv3x3_bf16_kernel on 1024 -> 1024 @32^2 (18.9 MB of weights, 125 tiles) for: production tile map; n-tiles pinned
# to XCDs (SRGD_CONV3_XCD_PIN_KB=1024); pinned + non-temporal halo DMAs and output stores (variant build `nt`); next to the
# un-profiled throughput and the in-kernel clock (stamp builds).  PMC passes carry counters only (no sys / hip tracing).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_tcc; mkdir -p $O; V=$R/srgd_amd/variants
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\(HIT\|MISS\|RDREQ\|REQ\|READ\)[A-Za-z0-9_]*" | sort -u > $O/tcc_counter_names.txt
SHAPE="3x3 1024->1024 @32"
run() {   # name, env PIN, lib suffix ("" = shipped)
  local NAME=$1 PIN=$2 LIB=$3 SL=$4
  local L=$R/srgd_amd/libsrgd_hip.so; [ -n "$LIB" ] && L=$V/libsrgd_hip_$LIB.so
  for PASS in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    TAG=$(echo $PASS | tr ' ' '-')
    SRGD_CONV3_XCD_PIN_KB=$PIN SRGD_HIP_LIB=$L rocprofv3 --kernel-trace --pmc $PASS -d $O/p_${NAME}_$TAG -o p -- python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 5 --impls 2 > $O/p_${NAME}_$TAG.log 2>&1 || { tail -5 $O/p_${NAME}_$TAG.log; continue; }
    python3 - "$(find $O/p_${NAME}_$TAG -name '*.db' | head -1)" "$NAME" <<'PY' >> $O/tcc_table.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, n, avg in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_bf16_kernel%' group by counter_name"):
    print(f"{sys.argv[2]:28s} {name:24s} launches {n:3d}  mean per launch {avg:16.0f}")
PY
    rm -rf $O/p_${NAME}_$TAG
  done
  for K in 1 2; do SRGD_CONV3_XCD_PIN_KB=$PIN SRGD_HIP_LIB=$L python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 20 --impls 2 2>&1 | grep "3x3" | sed "s/^/$NAME: /" >> $O/tcc_table.txt; done
  SRGD_CONV3_XCD_PIN_KB=$PIN SRGD_HIP_LIB=$V/libsrgd_hip_$SL.so python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 20 --impls 2 2>&1 | grep stamps | tail -1 | sed "s/^/$NAME: /" >> $O/tcc_table.txt
}
run production 0 "" stamps
run xcd_pin 1024 "" stamps
run xcd_pin_nt 1024 nt ntstamps
run production_nt 0 nt ntstamps
cat $O/tcc_counter_names.txt | tr '\n' ' '; echo; cat $O/tcc_table.txt
