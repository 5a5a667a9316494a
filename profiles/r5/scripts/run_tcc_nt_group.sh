# n-tile grouping inside an XCD band (SRGD_CONV3_NT_GROUP): TCC misses, throughput and in-kernel clock on the deep 3x3 layers
# ARCHIVED (round 6): needs profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch applied - see README.md here.
grep -q SRGD_CONV3_NT_GROUP srgd_amd/csrc/conv3x3_bf16.hip || { echo "apply profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch first: the knobs this script drives are not in this tree"; exit 1; }
# (needs the knobs of profiles/r5/conv3x3_bf16_nt_group_and_nt_policy_knobs.patch applied to srgd_amd/csrc/conv3x3_bf16.hip: they were removed after this measurement)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_tcc2; mkdir -p $O; V=$R/srgd_amd/variants
cd /tmp && export TMPDIR=/tmp
for SHAPE in "3x3 1024->1024 @32" "3x3 512->512 @64"; do
for G in 0 4 2 1; do
  NAME="$(echo $SHAPE | tr ' >' '__') G=$G"
  SRGD_CONV3_NT_GROUP=$G rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/p -o p -- python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 5 --impls 1,2 > $O/p.log 2>&1 || { tail -5 $O/p.log; }
  grep "3x3" $O/p.log | cut -c1-30,100-190 | sed "s/^/$NAME check: /" >> $O/table.txt
  python3 - "$(find $O/p -name '*.db' | head -1)" "$NAME" <<'PY' >> $O/table.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, n, avg in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3_bf16_kernel%' group by counter_name"):
    print(f"{sys.argv[2]:34s} {name:14s} launches {n:3d}  mean per launch {avg:14.0f}")
PY
  rm -rf $O/p
  for K in 1 2; do SRGD_CONV3_NT_GROUP=$G python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 20 --impls 2 2>&1 | grep "3x3" | sed "s/^/$NAME: /" >> $O/table.txt; done
  SRGD_CONV3_NT_GROUP=$G SRGD_HIP_LIB=$V/libsrgd_hip_stamps.so python3 $R/tools/bench_conv.py --only "$SHAPE" --batch 125 --iters 20 --impls 2 2>&1 | grep stamps | tail -1 | cut -c1-60,200-300 | sed "s/^/$NAME: /" >> $O/table.txt
done; done
cat $O/table.txt
