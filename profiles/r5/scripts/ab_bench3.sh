# Same-box A/B of the whole bench over THREE libraries: tools/ab_bench3.sh <out-dir> <libA> <libB> <libC> [bench args...] (two alternating rounds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; A=$2; B=$3; C3=$4; shift 4
for V in $A $B $C3 $A $B $C3; do
  L=$R/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$R/srgd_amd/libsrgd_hip.so
  N=$(ls $O | grep -c "^bench_$V")
  SRGD_HIP_LIB=$L timeout -k 10 500 python bench.py --no_cpu_baseline "$@" > $O/bench_${V}_$N.json 2>$O/bench_${V}_$N.err || { tail $O/bench_${V}_$N.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_${V}_$N.json").read().strip().splitlines()[-1]); print("$V", round(d["value"],4), round(d["ms_per_step"],1), "gn share", d["kernel_time_share"].get("groupnorm_silu"), "gn GB/s", d["hbm_kernels"]["groupnorm_silu"]["achieved"])
PY
done
