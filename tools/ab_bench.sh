# Same-box A/B of the whole bench: tools/ab_bench.sh <out-dir> <libA> <libB> [bench args...]   (alternating, two runs each; lib "default" = the shipped one)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; A=$2; B=$3; shift 3
for V in $A $B $A $B; do
  L=$R/srgd_amd/variants/libsrgd_hip_$V.so; [ $V = default ] && L=$R/srgd_amd/libsrgd_hip.so
  N=$(ls $O | grep -c "^bench_$V")
  SRGD_HIP_LIB=$L timeout -k 10 500 python bench.py --no_cpu_baseline "$@" > $O/bench_${V}_$N.json 2>$O/bench_${V}_$N.err || { tail $O/bench_${V}_$N.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/bench_${V}_$N.json").read().strip().splitlines()[-1]); print("$V", round(d["value"],4), round(d["ms_per_step"],1), round(d["roofline"]["achieved"]), {k: round(v,4) for k,v in (d.get("kernel_time_share") or {}).items()})
PY
done
