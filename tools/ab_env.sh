# Same-box A/B of the whole bench over environment settings: tools/ab_env.sh <out-dir> "<ENV=val ...>" "<ENV=val ...>" [bench args]  (two alternating rounds; "-" = no setting)
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O; A="$2"; B="$3"; shift 3
for R in 0 1; do for V in A B; do
  E="$A"; [ $V = B ] && E="$B"; [ "$E" = "-" ] && E=""
  env $E timeout -k 10 600 python bench.py --no_cpu_baseline "$@" > $O/${V}_$R.json 2>$O/${V}_$R.err || { tail $O/${V}_$R.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/${V}_$R.json").read().strip().splitlines()[-1]); print("$V [$E]", round(d["value"],4), round(d["ms_per_step"],1), {k: round(v,4) for k,v in (d.get("kernel_time_share") or {}).items() if v > 0.04})
PY
done; done
