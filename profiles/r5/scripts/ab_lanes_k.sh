# Round 5: how many concurrent step lanes (srgd_amd.lanes) pay for one HR tile (25 / 16 tiles per step)?  Same box, two rounds.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_lanes_k; mkdir -p $O
for R in 0 1; do for L in 1 2 3 4; do
  SRGD_STEP_LANES=$L timeout -k 10 600 python bench.py --no_cpu_baseline --no_profile --images 1 > $O/l${L}_$R.json 2>$O/l${L}_$R.err || { tail $O/l${L}_$R.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/l${L}_$R.json").read().strip().splitlines()[-1]); print("lanes=$L", round(d["value"],4), round(d["ms_per_step"],1), d["config"].get("step_lanes"))
PY
done; done
