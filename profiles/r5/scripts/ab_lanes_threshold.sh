# Where does the second step lane stop paying?  bench.py --images N (25 N / 16 N tiles per step), SRGD_STEP_LANES=1 vs 2, same box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_lanes_thr; mkdir -p $O
for N in 2 3 4; do for L in 1 2 1 2; do
  K=$(ls $O | grep -c "^n${N}_l${L}_")
  SRGD_STEP_LANES=$L timeout -k 10 600 python bench.py --no_cpu_baseline --no_profile --images $N --steps $((2*N)) --warmup $N > $O/n${N}_l${L}_$K.json 2>$O/n${N}_l${L}_$K.err || { tail $O/n${N}_l${L}_$K.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/n${N}_l${L}_$K.json").read().strip().splitlines()[-1]); print("images=$N lanes=$L", round(d["value"],4), round(d["ms_per_step"],1), d["config"].get("step_lanes"), d["config"].get("tiles_per_unet_launch"))
PY
done; done
