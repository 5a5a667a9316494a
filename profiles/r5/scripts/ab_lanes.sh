# Round 5: two concurrent step lanes (srgd_amd.lanes) against one lane, same box, alternating: bench.py --images 1 (configs[1] as
# written: one HR tile, 25 / 16 tiles per step), configs[4] fp8 with one HR tile (50 / 32 samples per step), and the lock-step default
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_lanes; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q -k "two_step_lanes or independent_of_batch_size or device_noise_mode" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
run() {  # name lanes args...
  N=$1; L=$2; shift 2
  SRGD_STEP_LANES=$L timeout -k 10 600 python bench.py --no_cpu_baseline --no_profile "$@" > $O/$N.json 2>$O/$N.err || { tail $O/$N.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/$N.json").read().strip().splitlines()[-1]); print("$N", "lanes=$L", round(d["value"],4), round(d["ms_per_step"],1), d["config"].get("step_lanes"))
PY
}
for R in 0 1; do
  run images1_one_$R 1 --images 1
  run images1_auto_$R "" --images 1
done
for R in 0 1; do
  run fp8_images1_one_$R 1 --images 1 --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 --steps 3 --warmup 1
  run fp8_images1_auto_$R "" --images 1 --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 --steps 3 --warmup 1
done
run default_auto "" 
run default_two 2
