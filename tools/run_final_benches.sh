# Final bench lines of a round -> gpurun_out/<tag>_final: default bf16 incl. cpu_baseline, the same with ONE HR tile alone (configs[1] as
# written: 25 / 16 tiles per launch), configs[4] in three precisions, fp32, forced-dist (RCCL at world 1), configs[3] canvas.
#   bash tools/run_final_benches.sh r5
cd $GRAFT_REPO_ROOT; T=${1:-r6}; O=gpurun_out/${T}_final; mkdir -p $O
python bench.py > $O/bench_${T}_default.json 2>$O/bench_${T}_default.err || { tail $O/bench_${T}_default.err; exit 1; }
python bench.py --no_cpu_baseline --images 1 > $O/bench_${T}_images1.json 2>$O/err_images1.txt || { tail $O/err_images1.txt; exit 1; }
for P in bf16 fp8 fp8_mixed; do
  python bench.py --no_cpu_baseline --precision $P --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_${T}_config5_$P.json 2>$O/err_$P.txt || { tail $O/err_$P.txt; exit 1; }
done
python bench.py --no_cpu_baseline --precision fp32 --steps 2 --warmup 1 --images 2 > $O/bench_${T}_fp32.json 2>$O/err_fp32.txt || { tail $O/err_fp32.txt; exit 1; }
python bench.py --no_cpu_baseline --precision f16x3 > $O/bench_${T}_f16x3.json 2>$O/err_f16x3.txt || { tail $O/err_f16x3.txt; exit 1; }
python bench.py --no_cpu_baseline --precision f16mx2 > $O/bench_${T}_f16mx2.json 2>$O/err_f16mx2.txt || { tail $O/err_f16mx2.txt; exit 1; }
python bench.py --no_cpu_baseline --precision f16x3 --images 1 > $O/bench_${T}_f16x3_images1.json 2>$O/err_f16x3_1.txt || { tail $O/err_f16x3_1.txt; exit 1; }
SRGD_FORCE_DIST=1 python bench.py --gpus 1 --no_cpu_baseline --no_profile > $O/bench_${T}_forced_dist_nccl_world1_tiles.json 2>$O/err_fd1.txt || { tail $O/err_fd1.txt; exit 1; }
python bench.py --no_cpu_baseline --no_profile --workload canvas --lr_size 2048 --steps 1 --warmup 0 > $O/bench_${T}_config4_canvas8192_1gpu.json 2>$O/err_c4.txt || { tail $O/err_c4.txt; exit 1; }
SRGD_FORCE_DIST=1 python bench.py --gpus 1 --no_cpu_baseline --no_profile --workload canvas --lr_size 2048 --steps 1 --warmup 0 > $O/bench_${T}_forced_dist_nccl_world1_canvas.json 2>$O/err_fd2.txt || { tail $O/err_fd2.txt; exit 1; }
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_${T}_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print(f.split("/")[-1], round(d["value"],4), d["unit"], round(d["ms_per_step"],1), r.get("achieved"), r.get("frac"), d.get("exchange_ms"), d.get("exchange_share"))
PY
