R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
B="timeout -k 10 500 python bench.py --no_cpu_baseline"
$B --steps 5 --warmup 5 --precision bf16 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_r3_config5_bf16.json 2>$O/err.log &&
$B --steps 5 --warmup 5 --precision fp8 --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_r3_config5_fp8.json 2>>$O/err.log &&
$B --steps 5 --warmup 5 --precision fp8_mixed --ddpm_steps 100 --class_cond_scale 2.0 > $O/bench_r3_config5_fp8_mixed.json 2>>$O/err.log &&
$B --steps 1 --warmup 0 --workload canvas --lr_size 2048 > $O/bench_r3_config4_canvas8192_1gpu.json 2>>$O/err.log &&
SRGD_FORCE_DIST=1 $B --steps 5 --warmup 5 > $O/bench_r3_forced_dist_nccl_world1_tiles.json 2>>$O/err.log &&
SRGD_FORCE_DIST=1 $B --steps 1 --warmup 1 --workload canvas --lr_size 512 > $O/bench_r3_forced_dist_nccl_world1_canvas.json 2>>$O/err.log &&
$B --steps 5 --warmup 5 --precision fp32 > $O/bench_r3_fp32.json 2>>$O/err.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['metric'][:40], round(d['value'],4), d['unit'], d.get('dist_backend'), d.get('forced_dist'), d.get('tile_allgathers'), d.get('gathered_hr_tiles'), round(d.get('roofline',{}).get('achieved',0)))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
