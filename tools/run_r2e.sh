for st in 0 -1 16 64 128; do
  if [ $st = -1 ]; then unset SRGD_CONV3_STAGGER; else export SRGD_CONV3_STAGGER=$st; fi
  echo "== stagger knob $st"
  python tools/bench_conv_fp8.py --batch 125 --iters 5 --only "128->128 @256" 2>&1 | tail -1
  python tools/bench_conv_fp8.py --batch 125 --iters 5 --only "1024->1024" 2>&1 | tail -1
  python tools/bench_conv_fp8.py --batch 125 --iters 5 --only "256->256 @128" 2>&1 | tail -1
done
