cd $GRAFT_REPO_ROOT
for W in 0 1; do
  echo "== ONE_WG=$W"
  SRGD_CONV3_ONE_WG=$W SRGD_CONV3_STAMPS=1 python tools/bench_conv.py --batch 125 --iters 100 --only "3x3 128->128 @256" --impls 2 2>&1 | grep -v amdgpu | tail -2
  SRGD_CONV3_ONE_WG=$W SRGD_CONV3_STAMPS=1 python tools/bench_conv.py --batch 125 --iters 100 --only "3x3 1024->1024 @32" --impls 2 2>&1 | grep -v amdgpu | tail -2
done
