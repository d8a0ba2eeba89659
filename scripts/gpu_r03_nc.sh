# LDS node cache in more kernels (K9 trace, K10, K13 replays): -DZR_NODE_CACHE_MORE build against the default
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.3}}))'
for lib in libzetaray_amd.so libzr_nc_more.so; do
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib
  for a in "--config 4" "--integrator restir_gi --scene synthetic" "--integrator pt --scene synthetic" "--config 3" "--config pt"; do
    echo "== $lib $a"; timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --settle 8 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
