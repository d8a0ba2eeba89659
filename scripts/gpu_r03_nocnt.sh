# what the per-wave ray-counter atomics cost: a build that skips FlushRayCounters (libzr_nocnt.so, -DZR_NO_RAY_COUNTERS) against the default
R=$GRAFT_REPO_ROOT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.05}}))'
for a in "" "--config 3" "--config 2a" "--config 4"; do
  for lib in libzetaray_amd.so libzr_nocnt.so; do
    export ZETARAY_AMD_LIB=$R/zetaray_amd/$lib
    echo "== $lib $a"; timeout 600 python bench.py --gpus 1 --steps 48 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
