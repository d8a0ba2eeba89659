# quick A/B: per-kernel ms/frame of the default library (and optional variant libraries LIBS="libzr_x.so ...")
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_frame"])'
for lib in libzetaray_amd.so $LIBS; do
  [ -f zetaray_amd/$lib ] || continue
  echo "== $lib"
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib
  timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
