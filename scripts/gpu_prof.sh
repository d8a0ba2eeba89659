# Section profile of the ReSTIR PT kernels with a -DZR_PROF build of the library (zetaray_amd/libzr_prof.so: s_memtime section timers +
# traversal vote statistics, zr_dev_scene.h ProfScope).  Prints bench.py's "prof" object per workload.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({"ms": d["ms_per_step"], "kernels": d["roofline"]["kernel_ms_per_frame"], "prof": d.get("prof")}, indent=1))'
for lib in ${LIBS:-libzr_prof.so}; do
  [ -f zetaray_amd/$lib ] || continue
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib
  echo "== $lib cornell"
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P" | tee gpurun_out/prof_${lib%.so}_cornell.json
  echo "== $lib atrium"
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P" | tee gpurun_out/prof_${lib%.so}_atrium.json
done
