# compiler-variant libraries against the default build (scheduler strategy, -O2): Cornell + atrium ReSTIR PT, Cornell ReSTIR GI
R=$GRAFT_REPO_ROOT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.4}}))'
for lib in libzetaray_amd.so libzr_maxilp.so libzr_minreg.so libzr_o2.so; do
  [ -f zetaray_amd/$lib ] || continue
  export ZETARAY_AMD_LIB=$R/zetaray_amd/$lib
  for a in "" "--config 4" "--config 3"; do
    echo "== $lib $a"; timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
