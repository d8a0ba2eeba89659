cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("gbuffer","trace","pt_shade","rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi","rdi_temporal","rdi_spatial","sdi_temporal","sdi_spatial")})'
for lib in libzetaray_amd.so $LIBS; do
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/$lib
  echo "== $lib"
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --direct --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
