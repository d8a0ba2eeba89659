cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("gbuffer","trace","pt_shade","rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi")})'
run() {
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
echo "== default (sign-selected slabs)"; run
ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_v23.so; export ZETARAY_AMD_LIB
echo "== v23"; run
unset ZETARAY_AMD_LIB
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
