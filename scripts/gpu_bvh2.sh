# second round of traversal experiments: exact-sweep SAH for small ranges (ZR_BVH_SWEEP) and the whole-leaf triangle phase (variant library)
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("gbuffer","trace","pt_shade","rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi")})'
run() {
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
echo "== default (leaf 2)"; run
echo "== sweep below 32"; ZR_BVH_SWEEP=32 run
echo "== sweep below 256"; ZR_BVH_SWEEP=256 run
export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_leaf2.so
echo "== whole-leaf tri phase"; run
echo "== whole-leaf tri phase + sweep 256"; ZR_BVH_SWEEP=256 run
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "atrium or baseline or traversal or trace" 2>&1 | tail -3
