cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "device_built_bvh or device_refit" 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.5}}))'
for b in host device; do
  echo "== ZR_BVH_BUILD=$b"; ZR_BVH_BUILD=$b timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --settle 8 --no-cpu-baseline --config 4 2>&1 | tail -1 | python -c "$P"
done
ZR_SCENE_UPDATE=rebuild timeout 600 python tools/refit_bench.py 2>&1 | tail -1 | cut -c1-900
