# four tiles per block in k_rpt_light (a quarter of the list atomics): parity, then the atrium frame (before: rpt_classify_temporal 0.093, rpt_spatial_search 0.101 ms)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -x -k "restir_pt or rpt or reference_passes or variants or tile_split" 2>&1 | grep -E "passed|failed" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if "classify" in a or "search" in a or "replay" in a}}))'
for a in "--config 4" "" "--config 4k"; do
  echo "== $a"; timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
done
