# dynamic work fetch in the replay kernels (K13): parity, then the atrium frame (before: rpt_replay_temporal 1.27 ms, rpt_replay_spatial 0.88 ms at 1080p)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -x -k "restir_pt or rpt or reference_passes or variants" 2>&1 | grep -E "passed|failed" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.3 or "replay" in a}}))'
for a in "--config 4" "" "--config 4k" "--scene synthetic --textured"; do
  echo "== $a"; timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
done
