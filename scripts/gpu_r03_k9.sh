# K9 wavefront stages allocate queue slots per block (AllocSlotsBlock) instead of per wave: parity, then the path tracer on Cornell and the atrium
# (before: Cornell 1.79 ms: pt_init 0.249, trace 0.78, pt_shade 0.68; atrium 9.7 ms)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -x -k "path_trac or pathtrace or pt_ or k9 or golden or config1 or reference_passes or smoke or textured" 2>&1 | grep -E "passed|failed" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": k}))'
for a in "--config pt" "--integrator pt --scene synthetic"; do
  echo "== $a"; timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
done
