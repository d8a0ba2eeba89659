# K11 with per-bounce path compaction (ZR_K11=compact) against the megakernel: parity (the ReSTIR PT GPU tests with the switch on), then timings
R=$GRAFT_REPO_ROOT
cd $R
ZR_K11=compact timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -x -k "restir_pt or rpt or reference_passes" 2>&1 | tail -5
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(json.dumps({"ms": d["ms_per_step"], "k": {a: b for a, b in k.items() if b > 0.2}}))'
for mode in inline compact; do
  for a in "" "--config 4" "--scene tests/golden/cornell.npz"; do
    echo "== K11 $mode $a"; ZR_K11=$mode timeout 600 python bench.py --gpus 1 --steps 32 --warmup 8 --settle 16 --no-cpu-baseline $a 2>&1 | tail -1 | python -c "$P"
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_multi_rank" 2>&1 | tail -5
