#!/bin/bash
# which K12 map (if any) should schedule the fused temporal reconnect kernel: ZR_TEMPORAL_MAP = 0 none, 1 CtN, 2 NtC
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("rpt_reconnect_temporal","rpt_reconnect_spatial","rpt_sort_temporal","rpt_sort_spatial")})'
for m in 0 1 2; do
  export ZR_TEMPORAL_MAP=$m; echo "== ZR_TEMPORAL_MAP=$m"
  python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
done
unset ZR_TEMPORAL_MAP
python -m pytest tests/test_ref_passes.py -q -m gpu -k "rpt" > gpurun_out/pt.log 2>&1; grep -E "passed|failed" gpurun_out/pt.log
ZR_TEMPORAL_MAP=1 python -m pytest tests/test_ref_passes.py tests/test_gpu_parity.py -q -m gpu -k "rpt or restir_pt" > gpurun_out/pt1.log 2>&1; grep -E "passed|failed" gpurun_out/pt1.log
