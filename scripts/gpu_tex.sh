#!/bin/bash
# occupancy of the TEXTURED K11 permutation: default (compiler's choice, 2 waves) vs variant libraries; the textured parity tests decide whether a
# variant is usable at all (ROCm 7.2 miscompiled the forced 3-wave build in round 1)
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],3) for a in k if a in ("rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial")})'
run() { python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --textured --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; }
echo "== default"; run
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -k "textured" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
