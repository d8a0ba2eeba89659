#!/bin/bash
# occupancy variants of the large-scene / textured K11 builds (variant libraries zetaray_amd/libzr_<name>.so)
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],3) for a in k if a in ("rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial")})'
run() { python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --textured --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
        python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; }
echo "== default"; run
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -k "textured or large or atrium" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
