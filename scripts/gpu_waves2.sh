#!/bin/bash
# occupancy of the TEXTURED K9 shade kernel and of k_rgi on textured scenes (variant libraries zetaray_amd/libzr_<name>.so)
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],3) for a in k if a in ("pt_shade","trace","rgi","gbuffer")})'
run() { python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --textured --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
        python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --textured --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; }
echo "== default"; run
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py tests/test_ref_passes.py -q -m gpu -k "textured" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
