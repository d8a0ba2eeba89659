#!/bin/bash
# DI kernels' block size A/B
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],4) for a in k if a in ("rdi_temporal","rdi_spatial","sdi_temporal","sdi_spatial")})'
run() {
  python bench.py --gpus 1 --steps 64 --warmup 8 --direct --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 64 --warmup 8 --scene tests/golden/cornell.npz --integrator pt --sky-direct --di-only --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 16 --warmup 4 --direct --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
echo "== default"; run
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "di" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
