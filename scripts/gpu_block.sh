#!/bin/bash
# block size of the heavy ReSTIR PT kernels: default library vs variant libraries (zetaray_amd/libzr_<name>.so), full frame and a 1/8 tile
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],4) for a in k if a in ("rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi","gbuffer")})'
run() {
  python bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 256 --warmup 32 --width 480 --height 544 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 32 --warmup 4 --width 480 --height 544 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
echo "== default"; run
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "restir_pt or rpt" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
