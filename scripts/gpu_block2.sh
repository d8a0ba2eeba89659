#!/bin/bash
# k_rgi block size A/B (default library vs variants) + the K11 one-wave-block default re-checked on the full GPU parity subset
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:round(k[a],4) for a in k if a in ("rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi","gbuffer")})'
run() {
  python bench.py --gpus 1 --steps 128 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  python bench.py --gpus 1 --steps 16 --warmup 4 --integrator restir_gi --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
echo "== default"; run
python bench.py --gpus 1 --steps 128 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
for v in "$@"; do export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so; echo "== $v"; run
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gi" > gpurun_out/pt_$v.log 2>&1; grep -E "passed|failed" gpurun_out/pt_$v.log
done
unset ZETARAY_AMD_LIB
python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log
