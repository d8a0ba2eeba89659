# traversal experiments: PT parity tests, then synthetic-atrium benches (K9 PT, ReSTIR PT) and the Cornell ones
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt_synth.json
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_synth.json
timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt.json
