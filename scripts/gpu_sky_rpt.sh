cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_sky.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --sky-direct --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_skydi.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_gi_sky.json
