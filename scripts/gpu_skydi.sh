cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --integrator pt --sky-direct --di-only --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_skydi.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --scene tests/golden/cornell.npz --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt_sky.json
