# textured permutations: parity tests, untextured regression check, textured vs untextured atrium benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_gi.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt_synth.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --integrator pt --textured --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt_synth_tex.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_synth.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --textured --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_synth_tex.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 2 --scene synthetic --integrator restir_gi --textured --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_gi_synth_tex.json
