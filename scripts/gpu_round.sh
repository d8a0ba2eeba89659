# one GPU visit: parity tests, smoke, bench (all integrators, Cornell + synthetic atrium), rocprof kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 2>&1 | tail -1 | tee gpurun_out/bench_rpt.json
timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --direct --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_di.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_gi.json
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rpt_synth.json
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_pt_synth.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rpt -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 16 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_rpt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pt_synth -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --scene synthetic --integrator pt > $GRAFT_REPO_ROOT/gpurun_out/prof_pt_synth.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_rpt -name "*results.db" | head -1); python tools/rocpd_summary.py stats $DB gpurun_out/prof_rpt_summary.csv | head -30
DB=$(find gpurun_out/prof_pt_synth -name "*results.db" | head -1); python tools/rocpd_summary.py stats $DB gpurun_out/prof_pt_synth_summary.csv | head -12
