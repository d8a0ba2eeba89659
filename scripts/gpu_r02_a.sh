# round-2 GPU visit A: parity tests (incl. the BASELINE-config tests), smoke, default bench, rocprof kernel stats, one SQ PMC pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/r02a_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 2>&1 | tail -1 | tee gpurun_out/r02a_bench_rpt.json
timeout 600 python bench.py --gpus 1 --steps 16 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02a_bench_rpt_atrium.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02a_prof_rpt -- python $R/bench.py --gpus 1 --steps 16 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02a_prof_rpt.log 2>&1
cd $R
DB=$(find gpurun_out/r02a_prof_rpt -name "*results.db" | head -1); python tools/rocpd_summary.py stats $DB gpurun_out/r02a_kernel_stats_rpt1080p.csv | head -20
cd /tmp
for scene in cornell atrium; do
  if [ $scene = atrium ]; then EXTRA="--scene synthetic"; else EXTRA=""; fi
  CMD="python $R/bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline $EXTRA"
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES -d $R/gpurun_out/r02a_pmc_sq_$scene -- $CMD > $R/gpurun_out/r02a_pmc_sq_$scene.log 2>&1
  DB=$(find $R/gpurun_out/r02a_pmc_sq_$scene -name "*results.db" | head -1); python $R/tools/rocpd_summary.py pmc $DB $R/gpurun_out/r02a_pmc_sq_$scene.csv | grep -E "k_rpt|k_gbuffer" | head -60
done
rm -rf $R/gpurun_out/r02a_prof_rpt $R/gpurun_out/r02a_pmc_sq_cornell $R/gpurun_out/r02a_pmc_sq_atrium
