# Round-2 profile set: for each workload a --kernel-trace --stats run and three separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE, the SQ set) of the same bench.py command; summaries -> gpurun_out/r02_*, copied into profiles/ by hand.
#   WORKLOADS="rpt_cornell rpt_atrium gi_cornell pt_cornell gi_atrium pt_atrium" bash scripts/gpu_r02_profiles.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES"
for wl in ${WORKLOADS:-rpt_cornell rpt_atrium}; do
  case $wl in
    rpt_cornell) ARGS="";;
    rpt_atrium) ARGS="--scene synthetic";;
    gi_cornell) ARGS="--integrator restir_gi";;
    gi_atrium) ARGS="--integrator restir_gi --scene synthetic";;
    pt_cornell) ARGS="--integrator pt";;
    pt_atrium) ARGS="--integrator pt --scene synthetic";;
  esac
  CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline $ARGS"
  O=$R/gpurun_out/r02_$wl
  rm -rf ${O}_*
  timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- $CMD > ${O}_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d ${O}_fetch -- $CMD > ${O}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d ${O}_write -- $CMD > ${O}_write.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ -d ${O}_sq -- $CMD > ${O}_sq.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/r02_kernel_stats_$wl.csv > /dev/null
  for k in fetch write sq; do python $R/tools/rocpd_summary.py pmc $(find ${O}_$k -name "*results.db" | head -1) ${O}_$k.csv > /dev/null; done
  echo "== $wl"
  python $R/tools/pmc_profile.py ${O}_fetch.csv ${O}_write.csv ${O}_sq.csv $R/gpurun_out/r02_pmc_$wl.json
  rm -rf ${O}_stats ${O}_fetch ${O}_write ${O}_sq
done
