# PMC passes for the K9 trace kernel on the synthetic atrium (separate runs, --pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --scene synthetic --integrator ${INTEG:-pt}"
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM -d $R/gpurun_out/pmct_sq -- $CMD > $R/gpurun_out/pmct_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT -d $R/gpurun_out/pmct_sq2 -- $CMD > $R/gpurun_out/pmct_sq2.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $R/gpurun_out/pmct_tcc -- $CMD > $R/gpurun_out/pmct_tcc.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmct_fetch -- $CMD > $R/gpurun_out/pmct_fetch.log 2>&1
cd $R
for k in sq sq2 tcc fetch; do DB=$(find gpurun_out/pmct_$k -name "*results.db" | head -1); python tools/rocpd_summary.py pmc $DB gpurun_out/pmct_${k}_summary.csv > /dev/null; tail -3 gpurun_out/pmct_$k.log; done
grep -E "k_trace|k_pt_shade|k_rpt_pathtrace" gpurun_out/pmct_*_summary.csv
