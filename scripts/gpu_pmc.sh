# PMC passes for the ReSTIR PT bench (separate runs, --pmc only; see MI355X_MICROARCH.md "rocprofv3 PMC slots")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM -d $R/gpurun_out/pmc_sq -- $CMD > $R/gpurun_out/pmc_sq.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -- $CMD > $R/gpurun_out/pmc_write.log 2>&1
cd $R
for k in sq fetch write; do DB=$(find gpurun_out/pmc_$k -name "*results.db" | head -1); python tools/rocpd_summary.py pmc $DB gpurun_out/pmc_${k}_summary.csv > /dev/null; done
grep -E "k_rpt_pathtrace|k_rpt_stc|k_rpt_temporal|k_gbuffer" gpurun_out/pmc_sq_summary.csv gpurun_out/pmc_fetch_summary.csv gpurun_out/pmc_write_summary.csv
