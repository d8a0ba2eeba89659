# PMC passes (separate runs, --pmc only; see MI355X_MICROARCH.md "rocprofv3 PMC slots") for the default bench (ReSTIR PT, Cornell 1080p):
# HBM-side traffic per kernel launch -> gpurun_out/pmc_{fetch,write}_summary.csv; tools/pmc_traffic.py turns them into profiles/*.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS}"
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -- $CMD > $R/gpurun_out/pmc_write.log 2>&1
cd $R
for k in fetch write; do DB=$(find gpurun_out/pmc_$k -name "*results.db" | head -1); python tools/rocpd_summary.py pmc $DB gpurun_out/pmc_${k}_summary.csv > /dev/null; done
grep -E "k_rpt|k_gbuffer|k_trace|k_pt_|k_rgi|k_sdi|k_rdi" gpurun_out/pmc_fetch_summary.csv gpurun_out/pmc_write_summary.csv
