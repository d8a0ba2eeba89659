R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_multi_rank" 2>&1 | grep -E "Assertion|assert |Error|passed|failed" | head -20
cd /tmp && export TMPDIR=/tmp
for mode in compact inline; do
export ZR_K11=$mode
O=$R/gpurun_out/r03_compact_$mode
rm -rf ${O}_stats
timeout 600 rocprofv3 --kernel-trace --stats -d ${O}_stats -- python $R/bench.py --gpus 1 --steps 8 --warmup 2 --settle 8 --no-cpu-baseline > ${O}_stats.log 2>&1
python $R/tools/rocpd_summary.py stats $(find ${O}_stats -name "*results.db" | head -1) $R/gpurun_out/r03_kernel_stats_compact_$mode.csv > /dev/null
rm -rf ${O}_stats
echo "== $mode"; grep -E "k_rpt_pt|k_rpt_pathtrace" $R/gpurun_out/r03_kernel_stats_compact_$mode.csv
done
