# the round's closing GPU visit: the r03 profile set of the final sources -> profiles/ (in the box's copy too, so that the bench lines that follow
# report traffic / valu from a matching profile), every bench preset, then the whole GPU test suite and the smoke test
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
WORKLOADS="rpt_cornell rpt_atrium gi_cornell" bash scripts/gpu_r03_profiles.sh 2>&1 | grep -E "^==|k_rpt_pathtrace|k_rgi |k_rpt_stc|k_rpt_temporal" | cut -c1-220
cp gpurun_out/r03_pmc_rpt_cornell.json gpurun_out/r03_pmc_rpt_atrium.json gpurun_out/r03_pmc_gi_cornell.json profiles/
for c in default 2a 2b 3 pt 4 4k 5; do
  A="--no-cpu-baseline --config $c"; [ $c = default ] && A=""
  timeout 900 python bench.py --gpus 1 $A 2>&1 | tail -1 > gpurun_out/r03_bench_$c.json
  python -c "
import json; d=json.load(open('gpurun_out/r03_bench_$c.json')); r=d['roofline']
print('$c', d['ms_per_step'], 'ms', d['value'], 'Mrays/s', 'dom', r['kernel'], r['avg_launch_ms'], 'frac', r['frac'], 'bound', r['bound'], 'traffic', r['traffic'], 'x plane', r.get('traffic_over_plane_bytes'), 'valu', (r['valu'] or {}).get('busy_frac'))"
done
timeout 1400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
