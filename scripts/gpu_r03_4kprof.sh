# PMC profile set of the 3840 x 2160 atrium frame with the denoise pass (config 5), then the two 4K bench lines with the matching profile in place
R=$GRAFT_REPO_ROOT
cd $R
WORKLOADS="rpt_3840x2160_atrium" bash scripts/gpu_r03_profiles.sh 2>&1 | grep -E "^==|k_rpt_pathtrace|k_svgf|k_rpt_stc|k_rpt_temporal|k_rpt_replay" | cut -c1-220
cp gpurun_out/r03_pmc_rpt_3840x2160_atrium.json profiles/
for c in 4k 5; do
  timeout 900 python bench.py --gpus 1 --no-cpu-baseline --config $c 2>&1 | tail -1 > gpurun_out/r03_bench_$c.json
  python -c "
import json; d=json.load(open('gpurun_out/r03_bench_$c.json')); r=d['roofline']
print('$c', d['ms_per_step'], 'ms', 'dom', r['kernel'], r['avg_launch_ms'], 'traffic', r['traffic'], 'x plane', r.get('traffic_over_plane_bytes'), 'valu', (r['valu'] or {}).get('busy_frac'))"
done
