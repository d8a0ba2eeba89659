# a round-3 GPU visit: full GPU test suite, default bench line + presets, r03 profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout 1400 python -m pytest tests -q -m gpu 2>&1 | tail -6
for c in ${CONFIGS:-default 4}; do
  A="--no-cpu-baseline --config $c"; [ $c = default ] && A=""      # the default line is the driver's: whole contract, cpu_baseline included
  timeout 900 python bench.py --gpus 1 $A ${BENCH_ARGS} 2>&1 | tail -1 > gpurun_out/r03_bench_$c.json
  python -c "
import json,sys; d=json.load(open('gpurun_out/r03_bench_$c.json')); r=d['roofline']
print('$c', d['ms_per_step'], 'ms', d['value'], 'Mrays/s', 'dom', r['kernel'], r['avg_launch_ms'], 'frac', r['frac'], {k:v for k,v in r['kernel_ms_per_frame'].items() if v>0.05})"
done
[ -n "$SKIP_PROFILES" ] || WORKLOADS="${WORKLOADS:-rpt_cornell}" bash scripts/gpu_r03_profiles.sh 2>&1 | tail -12
[ -n "$SKIP_TESTS" ] || timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
