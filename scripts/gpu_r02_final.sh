# Round-2 final measurement set on the final build: the r02 PMC / kernel-stats profiles first (bench.py then reads its traffic / valu numbers from
# them), then one bench line per workload -> gpurun_out/r02_bench_*.json (copied into profiles/ by hand)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WORKLOADS="rpt_cornell rpt_atrium gi_cornell pt_cornell" bash scripts/gpu_r02_profiles.sh 2>&1 | grep -E "^==|k_rpt_pathtrace|k_rpt_stc|k_rpt_temporal|k_rgi|k_trace_simple|k_pt_shade"
cp gpurun_out/r02_pmc_*.json profiles/ 2>/dev/null
B="timeout 600 python bench.py --gpus 1"
run() { name=$1; shift; $B "$@" 2>&1 | tail -1 > gpurun_out/r02_bench_$name.json; python -c "
import json,sys; d=json.load(open('gpurun_out/r02_bench_$name.json')); r=d.get('roofline',{}); print('$name', d['ms_per_step'], d['value'], r.get('kernel'), r.get('bound'), r.get('frac'), (r.get('valu') or {}).get('issue_frac'))"; }
run rpt1080p --steps 256 --warmup 64
run rpt_atrium1080p --steps 32 --warmup 8 --scene synthetic --no-cpu-baseline
run gi1080p --steps 128 --warmup 16 --integrator restir_gi --no-cpu-baseline
run pt1080p --steps 128 --warmup 16 --integrator pt --no-cpu-baseline
run rpt_di1080p --steps 64 --warmup 8 --direct --no-cpu-baseline
run pt_atrium1080p --steps 32 --warmup 8 --integrator pt --scene synthetic --no-cpu-baseline
run gi_atrium1080p --steps 32 --warmup 8 --integrator restir_gi --scene synthetic --no-cpu-baseline
run rpt_sunsky_cornell1080p --steps 64 --warmup 8 --scene tests/golden/cornell.npz --no-cpu-baseline
run skydi_cornell1080p --steps 64 --warmup 8 --scene tests/golden/cornell.npz --integrator pt --sky-direct --di-only --no-cpu-baseline
run rpt_atrium_textured1080p --steps 16 --warmup 4 --scene synthetic --textured --no-cpu-baseline
