#!/bin/bash
# A/B of the intra-wave work stealing in Traverse (zr_dev_scene.h): default library vs a -DZR_STEAL=0 build, + the vote / steal statistics of a
# -DZR_PROF build.  Variant libraries: /tmp/prof/build.sh <name> <flags> -> zetaray_amd/libzr_<name>.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_frame"]; print(d["ms_per_step"], {a:k[a] for a in k if a in ("gbuffer","trace","pt_shade","rpt_pathtrace","rpt_reconnect_temporal","rpt_reconnect_spatial","rgi")})'
run() {
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 4 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --scene synthetic --integrator pt --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
  timeout 600 python bench.py --gpus 1 --steps 64 --warmup 16 --integrator restir_gi --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"
}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/steal_pytest.log 2>&1; tail -n 3 gpurun_out/steal_pytest.log
echo "== steal (default)"; run
for v in "$@"; do
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_$v.so
  echo "== $v"; run
done
if [ -f zetaray_amd/libzr_profsteal.so ]; then
  export ZETARAY_AMD_LIB=$GRAFT_REPO_ROOT/zetaray_amd/libzr_profsteal.so
  timeout 600 python bench.py --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get("prof")))' > gpurun_out/profsteal_cornell.json
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scene synthetic --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get("prof")))' > gpurun_out/profsteal_atrium.json
  python - <<'PY'
import json
for n in ("cornell","atrium"):
    d=json.load(open(f"gpurun_out/profsteal_{n}.json")) or {}
    for k,v in d.items():
        print(n,k,{a:v.get(a) for a in ("trav_frac","trav_lane_util","node_lane_util","tri_lane_util","iters_per_call","rays_per_call","steals_per_call","pairs_per_steal","nodes_per_ray")})
PY
fi
