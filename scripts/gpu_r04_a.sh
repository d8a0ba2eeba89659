# Round-4 GPU visit A: the new parity cases (num_spatial_passes 0 / 2 vs the reference shaders' goldens), the tolerance-mode build's parity report,
# and contract vs fast arithmetic on Cornell + atrium (bench lines with per-kernel ms).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ref_passes.py -m gpu -q -x -k "two_spatial or no_spatial" > $O/r04a_tests_spatial.log 2>&1; tail -3 $O/r04a_tests_spatial.log
timeout 900 python -m pytest tests/test_fast_arith.py -m gpu -q > $O/r04a_tests_fast.log 2>&1; tail -15 $O/r04a_tests_fast.log
cp $O/fast_arith_parity.json $O/r04a_fast_arith_parity.json 2>/dev/null
timeout 600 python bench.py > $O/r04a_bench_default.json 2> $O/r04a_bench_default.err; tail -c 600 $O/r04a_bench_default.json
timeout 300 python bench.py --arith fast --no-extra-workloads --no-cpu-baseline > $O/r04a_bench_fast_cornell.json 2> $O/r04a_bench_fast_cornell.err
timeout 300 python bench.py --arith fast --config 4 --no-cpu-baseline > $O/r04a_bench_fast_atrium.json 2> $O/r04a_bench_fast_atrium.err
timeout 300 python bench.py --arith fast --config 3 --no-cpu-baseline > $O/r04a_bench_fast_gi.json 2>/dev/null
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/r04a_bench_contract_gi.json 2>/dev/null
timeout 300 python bench.py --arith fast --config pt --no-cpu-baseline > $O/r04a_bench_fast_pt.json 2>/dev/null
timeout 300 python bench.py --config pt --no-cpu-baseline > $O/r04a_bench_contract_pt.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r04a_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unparsable", e); continue
    print(os.path.basename(f), d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_per_frame"])
    for x in d.get("extra_workloads", []):
        print("   extra", x["preset"], x["ms_per_step"], x["value"], x["roofline"]["kernel_ms_per_frame"], x["cpu_baseline"])
PY
