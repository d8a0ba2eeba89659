# Round-4 closing set, third edition (parameter validation, threading test, pick): the set of gpu_r04_g.sh + PMC profiles of the GI, path-tracer and 3840 x 2160 workloads
# ReSTIR PT workloads (bench.py reports traffic / valu only from a profile whose source hash matches), every bench preset, the default line.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04m_gpu_suite.log 2>&1; tail -8 $O/r04m_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04m_smoke.log 2>&1; tail -2 $O/r04m_smoke.log
timeout 300 python tools/post_chain_bench.py > $O/r04m_post_chain.jsonl 2>/dev/null; cat $O/r04m_post_chain.jsonl | cut -c1-900
TAG=r04 WORKLOADS="rpt_cornell rpt_atrium gi_cornell pt_cornell rpt_3840x2160_atrium" bash scripts/gpu_r04_profiles.sh > $O/r04m_profiles.log 2>&1; tail -4 $O/r04m_profiles.log
cd $R
mkdir -p profiles && cp $O/r04_pmc_*.json profiles/ 2>/dev/null
for c in 2a 2b 3 4 4k 5 pt; do timeout 400 python bench.py --config $c --no-cpu-baseline > $O/r04_bench_$c.json 2>/dev/null; done
timeout 900 python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err
timeout 300 python bench.py --arith fast --no-extra-workloads --no-cpu-baseline > $O/r04_bench_fast.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r04_bench_*.json")):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unparsable", e); continue
    r = d["roofline"]
    print(os.path.basename(f), d["ms_per_step"], d["value"], r["kernel"], r["frac"], r.get("traffic"), r.get("traffic_over_plane_bytes"), (r.get("valu") or {}).get("busy_frac"))
    for x in d.get("extra_workloads", []):
        print("   extra", x["preset"], x["ms_per_step"], x["value"], x["roofline"]["frac"], x["roofline"].get("traffic"), x["cpu_baseline"]["value"] if x.get("cpu_baseline") else None)
PY
