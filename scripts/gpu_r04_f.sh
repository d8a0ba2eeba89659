# Round-4 GPU visit F: the whole GPU suite on the build with denoise definition 3, the replay chunking and the background rebuild; post-chain timings; the
# per-tile kernel times of the 8-way split (replay chunking); config 5.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r04f_gpu_suite.log 2>&1; tail -8 $O/r04f_gpu_suite.log
timeout 300 python tools/post_chain_bench.py > $O/r04f_post_chain.jsonl 2>/dev/null; cat $O/r04f_post_chain.jsonl
timeout 400 python bench.py --config 5 --no-cpu-baseline > $O/r04f_bench_5.json 2>/dev/null
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/r04f_bench_4.json 2>/dev/null
timeout 300 python bench.py --no-extra-workloads --no-cpu-baseline > $O/r04f_bench_cornell.json 2>/dev/null
timeout 300 python scripts/tile_kernels.py --scene synthetic > $O/r04f_tiles_atrium.jsonl 2>/dev/null
timeout 300 python scripts/tile_kernels.py > $O/r04f_tiles_cornell.jsonl 2>/dev/null
python - <<'PY'
import json, os, glob
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r04f_bench_*.json")):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unparsable", e); continue
    print(os.path.basename(f), d["ms_per_step"], d["value"], {k: v for k, v in d["roofline"]["kernel_ms_per_frame"].items() if v > 0.05})
for f in ("r04f_tiles_atrium", "r04f_tiles_cornell"):
    rows = [json.loads(l) for l in open(f"{O}/{f}.jsonl") if l.startswith("{")]
    if not rows: print(f, "no rows"); continue
    worst = max(rows, key=lambda r: r["sum_ms"])
    print(f, "slowest tile", worst["sum_ms"], "mean", round(sum(r["sum_ms"] for r in rows) / len(rows), 3), {k: v for k, v in worst["k"].items() if v > 0.05})
PY
