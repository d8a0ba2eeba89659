# Round-4 GPU visit D: denoise on tiles (GPU), background SAH rebuild (parity + the moving-instance frame cost on the atrium), config 5 with the
# denoise pass, per-tile kernel times of the 8-way split (megakernel vs per-bounce compaction), a-trous step 4 from an LDS tile.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise.py tests/test_gpu_parity.py -m gpu -q -k "denoise or background_sah or device_refit or stream_ordered or device_built" > $O/r04d_tests.log 2>&1; tail -6 $O/r04d_tests.log
timeout 400 python bench.py --config 5 --no-cpu-baseline > $O/r04d_bench_5.json 2> $O/r04d_bench_5.err; tail -c 1500 $O/r04d_bench_5.json; tail -3 $O/r04d_bench_5.err
timeout 300 python tools/post_chain_bench.py > $O/r04d_post_chain.jsonl 2>/dev/null; cat $O/r04d_post_chain.jsonl
ZR_DENOISE=lds4 timeout 300 python tools/post_chain_bench.py > $O/r04d_post_chain_lds4.jsonl 2>/dev/null; cat $O/r04d_post_chain_lds4.jsonl
REFIT_MOVING_FRAMES=64 timeout 300 python tools/refit_bench.py > $O/r04d_refit.jsonl 2>/dev/null
ZR_SCENE_UPDATE=refit_sah REFIT_MOVING_FRAMES=64 timeout 300 python tools/refit_bench.py >> $O/r04d_refit.jsonl 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for l in open(O + "/r04d_refit.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["mode"], d.get("background_rebuilds"), "update", d["update_ms_median"], "static", d["frame_ms_static"], "moving", d["frame_ms_moving"], d["kernel_ms_moving"].get("rpt_pathtrace"))
PY
timeout 300 python scripts/tile_kernels.py --scene synthetic > $O/r04d_tiles_atrium.jsonl 2>/dev/null
ZR_K11=compact timeout 300 python scripts/tile_kernels.py --scene synthetic > $O/r04d_tiles_atrium_compact.jsonl 2>/dev/null
timeout 300 python scripts/tile_kernels.py > $O/r04d_tiles_cornell.jsonl 2>/dev/null
ZR_K11=compact timeout 300 python scripts/tile_kernels.py > $O/r04d_tiles_cornell_compact.jsonl 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in ("r04d_tiles_atrium", "r04d_tiles_atrium_compact", "r04d_tiles_cornell", "r04d_tiles_cornell_compact"):
    rows = [json.loads(l) for l in open(f"{O}/{f}.jsonl") if l.startswith("{")]
    if not rows: print(f, "no rows"); continue
    worst = max(rows, key=lambda r: r["sum_ms"])
    print(f, "slowest tile", worst["sum_ms"], "mean", round(sum(r["sum_ms"] for r in rows) / len(rows), 3), {k: v for k, v in worst["k"].items() if v > 0.05})
PY
