# Round-4 GPU visit E: background SAH rebuild (parity + the moving-instance frame cost on the atrium, refit alone next to it), the dynamic-scene and
# denoise tests, the default bench line with both extra workloads.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_denoise.py tests/test_gpu_parity.py -m gpu -q -k "denoise or background_sah or device_refit or stream_ordered or device_built or moving" > $O/r04e_tests.log 2>&1; tail -6 $O/r04e_tests.log
REFIT_MOVING_FRAMES=64 timeout 300 python tools/refit_bench.py > $O/r04e_refit.jsonl 2>/dev/null
ZR_SCENE_UPDATE=refit_sah REFIT_MOVING_FRAMES=64 timeout 300 python tools/refit_bench.py >> $O/r04e_refit.jsonl 2>/dev/null
ZR_SCENE_UPDATE=rebuild_host REFIT_MOVING_FRAMES=12 timeout 300 python tools/refit_bench.py >> $O/r04e_refit.jsonl 2>/dev/null
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for l in open(O + "/r04e_refit.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["mode"], d.get("background_rebuilds"), "update", d["update_ms_median"], "static", d["frame_ms_static"], "moving", d["frame_ms_moving"], d["kernel_ms_moving"].get("rpt_pathtrace"))
PY
timeout 900 python bench.py > $O/r04e_bench_default.json 2> $O/r04e_bench_default.err; tail -3 $O/r04e_bench_default.err
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
d = json.loads(open(O + "/r04e_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"])
for x in d.get("extra_workloads", []):
    print("   extra", x["preset"], x["ms_per_step"], x["value"], x["roofline"]["kernel"], x["roofline"]["frac"], {k: v for k, v in x["roofline"]["kernel_ms_per_frame"].items() if v > 0.2}, x["cpu_baseline"])
PY
