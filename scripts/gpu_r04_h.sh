#!/bin/bash
# round 4, visit H: moved instances in subtrees of their own + topology-only install of background trees (zr_bvh.h Build ownSubtree, zr_api.hip
# k_install_topology): parity tests of the dynamic-scene paths, then the moving atrium with and without the grouping
mkdir -p gpurun_out
export ZR_BVH_TIMING=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "background_sah or device_refit or stream_ordered or moving" 2>&1 | tail -5 > gpurun_out/r04h_tests.log
cat gpurun_out/r04h_tests.log
rm -f gpurun_out/r04h_refit.jsonl
for g in 1 0; do
  ZR_BVH_GROUP=$g ZR_SCENE_UPDATE=refit_sah REFIT_MOVING_FRAMES=64 timeout 600 python tools/refit_bench.py 2> gpurun_out/r04h_refit_err_$g.log | tail -1 >> gpurun_out/r04h_refit.jsonl
  grep "background tree" gpurun_out/r04h_refit_err_$g.log | tail -3
done
python - <<'P'
import json
for l in open("gpurun_out/r04h_refit.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in ("mode", "group", "background_rebuilds", "update_ms_median", "update_ms_mean", "frame_ms_static", "frame_ms_moving", "kernel_ms_moving")})
    print(d["update_ms"]); print(d["frame_ms_moving_series"])
P
