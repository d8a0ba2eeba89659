# host SAH build, one thread against the forked build: zr_scene_create time of the atrium (and that the tree is the same)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, time, sys
sys.path.insert(0, ".")
from zetaray_amd import api, scene_io
sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
os.environ["ZR_BVH_TIMING"] = "1"
for nt in ("1", "4", "16", ""):
    if nt: os.environ["ZR_BVH_THREADS"] = nt
    else: os.environ.pop("ZR_BVH_THREADS", None)
    t = time.perf_counter(); s = api.Scene(sc); dt = time.perf_counter() - t
    print("ZR_BVH_THREADS=%s: zr_scene_create %.1f ms, bvh %s" % (nt or "(default)", dt * 1e3, list(s.bvh_info())), flush=True)
    del s
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "device_refit or device_built or atrium or large" 2>&1 | grep -E "passed|failed" | tail -2
