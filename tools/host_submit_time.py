"""Host time to SUBMIT one ReSTIR PT frame through the Python binding (frame constants + the ctypes calls) against the device time of the frame: if the first
approaches the second the benchmark measures the host.  python tools/host_submit_time.py [--overlap 0|1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zetaray_amd import api, scene_io, tiling, wire

ap = argparse.ArgumentParser()
ap.add_argument("--overlap", type=int, default=1)
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
W, H = 1920, 1080
t = tiling.TiledRestirPT(sc, W, H, 1, 0, params=wire.default_params())
if a.overlap:
    t.enable_frame_overlap(True)
for f in range(1, 65):
    t.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives)))
torch.cuda.synchronize()
t0 = time.perf_counter()
tc = 0.0
for f in range(65, 65 + a.frames):
    c0 = time.perf_counter()
    cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives))
    tc += time.perf_counter() - c0
    t.render_frame(cb)
t_submit = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(json.dumps({"overlap": bool(a.overlap), "frames": a.frames, "host_submit_ms_per_frame": round(t_submit / a.frames * 1e3, 4), "of_which_frame_constants_ms": round(tc / a.frames * 1e3, 4),
                  "wall_ms_per_frame": round(t_total / a.frames * 1e3, 4)}))
