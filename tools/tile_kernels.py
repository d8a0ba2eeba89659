"""per-kernel GPU time of each rank's tile of the 8-way split (one device, tiles rendered in turn, halos exchanged in process): where a small tile's
time goes.  python tools/tile_kernels.py [--scene synthetic]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zetaray_amd import api, scene_io, tiling, wire

ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="cornell"); ap.add_argument("--world", type=int, default=8); a = ap.parse_args()
W, H = 1920, 1080
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prm = wire.default_params()
if a.scene == "synthetic":
    sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium"); prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512; cam = dict(cam_pos=(0, 0, -3.5))
else:
    sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz")); cam = {}
ranks = [tiling.TiledRestirPT(sc, W, H, a.world, r, params=prm) for r in range(a.world)]
for r in ranks:
    r.r.p_gbuffer.enable_timing(True); r.r.p_indirect.enable_timing(True)
acc = [dict() for _ in ranks]
for f in range(1, 25):
    cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam)
    tiling.render_frame_in_process(ranks, cb)
    torch.cuda.synchronize()
    if f > 12:
        for i, r in enumerate(ranks):
            for name, (ms, n) in {**r.r.p_gbuffer.timings(), **r.r.p_indirect.timings()}.items():
                acc[i].setdefault(name, []).append(ms)
for i, d in enumerate(acc):
    k = {n: round(float(np.mean(v)), 3) for n, v in d.items()}
    print(json.dumps({"rank": i, "tile": ranks[i].tile, "sum_ms": round(sum(k.values()), 3), "k": k}))
