"""Per-tile GPU time of the N-way screen split, measured on ONE device (every rank's tile rendered in turn, halos exchanged in process): how well
balanced the tiles are bounds what `bench.py --gpus N` can reach.  Prints one JSON line.  python tools/tile_balance.py [--scene synthetic] [--world 8]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zetaray_amd import api, scene_io, tiling, wire


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames", type=int, default=10)
    a = ap.parse_args()
    W, H = 1920, 1080
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prm = wire.default_params()
    if a.scene == "synthetic":
        sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
        prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
        cam = dict(cam_pos=(0, 0, -3.5))
    else:
        sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
        cam = {}
    ranks = [tiling.TiledRestirPT(sc, W, H, a.world, r, params=prm) for r in range(a.world)]
    t = np.zeros((a.world, 2))
    n = 0
    for f in range(1, a.frames + 1):
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam)
        for i, r in enumerate(ranks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.stage_temporal(cb)
            torch.cuda.synchronize()
            if f > 3:
                t[i, 0] += time.perf_counter() - t0
        tiling.exchange_in_process(ranks, api.HALO_POST_TEMPORAL)
        for i, r in enumerate(ranks):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.stage_spatial(cb)
            torch.cuda.synchronize()
            if f > 3:
                t[i, 1] += time.perf_counter() - t0
        tiling.exchange_in_process(ranks, api.HALO_FINAL)
        n += f > 3
    ms = (t.sum(axis=1) / n * 1e3)
    print(json.dumps({"scene": a.scene, "world": a.world, "tile_ms": [round(float(x), 3) for x in ms], "rects": [tiling.tile_rect(W, H, a.world, r) for r in range(a.world)],
                      "max_ms": round(float(ms.max()), 3), "mean_ms": round(float(ms.mean()), 3), "sum_ms": round(float(ms.sum()), 3)}))


if __name__ == "__main__":
    main()
