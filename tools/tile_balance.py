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
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--layout", choices=["equal", "cost"], default="equal",
                    help="cost: the kd-split of tiling.balanced_layout on the per-cell ray counts of 12 full-frame probe frames")
    a = ap.parse_args()
    W, H = a.width, a.height
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prm = wire.default_params()
    if a.scene == "synthetic":
        sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
        prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
        cam = dict(cam_pos=(0, 0, -3.5))
    else:
        sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
        cam = {}
    layout = None
    full_ms = None
    if a.layout == "cost":
        probe = tiling.TiledRestirPT(sc, W, H, 1, 0, params=prm)
        probe.r.p_indirect.enable_cost_map(True)
        for f in range(1, 13):
            probe.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam))
        torch.cuda.synchronize()
        layout = tiling.choose_layout(W, H, a.world, probe.owned_cost_cells())
        probe.r.p_indirect.enable_cost_map(False)
        t0 = time.perf_counter()
        for f in range(13, 21):
            probe.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam))
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) / 8 * 1e3
        del probe
    ranks = [tiling.TiledRestirPT(sc, W, H, a.world, r, params=prm, layout=layout) for r in range(a.world)]
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
    print(json.dumps({"scene": a.scene, "world": a.world, "size": [W, H], "layout": a.layout, "layout_chosen": ("kd-split" if layout is not None else "grid"), "single_device_frame_ms": (round(full_ms, 3) if full_ms else None),
                      "tile_ms": [round(float(x), 3) for x in ms], "rects": [tiling.tile_rect(W, H, a.world, r, layout) for r in range(a.world)],
                      "max_ms": round(float(ms.max()), 3), "mean_ms": round(float(ms.mean()), 3), "sum_ms": round(float(ms.sum()), 3)}))


if __name__ == "__main__":
    main()
