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
    ap.add_argument("--uniform", type=float, default=None, help="tiling.choose_layout's per-pixel term (default: the library's)")
    ap.add_argument("--min-gain", type=float, default=None, help="tiling.choose_layout's threshold for leaving the equal-area grid (default: the library's 1.15)")
    ap.add_argument("--halves", action="store_true",
                    help="VERDICT r4 item 3(c): every device's tile of the equal-area split cut in two (32-px aligned, the longer side), the halves rendered "
                         "CONCURRENTLY on two streams -- one round of waves lasts as long as its slowest wave, a second independent half fills the slots its tail leaves idle")
    ap.add_argument("--overlap", action="store_true",
                    help="VERDICT r5 item 2: per tile, the THROUGHPUT of its own frames rendered back to back (no wait between the stages or frames; aprons as the lockstep "
                         "warm-up frames left them) in the plain order and with frame overlap (K1 + K11 of frame N + 1 on a second stream beside the spatial stage of frame N)")
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
        layout = tiling.choose_layout(W, H, a.world, probe.owned_cost_cells(), **({} if a.uniform is None else dict(uniform=a.uniform)), **({} if a.min_gain is None else dict(min_gain=a.min_gain)))
        probe.r.p_indirect.enable_cost_map(False)
        t0 = time.perf_counter()
        for f in range(13, 21):
            probe.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam))
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) / 8 * 1e3
        del probe
    full_ov = None
    if a.overlap:
        # the single-device frame the tiles are compared with: plain and overlapped, 16 frames back to back after 12 to settle
        probe = tiling.TiledRestirPT(sc, W, H, 1, 0, params=prm)
        res = []
        for mode in (0, 1):
            probe.enable_frame_overlap(bool(mode))
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for f in range(16):
                    probe.render_frame(scene_io.make_frame_constants(W, H, frame_num=100 + (2 * mode + rep) * 16 + f, num_emissives=len(sc.emissives), **cam))
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 16 * 1e3
            res.append(dt)
        full_ms, full_ov = res
        del probe
    if a.halves:
        halves = []
        for r in range(a.world):
            x0, y0, tw, th = tiling.tile_rect(W, H, a.world, r, layout)
            if th >= tw:
                c = max(64, min(th - 64, (th // 2 + 31) // 32 * 32))
                halves += [(x0, y0, tw, c), (x0, y0 + c, tw, th - c)]
            else:
                c = max(64, min(tw - 64, (tw // 2 + 31) // 32 * 32))
                halves += [(x0, y0, c, th), (x0 + c, y0, tw - c, th)]
        tiles = [tiling.TiledRestirPT(sc, W, H, 2 * a.world, r, params=prm, layout=halves) for r in range(2 * a.world)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        t = np.zeros((a.world, 2))
        n = 0
        for f in range(1, a.frames + 1):
            cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **cam)
            for d in range(a.world):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(2):
                    tiles[2 * d + k].stage_temporal(cb, streams[k].cuda_stream)
                torch.cuda.synchronize()
                if f > 3:
                    t[d, 0] += time.perf_counter() - t0
            tiling.exchange_in_process(tiles, api.HALO_POST_TEMPORAL)
            for d in range(a.world):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(2):
                    tiles[2 * d + k].stage_spatial(cb, streams[k].cuda_stream)
                torch.cuda.synchronize()
                if f > 3:
                    t[d, 1] += time.perf_counter() - t0
            tiling.exchange_in_process(tiles, api.HALO_FINAL)
            n += f > 3
        ms = t.sum(axis=1) / n * 1e3
        print(json.dumps({"scene": a.scene, "world": a.world, "size": [W, H], "layout": "halves on two streams", "tile_ms": [round(float(x), 3) for x in ms], "rects": halves,
                          "max_ms": round(float(ms.max()), 3), "mean_ms": round(float(ms.mean()), 3), "sum_ms": round(float(ms.sum()), 3)}))
        return
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
    if a.overlap:
        # every tile alone, M frames back to back: plain order, then overlapped (the kernels of an 8-way tile are ONE round of workgroups each: the plain order
        # pays every kernel's tail, the overlapped one fills it with the other half's workgroups)
        M = 24
        thr = np.zeros((a.world, 2))
        # one second stream for all tile objects of this process: HIP multiplexes a process's streams onto a handful of hardware queues, and with a stream per
        # tile object (8 here; a real rank has one) some tile's "second" stream lands on the queue of the null stream and overlaps nothing
        shared = torch.cuda.Stream()
        for i, r in enumerate(ranks):
            for mode in (0, 1):
                r.enable_frame_overlap(bool(mode))
                if mode:
                    r.r._overlap_stream = shared.cuda_stream
                for rep in range(2):      # (first repetition: warm-up)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(M):
                        cb = scene_io.make_frame_constants(W, H, frame_num=a.frames + 1 + (2 * mode + rep) * M + k, num_emissives=len(sc.emissives), **cam)
                        r.stage_temporal(cb)
                        r.stage_spatial(cb)
                    torch.cuda.synchronize()
                    thr[i, mode] = (time.perf_counter() - t0) / M * 1e3
            r.enable_frame_overlap(False)
        print(json.dumps({"scene": a.scene, "world": a.world, "size": [W, H], "layout": a.layout, "single_device_frame_ms": round(full_ms, 3), "single_device_frame_ms_overlapped": round(full_ov, 3),
                          "slowest_tile_bound_plain": round(full_ms / float(thr[:, 0].max()), 2), "slowest_tile_bound_overlapped": round(full_ms / float(thr[:, 1].max()), 2),
                          "slowest_tile_bound_overlapped_vs_overlapped_single": round(full_ov / float(thr[:, 1].max()), 2),
                          "tile_ms_stage_by_stage": [round(float(x), 3) for x in ms], "rects": [tiling.tile_rect(W, H, a.world, r, layout) for r in range(a.world)],
                          "tile_ms_back_to_back_plain": [round(float(x), 3) for x in thr[:, 0]], "tile_ms_back_to_back_overlapped": [round(float(x), 3) for x in thr[:, 1]],
                          "max_ms_plain": round(float(thr[:, 0].max()), 3), "max_ms_overlapped": round(float(thr[:, 1].max()), 3),
                          "sum_ms_plain": round(float(thr[:, 0].sum()), 3), "sum_ms_overlapped": round(float(thr[:, 1].sum()), 3)}))
        return
    print(json.dumps({"scene": a.scene, "world": a.world, "size": [W, H], "layout": a.layout, "layout_chosen": ("kd-split" if layout is not None else "grid"), "single_device_frame_ms": (round(full_ms, 3) if full_ms else None),
                      "tile_ms": [round(float(x), 3) for x in ms], "rects": [tiling.tile_rect(W, H, a.world, r, layout) for r in range(a.world)],
                      "max_ms": round(float(ms.max()), 3), "mean_ms": round(float(ms.mean()), 3), "sum_ms": round(float(ms.sum()), 3)}))


if __name__ == "__main__":
    main()
