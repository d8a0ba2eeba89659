"""ReSTIR PT over 2 devices on the CPU (gloo, world size 2): each rank renders its 32-px-aligned tile + 32-px apron with the
host executor, exchanges reservoir halos with torch.distributed P2P exactly as zetaray_amd/tiling.py does on RCCL
(post-temporal planes before the spatial stage, final planes after it), and the stitched radiance AND reservoir planes must
be bit-identical to the single-process full-frame run -- with a moving camera, so the temporal passes really read across
the tile border."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANES = ("A", "B", "C", "D", "E", "F", "G")


def _frames(scene_io, sc, w, h, n):
    cbs, prev = [], None
    for f in range(1, n + 1):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives),
                                           cam_pos=(0.04 * f, 1.2, -4.043 + 0.02 * f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        cbs.append(cb)
    return cbs


def _worker(rank, world, port, w, h, nframes, out_path, layout=None, spatial_passes=1):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import zro
    from tests.hostexec import zhx
    from zetaray_amd import scene_io, tiling, wire
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    o = zro.OracleScene(sc)
    hx = zhx.HostExecScene(sc, o.alias)
    tile = tiling.tile_rect(w, h, world, rank, layout)
    ext = tiling.extended_rect(w, h, tile)
    plan = tiling.halo_plan(w, h, world, rank, layout=layout)
    r = zhx.HostExecRPT(hx, ext[2], ext[3], ext=ext, owned=tile)
    prm = wire.default_params()
    prm.num_spatial_passes = spatial_passes

    def local(rect):
        return rect[0] - ext[0], rect[1] - ext[1], rect[2], rect[3]

    def exchange(which):
        for name in PLANES:
            full = r.plane(name, which)
            ops, recvs = [], []
            for peer, send, recv in plan:
                if send:
                    x, y, sw, sh = local(send)
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(full[y:y + sh, x:x + sw])), peer))
                if recv:
                    buf = torch.zeros((recv[3], recv[2], full.shape[2]), dtype=torch.from_numpy(full[:1, :1]).dtype)
                    recvs.append((recv, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for recv, buf in recvs:
                x, y, rw, rh = local(recv)
                full[y:y + rh, x:x + rw] = buf.numpy()
                r.write_plane_rect(name, which, full, (x, y, rw, rh))

    for cb in _frames(scene_io, sc, w, h, nframes):
        r.render_stage(cb, prm, 1)
        exchange(1)          # between the stages the post-temporal set is "which = 1"
        r.render_stage(cb, prm, 2)
        if spatial_passes == 2:
            exchange(1)      # the first round's outputs are the current set now: what the second round reads at neighbouring pixels
            r.render_stage(cb, prm, 4)
        exchange(0)          # after the frame: the set the next frame reads as previous
    x, y, tw, th = local(tile)
    res = {"tile": np.array(tile), "final": r.final[y:y + th, x:x + tw].copy(), "rays": np.array(r.counters)}
    for name in PLANES:
        res[name] = r.plane(name)[y:y + th, x:x + tw].copy()
    np.savez(out_path + f".{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("layout,spatial_passes", [(None, 1), ([(0, 0, 96, 64), (96, 0, 32, 64)], 1), (None, 2)],
                         ids=["equal-area grid", "uneven cost-balanced style split", "two spatial rounds (a third exchange between them)"])
def test_restir_pt_tile_split_with_halo_exchange_is_bit_identical(tmp_path, layout, spatial_passes):
    """layout None: tiling.tile_rect's grid; the second case is the kind of split tiling.balanced_layout produces (32-px-aligned tiles of
    different sizes): the halo plan and the stitched result must not depend on the tiles being equal"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h, nframes, world = 128, 64, 4, 2
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, port, w, h, nframes, out, layout, spatial_passes), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import zro
    from zetaray_amd import scene_io, wire
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    o = zro.OracleScene(sc)
    ref = zro.OracleRPT(o, w, h)
    prm = wire.default_params()
    prm.num_spatial_passes = spatial_passes
    rays = np.zeros(2, np.int64)
    for cb in _frames(scene_io, sc, w, h, nframes):
        want = ref.render(cb, prm)
    last_rays = np.array(ref.counters)
    got_rays = np.zeros(2, np.int64)
    for rank in range(world):
        d = np.load(out + f".{rank}.npz")
        x0, y0, tw, th = [int(v) for v in d["tile"]]
        assert np.array_equal(d["final"].view(np.uint32), want[y0:y0 + th, x0:x0 + tw].view(np.uint32)), f"rank {rank}: radiance differs"
        for name in PLANES:
            a, b = d[name], ref.plane(name)[y0:y0 + th, x0:x0 + tw]
            if name == "A":
                a, b = a & 0xffffff, b & 0xffffff
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"rank {rank}: reservoir plane {name} differs"
        got_rays += d["rays"]
    assert tuple(got_rays) == tuple(last_rays)


def test_halo_plan_is_symmetric_and_covers_the_apron():
    sys.path.insert(0, ROOT)
    from zetaray_amd import tiling
    W, H = 1920, 1080
    rng = np.random.default_rng(5)
    blob = np.fromfunction(lambda y, x: np.exp(-((x - 20) ** 2 + (y - 12) ** 2) / 90.0), ((H + 31) // 32, (W + 31) // 32))
    layouts = [(n, None) for n in (2, 4, 8)] + [(n, tiling.balanced_layout(W, H, n, blob + 0.01 * rng.random(blob.shape))) for n in (3, 6, 8)]
    for n, layout in layouts:
        for r in range(n):
            tile = tiling.tile_rect(W, H, n, r, layout)
            ext = tiling.extended_rect(W, H, tile)
            cover = np.zeros((H, W), np.int32)
            cover[tile[1]:tile[1] + tile[3], tile[0]:tile[0] + tile[2]] = 1
            for peer, send, recv in tiling.halo_plan(W, H, n, r, layout=layout):
                back = [(s2, r2) for p2, s2, r2 in tiling.halo_plan(W, H, n, peer, layout=layout) if p2 == r][0]
                assert back == (recv, send)
                if recv:
                    cover[recv[1]:recv[1] + recv[3], recv[0]:recv[0] + recv[2]] += 1
            assert (cover[ext[1]:ext[1] + ext[3], ext[0]:ext[0] + ext[2]] == 1).all()
