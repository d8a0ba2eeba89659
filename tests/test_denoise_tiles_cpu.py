"""The denoise pass on screen tiles (BASELINE config 5: "ReSTIR PT + SVGF denoise tile pass, 8 x MI355X"): every device filters its tile + 32-px apron
and exchanges halos where the next step's stencil would otherwise read something inexact (zetaray_amd/tiling.py denoise_schedule; the a-trous
stencil of the fifth iteration alone reaches 32 px).  Here on the CPU with the pass's stage functions run by the host executor (tests/hostexec: the
same zr_svgf.h the kernels compile): the stitched output, colour history and moments of 4 tiles (one process) and of 2 ranks over gloo must be the
FULL-FRAME ORACLE's (oracle/zro_svgf.h), bit for bit, over frames with sideways motion (the temporal step reads history across tile borders) and a
history reset.  The apron's signal and history are garbage until the exchange brings the owners' values -- as on a device, where the indirect
pass shades owned pixels only."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLT_MAX = np.float32(3.4028234663852886e38)
ITER = 5


def _inputs(W, H, nframes, seed=11):
    """full-frame planes, identical on every rank: a G-buffer with a depth / normal edge and a block of misses, noisy frames, 1.5-px motion from frame 2"""
    sys.path.insert(0, ROOT)
    from tests.test_denoise import _planes
    rng = np.random.default_rng(seed)
    depth, normal = _planes(H, W, rng)
    mv = np.uint32(int(round(1.5 / W * 32767.0)) & 0xffff) | (np.uint32(int(round(-0.5 / H * 32767.0)) & 0xffff) << np.uint32(16))
    frames = []
    for f in range(nframes):
        sig = np.zeros((H, W, 4), np.float32)
        sig[..., :3] = rng.uniform(0.0, 3.0, (H, W, 3)).astype(np.float32) * np.where(np.arange(W)[None, :, None] < W // 2, 1.0, 0.2).astype(np.float32)
        if f == 1:
            sig[H // 2, W // 2 - 1, 1] = np.inf          # a firefly that overflowed: sanitised, must not poison the history
        frames.append((sig, np.full((H, W), mv if f >= 2 else 0, np.uint32), f not in (0, 4)))
    return depth, normal, frames


class _Tile:
    """one device's share: window = tile + apron, its denoise state, and the strips it trades"""

    def __init__(self, W, H, world, rank, layout=None):
        from tests.hostexec import zhx
        from zetaray_amd import tiling
        self.tile = tiling.tile_rect(W, H, world, rank, layout)
        self.ext = tiling.extended_rect(W, H, self.tile)
        self.plan = tiling.halo_plan(W, H, world, rank, layout=layout)
        self.dn = zhx.HostExecDenoise(W, H, window=self.ext, iterations=ITER)
        self.sig = None

    def cut(self, full):
        x, y, w, h = self.ext
        return np.ascontiguousarray(full[y:y + h, x:x + w])

    def begin_frame(self, sig_full, rng):
        """the window's signal: the owned tile's pixels, garbage in the apron (nobody shaded it here)"""
        x, y, w, h = self.ext
        self.sig = rng.uniform(50.0, 90.0, (h, w, 4)).astype(np.float32)
        tx, ty, tw, th = self.tile
        self.sig[ty - y:ty - y + th, tx - x:tx - x + tw] = sig_full[ty:ty + th, tx:tx + tw]

    def planes_of(self, which):
        """the arrays an exchange moves (window-sized), in the order of zr_api.hip HaloPlanes"""
        from zetaray_amd import api
        if which == api.HALO_DENOISE_INPUT:
            return [("signal", self.sig), ("history", self.dn.plane("history")), ("moments", self.dn.plane("moments"))]
        return [("iter", self.dn.plane("iter"))]

    def local(self, rect):
        return rect[0] - self.ext[0], rect[1] - self.ext[1], rect[2], rect[3]

    def put(self, name, full, rect_local):
        x, y, w, h = rect_local
        if name == "signal":
            self.sig[y:y + h, x:x + w] = full[y:y + h, x:x + w]
        else:
            self.dn.write_plane_rect(name, full, rect_local)

    def owned(self, arr):
        x, y, w, h = self.local(self.tile)
        return arr[y:y + h, x:x + w].copy()


def _run_schedule(tiles, depth, normal, motion, prev, valid, exchange):
    from zetaray_amd import tiling
    for kind, v in tiling.denoise_schedule(ITER):
        if kind == "exchange":
            exchange(v)
        else:
            for t in tiles:
                t.dn.render(t.sig, t.cut(depth), t.cut(normal), t.cut(motion), t.cut(prev[0]), t.cut(prev[1]), temporal_valid=valid, steps=v)


def _oracle_frames(W, H, depth, normal, frames):
    from oracle import zro
    hc, hm = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 2), np.float32)
    outs = []
    for sig, motion, valid in frames:
        o, hc, hm = zro.svgf(sig, depth, normal, motion, depth, normal, hc, hm, temporal_valid=valid, iterations=ITER)
        outs.append((o, hc.copy(), hm.copy()))
    return outs


@pytest.mark.parametrize("world,layout", [(4, None), (3, [(0, 0, 96, 96), (96, 0, 64, 64), (96, 64, 64, 32)])], ids=["2 x 2 grid", "uneven split"])
def test_denoise_on_tiles_in_one_process_equals_the_full_frame_oracle(world, layout):
    sys.path.insert(0, ROOT)
    W, H, nframes = 160, 96, 6
    depth, normal, frames = _inputs(W, H, nframes)
    want = _oracle_frames(W, H, depth, normal, frames)
    tiles = [_Tile(W, H, world, r, layout) for r in range(world)]
    rng = np.random.default_rng(3)

    def exchange(which):
        sent = {(t_i, peer): [(name, arr[y:y + h, x:x + w].copy()) for name, arr in t.planes_of(which) for (x, y, w, h) in [t.local(send)]]
                for t_i, t in enumerate(tiles) for peer, send, recv in t.plan if send}
        for t_i, t in enumerate(tiles):
            for peer, send, recv in t.plan:
                if recv:
                    x, y, w, h = t.local(recv)
                    for (name, strip), (_, mine) in zip(sent[(peer, t_i)], t.planes_of(which)):
                        full = mine.copy()
                        full[y:y + h, x:x + w] = strip
                        t.put(name, full, (x, y, w, h))

    for f, (sig, motion, valid) in enumerate(frames):
        for t in tiles:
            t.begin_frame(sig, rng)
        _run_schedule(tiles, depth, normal, motion, (depth, normal), valid, exchange)
        o, hc, hm = want[f]
        for r, t in enumerate(tiles):
            x, y, w, h = t.tile
            for name, full in (("out", o), ("history", hc), ("moments", hm)):
                got = t.owned(t.dn.plane(name))
                assert np.array_equal(got.view(np.uint32), full[y:y + h, x:x + w].view(np.uint32)), f"frame {f} tile {r}: {name}"
    assert np.isfinite(want[-1][0]).all()


def _worker(rank, world, port, W, H, nframes, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    depth, normal, frames = _inputs(W, H, nframes)
    t = _Tile(W, H, world, rank)
    rng = np.random.default_rng(100 + rank)

    def exchange(which):
        for idx in range(len(t.planes_of(which))):
            name, arr = t.planes_of(which)[idx]
            ops, recvs = [], []
            for peer, send, recv in t.plan:
                if send:
                    x, y, w, h = t.local(send)
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(arr[y:y + h, x:x + w])), peer))
                if recv:
                    buf = torch.zeros((recv[3], recv[2], arr.shape[2]), dtype=torch.float32)
                    recvs.append((recv, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for recv, buf in recvs:
                x, y, w, h = t.local(recv)
                full = arr.copy()
                full[y:y + h, x:x + w] = buf.numpy()
                t.put(name, full, (x, y, w, h))

    res = {"tile": np.array(t.tile)}
    for f, (sig, motion, valid) in enumerate(frames):
        t.begin_frame(sig, rng)
        _run_schedule([t], depth, normal, motion, (depth, normal), valid, exchange)
        for name in ("out", "history", "moments"):
            res[f"{name}{f}"] = t.owned(t.dn.plane(name))
    np.savez(out_path + f".{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


def test_denoise_on_two_ranks_over_gloo_equals_the_full_frame_oracle(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H, nframes, world = 128, 64, 5, 2
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, port, W, H, nframes, out), nprocs=world, join=True)
    depth, normal, frames = _inputs(W, H, nframes)
    want = _oracle_frames(W, H, depth, normal, frames)
    for rank in range(world):
        d = np.load(out + f".{rank}.npz")
        x, y, w, h = [int(v) for v in d["tile"]]
        for f in range(nframes):
            for name, full in zip(("out", "history", "moments"), want[f]):
                assert np.array_equal(d[f"{name}{f}"].view(np.uint32), full[y:y + h, x:x + w].view(np.uint32)), f"rank {rank} frame {f}: {name}"


def test_denoise_schedule_margins():
    """the schedule never lets a step read beyond what is exact: variance 3 px, a-trous i 2 * 2^i px, an exchange restores the whole 32-px apron"""
    sys.path.insert(0, ROOT)
    from zetaray_amd import api, tiling
    for it in range(0, 6):
        margin, seen = 0, 0
        for kind, v in tiling.denoise_schedule(it):
            if kind == "exchange":
                margin = 32
                continue
            if v & api.STAGE_DENOISE_VARIANCE:
                assert v & api.STAGE_DENOISE_TEMPORAL
                margin -= 3
            for i in range(it):
                if v & api.stage_denoise_atrous(i):
                    assert i == seen
                    margin -= 2 << i
                    seen += 1
            assert margin >= 0, (it, v, margin)
        assert seen == it
    with pytest.raises(ValueError):
        tiling.denoise_schedule(6)
