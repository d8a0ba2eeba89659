"""N > 1 path on the CPU: world-size-2 gloo job.  Each rank renders its 32-px-aligned screen tile with the host
executor (same stage functions as the HIP kernels, global pixel coordinates for RNG seeds / group ids); the tiles are
gathered with torch.distributed and must stitch to the bit-identical single-process image."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import tile_rect
    from oracle import zro
    from tests.hostexec import zhx
    from zetaray_amd import scene_io, wire
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    o = zro.OracleScene(sc)
    hx = zhx.HostExecScene(sc, o.alias)
    cb = scene_io.make_frame_constants(w, h, frame_num=4, num_emissives=len(sc.emissives))
    tile = tile_rect(w, h, world, rank)
    _, planes = hx.gbuffer(cb, tile=tile)
    final, cnt = hx.pathtrace(cb, planes, wire.default_params(), tile=tile)
    # gather tiles (padded to a common shape) and ray counters on rank 0
    pad = torch.zeros((h, w, 4), dtype=torch.float32)
    pad[:tile[3], :tile[2]] = torch.from_numpy(final)
    meta = torch.tensor(list(tile) + [cnt[0], cnt[1]], dtype=torch.int64)
    tiles = [torch.zeros_like(pad) for _ in range(world)] if rank == 0 else None
    metas = [torch.zeros_like(meta) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, tiles, dst=0)
    dist.gather(meta, metas, dst=0)
    if rank == 0:
        img = np.zeros((h, w, 4), np.float32)
        rays = np.zeros(2, np.int64)
        for t, m in zip(tiles, metas):
            x0, y0, tw, th, nc, ns = [int(v) for v in m]
            img[y0:y0 + th, x0:x0 + tw] = t.numpy()[:th, :tw]
            rays += [nc, ns]
        np.savez(out_path, img=img, rays=rays)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_split_is_bit_identical(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w, h = 160, 96
    out = str(tmp_path / "stitched.npz")
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    got = np.load(out)
    sys.path.insert(0, ROOT)
    from oracle import zro
    from zetaray_amd import scene_io, wire
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    o = zro.OracleScene(sc)
    cb = scene_io.make_frame_constants(w, h, frame_num=4, num_emissives=len(sc.emissives))
    _, planes = o.gbuffer(cb)
    want, cnt = o.pathtrace(cb, planes, wire.default_params())
    assert np.array_equal(got["img"].view(np.uint32), want.view(np.uint32))
    assert tuple(got["rays"]) == tuple(cnt)


def test_tile_rects_cover_the_frame():
    from bench import tile_rect
    for n in (1, 2, 4, 8):
        cover = np.zeros((1080, 1920), np.int32)
        for r in range(n):
            x0, y0, tw, th = tile_rect(1920, 1080, n, r)
            assert x0 % 32 == 0 and y0 % 32 == 0 and tw > 0 and th > 0
            cover[y0:y0 + th, x0:x0 + tw] += 1
        assert (cover == 1).all()


def test_balanced_layout_partitions_the_frame():
    """tiling.balanced_layout: n 32-px-aligned rects that tile the frame, balanced cost, symmetric halo plans (both sides of an exchange
    derive the same strips), for even and odd rank counts and for a map with empty regions (the Cornell box at 16:9)"""
    import numpy as np
    from zetaray_amd import tiling
    W, H = 1920, 1080
    gh, gw = (H + 31) // 32, (W + 31) // 32
    rng = np.random.default_rng(3)
    cost = np.zeros((gh, gw))
    cost[:, 14:46] = 1.0 + rng.random((gh, 32))
    cost[8:26, 24:36] *= 4.0
    for n in (2, 3, 4, 6, 8):
        L = tiling.balanced_layout(W, H, n, cost)
        cover = np.zeros((H, W), np.int32)
        for x0, y0, tw, th in L:
            assert x0 % 32 == 0 and y0 % 32 == 0 and tw >= 64 and th >= 64
            cover[y0:y0 + th, x0:x0 + tw] += 1
        assert (cover == 1).all()
        share = np.array([cost[y0 // 32:(y0 + th + 31) // 32, x0 // 32:(x0 + tw + 31) // 32].sum() for x0, y0, tw, th in L]) / cost.sum()
        assert share.max() * n < 1.25, (n, share)
        eq = np.array([cost[y0 // 32:(y0 + th + 31) // 32, x0 // 32:(x0 + tw + 31) // 32].sum()
                       for x0, y0, tw, th in [tiling.tile_rect(W, H, n, r) for r in range(n)]]) / cost.sum() if n in (2, 4, 8) else None
        if eq is not None and n == 8:
            assert share.max() < eq.max()          # better than the equal-area grid on this map
        for r in range(n):
            for peer, send, recv in tiling.halo_plan(W, H, n, r, layout=L):
                back = [(s2, r2) for p, s2, r2 in tiling.halo_plan(W, H, n, peer, layout=L) if p == r]
                assert back and back[0] == (recv, send)


def test_final_halo_policy_decision():
    """tiling.frame_reads_history_across_tiles: the post-frame reservoir halo is only needed by a ReSTIR PT frame whose camera or scene moved"""
    from zetaray_amd import scene_io, tiling
    a = scene_io.make_frame_constants(256, 128, frame_num=3, num_emissives=2)
    assert not tiling.frame_reads_history_across_tiles("restir_pt", a, False)
    assert tiling.frame_reads_history_across_tiles("restir_pt", a, True)                     # an instance, a light or a material changed
    assert tiling.frame_reads_history_across_tiles("restir_gi", a, False)                    # GI draws temporal candidates from a neighbourhood
    b = scene_io.make_frame_constants(256, 128, frame_num=4, num_emissives=2, cam_pos=(0.1, 1.2, -4.0))
    b["prev_view"], b["prev_view_inv"] = a["curr_view"], a["curr_view_inv"]
    assert tiling.frame_reads_history_across_tiles("restir_pt", b, False)                    # the camera moved
    c = scene_io.make_frame_constants(256, 128, frame_num=5, num_emissives=2, jitter=(0.25, -0.25))
    c["prev_camera_jitter"] = np.float32([0.0, 0.0])
    assert tiling.frame_reads_history_across_tiles("restir_pt", c, False)                    # only the jitter changed: still a different reprojection


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` without WORLD_SIZE in the environment (the shape of the driver's N = 1 command) re-executes itself as N ranks under
    torch.distributed.run on the loopback address instead of asserting; with WORLD_SIZE set (the driver's torchrun launch) it does not.  The launch
    itself needs GPUs (tests/test_gpu_parity.py::test_bench_multi_rank_protocol_on_one_gpu runs it); here the command it would run is shown."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ZR_BENCH_LAUNCH_ECHO"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    cmd = json.loads(res.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_tiled_denoise_pass_lives_on_the_tile_device():
    """TiledRestirPT.enable_denoise must create the denoise pass on the rank's own device (like the direct-lighting passes), or every
    rank with local_rank != 0 gets a pass on device 0 next to a scene and G-buffer on local_rank.  No GPU here: the renderer is a recorder."""
    sys.path.insert(0, ROOT)
    from zetaray_amd import tiling

    class _Dev:
        index = 3

    class _Rec:
        gbuffer = None
        p_denoise = "set"

        def enable_denoise(self, prm, device=0):
            self.seen = device
            return "pass"

    t = tiling.TiledRestirPT.__new__(tiling.TiledRestirPT)
    t.r, t.device, t.world, t.native = _Rec(), _Dev(), 1, None
    t.api = type("A", (), {"STAGE_DENOISE_MASK": 0})
    assert t.enable_denoise() == "pass" and t.r.seen == 3 and t.r.p_denoise is None
