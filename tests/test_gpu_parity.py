"""GPU parity tests proper: the HIP path through the C-ABI vs the CPU oracle, bit-exact (ABI arithmetic contract,
include/zr_detmath.h).  Run on the GPU box with -m gpu."""
import os

import numpy as np
import pytest

from zetaray_amd import scene_io, wire

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from zetaray_amd import api
    assert api.device_count() >= 1, "no HIP device visible"
    return api


def _frame(scene, w, h, frame=1, **kw):
    return scene_io.make_frame_constants(w, h, frame_num=frame, num_emissives=len(scene.emissives), **kw)


@pytest.mark.parametrize("w,h", [(64, 64), (200, 120)])
def test_gbuffer_bit_exact(api, cornell_emissive, oracle_emissive, w, h):
    cb = _frame(cornell_emissive, w, h)
    r = api.Renderer(cornell_emissive, w, h)
    r.p_gbuffer.render(cb, r.scene, r.gbuffer)
    got, _ = r.gbuffer.download()
    want, _ = oracle_emissive.gbuffer(cb)
    for name, a, b in zip(wire.GB_PLANE_NAMES, got, want):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"G-buffer plane {name} differs"


def test_alias_table_bit_exact(api, cornell_emissive, oracle_emissive):
    cb = _frame(cornell_emissive, 32, 32)
    r = api.Renderer(cornell_emissive, 32, 32)
    r.p_prelight.render(cb, r.scene)
    got = r.scene.get_alias_table()
    assert np.array_equal(got.view(np.uint8), oracle_emissive.alias.view(np.uint8))


@pytest.mark.parametrize("w,h,frame", [(64, 64, 1), (160, 96, 7)])
def test_path_tracer_bit_exact(api, cornell_emissive, oracle_emissive, w, h, frame):
    """K1 + K2 + K9 through the C-ABI: radiance bit-exact (tolerance 0), ray counters equal."""
    cb = _frame(cornell_emissive, w, h, frame)
    prm = wire.default_params()
    r = api.Renderer(cornell_emissive, w, h, params=prm)
    r.render_frame(cb)
    got = r.final()
    n_closest, n_shadow = r.p_indirect.read_counters()
    _, planes = oracle_emissive.gbuffer(cb)
    want, cnt = oracle_emissive.pathtrace(cb, planes, prm)
    assert not np.isnan(got).any()
    mism = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert mism == 0, f"{mism} radiance floats differ, max abs {np.abs(got - want).max()}"
    assert (n_closest, n_shadow) == (cnt[0], cnt[1])


def test_accumulation_semantics(api, cornell_emissive, oracle_emissive):
    """Accumulate && CameraStatic: FINAL += li (PathTracer.hlsl:205-211); linearity of the accumulated image."""
    w = h = 48
    prm = wire.default_params()
    r = api.Renderer(cornell_emissive, w, h, params=prm)
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(1, 4):
        cb = _frame(cornell_emissive, w, h, f, accumulate=1, camera_static=1, num_frames_static=f)
        r.render_frame(cb)
        _, planes = oracle_emissive.gbuffer(cb)
        acc, _ = oracle_emissive.pathtrace(cb, planes, prm, final=acc)
    got = r.final()
    assert np.array_equal(got.view(np.uint32), acc.view(np.uint32))


def test_trace_closest_matches_oracle(api, cornell_emissive, oracle_emissive):
    import torch
    rng = np.random.default_rng(7)
    n = 20000
    o = rng.uniform([-1, 0.05, -1], [1, 2, 1], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.zeros((n, 1), np.float32), d, np.full((n, 1), 3.0e38, np.float32)], 1).astype(np.float32)
    want = oracle_emissive.trace_closest(rays)
    sc = api.Scene(cornell_emissive)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    api._check(api.lib().zr_trace_closest(sc.h, None, d_rays.data_ptr(), n, 3, d_hits.data_ptr()))
    torch.cuda.synchronize()
    got = d_hits.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)


def test_full_resolution_properties(api, cornell_emissive):
    """1920x1080 (BASELINE config size): determinism (two renders identical), no NaN, miss pixels exactly zero."""
    w, h = 1920, 1080
    cb = _frame(cornell_emissive, w, h, 3)
    r = api.Renderer(cornell_emissive, w, h)
    r.render_frame(cb)
    a = r.final()
    r.render_frame(cb)
    b = r.final()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.isnan(a).any() and (a >= 0).all()
    planes, _ = r.gbuffer.download()
    miss = planes[7] > 1e30
    assert (a[miss] == 0).all()
    n_closest, n_shadow = r.p_indirect.read_counters()
    assert n_closest > w * h and n_shadow > 0


def test_errors_are_loud(api, cornell_emissive):
    r = api.Renderer(cornell_emissive, 32, 32)
    cb = _frame(cornell_emissive, 16, 16)
    with pytest.raises(api.ZetaRayError):
        r.p_gbuffer.render(cb, r.scene, r.gbuffer)         # G-buffer tile larger than the render target
    p = wire.default_params()
    p.max_non_tr_bounces = 99
    with pytest.raises(api.ZetaRayError):
        r.p_indirect.set_params(p)                          # invalid parameter -> explicit error, never silent
    sky_di = api.Pass(api.PASS_DI_SKY, 32, 32)
    with pytest.raises(api.ZetaRayError):
        sky_di.render(_frame(cornell_emissive, 32, 32), r.scene, r.gbuffer)   # no sky-view LUT bound -> explicit error


def test_russian_roulette_and_materials_on_gpu(api):
    """Synthetic scene (metal / coat / glass / thin-walled), RR active: HIP == oracle bit for bit."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    for frame, nb, gb in [(1, 3, 4), (3, 6, 8)]:
        prm = wire.default_params()
        prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gb
        w, h = 96, 64
        cb = scene_io.make_frame_constants(w, h, frame_num=frame, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
        r = api.Renderer(sc, w, h, params=prm)
        r.render_frame(cb)
        got = r.final()
        cnt_gpu = r.p_indirect.read_counters()
        ga, planes = o.gbuffer(cb)
        gg, _ = r.gbuffer.download()
        for name, a, b in zip(wire.GB_PLANE_NAMES, ga, gg):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
        want, cnt = o.pathtrace(cb, planes, prm)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert tuple(cnt_gpu) == tuple(cnt)


def test_tile_split_on_gpu(api, cornell_emissive, oracle_emissive):
    """Two 32-px-aligned tiles rendered by separate Renderer objects stitch to the single-tile image."""
    w, h = 128, 64
    cb = _frame(cornell_emissive, w, h, 2)
    _, planes = oracle_emissive.gbuffer(cb)
    want, _ = oracle_emissive.pathtrace(cb, planes, wire.default_params())
    img = np.zeros_like(want)
    for x0 in (0, 64):
        r = api.Renderer(cornell_emissive, 64, h, tile_origin=(x0, 0))
        r.render_frame(cb)
        img[:, x0:x0 + 64] = r.final()
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))


RPT_PLANES = ("A", "B", "C", "D", "E", "F", "G", "neighbor", "map_ctn", "map_ntc")      # reservoirs, K15 neighbour, K12 thread maps


def _rpt_compare(api, scene, oscene, w, h, prm, frames, cam=None, reset_at=None, overlap=False, cam_of_frame=None):
    from oracle import zro
    r = api.Renderer(scene, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    if overlap:
        r.enable_frame_overlap(True, carry=True)      # (carry: the unused bytes of the reservoir records too, so whole planes compare equal)
    o = zro.OracleRPT(oscene, w, h)
    prev = None
    for f in range(1, frames + 1):
        cb = _frame(scene, w, h, f, **(cam_of_frame(f) if cam_of_frame else (cam or {})))
        if cam_of_frame is not None:
            if prev is not None:
                cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
            prev = cb.copy()
        if prm.presampling:
            oscene.presample(f, int(prm.num_sample_sets), int(prm.sample_set_size))
        if len(scene.emissives) == 0:
            oscene.sky_lut(cb, 256, 128)      # sun + sky lighting samples the sky-view LUT (K17)
        if reset_at == f:
            r.p_indirect.reset_temporal(); o.reset_temporal()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        want = o.render(cb, prm)
        assert r.p_indirect.read_counters() == o.counters, f"frame {f}: ray counters differ"
        assert not np.isnan(got).any()
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ, max abs {np.abs(got - want).max()}"
        for nm in RPT_PLANES:
            a, b = r.p_indirect.download_plane(nm), o.plane(nm)
            if nm == "A":
                a, b = a & 0xffffff, b & 0xffffff
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {f}: reservoir plane {nm} differs"
    return got


@pytest.mark.parametrize("w,h", [(64, 48), (200, 120)])
def test_restir_pt_bit_exact(api, cornell_emissive, oracle_emissive, w, h):
    """K11 + K13-K16 through the C-ABI over 4 frames (initial candidates, temporal, spatial, boiling suppression):
    radiance and the persistent reservoir planes bit-exact vs the oracle."""
    got = _rpt_compare(api, cornell_emissive, oracle_emissive, w, h, wire.default_params(), 4, reset_at=4)
    assert got[..., :3].max() > 0


def _overlap_digest(api, scene, w, h, prm, frames, overlap, cam_of_frame=None, denoise=False, toggle_at=(), carry=True):
    """sha1 per frame over FINAL, every reservoir plane, the target plane (and the denoised image) of a ReSTIR PT sequence; and over FINAL alone"""
    import hashlib
    r = api.Renderer(scene, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    if denoise:
        r.enable_denoise()
    if overlap:
        r.enable_frame_overlap(True, carry)
    out, fin, prev = [], [], None
    for f in range(1, frames + 1):
        if f in toggle_at:
            overlap = not overlap
            r.enable_frame_overlap(overlap, carry)
        cb = _frame(scene, w, h, f, **(cam_of_frame(f) if cam_of_frame else {}))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb)
        hh = hashlib.sha1(r.final().tobytes())
        fin.append(hh.hexdigest())
        for nm in ("B", "C", "D", "E", "F", "G", "target"):
            hh.update(r.p_indirect.download_plane(nm).tobytes())
        hh.update((r.p_indirect.download_plane("A") & 0xffffff).tobytes())
        if denoise:
            hh.update(r.p_denoise.download_plane("denoised").tobytes())
        out.append(hh.hexdigest())
    return out, fin, r.p_indirect.read_counters()


def test_frame_overlap_changes_nothing(api, cornell_emissive, oracle_emissive, cornell_sky):
    """Frame overlap (zr_pass_set_frame_overlap: the G-buffer, PreLighting and K11 of frame N + 1 on a second stream beside K15 / K12 / K13 / K16 of frame N, a
    third reservoir set, second target / FINAL planes, event-ordered stages) against the oracle and against the plain order:
      * small frames vs the CPU oracle with the switch on: moving camera + a temporal reset, num_spatial_passes 0 / 1 / 2, sun + sky lighting (the sky LUT is
        rendered before either half), a materials scene with presampled light sets (K3 every frame on the first half's stream);
      * 1920 x 1080, where the two halves really run side by side (each alone fills the device): 6 frames with the camera moving from frame 4, every plane
        of every frame hashes to the digest of the plain order; the same with the denoise pass consuming each frame, and with the switch flipped on and
        off in the middle of the sequence (the plane roles carry over); ray counters identical.
    These run ZR_FRAME_OVERLAP_CARRY (whole planes comparable).  The product mode leaves the UNUSED bytes of the reservoir records to the rotation of the three
    sets: its FINAL image and ray counters over 12 frames equal the plain order's."""
    from oracle import zro
    cam = lambda f: dict(cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043))
    _rpt_compare(api, cornell_emissive, oracle_emissive, 200, 120, wire.default_params(), 5, reset_at=4, overlap=True, cam_of_frame=cam)
    for nsp in (0, 2):
        prm = wire.default_params()
        prm.num_spatial_passes = nsp
        _rpt_compare(api, cornell_emissive, oracle_emissive, 96, 64, prm, 4, overlap=True, cam_of_frame=cam)
    _rpt_compare(api, cornell_sky, zro.OracleScene(cornell_sky), 96, 64, wire.default_params(), 3, overlap=True, cam=dict(cam_pos=(0.0, 1.2, -4.043)))
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 16, 64
    _rpt_compare(api, sc, zro.OracleScene(sc, force_bvh=True), 96, 64, prm, 3, overlap=True, cam=dict(cam_pos=(0, 0, -3.5)))
    # at size: overlapped == plain, frame by frame
    w, h = 1920, 1080
    cam = lambda f: dict(cam_pos=(0.02 * max(0, f - 3), 1.2, -4.043))
    ref = {}
    for dn in (False, True):
        plain, fin0, c0 = _overlap_digest(api, cornell_emissive, w, h, wire.default_params(), 6, False, cam, denoise=dn)
        over, _, c1 = _overlap_digest(api, cornell_emissive, w, h, wire.default_params(), 6, True, cam, denoise=dn)
        assert plain == over and c0 == c1, (dn, [a == b for a, b in zip(plain, over)], c0, c1)
        ref[dn] = (plain, fin0, c0)
    mixed, _, c2 = _overlap_digest(api, cornell_emissive, w, h, wire.default_params(), 6, False, cam, toggle_at=(3, 5))
    assert mixed == ref[False][0] and c2 == ref[False][2], [a == b for a, b in zip(mixed, ref[False][0])]
    # the product mode (no carry): the bytes a reservoir record does not use may differ, nothing else -- FINAL of every frame and the ray counters are the plain order's
    # (a used byte that differed would change a later frame's radiance: the reservoirs of frame N are the temporal and spatial inputs of frames N + 1 ...)
    plain12, fin12, c12 = _overlap_digest(api, cornell_emissive, w, h, wire.default_params(), 12, False, cam)
    _, finp, cp = _overlap_digest(api, cornell_emissive, w, h, wire.default_params(), 12, True, cam, carry=False)
    assert finp == fin12 and cp == c12, ([a == b for a, b in zip(finp, fin12)], cp, c12)


def _back_to_back(api, scene, w, h, prm, frames, mode, cam_of_frame, mover=None, light=None):
    """`frames` ReSTIR PT frames submitted WITHOUT a host wait in between (what bench.py times; the digests of _overlap_digest wait for every frame, so there
    the two halves of consecutive frames never actually share the device), optionally with an instance moved before every frame from the third on;
    mode: None = plain order, "carry" / "product" = zr_pass_set_frame_overlap.  Returns the digest of the last frame's planes, of its FINAL alone, the ray counters."""
    import hashlib
    import copy
    sc = copy.copy(scene)       # the arrays the sequence rewrites are this run's own (the fixture is shared)
    sc.instances, sc.instance_to_world, sc.emissives = scene.instances.copy(), scene.instance_to_world.copy(), scene.emissives.copy()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    if mode is not None:
        r.enable_frame_overlap(True, mode == "carry")
    prev, xf = None, {}
    t0 = sc.instances["translation"][mover].copy() if mover is not None else None
    l0 = sc.instances["translation"][light].copy() if light is not None else None
    for f in range(1, frames + 1):
        if mover is not None and f >= 3:
            ang = 0.05 * (f - 2)
            scene_io.move_instance(sc, mover, translation=t0 + np.float32([0.03 * (f - 2), 0.01 * (f - 2), -0.02 * (f - 2)]),
                                   rotation=np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32), xform_of=xf)
            r.scene.update_instances(sc.instances, sc.instance_to_world)
        if light is not None and f >= 3:
            a = 0.05 * (f - 2)
            inst, xw, first, tris = scene_io.move_emissive_instance(sc, light, translation=l0 + np.float32([0.02 * (f - 2), -0.01 * (f - 2), 0.015 * (f - 2)]),
                                                                     rotation=np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], np.float32), xform_of=xf)
            r.scene.update_emissives(tris, first); r.scene.update_instances(inst, xw)
            if f % 4 == 0:
                r.invalidate_alias_table()       # the next PRELIGHTING render rebuilds it (on the first half's stream under overlap)
        cb = _frame(sc, w, h, f, **cam_of_frame(f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb)
    fin = r.final()
    hh = hashlib.sha1(fin.tobytes())
    for nm in ("B", "C", "D", "E", "F", "G", "target"):
        hh.update(r.p_indirect.download_plane(nm).tobytes())
    hh.update((r.p_indirect.download_plane("A") & 0xffffff).tobytes())
    return hh.hexdigest(), hashlib.sha1(fin.tobytes()).hexdigest(), r.p_indirect.read_counters()


def test_frame_overlap_back_to_back_frames(api, cornell_emissive):
    """Frames in flight for real: 24 frames of the 1920 x 1080 Cornell box with a moving camera, and 16 of a 3000-triangle materials scene whose largest
    instance moves every frame (zr_scene_update_instances between the frames: the refit and the previous-structure swap are ordered against BOTH streams),
    submitted without a host wait.  Carry mode: every plane of the last frame equals the plain order's; product mode: its FINAL and the ray counters do.
    The same with the Cornell box's light moving (zr_scene_update_emissives + _instances before every frame, the alias table invalidated every fourth).
    Two or three runs of each overlapped sequence, so that a race has more than one chance to show."""
    w, h = 1920, 1080
    cam = lambda f: dict(cam_pos=(0.01 * max(0, f - 3), 1.2, -4.043))
    plain = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 24, None, cam)
    for rep in range(3):
        carry = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 24, "carry", cam)
        assert carry == plain, (rep, carry, plain)
        prod = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 24, "product", cam)
        assert prod[1:] == plain[1:], (rep, prod, plain)
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))
    cam2 = lambda f: dict(cam_pos=(0.005 * max(0, f - 3), 0, -3.5))
    plain = _back_to_back(api, sc, w, h, wire.default_params(), 16, None, cam2, mover=idx)
    for rep in range(3):
        carry = _back_to_back(api, sc, w, h, wire.default_params(), 16, "carry", cam2, mover=idx)
        assert carry == plain, ("moving instance", rep, carry, plain)
        prod = _back_to_back(api, sc, w, h, wire.default_params(), 16, "product", cam2, mover=idx)
        assert prod[1:] == plain[1:], ("moving instance", rep, prod, plain)
    # bench.py's own path (a TiledRestirPT of one rank: stage_temporal + stage_spatial per frame) in the product mode
    import hashlib
    from zetaray_amd import tiling
    plain = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 24, None, cam)
    for rep in range(2):
        t = tiling.TiledRestirPT(cornell_emissive, w, h, 1, 0, params=wire.default_params())
        t.enable_frame_overlap(True)
        prev = None
        for f in range(1, 25):
            cb = _frame(cornell_emissive, w, h, f, **cam(f))
            if prev is not None:
                cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
            prev = cb.copy()
            t.render_frame(cb)
        assert (hashlib.sha1(t.r.final().tobytes()).hexdigest(), t.r.p_indirect.read_counters()) == plain[1:], ("tiled, one rank", rep)
        del t
    # ... and of four ranks' tile objects on this one device with every halo exchange in between (no host wait anywhere): the stitched FINAL of frame 12
    ref12 = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 12, None, cam)
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, 4, k, params=wire.default_params()) for k in range(4)]
    for t in ranks:
        t.enable_frame_overlap(True)
    prev = None
    for f in range(1, 13):
        cb = _frame(cornell_emissive, w, h, f, **cam(f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        tiling.render_frame_in_process(ranks, cb)
    img = np.zeros((h, w, 4), np.float32)
    for t in ranks:
        (x0, y0, tw, th), tile = t.final_tile()
        img[y0:y0 + th, x0:x0 + tw] = tile
    assert hashlib.sha1(img.tobytes()).hexdigest() == ref12[1], "four tiles, overlapped, back to back"
    del ranks
    # a moving light: emissive records + its instance updated before every frame, the alias table rebuilt every fourth
    lidx = [i for i in range(len(cornell_emissive.instances)) if cornell_emissive.instances["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    plain = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 16, None, cam, light=lidx)
    for rep in range(2):
        carry = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 16, "carry", cam, light=lidx)
        assert carry == plain, ("moving light", rep, carry, plain)
        prod = _back_to_back(api, cornell_emissive, w, h, wire.default_params(), 16, "product", cam, light=lidx)
        assert prod[1:] == plain[1:], ("moving light", rep, prod, plain)


def test_frame_overlap_needs_its_tracked_gbuffer(api, cornell_emissive):
    """Two frames in flight need the G-buffer's third plane set and stream tracking, which zr_pass_set_frame_overlap gives to the G-buffer it is handed: a frame rendered
    from ANOTHER G-buffer (a renderer that replaced it on a resize and forgot) fails loudly instead of racing; handing the new one over makes it work."""
    w, h = 96, 64
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    r.enable_frame_overlap(True)
    for f in (1, 2):
        r.render_frame(_frame(cornell_emissive, w, h, f))
    other = api.GBuffer(w, h)
    cb = _frame(cornell_emissive, w, h, 3)
    r.p_gbuffer.render(cb, r.scene, other)
    with pytest.raises(api.ZetaRayError):
        r.p_indirect.render_stage(cb, r.scene, other, api.STAGE_CANDIDATES)
    r.gbuffer = other
    r.enable_frame_overlap(True)
    for f in (3, 4, 5):
        r.render_frame(_frame(cornell_emissive, w, h, f))
    assert np.isfinite(r.final()).all() and r.final()[..., :3].max() > 0


def test_restir_pt_thread_sort_on_partial_tiles(api, cornell_emissive, oracle_emissive):
    """K12 at 150 x 90 (partial 32 x 32 tiles on the right and bottom boundaries: the transposed right-boundary groups, the one-to-one last
    group, in-image pixels of boundary groups in the k >= 5 bucket) with a camera that starts moving at frame 3: both thread maps, the
    sorted Reconnect_StC waves (boiling suppression over the pixels K12 put together) and everything downstream equal the oracle's, which
    equals the reference's own Sort + reconnect shaders (tests/test_ref_passes.py live pin); then the same with the sort flags off."""
    from oracle import zro
    w, h = 150, 90
    for flags_off in (0, wire.IND_SORT_TEMPORAL | wire.IND_SORT_SPATIAL):
        prm = wire.default_params()
        prm.flags &= ~flags_off
        r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
        o = zro.OracleRPT(oracle_emissive, w, h)
        prev = None
        for f in range(1, 5):
            cb = _frame(cornell_emissive, w, h, f, cam_pos=(0.07 * max(0, f - 2), 1.2, -4.043))
            if prev is not None:
                cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
            prev = cb.copy()
            r.render_frame(cb)
            want = o.render(cb, prm)
            assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f} (flags_off {flags_off})"
            for nm in RPT_PLANES:
                a, b = r.p_indirect.download_plane(nm), o.plane(nm)
                if nm == "A":
                    a, b = a & 0xffffff, b & 0xffffff
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {f}: plane {nm} (flags_off {flags_off})"
        m = r.p_indirect.download_plane("map_ntc").reshape(h, w)
        if not flags_off:
            assert int(((m & 0x7fff) != (31 | (31 << 7))).sum()) > 1000      # the spatial sort really permuted pixels


def test_restir_pt_materials_and_rr_on_gpu(api):
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 6, 8
    _rpt_compare(api, sc, o, 96, 64, prm, 3, cam=dict(cam_pos=(0, 0, -3.5)))


def test_restir_pt_full_resolution_properties(api, cornell_emissive):
    """1080p: finite, deterministic (two renderers, same frames -> identical bits), temporal reuse lowers variance,
    mean radiance stays within the reuse bias band of the no-reuse estimate."""
    w, h = 1920, 1080
    prm = wire.default_params()
    imgs = []
    for rep in range(2):
        r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
        for f in range(1, 4):
            r.render_frame(_frame(cornell_emissive, w, h, f))
        imgs.append(r.final())
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    assert np.isfinite(imgs[0]).all()
    p0 = wire.default_params()
    p0.flags &= ~(wire.IND_TEMPORAL_RESAMPLE | wire.IND_SPATIAL_RESAMPLE)
    r0 = api.Renderer(cornell_emissive, w, h, params=p0, integrator=api.INTEGRATOR_RESTIR_PT)
    r0.render_frame(_frame(cornell_emissive, w, h, 3))
    base = r0.final()
    lum = lambda a: a[..., :3] @ np.array([0.2126, 0.7152, 0.0722], np.float32)
    m_reuse, m_base = lum(imgs[0]).mean(), lum(base).mean()
    assert 0.7 < m_reuse / m_base < 1.5, (m_reuse, m_base)
    # reuse must reduce noise: compare the high-frequency energy of the two images
    def hf(a):
        l = lum(a)
        return np.abs(l[:, 1:] - l[:, :-1]).mean()
    # (three frames of reuse; with the K12-sorted waves of Reconnect_StC -- the reference's default -- the ratio is 0.80, unsorted 0.77)
    assert hf(imgs[0]) < 0.85 * hf(base)


@pytest.mark.parametrize("overlap", [False, True])
def test_restir_pt_tile_split_with_halo_exchange_on_gpu(api, cornell_emissive, oracle_emissive, overlap):
    """4 tiles (2x2) of one frame sequence on one device: each tile object renders its tile + apron, halos move through
    zr_pass_halo_pack / unpack (the buffers RCCL would carry); stitched radiance == the full-frame oracle, moving camera.  200 x 120:
    tile boundaries at 128 / 64, partial 32 x 32 sort tiles on the right and bottom image boundaries (K12's edge rules under tiling).
    overlap: every tile with frame overlap on (its G-buffer, PreLighting and K11 on a stream of its own, beside the previous frame's exchange and
    spatial stage: what bench.py runs on N devices)."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 200, 120, 4
    prm = wire.default_params()
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, world, r, params=prm) for r in range(world)]
    if overlap:
        for r in ranks:
            r.enable_frame_overlap(True)
    o = zro.OracleRPT(oracle_emissive, w, h)
    prev = None
    for f in range(1, 5):
        cb = _frame(cornell_emissive, w, h, f, cam_pos=(0.05 * f, 1.2, -4.043 + 0.02 * f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        for r in ranks:
            r.stage_temporal(cb)
        tiling.exchange_in_process(ranks, api.HALO_POST_TEMPORAL)
        for r in ranks:
            r.stage_spatial(cb)
        tiling.exchange_in_process(ranks, api.HALO_FINAL)
        want = o.render(cb, prm)
        img = np.zeros_like(want)
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"


def test_restir_pt_cost_balanced_tile_split_on_gpu(api, cornell_emissive, oracle_emissive):
    """The cost-balanced screen split: per-cell ray counters of a full-frame probe (zr_pass_read_cost_map) -> tiling.balanced_layout (kd-split,
    6 ranks: an uneven count, tiles of different sizes) -> the same halo protocol.  The cost map (wave lifetimes of K11 / K14 / K16 per cell) is denser where the
    box is, the six tiles' costs are balanced, and the stitched radiance is still bit-identical to the full-frame oracle with a moving camera."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 416, 288, 6
    prm = wire.default_params()
    probe = tiling.TiledRestirPT(cornell_emissive, w, h, 1, 0, params=prm)
    probe.r.p_indirect.enable_cost_map(True)
    for f in range(1, 5):
        probe.render_frame(_frame(cornell_emissive, w, h, f))
    cost = probe.owned_cost_cells()
    assert cost.shape == ((h + 31) // 32, (w + 31) // 32) and cost.min() > 0            # every cell's waves report their lifetime
    gw = cost.shape[1]
    assert cost[:, gw // 2 - 2:gw // 2 + 2].mean() > 2 * cost[:, :2].mean()              # the box in the middle costs more than the empty sides
    layout = tiling.balanced_layout(w, h, world, cost)
    assert sum(t[2] * t[3] for t in layout) == w * h and all(t[0] % 32 == 0 and t[1] % 32 == 0 for t in layout)
    shares = np.array([cost[t[1] // 32:(t[1] + t[3] + 31) // 32, t[0] // 32:(t[0] + t[2] + 31) // 32].sum() for t in layout]) / cost.sum()
    eq = np.array([cost[t[1] // 32:(t[1] + t[3] + 31) // 32, t[0] // 32:(t[0] + t[2] + 31) // 32].sum()
                   for t in [tiling.tile_rect(w, h, 4, r) for r in range(4)]]) / cost.sum()
    assert shares.max() * world < 1.6, shares            # no tile carries more than 1.6 x its fair share (cells are coarse at this size)
    del probe
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, world, r, params=prm, layout=layout) for r in range(world)]
    o = zro.OracleRPT(oracle_emissive, w, h)
    prev = None
    for f in range(1, 5):
        cb = _frame(cornell_emissive, w, h, f, cam_pos=(0.05 * f, 1.2, -4.043 + 0.02 * f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        for r in ranks:
            r.stage_temporal(cb)
        tiling.exchange_in_process(ranks, api.HALO_POST_TEMPORAL)
        for r in ranks:
            r.stage_spatial(cb)
        tiling.exchange_in_process(ranks, api.HALO_FINAL)
        want = o.render(cb, prm)
        img = np.zeros_like(want)
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"


def test_presampled_light_sets_on_gpu(api):
    """K3 on the GPU + the presampled NEE branches of K9 and K11 (PreLighting regenerates the sets every frame)."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 32, 128
    w, h = 96, 64
    rp = api.Renderer(sc, w, h, params=prm)
    rr = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    orpt = zro.OracleRPT(o, w, h)
    for f in (1, 2, 3):
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        o.presample(f, 32, 128)
        rp.render_frame(cb)
        _, planes = o.gbuffer(cb)
        want, _ = o.pathtrace(cb, planes, prm)
        assert np.array_equal(rp.final().view(np.uint32), want.view(np.uint32)), f"K9 frame {f}"
        rr.render_frame(cb)
        want2 = orpt.render(cb, prm)
        assert np.array_equal(rr.final().view(np.uint32), want2.view(np.uint32)), f"ReSTIR PT frame {f}"
    # a pass asked to use presampled sets that were never generated must fail loudly
    r3 = api.Renderer(sc, w, h, params=wire.default_params())
    r3.render_frame(cb)
    r3.p_indirect.set_params(prm)
    with pytest.raises(api.ZetaRayError):
        r3.p_indirect.render(cb, r3.scene, r3.gbuffer)


@pytest.mark.parametrize("w,h", [(72, 48), (200, 120)])
def test_restir_di_bit_exact(api, cornell_emissive, oracle_emissive, w, h):
    """K5 + K6 (ReSTIR DI, emissive lights) through the C-ABI, 5 frames with a camera that starts moving at frame 3
    (disocclusion -> 4-sample spatial branch): radiance, reservoir planes and ray counters bit-exact vs the oracle."""
    from oracle import zro
    prm = wire.default_params_di()
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params())
    di = r.enable_direct(prm)
    o = zro.OracleRDI(oracle_emissive, w, h)
    prev = None
    for f in range(1, 6):
        cb = _frame(cornell_emissive, w, h, f, cam_pos=(0.06 * max(0, f - 2), 1.2, -4.043))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        di.read_counters(reset=True)
        r.render_frame(cb)
        got = di.download()
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert di.read_counters() == o.counters
        for nm, onm in (("di_A", "A"), ("di_B", "B"), ("di_target", "target")):
            assert np.array_equal(di.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: DI plane {onm}"
    assert got[..., :3].max() > 0


def test_restir_di_materials_presampled_on_gpu(api):
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    w, h = 96, 64
    prm = wire.default_params_di()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 32, 128
    ip = wire.default_params()
    ip.presampling, ip.num_sample_sets, ip.sample_set_size = 1, 32, 128
    r = api.Renderer(sc, w, h, params=ip)
    di = r.enable_direct(prm)
    odi = zro.OracleRDI(o, w, h)
    for f in (1, 2, 3):
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        o.presample(f, 32, 128)
        r.render_frame(cb)
        want = odi.render(cb, prm)
        assert np.array_equal(di.download().view(np.uint32), want.view(np.uint32)), f"frame {f}"


@pytest.mark.parametrize("alpha_min", [0.25, 0.8])
def test_restir_di_half_vector_copy_shift_on_gpu(api, alpha_min):
    """USE_HALF_VECTOR_COPY_SHIFT (ReSTIR_DI/Params.hlsli:12) through the C-ABI: glossy lobes below alpha_min keep their half vector across temporal
    and spatial reuse (Resampling.hlsli:130-317, PairwiseMIS.hlsli:37-214); radiance, reservoir planes (incl. the oct-encoded half vector and lobe bits)
    and ray counters against the oracle, which is pinned on the reference's shaders compiled with the switch on (tests/test_ref_passes.py di_half_vector*)."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    w, h = 160, 96
    prm = wire.default_params_di()
    prm.flags |= wire.DI_HALF_VECTOR_COPY_SHIFT
    prm.alpha_min = alpha_min
    r = api.Renderer(sc, w, h, params=wire.default_params())
    di = r.enable_direct(prm)
    odi = zro.OracleRDI(o, w, h)
    prev, shifted = None, 0
    for f in range(1, 6):
        cb = _frame(sc, w, h, f, cam_pos=(0.04 * max(0, f - 2), 0.01 * max(0, f - 2), -3.5))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        di.read_counters(reset=True)
        r.render_frame(cb)
        want = odi.render(cb, prm)
        assert np.array_equal(di.download().view(np.uint32), want.view(np.uint32)), f"frame {f}"
        assert di.read_counters() == odi.counters
        for nm, onm in (("di_A", "A"), ("di_B", "B"), ("di_target", "target")):
            assert np.array_equal(di.download_plane(nm).view(np.uint8), odi.plane(onm).view(np.uint8)), f"frame {f}: DI plane {onm}"
        shifted += int(((odi.plane("A").view(np.uint32).reshape(-1, 4)[:, 2] >> 21) & 1).sum())      # metadata (A.z >> 16) bit 5 = the reservoir carries a half vector
    assert shifted > 0, "no reservoir used the half-vector shift: the case does not exercise it"


def test_restir_gi_bit_exact(api, cornell_emissive, oracle_emissive):
    """K10 (ReSTIR GI) through the C-ABI, 5 frames, camera moving from frame 3: radiance, reservoir planes, ray counters."""
    from oracle import zro
    w, h = 200, 120
    prm = wire.default_params()
    prm.flags |= wire.IND_STOCHASTIC_MULTI_BOUNCE
    r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    o = zro.OracleRGI(oracle_emissive, w, h)
    prev = None
    for f in range(1, 6):
        cb = _frame(cornell_emissive, w, h, f, cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert r.p_indirect.read_counters() == o.counters
        for nm, onm in (("gi_A", "A"), ("gi_B", "B"), ("gi_C", "C")):
            assert np.array_equal(r.p_indirect.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: GI plane {onm}"


def test_restir_gi_materials_rr_on_gpu(api):
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    w, h = 96, 64
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 5, 7
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    ogi = zro.OracleRGI(o, w, h)
    for f in (1, 2, 3):
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        r.render_frame(cb)
        want = ogi.render(cb, prm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}"


def test_compositing(api, cornell_emissive, oracle_emissive):
    """Compositing.hlsl: (emissive DI + indirect * !emissive) / NumFramesCameraStatic; miss pixels 0 when not accumulating."""
    from oracle import zro
    w, h = 96, 64
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(wire.default_params_di())
    comp = r.enable_compositing()
    for f in range(1, 4):
        cb = _frame(cornell_emissive, w, h, f, accumulate=1, camera_static=1, num_frames_static=f)
        r.render_frame(cb)
    ind, d, out = r.final(), di.download(), comp.download()
    gb, _ = r.gbuffer.download()
    mr = gb[wire.GB_PLANE_NAMES.index("metallic_roughness")]
    emissive = ((mr & 0xff) & 2) != 0
    want = (d[..., :3] + ind[..., :3] * (~emissive)[..., None]) / np.float32(3)
    assert np.array_equal(out[..., :3].view(np.uint32), want.astype(np.float32).view(np.uint32))
    assert out[..., :3].max() > 0


# ------------------------------------------------------------------ sun / sky (K17 + NEE_EMISSIVE == 0 path tracer)
@pytest.fixture(scope="module")
def cornell_sky():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell.npz"))


def test_sky_view_lut_bit_exact(api, cornell_sky):
    """K17 through ZR_PASS_SKY vs the oracle: every R11G11B10F texel of the 256 x 128 LUT, two sun positions."""
    from oracle import zro
    orc = zro.OracleScene(cornell_sky)
    sc = api.Scene(cornell_sky)
    p = api.Pass(api.PASS_SKY, 256, 128)
    for sun in ((0.6565358, -0.0560669, 0.752208233), (0.2, -0.9, -0.3)):
        cb = scene_io.make_frame_constants(64, 64)
        sd = np.array(sun, np.float32)
        cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        p.render(cb, sc, None)
        got = p.download_plane("sky_lut")[..., 0]
        want = orc.sky_lut(cb, 256, 128)
        assert np.array_equal(got, want), f"{int((got != want).sum())} LUT texels differ"


@pytest.mark.parametrize("w,h,frame", [(64, 48, 1), (200, 120, 4)])
def test_path_tracer_sun_sky_bit_exact(api, cornell_sky, w, h, frame):
    """The reference's default Cornell box (no emissives): K17 + K1 + K9 with sun / sky NEE, radiance and counters bit-exact."""
    from oracle import zro
    orc = zro.OracleScene(cornell_sky)
    cb = scene_io.make_frame_constants(w, h, frame_num=frame, num_emissives=0)
    prm = wire.default_params()
    r = api.Renderer(cornell_sky, w, h, params=prm)
    r.render_frame(cb)
    got = r.final()
    n_closest, n_shadow = r.p_indirect.read_counters()
    orc.sky_lut(cb, 256, 128)
    _, planes = orc.gbuffer(cb)
    want, cnt = orc.pathtrace(cb, planes, prm)
    mism = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert mism == 0, f"{mism} radiance floats differ, max abs {np.abs(got - want).max()}"
    assert (n_closest, n_shadow) == (cnt[0], cnt[1])
    assert got[..., :3].sum() > 0


@pytest.mark.parametrize("scene_kind,w,h", [("cornell", 72, 48), ("cornell", 200, 120), ("glossy", 64, 48)])
def test_sky_di_bit_exact(api, cornell_sky, scene_kind, w, h):
    """K7 + K8 (sun + sky ReSTIR DI) through the C-ABI, 5 frames, camera moving from frame 3: radiance, the four reservoir planes
    and ray counters bit-exact vs the oracle.  "glossy": open-top clutter of metal / coat / glass (half-vector copy shift)."""
    from oracle import zro
    if scene_kind == "cornell":
        sc, cam0, sun = cornell_sky, (0.0, 1.2, -4.043), None
        osc = zro.OracleScene(sc)
    else:
        sc, cam0, sun = scene_io.make_synthetic_scene(num_tris=1500, num_emissive=0, seed=5, open_top=True), (0.0, 2.0, -3.5), (0.3, -0.8, 0.4)
        osc = zro.OracleScene(sc, force_bvh=True)
    prm = wire.default_params_sky_di()
    r = api.Renderer(sc, w, h, params=wire.default_params())
    r.skip_indirect = True
    di = r.enable_sky_direct(prm)
    o = zro.OracleSDI(osc, w, h)
    prev = None
    for f in range(1, 6):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(cam0[0] + 0.06 * max(0, f - 2), cam0[1], cam0[2]))
        if sun is not None:
            sd = np.array(sun, np.float32)
            cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        di.read_counters(reset=True)
        r.render_frame(cb)
        got = di.download()
        osc.sky_lut(cb, 256, 128)
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert di.read_counters() == o.counters
        for nm, onm in (("sdi_A", "A"), ("sdi_B", "B"), ("sdi_C", "C"), ("sdi_target", "target")):
            assert np.array_equal(di.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: sky DI plane {onm}"
    assert got[..., :3].max() > 0


def test_restir_gi_sun_sky_bit_exact(api, cornell_sky):
    """K10 with sun + sky NEE (no emissive triangles) through the C-ABI, 4 frames, camera moving from frame 3."""
    from oracle import zro
    w, h = 120, 80
    osc = zro.OracleScene(cornell_sky)
    prm = wire.default_params()
    r = api.Renderer(cornell_sky, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    o = zro.OracleRGI(osc, w, h)
    prev = None
    for f in range(1, 5):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        osc.sky_lut(cb, 256, 128)
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert r.p_indirect.read_counters() == o.counters
    assert got[..., :3].max() > 0


@pytest.mark.parametrize("kind", ["di", "sky_di", "restir_gi"])
def test_tile_split_with_halo_exchange_other_passes_on_gpu(api, cornell_emissive, oracle_emissive, cornell_sky, kind):
    """SURVEY 8(e) for the other passes with cross-pixel reuse: 4 tiles (2x2) on one device, reservoir halos through
    zr_pass_halo_pack / unpack (ReSTIR DI 24 B/px and sun + sky DI 13 B/px between their stages, ReSTIR GI 40 B/px after the frame);
    stitched radiance == the full-frame oracle, moving camera."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 192, 128, 4
    if kind == "di":
        sc, osc, pp = cornell_emissive, oracle_emissive, wire.default_params_di()
        o = zro.OracleRDI(osc, w, h)
    elif kind == "sky_di":
        sc, pp = cornell_sky, wire.default_params_sky_di()
        osc = zro.OracleScene(sc)
        o = zro.OracleSDI(osc, w, h)
    else:
        sc, osc, pp = cornell_emissive, oracle_emissive, wire.default_params()
        o = zro.OracleRGI(osc, w, h)
    ranks = [tiling.TiledRestirPT(sc, w, h, world, r, params=wire.default_params(), kind=kind, pass_params=pp) for r in range(world)]
    assert ranks[0].bpp == {"di": 24, "sky_di": 13, "restir_gi": 40}[kind]
    post, final = ranks[0].EXCHANGES[kind]
    prev = None
    for f in range(1, 5):
        cb = _frame(sc, w, h, f, cam_pos=(0.05 * f, 1.2, -4.043 + 0.02 * f))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        for r in ranks:
            r.stage_temporal(cb)
        if post:
            tiling.exchange_in_process(ranks, api.HALO_POST_TEMPORAL)
        for r in ranks:
            r.stage_spatial(cb)
        if final:
            tiling.exchange_in_process(ranks, api.HALO_FINAL)
        if kind == "sky_di":
            osc.sky_lut(cb, 256, 128)
        want = o.render(cb, pp)
        img = np.zeros_like(want)
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"{kind} frame {f}: {mism} pixels differ"


def test_light_voxel_grid_and_restir_gi_lvg_on_gpu(api):
    """K4 through PRELIGHTING (use_lvg) + the ReSTIR_GI_LVG variant through the C-ABI: grid samples and 3 GI frames bit-exact."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=2000, num_emissive=600, seed=3)
    osc = zro.OracleScene(sc, force_bvh=True)
    w, h = 96, 64
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 16, 64
    prm.use_lvg = 1
    dim, ext, off = (8, 4, 10), (0.6, 0.45, 0.6), 0.1
    prm.lvg_grid_dim = dim[0] | (dim[1] << 10) | (dim[2] << 20)
    prm.lvg_extents[:] = ext
    prm.lvg_offset_y = off
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    o = zro.OracleRGI(osc, w, h)
    for f in range(1, 4):
        cb = _frame(sc, w, h, f, cam_pos=(0.0, 0.0, -3.5))
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        osc.presample(f, 16, 64)
        want_grid = osc.build_lvg(cb, dim, ext, off)
        assert np.array_equal(r.scene.get_light_voxel_grid(dim).view(np.uint8), want_grid.view(np.uint8)), f"frame {f}: light voxel grid differs"
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert r.p_indirect.read_counters() == o.counters


@pytest.mark.parametrize("kind,w,h", [("cornell", 64, 48), ("cornell", 200, 120), ("glossy", 64, 48)])
def test_restir_pt_sun_sky_bit_exact(api, cornell_sky, kind, w, h):
    """K11-K16 with sun + sky lighting (the NEE_EMISSIVE == 0 shader variants: the reference's default Cornell box) through the
    C-ABI, 5 frames, camera moving from frame 4: radiance, the 7 reservoir planes and ray counters bit-exact vs the oracle."""
    from oracle import zro
    if kind == "cornell":
        sc, cam0, sun = cornell_sky, (0.0, 1.2, -4.043), None
        osc = zro.OracleScene(sc)
    else:
        sc, cam0, sun = scene_io.make_synthetic_scene(num_tris=1500, num_emissive=0, seed=5, open_top=True), (0.0, 2.0, -3.5), (0.3, -0.8, 0.4)
        osc = zro.OracleScene(sc, force_bvh=True)
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    o = zro.OracleRPT(osc, w, h)
    prev = None
    for f in range(1, 6):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(cam0[0] + 0.05 * max(0, f - 3), cam0[1], cam0[2]))
        if sun is not None:
            sd = np.array(sun, np.float32)
            cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        osc.sky_lut(cb, 256, 128)
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert r.p_indirect.read_counters() == o.counters
        for nm in "ABCDEFG":
            assert np.array_equal(r.p_indirect.download_plane(nm).view(np.uint8), o.plane(nm).view(np.uint8)), f"frame {f}: plane {nm}"
    assert got[..., :3].max() > 0


def test_firefly_filter_on_gpu(api, cornell_emissive):
    """Compositing + FireflyFilter (3 x 3 luminance clamp, LDS-tiled) through the C-ABI vs the oracle on the same composited image; the
    filter only ever replaces a pixel by one of its neighbours' colours; prints the achieved bandwidth at 3840 x 2160."""
    from oracle import zro
    w, h = 200, 120
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    r.enable_direct(wire.default_params_di())
    plain = r.enable_compositing()
    cb = _frame(cornell_emissive, w, h, 1)
    r.render_frame(cb)
    unfiltered = plain.download()
    gb, _ = r.gbuffer.download()
    depth = np.asarray(gb[wire.GB_PLANE_NAMES.index("depth")]).reshape(h, w)
    filt = r.enable_compositing(firefly_filter=True)
    filt.render(cb, r.scene, r.gbuffer)
    got = filt.download()
    want = zro.firefly_filter(unfiltered, depth)
    assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))
    changed = (got[..., :3] != unfiltered[..., :3]).any(axis=2)
    assert changed.any() and changed.mean() < 0.5
    # bandwidth of the stencil at 4K (20 B read + 16 B written per pixel)
    W4, H4 = 3840, 2160
    r4 = api.Renderer(cornell_emissive, W4, H4, params=wire.default_params(), integrator=api.INTEGRATOR_PATH_TRACING)
    c4 = r4.enable_compositing(firefly_filter=True)
    cb4 = _frame(cornell_emissive, W4, H4, 1)
    r4.render_frame(cb4)
    c4.enable_timing(True)
    best = 1e9
    for _ in range(5):
        c4.render(cb4, r4.scene, r4.gbuffer)
        import torch
        torch.cuda.synchronize()
        best = min(best, c4.timings()["firefly_filter"][0])
    gbs = 36.0 * W4 * H4 / (best * 1e-3) / 1e9
    print(f"firefly filter 3840x2160: {best * 1e3:.1f} us, {gbs:.0f} GB/s algorithmic ({gbs / 8000:.1%} of the 8 TB/s roofline)")
    assert gbs > 500


# ------------------------------------------------------------------------------------------------ material textures
def _textured_scene(num_emissive):
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=num_emissive, seed=11, open_top=(num_emissive == 0))
    offs = scene_io.add_test_textures(sc)
    return sc, offs


def _textured_frames(sc, offs, w, h, n, cam0, sun=None):
    prev = None
    for f in range(1, n + 1):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives),
                                           cam_pos=(cam0[0] + 0.05 * max(0, f - 2), cam0[1], cam0[2]))
        scene_io.set_texture_heap_offsets(cb, offs)
        if sun is not None:
            sd = np.array(sun, np.float32)
            cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        yield f, cb


@pytest.mark.parametrize("dof", [False, True])
def test_textured_gbuffer_and_emissive_power_on_gpu(api, dof):
    """K1 with base-colour / normal / metallic-roughness / emissive maps, UV differentials and alpha-tested primary rays, and K2 +
    alias table over emissive-textured triangles, through the C-ABI: bit-exact vs the oracle."""
    from oracle import zro
    sc, offs = _textured_scene(1500)
    w, h = 200, 120
    _, cb = next(_textured_frames(sc, offs, w, h, 1, (0.3, 0.2, -3.6)))
    if dof:
        cb["dof"], cb["focus_depth"], cb["lens_radius"], cb["camera_ray_uv_grads_scale"] = 1, 3.0, 0.05, 0.75
    orc = zro.OracleScene(sc, force_bvh=True, cb=cb)
    r = api.Renderer(sc, w, h)
    r.p_prelight.render(cb, r.scene)
    assert np.array_equal(r.scene.get_alias_table().view(np.uint8), orc.alias.view(np.uint8))
    r.p_gbuffer.render(cb, r.scene, r.gbuffer)
    got, _ = r.gbuffer.download()
    want, _ = orc.gbuffer(cb)
    for name, a, b in zip(wire.GB_PLANE_NAMES, got, want):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"G-buffer plane {name} differs"
    # table offsets that point outside the heap are refused, not read
    bad = cb.copy()
    bad["emissive_maps_desc_heap_offset"] = 1000
    with pytest.raises(api.ZetaRayError):
        r.p_gbuffer.render(bad, r.scene, r.gbuffer)


@pytest.mark.parametrize("integrator", ["pt", "restir_gi", "restir_pt"])
@pytest.mark.parametrize("lights", ["emissive", "sun_sky"])
def test_textured_integrators_on_gpu(api, integrator, lights):
    """The TEXTURED kernel permutations (ray differentials through every bounce, replay and reconnection shift; texture LODs from
    them) x both NEE_EMISSIVE permutations, through the C-ABI over 4 frames with a moving camera: radiance, persistent reservoir
    planes and ray counters bit-exact vs the oracle."""
    from oracle import zro
    sc, offs = _textured_scene(1500 if lights == "emissive" else 0)
    w, h = 96, 64
    cam0, sun = ((0.3, 0.2, -3.6), None) if lights == "emissive" else ((0.0, 2.0, -3.5), (0.3, -0.8, 0.4))
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 5, 7
    kind = {"pt": api.INTEGRATOR_PATH_TRACING, "restir_gi": api.INTEGRATOR_RESTIR_GI, "restir_pt": api.INTEGRATOR_RESTIR_PT}[integrator]
    r = api.Renderer(sc, w, h, params=prm, integrator=kind)
    orc = None
    o = None
    for f, cb in _textured_frames(sc, offs, w, h, 4, cam0, sun):
        if orc is None:
            orc = zro.OracleScene(sc, force_bvh=True, cb=cb)
            o = {"pt": None, "restir_gi": zro.OracleRGI, "restir_pt": zro.OracleRPT}[integrator]
            o = o(orc, w, h) if o else None
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        if lights == "sun_sky":
            orc.sky_lut(cb, 256, 128)
        if o is None:
            keep, planes = orc.gbuffer(cb)
            want, cnt = orc.pathtrace(cb, planes, prm)
        else:
            want, cnt = o.render(cb, prm), o.counters
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        assert tuple(r.p_indirect.read_counters()) == tuple(cnt)
        if integrator == "restir_pt":
            for nm in "ABCDEFG":
                pa, pb = r.p_indirect.download_plane(nm), o.plane(nm)
                if nm == "A":
                    pa, pb = pa & 0xffffff, pb & 0xffffff
                assert np.array_equal(pa.view(np.uint8), pb.view(np.uint8)), f"frame {f}: plane {nm}"
    assert got[..., :3].max() > 0


def test_taa_on_gpu(api, cornell_emissive, oracle_emissive):
    """ZR_PASS_TAA after Compositing (ReSTIR PT, moving + jittered camera, 5 frames): the RGBA16F output == the oracle's TAA.hlsl
    restatement run on the same composited signal and G-buffer planes, every frame; reset_temporal restarts the history."""
    from oracle import zro
    w, h = 200, 120
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    taa = r.enable_taa(0.1)
    hist = np.zeros((h, w, 4), np.uint16)
    prev = None
    for f in range(1, 6):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives), cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043),
                                           jitter=(0.25 * ((f * 7) % 4 - 1.5) / 2, 0.25 * ((f * 3) % 4 - 1.5) / 2))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        if f == 4:
            taa.reset_temporal()
        r.render_frame(cb)
        signal = r.p_composit.download()
        planes, _ = r.gbuffer.download()
        want = zro.taa(signal, planes[7].reshape(h, w), planes[3].reshape(h, w), hist, 0.1, f not in (1, 4))
        got = taa.download_plane("taa")
        assert np.array_equal(got[..., :3], want[..., :3]), f"frame {f}"
        if f in (1, 4):
            assert np.array_equal(got[..., :3], signal[..., :3].astype(np.float16).view(np.uint16))
        hist = got
    assert got[..., :3].view(np.float16).astype(np.float32).max() > 0


def test_taa_with_overflowing_and_nan_history_on_gpu(api, cornell_emissive):
    """The TAA kernel converts its 36 history texels per pixel with plain v_cvt_f32_f16 and re-runs the exact path only when one of them holds an Inf or
    a NaN (zr_taa.h LoadHistoryTexel<EXACT>).  Here the history really holds them: a synthetic signal with values beyond fp16's range (-> Inf in the RGBA16F
    output, i.e. in the next frame's history), NaNs and Infs, over the Cornell box's G-buffer with a moving camera: every frame == the oracle's TAA.hlsl
    restatement, bit for bit, and the special values are really there."""
    import torch
    from oracle import zro
    w, h = 160, 96
    prm = wire.default_params()
    prm.taa_blend_weight = 0.1
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params())
    taa = api.Pass(api.PASS_TAA, w, h, params=prm)
    rng = np.random.default_rng(17)
    hist = np.zeros((h, w, 4), np.uint16)
    prev, saw_special = None, 0
    for f in range(1, 6):
        cb = _chain(_frame(cornell_emissive, w, h, f, cam_pos=(0.04 * f, 1.2, -4.043)), prev)
        prev = cb.copy()
        r.p_gbuffer.render(cb, r.scene, r.gbuffer)
        sig = np.zeros((h, w, 4), np.float32)
        sig[..., :3] = rng.uniform(0.0, 2.0, (h, w, 3)).astype(np.float32)
        hot = rng.random((h, w)) < 0.02
        sig[hot, 0] = 3.0e5                      # beyond fp16: the stored output is +Inf
        sig[rng.random((h, w)) < 0.005, 1] = np.nan
        sig[rng.random((h, w)) < 0.005, 2] = np.inf
        dev = torch.from_numpy(sig).cuda()
        taa.set_input(api.IN_TAA_SIGNAL, dev.data_ptr())
        taa.render(cb, r.scene, r.gbuffer)
        planes, _ = r.gbuffer.download()
        want = zro.taa(sig, planes[7].reshape(h, w), planes[3].reshape(h, w), hist, 0.1, f > 1)
        got = taa.download_plane("taa")
        assert np.array_equal(got[..., :3], want[..., :3]), f"frame {f}: {int((got[..., :3] != want[..., :3]).any(-1).sum())} pixels differ"
        hist = got
        saw_special += int(((got[..., :3] & 0x7c00) == 0x7c00).sum())
    assert saw_special > 50


def test_half_conversion_instructions_match_portable_code(api):
    """The kernels' fp32 <-> fp16 conversions use v_cvt_f16_f32 / v_cvt_f32_f16; on the device they must agree with the portable
    code (== the oracle's, pinned to the reference's half in test_ref_pins.py) for all 2^32 / 2^16 bit patterns."""
    import ctypes as C
    a, b = C.c_uint64(), C.c_uint64()
    L = api.lib()
    L.zr_selftest_half_conversions.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    assert L.zr_selftest_half_conversions(0, C.byref(a), C.byref(b)) == 0
    assert (a.value, b.value) == (0, 0)


def test_restir_pt_uncached_kernel_build_on_gpu(api):
    """K11 has two builds: 4 waves per SIMD + the top of the tree in LDS (the default for every scene with a tree since round 6: every other ReSTIR PT test runs it) and
    the one without the node cache (3 waves for the general material class); zr_debug_set_large_scene_nodes(0x7FFFFFFF) selects the latter: same bit-exact
    comparison as the materials / RR test."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 6, 8
    assert api.lib().zr_debug_set_large_scene_nodes(0x7FFFFFFF) == 0
    try:
        _rpt_compare(api, sc, o, 96, 64, prm, 3, cam=dict(cam_pos=(0, 0, -3.5)))
    finally:
        api.lib().zr_debug_set_large_scene_nodes(0)


def test_material_class_kernels_change_nothing(api, cornell_emissive, oracle_emissive, cornell_sky):
    """K11 / K14 / K16 (and K9, K5 - K8, K10) have a second permutation for scenes of the plain material class (no metal, transmission, thin wall, coat or texture in
    the whole material table: the Cornell boxes), compiled without the code of those lobes.  The other ReSTIR PT tests of this file run it on
    the Cornell box by default; here the same box renders with the general kernels (zr_debug_set_material_class_kernels(0)) in both instantiations of K11 (with and
    without the LDS node cache), with the PLAIN node-cached one, and with sun + sky lighting through the general kernels: all bit-exact against the oracle, hence
    identical to each other.  A scene with one metallic material is of the general class."""
    from oracle import zro
    L = api.lib()
    assert api.Scene(cornell_emissive).material_class() == 1 and api.Scene(cornell_sky).material_class() == 1
    assert api.Scene(scene_io.make_synthetic_scene(num_tris=300, num_emissive=100, seed=3)).material_class() == 0
    try:
        for enable, large in ((0, 0), (1, 1), (0, 1)):
            assert L.zr_debug_set_material_class_kernels(enable) == 0 and L.zr_debug_set_large_scene_nodes(1 if large else 0x7FFFFFFF) == 0
            _rpt_compare(api, cornell_emissive, oracle_emissive, 200, 120, wire.default_params(), 4)
        assert L.zr_debug_set_material_class_kernels(0) == 0 and L.zr_debug_set_large_scene_nodes(0) == 0
        w, h = 96, 64
        osc = zro.OracleScene(cornell_sky)
        r = api.Renderer(cornell_sky, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
        o = zro.OracleRPT(osc, w, h)
        for f in range(1, 4):
            cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.0, 1.2, -4.043))
            r.render_frame(cb)
            osc.sky_lut(cb, 256, 128)
            want = o.render(cb, wire.default_params())
            assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"sun + sky, general kernels, frame {f}"
        # K9, K5 / K6, K7 / K8 and K10 have the permutation too: their Cornell tests above ran it; here the general kernels on the same scenes
        test_path_tracer_bit_exact(api, cornell_emissive, oracle_emissive, 160, 96, 7)
        test_path_tracer_sun_sky_bit_exact(api, cornell_sky, 200, 120, 4)
        test_restir_di_bit_exact(api, cornell_emissive, oracle_emissive, 72, 48)
        test_sky_di_bit_exact(api, cornell_sky, "cornell", 72, 48)
        test_restir_gi_bit_exact(api, cornell_emissive, oracle_emissive)
        test_restir_gi_sun_sky_bit_exact(api, cornell_sky)
    finally:
        L.zr_debug_set_material_class_kernels(1); L.zr_debug_set_large_scene_nodes(0)


# ------------------------------------------------------------------ parity on the BASELINE configurations
def test_baseline_config_cornell_1080p_restir_pt_bit_exact(api, cornell_emissive, oracle_emissive):
    """The bench workload itself: Cornell (emissive) 1920 x 1080, ReSTIR PT defaults (3 / 4 bounces, temporal + spatial reuse, boiling
    suppression), frames 1-3, FULL frame vs the oracle, tolerance 0: radiance, the 7 persistent reservoir planes + spatial neighbour
    plane, and the ray counters of every frame (IndirectLighting.cpp:877-1004)."""
    got = _rpt_compare(api, cornell_emissive, oracle_emissive, 1920, 1080, wire.default_params(), 3)
    assert got[..., :3].max() > 0


def _chain(cb, prev):
    if prev is not None:
        cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
    return cb


@pytest.mark.parametrize("half_vector", [False, True])
def test_baseline_config2_cornell_1080p_restir_di_emissive_bit_exact(api, cornell_emissive, oracle_emissive, half_vector):
    """BASELINE config 2 (emissive half): Cornell (emissive) 1920 x 1080, ReSTIR DI K5 + K6, frames 1-3 with the camera moving on frame 3,
    FULL frame vs the oracle, tolerance 0: radiance, both reservoir planes + target, ray counters (DirectLighting.cpp:100-190).
    half_vector: the same with USE_HALF_VECTOR_COPY_SHIFT on (alpha_min 1: the boxes' rough gloss lobes qualify whenever a BSDF-sampled candidate wins)."""
    from oracle import zro
    w, h = 1920, 1080
    prm = wire.default_params_di()
    if half_vector:
        prm.flags |= wire.DI_HALF_VECTOR_COPY_SHIFT
        prm.alpha_min = 1.0
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params())
    r.skip_indirect = True
    di = r.enable_direct(prm)
    o = zro.OracleRDI(oracle_emissive, w, h)
    prev = None
    for f in range(1, 4):
        cb = _chain(_frame(cornell_emissive, w, h, f, cam_pos=(0.06 * max(0, f - 2), 1.2, -4.043)), prev)
        prev = cb.copy()
        di.read_counters(reset=True)
        r.render_frame(cb)
        got = di.download()
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} of {w * h} pixels differ"
        assert di.read_counters() == o.counters
        for nm, onm in (("di_A", "A"), ("di_B", "B"), ("di_target", "target")):
            assert np.array_equal(di.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: DI plane {onm}"
    assert got[..., :3].max() > 0
    if half_vector:
        assert int(((o.plane("A").view(np.uint32).reshape(-1, 4)[:, 2] >> 21) & 1).sum()) > 0, "no reservoir carries a half vector: the case does not exercise the shift"


def test_baseline_config2_cornell_1080p_sky_di_bit_exact(api, cornell_sky):
    """BASELINE config 2 (sun + sky half, the reference's default Cornell box): 1920 x 1080 SkyDI K7 + K8, frames 1-3, camera moving on
    frame 3, FULL frame vs the oracle, tolerance 0: radiance, the four reservoir planes, ray counters (SkyDI.cpp:72-180)."""
    from oracle import zro
    w, h = 1920, 1080
    osc = zro.OracleScene(cornell_sky)
    prm = wire.default_params_sky_di()
    r = api.Renderer(cornell_sky, w, h, params=wire.default_params())
    r.skip_indirect = True
    di = r.enable_sky_direct(prm)
    o = zro.OracleSDI(osc, w, h)
    prev = None
    for f in range(1, 4):
        cb = _chain(scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.06 * max(0, f - 2), 1.2, -4.043)), prev)
        prev = cb.copy()
        di.read_counters(reset=True)
        r.render_frame(cb)
        got = di.download()
        osc.sky_lut(cb, 256, 128)
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} of {w * h} pixels differ"
        assert di.read_counters() == o.counters
        for nm, onm in (("sdi_A", "A"), ("sdi_B", "B"), ("sdi_C", "C"), ("sdi_target", "target")):
            assert np.array_equal(di.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: sky DI plane {onm}"
    assert got[..., :3].max() > 0


def test_baseline_config3_cornell_1080p_restir_gi_bit_exact(api, cornell_emissive, oracle_emissive):
    """BASELINE config 3: Cornell (emissive) 1920 x 1080, ReSTIR GI with 3 non-transmissive bounces, temporal resampling, M_max 10
    (IndirectLighting.cpp:277-368), frames 1-3, camera moving on frame 3, FULL frame vs the oracle, tolerance 0: radiance, the three
    reservoir planes, ray counters."""
    from oracle import zro
    w, h = 1920, 1080
    prm = wire.default_params()
    assert prm.max_non_tr_bounces == 3 and prm.m_max_temporal == 10
    r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    o = zro.OracleRGI(oracle_emissive, w, h)
    prev = None
    for f in range(1, 4):
        cb = _chain(_frame(cornell_emissive, w, h, f, cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043)), prev)
        prev = cb.copy()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        want = o.render(cb, prm)
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} of {w * h} pixels differ"
        assert r.p_indirect.read_counters() == o.counters
        for nm, onm in (("gi_A", "A"), ("gi_B", "B"), ("gi_C", "C")):
            assert np.array_equal(r.p_indirect.download_plane(nm).view(np.uint8), o.plane(onm).view(np.uint8)), f"frame {f}: GI plane {onm}"
    assert got[..., :3].max() > 0


@pytest.fixture(scope="module")
def atrium():
    """BASELINE config 4's scene class (bench.py --scene synthetic): 380 588 triangles of which 100 000 emissive"""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
    return sc, zro.OracleScene(sc, force_bvh=True)


def test_baseline_config_atrium_restir_pt_bit_exact(api, atrium):
    """The 380k-triangle / 100k-light atrium with the presampled light sets the reference would use (128 x 512, PreLighting.cpp:289-297)
    at 480 x 270 for 3 frames, the large-scene threshold at its default: the BVH has >= 16k nodes, so the 4-wave build of K11
    (k_rpt_pathtrace_w4) and traversal stacks deeper than the 8 LDS entries really run.  Radiance, reservoir planes, counters: tolerance 0."""
    sc, o = atrium
    handle = api.Scene(sc)
    nodes, tris, depth = handle.bvh_info()
    handle.close()
    assert nodes >= 16384 and tris == sc.num_tris, (nodes, tris)
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
    w, h = 480, 270
    from oracle import zro
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    orpt = zro.OracleRPT(o, w, h)
    for f in (1, 2, 3):
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        o.presample(f, 128, 512)
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        got = r.final()
        want = orpt.render(cb, prm)
        assert r.p_indirect.read_counters() == orpt.counters, f"frame {f}: ray counters differ"
        mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ, max abs {np.abs(got - want).max()}"
        for nm in RPT_PLANES:
            a, b = r.p_indirect.download_plane(nm), orpt.plane(nm)
            if nm == "A":
                a, b = a & 0xffffff, b & 0xffffff
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {f}: reservoir plane {nm} differs"
    assert got[..., :3].max() > 0
    # k > 2 reconnections (replay work lists) must be live on this scene
    k = (r.p_indirect.download_plane("A")[..., 0] & 0xf)
    assert (k > 0).any()


def test_baseline_config_atrium_path_tracer_and_gi_bit_exact(api, atrium):
    """Same scene, the wavefront K9 path tracer and ReSTIR GI (K10) at 320 x 180, 2 frames each."""
    from oracle import zro
    sc, o = atrium
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
    w, h = 320, 180
    rp = api.Renderer(sc, w, h, params=prm)
    rg = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_GI)
    ogi = zro.OracleRGI(o, w, h)
    for f in (1, 2):
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        o.presample(f, 128, 512)
        rp.render_frame(cb)
        _, planes = o.gbuffer(cb)
        want, cnt = o.pathtrace(cb, planes, prm)
        assert np.array_equal(rp.final().view(np.uint32), want.view(np.uint32)), f"K9 frame {f}"
        assert rp.p_indirect.read_counters() == (cnt[0], cnt[1])
        rg.render_frame(cb)
        wantg = ogi.render(cb, prm)
        assert np.array_equal(rg.final().view(np.uint32), wantg.view(np.uint32)), f"ReSTIR GI frame {f}"


# ---- at-size parity for BASELINE configs 4 and 5 without a full CPU frame: scattered windows of the full-resolution atrium (tests/window_parity.py)
class _GpuFullFrame:
    """the full-resolution frame on the device, staged like a tile of the multi-GPU split: what tests/window_parity.py compares the windows with"""

    def __init__(self, api, sc, W, H, prm):
        self.api = api
        self.r = api.Renderer(sc, W, H, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
        self.p = self.r.p_indirect
        self.p.enable_cost_map(2)      # ZR_COST_MAP_RAYS

    def stage1(self, cb):
        r = self.r
        r.p_gbuffer.render(cb, r.scene, r.gbuffer)
        r.p_prelight.render(cb, r.scene, None)
        self.p.read_cost_map(reset=True)
        self.p.render_stage(cb, r.scene, r.gbuffer, self.api.STAGE_TEMPORAL)

    def stage2(self, cb):
        self.p.render_stage(cb, self.r.scene, self.r.gbuffer, self.api.STAGE_SPATIAL)
        full_a = self.p.download_plane("A")
        assert not (full_a >> 24).any(), "plane A, byte 3: the library never writes it (zero since zr_pass_init)"
        return self.r.final(), self.p.read_cost_map(reset=True)

    def rect_planes(self, which, rect):
        """the reservoir planes of `rect` (global pixel coordinates) of set `which` (1 = post-temporal, between the stages; 0 = final) as the library
        packs them for a neighbouring device (zr_pass_halo_pack): plane after plane, each a dense w x h block"""
        import torch
        from tests.window_parity import HALO_PLANES
        n = rect[2] * rect[3]
        buf = torch.empty(n * 62, dtype=torch.uint8, device="cuda")
        self.p.halo_pack(self.r.gbuffer, self.api.HALO_POST_TEMPORAL if which == 1 else self.api.HALO_FINAL, rect, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        host, out, off = buf.cpu().numpy(), {}, 0
        for name, dt, ch, nbytes in HALO_PLANES:
            out[name] = host[off:off + n * nbytes].view(dt).reshape(rect[3], rect[2], ch).copy()
            off += n * nbytes
        assert off == host.size
        return out


def _atrium_windows_parity(api, atrium, W, H, windows, frames=3):
    from tests.window_parity import windows_parity
    sc, o = atrium
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
    cams = [(0.0, 0, -3.5)] * (frames - 1) + [(0.04, 0, -3.5)]      # the camera starts moving at the last frame: its temporal stage reprojects into the apron
    # oracle=o: the windows are rendered by the oracle too (zro.OracleRPTWindows) -- GPU == host-executed stage functions == oracle at the quoted size
    rays = windows_parity(_GpuFullFrame(api, sc, W, H, prm), sc, o.alias, W, H, windows, prm, cams, oracle=o)
    assert rays > 0


def test_atrium_1080p_windows_restir_pt_bit_exact(api, atrium):
    """BASELINE config 4 at its quoted size: the 380k-triangle / 100k-light atrium at 1920 x 1080, ReSTIR PT, 3 frames, eight scattered 64 x 64
    windows (one in the frame's corner, one ending in the partial 32 x 32 sort tile at the bottom edge: 1080 = 33 x 32 + 24)."""
    windows = [(0, 0, 64, 64), (960, 512, 64, 64), (1856, 0, 64, 64), (320, 256, 64, 64), (1408, 768, 64, 64), (640, 1024, 64, 56), (1856, 1024, 64, 56),
               (1152, 128, 64, 64)]
    _atrium_windows_parity(api, atrium, 1920, 1080, windows)


def test_atrium_2160p_windows_restir_pt_bit_exact(api, atrium):
    """BASELINE config 5's resolution: the atrium at 3840 x 2160, ReSTIR PT, 3 frames, four windows incl. one on the partial sort tiles of the
    bottom edge (2160 = 67 x 32 + 16) and one in the far corner."""
    windows = [(1920, 1056, 64, 64), (512, 320, 64, 64), (3008, 2112, 64, 48), (3776, 2112, 64, 48)]
    _atrium_windows_parity(api, atrium, 3840, 2160, windows)


@pytest.mark.parametrize("scene_name,golden", [("cornell.npz", "config1_cornell_256.npz"), ("cornell_emissive.npz", "config1_cornell_emissive_256.npz")])
def test_baseline_config1_goldens_on_gpu(api, scene_name, golden):
    """BASELINE config 1 (256 x 256, K9, 1 spp, frame 1, jitter off, default sun) against the COMMITTED fixtures of tests/golden/
    (tools/make_goldens.py): G-buffer planes, sky-view LUT, FINAL and ray counters, both Cornell scenes."""
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", scene_name))
    g = np.load(os.path.join(ROOT, "tests", "golden", golden))
    w = h = 256
    cb = scene_io.make_frame_constants(w, h, frame_num=1, num_emissives=len(sc.emissives))
    r = api.Renderer(sc, w, h, params=wire.default_params())
    r.render_frame(cb)
    planes, _ = r.gbuffer.download()
    for name, a in zip(wire.GB_PLANE_NAMES, planes):
        assert np.array_equal(a.view(np.uint8), g["gb_" + name].view(np.uint8)), f"G-buffer plane {name} differs from the golden"
    if "sky_lut" in g.files:
        assert np.array_equal(r.p_sky.download_plane("sky_lut")[..., 0], g["sky_lut"])
    got = r.final()
    assert np.array_equal(got.view(np.uint32), g["final"].view(np.uint32))
    assert tuple(r.p_indirect.read_counters()) == tuple(int(x) for x in g["counters"])


def test_emissive_di_and_compositing_show_the_sky_behind_missing_geometry(api, cornell_emissive):
    """ReSTIR_DI_Temporal.hlsl:274-286 and Compositing.hlsl:43-48: pixels without geometry take Light::Le_SkyWithSunDisk (sun disk or
    sky-view LUT along the pixel's camera ray) once the scene has a sky-view LUT -- accumulated over static frames by the DI pass, written
    directly by Compositing otherwise.  Emissive Cornell box at 16:9 (the box leaves side columns empty) with a ZR_PASS_SKY render."""
    from oracle import zro
    w, h = 128, 72
    osc = zro.OracleScene(cornell_emissive)
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    prm = wire.default_params_di()
    di = r.enable_direct(prm)
    comp = r.enable_compositing()
    r.p_sky = api.Pass(api.PASS_SKY, 256, 128)
    odi, opt = zro.OracleRDI(osc, w, h), zro.OracleRPT(osc, w, h)
    sun = np.array((0.05, -0.25, 0.97), np.float32)      # low sun in front of the camera: its disk lands in an empty side column or the sky does
    for f in range(1, 4):
        acc = f > 1      # frame 1: Compositing writes the sky; frames 2-3: the DI pass accumulates it
        cb = _frame(cornell_emissive, w, h, f, accumulate=int(acc), camera_static=int(acc), num_frames_static=max(1, f - 1))
        cb["sun_dir"] = sun / np.float32(np.linalg.norm(sun))
        r.render_frame(cb)
        osc.sky_lut(cb, 256, 128)
        want_di = odi.render(cb, prm)
        want_ind = opt.render(cb, wire.default_params())
        got_di = di.download()
        assert np.array_equal(got_di.view(np.uint32), want_di.view(np.uint32)), f"emissive DI, frame {f}"
        planes, _keep = osc.gbuffer(cb)
        want = zro.composite(osc, cb, planes[2].reshape(h, w), emissive_di=want_di, indirect=want_ind, out=None if f == 1 else want)
        got = comp.download()
        assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)), f"composited, frame {f}"
        miss = ((planes[2].reshape(h, w) & 0xff) & 4) != 0
        assert miss.any() and (got[miss][:, :3] > 0).all(), "sky must be visible where there is no geometry"


def test_fused_halo_transfer_and_rccl_exchange_on_one_gpu(api, cornell_emissive):
    """zr_pass_halo_pack_all / _unpack_all (one kernel for every plane x every rect) against the per-plane copy path, then the C++ HaloExchange
    (libzetaray_host.so: pack kernel -> grouped ncclSend / ncclRecv over RCCL -> unpack kernel, nothing waits on the host) run by a world
    of one rank that talks to itself: afterwards the reservoirs of the receiving strips equal those of the sending strips, plane by plane."""
    import torch
    from zetaray_amd import tiling
    w, h = 256, 128
    prm = wire.default_params()
    r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    for f in (1, 2):
        r.render_frame(_frame(cornell_emissive, w, h, f))
    p = r.p_indirect
    bpp = p.halo_bytes_per_pixel()
    assert bpp == 62
    rects = [(0, 0, 32, 128), (64, 32, 96, 64)]
    # (1) fused pack == per-plane pack, block by block
    offs, total = [], 0
    for (x0, y0, rw, rh) in rects:
        offs.append(total)
        total += (rw * rh * bpp + 15) // 16 * 16
    fused = torch.zeros(total, dtype=torch.uint8, device="cuda")
    p.halo_all(r.gbuffer, api.HALO_FINAL, [(x0, y0, rw, rh, o) for (x0, y0, rw, rh), o in zip(rects, offs)], fused.data_ptr(), total, pack=True)
    for (x0, y0, rw, rh), o in zip(rects, offs):
        ref = torch.zeros(rw * rh * bpp, dtype=torch.uint8, device="cuda")
        p.halo_pack(r.gbuffer, api.HALO_FINAL, (x0, y0, rw, rh), ref.data_ptr(), ref.numel())
        torch.cuda.synchronize()
        assert torch.equal(fused[o:o + ref.numel()], ref)
    # (2) self-exchange through RCCL: the strip x 0..31 and the block (64..127, 32..95) are sent, and received at x 224..255 and (128..191, 64..127)
    names = ["A", "B", "C", "D", "E", "F", "G"]
    before = {n: p.download_plane(n) for n in names}
    plan = [(0, (0, 0, 32, 128), (224, 0, 32, 128)), (0, (64, 32, 64, 64), (128, 64, 64, 64))]
    nh = tiling.NativeHalo(p, r.gbuffer, 0, 1, 0, plan)
    assert nh.send_bytes == (32 * 128 + 64 * 64) * bpp
    nh.run(api.HALO_FINAL)
    torch.cuda.synchronize()
    after = {n: p.download_plane(n) for n in names}
    nh.close()
    for n in names:
        a, b = before[n], after[n]
        assert np.array_equal(b[0:128, 224:256], a[0:128, 0:32]), n
        assert np.array_equal(b[64:128, 128:192], a[32:96, 64:128]), n
        untouched = np.ones((h, w), bool)
        untouched[0:128, 224:256] = False
        untouched[64:128, 128:192] = False
        assert np.array_equal(b[untouched], a[untouched]), n + ": pixels outside the receive rects changed"
    assert any(before[n][0:128, 0:32].any() for n in names)


@pytest.mark.parametrize("kind", ["restir_pt", "di"])
def test_halo_exchange_before_the_first_frame(api, cornell_emissive, kind):
    """TiledRestirPT's first-contact exchange (tiling.py: one exchange through the C++ RCCL node right after it is created, so that a transport that
    cannot work on this node is replaced before a frame depends on it) runs on a pass that has not rendered yet: the planes exist, the exchange
    succeeds and moves zeros, and the frames rendered afterwards are the ones a pass without that exchange renders."""
    import hashlib
    import torch
    from zetaray_amd import tiling
    w, h = 256, 128
    ip = wire.default_params()
    digests = []
    for trial in (False, True):
        r = api.Renderer(cornell_emissive, w, h, params=ip, integrator=api.INTEGRATOR_RESTIR_PT if kind == "restir_pt" else api.INTEGRATOR_PATH_TRACING)
        p = r.p_indirect
        if kind == "di":
            p = r.enable_direct(wire.default_params_di())
            r.skip_indirect = True
        if trial:
            nh = tiling.NativeHalo(p, r.gbuffer, 0, 1, 0, [(0, (0, 0, 32, 128), (224, 0, 32, 128))])
            nh.run(api.HALO_POST_TEMPORAL)
            torch.cuda.synchronize()
            nh.close()
        for f in (1, 2, 3):
            r.render_frame(_frame(cornell_emissive, w, h, f))
        digests.append(hashlib.sha1((p.download() if kind == "di" else r.final()).tobytes()).hexdigest())
    assert digests[0] == digests[1]


def test_device_refit_of_a_large_dynamic_scene(api):
    """zr_scene_update_instances refits the BVH on the device (triangles re-transformed, node boxes re-quantised level by level).  3000-triangle
    materials scene, three instances moving / rotating / scaling over 4 frames: G-buffer, ReSTIR PT radiance, reservoir planes and ray counters equal
    the oracle's (which rebuilds its own tree from the new matrices), and an update takes milliseconds, not a host rebuild."""
    import time
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    w, h = 96, 64
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    osc = zro.OracleScene(sc, force_bvh=True)
    opt = zro.OracleRPT(osc, w, h)
    assert r.scene.bvh_info()[0] > 100
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))        # the largest clutter instance (instance 0 is the room)
    assert sc.instance_num_tris[idx] > 200
    t0, xf = sc.instances["translation"][idx].copy(), {}
    prev = None
    for f in range(1, 5):
        if f >= 2:
            ang = 0.12 * (f - 1)
            q = np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32)
            scene_io.move_instance(sc, idx, translation=t0 + np.float32([0.08 * (f - 1), 0.03 * (f - 1), -0.05 * (f - 1)]), rotation=q,
                                   scale=np.float32([1.0 + 0.04 * (f - 1)] * 3), xform_of=xf)
            t_upd = time.perf_counter()
            r.scene.update_instances(sc.instances, sc.instance_to_world)
            dt = time.perf_counter() - t_upd
            assert dt < 0.25, f"update took {dt * 1e3:.1f} ms"
            osc.update_instances(sc.instances, sc.instance_to_world)
        cb = _frame(sc, w, h, f, cam_pos=(0, 0, -3.5))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb)
        want = opt.render(cb, prm)
        planes, _ = r.gbuffer.download()
        oplanes, _ = osc.gbuffer(cb)
        for n, a, b in zip(wire.GB_PLANE_NAMES, planes, oplanes):
            assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), f"frame {f}: G-buffer plane {n}"
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}"
    assert r.p_indirect.read_counters() is not None


def test_background_sah_rebuild_keeps_dynamic_frames_bit_exact(api):
    """zr_scene_set_background_rebuild: while instances move, the host's SAH builder runs on a thread on a snapshot of the
    transforms and a later update_instances swaps its tree in -- topology from the snapshot, triangles and boxes refit to the update at hand; the
    device refit covers every frame in between.  3000-triangle materials scene, three instances' worth of motion over 11 frames, an install every
    second update (the test waits for the builder so that the schedule is deterministic): G-buffer, ReSTIR PT radiance, reservoir planes and ray
    counters equal the oracle's every frame -- nothing depends on which tree a frame was traced through."""
    import time
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    w, h = 96, 64
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    r.scene.set_background_rebuild(True)
    osc = zro.OracleScene(sc, force_bvh=True)
    opt = zro.OracleRPT(osc, w, h)
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))
    t0, xf = sc.instances["translation"][idx].copy(), {}
    prev = None
    for f in range(1, 12):
        if f >= 2:
            ang = 0.12 * (f - 1)
            q = np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32)
            scene_io.move_instance(sc, idx, translation=t0 + np.float32([0.05 * (f - 1), 0.02 * (f - 1), -0.03 * (f - 1)]), rotation=q, xform_of=xf)
            r.scene.update_instances(sc.instances, sc.instance_to_world)
            osc.update_instances(sc.instances, sc.instance_to_world)
            t_wait = time.perf_counter()
            while r.scene.background_rebuild_stats()[2] == 1 and time.perf_counter() - t_wait < 20.0:
                time.sleep(0.005)
        cb = _chain(_frame(sc, w, h, f, cam_pos=(0, 0, -3.5)), prev)
        prev = cb.copy()
        r.p_indirect.read_counters(reset=True)
        r.render_frame(cb)
        want = opt.render(cb, prm)
        planes, _ = r.gbuffer.download()
        oplanes, _ = osc.gbuffer(cb)
        for n, a, b in zip(wire.GB_PLANE_NAMES, planes, oplanes):
            assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), f"frame {f}: G-buffer plane {n}"
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}"
        assert r.p_indirect.read_counters() == opt.counters, f"frame {f}: ray counters"
        for nm in ("A", "B", "C", "D", "E", "F", "G"):
            a, b = r.p_indirect.download_plane(nm), opt.plane(nm)
            if nm == "A":
                a, b = a & 0xffffff, b & 0xffffff
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {f}: reservoir plane {nm}"
    started, installed, _ = r.scene.background_rebuild_stats()
    assert started >= 3 and installed >= 3, (started, installed)


def test_device_built_bvh_traces_identically(api, monkeypatch):
    """ZR_BVH_BUILD=device: the acceleration structure is built on the GPU (LBVH: Morton sort + breadth-first 4-wide topology + per-level boxes,
    zr_tu_bvh.hip).  Query results do not depend on the tree, so 20 000 closest-hit rays against the oracle's own BVH2 and a ReSTIR PT sequence with
    materials / Russian roulette stay bit-exact; then the tree is REBUILT on the device every frame for a moving instance (ZR_SCENE_UPDATE=rebuild)."""
    from oracle import zro
    monkeypatch.setenv("ZR_BVH_BUILD", "device")
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    osc = zro.OracleScene(sc, force_bvh=True)
    handle = api.Scene(sc)
    nodes, tris, depth = handle.bvh_info()
    assert nodes > 300 and tris == sc.num_tris and 4 <= depth <= 20, (nodes, tris, depth)
    import torch
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform(-2.5, 2.5, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((n, 1), 1e-4), d, np.full((n, 1), 3.0e38)], 1).astype(np.float32)
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    api._check(api.lib().zr_trace_closest(handle.h, None, d_rays.data_ptr(), n, 3, d_hits.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_hits.cpu().numpy().view(np.uint32), osc.trace_closest(rays))
    handle.close()
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 6, 8
    _rpt_compare(api, sc, osc, 96, 64, prm, 3, cam=dict(cam_pos=(0, 0, -3.5)))
    # per-frame rebuild of a moving instance
    monkeypatch.setenv("ZR_SCENE_UPDATE", "rebuild")
    w, h = 96, 64
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    opt = zro.OracleRPT(osc, w, h)
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))
    t0, xf = sc.instances["translation"][idx].copy(), {}
    prev = None
    for f in range(1, 5):
        if f >= 2:
            ang = 0.12 * (f - 1)
            scene_io.move_instance(sc, idx, translation=t0 + np.float32([0.08 * (f - 1), 0.03 * (f - 1), -0.05 * (f - 1)]),
                                   rotation=np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32), xform_of=xf)
            r.scene.update_instances(sc.instances, sc.instance_to_world)
            osc.update_instances(sc.instances, sc.instance_to_world)
        cb = _chain(_frame(sc, w, h, f, cam_pos=(0, 0, -3.5)), prev)
        prev = cb.copy()
        r.render_frame(cb)
        want = opt.render(cb, prm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f} (device rebuild)"


def test_device_bvh_build_respects_the_depth_cap(api, monkeypatch):
    """The device builder never produces a tree deeper than the traversal stack can take (21 levels): a node whose key range could not fit below
    the cap if it were cut at a Morton bit is cut into four equal parts (zr_tu_bvh.hip; before round 5 a deeper tree was an error).  A cascade deep
    enough to reach level 21 needs > 60 key bits of structure, so the test lowers the cap instead (zr_debug_set_bvh_depth_cap): the 3000-triangle
    scene, whose LBVH is naturally ~10 levels deep, built with a cap of 7 -- the balanced cuts take over from level 1 or 2 on -- must come out at
    <= 7 levels, with every triangle in it (20 000 closest-hit rays == the oracle's BVH2) and a bit-exact ReSTIR PT sequence."""
    from oracle import zro
    import torch
    monkeypatch.setenv("ZR_BVH_BUILD", "device")
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    osc = zro.OracleScene(sc, force_bvh=True)
    natural = api.Scene(sc)
    _, _, depth0 = natural.bvh_info()
    natural.close()
    assert api.lib().zr_debug_set_bvh_depth_cap(7) == 0
    try:
        handle = api.Scene(sc)
        nodes, tris, depth = handle.bvh_info()
        assert tris == sc.num_tris and depth <= 7 < depth0, (nodes, tris, depth, depth0)
        rng = np.random.default_rng(6)
        n = 20000
        o = rng.uniform(-2.5, 2.5, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, np.full((n, 1), 1e-4), d, np.full((n, 1), 3.0e38)], 1).astype(np.float32)
        d_rays = torch.from_numpy(rays).cuda()
        d_hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        api._check(api.lib().zr_trace_closest(handle.h, None, d_rays.data_ptr(), n, 3, d_hits.data_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(d_hits.cpu().numpy().view(np.uint32), osc.trace_closest(rays))
        handle.close()
        prm = wire.default_params()
        _rpt_compare(api, sc, osc, 96, 64, prm, 2, cam=dict(cam_pos=(0, 0, -3.5)))
    finally:
        api.lib().zr_debug_set_bvh_depth_cap(0)


def test_moving_light_on_gpu(api):
    """zr_scene_update_emissives: the Cornell box's light quad translates and turns over frames 2-5 (its EmissiveTriangle records re-derived from the
    object-space ones like SceneCore::UpdateEmissivePositions does -- decode, transform, re-encode, pinned to the reference's code in
    tests/test_scene_io.py -- and its instance moved through zr_scene_update_instances so the BVH follows).  ReSTIR PT (reconnections onto the
    light: case 2 / 3 vertices, MoveXk) and ReSTIR DI (light samples reused across frames) stay bit-exact vs the oracle; the alias table is not
    rebuilt, like the reference."""
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_emissive.npz"))
    w, h = 128, 96
    prm, dprm = wire.default_params(), wire.default_params_di()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, h), zro.OracleRDI(osc, w, h)
    idx = [i for i in range(len(sc.instances)) if sc.instances["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    t0, xf = sc.instances["translation"][idx].copy(), {}
    before = sc.emissives.copy()
    prev = None
    for f in range(1, 6):
        if f >= 2:
            a = 0.2 * (f - 1)
            inst, xw, first, tris = scene_io.move_emissive_instance(sc, idx, translation=t0 + np.float32([0.05 * (f - 1), -0.02 * (f - 1), 0.03 * (f - 1)]),
                                                                     rotation=np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], np.float32), xform_of=xf)
            r.scene.update_emissives(tris, first); r.scene.update_instances(inst, xw)
            osc.update_emissives(tris, first); osc.update_instances(inst, xw)
        cb = _frame(sc, w, h, f)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb)
        want, want_di = opt.render(cb, prm), odi.render(cb, dprm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}: ReSTIR PT"
        assert np.array_equal(di.download().view(np.uint32), want_di.view(np.uint32)), f"frame {f}: ReSTIR DI"
        for nm in RPT_PLANES:
            a_, b_ = r.p_indirect.download_plane(nm), opt.plane(nm)
            if nm == "A":
                a_, b_ = a_ & 0xffffff, b_ & 0xffffff
            assert np.array_equal(a_.view(np.uint8), b_.view(np.uint8)), f"frame {f}: reservoir plane {nm}"
    assert not np.array_equal(before.view(np.uint8), sc.emissives.view(np.uint8))      # the light really moved


def test_stream_ordered_scene_updates_across_streams(api):
    """zr_scene_update_instances_async / _emissives_async: the moving light of the test above, but every update is ENQUEUED on one non-blocking
    stream and every frame rendered on another, six frames back to back without a host synchronisation in between (the library orders the two
    streams with events: a render waits for the last update, an update for the last render of each stream).  Afterwards every frame's ReSTIR PT
    and ReSTIR DI image -- copied out on the render stream -- equals the oracle's."""
    import torch
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_emissive.npz"))
    w, h = 128, 96
    prm, dprm = wire.default_params(), wire.default_params_di()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, h), zro.OracleRDI(osc, w, h)
    idx = [i for i in range(len(sc.instances)) if sc.instances["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    t0, xf = sc.instances["translation"][idx].copy(), {}
    s_upd, s_ren = torch.cuda.Stream(), torch.cuda.Stream()
    frames, wants, prev = [], [], None
    nbytes = w * h * 16
    import ctypes as C
    # the HIP runtime this process already uses (a second copy loaded by name would not know torch's streams)
    hip_path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    hip = C.CDLL(hip_path)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    for f in range(1, 7):
        if f >= 2:
            a = 0.2 * (f - 1)
            inst, xw, first, tris = scene_io.move_emissive_instance(sc, idx, translation=t0 + np.float32([0.05 * (f - 1), -0.02 * (f - 1), 0.03 * (f - 1)]),
                                                                     rotation=np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], np.float32), xform_of=xf)
            t_dev, i_dev, x_dev = tris.copy(), inst.copy(), np.array(xw, np.float32).copy()
            r.scene.update_emissives(t_dev, first, stream=s_upd.cuda_stream)
            r.scene.update_instances(i_dev, x_dev, stream=s_upd.cuda_stream)
            t_dev.view(np.uint8)[:] = 0xff; i_dev.view(np.uint8)[:] = 0xff; x_dev[:] = 0      # the caller's arrays may be reused at once (pinned staging ring)
            osc.update_emissives(tris, first); osc.update_instances(inst, xw)
        cb = _frame(sc, w, h, f)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb, stream=s_ren.cuda_stream)
        # snapshot both images on the render stream (device-to-device, stream-ordered)
        snap = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
        torch.cuda.current_stream().synchronize()          # the allocation's fill, not the renders
        pt_ptr, di_ptr = r.p_indirect.output_ptr()[0], di.output_ptr()[0]      # (output planes are allocated by the first render)
        assert hip.hipMemcpyAsync(snap.data_ptr(), pt_ptr, nbytes, 3, s_ren.cuda_stream) == 0
        assert hip.hipMemcpyAsync(snap.data_ptr() + nbytes, di_ptr, nbytes, 3, s_ren.cuda_stream) == 0
        frames.append(snap)
        wants.append((opt.render(cb, prm).copy(), odi.render(cb, dprm).copy()))
    torch.cuda.synchronize()
    for f, (snap, (want, want_di)) in enumerate(zip(frames, wants), 1):
        got = snap.cpu().numpy().view(np.float32).reshape(2, h, w, 4)
        assert np.array_equal(got[0].view(np.uint32), want.view(np.uint32)), f"frame {f}: ReSTIR PT"
        assert np.array_equal(got[1].view(np.uint32), want_di.view(np.uint32)), f"frame {f}: ReSTIR DI"


def test_emissive_material_change_rebuilds_the_alias_table(api):
    """SceneCore::UpdateEmissiveMaterial + PreLighting's stale-materials path (PreLighting.cpp:266): one of the Cornell light's two triangles gets
    8 x its strength at frame 3 (zr_scene_update_emissives with the rewritten record, zr_scene_invalidate_alias_table); the next PRELIGHTING render
    re-estimates the powers (K2) and rebuilds the table -- alias table, ReSTIR DI and ReSTIR PT equal the oracle's before and after."""
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_emissive.npz"))
    w, h = 96, 64
    prm, dprm = wire.default_params(), wire.default_params_di()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, h), zro.OracleRDI(osc, w, h)
    tables = []
    for f in range(1, 5):
        if f == 3:
            e = sc.emissives[0:1].copy()
            s_old = np.uint16(int(e["packed_b"][0]) >> 16).view(np.float16)
            s_new = int(np.float16(np.float32(s_old) * np.float32(8.0)).view(np.uint16))
            e["packed_b"] = (int(e["packed_b"][0]) & 0xFFFF) | (s_new << 16)
            e["packed_a"] = (int(e["packed_a"][0]) & 0x0FFFFFFF) | ((s_new & 0xF) << 28)
            sc.emissives[0:1] = e
            r.scene.update_emissives(e, 0); r.invalidate_alias_table()
            osc.update_emissives(e, 0); osc.rebuild_alias_table()
        cb = _frame(sc, w, h, f)
        r.render_frame(cb)
        tables.append(r.scene.get_alias_table().copy())
        assert np.array_equal(tables[-1].view(np.uint8), np.asarray(osc.alias).view(np.uint8)), f"frame {f}: alias table"
        want, want_di = opt.render(cb, prm), odi.render(cb, dprm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}: ReSTIR PT"
        assert np.array_equal(di.download().view(np.uint32), want_di.view(np.uint32)), f"frame {f}: ReSTIR DI"
    assert np.array_equal(tables[0].view(np.uint8), tables[1].view(np.uint8)) and not np.array_equal(tables[1].view(np.uint8), tables[2].view(np.uint8))


def test_deferred_alias_table_rebuild_never_blocks(api):
    """zr_scene_invalidate_alias_table_deferred (the reference's steady state, PreLighting.cpp:527-540): after an emissive-material change the OLD
    table keeps being sampled -- the frame rendered right after the change equals the oracle with the new light record and the old table -- and the
    rebuilt table arrives without any render call waiting: once the read-back has landed (a device synchronisation between
    the frames stands in for time) the next PRELIGHTING render uploads the rebuilt table, and every frame equals the oracle run with the same schedule."""
    import torch
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_emissive.npz"))
    w, h = 96, 64
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    osc = zro.OracleScene(sc)
    opt = zro.OracleRPT(osc, w, h)
    old_table = None
    for f in range(1, 8):
        if f == 3:
            e = sc.emissives[0:1].copy()
            s_old = np.uint16(int(e["packed_b"][0]) >> 16).view(np.float16)
            s_new = int(np.float16(np.float32(s_old) * np.float32(8.0)).view(np.uint16))
            e["packed_b"] = (int(e["packed_b"][0]) & 0xFFFF) | (s_new << 16)
            e["packed_a"] = (int(e["packed_a"][0]) & 0x0FFFFFFF) | ((s_new & 0xF) << 28)
            sc.emissives[0:1] = e
            old_table = r.scene.get_alias_table().copy()
            r.scene.update_emissives(e, 0); r.invalidate_alias_table(deferred=True)
            osc.update_emissives(e, 0)                       # the oracle keeps its old table for now
        if f == 4:
            osc.rebuild_alias_table()        # frame 3 started the read-back and the device went idle: frame 4's PRELIGHTING render picks it up
        cb = _frame(sc, w, h, f)
        r.render_frame(cb)
        torch.cuda.synchronize()             # (stands in for time passing; no render call above waited for the device)
        want = opt.render(cb, prm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}"
        if f == 3:
            assert np.array_equal(r.scene.get_alias_table().view(np.uint8), old_table.view(np.uint8)), "frame 3 still samples the old table"
    assert np.array_equal(r.scene.get_alias_table().view(np.uint8), np.asarray(osc.alias).view(np.uint8))
    assert not np.array_equal(old_table.view(np.uint8), np.asarray(osc.alias).view(np.uint8))


def test_material_edit_between_frames(api):
    """SceneCore::UpdateMaterial: at frame 3 every non-emissive material of the Cornell box turns into a rough metal with another base colour
    (zr_scene_update_materials), at frame 5 the original materials return; G-buffer, ReSTIR PT and ReSTIR DI equal the oracle's before and after
    (temporal reuse across the edits included; the material class goes plain -> general -> plain, the kernels PLAIN -> general at frame 3 -> PLAIN at frame 6)."""
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_emissive.npz"))
    w, h = 96, 64
    prm, dprm = wire.default_params(), wire.default_params_di()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, h), zro.OracleRDI(osc, w, h)
    imgs = []
    original = sc.materials.copy()
    for f in range(1, 8):
        if f == 5:
            # ... and back: the scene is of the plain class again, but frame 5's temporal reuse reads frame 4's G-buffer, whose pixels carry the metals' flags --
            # the pass keeps the general kernels for that frame (zr_gbuffer tracks the class each plane set was rendered under) and returns to the PLAIN ones at frame 6
            sc.materials[:] = original
            r.scene.update_materials(original, 0); osc.update_materials(original, 0)
            assert r.scene.material_class() == 1
        if f == 3:
            mats = sc.materials.copy()
            for i in range(1, len(mats)):
                if int(mats[i]["emissive_factor_normal_scale"]) & 0xFFFFFF:
                    continue      # the light keeps its material (its power enters the alias table)
                if i % 2:
                    mats[i] = scene_io.pack_material(base_color=(0.2 + 0.1 * (i % 5), 0.7, 0.3, 1.0), metallic=1.0, roughness=0.35)
                else:
                    # the metal bit alone, roughness kept: the temporal passes' similarity tests (roughness, transmission) still accept the pixel across the edit,
                    # so the frames after each edit shift samples onto surfaces of the OTHER material class (what the G-buffer's class tracking is for)
                    mats[i]["coat_color_flags"] |= np.uint32(1 << 24)
            sc.materials[:] = mats
            assert r.scene.material_class() == 1      # frames 1-2 ran the PLAIN kernel permutations ...
            r.scene.update_materials(mats, 0); osc.update_materials(mats, 0)
            assert r.scene.material_class() == 0      # ... frames 3-4 the general ones, over the same reservoirs
        cb = _frame(sc, w, h, f)
        r.render_frame(cb)
        planes, _ = r.gbuffer.download()
        oplanes, _ = osc.gbuffer(cb)
        for n, a, b in zip(wire.GB_PLANE_NAMES, planes, oplanes):
            assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), f"frame {f}: G-buffer plane {n}"
        want, want_di = opt.render(cb, prm), odi.render(cb, dprm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}: ReSTIR PT"
        assert np.array_equal(di.download().view(np.uint32), want_di.view(np.uint32)), f"frame {f}: ReSTIR DI"
        imgs.append(planes[0].copy())
    assert not np.array_equal(np.asarray(imgs[1]), np.asarray(imgs[2]))      # the base-colour plane changed with the edit


def test_bench_multi_rank_protocol_on_one_gpu():
    """`bench.py --gpus 2`, `--gpus 4` and `--gpus 8` end to end on a box with ONE GPU: the ranks are real processes launched by torch.distributed.run exactly as the
    driver launches them, but share device 0 and talk over gloo (ZR_BENCH_SHARED_GPU=1, halo strips staged through the host).  What runs is the
    multi-GPU orchestration -- tile split, halo plan, probe frames with the cost map, choose_layout + re-tiling, barrier / max-over-ranks timing,
    whole-job ray count -- everything except RCCL itself (exercised by test_fused_halo_transfer_and_rccl_exchange_on_one_gpu)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, ZR_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    singles = []
    # (the last run adds the denoise pass: BASELINE config 5's shape on two ranks -- the three denoise halo exchanges of tiling.denoise_schedule per frame)
    # ... and BASELINE config 2 (the DI passes alone, sharded as the pass whose reservoirs cross tile borders: `--config 2a / 2b --gpus 2`)
    for n, port, settle, more in ((1, 0, 4, []), (1, 0, 16, []), (2, 29631, 4, []), (4, 29632, 4, []), (8, 29634, 4, []), (2, 29633, 4, ["--denoise"]),
                                  (2, 29635, 4, ["--config", "2a"]), (2, 29636, 4, ["--config", "2b"]), (8, 29637, 4, ["--denoise"])):
        # n == 2 without --denoise is started as plain `python bench.py --gpus 2` (no launcher: bench.py starts its ranks itself, the shape of
        # the driver's N = 1 command); the other N > 1 runs go through torch.distributed.run the way the driver launches them
        self_launch = n == 2 and not more
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] if (n == 1 or self_launch) else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
             os.path.join(ROOT, "bench.py")]
        cmd += ["--gpus", str(n), "--steps", "4", "--warmup", "2", "--settle", str(settle), "--ramp-s", "0", "--no-general-kernels", "--width", "512", "--height", "288", "--no-cpu-baseline"] + more
        res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "strong"
        assert ("denoise pass" in d["config"]["workload"]) == ("--denoise" in more)
        # the device as the process found it: the probes ran and say something
        ds = d["device_state"]
        assert ds["compute_units"] > 0 and ds["probe_copy_GBs"] > 100 and ds["probe_fma_tflops"] > 1 and ds["probe_sclk_mhz_under_load"] > 100, ds
        if n == 1:
            assert d["config"]["kernel_class"] == "plain" and "multi_gpu" not in d
            singles.append(d["config"]["rays_per_frame"])
            continue
        # first-contact block of an N > 1 line: one row per rank with its device, tile, halo bytes and the time of one exchange; per-rank probe rows
        mg = d["multi_gpu"]
        assert mg["communicator_size"] == n and len(mg["ranks"]) == n and sorted(q["rank"] for q in mg["ranks"]) == list(range(n)), mg
        assert all(q["exchange_error"] is None and q["exchange_ms"] > 0 and q["halo_bytes_sent_per_exchange"] > 0 and q["peers"] >= 1 and q["transport"] == "torch_p2p" for q in mg["ranks"]), mg
        assert mg["exchanges_per_frame"] >= 1 and len(ds["ranks"]) == n and all(q["copy_GBs"] > 100 for q in ds["ranks"]), (mg, ds["ranks"])
        if "--config" in more:
            # config 2 on two ranks: the DI pass is the halo pass (24 / 13 B per pixel, one exchange per frame), every rank counted rays
            assert mg["halo_pass"] == ("di" if more[1] == "2a" else "sky_di") and mg["ranks"][0]["halo_bytes_per_pixel"] == (24 if more[1] == "2a" else 13), mg
            assert mg["exchanges_per_frame"] == 1 and d["config"]["integrator"] in ("+restir_di", "+sky_di"), d["config"]
            continue
        if "--denoise" in more:
            assert mg["exchanges_per_frame"] >= 4, mg      # post-temporal + the three denoise exchanges of tiling.denoise_schedule
        assert "screen tiles" in d["config"]["parallelism"] and ("cost-balanced" in d["config"]["parallelism"] or "equal-area" in d["config"]["parallelism"])
        assert d["config"]["halo_transport"] == "torch_p2p"      # (this rig; rccl_cpp on distinct devices: test_bench_two_ranks_over_rccl_when_two_gpus_are_visible)
        # the whole-job roofline object of an N > 1 line: every rank's K11 timed on its tile, N x the HBM peak
        rf = d["roofline"]
        assert rf["kernel"] == "rpt_pathtrace" and rf["peak"] == 8000.0 * n and 0 < rf["frac"] < 1 and len(rf["per_rank_avg_launch_ms"]) == n and min(rf["per_rank_avg_launch_ms"]) > 0
        # The same frames are traced whatever the split, so the whole-job ray count per frame is the single-device one with equally aged
        # reservoirs: 4 settle frames when the probe's cost map made the ranks re-tile (fresh passes), 12 probe + 4 settle frames when they kept
        # the grid.  (Equal frame numbers give bit-identical frames, hence equal counts; the probe frames shift the numbering in the second case.)
        rel = min(abs(d["config"]["rays_per_frame"] / s1 - 1) for s1 in singles)
        assert rel < 0.03, (d["config"]["rays_per_frame"], singles, d["config"]["parallelism"])


_VARIANT_SNIPPET = r"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from zetaray_amd import api, scene_io, wire
sc = scene_io.load_npz(os.path.join(sys.argv[1], "tests", "golden", sys.argv[2]))
w, h = 208, 128
prm = wire.default_params()
r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
hh = hashlib.sha1()
for f in range(1, 5):
    cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0.03 * f, 1.2, -4.043))
    r.render_frame(cb)
    hh.update(r.final().tobytes())
    for nm in ("B", "C", "D", "E", "F", "G", "target"):
        hh.update(r.p_indirect.download_plane(nm).tobytes())
print("DIGEST", hh.hexdigest())
"""


@pytest.mark.parametrize("scene", ["cornell_emissive.npz", "cornell.npz"])
def test_k11_kernel_variants_produce_identical_frames(scene):
    """The experiments build (libzetaray_amd_exp.so) has three selectable forms of K11 (ZR_K11, read once per process): the inline megakernel (the product's), block-pooled traces (`pool`, emissive
    permutation) and a kernel per bounce with path compaction (`compact`).  Four ReSTIR PT frames with a moving camera: radiance and every
    reservoir plane hash to the same digest under each."""
    import subprocess
    import sys
    exp = os.path.join(ROOT, "zetaray_amd", "libzetaray_amd_exp.so")
    if not os.path.exists(exp):
        pytest.skip("the K11 forms other than the inline megakernel exist only in the experiments build (make -C zetaray_amd/csrc experiments)")
    digests = {}
    for mode in ("inline", "compact", "pool"):
        res = subprocess.run([sys.executable, "-c", _VARIANT_SNIPPET, ROOT, scene], env=dict(os.environ, ZR_K11=mode, ZETARAY_AMD_LIB=exp), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
        digests[mode] = [l for l in res.stdout.splitlines() if l.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The real multi-GPU path -- one process per GPU, halo strips through the C++ HaloExchange node over RCCL (grouped ncclSend / ncclRecv on the
    pass's stream, unique id bootstrapped over the launcher's process group): `bench.py --gpus 2` launched the way the driver launches it.  Needs two
    devices; on a one-GPU box the protocol is covered by test_bench_multi_rank_protocol_on_one_gpu and RCCL itself by the one-rank exchange test."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--settle", "4", "--width", "512", "--height", "288", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["halo_transport"] == "rccl_cpp", d["config"]


def test_final_halo_is_exchanged_only_by_frames_that_read_it(api, cornell_emissive, oracle_emissive):
    """TiledRestirPT.render_frame's exchange policy (tiling.render_frame_in_process is the same policy for tile objects in one process): the
    FINAL halo moves at the start of the frame that reprojects across tiles -- moving camera: two exchanges per frame, unmoved camera and
    scene: one -- and the stitched image stays bit-identical to the full-frame oracle through moving -> static -> moving -> moved-light frames."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 200, 120, 4
    prm = wire.default_params()
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, world, r, params=prm) for r in range(world)]
    o = zro.OracleRPT(oracle_emissive, w, h)
    cams = [(0.05, 1.2, -4.02), (0.10, 1.2, -4.00), (0.10, 1.2, -4.00), (0.10, 1.2, -4.00), (0.16, 1.2, -3.98), (0.16, 1.2, -3.98)]
    prev, counts = None, []
    for f, cam in enumerate(cams, 1):
        cb = _frame(cornell_emissive, w, h, f, cam_pos=cam)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        counts.append(tiling.render_frame_in_process(ranks, cb))
        want = o.render(cb, prm)
        img = np.zeros_like(want)
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
    # frame 1: nothing to fetch yet; 2, 5: the camera moved; 3, 4, 6: it did not
    assert counts == [1, 2, 1, 1, 2, 1], counts


def test_tile_split_with_a_thin_lens_camera_exchanges_the_final_halo(api, cornell_emissive, oracle_emissive):
    """with cb.dof the primary hit lies off the pinhole ray the reprojection assumes, so the G-buffer's motion vectors are non-zero on a
    perfectly static frame and the temporal stage reads neighbouring pixels' history across tile borders.  The exchange policy must therefore move the
    FINAL halo every frame (two exchanges per frame from frame 2 on) although neither view, jitter nor scene change -- and the stitched image of four
    tiles stays bit-identical to the full-frame oracle."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 200, 120, 4
    prm = wire.default_params()
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, world, r, params=prm) for r in range(world)]
    o = zro.OracleRPT(oracle_emissive, w, h)
    prev, counts = None, []
    for f in range(1, 5):
        cb = _frame(cornell_emissive, w, h, f)
        cb["dof"], cb["focus_depth"], cb["lens_radius"] = 1, 3.0, 0.08
        cb = _chain(cb, prev)
        prev = cb.copy()
        counts.append(tiling.render_frame_in_process(ranks, cb))
        want = o.render(cb, prm)
        img = np.zeros_like(want)
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
    assert counts == [1, 2, 2, 2], counts


def test_tile_split_with_two_spatial_rounds(api, cornell_emissive, oracle_emissive):
    """num_spatial_passes = 2 on tiles: the second round reads the first round's outputs at neighbouring pixels, so it is its own stage
    (ZR_STAGE_SPATIAL2) behind one more exchange of the set the next stage reads.  Four tile objects, moving camera: stitched radiance and reservoir
    planes bit-identical to the full-frame oracle (which runs the reference's host loop), three exchanges per frame once reuse is on."""
    from oracle import zro
    from zetaray_amd import tiling
    w, h, world = 200, 120, 4
    prm = wire.default_params()
    prm.num_spatial_passes = 2
    ranks = [tiling.TiledRestirPT(cornell_emissive, w, h, world, r, params=prm) for r in range(world)]
    o = zro.OracleRPT(oracle_emissive, w, h)
    prev, counts = None, []
    for f in range(1, 5):
        cb = _chain(_frame(cornell_emissive, w, h, f, cam_pos=(0.04 * f, 1.2, -4.03)), prev)
        prev = cb.copy()
        counts.append(tiling.render_frame_in_process(ranks, cb))
        want = o.render(cb, prm)
        img = np.zeros_like(want)
        planes = {nm: np.zeros_like(o.plane(nm)) for nm in ("A", "B", "C", "D", "E", "F", "G")}
        for r in ranks:
            (x0, y0, tw, th), t = r.final_tile()
            img[y0:y0 + th, x0:x0 + tw] = t
            ex0, ey0 = r.ext[0], r.ext[1]
            for nm in planes:
                planes[nm][y0:y0 + th, x0:x0 + tw] = r.r.p_indirect.download_plane(nm)[y0 - ey0:y0 - ey0 + th, x0 - ex0:x0 - ex0 + tw]
        mism = int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert mism == 0, f"frame {f}: {mism} pixels differ"
        for nm, got in planes.items():
            a, b = got, o.plane(nm)
            if nm == "A":
                a, b = a & 0xffffff, b & 0xffffff
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {f}: reservoir plane {nm}"
    assert counts == [2, 3, 3, 3], counts


def test_restir_pt_one_round_grid_is_bit_exact(api, cornell_emissive, oracle_emissive):
    """512 x 480 is 3840 one-wave blocks of K11: more than the 3-wave build keeps resident (3072), fewer than the 4-wave build does (4096) -- the
    grid size at which the pass switches K11 to its 4-wave build (zr_api.hip FewerRoundsAtFourWaves; an 8-way tile of a 1080p frame is such a
    grid).  Same planes, same radiance as the oracle."""
    got = _rpt_compare(api, cornell_emissive, oracle_emissive, 512, 480, wire.default_params(), 2)
    assert got[..., :3].max() > 0


def test_frames_without_any_geometry_on_gpu(api, cornell_emissive, oracle_emissive):
    """Every primary ray misses (the camera looks straight up from outside the box): ReSTIR PT, ReSTIR GI and the K9 path tracer over three frames
    -- work lists empty, every wave of every kernel idle after its first test -- still equal the oracle, with zero radiance and only primary rays."""
    from oracle import zro
    w, h = 208, 96
    prm = wire.default_params()
    cam = dict(view_dir=(0, 1, 0), up=(0, 0, 1))
    got = _rpt_compare(api, cornell_emissive, oracle_emissive, w, h, prm, 3, cam=cam)
    assert not got[..., :3].any()
    for integ in (api.INTEGRATOR_RESTIR_GI, api.INTEGRATOR_PATH_TRACING):
        r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=integ)
        for f in range(1, 4):
            r.render_frame(_frame(cornell_emissive, w, h, f, **cam))
        assert not r.final()[..., :3].any()
        n_closest, n_shadow = r.p_indirect.read_counters()
        assert n_shadow == 0 and n_closest == 0, (integ, n_closest, n_shadow)


def test_scene_without_bvh_nodes_on_gpu(api, cornell_emissive):
    """Six triangles (a wall, the light, a wall of the Cornell box): BvhBuilder emits no nodes and the kernels traverse the single leaf that holds
    everything (kWholeSceneLeaf).  G-buffer, ReSTIR PT (planes included) and the K9 path tracer == the oracle."""
    from oracle import zro
    from tests.test_rpt_cpu import _six_triangle_scene
    sc = _six_triangle_scene(cornell_emissive)
    osc = zro.OracleScene(sc)
    w, h = 200, 120
    prm = wire.default_params()
    got = _rpt_compare(api, sc, osc, w, h, prm, 3)
    assert (got[..., :3].sum(-1) > 0).sum() > 200
    r = api.Renderer(sc, w, h, params=prm)
    assert r.scene.bvh_info()[0] == 0
    cb = _frame(sc, w, h, 1)
    r.render_frame(cb)
    _, planes = osc.gbuffer(cb)
    want, _ = osc.pathtrace(cb, planes, prm)
    assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32))


def test_pick_pixel_reports_the_mesh_under_the_cursor(api, cornell_emissive, oracle_emissive):
    """zr_pass_pick_pixel / zr_pass_read_pick / zr_pass_clear_pick = GBufferRT::PickPixel, the pick read-back buffer, ClearPick (GBufferRT.h:36-46;
    GBufferRT_Inline.hlsl:241-242 writes hitMeshIdx or UINT32_MAX).  Cornell box and the 3000-triangle materials scene: picked mesh == the oracle's for a
    grid of pixels, a miss reads UINT32_MAX, on a screen tile the pixel is named in render-target coordinates (and a tile that does not hold it has
    nothing to read), the G-buffer planes are what they are without a pick, and after ClearPick the pass stops writing."""
    from oracle import zro
    w, h = 160, 96
    for sc, osc, cam in ((cornell_emissive, oracle_emissive, {}), (None, None, dict(cam_pos=(0, 0, -3.5)))):
        if sc is None:
            sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
            osc = zro.OracleScene(sc, force_bvh=True)
        r = api.Renderer(sc, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
        cb = _frame(sc, w, h, 2, **cam)
        seen = set()
        for y in range(3, h, 13):
            for x in range(2, w, 17):
                r.p_gbuffer.pick_pixel(x, y)
                r.p_gbuffer.render(cb, r.scene, r.gbuffer)
                got, want = r.p_gbuffer.read_pick(), osc.pick(cb, x, y)
                assert got == want, (x, y, got, want)
                seen.add(got)
        assert len(seen - {0xffffffff}) >= 5
        planes, _ = r.gbuffer.download()
        oplanes, _ = osc.gbuffer(cb)
        for n, a, b in zip(wire.GB_PLANE_NAMES, planes, oplanes):
            assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), n
        r.p_gbuffer.clear_pick()
    # a miss
    cb_out = _frame(cornell_emissive, w, h, 2, cam_pos=(0.0, 1.0, -30.0), view_dir=(0, 0, -1))
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    r.p_gbuffer.pick_pixel(7, 9)
    r.p_gbuffer.render(cb_out, r.scene, r.gbuffer)
    assert r.p_gbuffer.read_pick() == 0xffffffff == oracle_emissive.pick(cb_out, 7, 9)
    # ClearPick: a later render over another view does not overwrite the value
    r.p_gbuffer.clear_pick()
    cb = _frame(cornell_emissive, w, h, 2)
    r.p_gbuffer.render(cb, r.scene, r.gbuffer)
    assert r.p_gbuffer.read_pick() == 0xffffffff
    # screen tile: render-target coordinates; a tile that does not hold the pixel reports that nothing was written
    tile = (64, 32, 96, 64)
    rt = api.Renderer(cornell_emissive, tile[2], tile[3], params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT, tile_origin=tile[:2])
    rt.p_gbuffer.pick_pixel(100, 60)
    rt.p_gbuffer.render(cb, rt.scene, rt.gbuffer)
    assert rt.p_gbuffer.read_pick() == oracle_emissive.pick(cb, 100, 60)
    rt.p_gbuffer.pick_pixel(10, 10)
    rt.p_gbuffer.render(cb, rt.scene, rt.gbuffer)
    with pytest.raises(RuntimeError):
        rt.p_gbuffer.read_pick()


def test_background_rebuild_edge_cases(api, cornell_emissive):
    """zr_scene_set_background_rebuild at its edges: (a) a scene of small instances in which every instance but one ends up with a subtree of its own (the boxes
    and walls of the Cornell scene all move a little each frame; only the light is left for the common tree), installs waited for; (b) the switch turned off
    while a build is in flight -- the finished tree is never installed and later updates keep refitting; (b') updates that repeat the same matrices start no build once the resting pose has its tree; (c) a scene below the
    builder's node threshold never starts a build.  G-buffer and ReSTIR PT radiance == oracle on every frame."""
    import copy
    import time
    from oracle import zro
    w, h = 96, 64
    prm = wire.default_params()

    def wait_builder(scene):
        t0 = time.perf_counter()
        while scene.background_rebuild_stats()[2] == 1 and time.perf_counter() - t0 < 20.0:
            time.sleep(0.005)

    def run(sc, frames, mover, switch_off_at=None, cam=None):
        r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
        r.scene.set_background_rebuild(True)
        osc = zro.OracleScene(sc, force_bvh=True)
        opt = zro.OracleRPT(osc, w, h)
        prev, xf = None, {}
        for f in range(1, frames + 1):
            if f >= 2:
                mover(sc, f, xf)
                r.scene.update_instances(sc.instances, sc.instance_to_world)
                osc.update_instances(sc.instances, sc.instance_to_world)
                if switch_off_at is None or f < switch_off_at:
                    wait_builder(r.scene)
                if f == switch_off_at:
                    r.scene.set_background_rebuild(False)
            cb = _chain(_frame(sc, w, h, f, **(cam or {})), prev)
            prev = cb.copy()
            r.render_frame(cb)
            want = opt.render(cb, prm)
            planes, _ = r.gbuffer.download()
            oplanes, _ = osc.gbuffer(cb)
            for n, a, b in zip(wire.GB_PLANE_NAMES, planes, oplanes):
                assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), f"frame {f}: G-buffer plane {n}"
            assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}"
        wait_builder(r.scene)
        return r.scene.background_rebuild_stats()

    # (a) every instance of the Cornell scene moves (by a hair: the light's emissive records are not updated, so it keeps its place)
    sc = copy.copy(cornell_emissive)
    sc.instances, sc.instance_to_world, sc._desc = cornell_emissive.instances.copy(), cornell_emissive.instance_to_world.copy(), None
    light = [i for i in range(len(sc.instances)) if not (sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE)]
    t0 = sc.instances["translation"].copy()

    def all_move(s, f, xf):
        for i in range(len(s.instances)):
            if i not in light:
                scene_io.move_instance(s, i, translation=t0[i] + np.float32([0.002 * (f - 1) * ((i % 3) - 1), 0.0, 0.001 * (f - 1)]), xform_of=xf)
    started, installed, _ = run(sc, 7, all_move)
    assert started >= 2 and installed >= 2, (started, installed)

    # (b) switched off while the builder runs: nothing more is installed, frames stay exact
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))
    t1 = sc.instances["translation"][idx].copy()

    def one_moves(s, f, xf):
        scene_io.move_instance(s, idx, translation=t1 + np.float32([0.04 * (f - 1), 0.0, -0.02 * (f - 1)]), xform_of=xf)
    started, installed, building = run(sc, 8, one_moves, switch_off_at=4, cam=dict(cam_pos=(0, 0, -3.5)))
    assert building != 1 and installed <= started and installed <= 2, (started, installed, building)

    # (b') the instance comes to rest after frame 3 while the host keeps handing over (the same) matrices every frame: one more build for the resting
    # pose, then none
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    t1 = sc.instances["translation"][idx].copy()

    def moves_then_rests(s, f, xf):
        k = min(f, 3)
        scene_io.move_instance(s, idx, translation=t1 + np.float32([0.04 * (k - 1), 0.0, -0.02 * (k - 1)]), xform_of=xf)
    started, installed, building = run(sc, 9, moves_then_rests, cam=dict(cam_pos=(0, 0, -3.5)))
    assert started == 2 and installed == 2 and building == 0, (started, installed, building)

    # (c) six triangles: no nodes, nothing to build
    from tests.test_rpt_cpu import _six_triangle_scene
    sc = _six_triangle_scene(cornell_emissive)
    t2 = sc.instances["translation"][0].copy()

    def wall_moves(s, f, xf):
        scene_io.move_instance(s, 0, translation=t2 + np.float32([0.0, 0.0, 0.01 * (f - 1)]), xform_of=xf)
    started, installed, _ = run(sc, 4, wall_moves)
    assert started == 0 and installed == 0


def test_argument_validation_sweep(api, cornell_emissive):
    """Every misuse below must come back as a ZetaRayError with a message -- never a crash, a hang or a silent no-op -- and must leave the objects
    usable: the frame rendered afterwards equals the one rendered before.  (The reference aborts through Check(); this ABI returns error codes)"""
    import torch
    w, h = 64, 64
    prm = wire.default_params()
    r = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    cb = _frame(cornell_emissive, w, h, 1)
    r.render_frame(cb)
    before = r.final().copy()
    E = api.ZetaRayError
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")

    def bad(fn, *a, **k):
        with pytest.raises(E) as ei:
            fn(*a, **k)
        assert str(ei.value) and "error" in str(ei.value)

    # scene updates: wrong counts / ranges
    bad(r.scene.update_instances, cornell_emissive.instances[:-1], cornell_emissive.instance_to_world[:-1])
    bad(r.scene.update_emissives, cornell_emissive.emissives, first=1)                                  # [1, n + 1) exceeds the scene's triangles
    bad(r.scene.update_materials, cornell_emissive.materials, first=len(cornell_emissive.materials))
    # passes: wrong kind for the call
    bad(r.p_indirect.pick_pixel, 3, 3)
    bad(r.p_gbuffer.pick_pixel, 0xffff, 1)
    bad(r.p_gbuffer.read_pick)                                                                          # nothing picked yet
    bad(r.p_indirect.render_stage, cb, r.scene, r.gbuffer, 0)                                           # no stage selected
    bad(r.p_indirect.render_stage, cb, r.scene, r.gbuffer, api.STAGE_DENOISE_TEMPORAL)                  # a denoise step bit on the integrator
    bad(r.p_indirect.set_input, 0, 0)                                                                   # not a compositing / post pass
    bad(r.p_indirect.download_raw, 9999, np.float32, (h, w, 4))                                         # unknown output id
    # parameters out of range
    for field, value in (("max_non_tr_bounces", 0), ("max_glossy_tr_bounces", 99), ("m_max_temporal", 0), ("num_spatial_passes", 3), ("tex_filter", 17)):
        p2 = wire.default_params()
        setattr(p2, field, value)
        bad(r.p_indirect.set_params, p2)
    p2 = wire.default_params()
    p2.presampling, p2.num_sample_sets = 1, 0
    bad(r.p_indirect.set_params, p2)
    # tiles: unaligned origin / owned rect, rects outside the pass, a transfer buffer that is too small
    bad(r.p_indirect.set_owned_rect, 16, 0, 32, 32)
    bad(r.p_indirect.set_owned_rect, 0, 0, 32, 0)
    bad(r.p_indirect.set_owned_rect, 0, 0, 4096, 4096)
    bad(r.p_indirect.halo_pack, r.gbuffer, api.HALO_POST_TEMPORAL, (48, 48, 32, 32), buf.data_ptr(), buf.numel())      # leaves the 64 x 64 pass
    bad(r.p_indirect.halo_pack, r.gbuffer, api.HALO_POST_TEMPORAL, (0, 0, 32, 32), buf.data_ptr(), 16)                    # 32 x 32 x 62 B do not fit 16 B
    bad(r.p_indirect.halo_pack, r.gbuffer, 77, (0, 0, 32, 32), buf.data_ptr(), buf.numel())                                # unknown halo set
    bad(r.p_indirect.resize, 0, 16)
    # a G-buffer of another size than the pass
    gb2 = api.GBuffer(32, 32)
    bad(r.p_indirect.render, _frame(cornell_emissive, w, h, 2), r.scene, gb2)
    # ... and everything still works: frame 1 again on fresh temporal state equals the first render
    r.p_indirect.reset_temporal()
    r2 = api.Renderer(cornell_emissive, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    r2.render_frame(cb)
    assert np.array_equal(r2.final().view(np.uint32), before.view(np.uint32))
    r.p_gbuffer.render(cb, r.scene, r.gbuffer)
    r.p_indirect.render(cb, r.scene, r.gbuffer)
    assert np.array_equal(r.final().view(np.uint32), before.view(np.uint32))


def test_argument_validation_sweep_of_the_other_passes(api, cornell_emissive):
    """The same for the post / direct-lighting passes and the scene object: unbound inputs, sizes that do not match, out-of-range parameters."""
    w, h = 64, 64
    E = api.ZetaRayError
    r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    cb = _frame(cornell_emissive, w, h, 1)
    r.render_frame(cb)

    def bad(fn, *a, **k):
        with pytest.raises(E) as ei:
            fn(*a, **k)
        assert str(ei.value)

    for kind in (api.PASS_TAA, api.PASS_DENOISE, api.PASS_AUTO_EXPOSURE, api.PASS_DISPLAY):
        p = api.Pass(kind, w, h)
        bad(p.render, cb, r.scene, r.gbuffer)                              # no input bound
        bad(p.set_input, 99, r.p_indirect.output_ptr()[0])                 # unknown input id
    dn = api.Pass(api.PASS_DENOISE, w, h)
    for field, value in (("svgf_iterations", 9), ("svgf_normal_power_log2", 40)):
        p2 = wire.default_params()
        setattr(p2, field, value)
        bad(dn.set_params, p2)
    dn.set_input(api.IN_DENOISE_SIGNAL, r.p_indirect.output_ptr()[0])
    gb2 = api.GBuffer(32, 32)
    bad(dn.render, cb, r.scene, gb2)                                       # G-buffer of another size
    comp = api.Pass(api.PASS_COMPOSITING, w, h)
    bad(comp.render, cb, r.scene, gb2)
    di = api.Pass(api.PASS_DI_EMISSIVE, w, h)
    p2 = wire.default_params_di()
    p2.m_max_temporal = 31
    bad(di.set_params, p2)
    sdi = api.Pass(api.PASS_DI_SKY, w, h)
    p2 = wire.default_params_sky_di()
    p2.m_max_spatial = 16
    bad(sdi.set_params, p2)
    # scene: an alias table of the wrong length, a voxel grid that was never built
    bad(r.scene.set_alias_table, r.scene.get_alias_table()[:-1])
    bad(r.scene.get_light_voxel_grid, (4, 4, 4))
    bad(r.scene.get_presampled_sets, 4, 8)
    # the renderer is unharmed
    r2 = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
    r2.render_frame(cb)
    r.p_indirect.reset_temporal()
    r.render_frame(cb)
    assert np.array_equal(r.final().view(np.uint32), r2.final().view(np.uint32))


def test_distinct_handles_driven_from_concurrent_host_threads(api, cornell_emissive):
    """The threading contract of the boundary (the reference records its passes on worker threads, nodes of one batch concurrently;
    "distinct handles may be driven from distinct host threads concurrently; a single handle is externally serialised").  Three host threads,
    each with a scene + G-buffer + passes of its own and its own non-blocking stream -- ReSTIR PT, ReSTIR GI and the path tracer, 5 frames with a
    moving camera, creation and destruction inside the threads, three rounds -- produce what the same work produces on one thread.  Then the
    reference's own concurrency: ONE scene and G-buffer, the DirectLighting and IndirectLighting passes of a frame enqueued from two threads on two streams."""
    import threading
    import torch
    w, h, n = 96, 64, 5
    sc_syn = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    jobs = [(cornell_emissive, api.INTEGRATOR_RESTIR_PT, {}), (cornell_emissive, api.INTEGRATOR_RESTIR_GI, {}), (sc_syn, api.INTEGRATOR_PATH_TRACING, dict(cam_pos=(0, 0, -3.5)))]

    def frames_of(sc, cam):
        prev, out = None, []
        for f in range(1, n + 1):
            kw = dict(cam)
            kw["cam_pos"] = tuple(np.float32(kw.get("cam_pos", (0.0, 1.2, -4.043))) + np.float32([0.02 * f, 0, 0]))
            cb = _chain(_frame(sc, w, h, f, **kw), prev)
            prev = cb.copy()
            out.append(cb)
        return out

    def work(job, out, idx, use_stream):
        sc, integ, cam = job
        try:
            st = torch.cuda.Stream() if use_stream else None
            r = api.Renderer(sc, w, h, params=wire.default_params(), integrator=integ)
            for cb in frames_of(sc, cam):
                r.render_frame(cb, stream=st.cuda_stream if st is not None else None)
            if st is not None:
                st.synchronize()
            out[idx] = r.final().copy()
        except Exception as e:      # surfaced by the assert below
            out[idx] = e

    serial = [None] * len(jobs)
    for i, j in enumerate(jobs):
        work(j, serial, i, False)
    assert all(isinstance(a, np.ndarray) for a in serial), serial
    for rnd in range(3):
        got = [None] * len(jobs)
        ts = [threading.Thread(target=work, args=(j, got, i, True)) for i, j in enumerate(jobs)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(120)
        for i in range(len(jobs)):
            assert isinstance(got[i], np.ndarray), (rnd, i, got[i])
            assert np.array_equal(got[i].view(np.uint32), serial[i].view(np.uint32)), (rnd, i)

    # one scene, one G-buffer, two passes of the same batch from two threads on two streams
    def shared(concurrent):
        r = api.Renderer(cornell_emissive, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
        r.enable_direct(wire.default_params_di())
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        errs = []
        for cb in frames_of(cornell_emissive, {}):
            r.p_gbuffer.render(cb, r.scene, r.gbuffer)
            r.p_prelight.render(cb, r.scene, None)
            torch.cuda.synchronize()

            def go(p, st):
                try:
                    p.render(cb, r.scene, r.gbuffer, st.cuda_stream)
                except Exception as e:
                    errs.append(e)
            if concurrent:
                ta, tb = threading.Thread(target=go, args=(r.p_direct, sa)), threading.Thread(target=go, args=(r.p_indirect, sb))
                ta.start(); tb.start(); ta.join(60); tb.join(60)
            else:
                go(r.p_direct, sa); go(r.p_indirect, sb)
            torch.cuda.synchronize()
        assert not errs, errs
        return r.p_direct.download().copy(), r.final().copy()
    d0, i0 = shared(False)
    for rnd in range(2):
        d1, i1 = shared(True)
        assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(i0.view(np.uint32), i1.view(np.uint32)), rnd
