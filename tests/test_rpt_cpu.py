"""ReSTIR PT (K11-K16) on CPU: the HIP stage functions (zr_rpt.h, run serially by tests/hostexec) against the oracle
(oracle/zro_rpt.h) -- bit-exact radiance AND bit-exact persistent state (all 7 reservoir planes) over several frames --
plus properties that pin the oracle itself: the no-reuse estimator agrees with the K9 path tracer in expectation, and
shifting a reservoir's path onto its own pixel reproduces its target with Jacobian 1."""
import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire

PLANES = ("A", "B", "C", "D", "E", "F", "G", "neighbor")


def _cb(sc, w, h, f, **kw):
    return scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), **kw)


def _assert_same_state(o, h, frame):
    for nm in PLANES:
        pa, pb = o.plane(nm), h.plane(nm)
        if nm == "A":      # RGBA8: the w channel is never written
            pa, pb = pa & 0xffffff, pb & 0xffffff
        assert np.array_equal(pa.view(np.uint8), pb.view(np.uint8)), f"frame {frame}: reservoir plane {nm} differs"


@pytest.fixture(scope="module")
def hx_emissive(cornell_emissive, oracle_emissive):
    return zhx.HostExecScene(cornell_emissive, oracle_emissive.alias)


@pytest.mark.parametrize("mode", ["full", "temporal", "none", "no_boiling"])
def test_rpt_cornell_bit_exact(cornell_emissive, oracle_emissive, hx_emissive, mode):
    w, h = 64, 48
    prm = wire.default_params()
    if mode == "temporal":
        prm.flags &= ~wire.IND_SPATIAL_RESAMPLE
    if mode == "none":
        prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
    if mode == "no_boiling":
        prm.flags &= ~wire.IND_BOILING_SUPPRESSION
    o, x = zro.OracleRPT(oracle_emissive, w, h), zhx.HostExecRPT(hx_emissive, w, h)
    for f in range(1, 5):
        cb = _cb(cornell_emissive, w, h, f)
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert not np.isnan(a).any()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}: radiance differs"
        _assert_same_state(o, x, f)
        assert o.counters == x.counters and o.counters[0] > 0
    assert a[..., :3].max() > 0


def test_rpt_moving_camera_and_reset(cornell_emissive, oracle_emissive, hx_emissive):
    """Camera translation between frames (motion vectors -> prevPixel != pixel, disocclusion at the borders), camera
    jitter, then ResetTemporal."""
    w, h = 72, 40
    prm = wire.default_params()
    o, x = zro.OracleRPT(oracle_emissive, w, h), zhx.HostExecRPT(hx_emissive, w, h)
    prev_cb = None
    for f in range(1, 6):
        cam = (0.05 * f, 1.2 + 0.02 * f, -4.043 + 0.03 * f)
        cb = _cb(cornell_emissive, w, h, f, cam_pos=cam, jitter=(0.25 * (f % 2), -0.125))
        if prev_cb is not None:
            for k in ("prev_view", "prev_view_inv", "prev_camera_jitter"):
                cb[k] = prev_cb[k.replace("prev_", "curr_")] if k != "prev_camera_jitter" else prev_cb["curr_camera_jitter"]
        prev_cb = cb.copy()
        if f == 4:
            o.reset_temporal(); x.reset_temporal()
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _assert_same_state(o, x, f)
    mv = o.prev[0][wire.GB_PLANE_NAMES.index("motion_vector")] if hasattr(wire, "GB_PLANE_NAMES") else None
    assert mv is None or np.any(mv != 0)


@pytest.fixture(scope="module")
def synthetic_small():
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    return sc, o, zhx.HostExecScene(sc, o.alias)


@pytest.mark.parametrize("nb,gb", [(3, 4), (6, 8)])
def test_rpt_all_material_classes(synthetic_small, nb, gb):
    """Metal, coat, specular / rough glass (GLOSSY_T reconnection limits, k > 2 replays, case 3 light reconnections),
    thin-walled transmission; (6, 8) bounces make Russian roulette and the 16x4-wave max trigger."""
    sc, osc, hx = synthetic_small
    w, h = 64, 48
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gb
    o, x = zro.OracleRPT(osc, w, h), zhx.HostExecRPT(hx, w, h)
    ks = set()
    for f in range(1, 5):
        cb = _cb(sc, w, h, f, cam_pos=(0, 0, -3.5))
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _assert_same_state(o, x, f)
        A = o.plane("A")[..., 0]
        k = A & 0xf
        lt_k = (A >> 14) & 3
        lt_k1 = (A >> 16) & 3
        for kk, c3, c2 in zip(k.ravel(), lt_k.ravel(), lt_k1.ravel()):
            if kk != 15:
                ks.add((int(kk) + 2, 3 if c3 else (2 if c2 else 1)))
    # the scene must exercise reconnections beyond the first indirect vertex and all three cases
    assert any(k > 2 for k, _ in ks) and {c for _, c in ks} == {1, 2, 3}, ks


@pytest.mark.parametrize("nb,gb", [(3, 4), (6, 8)])
def test_k11_carried_state_is_all_a_path_needs(synthetic_small, cornell_emissive, oracle_emissive, hx_emissive, nb, gb):
    """K11 with per-bounce path compaction (zr_kernels.h k_rpt_pt_first / k_rpt_pt_next, ZR_K11=compact) moves a live path between kernels as the
    78 words rpt::PtCarry enumerates.  Here the host executor rebuilds every live path from exactly those words at every bounce boundary -- the rest
    of the lane state is poison -- on the material scene (metal, coat, glass, thin walls, Russian roulette with (6, 8) bounces) and on the Cornell
    box: radiance and all reservoir planes still equal the oracle's."""
    sc, osc, hx = synthetic_small
    w, h = 64, 48
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gb
    zhx.set_k11_carry(True)
    try:
        for scene, oscene, hxs, cam in ((sc, osc, hx, dict(cam_pos=(0, 0, -3.5))), (cornell_emissive, oracle_emissive, hx_emissive, {})):
            o, x = zro.OracleRPT(oscene, w, h), zhx.HostExecRPT(hxs, w, h)
            for f in range(1, 4):
                cb = _cb(scene, w, h, f, **cam)
                a, b = o.render(cb, prm), x.render(cb, prm)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
                _assert_same_state(o, x, f)
    finally:
        zhx.set_k11_carry(False)


def test_k11_fused_stage_functions_equal_the_oracle(synthetic_small, cornell_emissive, oracle_emissive, hx_emissive):
    """The inline megakernel k_rpt_pathtrace compiles the FUSED forms of the K11 stage functions (PtInitLane_Fused / PtPhaseA_Fused, traversal in the
    middle of one function); the host executor normally runs the cut forms.  Here it runs the fused ones: radiance and all reservoir planes equal the
    oracle's on the material scene (metal, coat, glass, thin walls, Russian roulette) and on the Cornell box -- so an edit that reaches only one of
    the two forms (round 5: the wo-only BSDF terms prepared once per surface, the light direction's evaluation reused by the sampler pdf) shows here.
    (The sun + sky variant: test_rpt_sun_sky_bit_exact[fused].)"""
    sc, osc, hx = synthetic_small
    w, h = 64, 48
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 6, 8
    zhx.set_k11_fused(True)
    try:
        for scene, oscene, hxs, cam in ((sc, osc, hx, dict(cam_pos=(0, 0, -3.5))), (cornell_emissive, oracle_emissive, hx_emissive, {})):
            o, x = zro.OracleRPT(oscene, w, h), zhx.HostExecRPT(hxs, w, h)
            for f in range(1, 4):
                cb = _cb(scene, w, h, f, **cam)
                a, b = o.render(cb, prm), x.render(cb, prm)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
                _assert_same_state(o, x, f)
    finally:
        zhx.set_k11_fused(False)


def test_plain_material_class_changes_nothing_on_plain_scenes(cornell_emissive, oracle_emissive, hx_emissive):
    """The PLAIN kernel permutations (zr_kernels.h) set SceneView / GBuf / RBuf::plain = 1, which folds the metal / transmission / thin-wall / coat code out of
    InitSurface, LoadPixelSurfaceEx and LoadOffsetCtx.  On the device that path is only compared GPU == oracle; here the host executor runs the same stage
    functions with the class set (cut and fused forms of K11) on the Cornell box, emissive and sun + sky -- scenes of the plain class: radiance and every reservoir
    plane equal the oracle's, i.e. the general path's."""
    from zetaray_amd import scene_io
    import os
    w, h = 64, 48
    prm = wire.default_params()
    sky = scene_io.load_npz(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell.npz"))
    osky = zro.OracleScene(sky)
    hsky = zhx.HostExecScene(sky)
    zhx.set_material_class(True)
    try:
        for fused in (False, True):
            zhx.set_k11_fused(fused)
            for scene, oscene, hxs in ((cornell_emissive, oracle_emissive, hx_emissive), (sky, osky, hsky)):
                o, x = zro.OracleRPT(oscene, w, h), zhx.HostExecRPT(hxs, w, h)
                for f in range(1, 4):
                    cb = _cb(scene, w, h, f, cam_pos=(0.04 * max(0, f - 2), 1.2, -4.043))
                    if len(scene.emissives) == 0:
                        oscene.sky_lut(cb, 256, 128); hxs.sky_lut(cb, 256, 128)
                    a, b = o.render(cb, prm), x.render(cb, prm)
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f} (fused {fused})"
                    _assert_same_state(o, x, f)
    finally:
        zhx.set_material_class(False); zhx.set_k11_fused(False)


def test_k11_with_the_selected_reconnection_parked(synthetic_small, cornell_emissive, oracle_emissive, hx_emissive):
    """k_rpt_pathtrace_park (ZR_K11_PARK=1; zr_rpt.h RcPark): while a path is traced the reservoir's selected reconnection lives in a [word][lane]
    park (LDS on the device) -- Reservoir::Update stores winners there, the epilogue reads the last one back.  The host executor runs K11 that way,
    the lane's own copy never written: radiance and all reservoir planes still equal the oracle's, materials + Russian roulette and Cornell."""
    sc, osc, hx = synthetic_small
    w, h = 64, 48
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 6, 8
    zhx.set_k11_park(True)
    try:
        for scene, oscene, hxs, cam in ((sc, osc, hx, dict(cam_pos=(0, 0, -3.5))), (cornell_emissive, oracle_emissive, hx_emissive, {})):
            o, x = zro.OracleRPT(oscene, w, h), zhx.HostExecRPT(hxs, w, h)
            for f in range(1, 4):
                cb = _cb(scene, w, h, f, **cam)
                a, b = o.render(cb, prm), x.render(cb, prm)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
                _assert_same_state(o, x, f)
    finally:
        zhx.set_k11_park(False)


def test_rpt_initial_candidates_unbiased_vs_k9(cornell_emissive, oracle_emissive):
    """Without reuse K11 is a path tracer with a different RNG layout: its mean converges to K9's (pins the oracle's
    NEE / MIS / throughput bookkeeping against the independently pinned K9 restatement)."""
    w, h, n = 48, 32, 160
    prm = wire.default_params()
    p2 = wire.default_params()
    p2.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
    acc9, accp = np.zeros((h, w, 4), np.float64), np.zeros((h, w, 4), np.float64)
    rpt = zro.OracleRPT(oracle_emissive, w, h)
    for f in range(1, n + 1):
        cb = _cb(cornell_emissive, w, h, f)
        gb = oracle_emissive.gbuffer(cb)
        acc9 += oracle_emissive.pathtrace(cb, gb[1], prm)[0]
        accp += rpt.render(cb, p2, gb)
    m9, mp = acc9[..., :3].mean(axis=(0, 1)) / n, accp[..., :3].mean(axis=(0, 1)) / n
    assert np.all(np.abs(mp / m9 - 1) < 0.06), (m9, mp)


def test_rpt_self_shift_identity(synthetic_small):
    """Reconnection shift of a path onto its own pixel, k == 2 (no replay): target ~= stored target (L is fp16, w_k is
    oct32), Jacobian ~= 1, for case 1 and case 2.  (k > 2 goes through the r-buffer, whose context Load() resets the
    replay RNG -- Shift.hlsli:196-204, 700-713 -- so the reference's own shift is not an identity there.)"""
    sc, osc, _ = synthetic_small
    w, h = 64, 48
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 5, 6
    rpt = zro.OracleRPT(osc, w, h)
    cb = _cb(sc, w, h, 3, cam_pos=(0, 0, -3.5))
    rpt.render(cb, prm)
    o = rpt.self_shift(cb, prm).reshape(-1, 6)
    o = o[(o[:, 4] == 2) & (o[:, 1] > 0)]
    assert len(o) > 400 and set(o[:, 5].astype(int)) == {1, 2}
    ok = o[:, 0] > 0
    assert ok.mean() > 0.97
    tr, jr = o[ok, 0] / o[ok, 1], o[ok, 2] / o[ok, 3]
    assert np.median(np.abs(tr - 1)) < 2e-3 and np.quantile(np.abs(tr - 1), 0.85) < 0.03
    assert np.median(np.abs(jr - 1)) < 1e-3 and np.quantile(np.abs(jr - 1), 0.85) < 0.03


@pytest.mark.parametrize("carry", [False, True, "fused"], ids=["", "state carried through PtCarry at bounce boundaries", "fused"])
@pytest.mark.parametrize("kind", ["cornell", "glossy"])
def test_rpt_sun_sky_bit_exact(kind, carry):
    """NEE_EMISSIVE == 0 variants of K11-K16: no emissive triangles; NEE_NonEmissive (one RIS over sun / cosine-sky / BSDF-sky with
    Le_Sky as the lobe-RIS target), case-2 / case-3 reconnections to SUN / SKY, EstimateDirect_y_k_min_1 in the shifts, the
    partial (component-wise) reservoir plane writes of the non-emissive Write<>: radiance, all 7 planes, counters; moving camera."""
    import os
    from zetaray_amd import scene_io as sio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if kind == "cornell":
        sc, cam0, sun = sio.load_npz(os.path.join(root, "tests", "golden", "cornell.npz")), (0.0, 1.2, -4.043), None
        osc = zro.OracleScene(sc)
    else:
        sc, cam0, sun = sio.make_synthetic_scene(num_tris=1500, num_emissive=0, seed=5, open_top=True), (0.0, 2.0, -3.5), (0.3, -0.8, 0.4)
        osc = zro.OracleScene(sc, force_bvh=True)
    hx = zhx.HostExecScene(sc)
    w, h = 64, 48
    prm = wire.default_params()
    o, x = zro.OracleRPT(osc, w, h), zhx.HostExecRPT(hx, w, h)
    prev = None
    zhx.set_k11_fused(carry == "fused")      # the fused stage functions of the inline megakernel instead of the cut ones
    carry = carry is True
    zhx.set_k11_carry(carry)      # (the sun + sky variant traces its continuation ray at the top of PtPhaseA: the carried normal / transmissive flag)
    for f in range(1, 6):
        cb = sio.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(cam0[0] + 0.05 * max(0, f - 3), cam0[1], cam0[2]))
        if sun is not None:
            sd = np.array(sun, np.float32)
            cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        osc.sky_lut(cb, 256, 128)
        hx.sky_lut(cb, 256, 128)
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert not np.isnan(a).any()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}: radiance differs"
        for nm in "ABCDEFG":
            assert np.array_equal(o.plane(nm).view(np.uint8), x.plane(nm).view(np.uint8)), f"frame {f}: plane {nm} differs"
        assert o.counters == x.counters
    zhx.set_k11_carry(False); zhx.set_k11_fused(False)
    assert a[..., :3].max() > 0
    A = o.plane("A")[..., 0]
    assert (((A >> 16) & 3) != 0).any()          # some reservoirs reconnect into the sun / sky (case 2)


def test_frames_without_any_geometry(cornell_emissive, oracle_emissive, hx_emissive):
    """the camera looks straight up from outside the box: every primary ray misses.  ReSTIR PT (3 frames, so the temporal and spatial stages run over nothing but
    invalid pixels) == the oracle, nothing but the primary rays is traced and the radiance is zero."""
    w, h = 48, 32
    prm = wire.default_params()
    o, x = zro.OracleRPT(oracle_emissive, w, h), zhx.HostExecRPT(hx_emissive, w, h)
    for f in range(1, 4):
        cb = _cb(cornell_emissive, w, h, f, view_dir=(0, 1, 0), up=(0, 0, 1))      # straight up from outside the box
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _assert_same_state(o, x, f)
        assert not a[..., :3].any()


def _six_triangle_scene(cornell_emissive):
    """three quads of the Cornell box -- a wall, the light, another wall: 6 triangles, below BvhBuilder::kTinyScene, so the scene has NO nodes and every
    query runs against the single leaf that holds all triangles (zr_dev_scene.h kWholeSceneLeaf)"""
    import copy
    keep = [1, 6, 7]
    s2 = copy.copy(cornell_emissive)
    s2.instances = cornell_emissive.instances[keep].copy()
    s2.instance_to_world = cornell_emissive.instance_to_world[keep].copy()
    s2.instance_mask = cornell_emissive.instance_mask[keep].copy()
    s2.instance_num_tris = cornell_emissive.instance_num_tris[keep].copy()
    em = cornell_emissive.emissives.copy()
    for p in range(len(em)):
        em[p]["id"] = scene_io.pcg3d(keep.index(6), 0, p)[0]      # the light is instance 1 of this scene: its triangles' IDs hash the new index
    s2.emissives = em
    s2._desc = None
    return s2


def test_scene_without_bvh_nodes(cornell_emissive):
    """a 6-triangle scene (no BVH nodes: one leaf over everything): G-buffer and three ReSTIR PT frames, host-executed HIP stage functions == oracle"""
    sc = _six_triangle_scene(cornell_emissive)
    assert sc.num_tris == 6
    osc = zro.OracleScene(sc)
    hx = zhx.HostExecScene(sc, osc.alias)
    assert hx.bvh_digest()[1] == 0          # no nodes
    w, h = 64, 48
    prm = wire.default_params()
    o, x = zro.OracleRPT(osc, w, h), zhx.HostExecRPT(hx, w, h)
    for f in range(1, 4):
        cb = _cb(sc, w, h, f)
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _assert_same_state(o, x, f)
    assert (a[..., :3].sum(-1) > 0).sum() > 50


def test_window_parity_machinery_against_a_full_frame_host_executor(cornell_emissive, oracle_emissive):
    """CPU dry run of tests/window_parity.py (the at-size GPU tests of the 1080p / 2160p atrium): the full frame is played by a full-frame host
    executor instead of the GPU -- staged the same way, its rects cut out of its planes instead of zr_pass_halo_pack -- and the full-frame image it
    produces is the oracle's.  Windows in a corner, in the interior, and on the partial 32 x 32 tiles of the bottom edge; moving camera."""
    from oracle import zro
    from tests.hostexec import zhx
    from tests.window_parity import HALO_PLANES, windows_parity
    W, H = 160, 120
    prm = wire.default_params()

    class Full:
        def __init__(self):
            self.hx = zhx.HostExecScene(cornell_emissive, oracle_emissive.alias)
            self.r = zhx.HostExecRPT(self.hx, W, H)

        def stage1(self, cb):
            self.r.render_stage(cb, prm, 1)

        def stage2(self, cb):
            return self.r.render_stage(cb, prm, 2).copy(), None

        def rect_planes(self, which, rect):
            x, y, w, h = rect
            return {name: self.r.plane(name, which)[y:y + h, x:x + w].copy() for name, _, _, _ in HALO_PLANES}

    full = Full()
    cams = [(0.0, 1.2, -4.043), (0.0, 1.2, -4.043), (0.05, 1.2, -4.02), (0.1, 1.2, -4.0)]
    # oracle=...: the oracle renders the same windows (oracle/zro_rpt.h restricted to a rectangle of the full-size frame, zro.OracleRPTWindows) and
    # is compared with the full frame as well -- the machinery the GPU tests of the 1080p / 2160p atrium use against the GPU's frame
    rays = windows_parity(full, cornell_emissive, oracle_emissive.alias, W, H, [(0, 0, 64, 64), (64, 32, 64, 32), (96, 96, 64, 24)], prm, cams, oracle=oracle_emissive)
    assert rays > 0
    # ... and that full frame is the oracle's
    o = zro.OracleRPT(oracle_emissive, W, H)
    from zetaray_amd import scene_io
    prev = None
    for f, cam in enumerate(cams, 1):
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(cornell_emissive.emissives), cam_pos=cam)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        want = o.render(cb, prm)
    assert np.array_equal(full.r.final.view(np.uint32), want.view(np.uint32))


def test_moved_instances_in_subtrees_of_their_own():
    """zr_bvh.h Build(ownSubtree): what the product's background rebuild does once an instance has moved (the reference's static -> dynamic BLAS
    conversion, SceneCore.cpp:1038) -- its triangles leave the common SAH tree for a subtree of their own, joined near the root.  Host-executed HIP
    stage functions through such trees (one and two instances on their own, then every instance) == oracle over three frames of motion: hits do not
    depend on the tree.  Each of the three trees differs from the common one and from the others."""
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    osc = zro.OracleScene(sc, force_bvh=True)
    hx = zhx.HostExecScene(sc, osc.alias)
    common = hx.bvh_digest()
    w, h = 48, 32
    prm = wire.default_params()
    o, x = zro.OracleRPT(osc, w, h), zhx.HostExecRPT(hx, w, h)
    cand = sorted((i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE), key=lambda i: -int(sc.instance_num_tris[i]))
    t0, xf = {i: sc.instances["translation"][i].copy() for i in cand[:2]}, {}
    own_sets = {2: cand[:1], 3: cand[:2], 4: range(len(sc.instances))}
    prev, digests = None, set()
    for f in range(1, 5):
        if f >= 2:
            for k, i in enumerate(cand[:2]):
                ang = 0.1 * (f - 1) * (k + 1)
                scene_io.move_instance(sc, i, translation=t0[i] + np.float32([0.05 * (f - 1), 0.02 * k, -0.03 * (f - 1)]),
                                       rotation=np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32), xform_of=xf)
            own = np.zeros(len(sc.instances), np.uint8)
            own[list(own_sets[f])] = 1
            hx.set_own_subtree(own)
            hx.update_instances(sc.instances, sc.instance_to_world)
            osc.update_instances(sc.instances, sc.instance_to_world)
            d = hx.bvh_digest()
            assert d[2] == common[2] and d[0] != common[0]
            digests.add(d[0])
        cb = _cb(sc, w, h, f, cam_pos=(0, 0, -3.5))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _assert_same_state(o, x, f)
    assert len(digests) == 3


def test_picked_pixel_reports_the_mesh_under_it(cornell_emissive, oracle_emissive, hx_emissive):
    """GBufferRT::PickPixel (GBufferRT.h:36-46, GBufferRT_Inline.hlsl:241-242: g_pick[0] = hitMeshIdx, UINT32_MAX on a miss): the product's K1 stage
    function on the host == the oracle for a grid of pixels of the Cornell box (walls, boxes, the light, the open front's misses), on a screen tile
    too; the G-buffer itself is unchanged by a pending pick."""
    w, h = 96, 64
    cb = _cb(cornell_emissive, w, h, 3)
    seen = set()
    for y in range(2, h, 9):
        for x in range(1, w, 11):
            a, b = oracle_emissive.pick(cb, x, y), hx_emissive.pick(cb, x, y)
            assert a == b, (x, y, a, b)
            seen.add(a)
    assert len(seen - {0xffffffff}) >= 5 and max(seen - {0xffffffff}) < len(cornell_emissive.instances)      # several instances
    # a camera outside the box looking away: every primary ray misses
    cb_out = _cb(cornell_emissive, w, h, 3, cam_pos=(0.0, 1.0, -30.0), view_dir=(0, 0, -1))
    assert oracle_emissive.pick(cb_out, 5, 5) == hx_emissive.pick(cb_out, 5, 5) == 0xffffffff
    # on a tile: the pixel is named in render-target coordinates
    tile = (32, 32, 64, 32)
    assert hx_emissive.pick(cb, 40, 50, tile=tile) == oracle_emissive.pick(cb, 40, 50)
    pa, _ = hx_emissive.gbuffer(cb)
    pb, _ = oracle_emissive.gbuffer(cb)
    for n, a, b in zip(wire.GB_PLANE_NAMES, pa, pb):
        assert np.array_equal(np.asarray(a).view(np.uint8).reshape(-1), np.asarray(b).view(np.uint8).reshape(-1)), n
