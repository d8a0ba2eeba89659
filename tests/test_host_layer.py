"""C++ RenderPass / RenderGraph mirror (zetaray_amd/host): graph ordering on the CPU, a full frame on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from zetaray_amd import api
    api.lib()                                                    # torch + libzetaray_amd first (one HIP runtime)
    L = C.CDLL(os.path.join(ROOT, "zetaray_amd", "libzetaray_host.so"))
    L.zrh_graph_selftest.argtypes = [C.c_char_p, C.c_int]
    L.zrh_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    return L


def test_render_graph_orders_the_reference_node_set():
    """The reference's hot-path nodes with its resource dependencies (PathTracer.cpp:325-563): RT_AS_Build -> GBuffer,
    PreLighting -> EmissiveAliasTable -> {DirectLighting || Indirect} -> Compositing; every delegate runs once and
    never before its producers."""
    buf = C.create_string_buffer(1024)
    n = _lib().zrh_graph_selftest(buf, 1024)
    batches, log = buf.value.decode().split("#")
    assert n == 8
    b = [set(x.split(",")) for x in batches.split("|")]
    assert b[0] == {"RT_AS_Build", "Sky", "PreLighting"}
    assert b[1] == {"GBuffer", "EmissiveAliasTable"}
    assert b[2] == {"DirectLighting", "Indirect"}            # same batch: may record concurrently
    assert b[3] == {"Compositing"}
    order = log.split(",")
    pos = {name: i for i, name in enumerate(order)}
    assert len(pos) == 8
    for before, after in [("RT_AS_Build", "GBuffer"), ("PreLighting", "EmissiveAliasTable"), ("GBuffer", "Indirect"),
                          ("EmissiveAliasTable", "DirectLighting"), ("Sky", "Indirect"), ("Indirect", "Compositing"),
                          ("DirectLighting", "Compositing")]:
        assert pos[before] < pos[after]


@pytest.mark.gpu
def test_cpp_passes_render_a_frame(cornell_emissive, oracle_emissive):
    w, h = 80, 48
    cb = scene_io.make_frame_constants(w, h, frame_num=2, num_emissives=len(cornell_emissive.emissives))
    desc = cornell_emissive.desc()
    out = np.zeros((h, w, 4), np.float32)
    cbb = np.ascontiguousarray(cb)
    rc = _lib().zrh_render_frame(C.addressof(desc), cbb.ctypes.data, w, h, out.ctypes.data)
    assert rc == 0
    _, planes = oracle_emissive.gbuffer(cb)
    want, _ = oracle_emissive.pathtrace(cb, planes, wire.default_params())
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_render_restir_pt_sequence(cornell_emissive, oracle_emissive):
    """IndirectLighting::Init(INTEGRATOR::ReSTIR_PT) driven by the C++ RenderGraph for 3 frames == oracle frame 3."""
    from oracle import zro
    w, h, n = 80, 48, 3
    cbs = np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives)) for f in range(1, n + 1)])
    cbs = np.ascontiguousarray(cbs)
    desc = cornell_emissive.desc()
    out = np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    assert L.zrh_render_sequence(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, out.ctypes.data) == 0
    o = zro.OracleRPT(oracle_emissive, w, h)
    prm = wire.default_params()
    for f in range(n):
        want = o.render(cbs[f], prm)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_render_gi_and_di_sequence(cornell_emissive, oracle_emissive):
    """C++ RenderGraph with GBuffer -> PreLighting -> {DirectLighting (ReSTIR DI), Indirect (ReSTIR GI)} for 3 frames."""
    from oracle import zro
    w, h, n = 80, 48, 3
    cbs = np.ascontiguousarray(np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives)) for f in range(1, n + 1)]))
    desc = cornell_emissive.desc()
    out, dout = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    assert L.zrh_render_sequence2(C.addressof(desc), cbs.ctypes.data, n, w, h, 1, out.ctypes.data, dout.ctypes.data) == 0
    ogi, odi = zro.OracleRGI(oracle_emissive, w, h), zro.OracleRDI(oracle_emissive, w, h)
    for f in range(n):
        want = ogi.render(cbs[f], wire.default_params())
        dwant = odi.render(cbs[f], wire.default_params_di())
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dout.view(np.uint32), dwant.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_render_with_light_presampling():
    """PreLighting::SetLightPresamplingParams through the C++ mirror: K3 regenerates the presampled sets every frame (seeded by
    FrameNum) and Indirect (ReSTIR PT) + DirectLighting read them; 3 frames scheduled by the RenderGraph == the oracle's frame 3."""
    from oracle import zro
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    w, h, n = 96, 64, 3
    cbs = np.ascontiguousarray(np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
                                         for f in range(1, n + 1)]))
    desc = sc.desc()
    out, dout = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence3.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    assert L.zrh_render_sequence3(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, out.ctypes.data, dout.ctypes.data, 32, 128) == 0
    ip, dp = wire.default_params(), wire.default_params_di()
    for q in (ip, dp):
        q.presampling, q.num_sample_sets, q.sample_set_size = 1, 32, 128
    opt, odi = zro.OracleRPT(o, w, h), zro.OracleRDI(o, w, h)
    for f in range(n):
        o.presample(f + 1, 32, 128)
        want = opt.render(cbs[f], ip)
        dwant = odi.render(cbs[f], dp)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dout.view(np.uint32), dwant.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_render_sun_sky_sequence():
    """The reference's default frame through the C++ mirror: Sky (K17) -> GBuffer -> {SkyDI (K7/K8), Indirect (ReSTIR PT, sun + sky NEE)}
    scheduled by the RenderGraph for 3 frames == the oracle's frame 3 of both."""
    import os
    from oracle import zro
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell.npz"))
    osc = zro.OracleScene(sc)
    w, h, n = 80, 48, 3
    cbs = np.ascontiguousarray(np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0) for f in range(1, n + 1)]))
    desc = sc.desc()
    out, dout = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence_sky.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    assert L.zrh_render_sequence_sky(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, out.ctypes.data, dout.ctypes.data) == 0
    opt, osd = zro.OracleRPT(osc, w, h), zro.OracleSDI(osc, w, h)
    for f in range(n):
        osc.sky_lut(cbs[f], 256, 128)
        want = opt.render(cbs[f], wire.default_params())
        dwant = osd.render(cbs[f], wire.default_params_sky_di())
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dout.view(np.uint32), dwant.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_render_post_chain():
    """... followed by the post chain through the C++ mirror: Compositing (sky DI + indirect) -> TAA, scheduled by the RenderGraph over 4
    frames with a jittered, moving camera: the composited image and the RGBA16F TAA output == the oracle's."""
    import os
    from oracle import zro
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell.npz"))
    osc = zro.OracleScene(sc)
    w, h, n = 80, 48, 4
    cbl, prev = [], None
    for f in range(1, n + 1):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.04 * max(0, f - 2), 1.2, -4.043),
                                           jitter=(0.2 * (f % 3 - 1), 0.2 * (f % 2 - 0.5)))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        cbl.append(cb)
    cbs = np.ascontiguousarray(np.stack(cbl))
    desc = sc.desc()
    out, dout, comp = (np.zeros((h, w, 4), np.float32) for _ in range(3))
    taa = np.zeros((h, w, 4), np.uint16)
    L = _lib()
    L.zrh_render_sequence_sky_post.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 4
    assert L.zrh_render_sequence_sky_post(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, out.ctypes.data, dout.ctypes.data, comp.ctypes.data, taa.ctypes.data) == 0
    opt, osd = zro.OracleRPT(osc, w, h), zro.OracleSDI(osc, w, h)
    hist = np.zeros((h, w, 4), np.uint16)
    for f in range(n):
        osc.sky_lut(cbs[f], 256, 128)
        ind = opt.render(cbs[f], wire.default_params())
        sdi = osd.render(cbs[f], wire.default_params_sky_di())
        planes, _keep = osc.gbuffer(cbs[f])
        # Compositing.hlsl:30-125 without accumulation: sky DI + indirect * !emissive; pixels without geometry show the sky / sun disk
        signal = zro.composite(osc, cbs[f], planes[2].reshape(h, w), sky_di=sdi, indirect=ind)
        hist = zro.taa(signal, planes[7].reshape(h, w), planes[3].reshape(h, w), hist, 0.1, f > 0)
    assert np.array_equal(comp.view(np.uint32)[..., :3], signal.view(np.uint32)[..., :3])
    assert np.array_equal(taa[..., :3], hist[..., :3])


@pytest.mark.gpu
def test_cpp_passes_render_post_chain_to_display():
    """... and on to the end of the frame: Compositing -> TAA -> AutoExposure -> Display (Tony McMapface) through the C++ mirror, 4 frames,
    moving jittered camera.  The exposure texel, the display image and its sRGB8 back buffer == the oracle's chain, bit for bit."""
    import os
    from oracle import zro
    from zetaray_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell.npz"))
    osc = zro.OracleScene(sc)
    w, h, n = 80, 48, 4
    cbl, prev = [], None
    for f in range(1, n + 1):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.04 * max(0, f - 2), 1.2, -4.043),
                                           jitter=(0.2 * (f % 3 - 1), 0.2 * (f % 2 - 0.5)))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        cbl.append(cb)
    cbs = np.ascontiguousarray(np.stack(cbl))
    desc = sc.desc()
    out, dout, comp, disp = (np.zeros((h, w, 4), np.float32) for _ in range(4))
    taa = np.zeros((h, w, 4), np.uint16)
    srgb = np.zeros((h, w, 4), np.uint8)
    exposure = np.zeros(2, np.float32)
    lut = api.load_tonemap_lut()
    L = _lib()
    L.zrh_render_sequence_sky_display.argtypes = ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 4 +
                                                  [C.c_void_p, C.c_uint32, C.c_int] + [C.c_void_p] * 3)
    assert L.zrh_render_sequence_sky_display(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, out.ctypes.data, dout.ctypes.data, comp.ctypes.data, taa.ctypes.data,
                                             lut.ctypes.data, 48, wire.TONEMAP_NEUTRAL, exposure.ctypes.data, disp.ctypes.data, srgb.ctypes.data) == 0
    opt, osd = zro.OracleRPT(osc, w, h), zro.OracleSDI(osc, w, h)
    hist = np.zeros((h, w, 4), np.uint16)
    prm = wire.default_params()
    e = np.zeros(2, np.float32)
    for f in range(n):
        osc.sky_lut(cbs[f], 256, 128)
        ind = opt.render(cbs[f], wire.default_params())
        sdi = osd.render(cbs[f], wire.default_params_sky_di())
        planes, _keep = osc.gbuffer(cbs[f])
        signal = zro.composite(osc, cbs[f], planes[2].reshape(h, w), sky_di=sdi, indirect=ind)
        hist = zro.taa(signal, planes[7].reshape(h, w), planes[3].reshape(h, w), hist, 0.1, f > 0)
        _, e = zro.auto_exposure(hist, prm, cbs[f]["dt"], e)
    want, want_srgb = zro.display(hist, prm, (w, h), e, lut)
    assert np.array_equal(taa[..., :3], hist[..., :3])
    assert np.array_equal(exposure.view(np.uint32), e.view(np.uint32)) and exposure[0] > 0
    assert np.array_equal(disp.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(srgb, want_srgb)


@pytest.mark.gpu
def test_cpp_denoise_node_in_the_render_graph(cornell_emissive, oracle_emissive):
    """The C++ mirror's Denoise node (no reference counterpart; RenderPass-shaped like the others) scheduled after IndirectLighting by the
    RenderGraph, 4 frames of ReSTIR PT: its output == the oracle's denoise pass run on the oracle's ReSTIR PT frames and G-buffers."""
    from oracle import zro
    w, h, n, iters = 96, 64, 4, 3
    cbs = np.ascontiguousarray(np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives)) for f in range(1, n + 1)]))
    desc = cornell_emissive.desc()
    final, den = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    assert L.zrh_render_sequence_denoise(C.addressof(desc), cbs.ctypes.data, n, w, h, iters, final.ctypes.data, den.ctypes.data) == 0
    o = zro.OracleRPT(oracle_emissive, w, h)
    hc, hm = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 2), np.float32)
    prev = None
    for f in range(n):
        sig = o.render(cbs[f], wire.default_params()).copy()
        planes, _keep = oracle_emissive.gbuffer(cbs[f])
        depth, normal, motion = planes[7].reshape(h, w).copy(), planes[1].reshape(h, w).copy(), planes[3].reshape(h, w).copy()
        pd, pn = (depth, normal) if prev is None else prev
        want, hc, hm = zro.svgf(sig, depth, normal, motion, pd, pn, hc, hm, temporal_valid=f > 0, iterations=iters)
        prev = (depth, normal)
    assert np.array_equal(final.view(np.uint32), sig.view(np.uint32))
    assert np.array_equal(den.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cpp_passes_take_the_reference_ui_parameters(cornell_emissive, oracle_emissive):
    """The knobs the reference's settings UI turns (IndirectLighting.cpp:1468-1600 and DirectLighting.cpp:374-410 *Callback members) as setters of the
    C++ mirror: a ReSTIR PT + ReSTIR DI run with nearly every one of them off its default -- 2 / 3 bounces, no Russian roulette, TWO spatial rounds,
    M_max 6 / 5, temporal sort off, boiling suppression off, path regularisation on, Alpha_min 0.2 (stored squared); DI with M_max 12, no extra
    disocclusion samples, deterministic spatial, Alpha_min 0.1 -- equals the oracle run with the same parameter block, bit for bit, after 4 frames
    with a moving camera."""
    from oracle import zro
    w, h, n = 80, 48, 4
    cbs = np.ascontiguousarray(np.stack([scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives), cam_pos=(0.02 * f, 1.2, -4.043))
                                         for f in range(1, n + 1)]))
    for f in range(1, n):
        cbs[f]["prev_view"], cbs[f]["prev_view_inv"], cbs[f]["prev_camera_jitter"] = cbs[f - 1]["curr_view"], cbs[f - 1]["curr_view_inv"], cbs[f - 1]["curr_camera_jitter"]

    class Tuning(C.Structure):
        _fields_ = [(k, C.c_int) for k in ("max_non_tr", "max_glossy_tr", "stochastic_multibounce", "russian_roulette", "temporal", "spatial_passes", "m_max_t", "m_max_s",
                                           "sort_temporal", "sort_spatial", "boiling_suppression", "path_regularization")] + [("alpha_min", C.c_float)] + \
                   [(k, C.c_int) for k in ("di_temporal", "di_spatial", "di_m_max", "di_extra_disocclusion", "di_stochastic_spatial")] + [("di_alpha_min", C.c_float)]
    t = Tuning(2, 3, -1, 0, 1, 2, 6, 5, 0, 1, 0, 1, 0.2, 1, 1, 12, 0, 0, 0.1)
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces, prm.m_max_temporal, prm.m_max_spatial, prm.num_spatial_passes = 2, 3, 6, 5, 2
    prm.flags = wire.IND_TEMPORAL_RESAMPLE | wire.IND_SPATIAL_RESAMPLE | wire.IND_PATH_REGULARIZATION | wire.IND_SORT_SPATIAL
    prm.alpha_min = float(np.float32(0.2) * np.float32(0.2))
    dip = wire.default_params_di()
    dip.flags = wire.IND_TEMPORAL_RESAMPLE | wire.IND_SPATIAL_RESAMPLE
    dip.m_max_temporal = 12
    dip.alpha_min = float(np.float32(0.1) * np.float32(0.1))
    desc = cornell_emissive.desc()
    out, dout = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    L = _lib()
    L.zrh_render_sequence_tuned.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.zrh_render_sequence_tuned(C.addressof(desc), cbs.ctypes.data, n, w, h, 2, C.byref(t), out.ctypes.data, dout.ctypes.data) == 0
    o, odi = zro.OracleRPT(oracle_emissive, w, h), zro.OracleRDI(oracle_emissive, w, h)
    for f in range(n):
        want = o.render(cbs[f], prm)
        dwant = odi.render(cbs[f], dip)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dout.view(np.uint32), dwant.view(np.uint32))
    # ... and the defaults are not what was just rendered (the knobs did something)
    o2 = zro.OracleRPT(oracle_emissive, w, h)
    for f in range(n):
        plain = o2.render(cbs[f], wire.default_params())
    assert not np.array_equal(plain.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cpp_indirect_pass_as_two_graph_nodes_overlaps_frames(cornell_emissive, oracle_emissive):
    """IndirectLighting::SetFrameOverlap + RenderCandidates / RenderReuse: the ReSTIR PT pass as two nodes of the C++ RenderGraph -- GBuffer, PreLighting and
    Indirect.Candidates on the async-compute queue, Indirect.Reuse on the direct queue -- with the frames of the sequence submitted back to back (one wait at
    the end).  5 frames, camera moving from frame 3: FINAL of the last frame == the oracle (carry and product mode), == the single-node graph; the graph's
    batches show the two halves; the same at 1280 x 720 (where the halves really run side by side) against the single-node graph, with light presampling on
    a materials scene."""
    from oracle import zro
    L = _lib()
    L.zrh_render_sequence_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]

    def cbs_of(sc, w, h, n, cam0):
        out, prev = [], None
        for f in range(1, n + 1):
            cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(cam0[0] + 0.04 * max(0, f - 2), cam0[1], cam0[2]))
            if prev is not None:
                cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
            prev = cb.copy()
            out.append(cb)
        return np.ascontiguousarray(np.stack(out))

    def run(sc, cbs, w, h, mode, sets=0, size=0):
        desc = sc.desc()
        out = np.zeros((h, w, 4), np.float32)
        buf = C.create_string_buffer(512)
        assert L.zrh_render_sequence_overlap(C.addressof(desc), cbs.ctypes.data, len(cbs), w, h, mode, out.ctypes.data, sets, size, buf, 512) == 0
        return out, buf.value.decode()

    w, h, n = 96, 64, 5
    cbs = cbs_of(cornell_emissive, w, h, n, (0.0, 1.2, -4.043))
    o = zro.OracleRPT(oracle_emissive, w, h)
    prm = wire.default_params()
    for f in range(n):
        want = o.render(cbs[f], prm)
    for mode in (0, 1, 2):
        got, batches = run(cornell_emissive, cbs, w, h, mode)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"overlap mode {mode}"
        b = [set(x.split(",")) for x in batches.split("|")]
        if mode:
            assert b == [{"GBuffer", "PreLighting"}, {"Indirect.Candidates"}, {"Indirect.Reuse"}], batches
        else:
            assert b == [{"GBuffer", "PreLighting"}, {"Indirect"}], batches
    # at a size where each half fills the device, and with PreLighting's K3 regenerating the light sets every frame on the first half's queue
    w, h, n = 1280, 720, 6
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    for scene, cam0, sets, size in ((cornell_emissive, (0.0, 1.2, -4.043), 0, 0), (sc, (0.0, 0.0, -3.5), 16, 64)):
        cbs = cbs_of(scene, w, h, n, cam0)
        plain, _ = run(scene, cbs, w, h, 0, sets, size)
        for mode in (1, 2):
            got, _ = run(scene, cbs, w, h, mode, sets, size)
            assert np.array_equal(got.view(np.uint32), plain.view(np.uint32)), (mode, sets)
        assert plain[..., :3].max() > 0
