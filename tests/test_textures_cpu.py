"""Material textures (`-m "not gpu"`): the ABI's filtering arithmetic (include/zr_texture.h) against an independent numpy
restatement, and the textured K1 G-buffer (UV differentials, base colour / normal / metallic-roughness / emissive maps,
alpha-tested primary rays), K2 emissive power and emissive-textured light sampling -- HIP stage functions run by the serial
host executor vs the oracle, bit-exact."""
import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire


@pytest.fixture(scope="module")
def textured():
    sc = scene_io.make_synthetic_scene(num_tris=1024, num_emissive=64, seed=11)
    offs = scene_io.add_test_textures(sc)
    return sc, offs


def _cb(offs, w=48, h=32, **kw):
    cb = scene_io.make_frame_constants(w, h, cam_pos=(0.3, 0.2, -3.6), **kw)
    return scene_io.set_texture_heap_offsets(cb, offs)


@pytest.fixture(scope="module")
def pair(textured):
    sc, offs = textured
    cb = _cb(offs)
    orc = zro.OracleScene(sc, force_bvh=True, cb=cb)
    hx = zhx.HostExecScene(sc, alias=orc.alias)
    hx.latch_heap_offsets(cb)
    return orc, hx


def _srgb(b):
    c = b / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def _mip(sc, t, m):
    d = sc.textures[t]
    w, h, off = int(d["width"]), int(d["height"]), int(d["offset"])
    ch = 2 if d["format"] == wire.TEX_RG8 else 4
    for _ in range(m):
        off += w * h * ch
        w, h = max(1, w // 2), max(1, h // 2)
    return sc.texels[off:off + w * h * ch].reshape(h, w, ch)


def _decode(sc, t, texels):
    d = sc.textures[t]
    a = texels.astype(np.float64)
    if d["format"] == wire.TEX_RG8:
        return np.concatenate([a / 255.0, np.zeros_like(a[..., :1]), np.ones_like(a[..., :1])], -1)
    rgb = _srgb(a[..., :3]) if d["format"] == wire.TEX_RGBA8_SRGB else a[..., :3] / 255.0
    return np.concatenate([rgb, a[..., 3:] / 255.0], -1)


def _bilinear_ref(sc, t, m, uv):
    img = _decode(sc, t, _mip(sc, t, m))
    h, w = img.shape[:2]
    x = (uv[:, 0] - np.floor(uv[:, 0])) * w - 0.5
    y = (uv[:, 1] - np.floor(uv[:, 1])) * h - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    tx, ty = (x - x0)[:, None], (y - y0)[:, None]
    c = lambda xx, yy: img[yy % h, xx % w]
    top = c(x0, y0) * (1 - tx) + c(x0 + 1, y0) * tx
    bot = c(x0, y0 + 1) * (1 - tx) + c(x0 + 1, y0 + 1) * tx
    return top * (1 - ty) + bot * ty


def test_mip_chain_layout(textured):
    sc, _ = textured
    assert sc.textures["num_mips"].tolist() == [7, 6, 6, 5, 4]           # 64x32, 32x32, 32x32, 24x12, 8x8
    assert (sc.textures["offset"] % 4 == 0).all()
    m1 = _mip(sc, 0, 1).astype(int)
    m0 = _mip(sc, 0, 0).astype(int)
    box = (m0[0::2, 0::2] + m0[1::2, 0::2] + m0[0::2, 1::2] + m0[1::2, 1::2] + 2) // 4
    assert np.array_equal(m1, box)


@pytest.mark.parametrize("tex", [0, 1, 2, 3, 4])
def test_filtering_matches_numpy(pair, textured, tex):
    orc, hx = pair
    sc, _ = textured
    rng = np.random.default_rng(3 + tex)
    uv = rng.uniform(-3, 3, (400, 2)).astype(np.float32)
    # point: the texel under uv
    img = _decode(sc, tex, _mip(sc, tex, 0))
    h, w = img.shape[:2]
    fx, fy = uv[:, 0] - np.floor(uv[:, 0]), uv[:, 1] - np.floor(uv[:, 1])
    ref = img[np.minimum((fy * h).astype(int), h - 1), np.minimum((fx * w).astype(int), w - 1)]
    got = orc.tex_sample(tex, 0, uv)
    assert np.allclose(got, ref, atol=1e-6)
    assert np.array_equal(got, hx.tex_sample(tex, 0, uv))
    # bilinear at integer lods, trilinear in between
    nm = int(sc.textures["num_mips"][tex])
    for lod in (0.0, 1.0, float(nm - 1)):
        g = np.zeros((len(uv), 4), np.float32); g[:, 0] = lod
        got = orc.tex_sample(tex, 1, uv, g)
        assert np.allclose(got, _bilinear_ref(sc, tex, int(lod), uv), atol=2e-6)
        assert np.array_equal(got, hx.tex_sample(tex, 1, uv, g))
    g = np.zeros((len(uv), 4), np.float32); g[:, 0] = 0.25
    got = orc.tex_sample(tex, 1, uv, g)
    ref = 0.75 * _bilinear_ref(sc, tex, 0, uv) + 0.25 * _bilinear_ref(sc, tex, 1, uv)
    assert np.allclose(got, ref, atol=2e-6)
    # lod beyond the chain clamps to the last mip; NaN lod -> 0
    g[:, 0] = 40.0
    assert np.allclose(orc.tex_sample(tex, 1, uv, g), _bilinear_ref(sc, tex, nm - 1, uv), atol=2e-6)
    g[:, 0] = np.nan
    assert np.allclose(orc.tex_sample(tex, 1, uv, g), _bilinear_ref(sc, tex, 0, uv), atol=2e-6)


def test_sample_grad_properties(pair, textured):
    orc, hx = pair
    sc, _ = textured
    rng = np.random.default_rng(9)
    uv = rng.uniform(-2, 2, (300, 2)).astype(np.float32)
    w, h = float(sc.textures["width"][0]), float(sc.textures["height"][0])
    # isotropic footprint of 2 texels -> N = 1, lod = 1: equals SampleLevel(1)
    g = np.zeros((300, 4), np.float32); g[:, 0] = 2.0 / w; g[:, 3] = 2.0 / h
    lv = np.zeros((300, 4), np.float32); lv[:, 0] = 1.0
    assert np.allclose(orc.tex_sample(0, 2, uv, g), orc.tex_sample(0, 1, uv, lv), atol=1e-6)
    # 4:1 anisotropy: 4 taps at the minor axis' lod (= lod 0 here), spread along the major axis
    g = np.zeros((300, 4), np.float32); g[:, 0] = 4.0 / w; g[:, 3] = 1.0 / h
    taps = [_bilinear_ref(sc, 0, 0, uv + np.array([s * 4.0 / w, 0.0])) for s in (-0.375, -0.125, 0.125, 0.375)]
    assert np.allclose(orc.tex_sample(0, 2, uv, g), np.mean(taps, 0), atol=3e-6)
    # degenerate / hostile gradients are finite and identical on both sides
    for bad in (0.0, np.inf, np.nan, 65504.0, -1e30):
        g = np.full((300, 4), bad, np.float32)
        a, b = orc.tex_sample(0, 2, uv, g), hx.tex_sample(0, 2, uv, g)
        assert np.isfinite(a).all() and np.array_equal(a, b)
    g = rng.normal(0, 0.05, (300, 4)).astype(np.float32)
    assert np.array_equal(orc.tex_sample(1, 2, uv, g), hx.tex_sample(1, 2, uv, g))
    nonfinite = np.array([[np.nan, 0.5], [np.inf, -np.inf], [1e38, 3e38]], np.float32)
    assert np.isfinite(orc.tex_sample(0, 1, nonfinite)).all()


def test_gbuffer_textured_bit_exact(pair, textured):
    orc, hx = pair
    sc, offs = textured
    for dof in (False, True):
        cb = _cb(offs, num_emissives=len(sc.emissives))
        if dof:
            cb["dof"], cb["focus_depth"], cb["lens_radius"] = 1, 3.0, 0.05
            cb["camera_ray_uv_grads_scale"] = 0.75
        ao, _ = orc.gbuffer(cb)
        ah, _ = hx.gbuffer(cb)
        for k, (a, b) in enumerate(zip(ao, ah)):
            assert np.array_equal(a, b), f"plane {k}"
    # the maps are visible in the planes: the untextured scene differs in base colour, normal, metallic-roughness
    plain = scene_io.make_synthetic_scene(num_tris=1024, num_emissive=64, seed=11)
    ap, _ = zro.OracleScene(plain, force_bvh=True).gbuffer(_cb(offs))
    cbt = _cb(offs)
    at, _ = orc.gbuffer(cbt)
    for plane in (0, 1, 2, 4):       # base colour, normal, metallic-roughness, emissive colour
        assert not np.array_equal(ap[plane], at[plane]), plane
    # alpha test: some primary rays pass through the non-opaque instance -> depth differs from the opaque scene somewhere
    assert not np.array_equal(ap[7], at[7])


def test_emissive_power_textured(pair, textured):
    orc, hx = pair
    sc, _ = textured
    po, ph = orc.estimate_power(), hx.estimate_power()
    assert np.array_equal(po, ph)
    plain = scene_io.make_synthetic_scene(num_tris=1024, num_emissive=64, seed=11)
    pp = zro.OracleScene(plain, force_bvh=True).estimate_power()
    assert (po <= pp * 1.0001).all() and (po < pp).any() and (po > 0).all()      # the map only darkens (texels <= 1)


def _params(**kw):
    p = wire.default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_pathtracer_textured_bit_exact(pair, textured):
    """K9 with ray differentials driving the base-colour / metallic-roughness LODs at every bounce, emissive-textured lights."""
    orc, hx = pair
    sc, offs = textured
    cb = _cb(offs, w=32, h=24, num_emissives=len(sc.emissives))
    planes_a, planes = orc.gbuffer(cb)
    prm = _params(max_non_tr_bounces=4, max_glossy_tr_bounces=5)
    fo, _ = orc.pathtrace(cb, planes, prm)
    fh, _ = hx.pathtrace(cb, planes, prm)
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))
    assert np.isfinite(fo).all() and fo[..., :3].max() > 0
    # and the textures matter: the same G-buffer shaded without the heap differs
    plain = scene_io.make_synthetic_scene(num_tris=1024, num_emissive=64, seed=11)
    fp, _ = zro.OracleScene(plain, force_bvh=True).pathtrace(cb, planes, prm)
    assert not np.array_equal(fp, fo)


def _frames(sc, offs, w, h, n):
    prev = None
    for f in range(1, n + 1):
        cb = _cb(offs, w=w, h=h, frame_num=f, num_emissives=len(sc.emissives))
        cb["camera_pos"][0] += 0.04 * max(0, f - 2)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        yield f, cb


def test_restir_gi_textured_bit_exact(pair, textured):
    orc, hx = pair
    sc, offs = textured
    w, h = 40, 32
    prm = _params(max_non_tr_bounces=4, max_glossy_tr_bounces=5)
    o, x = zro.OracleRGI(orc, w, h), zhx.HostExecRGI(hx, w, h)
    for f, cb in _frames(sc, offs, w, h, 3):
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        for nm in ("A", "B", "C"):
            assert np.array_equal(o.plane(nm).view(np.uint8), x.plane(nm).view(np.uint8)), f"frame {f}: GI plane {nm}"
        assert o.counters == x.counters
    assert np.isfinite(a).all() and a[..., :3].max() > 0


RPT_PLANES = ["A", "B", "C", "D", "E", "F", "G"]


def _rpt_same(o, x, f):
    for nm in RPT_PLANES:
        pa, pb = o.plane(nm), x.plane(nm)
        if nm == "A":      # RGBA8: the w channel is never written
            pa, pb = pa & 0xffffff, pb & 0xffffff
        assert np.array_equal(pa.view(np.uint8), pb.view(np.uint8)), f"frame {f}: reservoir plane {nm}"
    assert o.counters == x.counters


@pytest.mark.parametrize("mode", ["full", "none"])
def test_restir_pt_textured_bit_exact(pair, textured, mode):
    """K11-K16 with ray differentials through path tracing, replay (r-buffer uv-gradient channel) and the reconnection
    shift's isotropic LOD; moving camera, 6 / 8 bounces so that k > 2 replays and Russian roulette occur."""
    orc, hx = pair
    sc, offs = textured
    w, h = 40, 32
    prm = _params(max_non_tr_bounces=6, max_glossy_tr_bounces=8)
    if mode == "none":
        prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
    o, x = zro.OracleRPT(orc, w, h), zhx.HostExecRPT(hx, w, h)
    ks = set()
    for f, cb in _frames(sc, offs, w, h, 4):
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        _rpt_same(o, x, f)
        ks |= set((o.plane("A")[..., 0] & 0xf).ravel().tolist())
    assert np.isfinite(a).all() and a[..., :3].max() > 0
    assert any(k not in (0, 15) for k in ks), ks         # reconnections beyond the first indirect vertex occurred


def test_sun_sky_textured_bit_exact():
    """NEE_EMISSIVE == 0 x TEXTURED permutations of K9, K10 and K11-K16 (the reference's default cornell.gltf is exactly this:
    sun + sky over a textured floor)."""
    sc = scene_io.make_synthetic_scene(num_tris=1200, num_emissive=0, seed=5, open_top=True)
    offs = scene_io.add_test_textures(sc)
    orc, hx = zro.OracleScene(sc, force_bvh=True), zhx.HostExecScene(sc)
    w, h = 40, 32
    prm = _params()
    sd = np.array((0.3, -0.8, 0.4), np.float32)
    po, px_ = zro.OracleRPT(orc, w, h), zhx.HostExecRPT(hx, w, h)
    go, gx = zro.OracleRGI(orc, w, h), zhx.HostExecRGI(hx, w, h)
    prev = None
    for f in range(1, 4):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.05 * max(0, f - 2), 2.0, -3.5))
        scene_io.set_texture_heap_offsets(cb, offs)
        cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        orc.sky_lut(cb, 256, 128); hx.sky_lut(cb, 256, 128)
        a, b = po.render(cb, prm), px_.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"RPT frame {f}"
        _rpt_same(po, px_, f)
        a, b = go.render(cb, prm), gx.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"GI frame {f}"
    keep, planes = orc.gbuffer(cb)          # (the ctypes plane table points into `keep`)
    fo, _ = orc.pathtrace(cb, planes, prm)
    fh, _ = hx.pathtrace(cb, planes, prm)
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32)) and fo[..., :3].max() > 0
