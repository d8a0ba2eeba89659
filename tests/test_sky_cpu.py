"""Sun / sky (`-m "not gpu"`): the sky-view LUT (K17), Le_Sky / Le_Sun, and the K9 path tracer with sun + sky next-event
estimation (NEE_EMISSIVE == 0 shader variants) -- HIP stage functions run by the serial host executor vs the oracle."""
import os

import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cornell_sky():
    return scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell.npz"))


@pytest.fixture(scope="module")
def pair(cornell_sky):
    return zro.OracleScene(cornell_sky), zhx.HostExecScene(cornell_sky)


def _unpack_r11g11b10(v):
    def uf(bits, m):
        e = bits >> m
        man = bits & ((1 << m) - 1)
        return np.where(e == 0, man / float(1 << m) * 2.0 ** -14, (1 + man / float(1 << m)) * 2.0 ** (e.astype(np.float64) - 15))
    return np.stack([uf(v & 0x7FF, 6), uf((v >> 11) & 0x7FF, 6), uf(v >> 22, 5)], -1)


def test_sky_lut_bit_exact_and_plausible(pair):
    orc, hx = pair
    cb = scene_io.make_frame_constants(64, 64)
    lo = orc.sky_lut(cb, 256, 128)
    lh = hx.sky_lut(cb, 256, 128)
    assert np.array_equal(lo, lh)
    rgb = _unpack_r11g11b10(lo)
    assert np.isfinite(rgb).all() and rgb.max() > 0
    # the default sun sits 3 degrees above the horizon: the horizon rows are brighter than the zenith row, blue > red overhead
    assert rgb[60:68].mean() > rgb[0].mean()
    assert rgb[0, :, 2].mean() > rgb[0, :, 0].mean()
    # below the horizon the planet blocks the view ray early: darker than just above it
    assert rgb[100].mean() < rgb[62].mean()
    # a different sun -> a different LUT
    cb2 = cb.copy()
    cb2["sun_dir"] = np.array([0.0, -1.0, 0.0], np.float32)
    assert not np.array_equal(orc.sky_lut(cb2, 64, 32), orc.sky_lut(cb, 64, 32))
    orc.sky_lut(cb, 256, 128)


def test_le_sky_and_le_sun(pair):
    orc, hx = pair
    cb = scene_io.make_frame_constants(64, 64)
    lut = orc.sky_lut(cb, 256, 128)
    hx.sky_lut(cb, 256, 128)
    rng = np.random.default_rng(7)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d = np.concatenate([d, np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], np.float32)])
    a, b = orc.le_sky(d), hx.le_sky(d)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rgb = _unpack_r11g11b10(lut)
    assert a.min() >= 0 and a.max() <= rgb.max() + 1e-6          # bilinear interpolation stays inside the texel range
    # straight up lands on v = 0, i.e. half-way between the first row and -- wrap addressing, as g_samLinearWrap -- the last
    assert np.allclose(a[4096], 0.5 * (rgb[0].mean(0) + rgb[-1].mean(0)), rtol=0.02)
    pos = (rng.random((256, 3)).astype(np.float32) - 0.5) * np.float32(20.0)
    s1, s2 = orc.le_sun(cb, pos), hx.le_sun(cb, pos)
    assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
    assert (s1 > 0).all() and (s1 <= float(cb["sun_illuminance"])).all()
    assert (s1[:, 0] > s1[:, 2]).all()                            # low sun: blue is scattered out more than red


@pytest.mark.parametrize("w,h,frame", [(64, 48, 1), (96, 64, 3)])
def test_path_tracer_sun_sky_bit_exact(pair, cornell_sky, w, h, frame):
    orc, hx = pair
    assert len(cornell_sky.emissives) == 0
    cb = scene_io.make_frame_constants(w, h, frame_num=frame, num_emissives=0)
    orc.sky_lut(cb, 256, 128)
    hx.sky_lut(cb, 256, 128)
    ga, gp = orc.gbuffer(cb)
    ha, hp = hx.gbuffer(cb)
    for name, x, y in zip(wire.GB_PLANE_NAMES, ga, ha):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), name
    prm = wire.default_params()
    fo, co = orc.pathtrace(cb, gp, prm)
    fh, ch = hx.pathtrace(cb, hp, prm)
    assert co == ch
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))
    assert fo[..., :3].sum() > 0
    assert co[1] > 0          # visibility rays were traced
