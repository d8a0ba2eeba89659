"""ReSTIR DI for sun + sky (K7 / K8) on CPU: HIP stage functions (zr_sdi.h, run serially by tests/hostexec) against the oracle
(oracle/zro_sdi.h): bit-exact radiance, reservoir planes (A metadata, B oct32 direction / local half vector, C weights, target)
and ray counters over multi-frame sequences with a moving camera; accumulation of the sky radiance on miss pixels; plus the
property that pins the oracle: the estimator's mean does not depend on which reuse passes are on."""
import os

import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(o, x, f):
    for nm in ("A", "B", "C", "target"):
        assert np.array_equal(o.plane(nm).view(np.uint8), x.plane(nm).view(np.uint8)), f"frame {f}: sky DI plane {nm} differs"
    assert o.counters == x.counters


@pytest.fixture(scope="module")
def cornell_sky():
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell.npz"))
    return sc, zro.OracleScene(sc), zhx.HostExecScene(sc)


def _sequence(sc, osc, hx, w, h, prm, frames, move_from=3, cam0=(0.0, 1.2, -4.043), step=0.06, sun=None, **cbkw):
    o, x = zro.OracleSDI(osc, w, h), zhx.HostExecSDI(hx, w, h)
    prev = None
    a = None
    for f in range(1, frames + 1):
        cam = (cam0[0] + step * max(0, f - move_from + 1), cam0[1], cam0[2])
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=cam, **cbkw)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        if sun is not None:
            sd = np.array(sun, np.float32)
            cb["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
        prev = cb.copy()
        osc.sky_lut(cb, 256, 128)
        hx.sky_lut(cb, 256, 128)
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert not np.isnan(a).any()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}: radiance differs"
        _same(o, x, f)
    return a, o


@pytest.mark.parametrize("mode", ["full", "temporal", "none"])
def test_sdi_cornell_bit_exact(cornell_sky, mode):
    sc, osc, hx = cornell_sky
    prm = wire.default_params_sky_di()
    if mode == "temporal":
        prm.flags &= ~wire.IND_SPATIAL_RESAMPLE
    if mode == "none":
        prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
    a, o = _sequence(sc, osc, hx, 72, 48, prm, 5)
    assert a[..., :3].max() > 0
    A = o.plane("A")[..., 0]
    assert ((A >> 7) & 1).any() and ((A >> 4) & 1).any()            # valid reservoirs, some holding sky samples
    assert (A & 0xF).max() <= 15


def test_sdi_glossy_materials_half_vector_shift_bit_exact():
    """No emissives, metal / coat / glass / thin-walled surfaces, camera inside the soup: BSDF candidates on low-roughness lobes
    are stored as local half vectors (half-vector copy shift) and re-evaluated at temporal and spatial neighbours."""
    sc = scene_io.make_synthetic_scene(num_tris=1500, num_emissive=0, seed=5, open_top=True)
    osc, hx = zro.OracleScene(sc, force_bvh=True), zhx.HostExecScene(sc)
    prm = wire.default_params_sky_di()
    a, o = _sequence(sc, osc, hx, 64, 48, prm, 4, move_from=2, cam0=(0.0, 2.0, -3.5), step=0.05, sun=(0.3, -0.8, 0.4))
    A = o.plane("A")[..., 0]
    assert ((A >> 5) & 1).any(), "no reservoir used the half-vector copy shift"
    assert a[..., :3].max() > 0


def test_sdi_miss_pixels_accumulate_the_sky(cornell_sky):
    """flags.invalid pixels: final = prev * (N > 1) + Le_SkyWithSunDisk when accumulating, 0 otherwise (SkyDI_Temporal.hlsl:193-205)"""
    sc, osc, hx = cornell_sky
    w, h = 48, 32
    prm = wire.default_params_sky_di()
    o, x = zro.OracleSDI(osc, w, h), zhx.HostExecSDI(hx, w, h)
    imgs = []
    for f in range(1, 4):
        # look past the box: the upper rows see the sky
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0, cam_pos=(0.0, 1.2, -9.0), view_dir=(0.0, 0.5, 1.0),
                                           accumulate=1, camera_static=1, num_frames_static=f)
        osc.sky_lut(cb, 256, 128)
        hx.sky_lut(cb, 256, 128)
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
        imgs.append(a.copy())
    ga, _ = osc.gbuffer(cb)
    mr = np.asarray(ga[wire.GB_PLANE_NAMES.index("metallic_roughness")]).reshape(h, w, -1)[..., 0].astype(np.uint32)
    miss = ((mr & 0xFF) & 4) != 0           # GBuffer flags byte, bit 2 = invalid (no primary hit)
    assert miss.any() and not miss.all()
    sky = imgs[0][miss][:, :3]
    assert (sky > 0).all()
    assert np.allclose(imgs[2][miss][:, :3], 3.0 * sky, rtol=1e-6)      # frame 1 restarts (N = 1), frames 2 and 3 add


def test_sdi_mean_is_independent_of_reuse(cornell_sky):
    """RIS with MIS + pairwise-MIS reuse is unbiased: the frame-averaged image with temporal + spatial reuse matches the one
    without reuse far better than either matches zero (a sign / Jacobian / MIS error shifts the mean by tens of percent)."""
    sc, osc, _ = cornell_sky
    w, h = 40, 28
    means = {}
    for mode in ("none", "full"):
        prm = wire.default_params_sky_di()
        if mode == "none":
            prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
        o = zro.OracleSDI(osc, w, h)
        acc = np.zeros((h, w, 3), np.float64)
        n = 0
        for f in range(1, 41):
            cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0)
            osc.sky_lut(cb, 256, 128)
            img = o.render(cb, prm)
            if f > 4:
                acc += img[..., :3]
                n += 1
        means[mode] = acc / n
    a, b = means["none"].mean(), means["full"].mean()
    assert a > 0 and abs(a - b) / a < 0.05, (a, b)
