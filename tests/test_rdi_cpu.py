"""ReSTIR DI for emissive lights (K5 / K6) on CPU: HIP stage functions (zr_rdi.h, run serially by tests/hostexec) against
the oracle (oracle/zro_rdi.h): bit-exact radiance, reservoir planes and ray counters over multi-frame sequences; plus the
property that pins the oracle: the estimator's mean does not depend on which reuse passes are on (RIS with MIS is unbiased)."""
import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire


def _cb(sc, w, h, f, **kw):
    return scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), **kw)


def _same(o, x, f):
    for nm in ("A", "B", "target"):
        assert np.array_equal(o.plane(nm).view(np.uint8), x.plane(nm).view(np.uint8)), f"frame {f}: DI plane {nm} differs"
    assert o.counters == x.counters


@pytest.fixture(scope="module")
def hx_emissive(cornell_emissive, oracle_emissive):
    return zhx.HostExecScene(cornell_emissive, oracle_emissive.alias)


@pytest.mark.parametrize("mode", ["full", "temporal", "none", "plain_spatial"])
def test_rdi_cornell_bit_exact(cornell_emissive, oracle_emissive, hx_emissive, mode):
    w, h = 72, 48
    prm = wire.default_params_di()
    if mode == "temporal":
        prm.flags &= ~wire.IND_SPATIAL_RESAMPLE
    if mode == "none":
        prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
    if mode == "plain_spatial":
        prm.flags &= ~(wire.DI_STOCHASTIC_SPATIAL | wire.DI_EXTRA_DISOCCLUSION_SAMPLING)
    o, x = zro.OracleRDI(oracle_emissive, w, h), zhx.HostExecRDI(hx_emissive, w, h)
    prev = None
    for f in range(1, 6):
        # moving camera from frame 3 on: motion vectors, disocclusion at the borders -> the 4-sample spatial branch
        cam = (0.06 * max(0, f - 2), 1.2, -4.043)
        cb = _cb(cornell_emissive, w, h, f, cam_pos=cam)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        a, b = o.render(cb, prm), x.render(cb, prm)
        assert not np.isnan(a).any()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}: radiance differs"
        _same(o, x, f)
    assert a[..., :3].max() > 0


def test_rdi_materials_presampled_bit_exact():
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
    osc = zro.OracleScene(sc, force_bvh=True)
    hx = zhx.HostExecScene(sc, osc.alias)
    w, h = 64, 48
    for presample in (0, 1):
        prm = wire.default_params_di()
        prm.presampling, prm.num_sample_sets, prm.sample_set_size = presample, 16, 64
        o, x = zro.OracleRDI(osc, w, h), zhx.HostExecRDI(hx, w, h)
        for f in range(1, 4):
            cb = _cb(sc, w, h, f, cam_pos=(0, 0, -3.5))
            if presample:
                osc.presample(f, 16, 64); hx.presample(f, 16, 64)
            a, b = o.render(cb, prm), x.render(cb, prm)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"presample {presample} frame {f}"
            _same(o, x, f)


def test_rdi_mean_is_independent_of_reuse(cornell_emissive, oracle_emissive):
    w, h, n = 48, 32, 60
    lum = lambda a: a[..., :3] @ np.array([0.2126, 0.7152, 0.0722])
    means = {}
    for mode in ("none", "temporal", "full"):
        p = wire.default_params_di()
        if mode == "none":
            p.flags &= ~(wire.IND_TEMPORAL_RESAMPLE | wire.IND_SPATIAL_RESAMPLE)
        if mode == "temporal":
            p.flags &= ~wire.IND_SPATIAL_RESAMPLE
        acc = np.zeros((h, w), np.float64)
        for s in range(n):
            r = zro.OracleRDI(oracle_emissive, w, h)
            for j in range(1 if mode == "none" else 3):
                fin = r.render(_cb(cornell_emissive, w, h, 100 + 3 * s + j), p)
            acc += lum(fin)
        means[mode] = acc.mean() / n
    assert abs(means["temporal"] / means["none"] - 1) < 0.02 and abs(means["full"] / means["none"] - 1) < 0.02, means
