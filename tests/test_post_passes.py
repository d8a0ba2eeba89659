"""Post stack: auto exposure (AutoExposure_Histogram.hlsl, AutoExposure_WeightedAvg.hlsl) and display / tone mapping
(Display.hlsl, Tonemap.hlsli).  tests/golden/ref_post.npz holds what the REFERENCE's own shaders, compiled as C++ (oracle/_ref), produce on
the seeded inputs of tools/post_cases.py; the oracle restatement (CPU) and the HIP passes (GPU) must reproduce it bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import post_cases as pc  # noqa: E402
from oracle import zref, zro  # noqa: E402
from zetaray_amd import api  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "ref_post.npz")


def _i(x):
    return int(np.asarray(x).reshape(-1)[0])


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same(got, want, what):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = bits(got) != bits(want)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values differ, first at {np.argwhere(bad)[0]}: {got[tuple(np.argwhere(bad)[0])]} vs {want[tuple(np.argwhere(bad)[0])]}"


def ae_inputs(case):
    name, frames, dts, f16, over = case
    prm = pc.params(**over)
    imgs = [pc.hdr_image(**kw) for kw in frames]
    return name, [pc.to_half_bits(i) if f16 else i for i in imgs], dts, prm


def display_inputs(case):
    name, tm, ae, sat, agx, f16, disp = case
    img = pc.hdr_image(seed=11)
    return name, (pc.to_half_bits(img) if f16 else img), pc.params(tm, ae, sat, agx), pc.frame_constants(display=disp)


@pytest.mark.parametrize("case", pc.AE_CASES, ids=[c[0] for c in pc.AE_CASES])
def test_oracle_auto_exposure_reproduces_reference_shaders(case):
    gold = np.load(GOLD)
    name, imgs, dts, prm = ae_inputs(case)
    e = np.zeros(2, np.float32)
    for i, (img, dt) in enumerate(zip(imgs, dts)):
        hist, e = zro.auto_exposure(img, prm, np.float32(dt), e)
        assert_same(hist, gold[f"{name}/hist{i}"], f"{name} histogram, frame {i}")
        assert_same(e, gold[f"{name}/exposure{i}"], f"{name} exposure, frame {i}")
        assert int(hist.sum()) == img.shape[0] * img.shape[1]
    assert e[0] > 0 and np.isfinite(e).all()


@pytest.mark.parametrize("case", pc.DISPLAY_CASES, ids=[c[0] for c in pc.DISPLAY_CASES])
def test_oracle_display_reproduces_reference_shader(case):
    gold = np.load(GOLD)
    name, img, prm, cb = display_inputs(case)
    rgba, srgb = zro.display(img, prm, (_i(cb["display_width"]), _i(cb["display_height"])), pc.DISPLAY_EXPOSURE, api.load_tonemap_lut())
    assert_same(rgba, gold[f"{name}/rgba"], name)
    # the 8-bit back buffer: monotone in the linear value, 0 / 255 at the ends, NaN -> 0
    lin = rgba[..., :3]
    assert (srgb[..., 3] == 255).all()
    assert (srgb[..., :3][np.nan_to_num(lin, nan=0.0) <= 0] == 0).all() and (srgb[..., :3][lin >= 1] == 255).all()


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libzref_post.so")), reason="needs oracle/_ref (built from /root/reference)")
def test_oracle_post_matches_reference_shaders_live():
    """fresh inputs (not the golden ones), larger image, every tone mapper, both input formats, three adaptation steps"""
    ref = zref.RefPost()
    lut = api.load_tonemap_lut()
    e_ref = e_orc = np.zeros(2, np.float32)
    for f in range(3):
        img = pc.hdr_image(seed=100 + f, w=160, h=90, scale=[1.0, 6.0, 0.05][f])
        src = pc.to_half_bits(img) if f == 1 else img
        prm = pc.params(min_lum=[5e-3, 2e-3, 5e-2][f], lum_map_exp=[0.5, 0.8, 0.25][f])
        cb = pc.frame_constants(render=(160, 90), dt=[1 / 60, 1 / 24, 2.0][f])
        h_ref, e_ref = ref.auto_exposure(src, prm, cb, e_ref)
        h_orc, e_orc = zro.auto_exposure(src, prm, cb["dt"], e_orc)
        assert_same(h_orc, h_ref, f"live histogram {f}")
        assert_same(e_orc, e_ref, f"live exposure {f}")
        for tm in pc.TONEMAPPERS:
            p2 = pc.params(tm, f != 2, 0.75, 1.2)
            cbd = pc.frame_constants(render=(160, 90), display=(200, 113) if f == 0 else None)
            want = ref.display(src, p2, cbd, e_ref, lut)
            got, _ = zro.display(src, p2, (_i(cbd["display_width"]), _i(cbd["display_height"])), e_ref, lut)
            assert_same(got, want, f"live display {tm} frame {f}")


# ------------------------------------------------------------------------------------------------ HIP passes
def _upload(t, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.fixture(scope="module")
def tiny_scene():
    from zetaray_amd import scene_io
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    return api.Scene(sc)


@pytest.mark.gpu
@pytest.mark.parametrize("case", pc.AE_CASES, ids=[c[0] for c in pc.AE_CASES])
def test_hip_auto_exposure_reproduces_reference_shaders(case, tiny_scene):
    import torch
    gold = np.load(GOLD)
    name, imgs, dts, prm = ae_inputs(case)
    h, w = imgs[0].shape[:2]
    p = api.Pass(api.PASS_AUTO_EXPOSURE, w, h, params=prm)
    e_orc = np.zeros(2, np.float32)
    for i, (img, dt) in enumerate(zip(imgs, dts)):
        dev = _upload(torch, img)
        p.set_input(api.IN_POST_SIGNAL_F16 if img.dtype == np.uint16 else api.IN_POST_SIGNAL_F32, dev.data_ptr())
        p.render(pc.frame_constants(dt=dt), tiny_scene)
        torch.cuda.synchronize()
        hist = p.download_raw(api.OUT_AE_HISTOGRAM, np.uint32, (256,))
        e = p.download_raw(api.OUT_EXPOSURE, np.float32, (2,))
        assert_same(hist, gold[f"{name}/hist{i}"], f"{name} histogram, frame {i}")
        assert_same(e, gold[f"{name}/exposure{i}"], f"{name} exposure, frame {i}")
        _, e_orc = zro.auto_exposure(img, prm, np.float32(dt), e_orc)
        assert_same(e, e_orc, f"{name} vs oracle, frame {i}")
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", pc.DISPLAY_CASES, ids=[c[0] for c in pc.DISPLAY_CASES])
def test_hip_display_reproduces_reference_shader(case, tiny_scene):
    import torch
    gold = np.load(GOLD)
    name, img, prm, cb = display_inputs(case)
    dw, dh = _i(cb["display_width"]), _i(cb["display_height"])
    p = api.Pass(api.PASS_DISPLAY, dw, dh, params=prm)
    p.set_tonemap_lut()
    dev, exp = _upload(torch, img), _upload(torch, pc.DISPLAY_EXPOSURE)
    p.set_input(api.IN_POST_SIGNAL_F16 if img.dtype == np.uint16 else api.IN_POST_SIGNAL_F32, dev.data_ptr())
    p.set_input(api.IN_DISPLAY_EXPOSURE, exp.data_ptr())
    p.render(cb, tiny_scene)
    torch.cuda.synchronize()
    rgba = p.download_raw(api.OUT_DISPLAY, np.float32, (dh, dw, 4))
    srgb = p.download_raw(api.OUT_DISPLAY_SRGB8, np.uint8, (dh, dw, 4))
    assert_same(rgba, gold[f"{name}/rgba"], name)
    _, srgb_orc = zro.display(img, prm, (dw, dh), pc.DISPLAY_EXPOSURE, api.load_tonemap_lut())
    assert_same(srgb, srgb_orc, name + " sRGB8")
    p.close()


@pytest.mark.gpu
def test_hip_post_chain_1080p_properties(tiny_scene):
    """full size: the histogram counts every pixel exactly once, exposure is finite and positive, and scaling the image by k scales the
    adapted luminance's steady state by about k (the histogram is 254 bins wide) -- plus bit-equality with the oracle at 1080p"""
    import torch
    w, h = 1920, 1080
    img = pc.hdr_image(seed=7, w=w, h=h)
    prm = pc.params(adaptation_rate=1e6)          # adapt in one step
    p = api.Pass(api.PASS_AUTO_EXPOSURE, w, h, params=prm)
    cb = pc.frame_constants(render=(w, h))
    res = []
    for k in (1.0, 2.0):
        dev = _upload(torch, (img * np.float32([k, k, k, 1.0])).astype(np.float32))
        p.set_input(api.IN_POST_SIGNAL_F32, dev.data_ptr())
        p.render(cb, tiny_scene)
        torch.cuda.synchronize()
        hist = p.download_raw(api.OUT_AE_HISTOGRAM, np.uint32, (256,))
        e = p.download_raw(api.OUT_EXPOSURE, np.float32, (2,))
        assert int(hist.sum()) == w * h and np.isfinite(e).all() and e[0] > 0
        res.append((hist, e))
    h_orc, e_orc = zro.auto_exposure(img, prm, cb["dt"], np.zeros(2, np.float32))
    assert_same(res[0][0], h_orc, "1080p histogram vs oracle")
    assert_same(res[0][1], e_orc, "1080p exposure vs oracle")
    assert 1.5 < res[1][1][1] / res[0][1][1] < 2.5
    p.close()
