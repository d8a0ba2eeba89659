"""TAA (`-m "not gpu"`): the HIP stage function (zr_taa.h, run by the serial host executor) against the oracle's restatement of
TAA.hlsl, bit-exact over a moving-camera sequence, plus the properties the reference's filter has by construction."""
import os

import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _f16(a):
    return a.view(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def sequence():
    """5 frames of 1-spp path tracing over a moving camera (signal + G-buffer depth / motion)"""
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    orc = zro.OracleScene(sc)
    w, h = 72, 48
    frames, prev = [], None
    for f in range(1, 6):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0.05 * max(0, f - 2), 1.2, -4.043),
                                           jitter=(0.25 * ((f * 7) % 4 - 1.5) / 2, 0.25 * ((f * 3) % 4 - 1.5) / 2))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        arrays, planes = orc.gbuffer(cb)
        sig, _ = orc.pathtrace(cb, planes, wire.default_params())
        frames.append((sig.copy(), arrays[7].reshape(h, w).copy(), arrays[3].reshape(h, w).copy()))
    return w, h, frames


def test_taa_bit_exact_and_properties(sequence):
    w, h, frames = sequence
    ho = np.zeros((h, w, 4), np.uint16)
    hh = ho.copy()
    for f, (sig, depth, motion) in enumerate(frames):
        valid = f > 0
        no = zro.taa(sig, depth, motion, ho, 0.1, valid)
        nh = zhx.taa(sig, depth, motion, hh, 0.1, valid)
        assert np.array_equal(no, nh), f"frame {f}"
        if f == 0:      # TemporalIsValid == 0: the signal itself, rounded to half
            assert np.array_equal(no[..., :3], sig[..., :3].astype(np.float16).view(np.uint16))
        ho, hh = no, nh
    out = _f16(ho)[..., :3]
    assert np.isfinite(out).all() and out.max() > 0
    # temporal accumulation reduces the frame-to-frame noise of the 1-spp signal
    sig_last = frames[-1][0][..., :3]
    geo = frames[-1][1] != np.float32(3.402823466e+38)
    lum = lambda a: a @ np.array([0.2126, 0.7152, 0.0722], np.float32)
    def rough(a):
        l = lum(a)
        return np.abs(l[1:-1, 1:-1] - 0.25 * (l[:-2, 1:-1] + l[2:, 1:-1] + l[1:-1, :-2] + l[1:-1, 2:]))[geo[1:-1, 1:-1]].mean()
    assert rough(out) < 0.6 * rough(sig_last)


def test_taa_static_converges_to_reconstruction_and_clips_history():
    """Constant signal + zero motion: output == signal; a wildly different history is clipped to the neighbourhood."""
    w, h = 16, 12
    sig = np.zeros((h, w, 4), np.float32); sig[..., :3] = (0.25, 0.5, 0.125)
    depth = np.full((h, w), 2.0, np.float32)
    motion = np.zeros((h, w), np.uint32)
    hist = np.zeros((h, w, 4), np.uint16); hist[..., :3] = np.float16(40.0).view(np.uint16)
    o = zro.taa(sig, depth, motion, hist, 0.1, True)
    assert np.array_equal(o, zhx.taa(sig, depth, motion, hist, 0.1, True))
    assert np.allclose(_f16(o)[..., :3], sig[..., :3], atol=2e-3)       # zero variance -> history clipped onto the mean
    # pixels without geometry pass the signal through
    depth[3, 4] = np.float32(3.402823466e+38)
    o = zro.taa(sig, depth, motion, hist, 0.1, True)
    assert np.array_equal(o[3, 4, :3], sig[3, 4, :3].astype(np.float16).view(np.uint16))
