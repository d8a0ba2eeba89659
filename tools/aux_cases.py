"""Inputs of the auxiliary-pass pins (PreLighting K2 / K3 / K4, SkyViewLUT K17, the ReSTIR_GI_LVG permutation, Compositing, FireflyFilter,
TAA), shared by tools/make_ref_aux_goldens.py -- which runs the REFERENCE's own shaders (oracle/_ref/libzref_aux.so, libzref_gi_e1l.so) --
and tests/test_ref_aux.py (oracle on the CPU, HIP library on the GPU).  Everything is generated from seeds; only the reference's
outputs are committed (tests/golden/ref_aux.npz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_pass_cases as rc  # noqa: E402
from zetaray_amd import scene_io, wire  # noqa: E402

W, H = 96, 64
FLT_MAX = np.float32(3.4028234663852886e38)
COMPOSIT_SKY_DI, COMPOSIT_INDIRECT, COMPOSIT_EMISSIVE_DI = 1 << 1, 1 << 2, 1 << 5      # CB_COMPOSIT_FLAGS, Compositing_Common.h:12-19


# ---------------------------------------------------------------- PreLighting
def textured_scene():
    """150 emissive triangles with an emissive map (K2's Monte Carlo branch, K3's Le_EmissiveTriangle texture fetch)"""
    sc, force, kw = rc._scene("textured")
    cb = scene_io.make_frame_constants(W, H, frame_num=1, num_emissives=len(sc.emissives), **kw)
    scene_io.set_texture_heap_offsets(cb, rc.TEX_OFFSETS["textured"])
    return sc, force, cb


K3_FRAME, K3_SETS = 3, (16, 64)


def lvg_scene():
    sc = scene_io.make_synthetic_scene(num_tris=2000, num_emissive=600, seed=3)
    return sc, True


LVG_DIM, LVG_EXT, LVG_OFF = (8, 4, 10), (0.6, 0.45, 0.6), 0.1
GI_W, GI_H, GI_FRAMES = 64, 48, 3


def lvg_params():
    p = wire.default_params()
    p.presampling, p.num_sample_sets, p.sample_set_size = 1, 16, 64
    p.use_lvg = 1
    p.lvg_grid_dim = LVG_DIM[0] | (LVG_DIM[1] << 10) | (LVG_DIM[2] << 20)
    p.lvg_extents[:] = LVG_EXT
    p.lvg_offset_y = LVG_OFF
    return p


def lvg_frame(sc, f, w=GI_W, h=GI_H):
    return scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0.0, 0.0, -3.5))


# ---------------------------------------------------------------- sky
def sky_frames():
    a = scene_io.make_frame_constants(W, H, frame_num=1, num_emissives=0)
    b = a.copy()
    sd = np.array([0.2, -0.12, 0.97], np.float32)          # a low sun
    b["sun_dir"] = sd / np.float32(np.linalg.norm(sd))
    b["sun_illuminance"] = np.float32(35.0)
    return {"k17_default": a, "k17_low_sun": b}


# ---------------------------------------------------------------- Compositing / FireflyFilter / TAA
def hdr(seed, w=W, h=H, scale=1.0):
    """a lighting term: log-normal radiance with a few fireflies (RGBA32F)"""
    rng = np.random.default_rng(seed)
    a = np.exp(rng.normal(-1.0, 1.5, (h, w, 4))).astype(np.float32) * np.float32(scale)
    ys, xs = rng.integers(0, h, 30), rng.integers(0, w, 30)
    a[ys, xs, :3] *= np.float32(300.0)
    return a


def post_scene(kind):
    if kind == "sky":        # open top: miss pixels show Le_SkyWithSunDisk
        return scene_io.make_synthetic_scene(num_tris=1500, num_emissive=0, seed=5, open_top=True), True
    return scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz")), False


POST_FRAMES = 4


def post_frame(kind, sc, f):
    """moving + jittered camera; frame 4 accumulates (Accumulate && CameraStatic, NumFramesCameraStatic = 3)"""
    acc = int(f == 4)
    pos, vd = ((0.05 * max(0, f - 2), 2.0, -3.5), (0, 0.35, 1)) if kind == "sky" else ((0.05 * max(0, f - 2), 1.2, -4.043), (0, 0, 1))
    cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), cam_pos=pos, view_dir=vd,
                                       jitter=(0.25 * ((f * 7) % 4 - 1.5) / 2, 0.25 * ((f * 3) % 4 - 1.5) / 2),
                                       accumulate=acc, camera_static=acc, num_frames_static=3 if acc else 0)
    return cb


def chain_prev(cb, prev):
    if prev is not None:
        cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
    return cb


def post_terms(kind, f):
    """(sky_di, emissive_di, indirect, flags)"""
    di, ind = hdr(10 + f), hdr(20 + f, scale=0.5)
    if kind == "sky":
        return di, None, ind, COMPOSIT_SKY_DI | COMPOSIT_INDIRECT
    return None, di, ind, COMPOSIT_EMISSIVE_DI | COMPOSIT_INDIRECT


TAA_BLEND = 0.1
TAA_INVALID = (1, 3)       # frames on which TemporalIsValid is 0 (first frame; a reset before frame 3)
