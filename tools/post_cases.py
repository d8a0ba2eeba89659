"""Inputs of the post-stack parity cases (auto exposure + display / tone mapping), shared by tools/make_ref_post_goldens.py (which runs the
reference's own shaders, oracle/_ref/libzref_post.so) and tests/test_post_passes.py (oracle, HIP library).  Everything is generated from a
seed: only the reference's OUTPUTS are committed (tests/golden/ref_post.npz)."""
import numpy as np

from zetaray_amd import scene_io, wire

RENDER = (96, 54)          # w, h
DISPLAY_UPSCALED = (144, 81)
TONEMAPPERS = ["none", "neutral", "agx_default", "agx_golden", "agx_punchy", "agx_custom"]


def hdr_image(seed, w=RENDER[0], h=RENDER[1], scale=1.0):
    """An HDR test image: log-normal radiance with dark / black / negative regions and a few very bright texels (RGBA32F)."""
    rng = np.random.default_rng(seed)
    img = np.exp(rng.normal(-1.0, 1.6, (h, w, 3))).astype(np.float32) * np.float32(scale)
    img[: h // 6] *= np.float32(1e-3)                    # a dark band: many pixels fall into the low bins
    img[h // 6: h // 5] = 0.0                            # black: bin 0 ("excluded")
    img[-2:, : w // 4] *= np.float32(-1.0)               # negative radiance: luminance <= EPS
    ys, xs = rng.integers(0, h, 24), rng.integers(0, w, 24)
    img[ys, xs] *= np.float32(400.0)                     # fireflies: saturate the top bin, exercise the tone mappers' shoulders
    a = np.ones((h, w, 1), np.float32)
    return np.concatenate([img, a], axis=2)


def to_half_bits(img):
    return np.ascontiguousarray(img.astype(np.float16).view(np.uint16))


def frame_constants(render=RENDER, display=None, dt=1.0 / 60.0):
    cb = scene_io.make_frame_constants(render[0], render[1])
    cb["dt"] = np.float32(dt)
    d = display or render
    cb["display_width"], cb["display_height"] = d
    return cb


def params(tonemapper="neutral", auto_exposure=True, saturation=1.0, agx_exp=1.0, **ae):
    p = wire.default_params()
    p.display_tonemapper = TONEMAPPERS.index(tonemapper)
    p.display_auto_exposure = int(auto_exposure)
    p.display_saturation, p.display_agx_exp = saturation, agx_exp
    for k, v in ae.items():
        setattr(p, "ae_" + k, v)
    return p


# auto exposure: (name, image kwargs per frame, dt per frame, f16 input?, parameter overrides)
AE_CASES = [
    ("ae_adapt_f32", [dict(seed=1), dict(seed=2, scale=3.0), dict(seed=3, scale=0.2), dict(seed=3, scale=0.2)], [1 / 60, 1 / 60, 1 / 30, 0.5], False, {}),
    ("ae_f16_params", [dict(seed=4), dict(seed=5, scale=8.0)], [1 / 144, 1 / 144], True, dict(min_lum=1e-2, max_lum=8.0, lum_map_exp=0.35, adaptation_rate=2.5)),
]
# display: (name, tonemapper, auto exposure, saturation, agx exponent, f16 input?, display size)
DISPLAY_CASES = [(f"display_{t}", t, True, 1.0, 1.0, False, None) for t in TONEMAPPERS] + [
    ("display_neutral_desat_f16_upscaled", "neutral", True, 0.6, 1.0, True, DISPLAY_UPSCALED),
    ("display_agx_custom_params", "agx_custom", False, 1.3, 0.85, True, None),
]
DISPLAY_EXPOSURE = np.array([0.37, 1.9], np.float32)      # the exposure texel the display cases read
