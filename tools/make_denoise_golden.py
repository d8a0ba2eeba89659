"""tests/golden/denoise_spec.npz: the denoise pass's specification as vectors.  The pass has no reference counterpart (zr_svgf.h defines it), so what
pins it from round to round is this file: seeded synthetic inputs (a G-buffer with a depth / normal edge and misses, four noisy frames, a 1.5-px
motion from frame 2 on, a history reset, a NaN) and the outputs of oracle/zro_svgf.h for them -- output, colour history, moments of every frame.
tests/test_denoise.py requires oracle == host-executed HIP stage functions == these arrays, bit for bit.  python tools/make_denoise_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs():
    from tests.test_denoise import _planes
    h, w = 40, 56
    rng = np.random.default_rng(20260926)
    depth, normal = _planes(h, w, rng)
    mv = np.uint32(int(round(1.5 / w * 32767.0)) & 0xffff) | (np.uint32(int(round(-0.5 / h * 32767.0)) & 0xffff) << np.uint32(16))
    frames = []
    for f in range(4):
        sig = np.zeros((h, w, 4), np.float32)
        sig[..., :3] = rng.uniform(0.0, 3.0, (h, w, 3)).astype(np.float32) * np.where(np.arange(w)[None, :, None] < w // 2, 1.0, 0.2).astype(np.float32)
        if f == 2:
            sig[5, 7, 1] = np.nan
        frames.append((sig, np.full((h, w), mv if f >= 2 else 0, np.uint32), f not in (0, 3)))
    return depth, normal, frames, dict(iterations=4, sigma_l=3.0, normal_power_log2=6)


def main():
    from oracle import zro
    depth, normal, frames, kw = inputs()
    h, w = depth.shape
    hc, hm = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 2), np.float32)
    out = {"depth": depth, "normal": normal}
    for f, (sig, motion, valid) in enumerate(frames):
        o, hc, hm = zro.svgf(sig, depth, normal, motion, depth, normal, hc, hm, temporal_valid=valid, **kw)
        out[f"signal{f}"], out[f"motion{f}"] = sig, motion
        out[f"out{f}"], out[f"hist{f}"], out[f"mom{f}"] = o, hc.copy(), hm.copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "denoise_spec.npz"), **out)
    print("wrote tests/golden/denoise_spec.npz", {k: v.shape for k, v in out.items() if k.endswith("0")})


if __name__ == "__main__":
    main()
