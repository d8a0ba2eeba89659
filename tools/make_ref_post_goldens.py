"""tests/golden/ref_post.npz <- the REFERENCE's auto-exposure and display shaders (oracle/_ref/libzref_post.so, built from /root/reference
by oracle/_ref.mk) on the seeded inputs of tools/post_cases.py.  Run in the build container:  python tools/make_ref_post_goldens.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import post_cases as pc  # noqa: E402
from oracle import zref  # noqa: E402
from zetaray_amd import api  # noqa: E402


def run_ae(ref, case):
    name, frames, dts, f16, over = case
    prm = pc.params(**over)
    e = np.zeros(2, np.float32)
    out = {}
    for i, (kw, dt) in enumerate(zip(frames, dts)):
        img = pc.hdr_image(**kw)
        hist, e = ref.auto_exposure(pc.to_half_bits(img) if f16 else img, prm, pc.frame_constants(dt=dt), e)
        out[f"{name}/hist{i}"] = hist
        out[f"{name}/exposure{i}"] = e.copy()
    return out


def run_display(ref, case, lut):
    name, tm, ae, sat, agx, f16, disp = case
    prm = pc.params(tm, ae, sat, agx)
    img = pc.hdr_image(seed=11)
    rgba = ref.display(pc.to_half_bits(img) if f16 else img, prm, pc.frame_constants(display=disp), pc.DISPLAY_EXPOSURE, lut)
    return {f"{name}/rgba": rgba}


def main():
    ref = zref.RefPost()
    lut = api.load_tonemap_lut()
    out = {}
    for c in pc.AE_CASES:
        out.update(run_ae(ref, c))
    for c in pc.DISPLAY_CASES:
        out.update(run_display(ref, c, lut))
    dst = os.path.join(ROOT, "tests", "golden", "ref_post.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
