#!/usr/bin/env python
"""Parity of the tolerance-mode arithmetic build (libzetaray_amd_fast.so, `make -C zetaray_amd/csrc fast`; include/zr_detmath.h
ZR_ARITH_FAST) against the CPU oracle, in the two senses BASELINE.json's north_star states:

  * integer reservoir state after frame 1 (ReSTIR PT: k / M / lobes / light types of plane A, the replay seed and light ID of plane C, the NEE
    seed and mesh index of plane G; ReSTIR DI: M and lightIdx of plane A) -- the share of pixels on which it is EQUAL to the oracle's;
  * the radiance of the N-frame accumulated image -- relative L2 of the whole image, and the share of pixels whose own relative error exceeds 5 %.

Runs in its own process because the library is selected at load time (ZETARAY_AMD_LIB); prints one JSON object.  tests/test_fast_arith.py
asserts on it; scripts/gpu_r04_*.sh record it under profiles/.  Test infrastructure: imports the oracle as the checker.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=192)
    ap.add_argument("--height", type=int, default=108)
    ap.add_argument("--frames", type=int, default=256)
    args = ap.parse_args()
    os.environ.setdefault("ZETARAY_AMD_LIB", os.path.join(ROOT, "zetaray_amd", "libzetaray_amd_fast.so"))
    from zetaray_amd import api, scene_io, wire
    from oracle import zro
    assert os.path.basename(api.LIB_PATH).startswith("libzetaray_amd_fast"), api.LIB_PATH
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    w, h = args.width, args.height
    prm = wire.default_params()
    out = {"lib": os.path.basename(api.LIB_PATH), "width": w, "height": h, "frames": args.frames}

    o = zro.OracleScene(sc)
    # ---- ReSTIR PT: frame-1 integer state, then the accumulated image
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    orpt = zro.OracleRPT(o, w, h)
    acc_g = np.zeros((h, w, 3), np.float64)
    acc_o = np.zeros((h, w, 3), np.float64)
    for f in range(1, args.frames + 1):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives))
        r.render_frame(cb)
        got = r.final()
        want = orpt.render(cb, prm)
        acc_g += got[..., :3]
        acc_o += want[..., :3]
        if f == 1:
            eq = np.ones((h, w), bool)
            for nm, sel in (("A", None), ("C", (1, 2)), ("G", (0, 1))):
                a, b = r.p_indirect.download_plane(nm), orpt.plane(nm)
                if nm == "A":
                    a, b = (a & 0xffffff), (b & 0xffffff)
                    e = (a == b).reshape(h, w, -1).all(axis=2)
                else:
                    e = (a.reshape(h, w, -1)[..., list(sel)] == b.reshape(h, w, -1)[..., list(sel)]).all(axis=2)
                out[f"rpt_frame1_plane_{nm}_equal_share"] = round(float(e.mean()), 6)
                eq &= e
            out["rpt_frame1_integer_state_equal_share"] = round(float(eq.mean()), 6)
            out["rpt_frame1_radiance_bit_equal_share"] = round(float((got.view(np.uint32) == want.view(np.uint32)).all(axis=2).mean()), 6)
            out["rpt_frame1_ray_counters"] = {"fast": list(r.p_indirect.read_counters()), "oracle": list(orpt.counters)}
    mg, mo = acc_g / args.frames, acc_o / args.frames
    out["rpt_accum_rel_l2"] = float(np.sqrt(((mg - mo) ** 2).sum() / (mo ** 2).sum()))
    px_err = np.sqrt(((mg - mo) ** 2).sum(axis=2)) / np.maximum(np.sqrt((mo ** 2).sum(axis=2)), 1e-3)
    out["rpt_accum_px_rel_err_p50"] = float(np.percentile(px_err, 50))
    out["rpt_accum_px_rel_err_p99"] = float(np.percentile(px_err, 99))
    out["rpt_accum_px_share_above_5pct"] = float((px_err > 0.05).mean())
    out["rpt_accum_mean_radiance"] = {"fast": float(mg.mean()), "oracle": float(mo.mean())}

    # ---- ReSTIR DI (emissive): frame-1 light picks
    dip = wire.default_params_di()
    r2 = api.Renderer(sc, w, h, params=prm)
    r2.enable_direct(dip)
    r2.skip_indirect = True
    odi = zro.OracleRDI(o, w, h)
    cb = scene_io.make_frame_constants(w, h, frame_num=1, num_emissives=len(sc.emissives))
    r2.render_frame(cb)
    odi.render(cb, dip)
    a = r2.p_direct.download_plane("di_A").reshape(h, w, 4)
    b = odi.plane("A").reshape(h, w, 4)
    out["rdi_frame1_lightIdx_equal_share"] = round(float((a[..., 3] == b[..., 3]).mean()), 6)
    out["rdi_frame1_M_equal_share"] = round(float(((a[..., 2] >> 16) == (b[..., 2] >> 16)).mean()), 6)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
