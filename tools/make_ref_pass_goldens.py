"""Generates tests/golden/ref_pass_*.npz: outputs of the REFERENCE's own shader passes compiled as C++ (oracle/_ref/libzref_k1.so,
libzref_k9_*.so, libzref_rpt_*.so; `make -C oracle -f _ref.mk`) on the scenarios of tools/ref_pass_cases.py.

    ref_pass_gbuffer.npz   K1 (GBufferRT_Inline.hlsl): the 10 G-buffer planes of frame 1 of every scene
    ref_pass_<case>.npz    K9 (PathTracer.hlsl): FINAL of every frame; ReSTIR PT (K11, K13-K16), ReSTIR GI (K10), ReSTIR DI (K5 / K6) and
                           sun + sky DI (K7 / K8): FINAL of every frame + the persistent reservoir planes after the last frame

Inputs the reference computes elsewhere are taken from the already-pinned parts: the alias table (bit-exact vs the reference's own
AliasTable_Build, tests/test_ref_pins.py), and K3 presampled sets / K17 sky LUT from the oracle (their building blocks are pinned in
tests/test_ref_hlsl_pins.py).  The committed files let machines without /root/reference (the GPU box) compare the oracle AND the HIP
product with the reference's outputs.  Run in the build container: python tools/make_ref_pass_goldens.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_pass_cases as RC  # noqa: E402
from oracle import zref, zro  # noqa: E402
from zetaray_amd import wire  # noqa: E402


def prepare(ref, o, sc, cb, f, prm):
    """scene-level inputs of frame f on the reference side"""
    if len(sc.emissives) == 0:
        ref.set_sky_lut(o.sky_lut(cb, 256, 128))
    elif f == 1:
        ref.set_alias_table(o.alias)
    if prm.presampling:
        ref.set_sample_sets(o.presample(f, prm.num_sample_sets, prm.sample_set_size), prm.num_sample_sets, prm.sample_set_size)


def make_ref(zref, sc, integ, prm, force_bvh):
    ps = bool(prm.presampling)
    if integ == "pt":
        return zref.RefPathTracer(sc, ps, force_bvh)
    if integ == "rpt":
        return zref.RefRestirPT(sc, RC.W, RC.H, ps, force_bvh)
    if integ == "gi":
        return zref.RefRestirGI(sc, RC.W, RC.H, ps, force_bvh)
    return zref.RefDirect(sc, RC.W, RC.H, sky=(integ == "sdi"), presampling=ps, force_bvh=force_bvh, half_vec=(integ == "di" and bool(prm.flags & wire.DI_HALF_VECTOR_COPY_SHIFT)))


def main():
    # optional arguments: case-name prefixes to (re)generate; the other cases' files and G-buffer entries are kept
    only = sys.argv[1:]
    gb_path = os.path.join(ROOT, "tests", "golden", "ref_pass_gbuffer.npz")
    gb_out = dict(np.load(gb_path)) if only and os.path.exists(gb_path) else {}
    for case in RC.CASES:
        if only and not any(case.startswith(o) for o in only):
            continue
        sc, force_bvh, integ, prm = RC.scene_and_params(case)
        o = zro.OracleScene(sc, force_bvh=force_bvh, cb=RC.first_cb(case))
        k1 = zref.RefGBuffer(sc, force_bvh)
        ref = make_ref(zref, sc, integ, prm, force_bvh)
        res = {}
        anim = RC.Animator(sc, light=case in RC.MOVING_LIGHT) if case in RC.ANIMATED else None
        for f, cb in RC.frames_of(case):
            if anim is not None and f >= 2:
                anim.apply(f, o, k1, ref)
            prepare(ref, o, sc, cb, f, prm)
            arrays, planes = k1.render(cb)
            if f == (3 if anim is not None else 1):          # animated cases: the G-buffer of a frame in motion (motion vectors, prev transform)
                for n, a in zip(wire.GB_PLANE_NAMES, arrays):
                    gb_out[f"{case}_{n}"] = a.copy()
            res[f"final_{f}"] = ref.render(cb, planes, prm) if integ == "pt" else ref.render(cb, prm, (arrays, planes))
        for nm in RC.PLANES[integ]:
            res["plane_" + nm] = ref.plane(nm)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"ref_pass_{case}.npz"), **res)
        print(case, {k: float(np.asarray(v, np.float64).mean()) for k, v in res.items() if k.startswith("final")})
    # GBufferRT::PickPixel: what the reference's K1 writes to g_pick[0] (hitMeshIdx, UINT32_MAX on a miss) for a grid of pixels of frame 1
    for case in RC.PICK_CASES:
        if only and not any(case.startswith(o) or o == "pick" for o in only):
            continue
        sc, force_bvh, _, _ = RC.scene_and_params(case)
        k1 = zref.RefGBuffer(sc, force_bvh)
        cb = RC.first_cb(case)
        gb_out[f"pick_{case}"] = np.array([[x, y, k1.pick(cb, x, y)] for (x, y) in RC.PICK_PIXELS], np.uint32)
        print("pick", case, sorted(set(int(v) for v in gb_out[f"pick_{case}"][:, 2])))
    np.savez_compressed(gb_path, **gb_out)


if __name__ == "__main__":
    main()
