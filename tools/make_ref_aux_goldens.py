"""tests/golden/ref_aux.npz <- the REFERENCE's auxiliary shaders (EstimateTriEmissivePower, PresampleEmissives, BuildLightVoxelGrid,
SkyViewLUT, ReSTIR_GI_LVG, Compositing, FireflyFilter, TAA) compiled as C++ (oracle/_ref/libzref_aux.so, libzref_gi_e1l.so, libzref_k1.so;
built from /root/reference by oracle/_ref.mk) on the seeded inputs of tools/aux_cases.py.
Run in the build container:  python tools/make_ref_aux_goldens.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import aux_cases as ac  # noqa: E402
from oracle import zref, zro  # noqa: E402


def prelighting(out):
    sc, force, cb = ac.textured_scene()
    ra = zref.RefAux(sc, force)
    power = ra.estimate_power(cb)
    out["k2_textured/power"] = power
    # the alias table between K2 and K3 is the reference's CPU code (pinned in tests/test_ref_pins.py); here: the oracle's equal restatement
    ra.set_alias_table(zro.alias_table_build(power))
    cb3 = cb.copy()
    cb3["frame_num"] = ac.K3_FRAME
    out["k3_textured/sets"] = ra.presample(cb3, *ac.K3_SETS)
    sc4, force4 = ac.lvg_scene()
    r4 = zref.RefAux(sc4, force4)
    cbp = ac.lvg_frame(sc4, 1)
    p4 = r4.estimate_power(cbp)
    out["k2_lights/power"] = p4
    alias = zro.alias_table_build(p4)
    r4.set_alias_table(alias)
    rg = zref.RefGBuffer(sc4, force4)
    rgi = zref.RefRestirGI(sc4, ac.GI_W, ac.GI_H, presampling=True, force_bvh=force4, lvg=True)
    rgi.set_alias_table(alias)
    prm = ac.lvg_params()
    for f in range(1, ac.GI_FRAMES + 1):
        cb = ac.lvg_frame(sc4, f)
        sets = r4.presample(cb, prm.num_sample_sets, prm.sample_set_size)
        grid = r4.build_lvg(cb, ac.LVG_DIM, ac.LVG_EXT, ac.LVG_OFF)
        rgi.set_sample_sets(sets, prm.num_sample_sets, prm.sample_set_size)
        rgi.set_lvg(grid, ac.LVG_DIM, ac.LVG_EXT, ac.LVG_OFF)
        out[f"gi_lvg/final{f}"] = rgi.render(cb, prm, rg.render(cb))
        if f == 2:
            out["k4/grid"] = grid
            out["k3_lights/sets"] = sets
    for nm in "ABC":
        out[f"gi_lvg/{nm}"] = rgi.plane(nm)


def sky(out):
    ra = zref.RefAux()
    for name, cb in ac.sky_frames().items():
        out[f"{name}/lut"] = ra.sky_lut(cb)


def post(out):
    for kind in ("sky", "emissive"):
        sc, force = ac.post_scene(kind)
        ra, rg = zref.RefAux(sc, force), zref.RefGBuffer(sc, force)
        lutgen = zref.RefAux()
        prev, comp = None, None
        hist = np.zeros((ac.H, ac.W, 4), np.uint16)
        for f in range(1, ac.POST_FRAMES + 1):
            cb = ac.chain_prev(ac.post_frame(kind, sc, f), prev)
            prev = cb.copy()
            ra.set_sky_lut(lutgen.sky_lut(cb))        # the Sky pass runs for every scene (miss pixels of an emissive scene show the sky too)
            arrays, planes = rg.render(cb)
            sdi, edi, ind, flags = ac.post_terms(kind, f)
            comp = ra.composite(cb, planes, sky_di=sdi, emissive_di=edi, indirect=ind, flags=flags, firefly=False, out=comp)
            out[f"post_{kind}/comp{f}"] = comp
            out[f"post_{kind}/firefly{f}"] = ra.composite(cb, planes, sky_di=sdi, emissive_di=edi, indirect=ind, flags=flags, firefly=True, out=None)
            hist = ra.taa(cb, comp, arrays[7].reshape(ac.H, ac.W), arrays[3].reshape(ac.H, ac.W), hist, ac.TAA_BLEND, f not in ac.TAA_INVALID)
            out[f"post_{kind}/taa{f}"] = hist[..., :3].copy()


def main():
    out = {}
    prelighting(out)
    sky(out)
    post(out)
    dst = os.path.join(ROOT, "tests", "golden", "ref_aux.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
