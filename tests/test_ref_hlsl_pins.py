"""Pins the oracle AND the product's HIP stage functions to the REFERENCE's own shader code.

The reference's HLSL-2021 headers (Source/ZetaRenderPass/Common/{Math,Sampling,RT,BSDF,BSDFSampling}.hlsli, Material.h) are compiled
as C++ in place by oracle/_ref.mk (oracle/ref_hlsl/hlsl2cpp.py rewrites only surface syntax; hlsl_shim.h maps HLSL intrinsics onto the
ABI's arithmetic contract, include/zr_detmath.h).  Function-level probes (oracle/zro_kat_layout.h) then run three ways:

    reference code (oracle/_ref/libzref_hlsl.so; stored in tests/golden/ref_hlsl_kat.npz by tools/make_ref_hlsl_goldens.py)
    the oracle restatement (oracle/zro_kat.h)
    the HIP stage functions of zetaray_amd/csrc compiled for the host (tests/hostexec/hx_kat.h)

and must agree bit for bit on every column the oracle / product implement.  A misreading of the HLSL -- the failure mode a twin
transcription cannot see -- shows up here.  What this cannot pin: vendor transcendentals, texture filtering hardware and the
driver's traversal; those are defined by the ABI for all three sides.
Also: offsetof / sizeof of zr_wire.h (through zr_wire_layout) and of wire.py's dtypes against the reference's RtCommon.h, Material.h,
Vertex.h, FrameConstants.h compiled in place (tests/golden/ref_layout.txt)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kat_inputs as K  # noqa: E402
from oracle import zro  # noqa: E402
from tests.hostexec import zhx  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "ref_hlsl_kat.npz"))
N_GOLDEN = 4096
RHO = np.fromfile(os.path.join(ROOT, "zetaray_amd", "assets", "rho_lut_u16.bin"), np.uint16)
RHO_DIM = np.array([64, 32, 16], np.uint32)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libzref_hlsl.so")

# columns of reference functions that are NOT on the hot path and therefore exist neither in the oracle nor in the product
# (Sampling::UniformSampleHemisphere / UniformSampleDisk / UniformSampleSphere, RT::BalanceHeuristic<T>, BSDF::GGXReflectance_Metal)
NOT_IMPLEMENTED = {"sampling": [0, 1, 2, 3, 12, 13, 16, 17, 18], "math": [], "rt": [6], "bsdf": [55, 56, 57]}
LOBE_COLS = {32: 25, 50: 46}      # BSDFSample.lobe is left unset by the reference when the sample failed (pdf column == 0)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _run(lib, prefix, fam, x):
    out = np.zeros((len(x), K.FAMILIES[fam][1]), np.float32)
    if fam == "bsdf" and prefix != "zrefh_kat_":
        getattr(lib, prefix + fam)(_p(RHO), _p(RHO_DIM), _p(x), _p(out), len(x))
    else:
        getattr(lib, prefix + fam)(_p(x), _p(out), len(x))
    return out.view(np.uint32)


def _compare(fam, want, got, who):
    cols = [c for c in range(want.shape[1]) if c not in NOT_IMPLEMENTED[fam]]
    eq = want == got
    if fam == "bsdf":
        for lobe_col, pdf_col in LOBE_COLS.items():
            eq[:, lobe_col] |= want[:, pdf_col].view(np.float32) == 0
    bad = {c: int((~eq[:, c]).sum()) for c in cols if not eq[:, c].all()}
    assert not bad, f"{who} differs from the reference's own code in {fam} columns {bad}"


@pytest.mark.parametrize("fam", list(K.FAMILIES))
def test_oracle_matches_reference_shader_code(fam):
    x = np.ascontiguousarray(K.FAMILIES[fam][0](N_GOLDEN))
    _compare(fam, GOLD[fam], _run(zro.lib(), "zro_kat2_", fam, x), "oracle")


@pytest.mark.parametrize("fam", list(K.FAMILIES))
def test_hip_stage_functions_match_reference_shader_code(fam):
    """the product's device functions (zr_dev_math.h, zr_dev_bsdf.h, zr_rpt.h), compiled for the host by tests/hostexec"""
    x = np.ascontiguousarray(K.FAMILIES[fam][0](N_GOLDEN))
    _compare(fam, GOLD[fam], _run(zhx.lib(), "zhx_kat_", fam, x), "HIP stage functions")


def test_probe_inputs_exercise_every_branch():
    """the BSDF probe must hit every lobe, reflection and transmission, TIR, invalid configurations, coated and uncoated surfaces"""
    g = GOLD["bsdf"]
    ok = g[:, 25].view(np.float32) > 0
    assert set(np.unique(g[ok, 32])) == {0, 1, 2, 3, 4}            # DIFFUSE_R, DIFFUSE_T, GLOSSY_R, GLOSSY_T, COAT all sampled
    assert (g[:, 21] == 1).sum() > 10                              # total internal reflection
    assert 0.2 < (g[:, 14] & 1).mean() < 0.8                       # valid and invalid wi
    assert ((g[:, 14] & 2) == 0).sum() > 100                       # transmission configurations
    assert (np.abs(g[:, 15:18].view(np.float32)).sum(axis=1) > 0).mean() > 0.4
    assert (g[:, 34].view(np.float32) > 0).mean() > 0.4            # BSDFSamplerPdf
    assert (g[:, 36].view(np.float32) > 0).mean() > 0.8            # EvalBSDFSampler replay


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no /root/reference on this machine): the committed goldens are used instead")
@pytest.mark.parametrize("fam", list(K.FAMILIES))
def test_live_reference_build_20000_fresh_inputs(fam):
    """with the reference present: fresh seeds, 20 000 inputs per family, reference code vs oracle vs HIP stage functions"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_hlsl_goldens as M
    L = M.ref_lib()
    x = np.ascontiguousarray(K.FAMILIES[fam][0](20000, seed=12345))
    want = M.run_ref(L, fam, x).view(np.uint32)
    if fam == "sampling":   # the stored goldens are what the reference build produces today
        x0 = np.ascontiguousarray(K.FAMILIES[fam][0](N_GOLDEN))
        assert np.array_equal(M.run_ref(L, fam, x0).view(np.uint32), GOLD[fam])
    _compare(fam, want, _run(zro.lib(), "zro_kat2_", fam, x), "oracle")
    _compare(fam, want, _run(zhx.lib(), "zhx_kat_", fam, x), "HIP stage functions")


# ------------------------------------------------------------------ layout pins
def _layout_lines(text):
    return {ln.split()[0]: tuple(int(v) for v in ln.split()[1:]) for ln in text.strip().splitlines()}


def test_wire_structs_match_reference_headers():
    """offsetof / sizeof of every field of zr_wire.h == the reference's own headers (RtCommon.h:47-131,302-332, Material.h:419-427,
    Vertex.h:8-14, FrameConstants.h:10-78) compiled in place (golden text; the live comparison runs when oracle/_ref exists)"""
    from zetaray_amd import api
    L = api.lib()
    L.zr_wire_layout.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 16)
    assert L.zr_wire_layout(buf, len(buf)) > 0
    mine = _layout_lines(buf.value.decode())
    ref = _layout_lines(open(os.path.join(ROOT, "tests", "golden", "ref_layout.txt")).read())
    assert len(ref) >= 112
    assert mine == ref, {k: (mine.get(k), ref.get(k)) for k in set(mine) | set(ref) if mine.get(k) != ref.get(k)}
    so = os.path.join(ROOT, "oracle", "_ref", "libzref.so")
    if os.path.exists(so):
        R = C.CDLL(so)
        b2 = C.create_string_buffer(1 << 16)
        assert R.zref_layout(b2, len(b2)) > 0
        assert _layout_lines(b2.value.decode()) == ref, "tests/golden/ref_layout.txt is stale: run tools/make_ref_hlsl_goldens.py"


def test_numpy_wire_dtypes_match_reference_headers():
    from zetaray_amd import wire
    ref = _layout_lines(open(os.path.join(ROOT, "tests", "golden", "ref_layout.txt")).read())
    cb = {k.split(".")[1]: v for k, v in ref.items() if k.startswith("cbFrameConstants.")}
    assert ref["cbFrameConstants"] == (wire.FRAME_CONSTANTS.itemsize,)
    # same declaration order in both: compare offset sequences
    offs = [wire.FRAME_CONSTANTS.fields[n][1] for n in wire.FRAME_CONSTANTS.names]
    assert offs == [v[0] for v in cb.values()]
    for name, dt in (("MeshInstance", wire.MESH_INSTANCE), ("EmissiveTriangle", wire.EMISSIVE_TRI), ("Material", wire.MATERIAL),
                     ("Vertex", wire.VERTEX), ("EmissiveLumenAliasTableEntry", wire.ALIAS_ENTRY), ("PresampledEmissiveTriangle", wire.PRESAMPLED_TRI),
                     ("VoxelSample", wire.VOXEL_SAMPLE)):
        assert ref[name] == (dt.itemsize,), name
        want = [v[0] for k, v in ref.items() if k.startswith(name + ".")]
        got = [dt.fields[n][1] for n in dt.names]
        assert got == want, (name, got, want)
