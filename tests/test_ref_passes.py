"""Parity against the REFERENCE's own shader passes.

tests/golden/ref_pass_*.npz hold what the reference's HLSL passes -- GBufferRT_Inline.hlsl (K1), PathTracer.hlsl (K9) and the ten ReSTIR PT
shaders (K11, K13-K16) with the host sequence of IndirectLighting::RenderReSTIR_PT, ReSTIR_GI.hlsl (K10), ReSTIR_DI_Temporal / _Spatial.hlsl
(K5 / K6) and SkyDI_Temporal / _Spatial.hlsl (K7 / K8) with their host sequences -- produce when compiled as C++ in place from
/root/reference (oracle/ref_hlsl/, `make -C oracle -f _ref.mk`, tools/make_ref_pass_goldens.py) over the ABI's definitions of what the
reference leaves to driver and hardware (traversal / intersection, transcendentals, texture filtering).

  * not gpu:  the oracle reproduces them bit for bit -- G-buffer planes, FINAL of every frame, the 7 reservoir planes + neighbour plane;
              where oracle/_ref is built (the build container) the reference passes are also re-run live against the stored files
  * gpu:      the HIP product through the C-ABI reproduces the same files bit for bit (tolerance 0)

Scenarios (tools/ref_pass_cases.py): Cornell with emissive quad lights (static and moving camera), Cornell with sun + sky, a 3 000-triangle
scene with metal / coat / glass / thin-walled materials, 6 / 8 bounces and Russian roulette, presampled light sets."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_pass_cases as RC  # noqa: E402
from zetaray_amd import wire  # noqa: E402

GB = np.load(os.path.join(ROOT, "tests", "golden", "ref_pass_gbuffer.npz"))


def _gold(case):
    return np.load(os.path.join(ROOT, "tests", "golden", f"ref_pass_{case}.npz"))


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8).ravel(), np.ascontiguousarray(b).view(np.uint8).ravel())


def _plane_a(x):
    """plane A as bytes (k | M << 4, lobes | lt_k << 6, lt_k+1 | x_k_in_motion << 2); the 4th byte is unused"""
    return np.ascontiguousarray(x).view(np.uint8).reshape(RC.H, RC.W, 4)[..., :3]


@pytest.mark.parametrize("case", list(RC.CASES))
def test_oracle_reproduces_reference_passes(case):
    from oracle import zro
    sc, force_bvh, integ, prm = RC.scene_and_params(case)
    g = _gold(case)
    o = zro.OracleScene(sc, force_bvh=force_bvh, cb=RC.first_cb(case))
    rpt = {"rpt": zro.OracleRPT, "gi": zro.OracleRGI, "di": zro.OracleRDI, "sdi": zro.OracleSDI}[integ](o, RC.W, RC.H) if integ != "pt" else None
    anim = RC.Animator(sc, light=case in RC.MOVING_LIGHT) if case in RC.ANIMATED else None
    for f, cb in RC.frames_of(case):
        if anim is not None and f >= 2:
            anim.apply(f, o)
        if len(sc.emissives) == 0:
            o.sky_lut(cb, 256, 128)
        if prm.presampling:
            o.presample(f, prm.num_sample_sets, prm.sample_set_size)
        arrays, planes = o.gbuffer(cb)
        if f == (3 if anim is not None else 1):
            for n, a in zip(wire.GB_PLANE_NAMES, arrays):
                assert _same(a, GB[f"{case}_{n}"]), f"K1 plane {n} differs from the reference shader's"
        got = o.pathtrace(cb, planes, prm)[0] if integ == "pt" else rpt.render(cb, prm, gb=(arrays, planes))
        assert _same(got, g[f"final_{f}"]), f"frame {f}: FINAL differs from the reference shaders'"
    for nm in RC.PLANES[integ]:
        a, b = rpt.plane(nm), g["plane_" + nm]
        if nm == "A" and integ == "rpt":
            a, b = _plane_a(a), _plane_a(b)
        assert _same(a, b), f"reservoir plane {nm} differs from the reference shaders'"


def _zref():
    from oracle import zref
    if not zref.available():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine): the committed reference outputs are used instead")
    return zref


@pytest.mark.parametrize("case", ["k9_materials_rr", "rpt_cornell_moving", "rpt_sun_sky", "gi_materials_rr", "di_materials", "sdi_cornell_moving",
                                  "rpt_moving_instance", "di_moving_instance", "rpt_two_spatial", "rpt_no_spatial", "di_half_vector", "di_half_vector_static"])
def test_live_reference_passes_match_stored_outputs(case):
    """re-runs the reference's compiled shaders: guards the stored files against a stale build"""
    zref = _zref()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_ref_pass_goldens as M
    from oracle import zro
    sc, force_bvh, integ, prm = RC.scene_and_params(case)
    g = _gold(case)
    o = zro.OracleScene(sc, force_bvh=force_bvh, cb=RC.first_cb(case))
    k1 = zref.RefGBuffer(sc, force_bvh)
    ref = M.make_ref(zref, sc, integ, prm, force_bvh)
    anim = RC.Animator(sc, light=case in RC.MOVING_LIGHT) if case in RC.ANIMATED else None
    for f, cb in RC.frames_of(case):
        if anim is not None and f >= 2:
            anim.apply(f, o, k1, ref)
        M.prepare(ref, o, sc, cb, f, prm)
        arrays, planes = k1.render(cb)
        got = ref.render(cb, planes, prm) if integ == "pt" else ref.render(cb, prm, (arrays, planes))
        assert _same(got, g[f"final_{f}"]), f"frame {f}"


def test_live_reference_restir_pt_larger_frame():
    """live only: 150 x 90 (partial 32 x 32 sort tiles on both boundaries), 5 frames, camera that starts moving at frame 3, reference shaders
    (incl. the four ReSTIR_PT_Sort dispatches and the map-driven reconnect passes) vs oracle incl. every reservoir plane and both thread maps"""
    zref = _zref()
    from oracle import zro
    from zetaray_amd import scene_io
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    w, h = 150, 90
    o = zro.OracleScene(sc)
    k1, ref, orpt = zref.RefGBuffer(sc), zref.RefRestirPT(sc, w, h), zro.OracleRPT(o, w, h)
    ref.set_alias_table(o.alias)
    prm = wire.default_params()
    prev = None
    for f in range(1, 6):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0.07 * max(0, f - 2), 1.2, -4.043))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        arrays, planes = k1.render(cb)
        oa, op = o.gbuffer(cb)
        for a, b in zip(arrays, oa):
            assert _same(a, b)
        assert _same(ref.render(cb, prm, (arrays, planes)), orpt.render(cb, prm, gb=(oa, op))), f"frame {f}"
        for nm in RC.RPT_PLANES:
            a, b = ref.plane(nm), orpt.plane(nm)
            if nm == "A":
                a, b = _plane_a(a.reshape(h, w, 4)[:RC.H, :RC.W].copy()), _plane_a(np.ascontiguousarray(b).view(np.uint8).reshape(h, w, 4)[:RC.H, :RC.W].copy())
            assert _same(a, b), f"frame {f}: plane {nm}"


# ------------------------------------------------------------------ the HIP product against the reference's outputs
@pytest.mark.gpu
@pytest.mark.parametrize("case", list(RC.CASES))
def test_hip_path_reproduces_reference_passes(case):
    from zetaray_amd import api
    sc, _, integ, prm = RC.scene_and_params(case)
    g = _gold(case)
    integrator = {"rpt": api.INTEGRATOR_RESTIR_PT, "gi": api.INTEGRATOR_RESTIR_GI}.get(integ, api.INTEGRATOR_PATH_TRACING)
    if integ in ("di", "sdi"):
        ip = wire.default_params()
        ip.presampling, ip.num_sample_sets, ip.sample_set_size = prm.presampling, prm.num_sample_sets, prm.sample_set_size
        r = api.Renderer(sc, RC.W, RC.H, params=ip)
        p = r.enable_direct(prm) if integ == "di" else r.enable_sky_direct(prm)
        r.skip_indirect = True
        names = {"A": "di_A", "B": "di_B"} if integ == "di" else {"A": "sdi_A", "B": "sdi_B", "C": "sdi_C"}
    else:
        r = api.Renderer(sc, RC.W, RC.H, params=prm, integrator=integrator)
        p = r.p_indirect
        names = {n: n for n in RC.RPT_PLANES} if integ == "rpt" else {"A": "gi_A", "B": "gi_B", "C": "gi_C"}
    anim = RC.Animator(sc, light=case in RC.MOVING_LIGHT) if case in RC.ANIMATED else None
    for f, cb in RC.frames_of(case):
        if anim is not None and f >= 2:
            anim.apply(f, r.scene)
        r.render_frame(cb)
        if f == (3 if anim is not None else 1):
            planes, _ = r.gbuffer.download()
            for n, a in zip(wire.GB_PLANE_NAMES, planes):
                assert _same(a, GB[f"{case}_{n}"]), f"K1 plane {n} differs from the reference shader's"
        assert _same(p.download(), g[f"final_{f}"]), f"frame {f}: FINAL differs from the reference shaders'"
    for nm in RC.PLANES[integ]:
        a, b = p.download_plane(names[nm]), g["plane_" + nm]
        if nm == "A" and integ == "rpt":
            a, b = _plane_a(a), _plane_a(b)
        assert _same(a, b), f"reservoir plane {nm} differs from the reference shaders'"


# ------------------------------------------------------------------ the HIP stage functions, executed on the host, against the reference's outputs
@pytest.mark.parametrize("case", ["rpt_cornell_moving", "rpt_moving_instance", "di_moving_instance", "sdi_moving_instance", "gi_cornell_moving", "rpt_sun_sky",
                                  "rpt_moving_light", "di_moving_light", "rpt_two_spatial", "rpt_two_spatial_materials", "rpt_no_spatial", "rpt_two_spatial_sun_sky",
                                  "di_half_vector", "di_half_vector_static", "di_materials"])
def test_hip_stage_functions_on_host_reproduce_reference_passes(case):
    """the product's device code (zr_stages.h, zr_rpt.h, zr_rdi.h, zr_sdi.h, zr_rgi.h) compiled for the host by tests/hostexec and run serially:
    catches a divergence from the reference's shaders without a GPU, incl. the dynamic-instance paths (previous BVH / mesh instances, MoveXk)"""
    from oracle import zro
    from tests.hostexec import zhx
    sc, force_bvh, integ, prm = RC.scene_and_params(case)
    g = _gold(case)
    o = zro.OracleScene(sc, force_bvh=force_bvh, cb=RC.first_cb(case))                 # only for the scene-level inputs (alias table, presampled sets, sky LUT)
    hx = zhx.HostExecScene(sc, alias=o.alias if len(sc.emissives) else None)
    run = {"rpt": zhx.HostExecRPT, "gi": zhx.HostExecRGI, "di": zhx.HostExecRDI, "sdi": zhx.HostExecSDI}[integ](hx, RC.W, RC.H)
    anim = RC.Animator(sc, light=case in RC.MOVING_LIGHT) if case in RC.ANIMATED else None
    for f, cb in RC.frames_of(case):
        if anim is not None and f >= 2:
            anim.apply(f, hx)
        if len(sc.emissives) == 0:
            hx.sky_lut(cb, 256, 128)
        gb = hx.gbuffer(cb)
        if f == (3 if anim is not None else 1):
            for n, a in zip(wire.GB_PLANE_NAMES, gb[0]):
                assert _same(a, GB[f"{case}_{n}"]), f"K1 plane {n} differs from the reference shader's"
        assert _same(run.render(cb, prm, gb=gb), g[f"final_{f}"]), f"frame {f}: FINAL differs from the reference shaders'"
    for nm in RC.PLANES[integ]:
        a, b = run.plane(nm), g["plane_" + nm]
        if nm == "A" and integ == "rpt":
            a, b = _plane_a(a), _plane_a(b)
        assert _same(a, b), f"reservoir plane {nm} differs from the reference shaders'"


@pytest.mark.parametrize("case", list(RC.PICK_CASES))
def test_oracle_picks_what_the_reference_shader_picks(case):
    """GBufferRT::PickPixel: the oracle's K1 reports the mesh the reference's GBufferRT_Inline.hlsl wrote to g_pick[0] (UINT32_MAX on a miss)"""
    from oracle import zro
    sc, force_bvh, _, _ = RC.scene_and_params(case)
    o = zro.OracleScene(sc, force_bvh=force_bvh, cb=RC.first_cb(case))
    cb = RC.first_cb(case)
    rows = GB[f"pick_{case}"]
    assert len(rows) == len(RC.PICK_PIXELS) and len(set(int(v) for v in rows[:, 2])) >= 3
    for x, y, want in rows:
        assert o.pick(cb, int(x), int(y)) == int(want), (int(x), int(y))


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(RC.PICK_CASES))
def test_hip_path_picks_what_the_reference_shader_picks(case):
    from zetaray_amd import api
    sc, _, _, prm = RC.scene_and_params(case)
    r = api.Renderer(sc, RC.W, RC.H, params=prm)
    cb = RC.first_cb(case)
    for x, y, want in GB[f"pick_{case}"]:
        r.p_gbuffer.pick_pixel(int(x), int(y))
        r.p_gbuffer.render(cb, r.scene, r.gbuffer)
        assert r.p_gbuffer.read_pick() == int(want), (int(x), int(y))
