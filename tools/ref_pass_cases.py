"""The scenarios rendered by the REFERENCE's own shader passes (oracle/zref.py) for tests/golden/ref_pass_*.npz, shared by the generator
(tools/make_ref_pass_goldens.py) and the tests that replay them on the oracle and on the GPU (tests/test_ref_passes.py)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 96, 64


def _scene(kind):
    from zetaray_amd import scene_io
    if kind == "cornell_emissive":
        return scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz")), False, {}
    if kind == "cornell":
        return scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell.npz")), False, {}
    if kind == "materials":           # metal / coat / glass / thin-walled instances, 150 lights
        return scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11), True, dict(cam_pos=(0, 0, -3.5))
    if kind == "materials_lights":    # the same geometry with 1500 lights (presampled sets)
        return scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11), True, dict(cam_pos=(0, 0, -3.5))
    if kind in ("textured", "textured_sky"):
        # base-colour / normal / metallic-roughness / emissive maps (procedural, mip chains), one alpha-tested instance: ray differentials, the
        # TEXTURE_FILTER samplers, TestOpacity.  `textured_sky` has no emissive triangles (sun + sky permutations)
        sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=0 if kind == "textured_sky" else 150, seed=11, open_top=(kind == "textured_sky"))
        TEX_OFFSETS[kind] = scene_io.add_test_textures(sc)
        return sc, True, dict(cam_pos=(0, 0, -3.5))
    raise KeyError(kind)


TEX_OFFSETS = {}      # scene kind -> the four descriptor-table offsets of its texture heap (cbFrameConstants::*MapsDescHeapOffset)


def _params(bounces=None, presample=None, flags_off=0, kind=None, tex_filter=None, spatial_passes=None, flags_on=0, m_max=None, alpha_min=None):
    from zetaray_amd import wire
    p = wire.default_params_di() if kind == "di" else (wire.default_params_sky_di() if kind == "sdi" else wire.default_params())
    if bounces:
        p.max_non_tr_bounces, p.max_glossy_tr_bounces = bounces
    if presample:
        p.presampling, p.num_sample_sets, p.sample_set_size = 1, presample[0], presample[1]
    p.flags &= ~flags_off
    p.flags |= flags_on
    if m_max:
        p.m_max_temporal, p.m_max_spatial = m_max
    if alpha_min is not None:      # the settings UI's value; the constant buffers hold its square (IndirectLighting.cpp:1593-1600, DirectLighting.cpp:404-408)
        p.alpha_min = float(np.float32(alpha_min) * np.float32(alpha_min))
    if tex_filter is not None:
        p.tex_filter = tex_filter
    if spatial_passes is not None:
        p.num_spatial_passes = spatial_passes
    return p


# name -> (scene kind, integrator, frames, params kwargs, moving camera?)
CASES = {
    "k9_cornell_emissive": ("cornell_emissive", "pt", 2, {}, False),
    "k9_materials_rr": ("materials", "pt", 2, dict(bounces=(6, 8)), False),
    "k9_presampled": ("materials_lights", "pt", 2, dict(presample=(32, 128)), False),
    "k9_sun_sky": ("cornell", "pt", 2, {}, False),
    "rpt_cornell_moving": ("cornell_emissive", "rpt", 4, {}, True),
    "rpt_materials_rr": ("materials_lights", "rpt", 3, dict(bounces=(6, 8)), False),
    "rpt_presampled": ("materials_lights", "rpt", 3, dict(presample=(32, 128)), False),
    "rpt_sun_sky": ("cornell", "rpt", 3, {}, False),
    # CB_IND_FLAGS::SORT_TEMPORAL / SORT_SPATIAL off: Reconnect_StC's waves are the 8 x 8 screen blocks (every other ReSTIR PT case runs the
    # reference's default, sorted)
    "rpt_unsorted": ("materials_lights", "rpt", 3, dict(flags_off=(1 << 6) | (1 << 7)), False),
    # m_numSpatialPasses = 2 / 0 (IndirectLighting.cpp:616-621, 860-870, 906): the second search / sort / replay / reconnect round on swapped
    # reservoir sets; no spatial reuse at all (the temporal pass writes the radiance)
    "rpt_two_spatial": ("cornell_emissive", "rpt", 4, dict(spatial_passes=2), True),
    "rpt_two_spatial_materials": ("materials_lights", "rpt", 3, dict(spatial_passes=2, bounces=(6, 8)), False),
    "rpt_no_spatial": ("materials_lights", "rpt", 3, dict(spatial_passes=0), False),
    "rpt_two_spatial_sun_sky": ("cornell", "rpt", 3, dict(spatial_passes=2), False),      # the NEE_EMISSIVE == 0 permutation (component-wise reservoir writes) through two rounds
    # the settings UI's knobs off their defaults (IndirectLighting.cpp:1468-1600 / DirectLighting.cpp:374-410 callbacks; the C++ mirror's setters): 2 / 3
    # bounces, no Russian roulette, M_max 6 / 5, temporal sort off, boiling suppression off, path regularisation ON, Alpha_min 0.2; DI: M_max 12, no extra
    # disocclusion samples, deterministic spatial neighbours, Alpha_min 0.1
    "rpt_tuned": ("materials_lights", "rpt", 3, dict(bounces=(2, 3), m_max=(6, 5), alpha_min=0.2, flags_off=(1 << 3) | (1 << 4) | (1 << 6), flags_on=(1 << 5)), False),
    "di_tuned": ("materials_lights", "di", 3, dict(m_max=(12, 20), alpha_min=0.1, flags_off=(1 << 8) | (1 << 9)), False),
    # ReSTIR GI: CB_IND_FLAGS::STOCHASTIC_MULTI_BOUNCE on, boiling suppression off, M_max 6, 2 / 3 bounces; sun + sky DI: M_max (sky) 8, M_max (sun) 2, Alpha_min 0.2
    "gi_tuned": ("materials_lights", "gi", 4, dict(bounces=(2, 3), m_max=(6, 8), flags_on=(1 << 2), flags_off=(1 << 4)), False),
    "sdi_tuned": ("cornell", "sdi", 4, dict(m_max=(8, 2), alpha_min=0.2), True),
    # depth of field (cbFrameConstants::DoF / LensRadius / FocusDepth; GBufferRT_Inline.hlsl:215-231: the primary ray leaves a lens sample, the depth plane
    # holds t, the motion vector is non-zero on a still frame): K1 + ReSTIR PT with a moving camera, and the textured path tracer (CameraRayUVGradsScale)
    "rpt_dof": ("cornell_emissive", "rpt", 3, {}, True),
    "k9_textured_dof": ("textured", "pt", 2, {}, False),
    # frame accumulation (cbFrameConstants::Accumulate && CameraStatic, NumFramesCameraStatic: ReSTIR_PT/Util.hlsli:141-159, ReSTIR_DI_Temporal.hlsl:274-303,
    # PathTracer.hlsl:205-211): FINAL sums the frames of a standing camera.  (ReSTIR PT and DI only: the path-tracer harness keeps no FINAL between frames;
    # K9's accumulation is the linearity test of tests/test_gpu_parity.py)
    "rpt_accumulate": ("cornell_emissive", "rpt", 4, {}, False),
    "di_accumulate": ("cornell_emissive", "di", 4, {}, False),
    # the "Temporal Resample" / "Spatial Resample" switches of every ReSTIR pass off (TemporalResamplingCallback / SpatialResamplingCallback of the four passes)
    "di_no_spatial": ("cornell_emissive", "di", 3, dict(flags_off=(1 << 1)), True),
    "di_no_reuse": ("materials_lights", "di", 3, dict(flags_off=(1 << 0) | (1 << 1)), False),
    "sdi_no_spatial": ("cornell", "sdi", 3, dict(flags_off=(1 << 1)), True),
    "sdi_no_reuse": ("cornell", "sdi", 2, dict(flags_off=(1 << 0) | (1 << 1)), False),
    "rpt_no_reuse": ("cornell_emissive", "rpt", 3, dict(flags_off=(1 << 0)), True),
    "gi_no_temporal": ("cornell_emissive", "gi", 3, dict(flags_off=(1 << 0)), True),
    "di_textured": ("textured", "di", 3, {}, False),      # emissive maps in the light samples of K5 / K6 (Le_EmissiveTriangle's texture fetch), the G-buffer of a textured scene
    "gi_cornell_moving": ("cornell_emissive", "gi", 4, {}, True),
    "gi_materials_rr": ("materials_lights", "gi", 3, dict(bounces=(6, 8)), False),
    "gi_presampled": ("materials_lights", "gi", 3, dict(presample=(32, 128)), False),
    "gi_sun_sky": ("cornell", "gi", 3, {}, False),
    "di_cornell_moving": ("cornell_emissive", "di", 5, {}, True),
    # (round 6) the reference's USE_HALF_VECTOR_COPY_SHIFT = 1 build of the emissive DI shaders (Params.hlsli:12; 0 in its tree): Alpha_min 0.8 / 0.6 (far above what a renderer would use: coverage) makes the metal (roughness
    # 0.35), glossy (0.15 - 0.25) and coat lobes of the materials scene reuse by half-vector copy; moving camera (temporal Jacobians), spatial pairwise MIS
    "di_half_vector": ("materials_lights", "di", 4, dict(flags_on=(1 << 11), alpha_min=0.8), False),      # (camera path: CB_KW below)
    "di_half_vector_static": ("materials_lights", "di", 3, dict(flags_on=(1 << 11), alpha_min=0.6, m_max=(12, 20)), False),
    "di_materials": ("materials_lights", "di", 3, {}, False),
    "di_presampled": ("materials_lights", "di", 3, dict(presample=(32, 128)), False),
    "sdi_cornell_moving": ("cornell", "sdi", 5, {}, True),
    # dynamic instance: the tall box slides and turns during frames 2-4 and rests afterwards (zr_scene_update_instances: previous
    # acceleration structure + mesh instances bound by the CtT passes / temporal shifts, MoveXk, x_k_in_motion)
    # material textures: the default ANISOTROPIC_4X sampler and the other TEXTURE_FILTER modes (IndirectLighting_Common.h:69-77)
    "k9_textured": ("textured", "pt", 2, {}, False),
    "rpt_textured": ("textured", "rpt", 2, {}, False),
    "gi_textured_trilinear": ("textured", "gi", 2, dict(tex_filter=1), False),
    "k9_textured_mip0": ("textured", "pt", 1, dict(tex_filter=0), False),
    "k9_textured_aniso2": ("textured", "pt", 1, dict(tex_filter=2), False),
    "k9_textured_aniso16": ("textured", "pt", 1, dict(tex_filter=4), False),
    "k9_textured_sky": ("textured_sky", "pt", 1, {}, False),
    "rpt_moving_instance": ("cornell_emissive", "rpt", 6, {}, False),
    # the light quad itself moves (SceneCore::UpdateEmissivePositions): reconnections onto a moving light, light samples reused across frames
    "rpt_moving_light": ("cornell_emissive", "rpt", 5, {}, False),
    "di_moving_light": ("cornell_emissive", "di", 5, {}, False),
    "di_moving_instance": ("cornell_emissive", "di", 6, {}, False),
    "sdi_moving_instance": ("cornell", "sdi", 5, {}, False),
    "gi_moving_instance": ("cornell_emissive", "gi", 5, {}, False),      # ReSTIR GI's temporal pass over a moving box (previous instance buffer, motion vectors of the mover)
    "gi_moving_light": ("cornell_emissive", "gi", 5, {}, False),
}
# per-case edits of the frame constants
CB_EDIT = {"rpt_dof": dict(dof=1, lens_radius=0.05, focus_depth=4.0), "k9_textured_dof": dict(dof=1, lens_radius=0.05, focus_depth=3.0, camera_ray_uv_grads_scale=0.75)}
_ACC = lambda f: dict(accumulate=1, camera_static=1 if f > 1 else 0, num_frames_static=f - 1)      # noqa: E731
# (the DI pass accumulates Le_SkyWithSunDisk at miss pixels, ReSTIR_DI_Temporal.hlsl:276-281, and an emissive-only scene binds no sky-view LUT: its camera stands inside the box)
CB_KW = {"rpt_accumulate": _ACC, "di_accumulate": lambda f: dict(_ACC(f), cam_pos=(0.0, 1.0, -0.9)),
         "di_half_vector": lambda f: dict(cam_pos=(0.04 * max(0, f - 2), 0.01 * max(0, f - 2), -3.5))}      # per-case, per-frame arguments of make_frame_constants
ANIMATED = {"rpt_moving_instance", "di_moving_instance", "sdi_moving_instance", "rpt_moving_light", "di_moving_light", "gi_moving_instance", "gi_moving_light"}
MOVING_LIGHT = {"rpt_moving_light", "di_moving_light", "gi_moving_light"}
RPT_PLANES = ("A", "B", "C", "D", "E", "F", "G", "neighbor", "map_ctn", "map_ntc")      # + the K12 thread maps of the last frame
PLANES = {"rpt": RPT_PLANES, "gi": ("A", "B", "C"), "di": ("A", "B"), "sdi": ("A", "B", "C"), "pt": ()}


# GBufferRT::PickPixel (tests/golden/ref_pass_gbuffer.npz pick_<case>: rows of x, y, picked mesh): frame 1 of these cases, a grid over the W x H target
PICK_CASES = ("rpt_cornell_moving", "rpt_materials_rr", "k9_sun_sky")
PICK_PIXELS = [(x, y) for y in range(3, H, 12) for x in range(2, W, 13)]


def animated_instance(sc):
    """index of the instance the *_moving_instance cases animate: the largest non-emissive mesh that is not a wall (a box)"""
    import numpy as _np
    from zetaray_amd import wire as _w
    cand = [i for i in range(len(sc.instances)) if sc.instance_mask[i] == _w.SUBGROUP_NON_EMISSIVE and sc.instance_num_tris[i] >= 10]
    return cand[-1]


def light_instance(sc):
    """the instance that carries the emissive triangles (the Cornell box's light quad)"""
    return [i for i in range(len(sc.instances)) if sc.instances["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]


class Animator:
    """per-frame instance updates of an animated case: call step(f) before rendering frame f >= 2; returns (instances, instance_to_world)"""

    def __init__(self, sc, light=False):
        import math
        self.sc, self.idx, self.xf, self.math, self.light = sc, (light_instance(sc) if light else animated_instance(sc)), {}, math, light
        self.emissive_update = None      # (first, records) of the last step when the light moves
        self.t0 = sc.instances["translation"][self.idx].copy()

    def step(self, f):
        from zetaray_amd import scene_io
        m = self.math
        if 2 <= f <= 4:
            t = self.t0 + np.array([0.06 * (f - 1), 0.0, 0.03 * (f - 1)], np.float32)
            a = 0.15 * (f - 1)
            q = np.array([0, m.sin(a / 2), 0, m.cos(a / 2)], np.float32)
            if self.light:
                inst, xw, first, tris = scene_io.move_emissive_instance(self.sc, self.idx, translation=t, rotation=q, xform_of=self.xf)
                self.emissive_update = (first, tris.copy())
                return inst, xw
            return scene_io.move_instance(self.sc, self.idx, translation=t, rotation=q, xform_of=self.xf)
        self.emissive_update = None
        return scene_io.move_instance(self.sc, self.idx, xform_of=self.xf)

    def apply(self, f, *scenes):
        """step(f) + hand the updates to every scene object (update_emissives before update_instances)"""
        inst, xw = self.step(f)
        for q in scenes:
            if self.emissive_update is not None:
                q.update_emissives(self.emissive_update[1], self.emissive_update[0])
            q.update_instances(inst, xw)


def frames_of(case):
    """yields (frame number, cbFrameConstants) with the previous-frame camera chained like a renderer does"""
    from zetaray_amd import scene_io
    kind, _, n, _, moving = CASES[case]
    sc, _, cam = _scene(kind)
    prev = None
    for f in range(1, n + 1):
        kw = dict(cam)
        if moving:
            kw["cam_pos"] = (0.05 * f, 1.2, -4.043 + 0.02 * f)
        kw.update(CB_KW[case](f) if case in CB_KW else {})
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), **kw)
        if kind in TEX_OFFSETS:
            scene_io.set_texture_heap_offsets(cb, TEX_OFFSETS[kind])
        for k, v in CB_EDIT.get(case, {}).items():
            cb[k] = v
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        yield f, cb


def first_cb(case):
    """frame 1's constants: what an OracleScene needs at construction to latch the texture-table offsets (the power estimate of textured
    emissive triangles -- K2, hence the alias table -- samples the emissive maps through them)"""
    return next(frames_of(case))[1]


def scene_and_params(case):
    kind, integ, n, pk, _ = CASES[case]
    sc, force_bvh, _ = _scene(kind)
    return sc, force_bvh, integ, _params(kind=integ, **pk)
