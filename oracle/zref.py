"""ORACLE tooling -- test infrastructure only.

ctypes wrappers over the REFERENCE's own shader passes compiled as C++ (oracle/_ref/libzref_k*.so, built by `make -C oracle -f _ref.mk`
from /root/reference; see oracle/ref_hlsl/).  Only available where /root/reference exists (the build container): tests use it live
there and fall back to the goldens it generated (tests/golden/ref_pass_*.npz, tools/make_ref_pass_goldens.py) elsewhere."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def available():
    return os.path.exists(os.path.join(HERE, "_ref", "libzref_k1.so"))


def _lib(name):
    if name not in _libs:
        L = C.CDLL(os.path.join(HERE, "_ref", f"libzref_{name}.so"))
        L.zrefp_scene_create.restype = C.c_void_p
        L.zrefp_scene_create.argtypes = [C.c_void_p, C.c_int]
        L.zrefp_scene_destroy.argtypes = [C.c_void_p]
        L.zrefp_scene_set_alias_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zrefp_scene_set_sky_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.zrefp_scene_set_sample_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _libs[name] = L
    return _libs[name]


class RefPass:
    """one compiled shader permutation + its own copy of the scene"""

    def __init__(self, name, scene, force_bvh=False):
        self.L = _lib(name)
        self._scene = scene
        self._desc = scene.desc()
        self.h = self.L.zrefp_scene_create(C.addressof(self._desc), int(force_bvh))
        self._keep = []

    def __del__(self):
        try:
            self.L.zrefp_scene_destroy(self.h)
        except Exception:
            pass

    def update_instances(self, instances, instance_to_world):
        """per-frame MeshInstance records + object-to-world matrices; the scene as it was becomes the previous one"""
        self.L.zrefp_scene_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        i, x = np.ascontiguousarray(instances), np.ascontiguousarray(instance_to_world, np.float32)
        assert self.L.zrefp_scene_update_instances(self.h, i.ctypes.data, x.ctypes.data, len(i)) == 0

    def update_emissives(self, triangles, first=0):
        """new EmissiveTriangle records of instances that moved (one emissive buffer: the previous-frame scene view sees them too)"""
        self.L.zrefp_scene_update_emissives.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        t = np.ascontiguousarray(triangles)
        assert self.L.zrefp_scene_update_emissives(self.h, t.ctypes.data, first, len(t)) == 0

    def set_alias_table(self, entries):
        e = np.ascontiguousarray(entries)
        self.L.zrefp_scene_set_alias_table(self.h, e.ctypes.data, len(e))

    def set_sky_lut(self, texels):
        t = np.ascontiguousarray(texels, np.uint32)
        self.L.zrefp_scene_set_sky_lut(self.h, t.ctypes.data, t.shape[1], t.shape[0])

    def set_lvg(self, grid, dim, extents, offset_y):
        """K4's output (BuildLightVoxelGrid.hlsl) for the USE_LVG permutation of ReSTIR GI"""
        self.L.zrefp_scene_set_lvg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        g, d, e = np.ascontiguousarray(grid), np.array(dim, np.uint32), np.array(extents, np.float32)
        self.L.zrefp_scene_set_lvg(self.h, g.ctypes.data, d.ctypes.data, e.ctypes.data, float(offset_y))

    def set_sample_sets(self, sets, num_sets, set_size):
        s = np.ascontiguousarray(sets)
        self.L.zrefp_scene_set_sample_sets(self.h, s.ctypes.data, num_sets, set_size)


class RefGBuffer(RefPass):
    """K1: GBufferRT_Inline.hlsl"""

    def __init__(self, scene, force_bvh=False):
        super().__init__("k1", scene, force_bvh)
        self.L.zrefp_gbuffer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]

    def render(self, cb):
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        self.L.zrefp_gbuffer_render(self.h, cbb.ctypes.data, C.addressof(planes))
        return arrays, planes


    def pick(self, cb, x, y):
        """GBufferRT::PickPixel(x, y) + one dispatch of the reference's K1: what the shader wrote to g_pick[0]"""
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        out = C.c_uint32(0)
        self.L.zrefp_gbuffer_render_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        self.L.zrefp_gbuffer_render_pick(self.h, cbb.ctypes.data, C.addressof(planes), int(x), int(y), C.byref(out))
        return int(out.value)


class RefPathTracer(RefPass):
    """K9: PathTracer.hlsl; the permutation follows the scene / params like IndirectLighting.h:251-300 picks the .cso"""

    def __init__(self, scene, presampling=False, force_bvh=False):
        name = "k9_e0" if len(scene.emissives) == 0 else ("k9_e1p" if presampling else "k9_e1")
        super().__init__(name, scene, force_bvh)
        self.L.zrefp_pathtrace_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def render(self, cb, gb_planes, params, final=None):
        w, h = int(cb["render_width"]), int(cb["render_height"])
        out = np.zeros((h, w, 4), np.float32) if final is None else np.ascontiguousarray(final.copy())
        cbb = np.ascontiguousarray(cb)
        self.L.zrefp_pathtrace_render(self.h, cbb.ctypes.data, C.addressof(gb_planes), C.addressof(params), out.ctypes.data)
        return out


class RefRestirPT(RefPass):
    """K11 + K13-K16: the reference's ReSTIR PT shaders driven by the restated host sequence (oracle/ref_hlsl/ref_rpt_host.cpp)"""
    PLANES = {"A": (0, np.uint8, 4), "B": (1, np.float32, 2), "C": (2, np.uint32, 4), "D": (3, np.uint32, 4), "E": (4, np.uint16, 1),
              "F": (5, np.float32, 2), "G": (6, np.uint32, 2), "neighbor": (7, np.uint8, 2), "target": (8, np.float32, 4),
              "map_ctn": (10, np.uint16, 1), "map_ntc": (11, np.uint16, 1)}      # K12 thread maps (ReSTIR_PT_Sort.hlsl)

    def __init__(self, scene, w, h, presampling=False, force_bvh=False):
        name = "rpt_e0" if len(scene.emissives) == 0 else ("rpt_e1p" if presampling else "rpt_e1")
        super().__init__(name, scene, force_bvh)
        L = self.L
        L.zrefp_rpt_create.restype = C.c_void_p
        L.zrefp_rpt_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zrefp_rpt_destroy.argtypes = [C.c_void_p]
        L.zrefp_rpt_reset_temporal.argtypes = [C.c_void_p]
        L.zrefp_rpt_render.argtypes = [C.c_void_p] * 7
        L.zrefp_rpt_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.w, self.h_ = w, h
        self.st = L.zrefp_rpt_create(w, h)
        self.prev = None

    def reset_temporal(self):
        self.L.zrefp_rpt_reset_temporal(self.st)

    def render(self, cb, params, gb):
        """gb = (arrays, planes) of THIS frame's G-buffer; the previous frame's is remembered"""
        out = np.zeros((self.h_, self.w, 4), np.float32)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        self.L.zrefp_rpt_render(self.h, self.st, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), out.ctypes.data)
        self.prev = gb
        return out

    def plane(self, name, which=0):
        idx, dt, ch = self.PLANES[name]
        buf = np.zeros((self.h_, self.w, ch), dt)
        self.L.zrefp_rpt_read_plane(self.st, which, idx, buf.ctypes.data)
        return buf


class RefRestirGI(RefPass):
    """K10: ReSTIR_GI.hlsl + the restated host (oracle/ref_hlsl/ref_gi_host.cpp)"""
    PLANES = {"A": (0, np.float32, 4), "B": (1, np.uint16, 4), "C": (2, np.float32, 4)}

    def __init__(self, scene, w, h, presampling=False, force_bvh=False, lvg=False):
        name = "gi_e0" if len(scene.emissives) == 0 else (("gi_e1l" if lvg else "gi_e1p") if presampling else "gi_e1")
        super().__init__(name, scene, force_bvh)
        L = self.L
        L.zrefp_gi_create.restype = C.c_void_p
        L.zrefp_gi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zrefp_gi_render.argtypes = [C.c_void_p] * 7
        L.zrefp_gi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.w, self.h_ = w, h
        self.st = L.zrefp_gi_create(w, h)
        self.prev = None

    def render(self, cb, params, gb):
        out = np.zeros((self.h_, self.w, 4), np.float32)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        self.L.zrefp_gi_render(self.h, self.st, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), out.ctypes.data)
        self.prev = gb
        return out

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        buf = np.zeros((self.h_, self.w, ch), dt)
        self.L.zrefp_gi_read_plane(self.st, idx, buf.ctypes.data)
        return buf


class RefDirect(RefPass):
    """K5 / K6 (emissive ReSTIR DI) or K7 / K8 (sun + sky ReSTIR DI): the reference's DI shaders + restated host (oracle/ref_hlsl/ref_di_host.cpp)"""

    def __init__(self, scene, w, h, sky=False, presampling=False, force_bvh=False, half_vec=False):
        # half_vec: the emissive DI shaders compiled with USE_HALF_VECTOR_COPY_SHIFT = 1 (libzref_di_e1h.so; the reference's tree has 0)
        assert not (half_vec and (sky or presampling))
        super().__init__("di_sky" if sky else ("di_e1h" if half_vec else ("di_e1p" if presampling else "di_e1")), scene, force_bvh)
        L = self.L
        L.zrefp_di_create.restype = C.c_void_p
        L.zrefp_di_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zrefp_di_render.argtypes = [C.c_void_p] * 7
        L.zrefp_di_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.w, self.h_, self.sky = w, h, sky
        self.st = L.zrefp_di_create(w, h)
        self.prev = None
        self.PLANES = ({"A": (0, np.uint8, 1), "B": (1, np.uint16, 2), "C": (2, np.float32, 2), "target": (8, np.uint16, 4)} if sky else
                       {"A": (0, np.uint32, 4), "B": (1, np.float32, 2), "target": (8, np.uint16, 4)})

    def render(self, cb, params, gb):
        out = np.zeros((self.h_, self.w, 4), np.float32)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        self.L.zrefp_di_render(self.h, self.st, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), out.ctypes.data)
        self.prev = gb
        return out

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        buf = np.zeros((self.h_, self.w, ch), dt)
        self.L.zrefp_di_read_plane(self.st, idx, buf.ctypes.data)
        return buf


class RefPost:
    """AutoExposure_Histogram / AutoExposure_WeightedAvg / Display.hlsl (libzref_post.so: no scene)"""

    def __init__(self):
        self.L = C.CDLL(os.path.join(HERE, "_ref", "libzref_post.so"))
        self.L.zrefp_auto_exposure.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.zrefp_display.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                         C.c_void_p, C.c_uint32, C.c_void_p]

    @staticmethod
    def _image(image):
        a = np.ascontiguousarray(image)
        assert a.ndim == 3 and a.shape[2] == 4 and a.dtype in (np.uint16, np.float32)
        return a, int(a.dtype == np.uint16)

    def auto_exposure(self, image, params, cb, exposure2=None):
        a, is16 = self._image(image)
        h, w = a.shape[:2]
        prm4 = np.array([params.ae_min_lum, np.float32(params.ae_max_lum) - np.float32(params.ae_min_lum), params.ae_lum_map_exp, params.ae_adaptation_rate], np.float32)
        hist = np.zeros(256, np.uint32)
        e = np.zeros(2, np.float32) if exposure2 is None else np.array(exposure2, np.float32).copy()
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_auto_exposure(a.ctypes.data, is16, w, h, cbb.ctypes.data, prm4.ctypes.data, hist.ctypes.data, e.ctypes.data) == 0
        return hist, e

    def display(self, image, params, cb, exposure2=None, lut=None):
        a, is16 = self._image(image)
        rh, rw = a.shape[:2]
        cbb = np.ascontiguousarray(cb)
        dw, dh = int(np.asarray(cbb["display_width"]).reshape(-1)[0]), int(np.asarray(cbb["display_height"]).reshape(-1)[0])
        out = np.zeros((dh, dw, 4), np.float32)
        e = None if exposure2 is None else np.array(exposure2, np.float32).copy()
        l = None if lut is None else np.ascontiguousarray(lut, np.uint32)
        dim = 0 if l is None else int(round(l.size ** (1.0 / 3.0)))
        assert self.L.zrefp_display(a.ctypes.data, is16, rw, rh, cbb.ctypes.data, None if e is None else e.ctypes.data, params.display_tonemapper,
                                    params.display_auto_exposure, params.display_saturation, params.display_agx_exp,
                                    None if l is None else l.ctypes.data, dim, out.ctypes.data) == 0
        return out


class RefAux(RefPass):
    """The reference's auxiliary shaders (libzref_aux.so): PreLighting K2 / K3 / K4 with a scene; SkyViewLUT (K17), Compositing +
    FireflyFilter and TAA.  `scene` may be None for the passes that need none (sky_lut, taa)."""

    def __init__(self, scene=None, force_bvh=False):
        if scene is not None:
            super().__init__("aux", scene, force_bvh)
        else:
            self.L, self.h = _lib("aux"), None
        L = self.L
        vp, u32 = C.c_void_p, C.c_uint32
        L.zrefp_estimate_power.argtypes = [vp, vp, vp, vp]
        L.zrefp_presample.argtypes = [vp, vp, u32, u32, vp]
        L.zrefp_build_lvg.argtypes = [vp, vp, vp, vp, C.c_float, vp]
        L.zrefp_sky_lut.argtypes = [vp, u32, u32, vp]
        L.zrefp_composite.argtypes = [vp, vp, vp, vp, vp, vp, u32, C.c_int, vp]
        L.zrefp_taa.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, C.c_float, C.c_int]

    def __del__(self):
        if self.h is not None:
            super().__del__()

    @staticmethod
    def halton_table():
        """the 64 Halton(2, 3) points of PreLighting::Init (PreLighting.cpp:236-243) from the reference's own Halton (libzref.so)"""
        L = C.CDLL(os.path.join(HERE, "_ref", "libzref.so"))
        L.zref_halton.restype = C.c_float
        L.zref_halton.argtypes = [C.c_int, C.c_int]
        return np.array([[L.zref_halton(i + 1, 2), L.zref_halton(i + 1, 3)] for i in range(64)], np.float32)

    def estimate_power(self, cb):
        out = np.zeros(len(self._desc_scene().emissives), np.float32)
        cbb, hal = np.ascontiguousarray(cb), self.halton_table()
        assert self.L.zrefp_estimate_power(self.h, cbb.ctypes.data, hal.ctypes.data, out.ctypes.data) == 0
        return out

    def _desc_scene(self):
        return self._scene

    def presample(self, cb, num_sets, set_size):
        from zetaray_amd import wire
        out = np.zeros(num_sets * set_size, wire.PRESAMPLED_TRI)
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_presample(self.h, cbb.ctypes.data, num_sets, set_size, out.ctypes.data) == 0
        return out

    def build_lvg(self, cb, dim, extents, offset_y):
        from zetaray_amd import wire
        d, e = np.array(dim, np.uint32), np.array(extents, np.float32)
        out = np.zeros((int(d[2]), int(d[1]), int(d[0]), 64), wire.VOXEL_SAMPLE)
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_build_lvg(self.h, cbb.ctypes.data, d.ctypes.data, e.ctypes.data, float(offset_y), out.ctypes.data) == 0
        return out

    def sky_lut(self, cb, w=256, h=128):
        out = np.zeros((h, w), np.uint32)
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_sky_lut(cbb.ctypes.data, w, h, out.ctypes.data) == 0
        return out

    def composite(self, cb, gb_planes, sky_di=None, emissive_di=None, indirect=None, flags=0, firefly=False, out=None):
        """gb_planes: the zr_gbuffer_planes struct of this frame; inputs (h, w, 4) f32 or None; flags = CB_COMPOSIT_FLAGS"""
        w, h = gb_planes.width, gb_planes.height
        ins = [None if p is None else np.ascontiguousarray(p, np.float32) for p in (sky_di, emissive_di, indirect)]
        o = np.zeros((h, w, 4), np.float32) if out is None else np.ascontiguousarray(out, np.float32).copy()
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_composite(self.h, cbb.ctypes.data, C.addressof(gb_planes), *[None if p is None else p.ctypes.data for p in ins], flags, int(firefly), o.ctypes.data) == 0
        return o

    def taa(self, cb, signal, depth, motion, prev_out, blend_weight=0.1, temporal_valid=True):
        sig = np.ascontiguousarray(signal, np.float32)
        h, w = sig.shape[:2]
        d, m, prev = np.ascontiguousarray(depth, np.float32), np.ascontiguousarray(motion, np.uint32), np.ascontiguousarray(prev_out, np.uint16)
        out = np.zeros((h, w, 4), np.uint16)
        cbb = np.ascontiguousarray(cb)
        assert self.L.zrefp_taa(cbb.ctypes.data, sig.ctypes.data, d.ctypes.data, m.ctypes.data, prev.ctypes.data, out.ctypes.data, w, h, float(blend_weight), int(bool(temporal_valid))) == 0
        return out
