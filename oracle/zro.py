"""ctypes binding of the CPU oracle (oracle/libzro.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libzro.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libzro.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.zro_scene_create.restype = C.c_void_p
        L.zro_scene_create.argtypes = [C.c_void_p, C.c_int]
        L.zro_scene_destroy.argtypes = [C.c_void_p]
        L.zro_scene_num_tris.argtypes = [C.c_void_p]
        L.zro_alias_table_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_kahan_sum.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.zro_scene_set_alias_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_estimate_power.argtypes = [C.c_void_p, C.c_void_p]
        L.zro_gbuffer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.zro_pathtrace_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zro_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_trace_closest_timed.restype = C.c_double
        L.zro_trace_closest_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_kat_unary.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_f32_to_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_f16_to_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_pcg3d.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_oct_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_oct_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.zro_kat_uniform_bounded.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.zro_kat_alias_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.zro_kat_bsdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _LIB = L
    return _LIB


class OracleScene:
    def __init__(self, scene, force_bvh=False):
        """scene: zetaray_amd.scene_io.Scene"""
        from zetaray_amd import wire
        self.scene = scene
        self._desc = scene.desc()
        self.h = lib().zro_scene_create(C.addressof(self._desc), int(force_bvh))
        self.alias = None
        if len(scene.emissives):
            power = self.estimate_power()
            self.power = power
            self.alias = alias_table_build(power)
            lib().zro_scene_set_alias_table(self.h, self.alias.ctypes.data, len(self.alias))

    def __del__(self):
        if getattr(self, "h", None):
            lib().zro_scene_destroy(self.h)
            self.h = None

    def estimate_power(self):
        out = np.zeros(len(self.scene.emissives), np.float32)
        lib().zro_estimate_power(self.h, out.ctypes.data)
        return out

    def gbuffer(self, cb):
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        lib().zro_gbuffer_render(self.h, cbb.ctypes.data, C.addressof(planes))
        return arrays, planes

    def pathtrace(self, cb, gb_planes, params, final=None):
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        if final is None:
            final = np.zeros((h, w, 4), np.float32)
        cnt = wire.Counters()
        cbb = np.ascontiguousarray(cb)
        lib().zro_pathtrace_render(self.h, cbb.ctypes.data, C.addressof(gb_planes), C.addressof(params),
                                   final.ctypes.data, C.addressof(cnt))
        return final, (cnt.n_closest, cnt.n_shadow)

    def trace_closest(self, rays, mask=3, timed=False):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros((len(rays), 4), np.uint32)
        if timed:
            dt = lib().zro_trace_closest_timed(self.h, rays.ctypes.data, len(rays), mask, hits.ctypes.data)
            return hits, dt
        lib().zro_trace_closest(self.h, rays.ctypes.data, len(rays), mask, hits.ctypes.data)
        return hits

    def trace_any(self, rays, mask=3):
        rays = np.ascontiguousarray(rays, np.float32)
        occ = np.zeros(len(rays), np.uint32)
        lib().zro_trace_any(self.h, rays.ctypes.data, len(rays), mask, occ.ctypes.data)
        return occ


def alias_table_build(power, align_phase=0):
    from zetaray_amd import wire
    power = np.ascontiguousarray(power, np.float32)
    out = np.zeros(len(power), wire.ALIAS_ENTRY)
    lib().zro_alias_table_build(power.ctypes.data, len(power), align_phase, out.ctypes.data)
    return out


def kahan_sum(data, align_phase=0):
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros(1, np.float32)
    lib().zro_kahan_sum(data.ctypes.data, len(data), align_phase, out.ctypes.data)
    return out[0]


def kat_unary(fn, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().zro_kat_unary(fn, x.ctypes.data, y.ctypes.data, x.size)
    return y
