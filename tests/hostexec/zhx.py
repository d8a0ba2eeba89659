"""ctypes binding of tests/hostexec/libzhx.so (TEST-ONLY serial executor of the HIP stage functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        # (one process at a time: pytest-xdist workers would otherwise relink the library while another worker loads it)
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", _HERE, "libzhx.so"])
            L = C.CDLL(os.path.join(_HERE, "libzhx.so"))
        L.zhx_scene_create.restype = C.c_void_p
        L.zhx_scene_create.argtypes = [C.c_void_p]
        L.zhx_scene_destroy.argtypes = [C.c_void_p]
        L.zhx_scene_set_alias.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.zhx_bvh_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zhx_estimate_power.argtypes = [C.c_void_p, C.c_void_p]
        L.zhx_latch_heap_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.zhx_tex_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zhx_gbuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.zhx_pathtrace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zhx_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zhx_set_tile_origin.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zhx_presample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zhx_build_lvg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.zhx_sky_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zhx_le_sky.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zhx_le_sun.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zhx_rpt_create.restype = C.c_void_p
        L.zhx_rpt_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_rpt_destroy.argtypes = [C.c_void_p]
        L.zhx_rpt_reset_temporal.argtypes = [C.c_void_p]
        L.zhx_rpt_render.argtypes = [C.c_void_p] * 8
        L.zhx_rpt_render_stage.argtypes = [C.c_void_p] * 8 + [C.c_int]
        L.zhx_rpt_set_owned_rect.argtypes = [C.c_uint32] * 4
        L.zhx_rpt_write_plane_rect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_uint32] * 4
        L.zhx_rpt_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.zhx_rdi_create.restype = C.c_void_p
        L.zhx_rdi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_rdi_destroy.argtypes = [C.c_void_p]
        L.zhx_rdi_reset_temporal.argtypes = [C.c_void_p]
        L.zhx_rdi_render.argtypes = [C.c_void_p] * 8
        L.zhx_rdi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.zhx_sdi_create.restype = C.c_void_p
        L.zhx_sdi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_sdi_destroy.argtypes = [C.c_void_p]
        L.zhx_sdi_reset_temporal.argtypes = [C.c_void_p]
        L.zhx_sdi_render.argtypes = [C.c_void_p] * 8
        L.zhx_sdi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.zhx_rgi_create.restype = C.c_void_p
        L.zhx_rgi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_rgi_destroy.argtypes = [C.c_void_p]
        L.zhx_rgi_reset_temporal.argtypes = [C.c_void_p]
        L.zhx_rgi_render.argtypes = [C.c_void_p] * 8
        L.zhx_rgi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


class HostExecScene:
    def __init__(self, scene, alias=None):
        self.scene = scene
        self._desc = scene.desc()
        self.h = lib().zhx_scene_create(C.addressof(self._desc))
        if alias is not None:
            self._alias = np.ascontiguousarray(alias)
            lib().zhx_scene_set_alias(self.h, self._alias.ctypes.data, len(self._alias))

    def __del__(self):
        if getattr(self, "h", None):
            lib().zhx_scene_destroy(self.h)
            self.h = None

    def bvh_digest(self):
        """(FNV-1a over the built 4-wide nodes + leaf-ordered triangles + stack bound, nodes, triangles, stack bound)"""
        L = lib()
        L.zhx_bvh_digest.restype = C.c_uint64
        L.zhx_bvh_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        n, t, st = C.c_uint32(), C.c_uint32(), C.c_uint32()
        return int(L.zhx_bvh_digest(self.h, C.byref(n), C.byref(t), C.byref(st))), n.value, t.value, st.value

    def update_instances(self, instances, instance_to_world):
        L = lib()
        L.zhx_scene_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        i, x = np.ascontiguousarray(instances), np.ascontiguousarray(instance_to_world, np.float32)
        L.zhx_scene_update_instances(self.h, i.ctypes.data, x.ctypes.data, len(i))

    def set_own_subtree(self, flags):
        """instances the next update_instances rebuild keeps in subtrees of their own (what the product's background rebuild does for instances that moved)"""
        L = lib()
        L.zhx_scene_set_own_subtree.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        f = np.ascontiguousarray(flags, np.uint8)
        L.zhx_scene_set_own_subtree(self.h, f.ctypes.data, len(f))

    def update_emissives(self, triangles, first=0):
        L = lib()
        L.zhx_scene_update_emissives.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        t = np.ascontiguousarray(triangles)
        L.zhx_scene_update_emissives(self.h, t.ctypes.data, first, len(t))

    def bvh_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().zhx_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def latch_heap_offsets(self, cb):
        cbb = np.ascontiguousarray(cb)
        lib().zhx_latch_heap_offsets(self.h, cbb.ctypes.data)

    def tex_sample(self, tex, mode, uv, g=None):
        """zr_texture.h on this scene's heap: mode 0 point, 1 SampleLevel (lod = g[:, 0]), 2 SampleGrad (g = ddx.uv, ddy.uv)"""
        uv = np.ascontiguousarray(uv, np.float32)
        g = np.zeros((len(uv), 4), np.float32) if g is None else np.ascontiguousarray(g, np.float32)
        out = np.zeros((len(uv), 4), np.float32)
        lib().zhx_tex_sample(self.h, C.c_uint32(tex), C.c_int(mode), uv.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                            C.c_uint32(len(uv)), out.ctypes.data_as(C.c_void_p))
        return out

    def estimate_power(self):
        out = np.zeros(len(self.scene.emissives), np.float32)
        lib().zhx_estimate_power(self.h, out.ctypes.data)
        return out

    def presample(self, frame_num, num_sets, set_size):
        from zetaray_amd import wire
        out = np.zeros(num_sets * set_size, wire.PRESAMPLED_TRI)
        lib().zhx_presample(self.h, frame_num, num_sets, set_size, out.ctypes.data)
        return out

    def build_lvg(self, cb, dim, extents, offset_y):
        """K4: build and bind the light voxel grid; returns (dz, dy, dx, 64) wire.VOXEL_SAMPLE records"""
        from zetaray_amd import wire
        d = np.array(dim, np.uint32)
        e = np.array(extents, np.float32)
        out = np.zeros((int(d[2]), int(d[1]), int(d[0]), 64), wire.VOXEL_SAMPLE)
        cbb = np.ascontiguousarray(cb)
        lib().zhx_build_lvg(self.h, cbb.ctypes.data, d.ctypes.data, e.ctypes.data, float(offset_y), out.ctypes.data)
        return out

    def sky_lut(self, cb, w=256, h=128):
        """K17: generate and bind the sky-view LUT; returns the R11G11B10F texels (h, w) uint32"""
        out = np.zeros((h, w), np.uint32)
        cbb = np.ascontiguousarray(cb)
        lib().zhx_sky_lut(self.h, cbb.ctypes.data, w, h, out.ctypes.data)
        return out

    def le_sky(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(d)
        lib().zhx_le_sky(self.h, d.ctypes.data, len(d), out.ctypes.data)
        return out

    @staticmethod
    def le_sun(cb, pos):
        p = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        out = np.zeros_like(p)
        cbb = np.ascontiguousarray(cb)
        lib().zhx_le_sun(cbb.ctypes.data, p.ctypes.data, len(p), out.ctypes.data)
        return out

    def gbuffer(self, cb, tile=None):
        """tile = (x0, y0, w, h) renders only that screen tile (planes have the tile's size)."""
        from zetaray_amd import wire
        x0, y0, w, h = tile if tile else (0, 0, int(cb["render_width"]), int(cb["render_height"]))
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        lib().zhx_set_tile_origin(x0, y0)
        lib().zhx_gbuffer(self.h, cbb.ctypes.data, C.addressof(planes))
        lib().zhx_set_tile_origin(0, 0)
        return arrays, planes

    def pick(self, cb, x, y, tile=None):
        """GBufferRT::PickPixel(x, y) + a G-buffer render through the product's stage function"""
        L = lib()
        L.zhx_pick_pixel.argtypes = [C.c_uint32, C.c_uint32]
        L.zhx_picked.restype = C.c_uint32
        L.zhx_pick_pixel(int(x), int(y))
        self.gbuffer(cb, tile)
        L.zhx_pick_pixel(0xffff, 0xffff)
        return int(L.zhx_picked())

    def pathtrace(self, cb, planes, params, final=None, tile=None):
        from zetaray_amd import wire
        x0, y0, w, h = tile if tile else (0, 0, int(cb["render_width"]), int(cb["render_height"]))
        lib().zhx_set_tile_origin(x0, y0)
        try:
            return self._pathtrace(cb, planes, params, final, w, h)
        finally:
            lib().zhx_set_tile_origin(0, 0)

    def _pathtrace(self, cb, planes, params, final, w, h):
        from zetaray_amd import wire
        if final is None:
            final = np.zeros((h, w, 4), np.float32)
        cnt = wire.Counters()
        cbb = np.ascontiguousarray(cb)
        lib().zhx_pathtrace(self.h, cbb.ctypes.data, C.addressof(planes), C.addressof(params), final.ctypes.data, C.addressof(cnt))
        return final, (cnt.n_closest, cnt.n_shadow)

    def trace_closest(self, rays, mask=3):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros((len(rays), 4), np.uint32)
        lib().zhx_trace_closest(self.h, rays.ctypes.data, len(rays), mask, hits.ctypes.data)
        return hits

    def trace_any(self, rays, mask=3):
        rays = np.ascontiguousarray(rays, np.float32)
        occ = np.zeros(len(rays), np.uint32)
        lib().zhx_trace_any(self.h, rays.ctypes.data, len(rays), mask, occ.ctypes.data)
        return occ


class HostExecRPT:
    """ReSTIR PT through the HIP stage functions, serially (mirror of oracle.zro.OracleRPT)."""
    PLANES = {"A": (0, np.uint32, 1), "B": (1, np.float32, 2), "C": (2, np.uint32, 4), "D": (3, np.uint32, 4),
              "E": (4, np.uint16, 1), "F": (5, np.float32, 2), "G": (6, np.uint32, 2), "target": (7, np.float32, 4),
              "neighbor": (8, np.uint8, 2), "map_ctn": (18, np.uint16, 1), "map_ntc": (19, np.uint16, 1),
              "ctn_A": (10, np.uint16, 4), "ctn_B": (11, np.uint32, 4), "ctn_C": (12, np.uint32, 4), "ctn_D": (13, np.uint16, 1),
              "ntc_A": (14, np.uint16, 4), "ntc_B": (15, np.uint32, 4), "ntc_C": (16, np.uint32, 4), "ntc_D": (17, np.uint16, 1)}

    def __init__(self, hxscene, w, h, ext=None, owned=None):
        """w, h = plane size (the extended tile's size when ext is given)"""
        self.hx, self.w, self.h = hxscene, w, h
        self.ext, self.owned = ext, owned
        self.r = lib().zhx_rpt_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zhx_rpt_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zhx_rpt_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        return self.render_stage(cb, params, 7, gb)

    def render_stage(self, cb, params, stages, gb=None):
        """stages: 1 = K11 + temporal, 2 = spatial (first round) + end of frame, 4 = the second spatial round (num_spatial_passes = 2), 7 = all.  With ext=(x0, y0, w, h) the planes cover that
        extended tile and `owned` is the rect this instance shades (multi-device split)."""
        from zetaray_amd import wire
        if stages & 1:
            if gb is None:
                gb = self.hx.gbuffer(cb, tile=self.ext)
            self._gb = gb
        gb = self._gb
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        L = lib()
        if self.ext is not None:
            L.zhx_set_tile_origin(self.ext[0], self.ext[1])
            L.zhx_rpt_set_owned_rect(*self.owned)
        try:
            L.zhx_rpt_render_stage(self.hx.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params),
                                   self.final.ctypes.data, C.addressof(cnt), stages)
        finally:
            L.zhx_set_tile_origin(0, 0)
            L.zhx_rpt_set_owned_rect(0, 0, 0, 0)
        c = getattr(self, "counters", (0, 0)) if not (stages & 1) else (0, 0)
        self.counters = (c[0] + cnt.n_closest, c[1] + cnt.n_shadow)
        last = 4 if (params.num_spatial_passes == 2 and (params.flags & 0x2)) else 2      # the stage after which the frame's G-buffer becomes "previous"
        if stages & last:
            self.prev = gb
        return self.final

    def write_plane_rect(self, name, which, full, rect_local):
        """copy rect_local = (x0, y0, w, h) (plane-local coordinates) of the full-size array `full` into the plane"""
        idx, dt, ch = self.PLANES[name]
        full = np.ascontiguousarray(full, dt)
        lib().zhx_rpt_write_plane_rect(self.r, which, idx, full.ctypes.data, *rect_local)

    def plane(self, name, which=0):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zhx_rpt_read_plane(self.r, which, idx, out.ctypes.data)
        return out


class HostExecRDI:
    """ReSTIR DI (emissive) through the HIP stage functions, serially (mirror of oracle.zro.OracleRDI)."""
    PLANES = {"A": (0, np.uint32, 4), "B": (1, np.float32, 2), "target": (2, np.float32, 4)}

    def __init__(self, hxscene, w, h):
        self.hx, self.w, self.h = hxscene, w, h
        self.r = lib().zhx_rdi_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zhx_rdi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zhx_rdi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.hx.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zhx_rdi_render(self.hx.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), self.final.ctypes.data,
                             C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zhx_rdi_read_plane(self.r, idx, out.ctypes.data)
        return out


class HostExecSDI:
    """Sun + sky ReSTIR DI through the HIP stage functions, serially (mirror of oracle.zro.OracleSDI)."""
    PLANES = {"A": (0, np.uint8, 1), "B": (1, np.uint16, 2), "C": (2, np.float32, 2), "target": (3, np.float32, 4)}

    def __init__(self, hxscene, w, h):
        self.hx, self.w, self.h = hxscene, w, h
        self.r = lib().zhx_sdi_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zhx_sdi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zhx_sdi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.hx.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zhx_sdi_render(self.hx.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), self.final.ctypes.data,
                             C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zhx_sdi_read_plane(self.r, idx, out.ctypes.data)
        return out


class HostExecRGI:
    """ReSTIR GI through the HIP stage functions, serially (mirror of oracle.zro.OracleRGI)."""
    PLANES = {"A": (0, np.float32, 4), "B": (1, np.uint16, 4), "C": (2, np.float32, 4)}

    def __init__(self, hxscene, w, h):
        self.hx, self.w, self.h = hxscene, w, h
        self.r = lib().zhx_rgi_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zhx_rgi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zhx_rgi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.hx.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zhx_rgi_render(self.hx.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params), self.final.ctypes.data,
                             C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zhx_rgi_read_plane(self.r, idx, out.ctypes.data)
        return out


def set_k11_carry(on):
    """K11 emulation: at every bounce boundary a live path is rebuilt from the words rpt::PtCarry moves (what the compacting kernels carry) and nothing else"""
    lib().zhx_set_k11_carry(int(bool(on)))


def set_k11_fused(on):
    """K11 emulation runs the fused stage functions (PtInitLane_Fused / PtPhaseA_Fused, the ones the inline megakernel compiles) instead of the cut ones"""
    lib().zhx_set_k11_fused(int(bool(on)))


def set_material_class(plain):
    """the ReSTIR PT stage functions as the PLAIN kernel permutations run them (SceneView / GBuf / RBuf::plain = 1); only for scenes whose materials are plain"""
    lib().zhx_set_material_class(int(bool(plain)))


def set_k11_park(on):
    """K11 emulation of k_rpt_pathtrace_park: the reservoir's selected reconnection lives in a [word][lane] park outside the lane (zr_rpt.h RcPark)"""
    lib().zhx_set_k11_park(int(bool(on)))


def svgf(signal_rgba, depth, normal, motion, prev_depth, prev_normal, hist_color, hist_moments, temporal_valid=True, alpha=0.2, alpha_moments=0.2,
         sigma_l=4.0, sigma_z=1.0, normal_power_log2=7, iterations=5):
    """the denoise pass's HIP stage functions (zr_svgf.h) run serially on the host; same interface as oracle.zro.svgf"""
    sig = np.ascontiguousarray(signal_rgba, np.float32)
    h, w = sig.shape[:2]
    d = np.ascontiguousarray(depth, np.float32); n = np.ascontiguousarray(normal, np.uint32); m = np.ascontiguousarray(motion, np.uint32)
    pd = np.ascontiguousarray(prev_depth, np.float32); pn = np.ascontiguousarray(prev_normal, np.uint32)
    hc = np.array(hist_color, np.float32, copy=True).reshape(h, w, 4); hm = np.array(hist_moments, np.float32, copy=True).reshape(h, w, 2)
    out = np.zeros((h, w, 4), np.float32)
    p4 = np.array([alpha, alpha_moments, sigma_l, sigma_z], np.float32)
    f = lib().zhx_svgf
    f.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    f(sig.ctypes.data, d.ctypes.data, n.ctypes.data, m.ctypes.data, pd.ctypes.data, pn.ctypes.data, hc.ctypes.data, hm.ctypes.data,
      int(bool(temporal_valid)), p4.ctypes.data, int(normal_power_log2), int(iterations), w, h, out.ctypes.data)
    return out, hc, hm


STAGE_DENOISE_TEMPORAL, STAGE_DENOISE_VARIANCE, STAGE_DENOISE_MASK = 1 << 8, 1 << 9, 0x3ff00


def stage_denoise_atrous(i):
    return 1 << (10 + i)


class HostExecDenoise:
    """The denoise pass's HIP stage functions with the pass's state, over a window of the frame and step by step (what zr_pass_render_stage runs on a
    device of the tile split): window = (x0, y0, w, h) of a frame (W, H); planes handed to render() are the window's."""
    PLANES = {"history": (0, 4), "moments": (1, 2), "iter": (2, 4), "out": (3, 4)}

    def __init__(self, W, H, window=None, **kw):
        self.W, self.H = W, H
        self.win = tuple(window) if window is not None else (0, 0, W, H)
        self.prm = dict(alpha=0.2, alpha_moments=0.2, sigma_l=4.0, sigma_z=1.0, normal_power_log2=7, iterations=5)
        self.prm.update(kw)
        L = lib()
        L.zhx_svgf_create.restype = C.c_void_p
        L.zhx_svgf_create.argtypes = [C.c_int] * 6
        self.h = C.c_void_p(L.zhx_svgf_create(self.win[0], self.win[1], self.win[2], self.win[3], W, H))

    def __del__(self):
        if getattr(self, "h", None):
            lib().zhx_svgf_destroy.argtypes = [C.c_void_p]
            lib().zhx_svgf_destroy(self.h)
            self.h = None

    def render(self, signal, depth, normal, motion, prev_depth, prev_normal, temporal_valid=True, steps=STAGE_DENOISE_MASK):
        ph, pw = self.win[3], self.win[2]
        arrs = [np.ascontiguousarray(signal, np.float32).reshape(ph, pw, 4), np.ascontiguousarray(depth, np.float32).reshape(ph, pw),
                np.ascontiguousarray(normal, np.uint32).reshape(ph, pw), np.ascontiguousarray(motion, np.uint32).reshape(ph, pw),
                np.ascontiguousarray(prev_depth, np.float32).reshape(ph, pw), np.ascontiguousarray(prev_normal, np.uint32).reshape(ph, pw)]
        p4 = np.array([self.prm["alpha"], self.prm["alpha_moments"], self.prm["sigma_l"], self.prm["sigma_z"]], np.float32)
        f = lib().zhx_svgf_render
        f.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        f(self.h, *[a.ctypes.data for a in arrs], int(bool(temporal_valid)), p4.ctypes.data, int(self.prm["normal_power_log2"]), int(self.prm["iterations"]), int(steps))

    def plane(self, name):
        which, ch = self.PLANES[name]
        out = np.zeros((self.win[3], self.win[2], ch), np.float32)
        lib().zhx_svgf_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib().zhx_svgf_read_plane(self.h, which, out.ctypes.data)
        return out

    def write_plane_rect(self, name, full, rect_local):
        which, ch = self.PLANES[name]
        full = np.ascontiguousarray(full, np.float32)
        assert full.shape == (self.win[3], self.win[2], ch), (full.shape, self.win)
        lib().zhx_svgf_write_plane_rect.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_uint32] * 4
        lib().zhx_svgf_write_plane_rect(self.h, which, full.ctypes.data, *[int(v) for v in rect_local])


def taa(signal_rgba, depth, motion, prev_out, blend_weight=0.1, temporal_valid=True):
    """TAA.hlsl on an RGBA32F signal (h, w, 4), depth (h, w) f32, motion (h, w) u32 (R16G16_SNORM), history (h, w, 4) f16 bits (u16);
    returns the new RGBA16F output as u16 (alpha = the history buffer's, untouched)."""
    sig = np.ascontiguousarray(signal_rgba, np.float32)
    h, w = sig.shape[:2]
    d = np.ascontiguousarray(depth, np.float32)
    m = np.ascontiguousarray(motion, np.uint32)
    prev = np.ascontiguousarray(prev_out, np.uint16)
    out = np.zeros((h, w, 4), np.uint16)
    f = lib().zhx_taa
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int]
    f(sig.ctypes.data, d.ctypes.data, m.ctypes.data, prev.ctypes.data, out.ctypes.data, w, h, float(blend_weight), int(bool(temporal_valid)))
    return out
