"""ctypes binding of the CPU oracle (oracle/libzro.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    # (one process at a time: pytest-xdist workers would otherwise relink the library while another worker loads it)
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
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
        L.zro_scene_latch_heap_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.zro_tex_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zro_gbuffer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.zro_pathtrace_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zro_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_trace_closest_timed.restype = C.c_double
        L.zro_trace_closest_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_presample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_firefly.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.zro_build_lvg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.zro_sky_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_le_sky.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zro_le_sun.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.zro_rpt_create.restype = C.c_void_p
        L.zro_rpt_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_rpt_destroy.argtypes = [C.c_void_p]
        L.zro_rpt_reset_temporal.argtypes = [C.c_void_p]
        L.zro_rpt_render.argtypes = [C.c_void_p] * 8
        L.zro_rpt_self_shift.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.zro_rpt_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.zro_rdi_create.restype = C.c_void_p
        L.zro_rdi_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_rdi_destroy.argtypes = [C.c_void_p]
        L.zro_rdi_reset_temporal.argtypes = [C.c_void_p]
        L.zro_rdi_render.argtypes = [C.c_void_p] * 8
        L.zro_rdi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.zro_sdi_create.restype = C.c_void_p
        L.zro_sdi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zro_sdi_destroy.argtypes = [C.c_void_p]
        L.zro_sdi_reset_temporal.argtypes = [C.c_void_p]
        L.zro_sdi_render.argtypes = [C.c_void_p] * 8
        L.zro_sdi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.zro_rgi_create.restype = C.c_void_p
        L.zro_rgi_create.argtypes = [C.c_uint32, C.c_uint32]
        L.zro_rgi_destroy.argtypes = [C.c_void_p]
        L.zro_rgi_reset_temporal.argtypes = [C.c_void_p]
        L.zro_rgi_render.argtypes = [C.c_void_p] * 8
        L.zro_rgi_read_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
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


def firefly_filter(rgba, depth):
    """FireflyFilter.hlsl on an (h, w, 4) float32 image with the (h, w) float32 view-depth plane (FLT_MAX = no geometry)"""
    rgba = np.ascontiguousarray(rgba, np.float32)
    depth = np.ascontiguousarray(depth, np.float32).reshape(rgba.shape[0], rgba.shape[1])
    out = np.zeros_like(rgba)
    lib().zro_firefly(rgba.ctypes.data, depth.ctypes.data, out.ctypes.data, rgba.shape[1], rgba.shape[0])
    return out


class OracleScene:
    def __init__(self, scene, force_bvh=False, cb=None):
        """scene: zetaray_amd.scene_io.Scene.  cb: frame constants whose texture descriptor-table offsets the power
        estimate (K2) should use; only matters for scenes with emissive textures at a non-zero table offset."""
        from zetaray_amd import wire
        self.scene = scene
        self._desc = scene.desc()
        self.h = lib().zro_scene_create(C.addressof(self._desc), int(force_bvh))
        if cb is not None:
            self.latch_heap_offsets(cb)
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

    def update_instances(self, instances, instance_to_world):
        """zr_scene_update_instances on the oracle: new MeshInstance records + object-to-world matrices; the scene as it was becomes
        the previous one (bound by the CtT passes of ReSTIR PT and the temporal shifts of the DI passes)"""
        i, x = np.ascontiguousarray(instances), np.ascontiguousarray(instance_to_world, np.float32)
        lib().zro_scene_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        assert lib().zro_scene_update_instances(self.h, i.ctypes.data, x.ctypes.data, len(i)) == 0

    def update_emissives(self, triangles, first=0):
        """zr_scene_update_emissives on the oracle"""
        t = np.ascontiguousarray(triangles)
        lib().zro_scene_update_emissives.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        assert lib().zro_scene_update_emissives(self.h, t.ctypes.data, first, len(t)) == 0

    def latch_heap_offsets(self, cb):
        cbb = np.ascontiguousarray(cb)
        lib().zro_scene_latch_heap_offsets(self.h, cbb.ctypes.data)

    def tex_sample(self, tex, mode, uv, g=None):
        """zr_texture.h on this scene's heap: mode 0 point, 1 SampleLevel (lod = g[:, 0]), 2 SampleGrad (g = ddx.uv, ddy.uv)"""
        uv = np.ascontiguousarray(uv, np.float32)
        g = np.zeros((len(uv), 4), np.float32) if g is None else np.ascontiguousarray(g, np.float32)
        out = np.zeros((len(uv), 4), np.float32)
        lib().zro_tex_sample(self.h, C.c_uint32(tex), C.c_int(mode), uv.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                            C.c_uint32(len(uv)), out.ctypes.data_as(C.c_void_p))
        return out

    def update_materials(self, materials, first=0):
        """zr_scene_update_materials on the oracle"""
        m = np.ascontiguousarray(materials)
        lib().zro_scene_update_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        assert lib().zro_scene_update_materials(self.h, m.ctypes.data, first, len(m)) == 0

    def rebuild_alias_table(self):
        """emissive materials changed (zr_scene_invalidate_alias_table + the next PRELIGHTING render): K2 + the alias-table build again"""
        self.power = self.estimate_power()
        self.alias = alias_table_build(self.power)
        lib().zro_scene_set_alias_table(self.h, self.alias.ctypes.data, len(self.alias))

    def estimate_power(self):
        out = np.zeros(len(self.scene.emissives), np.float32)
        lib().zro_estimate_power(self.h, out.ctypes.data)
        return out

    def presample(self, frame_num, num_sets, set_size):
        """K3: (re)generate the presampled light sets for this frame; returns them (wire.PRESAMPLED_TRI records)"""
        from zetaray_amd import wire
        out = np.zeros(num_sets * set_size, wire.PRESAMPLED_TRI)
        lib().zro_presample(self.h, frame_num, num_sets, set_size, out.ctypes.data)
        return out

    def build_lvg(self, cb, dim, extents, offset_y):
        """K4: build and bind the light voxel grid; returns (dz, dy, dx, 64) wire.VOXEL_SAMPLE records"""
        from zetaray_amd import wire
        d = np.array(dim, np.uint32)
        e = np.array(extents, np.float32)
        out = np.zeros((int(d[2]), int(d[1]), int(d[0]), 64), wire.VOXEL_SAMPLE)
        cbb = np.ascontiguousarray(cb)
        lib().zro_build_lvg(self.h, cbb.ctypes.data, d.ctypes.data, e.ctypes.data, float(offset_y), out.ctypes.data)
        return out

    def sky_lut(self, cb, w=256, h=128):
        """K17: generate and bind the sky-view LUT; returns the R11G11B10F texels (h, w) uint32"""
        out = np.zeros((h, w), np.uint32)
        cbb = np.ascontiguousarray(cb)
        lib().zro_sky_lut(self.h, cbb.ctypes.data, w, h, out.ctypes.data)
        return out

    def le_sky(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(d)
        lib().zro_le_sky(self.h, d.ctypes.data, len(d), out.ctypes.data)
        return out

    @staticmethod
    def le_sun(cb, pos):
        p = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        out = np.zeros_like(p)
        cbb = np.ascontiguousarray(cb)
        lib().zro_le_sun(cbb.ctypes.data, p.ctypes.data, len(p), out.ctypes.data)
        return out

    def gbuffer(self, cb):
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        lib().zro_gbuffer_render(self.h, cbb.ctypes.data, C.addressof(planes))
        return arrays, planes

    def pick(self, cb, x, y):
        """GBufferRT::PickPixel(x, y) + a G-buffer render: the mesh index under the pixel, 0xffffffff on a miss (GBufferRT_Inline.hlsl:241-242)"""
        from zetaray_amd import wire
        w, h = int(cb["render_width"]), int(cb["render_height"])
        arrays, planes = wire.alloc_gbuffer_planes(w, h)
        cbb = np.ascontiguousarray(cb)
        out = C.c_uint32(0xfffffffe)
        L = lib()
        L.zro_gbuffer_render_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zro_gbuffer_render_pick(self.h, cbb.ctypes.data, C.addressof(planes), int(x), int(y), C.byref(out))
        return int(out.value)

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


class OracleRPT:
    """Stateful ReSTIR PT renderer of the oracle (zro_rpt.h).  render() keeps the previous frame's G-buffer alive."""
    PLANES = {"A": (0, np.uint32, 1), "B": (1, np.float32, 2), "C": (2, np.uint32, 4), "D": (3, np.uint32, 4),
              "E": (4, np.uint16, 1), "F": (5, np.float32, 2), "G": (6, np.uint32, 2), "target": (7, np.float32, 4),
              "neighbor": (8, np.uint8, 2), "map_ctn": (10, np.uint16, 1), "map_ntc": (11, np.uint16, 1)}

    def __init__(self, oscene, w, h):
        from zetaray_amd import scene_io
        self.osc, self.w, self.h = oscene, w, h
        self.sample_set = scene_io.load_rpt_sample_set()
        self.r = lib().zro_rpt_create(w, h, self.sample_set.ctypes.data)
        self.prev = None          # (arrays, planes) of the previous frame
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zro_rpt_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zro_rpt_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        """gb: (arrays, planes) from OracleScene.gbuffer(cb); rendered here when None"""
        if gb is None:
            gb = self.osc.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        from zetaray_amd import wire
        cnt = wire.Counters()
        lib().zro_rpt_render(self.osc.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params),
                             self.final.ctypes.data, C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def self_shift(self, cb, params, which=0):
        out = np.zeros((self.h, self.w, 6), np.float32)
        cbb = np.ascontiguousarray(cb)
        lib().zro_rpt_self_shift(self.osc.h, self.r, cbb.ctypes.data, C.addressof(self.prev[1]), C.addressof(params), which, out.ctypes.data)
        return out

    def plane(self, name, which=0):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zro_rpt_read_plane(self.r, which, idx, out.ctypes.data)
        return out


class OracleRPTWindows:
    """The oracle's ReSTIR PT on scattered windows of ONE full-size frame (at-size parity, tests/window_parity.py): full-size planes and global pixel
    coordinates -- the oracle's passes unchanged but for the rectangle their loops visit (zro_rpt.h g_active) -- G-buffer rendered on window + apron,
    K11 / temporal / spatial stages on the owned window, the apron's reservoirs written in by the caller from the frame under test.  window(i) gives
    the per-window view with the interface of tests.hostexec.zhx.HostExecRPT (render_stage, plane, write_plane_rect, counters)."""
    HALO = {"A": (0, np.uint32, 1), "B": (1, np.float32, 2), "C": (2, np.uint32, 4), "D": (3, np.uint32, 4), "E": (4, np.uint16, 1), "F": (5, np.float32, 2), "G": (6, np.uint32, 2)}

    def __init__(self, oscene, W, H, windows):
        """windows: [(own, ext)], own / ext = (x0, y0, w, h) in frame coordinates, pairwise disjoint ext"""
        from zetaray_amd import scene_io, wire
        self.osc, self.W, self.H, self.windows = oscene, W, H, list(windows)
        self.sample_set = scene_io.load_rpt_sample_set()
        self.r = lib().zro_rpt_create(W, H, self.sample_set.ctypes.data)
        self.gb = [wire.alloc_gbuffer_planes(W, H), wire.alloc_gbuffer_planes(W, H)]
        self.gbi, self.frame_gb, self.have_prev = 0, None, False
        self.final = np.zeros((H, W, 4), np.float32)
        self.counters = [(0, 0)] * len(self.windows)
        # the physical reservoir set holding a window's post-temporal (between its stages) / final (after its stage 2) reservoirs: the windows share one
        # state and only the last window of a sweep commits the end-of-frame set flips, so the logical names are resolved here
        self.set_of = [[0, 0] for _ in self.windows]      # [window][which]: which = 1 post-temporal, 0 final
        L = lib()
        L.zro_gbuffer_render_rect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint32] * 4
        L.zro_rpt_render_stage.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_int]
        L.zro_rpt_curr_idx.argtypes = [C.c_void_p]
        L.zro_rpt_rw_plane_rect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_uint32] * 7 + [C.c_int]

    def __del__(self):
        if getattr(self, "r", None):
            lib().zro_rpt_destroy(self.r)
            self.r = None

    def _stage(self, i, cb, params, stage):
        from zetaray_amd import wire
        own, ext = self.windows[i]
        cbb = np.ascontiguousarray(cb)
        key = int(cb["frame_num"])
        if stage == 1:
            if self.frame_gb != key:      # first window of a new frame: the planes of the last frame become "previous"
                if self.frame_gb is not None:
                    self.gbi, self.have_prev = 1 - self.gbi, True
                self.frame_gb = key
            lib().zro_gbuffer_render_rect(self.osc.h, cbb.ctypes.data, C.addressof(self.gb[self.gbi][1]), ext[0], ext[1], ext[0] + ext[2], ext[1] + ext[3])
        prev = C.addressof(self.gb[1 - self.gbi][1]) if self.have_prev else None
        rect = (C.c_uint32 * 4)(own[0], own[1], own[0] + own[2], own[1] + own[3])
        cnt = wire.Counters()
        commit = 1 if (stage == 2 and i == len(self.windows) - 1) else 0
        before = lib().zro_rpt_curr_idx(self.r)
        after = lib().zro_rpt_render_stage(self.osc.h, self.r, cbb.ctypes.data, C.addressof(self.gb[self.gbi][1]), prev, C.addressof(params),
                                           self.final.ctypes.data, C.addressof(cnt), stage, rect, commit)
        if stage == 1:
            self.set_of[i][1] = before          # K11 and the temporal passes write the current set
        else:
            self.set_of[i][0] = 1 - after       # "the set the next frame reads as previous" once the flips are committed
        c = self.counters[i] if stage == 2 else (0, 0)
        self.counters[i] = (c[0] + cnt.n_closest, c[1] + cnt.n_shadow)

    def _rw(self, i, name, which, buf, rect_local, write):
        own, ext = self.windows[i]
        idx, dt, ch = self.HALO[name]
        x, y, w, h = rect_local
        lib().zro_rpt_rw_plane_rect(self.r, 2 + self.set_of[i][which], idx, buf.ctypes.data, ext[0], ext[1], ext[2], ext[0] + x, ext[1] + y, w, h, write)

    def window(self, i):
        return _OracleWindow(self, i)


class _OracleWindow:
    def __init__(self, parent, i):
        self.p, self.i = parent, i
        self.own, self.ext = parent.windows[i]

    @property
    def counters(self):
        return self.p.counters[self.i]

    def render_stage(self, cb, params, stage):
        """stage 1 = G-buffer on window + apron, K11 + temporal on the window; 2 = spatial + end of frame (the LAST window of the sweep commits it).
        Returns the full-size radiance array cropped to the extended rect."""
        self.p._stage(self.i, cb, params, stage)
        e = self.ext
        return self.p.final[e[1]:e[1] + e[3], e[0]:e[0] + e[2]]

    def plane(self, name, which=0):
        idx, dt, ch = self.p.HALO[name]
        out = np.zeros((self.ext[3], self.ext[2], ch), dt)
        self.p._rw(self.i, name, which, out, (0, 0, self.ext[2], self.ext[3]), 0)
        return out

    def write_plane_rect(self, name, which, full, rect_local):
        idx, dt, ch = self.p.HALO[name]
        self.p._rw(self.i, name, which, np.ascontiguousarray(full, dt), rect_local, 1)


class OracleRDI:
    """Stateful ReSTIR DI (emissive) renderer of the oracle (zro_rdi.h)."""
    PLANES = {"A": (0, np.uint32, 4), "B": (1, np.float32, 2), "target": (2, np.float32, 4)}

    def __init__(self, oscene, w, h):
        from zetaray_amd import scene_io
        self.osc, self.w, self.h = oscene, w, h
        self.sample_set = scene_io.load_rdi_sample_set()
        self.r = lib().zro_rdi_create(w, h, self.sample_set.ctypes.data)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zro_rdi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zro_rdi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.osc.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zro_rdi_render(self.osc.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params),
                             self.final.ctypes.data, C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zro_rdi_read_plane(self.r, idx, out.ctypes.data)
        return out


class OracleSDI:
    """Stateful sun + sky ReSTIR DI renderer of the oracle (zro_sdi.h).  The scene's sky LUT must be bound (sky_lut())."""
    PLANES = {"A": (0, np.uint8, 1), "B": (1, np.uint16, 2), "C": (2, np.float32, 2), "target": (3, np.float32, 4)}

    def __init__(self, oscene, w, h):
        self.osc, self.w, self.h = oscene, w, h
        self.r = lib().zro_sdi_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zro_sdi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zro_sdi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.osc.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zro_sdi_render(self.osc.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params),
                             self.final.ctypes.data, C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zro_sdi_read_plane(self.r, idx, out.ctypes.data)
        return out


class OracleRGI:
    """Stateful ReSTIR GI renderer of the oracle (zro_rgi.h)."""
    PLANES = {"A": (0, np.float32, 4), "B": (1, np.uint16, 4), "C": (2, np.float32, 4)}

    def __init__(self, oscene, w, h):
        self.osc, self.w, self.h = oscene, w, h
        self.r = lib().zro_rgi_create(w, h)
        self.prev = None
        self.final = np.zeros((h, w, 4), np.float32)

    def __del__(self):
        if getattr(self, "r", None):
            lib().zro_rgi_destroy(self.r)
            self.r = None

    def reset_temporal(self):
        lib().zro_rgi_reset_temporal(self.r)

    def render(self, cb, params, gb=None):
        from zetaray_amd import wire
        if gb is None:
            gb = self.osc.gbuffer(cb)
        cbb = np.ascontiguousarray(cb)
        prev = C.addressof(self.prev[1]) if self.prev is not None else None
        cnt = wire.Counters()
        lib().zro_rgi_render(self.osc.h, self.r, cbb.ctypes.data, C.addressof(gb[1]), prev, C.addressof(params),
                             self.final.ctypes.data, C.addressof(cnt))
        self.counters = (cnt.n_closest, cnt.n_shadow)
        self.prev = gb
        return self.final

    def plane(self, name):
        idx, dt, ch = self.PLANES[name]
        out = np.zeros((self.h, self.w, ch), dt)
        lib().zro_rgi_read_plane(self.r, idx, out.ctypes.data)
        return out


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


def composite(scene, cb, mr_plane, sky_di=None, emissive_di=None, indirect=None, out=None):
    """Compositing.hlsl main (in-scattering off).  mr_plane: the G-buffer's METALLIC_ROUGHNESS plane (h, w) u16 or (h, w, 2) u8; the DI / indirect
    terms are (h, w, 4) f32 or None; `out` = the composited texture's previous content (alpha is kept).  Miss pixels show
    Le_SkyWithSunDisk when a direct-lighting term is bound and the scene has a sky-view LUT."""
    mr = np.ascontiguousarray(mr_plane)
    h, w = mr.shape[:2]
    mr8 = mr.view(np.uint8).reshape(h, w, 2)
    planes = [None if p is None else np.ascontiguousarray(p, np.float32) for p in (sky_di, emissive_di, indirect)]
    o = np.zeros((h, w, 4), np.float32) if out is None else np.ascontiguousarray(out, np.float32).copy()
    cbb = np.ascontiguousarray(cb)
    f = lib().zro_composite
    f.argtypes = [C.c_void_p] * 7 + [C.c_uint32, C.c_uint32]
    f(scene.h, cbb.ctypes.data, mr8.ctypes.data, *[None if p is None else p.ctypes.data for p in planes], o.ctypes.data, w, h)
    return o


def _post_image(image):
    a = np.ascontiguousarray(image)
    assert a.ndim == 3 and a.shape[2] == 4 and a.dtype in (np.uint16, np.float32), "image: (h, w, 4) u16 (RGBA16F bits) or f32"
    return a, int(a.dtype == np.uint16)


def auto_exposure(image, params, dt, exposure2=None):
    """AutoExposure_Histogram.hlsl + AutoExposure_WeightedAvg.hlsl on an RGBA16F (u16 bits) or RGBA32F image.  `exposure2` = the
    persistent (exposure, adapted luminance) texel from the previous frame (zeros on the first).  Returns (hist[256] u32, exposure2 f32[2])."""
    a, is16 = _post_image(image)
    h, w = a.shape[:2]
    hist = np.zeros(256, np.uint32)
    e = np.zeros(2, np.float32) if exposure2 is None else np.array(exposure2, np.float32).copy()
    f = lib().zro_auto_exposure
    f.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    f(a.ctypes.data, is16, w, h, float(dt), C.addressof(params), hist.ctypes.data, e.ctypes.data)
    return hist, e


def display(image, params, display_size, exposure2=None, lut=None):
    """Display.hlsl mainPS (DisplayOption::DEFAULT) over display_size = (dw, dh) pixels.  Returns (rgba f32 (dh, dw, 4), srgb8 u8 (dh, dw, 4))."""
    a, is16 = _post_image(image)
    rh, rw = a.shape[:2]
    dw, dh = display_size
    out = np.zeros((dh, dw, 4), np.float32)
    srgb = np.zeros((dh, dw, 4), np.uint8)
    e = None if exposure2 is None else np.ascontiguousarray(exposure2, np.float32)
    l = None if lut is None else np.ascontiguousarray(lut, np.uint32)
    dim = 0 if l is None else int(round(l.size ** (1.0 / 3.0)))
    assert l is None or dim ** 3 == l.size
    f = lib().zro_display
    f.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    f(a.ctypes.data, is16, rw, rh, dw, dh, None if e is None else e.ctypes.data, C.addressof(params), None if l is None else l.ctypes.data, dim,
      out.ctypes.data, srgb.ctypes.data)
    return out, srgb


SVGF_DEFAULTS = dict(alpha=0.2, alpha_moments=0.2, sigma_l=4.0, sigma_z=1.0, normal_power_log2=7, iterations=5)


def svgf(signal_rgba, depth, normal, motion, prev_depth, prev_normal, hist_color, hist_moments, temporal_valid=True, **kw):
    """the denoise pass (zro_svgf.h; no reference counterpart) on one frame: signal (h, w, 4) f32, depth (h, w) f32, normal / motion (h, w) u32 (G-buffer
    planes), the previous frame's depth / normal, history colour (h, w, 4) f32 (rgb + length) and moments (h, w, 2) f32.
    Returns (out (h, w, 4) rgb + variance, new history colour, new history moments)."""
    prm = dict(SVGF_DEFAULTS, **kw)
    sig = np.ascontiguousarray(signal_rgba, np.float32)
    h, w = sig.shape[:2]
    d = np.ascontiguousarray(depth, np.float32); n = np.ascontiguousarray(normal, np.uint32); m = np.ascontiguousarray(motion, np.uint32)
    pd = np.ascontiguousarray(prev_depth, np.float32); pn = np.ascontiguousarray(prev_normal, np.uint32)
    hc = np.array(hist_color, np.float32, copy=True).reshape(h, w, 4); hm = np.array(hist_moments, np.float32, copy=True).reshape(h, w, 2)
    out = np.zeros((h, w, 4), np.float32)
    p4 = np.array([prm["alpha"], prm["alpha_moments"], prm["sigma_l"], prm["sigma_z"]], np.float32)
    f = lib().zro_svgf
    f.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    f(sig.ctypes.data, d.ctypes.data, n.ctypes.data, m.ctypes.data, pd.ctypes.data, pn.ctypes.data, hc.ctypes.data, hm.ctypes.data,
      int(bool(temporal_valid)), p4.ctypes.data, int(prm["normal_power_log2"]), int(prm["iterations"]), w, h, out.ctypes.data)
    return out, hc, hm


def taa(signal_rgba, depth, motion, prev_out, blend_weight=0.1, temporal_valid=True):
    """TAA.hlsl on an RGBA32F signal (h, w, 4), depth (h, w) f32, motion (h, w) u32 (R16G16_SNORM), history (h, w, 4) f16 bits (u16);
    returns the new RGBA16F output as u16 (alpha = the history buffer's, untouched)."""
    sig = np.ascontiguousarray(signal_rgba, np.float32)
    h, w = sig.shape[:2]
    d = np.ascontiguousarray(depth, np.float32)
    m = np.ascontiguousarray(motion, np.uint32)
    prev = np.ascontiguousarray(prev_out, np.uint16)
    out = np.zeros((h, w, 4), np.uint16)
    f = lib().zro_taa
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int]
    f(sig.ctypes.data, d.ctypes.data, m.ctypes.data, prev.ctypes.data, out.ctypes.data, w, h, float(blend_weight), int(bool(temporal_valid)))
    return out
