"""ctypes binding of libzetaray_amd.so (the C-ABI in include/zetaray_amd.h) plus thin Python mirrors of the
reference's pass surface (Init / OnWindowResized / ResetTemporal / Render / GetOutput).

Host plumbing only: everything that computes happens inside the HIP library.  The library has no CPU path -- every
compute entry point raises ZetaRayError (ZR_ERR_NO_DEVICE) when no MI355X is visible, and importing this module fails
loudly if libzetaray_amd.so has not been built (python -c "import __graft_entry__ as g; g.build()").
"""
import ctypes as C
import os

import numpy as np

from . import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZETARAY_AMD_LIB", os.path.join(_HERE, "libzetaray_amd.so"))     # (override: compiler-variant experiments)

PASS_GBUFFER, PASS_PRELIGHTING, PASS_DI_EMISSIVE, PASS_DI_SKY, PASS_INDIRECT, PASS_COMPOSITING, PASS_SKY, PASS_TAA, PASS_AUTO_EXPOSURE, PASS_DISPLAY, PASS_DENOISE = range(11)
IN_TAA_SIGNAL, OUT_TAA = 3, 41
IN_POST_SIGNAL_F16, IN_POST_SIGNAL_F32, IN_DISPLAY_EXPOSURE = 4, 5, 6
OUT_EXPOSURE, OUT_AE_HISTOGRAM, OUT_DISPLAY, OUT_DISPLAY_SRGB8 = 42, 43, 44, 45
IN_DENOISE_SIGNAL, OUT_DENOISED, OUT_DENOISE_HISTORY, OUT_DENOISE_MOMENTS = 7, 46, 47, 48
TONEMAP_LUT_PATH = os.path.join(_HERE, "assets", "tony_mc_mapface_rgb9e5.bin")


def load_tonemap_lut():
    """The Tony McMapface LUT of the NEUTRAL tone mapper: 48^3 R9G9B9E5_SHAREDEXP texels (tools/extract_assets.py)."""
    lut = np.fromfile(TONEMAP_LUT_PATH, np.uint32)
    assert lut.size == 48 ** 3, lut.size
    return lut

OUT_SKY_LUT = 40
IN_EMISSIVE_DI, IN_INDIRECT, IN_SKY_DI = range(3)
INTEGRATOR_PATH_TRACING, INTEGRATOR_RESTIR_GI, INTEGRATOR_RESTIR_PT = range(3)
OUT_FINAL = 0
# ReSTIR PT persistent state (zr_output): name -> (id, dtype, channels)
RPT_OUTPUTS_EXTRA = {"denoised": (46, np.float32, 4), "denoise_history": (47, np.float32, 4), "denoise_moments": (48, np.float32, 2),
                     "taa": (41, np.uint16, 4), "sky_lut": (40, np.uint32, 1), "sdi_A": (24, np.uint8, 1), "sdi_B": (25, np.uint16, 2), "sdi_C": (26, np.float32, 2),
                     "sdi_target": (27, np.float32, 4)}
RPT_OUTPUTS = {"A": (1, np.uint32, 1), "B": (2, np.float32, 2), "C": (3, np.uint32, 4), "D": (4, np.uint32, 4),
               "E": (5, np.uint16, 1), "F": (6, np.float32, 2), "G": (7, np.uint32, 2), "target": (8, np.float32, 4),
               "neighbor": (9, np.uint8, 2), "map_ctn": (18, np.uint16, 1), "map_ntc": (19, np.uint16, 1),
               "ctn_A": (10, np.uint16, 4), "ctn_B": (11, np.uint32, 4), "ctn_C": (12, np.uint32, 4), "ctn_D": (13, np.uint16, 1),
               "gi_A": (30, np.float32, 4), "gi_B": (31, np.uint16, 4), "gi_C": (32, np.float32, 4),
               "di_A": (20, np.uint32, 4), "di_B": (21, np.float32, 2), "di_target": (22, np.float32, 4),
               "ntc_A": (14, np.uint16, 4), "ntc_B": (15, np.uint32, 4), "ntc_C": (16, np.uint32, 4), "ntc_D": (17, np.uint16, 1)}

EXPORTS = [
    "zr_abi_version", "zr_last_error", "zr_device_count", "zr_device_probe_run", "zr_wire_layout", "zr_scene_create", "zr_scene_destroy", "zr_scene_update_instances", "zr_scene_update_emissives", "zr_scene_invalidate_alias_table", "zr_scene_update_materials",
    "zr_scene_set_background_rebuild", "zr_scene_background_rebuild_stats",
    "zr_scene_invalidate_alias_table_deferred", "zr_scene_update_instances_async", "zr_scene_update_emissives_async", "zr_scene_update_materials_async", "zr_scene_set_alias_table_async",
    "zr_scene_set_alias_table", "zr_alias_table_build", "zr_scene_get_alias_table", "zr_scene_get_light_voxel_grid", "zr_scene_get_presampled_sets", "zr_scene_bvh_info", "zr_pass_enable_cost_map", "zr_pass_read_cost_map", "zr_pass_debug_trip_stats", "zr_debug_set_large_scene_nodes", "zr_debug_set_bvh_depth_cap", "zr_debug_set_material_class_kernels", "zr_scene_material_class",
    "zr_gbuffer_create", "zr_gbuffer_destroy", "zr_gbuffer_set_tile_origin", "zr_gbuffer_download", "zr_gbuffer_device_plane",
    "zr_params_default", "zr_pass_create", "zr_pass_init", "zr_pass_resize", "zr_pass_reset_temporal", "zr_pass_pick_pixel", "zr_pass_clear_pick", "zr_pass_read_pick",
    "zr_pass_set_params", "zr_pass_render", "zr_pass_get_output", "zr_pass_download_output",
    "zr_pass_read_counters", "zr_pass_read_kernel_counters", "zr_pass_enable_timing", "zr_pass_get_timings", "zr_selftest_half_conversions", "zr_pass_destroy",
    "zr_trace_closest", "zr_trace_any",
    "zr_pass_set_owned_rect", "zr_pass_render_stage", "zr_pass_halo_pack", "zr_pass_halo_unpack", "zr_pass_halo_bytes_per_pixel", "zr_pass_set_input",
    "zr_pass_set_tonemap_lut", "zr_pass_halo_pack_all", "zr_pass_halo_unpack_all",
    "zr_pass_set_frame_overlap", "zr_pass_frame_overlap_stream", "zr_device_synchronize",
]
STAGE_TEMPORAL, STAGE_SPATIAL, STAGE_ALL = 1, 2, 3
STAGE_SPATIAL2 = 4          # ReSTIR PT, num_spatial_passes = 2 on tiles: the second round, behind one more HALO_POST_TEMPORAL exchange
STAGE_CANDIDATES, STAGE_TEMPORAL_REUSE = 8, 16      # ReSTIR PT: the TEMPORAL stage in its two halves (K11 / K12-K14): what frame overlap puts on two streams
HALO_POST_TEMPORAL, HALO_FINAL = 0, 1
HALO_DENOISE_INPUT, HALO_DENOISE_ITER = 2, 3                  # ZR_PASS_DENOISE on tiles (tiling.denoise_schedule): 40 / 16 B per pixel
STAGE_DENOISE_TEMPORAL, STAGE_DENOISE_VARIANCE, STAGE_DENOISE_MASK = 1 << 8, 1 << 9, 0x3ff00


def stage_denoise_atrous(i):
    return 1 << (10 + i)

HALO_BYTES_PER_PIXEL = 62


class ZetaRayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zetaray_amd error {code}: {msg}")
        self.code = code


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first "
                              f"(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 (same SONAME as the system
        # one this library links).  Whichever is loaded first serves both, and torch only finds its GPUs through its
        # own copy -- so torch goes first; libzetaray_amd.so then binds to the already-loaded runtime, which also makes
        # torch.cuda.synchronize() / torch streams and this library's launches share one device context.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.zr_last_error.restype = C.c_char_p
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.zr_device_count.argtypes = [vp]
        L.zr_scene_create.argtypes = [i32, vp, vp]
        L.zr_scene_destroy.argtypes = [vp]
        L.zr_scene_set_alias_table.argtypes = [vp, vp, u32]
        L.zr_alias_table_build.argtypes = [vp, u32, u32, vp]
        L.zr_scene_get_alias_table.argtypes = [vp, vp, u32]
        L.zr_scene_get_light_voxel_grid.argtypes = [vp, vp, vp, u32]
        L.zr_scene_get_presampled_sets.argtypes = [vp, vp, vp, u32]
        L.zr_scene_bvh_info.argtypes = [vp, vp, vp, vp]
        L.zr_gbuffer_create.argtypes = [i32, u32, u32, vp]
        L.zr_gbuffer_destroy.argtypes = [vp]
        L.zr_gbuffer_set_tile_origin.argtypes = [vp, u32, u32]
        L.zr_gbuffer_download.argtypes = [vp, vp, vp]
        L.zr_gbuffer_device_plane.argtypes = [vp, i32, vp]
        L.zr_params_default.argtypes = [vp]
        L.zr_pass_create.argtypes = [i32, i32, vp]
        L.zr_pass_init.argtypes = [vp, u32, u32, i32]
        L.zr_pass_resize.argtypes = [vp, u32, u32]
        L.zr_pass_reset_temporal.argtypes = [vp]
        L.zr_pass_set_params.argtypes = [vp, vp]
        L.zr_pass_render.argtypes = [vp, vp, vp, vp, vp]
        L.zr_pass_get_output.argtypes = [vp, i32, vp, vp, vp, vp]
        L.zr_pass_download_output.argtypes = [vp, i32, vp, vp, C.c_size_t]
        L.zr_pass_read_counters.argtypes = [vp, vp, vp, i32]
        L.zr_pass_read_kernel_counters.argtypes = [vp, vp, u32, vp, vp, vp, vp]
        L.zr_pass_set_input.argtypes = [vp, i32, vp]
        L.zr_pass_set_tonemap_lut.argtypes = [vp, vp, u32]
        L.zr_pass_halo_pack_all.argtypes = [vp, vp, vp, i32, vp, u32, vp, C.c_size_t]
        L.zr_pass_halo_unpack_all.argtypes = [vp, vp, vp, i32, vp, u32, vp, C.c_size_t]
        L.zr_pass_set_owned_rect.argtypes = [vp, u32, u32, u32, u32]
        L.zr_pass_render_stage.argtypes = [vp, vp, vp, vp, vp, i32]
        L.zr_pass_halo_pack.argtypes = [vp, vp, vp, i32, u32, u32, u32, u32, vp, C.c_size_t]
        L.zr_pass_halo_bytes_per_pixel.argtypes = [vp, vp]
        L.zr_pass_halo_unpack.argtypes = [vp, vp, vp, i32, u32, u32, u32, u32, vp, C.c_size_t]
        L.zr_pass_enable_timing.argtypes = [vp, i32]
        L.zr_pass_get_timings.argtypes = [vp, u32, vp, vp, vp, vp]
        L.zr_pass_destroy.argtypes = [vp]
        L.zr_pass_set_frame_overlap.argtypes = [vp, vp, i32]
        L.zr_pass_frame_overlap_stream.argtypes = [vp, vp]
        L.zr_device_synchronize.argtypes = [i32]
        L.zr_trace_closest.argtypes = [vp, vp, vp, u32, u32, vp]
        L.zr_trace_any.argtypes = [vp, vp, vp, u32, u32, vp]
        _LIB = L
    return _LIB


def _check(code):
    if code != 0:
        raise ZetaRayError(code, lib().zr_last_error().decode())


def device_count():
    n = C.c_int(0)
    _check(lib().zr_device_count(C.byref(n)))
    return n.value


def alias_table_build(power, align_phase=0):
    """zr_alias_table_build: host-side Vose build of the reference (PreLighting.cpp:27-158); needs no GPU."""
    power = np.ascontiguousarray(power, np.float32)
    out = np.zeros(len(power), wire.ALIAS_ENTRY)
    _check(lib().zr_alias_table_build(power.ctypes.data, len(power), align_phase, out.ctypes.data))
    return out


class Scene:
    def __init__(self, scene, device=0):
        self.host = scene
        self._desc = scene.desc()
        self.h = C.c_void_p()
        self.version = 0      # bumped by every update_*: lets a caller see whether anything moved since it last looked (tiling.TiledRestirPT)
        # do the instance records now on the device describe motion (prev transform != current)?  They keep doing so in every later frame until
        # the host uploads records at rest, whether or not update_instances is called again -- the G-buffer's motion vectors come from them
        self.instances_in_motion = self._records_in_motion(getattr(scene, "instances", None))
        _check(lib().zr_scene_create(device, C.addressof(self._desc), C.byref(self.h)))

    @staticmethod
    def _records_in_motion(instances):
        """MeshInstance records (wire.MESH_INSTANCE) whose previous transform differs from the current one: dTranslation (half3) non-zero, or
        PrevRotation / PrevScale != Rotation / Scale (RtCommon.h:47-64)"""
        if instances is None or len(instances) == 0:
            return False
        i = np.asarray(instances)
        return bool(((i["d_translation"] & 0x7fff) != 0).any() or (i["prev_rotation"] != i["rotation"]).any() or (i["prev_scale"] != i["scale"]).any())

    def update_instances(self, instances, instance_to_world, stream=False):
        """per-frame MeshInstance records + object-to-world matrices; last frame's instance buffer and BVH become the previous ones.
        stream=False: host-synchronous; a stream handle (or None = the null stream): enqueued, no host wait (zr_scene_update_instances_async)"""
        L = lib()
        self.version += 1
        i, x = np.ascontiguousarray(instances), np.ascontiguousarray(instance_to_world, np.float32)
        self.instances_in_motion = self._records_in_motion(i)
        if stream is False:
            L.zr_scene_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
            _check(L.zr_scene_update_instances(self.h, i.ctypes.data, x.ctypes.data, len(i)))
        else:
            L.zr_scene_update_instances_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
            _check(L.zr_scene_update_instances_async(self.h, stream, i.ctypes.data, x.ctypes.data, len(i)))

    def set_background_rebuild(self, on=True):
        """dynamic scenes: rebuild the SAH tree on a host thread for the instances' current transforms and swap it in at a later update_instances
        (zr_scene_set_background_rebuild); the device refit keeps running every frame in between"""
        lib().zr_scene_set_background_rebuild.argtypes = [C.c_void_p, C.c_int]
        _check(lib().zr_scene_set_background_rebuild(self.h, int(bool(on))))

    def background_rebuild_stats(self):
        """(builds started, builds installed, state: 0 idle / 1 building / 2 built, waiting for the next update)"""
        a, b, st = C.c_uint64(), C.c_uint64(), C.c_int()
        lib().zr_scene_background_rebuild_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib().zr_scene_background_rebuild_stats(self.h, C.byref(a), C.byref(b), C.byref(st)))
        return a.value, b.value, st.value

    def update_emissives(self, triangles, first=0, stream=False):
        """new EmissiveTriangle records for [first, first + len(triangles)) (instances that carry lights moved)"""
        L = lib()
        self.version += 1
        t = np.ascontiguousarray(triangles, wire.EMISSIVE_TRI)
        if stream is False:
            L.zr_scene_update_emissives.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
            _check(L.zr_scene_update_emissives(self.h, t.ctypes.data, first, len(t)))
        else:
            L.zr_scene_update_emissives_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
            _check(L.zr_scene_update_emissives_async(self.h, stream, t.ctypes.data, first, len(t)))

    def update_materials(self, materials, first=0, stream=False):
        """rewritten Material records for [first, first + len(materials))"""
        L = lib()
        self.version += 1
        m = np.ascontiguousarray(materials, wire.MATERIAL)
        if stream is False:
            L.zr_scene_update_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
            _check(L.zr_scene_update_materials(self.h, m.ctypes.data, first, len(m)))
        else:
            L.zr_scene_update_materials_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
            _check(L.zr_scene_update_materials_async(self.h, stream, m.ctypes.data, first, len(m)))

    def material_class(self):
        """1 when every material is an opaque, uncoated, non-metallic dielectric and the scene has no textures (the PLAIN kernel permutations), else 0"""
        out = C.c_uint32()
        lib().zr_scene_material_class.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib().zr_scene_material_class(self.h, C.byref(out)))
        return out.value

    def invalidate_alias_table_deferred(self):
        """the reference's steady-state form: the old table is sampled until the rebuilt one has been uploaded; no render call waits"""
        lib().zr_scene_invalidate_alias_table_deferred.argtypes = [C.c_void_p]
        _check(lib().zr_scene_invalidate_alias_table_deferred(self.h))

    def invalidate_alias_table(self):
        """emissive materials changed: the next PRELIGHTING render re-estimates the powers and rebuilds the alias table"""
        lib().zr_scene_invalidate_alias_table.argtypes = [C.c_void_p]
        _check(lib().zr_scene_invalidate_alias_table(self.h))

    def set_alias_table(self, entries):
        entries = np.ascontiguousarray(entries)
        _check(lib().zr_scene_set_alias_table(self.h, entries.ctypes.data, len(entries)))

    def get_alias_table(self):
        out = np.zeros(len(self.host.emissives), wire.ALIAS_ENTRY)
        _check(lib().zr_scene_get_alias_table(self.h, out.ctypes.data, len(out)))
        return out

    def get_light_voxel_grid(self, dim, stream=None):
        out = np.zeros((dim[2], dim[1], dim[0], 64), wire.VOXEL_SAMPLE)
        _check(lib().zr_scene_get_light_voxel_grid(self.h, stream, out.ctypes.data, out.size))
        return out

    def get_presampled_sets(self, num_sets, set_size, stream=None):
        out = np.zeros(num_sets * set_size, wire.PRESAMPLED_TRI)
        _check(lib().zr_scene_get_presampled_sets(self.h, stream, out.ctypes.data, len(out)))
        return out

    def bvh_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().zr_scene_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self.h:
            lib().zr_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GBuffer:
    def __init__(self, width, height, device=0):
        self.w, self.h_ = width, height
        self.h = C.c_void_p()
        _check(lib().zr_gbuffer_create(device, width, height, C.byref(self.h)))

    def set_tile_origin(self, x0, y0):
        _check(lib().zr_gbuffer_set_tile_origin(self.h, x0, y0))

    def download(self, stream=None):
        arrays, planes = wire.alloc_gbuffer_planes(self.w, self.h_)
        _check(lib().zr_gbuffer_download(self.h, stream, C.addressof(planes)))
        return arrays, planes

    def close(self):
        if self.h:
            lib().zr_gbuffer_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pass:
    """One RenderPass node (GBufferRT / PreLighting / IndirectLighting) behind the C-ABI."""

    def __init__(self, kind, width, height, integrator=INTEGRATOR_PATH_TRACING, device=0, params=None):
        self.kind = kind
        self.w, self.h_ = width, height
        self.h = C.c_void_p()
        _check(lib().zr_pass_create(kind, device, C.byref(self.h)))
        _check(lib().zr_pass_init(self.h, width, height, integrator))
        if params is not None:
            self.set_params(params)

    def set_params(self, params):
        _check(lib().zr_pass_set_params(self.h, C.addressof(params)))

    def resize(self, width, height):
        _check(lib().zr_pass_resize(self.h, width, height))
        self.w, self.h_ = width, height

    def reset_temporal(self):
        _check(lib().zr_pass_reset_temporal(self.h))

    # GBUFFER pass: GBufferRT::PickPixel / ClearPick / the pick read-back (GBufferRT.h:36-46)
    def pick_pixel(self, x, y):
        L = lib()
        L.zr_pass_pick_pixel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        _check(L.zr_pass_pick_pixel(self.h, int(x), int(y)))

    def clear_pick(self):
        L = lib()
        L.zr_pass_clear_pick.argtypes = [C.c_void_p]
        _check(L.zr_pass_clear_pick(self.h))

    def read_pick(self, stream=None):
        """mesh index written by the last G-buffer render over the picked pixel (0xffffffff: the primary ray missed)"""
        L = lib()
        L.zr_pass_read_pick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        out = C.c_uint32(0)
        _check(L.zr_pass_read_pick(self.h, stream, C.byref(out)))
        return int(out.value)

    def render(self, cb, scene, gbuffer=None, stream=None):
        cbb = np.ascontiguousarray(cb)
        _check(lib().zr_pass_render(self.h, stream, cbb.ctypes.data, scene.h, gbuffer.h if gbuffer is not None else None))

    def render_stage(self, cb, scene, gbuffer, stages, stream=None):
        cbb = np.ascontiguousarray(cb)
        _check(lib().zr_pass_render_stage(self.h, stream, cbb.ctypes.data, scene.h, gbuffer.h if gbuffer is not None else None, stages))

    def set_input(self, which, dev_ptr):
        _check(lib().zr_pass_set_input(self.h, which, dev_ptr))

    def set_frame_overlap(self, gbuffer, on=1):
        """ReSTIR PT: keep a third reservoir set + second target / FINAL planes so that this frame's CANDIDATES stage may run beside the previous frame's
        reuse stages on another stream (zetaray_amd.h zr_pass_set_frame_overlap); the G-buffer becomes stream-tracked"""
        _check(lib().zr_pass_set_frame_overlap(self.h, gbuffer.h, int(on)))      # 0 off, 1 = ZR_FRAME_OVERLAP, 2 = ZR_FRAME_OVERLAP_CARRY

    def frame_overlap_stream(self):
        """the pass-owned non-blocking stream for the GBUFFER / PRELIGHTING / CANDIDATES half of an overlapped frame (a hipStream_t as an integer)"""
        st = C.c_void_p()
        _check(lib().zr_pass_frame_overlap_stream(self.h, C.byref(st)))
        return st.value

    def set_tonemap_lut(self, lut=None):
        lut = np.ascontiguousarray(load_tonemap_lut() if lut is None else lut, np.uint32)
        dim = int(round(lut.size ** (1.0 / 3.0)))
        _check(lib().zr_pass_set_tonemap_lut(self.h, lut.ctypes.data, dim))

    def download_raw(self, which, dtype, shape, stream=None):
        out = np.zeros(shape, dtype)
        _check(lib().zr_pass_download_output(self.h, which, stream, out.ctypes.data, out.nbytes))
        return out

    def set_owned_rect(self, x0, y0, w, h):
        _check(lib().zr_pass_set_owned_rect(self.h, x0, y0, w, h))

    def halo_pack(self, gbuffer, which, rect, dev_ptr, nbytes, stream=None):
        _check(lib().zr_pass_halo_pack(self.h, stream, gbuffer.h, which, rect[0], rect[1], rect[2], rect[3], dev_ptr, nbytes))

    def halo_all(self, gbuffer, which, rects, dev_ptr, nbytes, pack=True, stream=None):
        """fused transfer: rects = [(x0, y0, w, h, byte offset into the buffer)]"""
        class Rect(C.Structure):
            _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32), ("offset", C.c_uint64)]
        arr = (Rect * max(1, len(rects)))(*[Rect(*r) for r in rects])
        f = lib().zr_pass_halo_pack_all if pack else lib().zr_pass_halo_unpack_all
        _check(f(self.h, stream, gbuffer.h, which, arr, len(rects), dev_ptr, nbytes))

    def halo_bytes_per_pixel(self):
        b = C.c_uint32()
        _check(lib().zr_pass_halo_bytes_per_pixel(self.h, C.byref(b)))
        return b.value

    def halo_unpack(self, gbuffer, which, rect, dev_ptr, nbytes, stream=None):
        _check(lib().zr_pass_halo_unpack(self.h, stream, gbuffer.h, which, rect[0], rect[1], rect[2], rect[3], dev_ptr, nbytes))

    def output_ptr(self, which=OUT_FINAL):
        dev = C.c_void_p()
        w, h, bpp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().zr_pass_get_output(self.h, which, C.byref(dev), C.byref(w), C.byref(h), C.byref(bpp)))
        return dev.value, w.value, h.value, bpp.value

    def download(self, which=OUT_FINAL, stream=None):
        out = np.zeros((self.h_, self.w, 4), np.float32)
        _check(lib().zr_pass_download_output(self.h, which, stream, out.ctypes.data, out.nbytes))
        return out

    def download_plane(self, name, stream=None):
        """ReSTIR PT reservoir / target / neighbour planes (see RPT_OUTPUTS)."""
        which, dt, ch = RPT_OUTPUTS[name] if name in RPT_OUTPUTS else RPT_OUTPUTS_EXTRA[name]
        out = np.zeros((self.h_, self.w, ch), dt)
        _check(lib().zr_pass_download_output(self.h, which, stream, out.ctypes.data, out.nbytes))
        return out

    def enable_cost_map(self, on=True):
        lib().zr_pass_enable_cost_map.argtypes = [C.c_void_p, C.c_int]
        _check(lib().zr_pass_enable_cost_map(self.h, int(on)))

    def read_cost_map(self, reset=True, stream=None):
        """(cells_h, cells_w) uint32: rays per 32 x 32-px cell of this pass's planes since the last reset"""
        cw, ch = (self.w + 31) // 32 + 1, (self.h_ + 31) // 32 + 1
        out = np.zeros((ch, cw), np.uint32)
        lib().zr_pass_read_cost_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        _check(lib().zr_pass_read_cost_map(self.h, stream, out.ctypes.data, cw, ch, int(reset)))
        return out

    def debug_trip_stats(self):
        """ZR_K11=trip diagnostic: (lanes alive at K11's bounce boundaries, lane slots of the waves that passed them, 32-bit words per path state)"""
        out = (C.c_uint64 * 3)()
        lib().zr_pass_debug_trip_stats.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib().zr_pass_debug_trip_stats(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def read_counters(self, reset=True, stream=None):
        c = wire.Counters()
        _check(lib().zr_pass_read_counters(self.h, stream, C.addressof(c), int(reset)))
        return c.n_closest, c.n_shadow

    def kernel_counters(self, stream=None):
        """{kernel name: (closest-hit queries, shadow queries)} accumulated since the last read_counters(reset=True)."""
        n = 16
        names = (C.c_char_p * n)()
        a = (C.c_uint64 * n)()
        b = (C.c_uint64 * n)()
        cnt = C.c_uint32()
        _check(lib().zr_pass_read_kernel_counters(self.h, stream, n, names, a, b, C.byref(cnt)))
        return {names[i].decode(): (a[i], b[i]) for i in range(cnt.value)}

    def enable_timing(self, on=True):
        _check(lib().zr_pass_enable_timing(self.h, int(on)))

    def timings(self):
        n = 64
        names = (C.c_char_p * n)()
        ms = (C.c_float * n)()
        launches = (C.c_uint32 * n)()
        cnt = C.c_uint32()
        _check(lib().zr_pass_get_timings(self.h, n, names, ms, launches, C.byref(cnt)))
        return {names[i].decode(): (ms[i], launches[i]) for i in range(min(cnt.value, n))}

    def close(self):
        if self.h:
            lib().zr_pass_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """Minimal host restatement of DefaultRenderer's per-frame order for the hot path
    (Source/ZetaRenderer/Default/PathTracer.cpp:325-563): GBuffer -> PreLighting (+ alias table) -> Indirect."""

    def __init__(self, scene_host, width, height, device=0, params=None, integrator=INTEGRATOR_PATH_TRACING,
                 tile_origin=(0, 0)):
        """width/height = size of the tile this renderer owns; tile_origin = its top-left pixel in the full target."""
        self.scene = Scene(scene_host, device)
        self.scene_host = scene_host
        self.gbuffer = GBuffer(width, height, device)
        if tile_origin != (0, 0):
            self.gbuffer.set_tile_origin(*tile_origin)
        self.p_gbuffer = Pass(PASS_GBUFFER, width, height, device=device)
        self.p_prelight = Pass(PASS_PRELIGHTING, width, height, device=device, params=params)
        self._presampling = bool(params is not None and params.presampling)
        self.p_indirect = Pass(PASS_INDIRECT, width, height, integrator, device=device, params=params)
        # Sky pass (K17): scenes without emissive triangles light with sun + sky, which sample the sky-view LUT
        self.p_sky = Pass(PASS_SKY, 256, 128, device=device) if len(scene_host.emissives) == 0 else None
        self.p_direct = None          # ReSTIR DI (emissive): enable_direct()
        self.p_sky_direct = None      # ReSTIR DI (sun + sky): enable_sky_direct()
        self.skip_indirect = False
        self.p_composit = None        # Compositing: enable_compositing()
        self._alias_ready = False
        self._overlap_stream = None   # frame overlap: enable_frame_overlap()
        self._device = device

    def enable_frame_overlap(self, on=True, carry=False):
        """(carry: ZR_FRAME_OVERLAP_CARRY -- the unused bytes of the reservoir records are carried over too, so every plane equals the plain order's byte
        for byte: what the parity tests compare with the oracle; costs one streaming copy per frame)
        Software-pipeline consecutive ReSTIR PT frames on two streams: the G-buffer, PreLighting and K11 of frame N + 1 go to a stream of the pass's
        own and run beside the search / sort / replay / reconnect kernels of frame N (the reference overlaps its direct and async-compute queues the same
        way, RenderGraph.cpp:442-541).  Bit-identical frames; throughput goes up, the latency of one frame does not go down.  Indirect (ReSTIR PT) + denoise
        only: the DI passes, Compositing and TAA keep single-buffered outputs that the next frame's first half would overwrite under their consumers."""
        assert self.p_direct is None and self.p_sky_direct is None and self.p_composit is None and getattr(self, "p_taa", None) is None, \
            "frame overlap covers GBuffer + PreLighting + Indirect (ReSTIR PT) [+ denoise]"
        _check(lib().zr_device_synchronize(self._device))
        self.p_indirect.set_frame_overlap(self.gbuffer, (2 if carry else 1) if on else 0)
        self._overlap_stream = self.p_indirect.frame_overlap_stream() if on else None

    def enable_compositing(self, device=0, firefly_filter=False):
        """add the Compositing pass: (DI + indirect * !emissive) / NumFramesCameraStatic, optionally followed by the firefly filter"""
        prm = None
        if firefly_filter:
            prm = wire.default_params()
            prm.flags |= wire.COMPOSIT_FIREFLY_FILTER
        self.p_composit = Pass(PASS_COMPOSITING, self.p_indirect.w, self.p_indirect.h_, device=device, params=prm)
        self._bind_post_inputs()
        return self.p_composit

    def _bind_post_inputs(self):
        """(re)bind the Compositing / TAA inputs to the current output planes of the lighting passes: called whenever a pass is
        added and at the start of every frame (output pointers change on Pass.resize; a term whose pass is absent stays unbound,
        which CompositePixel treats as 'term absent')"""
        if self.p_composit is None:
            return
        self.p_composit.set_input(IN_INDIRECT, None if self.skip_indirect else self.p_indirect.output_ptr()[0])
        self.p_composit.set_input(IN_EMISSIVE_DI, self.p_direct.output_ptr()[0] if self.p_direct is not None else None)
        self.p_composit.set_input(IN_SKY_DI, self.p_sky_direct.output_ptr()[0] if self.p_sky_direct is not None else None)
        if getattr(self, "p_taa", None) is not None:
            self.p_taa.set_input(IN_TAA_SIGNAL, self.p_composit.output_ptr()[0])

    def enable_denoise(self, params=None, device=0):
        """add the denoise pass (ZR_PASS_DENOISE: spatiotemporal variance-guided filter; no reference counterpart) on the indirect pass's FINAL image;
        it renders after the lighting passes, before Compositing reads anything (Compositing keeps reading the unfiltered planes).  Read the
        result with p_denoise.download_plane("denoised") (RGBA32F: rgb + variance)."""
        self.p_denoise = Pass(PASS_DENOISE, self.p_indirect.w, self.p_indirect.h_, device=device, params=params or wire.default_params())
        self.p_denoise.set_input(IN_DENOISE_SIGNAL, self.p_indirect.output_ptr()[0])
        return self.p_denoise

    def enable_taa(self, blend_weight=0.1, device=0):
        """add the TAA pass on the composited image (adds the Compositing pass if it is not there yet); read it with
        p_taa.download_plane("taa") (RGBA16F bits)"""
        if self.p_composit is None:
            self.enable_compositing(device=device)
        prm = wire.default_params()
        prm.taa_blend_weight = blend_weight
        self.p_taa = Pass(PASS_TAA, self.p_indirect.w, self.p_indirect.h_, device=device, params=prm)
        self._bind_post_inputs()
        return self.p_taa

    def enable_sky_direct(self, params=None, device=0):
        """add the SkyDI (sun + sky ReSTIR DI) pass; it renders after the Sky pass and the G-buffer"""
        if self.p_sky is None:
            self.p_sky = Pass(PASS_SKY, 256, 128, device=device)
        self.p_sky_direct = Pass(PASS_DI_SKY, self.p_indirect.w, self.p_indirect.h_, device=device, params=params)
        self._bind_post_inputs()
        return self.p_sky_direct

    def enable_direct(self, params=None, device=0):
        """add the DirectLighting (ReSTIR DI, emissive) pass; it renders after PreLighting, next to Indirect"""
        self.p_direct = Pass(PASS_DI_EMISSIVE, self.p_indirect.w, self.p_indirect.h_, device=device, params=params)
        self._bind_post_inputs()
        return self.p_direct

    def invalidate_alias_table(self, deferred=False):
        """emissive materials changed (records already handed to scene.update_emissives): PRELIGHTING re-estimates powers + rebuilds the table.
        deferred: the reference's steady-state form -- the old table is sampled until the new one has arrived, no render call waits"""
        if deferred:
            self.scene.invalidate_alias_table_deferred()
            self._alias_poll = 8          # PRELIGHTING renders for the next frames: the first starts the read-back, a later one picks it up
            return
        self.scene.invalidate_alias_table()
        self._alias_ready = False

    _SKY_FIELDS = ("sun_dir", "sun_illuminance", "planet_radius", "atmosphere_altitude", "g", "rayleigh_sigma_s_color", "rayleigh_sigma_s_scale",
                   "ozone_sigma_a_color", "ozone_sigma_a_scale", "mie_sigma_s", "mie_sigma_a")

    def render_sky(self, cb, stream=None):
        """K17 only when its inputs (sun, atmosphere) changed: the LUT does not depend on the camera"""
        if self.p_sky is None:
            return
        key = b"".join(np.asarray(cb[f]).tobytes() for f in self._SKY_FIELDS)
        if key != getattr(self, "_sky_key", None):
            self.p_sky.render(cb, self.scene, None, stream)
            self._sky_key = key

    def render_frame(self, cb, stream=None):
        if self._overlap_stream is not None:
            return self._render_frame_overlapped(cb, stream)
        self._bind_post_inputs()
        self.render_sky(cb, stream)
        self.p_gbuffer.render(cb, self.scene, self.gbuffer, stream)
        if not self._alias_ready or self._presampling or getattr(self, "_alias_poll", 0) > 0:      # presampled light sets are regenerated every frame (K3)
            self.p_prelight.render(cb, self.scene, None, stream)
            self._alias_ready = True
            self._alias_poll = max(0, getattr(self, "_alias_poll", 0) - 1)
        if self.p_direct is not None:
            self.p_direct.render(cb, self.scene, self.gbuffer, stream)
        if self.p_sky_direct is not None:
            self.p_sky_direct.render(cb, self.scene, self.gbuffer, stream)
        if not self.skip_indirect:
            self.p_indirect.render(cb, self.scene, self.gbuffer, stream)
        if getattr(self, "p_denoise", None) is not None:
            self.p_denoise.set_input(IN_DENOISE_SIGNAL, self.p_indirect.output_ptr()[0])
            self.p_denoise.render(cb, self.scene, self.gbuffer, stream)
        if self.p_composit is not None:
            self.p_composit.render(cb, self.scene, self.gbuffer, stream)
        if getattr(self, "p_taa", None) is not None:
            self.p_taa.render(cb, self.scene, self.gbuffer, stream)

    def _first_half(self, cb, stream, prelight):
        """the half of an overlapped frame that goes to the pass's own stream: [sky LUT,] G-buffer, PreLighting, K11 (zetaray_amd.h zr_pass_set_frame_overlap)"""
        a = self._overlap_stream
        if self.p_sky is not None:
            key = b"".join(np.asarray(cb[f]).tobytes() for f in self._SKY_FIELDS)
            if key != getattr(self, "_sky_key", None):
                # K17 rewrites the sky-view LUT that kernels of BOTH halves sample: a rare event (sun / atmosphere edits), ordered the blunt way
                _check(lib().zr_device_synchronize(self._device))
                self.p_sky.render(cb, self.scene, None, stream)
                _check(lib().zr_device_synchronize(self._device))
                self._sky_key = key
        self.p_gbuffer.render(cb, self.scene, self.gbuffer, a)
        if prelight:
            if not self._alias_ready:
                _check(lib().zr_device_synchronize(self._device))      # the alias table's first build is read by both halves
            self.p_prelight.render(cb, self.scene, None, a)          # (K3's presampled sets are read by K11 alone: same stream)
            if not self._alias_ready:
                _check(lib().zr_device_synchronize(self._device))
            self._alias_ready = True
            self._alias_poll = max(0, getattr(self, "_alias_poll", 0) - 1)
        self.p_indirect.render_stage(cb, self.scene, self.gbuffer, STAGE_CANDIDATES, a)

    def _render_frame_overlapped(self, cb, stream=None):
        """render_frame with frame overlap on: first half on the pass's stream, second half on `stream`"""
        assert self.p_direct is None and self.p_sky_direct is None and self.p_composit is None and not self.skip_indirect
        self._first_half(cb, stream, not self._alias_ready or self._presampling or getattr(self, "_alias_poll", 0) > 0)
        self.p_indirect.render_stage(cb, self.scene, self.gbuffer, STAGE_TEMPORAL_REUSE | STAGE_SPATIAL | STAGE_SPATIAL2, stream)
        if getattr(self, "p_denoise", None) is not None:
            self.p_denoise.set_input(IN_DENOISE_SIGNAL, self.p_indirect.output_ptr()[0])
            self.p_denoise.render(cb, self.scene, self.gbuffer, stream)

    def final(self):
        return self.p_indirect.download()
