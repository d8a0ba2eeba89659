"""CPU-side (`-m "not gpu"`) checks: the math contract, the wavefront stage functions of the HIP path executed by
the test-only serial host executor (tests/hostexec) against the oracle, BVH vs brute force, the C-ABI surface, and the
committed golden renders."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import zro
from tests.hostexec import zhx
from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ arithmetic contract (include/zr_detmath.h)
def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def test_detmath_accuracy():
    rng = np.random.default_rng(0)
    x = rng.uniform(-20, 20, 200000).astype(np.float32)
    assert np.abs(zro.kat_unary(0, x) - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(zro.kat_unary(1, x) - np.cos(x.astype(np.float64))).max() < 3e-7
    xe = rng.uniform(-80, 80, 200000).astype(np.float32)
    e = zro.kat_unary(2, xe)
    assert (np.abs(e - np.exp(xe.astype(np.float64))) / np.exp(xe.astype(np.float64))).max() < 3e-7
    xl = (10.0 ** rng.uniform(-30, 30, 200000)).astype(np.float32)
    assert np.abs(zro.kat_unary(3, xl) - np.log(xl.astype(np.float64))).max() < 1e-5
    assert (_ulp_diff(zro.kat_unary(3, xl), np.log(xl.astype(np.float64)).astype(np.float32)) <= 2).all()
    xa = rng.uniform(-50, 50, 200000).astype(np.float32)
    assert np.abs(zro.kat_unary(4, xa) - np.arctan(xa.astype(np.float64))).max() < 3e-7
    assert zro.kat_unary(2, np.array([0.0], np.float32))[0] == 1.0
    assert zro.kat_unary(3, np.array([1.0], np.float32))[0] == 0.0
    assert np.isneginf(zro.kat_unary(3, np.array([0.0], np.float32))[0])


def test_half_roundtrip_all_halves():
    h = np.arange(65536, dtype=np.uint16)
    f = np.zeros(65536, np.float32)
    zro.lib().zro_kat_f16_to_f32(h.ctypes.data, f.ctypes.data, len(h))
    want = h.view(np.float16).astype(np.float32)
    nan = np.isnan(want)
    assert np.array_equal(f[~nan].view(np.uint32), want[~nan].view(np.uint32)) and np.isnan(f[nan]).all()
    back = np.zeros(65536, np.uint16)
    zro.lib().zro_kat_f32_to_f16(f.ctypes.data, back.ctypes.data, len(f))
    assert np.array_equal(back[~nan], h[~nan])


def test_pcg3d_known_answers():
    v = np.array([[0, 0, 0], [1, 2, 3], [4, 0, 57], [0xFFFFFFFF, 7, 9]], np.uint32)
    out = np.zeros_like(v)
    zro.lib().zro_kat_pcg3d(v.ctypes.data, out.ctypes.data, len(v))
    for row, got in zip(v, out):
        assert tuple(int(g) for g in got) == scene_io.pcg3d(int(row[0]), int(row[1]), int(row[2]))


def test_uniform_bounded_is_in_range_and_uniform():
    out = np.zeros(60000, np.uint32)
    zro.lib().zro_kat_uniform_bounded(123, 7, out.ctypes.data, len(out))
    assert out.max() < 7
    counts = np.bincount(out, minlength=7)
    assert np.abs(counts / len(out) - 1 / 7).max() < 0.01


# ------------------------------------------------------------------ stage functions (host executor) vs oracle
@pytest.fixture(scope="module")
def hx_emissive(cornell_emissive, oracle_emissive):
    return zhx.HostExecScene(cornell_emissive, oracle_emissive.alias)


def test_power_estimate_bit_exact(hx_emissive, oracle_emissive):
    assert np.array_equal(hx_emissive.estimate_power().view(np.uint32), oracle_emissive.power.view(np.uint32))


@pytest.mark.parametrize("w,h,frame", [(64, 64, 1), (100, 56, 5)])
def test_gbuffer_and_path_tracer_bit_exact(hx_emissive, oracle_emissive, cornell_emissive, w, h, frame):
    cb = scene_io.make_frame_constants(w, h, frame_num=frame, num_emissives=len(cornell_emissive.emissives))
    ga, gp = oracle_emissive.gbuffer(cb)
    ha, hp = hx_emissive.gbuffer(cb)
    for name, a, b in zip(wire.GB_PLANE_NAMES, ga, ha):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    prm = wire.default_params()
    fo, co = oracle_emissive.pathtrace(cb, gp, prm)
    fh, ch = hx_emissive.pathtrace(cb, hp, prm)
    assert co == ch
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))
    assert fo[..., :3].sum() > 0


def test_dof_and_jitter_paths(hx_emissive, oracle_emissive, cornell_emissive):
    cb = scene_io.make_frame_constants(48, 48, frame_num=3, num_emissives=len(cornell_emissive.emissives), jitter=(0.25, -0.125))
    cb["dof"] = 1
    cb["lens_radius"] = 0.05
    cb["focus_depth"] = 4.0
    ga, gp = oracle_emissive.gbuffer(cb)
    ha, hp = hx_emissive.gbuffer(cb)
    for name, a, b in zip(wire.GB_PLANE_NAMES, ga, ha):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    prm = wire.default_params()
    fo, _ = oracle_emissive.pathtrace(cb, gp, prm)
    fh, _ = hx_emissive.pathtrace(cb, hp, prm)
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))


def _random_rays(n, seed, lo=(-1.2, 0.0, -1.2), hi=(1.2, 2.1, 1.2)):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.uniform(size=(n, 1)) < 0.5, 3.0e38, rng.uniform(0.1, 3.0, (n, 1)))
    return np.concatenate([o, np.full((n, 1), 1e-5), d, tmax], 1).astype(np.float32)


def test_traversal_bvh_vs_bruteforce(hx_emissive, oracle_emissive, cornell_emissive):
    """Closest hit with the ABI tie-break and any-hit do not depend on the acceleration structure: oracle brute force
    == oracle BVH2 == product SAH BVH (host-executed)."""
    rays = _random_rays(40000, 11)
    brute = oracle_emissive.trace_closest(rays)
    obvh = zro.OracleScene(cornell_emissive, force_bvh=True)
    assert np.array_equal(brute, obvh.trace_closest(rays))
    assert np.array_equal(brute, hx_emissive.trace_closest(rays))
    for mask in (1, 2):
        assert np.array_equal(oracle_emissive.trace_closest(rays, mask), hx_emissive.trace_closest(rays, mask))
        assert np.array_equal(oracle_emissive.trace_any(rays, mask), hx_emissive.trace_any(rays, mask))
    assert np.array_equal(oracle_emissive.trace_any(rays), hx_emissive.trace_any(rays))


def test_traversal_synthetic_scene():
    """A few thousand random + axis-aligned coplanar triangles (ties on t): brute force vs both BVHs."""
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=64, seed=3)
    o = zro.OracleScene(sc)                      # <= 256 tris -> brute force, otherwise BVH: force both
    ob = zro.OracleScene(sc, force_bvh=True)
    hx = zhx.HostExecScene(sc, o.alias)
    nodes, tris, depth = hx.bvh_info()
    assert tris == sc.num_tris and nodes > 100 and depth < 40
    rays = _random_rays(20000, 5, lo=(-4, -4, -4), hi=(4, 4, 4))
    a = ob.trace_closest(rays)
    assert np.array_equal(a, hx.trace_closest(rays))
    assert (a[:, 3] != 0xFFFFFFFF).mean() > 0.2


def test_empty_and_degenerate_inputs(hx_emissive, oracle_emissive):
    rays = np.zeros((0, 8), np.float32)
    assert oracle_emissive.trace_closest(rays).shape == (0, 4)
    # zero-length direction, tmax <= tmin, NaN origin: all must miss identically
    bad = np.array([[0, 1, 0, 0, 0, 0, 0, 1e30], [0, 1, 0, 1.0, 0, 0, 1, 0.5], [np.nan, 1, 0, 0, 0, 0, 1, 1e30]], np.float32)
    assert np.array_equal(oracle_emissive.trace_closest(bad), hx_emissive.trace_closest(bad))
    assert (oracle_emissive.trace_closest(bad)[:, 3] == 0xFFFFFFFF).all()


# ------------------------------------------------------------------ golden fixtures (regression pin of the oracle)
def test_golden_render(oracle_emissive, cornell_emissive):
    g = np.load(os.path.join(ROOT, "tests", "golden", "render_emissive_64.npz"))
    cb = scene_io.make_frame_constants(64, 64, frame_num=1, num_emissives=len(cornell_emissive.emissives))
    arrays, planes = oracle_emissive.gbuffer(cb)
    for name, a in zip(wire.GB_PLANE_NAMES, arrays):
        assert np.array_equal(a, g["gb_" + name]), name
    final, cnt = oracle_emissive.pathtrace(cb, planes, wire.default_params())
    assert np.array_equal(final.view(np.uint32), g["final"].view(np.uint32))
    assert tuple(cnt) == tuple(g["counters"])


@pytest.mark.parametrize("scene_name,golden", [("cornell.npz", "config1_cornell_256.npz"), ("cornell_emissive.npz", "config1_cornell_emissive_256.npz")])
def test_golden_config1_256(scene_name, golden):
    """BASELINE config 1 (256 x 256, K9 1 spp, frame 1, jitter off, default sun) for both Cornell scenes: the oracle
    reproduces the committed fixtures (tools/make_goldens.py) -- G-buffer planes, sky-view LUT, FINAL, ray counters.  The GPU test
    test_baseline_config1_goldens_on_gpu checks the HIP path against the same files."""
    from oracle import zro
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", scene_name))
    o = zro.OracleScene(sc)
    g = np.load(os.path.join(ROOT, "tests", "golden", golden))
    cb = scene_io.make_frame_constants(256, 256, frame_num=1, num_emissives=len(sc.emissives))
    if "sky_lut" in g.files:
        assert np.array_equal(o.sky_lut(cb, 256, 128), g["sky_lut"])
    arrays, planes = o.gbuffer(cb)
    for name, a in zip(wire.GB_PLANE_NAMES, arrays):
        assert np.array_equal(a, g["gb_" + name]), name
    final, cnt = o.pathtrace(cb, planes, wire.default_params())
    assert np.array_equal(final.view(np.uint32), g["final"].view(np.uint32))
    assert tuple(cnt) == tuple(g["counters"])
    assert final[..., :3].max() > 0


# ------------------------------------------------------------------ C-ABI surface
def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "zetaray_amd.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(zr_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 28
    from zetaray_amd import api
    L = api.lib()
    for name in declared:
        assert hasattr(L, name), f"libzetaray_amd.so does not export {name}"
    assert declared == set(api.EXPORTS)
    assert L.zr_abi_version() == 3


def test_no_device_is_a_loud_error(cornell_emissive):
    """This container has no GPU: compute entry points must fail with ZR_ERR_NO_DEVICE, never fall back."""
    from zetaray_amd import api
    if api.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(api.ZetaRayError) as e:
        api.Scene(cornell_emissive)
    assert e.value.code == 2
    with pytest.raises(api.ZetaRayError):
        api.GBuffer(64, 64)
    with pytest.raises(api.ZetaRayError):
        api.Pass(api.PASS_INDIRECT, 64, 64)


def test_wire_struct_sizes():
    assert wire.VERTEX.itemsize == 28 and wire.MESH_INSTANCE.itemsize == 64 and wire.MATERIAL.itemsize == 32
    assert wire.EMISSIVE_TRI.itemsize == 48 and wire.ALIAS_ENTRY.itemsize == 16 and wire.FRAME_CONSTANTS.itemsize == 544
    assert C.sizeof(wire.Params) == 128     # 64 + the auto-exposure (16 B) and display (16 B) blocks + tex_filter + the denoise block (24 B) + num_spatial_passes (ABI 3)


# ------------------------------------------------------------------ Russian roulette + special materials
@pytest.mark.parametrize("nb,gb,rr", [(5, 6, True), (5, 6, False)])
def test_russian_roulette_cornell(hx_emissive, oracle_emissive, cornell_emissive, nb, gb, rr):
    """RR (PathTracing.hlsli:62-72) only triggers from bounce 3: raise the bounce limits so it does; the 8x8-group
    max-throughput reduction of the wavefront path (per-group atomicMax + parked paths) must equal the oracle's
    virtual-wave model."""
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gb
    if not rr:
        prm.flags &= ~wire.IND_RUSSIAN_ROULETTE
    cb = scene_io.make_frame_constants(64, 64, frame_num=1, num_emissives=len(cornell_emissive.emissives))
    _, gp = oracle_emissive.gbuffer(cb)
    fo, co = oracle_emissive.pathtrace(cb, gp, prm)
    fh, ch = hx_emissive.pathtrace(cb, gp, prm)
    assert co == ch
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))


@pytest.fixture(scope="module")
def synthetic_small():
    sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=150, seed=11)
    o = zro.OracleScene(sc, force_bvh=True)
    return sc, o, zhx.HostExecScene(sc, o.alias)


@pytest.mark.parametrize("frame,nb,gb", [(1, 3, 4), (3, 6, 8)])
def test_synthetic_scene_all_material_classes(synthetic_small, frame, nb, gb):
    """Metal, coat, specular + rough glass with Beer-Lambert, thin-walled diffuse transmission; transmissive primary
    surfaces take the 4-bounce limit so RR triggers with the defaults."""
    sc, o, hx = synthetic_small
    prm = wire.default_params()
    prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gb
    cb = scene_io.make_frame_constants(96, 64, frame_num=frame, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
    ga, gp = o.gbuffer(cb)
    ha, _ = hx.gbuffer(cb)
    for name, a, b in zip(wire.GB_PLANE_NAMES, ga, ha):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    flags = set(np.unique(ga[2] & 0xff).tolist())
    assert {1, 16, 32, 128} <= flags or {9, 16, 32, 128} <= flags        # transmissive, subsurface, coated, metallic all visible
    fo, co = o.pathtrace(cb, gp, prm)
    fh, ch = hx.pathtrace(cb, gp, prm)
    assert co == ch
    assert np.array_equal(fo.view(np.uint32), fh.view(np.uint32))


def test_presampled_light_sets(synthetic_small):
    """K3 PresampleEmissives + the USE_PRESAMPLED_SETS NEE branches of K9 and K11: sets bit-exact, radiance bit-exact."""
    from oracle import zro as _zro
    from tests.hostexec import zhx as _zhx
    sc, o, hx = synthetic_small
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 16, 64
    w, h = 64, 48
    orpt, hrpt = _zro.OracleRPT(o, w, h), _zhx.HostExecRPT(hx, w, h)
    for f in (1, 2, 3):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
        a, b = o.presample(f, 16, 64), hx.presample(f, 16, 64)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        assert len(np.unique(a["idx"])) > 20 and (a["pdf"] > 0).all()
        _, planes = o.gbuffer(cb)
        want, cnt = o.pathtrace(cb, planes, prm)
        got, cnt2 = hx.pathtrace(cb, planes, prm)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and cnt == cnt2
        r1, r2 = orpt.render(cb, prm), hrpt.render(cb, prm)
        assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32)) and orpt.counters == hrpt.counters
    # presampling changes the light draws: the image must differ from the alias-table path
    p0 = wire.default_params()
    want0, _ = o.pathtrace(cb, planes, p0)
    assert not np.array_equal(want0, want)


def test_thread_sort_maps_are_tile_permutations(oracle_emissive, cornell_emissive):
    """K12 (ReSTIR_PT_Sort.hlsl) properties that hold for any input: inside the render target every thread-map entry decodes to a pixel of the
    same 32 x 32 tile, every pixel is assigned to exactly one thread (no pixel lost or processed twice at the partial boundary tiles), pixels
    flagged invalid carry the error bit, and the oracle and the host-executed HIP stage functions produce the same maps."""
    from oracle import zro
    from tests.hostexec import zhx
    w, h = 150, 90
    prm = wire.default_params()
    o = zro.OracleRPT(oracle_emissive, w, h)
    hx = zhx.HostExecScene(cornell_emissive, alias=oracle_emissive.alias)
    hr = zhx.HostExecRPT(hx, w, h)
    prev = None
    for f in range(1, 4):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(cornell_emissive.emissives), cam_pos=(0.07 * max(0, f - 1), 1.2, -4.043))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        a = o.render(cb, prm)
        b = hr.render(cb, prm, gb=hx.gbuffer(cb))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {f}"
    ys, xs = np.mgrid[0:h, 0:w]
    for name in ("map_ctn", "map_ntc"):
        m = o.plane(name).reshape(h, w).astype(np.int64)
        assert np.array_equal(m, hr.plane(name).reshape(h, w)), name
        px, py = xs + (m & 0x3f) - 31, ys + ((m >> 7) & 0x3f) - 31
        assert ((px >= 0) & (px < w) & (py >= 0) & (py < h)).all(), name
        assert np.array_equal(px // 32, xs // 32) and np.array_equal(py // 32, ys // 32), f"{name}: a pixel left its tile"
        assert len(np.unique(py * w + px)) == w * h, f"{name}: not a permutation"
    # the spatial maps of the last frame really sort: a third of the positions at least point elsewhere
    assert ((o.plane("map_ntc").reshape(h, w) & 0x7fff) != (31 | (31 << 7))).mean() > 0.3


def test_c_and_python_parameter_defaults_agree():
    """zr_params_default (what zr_pass_create and the C++ host start from) == wire.default_params(), field by field -- including the blocks of
    the post and denoise passes"""
    import ctypes as C
    from zetaray_amd import api, wire
    c = wire.Params()
    L = api.lib()
    L.zr_params_default.argtypes = [C.c_void_p]
    assert L.zr_params_default(C.byref(c)) == 0
    p = wire.default_params()
    for name, _ in wire.Params._fields_:
        a, b = getattr(c, name), getattr(p, name)
        if hasattr(a, "__len__"):
            a, b = list(a), list(b)
        if name in ("lvg_grid_dim", "lvg_extents", "lvg_offset_y", "use_lvg"):
            continue      # the light voxel grid block is filled in by the caller that turns it on (DefaultRendererImpl.h:73-77)
        assert a == b or (isinstance(a, float) and abs(a - b) <= 1e-7 * abs(b)), (name, a, b)


def test_host_bvh_build_is_independent_of_its_thread_count():
    """the forked SAH build (zr_bvh.h, ZR_BVH_THREADS): same 4-wide nodes, same leaf-ordered triangles, same stack bound for 1, 3 and 8 threads"""
    import os
    from tests.hostexec import zhx
    from zetaray_amd import scene_io
    sc = scene_io.make_synthetic_scene(num_tris=60000, num_emissive=2000, seed=3)
    old = os.environ.get("ZR_BVH_THREADS")
    try:
        digests = []
        for nt in ("1", "3", "8"):
            os.environ["ZR_BVH_THREADS"] = nt
            digests.append(zhx.HostExecScene(sc).bvh_digest())
    finally:
        if old is None:
            os.environ.pop("ZR_BVH_THREADS", None)
        else:
            os.environ["ZR_BVH_THREADS"] = old
    assert len(set(digests)) == 1 and digests[0][1] > 1000, digests


@pytest.mark.parametrize("header", ["zetaray_amd.h", "zr_wire.h", "zr_detmath.h", "zr_intersect.h", "zr_texture.h", "zr_srgb_table.h"])
def test_public_headers_are_plain_c(header, tmp_path):
    """The boundary is a C ABI: every header under include/ compiles on its own as C99 (-pedantic, warnings are errors) and as C++17 -- what a
    cgo / JNI / ctypes binding generator or the reference's C++ would feed on; no torch, HIP or C++ types in the signatures."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    inc = os.path.join(ROOT, "include")
    src_c, src_cpp = tmp_path / "t.c", tmp_path / "t.cpp"
    body = f'#include "{header}"\nint main(void) {{ return 0; }}\n'
    src_c.write_text(body)
    src_cpp.write_text(body)
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src_c)],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unused-function", "-fsyntax-only", "-I", inc, str(src_cpp)]):
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, " ".join(cmd) + "\n" + res.stderr[-3000:]


def test_clean_checkout_has_every_build_prerequisite(tmp_path):
    """the generated sample-set tables (zr_*_sample_set.inc) are git-ignored prerequisites; their generator rules once went missing and stale
    files in the builder's tree hid it.  A tree made of nothing but the tracked files must be able to plan every build (`make -n`): the product library,
    the tolerance-mode and experiments builds, the host executor and the oracle."""
    import shutil
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")) or shutil.which("git") is None:
        pytest.skip("not a git checkout (the GPU box gets a snapshot without .git)")
    co = tmp_path / "co"
    co.mkdir()
    ar = subprocess.run(["git", "-C", ROOT, "archive", "HEAD"], capture_output=True, check=True)
    subprocess.run(["tar", "-x", "-C", str(co)], input=ar.stdout, check=True)
    for sub, target in (("zetaray_amd/csrc", []), ("zetaray_amd/csrc", ["fast"]), ("zetaray_amd/csrc", ["experiments"]), ("tests/hostexec", []), ("oracle", [])):
        r = subprocess.run(["make", "-n", "-C", str(co / sub)] + target, capture_output=True, text=True)
        assert r.returncode == 0, f"{sub} {target}: {r.stderr[-800:]}"
