"""C++ scene ingestion (zetaray_amd/host/zr_scene_io.cpp, libzetaray_sceneio.so): glTF -> wire formats.

Pinned three ways: (1) against the reference's own transform math and EmissiveTriangle packing compiled in place (oracle/_ref/libzref.so:
affineTransformation x parent, decomposeSRT, quaternionFromRotationMat1, unorm4::FromNormalized, half3, RT::EmissiveTriangle), bit for bit on
random inputs; (2) against the Python loader (zetaray_amd/scene_io.py) on the reference's Cornell scenes; (3) on a synthetic glTF with a node
hierarchy, a matrix node, KHR material extensions and an emissive mesh.  Block decompression is checked against Pillow's decoders."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from zetaray_amd import scene_io, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_GLTF = "/root/reference/Assets/CornellBox"
ZREF = os.path.join(ROOT, "oracle", "_ref", "libzref.so")


def sio():
    L = scene_io._sceneio_lib()
    L.zrh_decompose_srt.argtypes = [C.c_void_p] * 4
    L.zrh_compose_world.argtypes = [C.c_void_p] * 5
    L.zrh_fill_mesh_instance.argtypes = [C.c_void_p] * 2
    L.zrh_pack_emissive_triangle.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32, C.c_int, C.c_void_p]
    L.zrh_bc7_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.zrh_bc5_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    return L


def to_ref(M34):
    """3 x 4 column-vector matrix -> the reference's 4 x 3 row-vector layout"""
    M = np.asarray(M34, np.float32).reshape(3, 4)
    return np.ascontiguousarray(np.vstack([M[:, :3].T, M[:, 3]]), np.float32)


def from_ref(M43):
    M = np.asarray(M43, np.float32).reshape(4, 3)
    return np.ascontiguousarray(np.hstack([M[:3].T, M[3].reshape(3, 1)]), np.float32)


def random_trs(rng):
    q = rng.normal(size=4).astype(np.float32)
    q /= np.float32(np.sqrt(np.float32(np.dot(q, q))))
    s = np.exp(rng.uniform(-2, 2, 3)).astype(np.float32)
    t = rng.uniform(-10, 10, 3).astype(np.float32)
    return s, q, t


@pytest.mark.skipif(not os.path.exists(ZREF), reason="needs oracle/_ref/libzref.so (built from /root/reference)")
def test_transform_math_matches_reference_code():
    """world = affineTransformation(s, q, t) x parent over 3-level hierarchies, then FillMeshInstanceData's decomposeSRT + quantisation:
    2000 random chains, every float and every quantised field identical to the reference's SSE code"""
    R = C.CDLL(ZREF)
    R.zref_compose_world.argtypes = [C.c_void_p] * 5
    R.zref_fill_mesh_instance.argtypes = [C.c_void_p] * 6
    L = sio()
    rng = np.random.default_rng(3)
    ident = np.hstack([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)])
    for it in range(2000):
        mine, ref = ident.copy(), to_ref(ident)
        for level in range(3):
            s, q, t = random_trs(rng)
            if it % 7 == 0:
                q = np.array([0, 0, 0, 1], np.float32)       # pure scale + translation
            if it % 5 == 0:
                s = np.full(3, s[0], np.float32)              # uniform scale keeps the chain a rotation + scale
            out = np.zeros((3, 4), np.float32)
            L.zrh_compose_world(s.ctypes.data, q.ctypes.data, t.ctypes.data, mine.ctypes.data, out.ctypes.data)
            rout = np.zeros((4, 3), np.float32)
            R.zref_compose_world(s.ctypes.data, q.ctypes.data, t.ctypes.data, ref.ctypes.data, rout.ctypes.data)
            assert np.array_equal(to_ref(out).view(np.uint32), rout.view(np.uint32)), (it, level)
            mine, ref = out, rout
            # decomposition + quantisation of this world matrix
            inst = np.zeros(1, wire.MESH_INSTANCE)
            L.zrh_fill_mesh_instance(mine.ctypes.data, inst.ctypes.data)
            s3, q4, t3 = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(3, np.float32)
            L.zrh_decompose_srt(mine.ctypes.data, s3.ctypes.data, q4.ctypes.data, t3.ctypes.data)
            rs, rq, rt = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(3, np.float32)
            rrot, rscale = np.zeros(4, np.uint16), np.zeros(3, np.uint16)
            R.zref_fill_mesh_instance(ref.ctypes.data, rs.ctypes.data, rq.ctypes.data, rt.ctypes.data, rrot.ctypes.data, rscale.ctypes.data)
            for a, b, what in ((s3, rs, "scale"), (q4, rq, "quaternion"), (t3, rt, "translation")):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (it, level, what, a, b)
            assert np.array_equal(inst["rotation"][0], rrot) and np.array_equal(inst["scale"][0], rscale)
            assert np.array_equal(inst["prev_rotation"][0], rrot) and np.array_equal(inst["translation"][0].view(np.uint32), rt.view(np.uint32))
            if it % 5 != 0:
                break       # non-uniform scale under a rotated parent shears: the reference does not support decomposing that either


@pytest.mark.skipif(not os.path.exists(ZREF), reason="needs oracle/_ref/libzref.so (built from /root/reference)")
def test_emissive_triangle_packing_matches_reference_code():
    R = C.CDLL(ZREF)
    R.zref_emissive_triangle.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32, C.c_int, C.c_void_p]
    L = sio()
    rng = np.random.default_rng(9)
    for it in range(3000):
        v = rng.uniform(-20, 20, (3, 3)).astype(np.float32)
        if it % 11 == 0:
            v[1] = v[0] + np.float32([1, 0, 0]) * np.float32(rng.uniform(0.01, 3))      # axis-aligned edge: octahedral fold edge cases
        uv = rng.uniform(-2, 3, 6).astype(np.float32)
        factor, tex = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 1 << 16))
        strength = int(np.float16(rng.uniform(0, 60)).view(np.uint16))
        tid, ds = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 2))
        a, b = np.zeros(1, wire.EMISSIVE_TRI), np.zeros(48, np.uint8)
        L.zrh_pack_emissive_triangle(v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, uv.ctypes.data, factor, tex, strength, tid, ds, a.ctypes.data)
        R.zref_emissive_triangle(v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, uv.ctypes.data, factor, tex, strength, tid, ds, b.ctypes.data)
        # the ctor leaves TriIDPatchedBit (24) clear; SceneCore.cpp:229-235 sets it when it replaces the triangle index by the PCG3d hash
        # (EmissiveTriangle::ResetID) -- the wire record is the patched one
        rb = b.view(wire.EMISSIVE_TRI).copy()
        rb["packed_a"] |= 1 << 24
        assert a.tobytes() == rb.tobytes(), (it, a, rb)
        # ... and the Python packer the fixtures were made with agrees
        p = scene_io.pack_emissive_triangle(v[0], v[1], v[2], uv.reshape(3, 2), factor, tex, strength, tid, bool(ds))
        assert p.tobytes() == a.tobytes(), it


@pytest.mark.skipif(not os.path.exists(ZREF), reason="needs oracle/_ref/libzref.so (built from /root/reference)")
def test_emissive_world_transform_matches_reference_code():
    """zrh_emissive_to_world (decode the 16-bit octahedral edges + half lengths, transform, re-encode: what SceneCore does to every emissive
    triangle of a transformed instance on the first frame and to moving ones every frame) against the reference's own
    EmissiveTriangle::LoadVertices -> mul(toWorld, v) -> StoreVertices, 8000 random triangles x random rotations / scales / translations."""
    R, L = C.CDLL(ZREF), sio()
    for f in (R.zref_emissive_to_world, L.zrh_emissive_to_world):
        f.argtypes = [C.c_void_p] * 3
    L.zrh_pack_emissive_triangle.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32, C.c_int, C.c_void_p]
    rng = np.random.default_rng(21)
    for it in range(8000):
        v = (rng.normal(size=(3, 3)) * rng.choice([0.01, 0.3, 2.0, 30.0])).astype(np.float32)
        if it % 7 == 0:
            v[1] = v[0] + np.float32([rng.normal(), 0, 0])
        v0, v1, v2 = (np.ascontiguousarray(x) for x in v)
        uv = rng.random(6).astype(np.float32)
        e = np.zeros(1, wire.EMISSIVE_TRI)
        L.zrh_pack_emissive_triangle(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, uv.ctypes.data, 0x804020, 0xffff, 0x4000, it, 1, e.ctypes.data)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        M = np.zeros((3, 4), np.float32)
        M[:, :3] = (q @ np.diag(rng.uniform(0.1, 3, 3))).astype(np.float32) if it % 11 else np.diag(rng.uniform(0.1, 3, 3)).astype(np.float32)
        M[:, 3] = (rng.normal(size=3) * 5).astype(np.float32)
        M4x3 = np.ascontiguousarray(np.vstack([M[:, :3].T, M[:, 3][None, :]]).astype(np.float32))      # the reference's row-vector layout
        a, b = np.zeros(1, wire.EMISSIVE_TRI), np.zeros(1, wire.EMISSIVE_TRI)
        L.zrh_emissive_to_world(e.ctypes.data, M.ctypes.data, a.ctypes.data)
        R.zref_emissive_to_world(e.ctypes.data, M4x3.ctypes.data, b.ctypes.data)
        assert a.tobytes() == b.tobytes(), (it, a, b)
        assert scene_io.emissive_to_world(e, M).tobytes() == a.tobytes()


@pytest.mark.skipif(not os.path.exists(REF_GLTF), reason="the reference's assets are not on this machine")
@pytest.mark.parametrize("name", ["cornell", "cornell_emissive"])
def test_native_loader_matches_python_loader_on_the_cornell_scenes(name):
    path = os.path.join(REF_GLTF, name + ".gltf")
    a = scene_io.load_gltf(path)
    b, offs = scene_io.load_gltf_native(path)
    for f in ("vertices", "indices", "instance_mask", "instance_num_tris", "emissives"):
        assert getattr(a, f).tobytes() == getattr(b, f).tobytes(), f
    # MeshInstance records: identical except for the floor's base-colour texture, which only the native loader binds
    for fld in wire.MESH_INSTANCE.names:
        if fld != "base_color_tex":
            assert a.instances[fld].tobytes() == b.instances[fld].tobytes(), fld
    ground = [i for i in range(len(b.instances)) if b.instances["base_color_tex"][i] != 0xFFFF]
    assert len(ground) == 1 and b.instances["base_color_tex"][ground[0]] == 0
    # object-to-world: the Python loader rotates basis vectors with Math::RotateVector, the native one builds rotationMatFromQuat like the
    # reference (pinned above): equal up to 1 ulp, exactly equal for the unrotated instances
    d = np.abs(a.instance_to_world.view(np.int32).astype(np.int64) - b.instance_to_world.view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d.reshape(len(d), -1).max(axis=1) == 0).sum() >= 8
    # materials: only the textured one differs (texture index 0 instead of none)
    diff = [i for i in range(len(a.materials)) if a.materials[i].tobytes() != b.materials[i].tobytes()]
    assert len(diff) == 1 and (int(b.materials[diff[0]]["base_color_tex_subsurf_coat_weight"]) & 0xFFFF) == 0
    # the checkerboard: BC7_UNORM_SRGB 1024 x 1024 with 11 mips, two colours in mip 0
    assert offs == dict(base_color=0, normal=1, metallic_roughness=1, emissive=1)
    assert len(b.textures) == 1 and (int(b.textures["width"][0]), int(b.textures["height"][0]), int(b.textures["num_mips"][0])) == (1024, 1024, 11)
    mip0 = b.texels[int(b.textures["offset"][0]):][:1024 * 1024 * 4].reshape(1024, 1024, 4)
    assert len(np.unique(mip0.reshape(-1, 4), axis=0)) == 2
    assert b.texels.size == sum(max(1, 1024 >> m) ** 2 * 4 for m in range(11))


def _write_gltf(tmp_path):
    """two-level hierarchy (rotated + uniformly scaled parent, translated child), a matrix node, an emissive quad with emissive_strength, a
    glass material with ior / transmission / clearcoat, MASK alpha mode"""
    pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    nrm = np.tile(np.float32([0, 1, 0]), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = pos.tobytes() + nrm.tobytes() + uv.tobytes() + idx.tobytes()
    (tmp_path / "quad.bin").write_bytes(blob)
    s45 = float(np.sin(np.pi / 8))
    c45 = float(np.cos(np.pi / 8))
    g = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 3]}],
        "nodes": [
            {"name": "parent", "rotation": [0, s45, 0, c45], "scale": [2, 2, 2], "translation": [1, 2, 3], "children": [1, 2]},
            {"name": "child_plain", "mesh": 0, "translation": [0.5, 0, -0.25]},
            {"name": "child_light", "mesh": 1, "translation": [0, 3, 0], "scale": [0.5, 0.5, 0.5]},
            {"name": "matrix_node", "mesh": 2, "matrix": [3, 0, 0, 0, 0, 3, 0, 0, 0, 0, 3, 0, -4, 1, 2, 1]},
        ],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": m}]} for m in (0, 1, 2)],
        "materials": [
            {"name": "plain", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.2, 0.1, 0.6], "metallicFactor": 1.0, "roughnessFactor": 0.35},
             "alphaMode": "MASK", "alphaCutoff": 0.4, "doubleSided": True},
            {"name": "light", "emissiveFactor": [1.0, 0.5, 0.25], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.5}},
             "pbrMetallicRoughness": {"metallicFactor": 0}},
            {"name": "glass", "pbrMetallicRoughness": {"metallicFactor": 0, "roughnessFactor": 0.05},
             "extensions": {"KHR_materials_ior": {"ior": 1.33}, "KHR_materials_transmission": {"transmissionFactor": 1.0},
                            "KHR_materials_clearcoat": {"clearcoatFactor": 0.7, "clearcoatRoughnessFactor": 0.2}}},
        ],
        "buffers": [{"uri": "quad.bin", "byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                        {"buffer": 0, "byteOffset": 96, "byteLength": 32}, {"buffer": 0, "byteOffset": 128, "byteLength": 12}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"}, {"bufferView": 3, "componentType": 5123, "count": 6, "type": "SCALAR"}],
    }
    p = tmp_path / "scene.gltf"
    p.write_text(json.dumps(g))
    return str(p), g, pos


def test_native_loader_rejects_malformed_files(tmp_path):
    """untrusted input: out-of-range vertex indices, short attribute accessors, a material index beyond the array, a node cycle and a JSON
    file that ends inside a number are errors, never out-of-bounds reads; KHR_materials_emissive_strength alone makes a
    primitive emissive like glTF.cpp:405-410"""
    import copy
    path, g, _ = _write_gltf(tmp_path)

    def load(mut, name, blob=None):
        g2 = copy.deepcopy(g)
        mut(g2)
        q = tmp_path / name
        q.write_text(json.dumps(g2) if blob is None else blob)
        return scene_io.load_gltf_native(str(q))

    def bad_index(g2):      # index 7 with 4 vertices
        idx = np.array([0, 1, 7, 0, 2, 3], np.uint16)
        blob = (tmp_path / "quad.bin").read_bytes()[:128] + idx.tobytes()
        (tmp_path / "quad_bad.bin").write_bytes(blob)
        g2["buffers"][0]["uri"] = "quad_bad.bin"
    for mut, name in ((bad_index, "bad_index.gltf"),
                      (lambda g2: g2["accessors"][1].update(count=3), "short_normals.gltf"),
                      (lambda g2: g2["meshes"][0]["primitives"][0].update(material=9), "bad_material.gltf"),
                      (lambda g2: g2["nodes"][1].update(children=[0]), "cycle.gltf")):
        with pytest.raises(Exception):
            load(mut, name)
    with pytest.raises(Exception):
        load(lambda g2: None, "truncated.gltf", blob=json.dumps(g)[:-40] + "12")
    # emissive strength without a factor or a texture: still an emissive primitive
    def strength_only(g2):
        g2["materials"][0]["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": 3.0}}
    sc, _ = load(strength_only, "strength_only.gltf")
    base, _ = scene_io.load_gltf_native(path)
    assert len(sc.emissives) == len(base.emissives) + 2


def test_native_loader_on_a_synthetic_hierarchy(tmp_path):
    path, g, pos = _write_gltf(tmp_path)
    sc, offs = scene_io.load_gltf_native(path)
    assert len(sc.instances) == 3 and len(sc.vertices) == 12 and len(sc.indices) == 18 and len(sc.textures) == 0
    # RH -> LH: z flipped, winding swapped
    assert np.array_equal(sc.vertices["pos"][:4], pos * np.float32([1, 1, -1]))
    assert sc.indices[:6].tolist() == [0, 2, 1, 0, 3, 2]
    # depth-first order: child_plain, child_light, matrix_node; the expected world matrices in float64
    def trs(t, q, s):
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4)
        M[:3, :3] = R @ np.diag(s)
        M[:3, 3] = t
        return M
    n = g["nodes"]
    def lhs(node):
        t = np.array(node.get("translation", [0, 0, 0]), float) * [1, 1, -1]
        q = np.array(node.get("rotation", [0, 0, 0, 1]), float) * [-1, -1, 1, 1]
        return trs(t, q, node.get("scale", [1, 1, 1]))
    parent = lhs(n[0])
    want = [parent @ lhs(n[1]), parent @ lhs(n[2])]
    Mm = np.array(n[3]["matrix"], float).reshape(4, 4).T
    Cz = np.diag([1, 1, -1, 1.0])
    want.append(Cz @ Mm @ Cz)
    for i, W in enumerate(want):
        got = sc.instance_to_world[i].reshape(3, 4)
        assert np.allclose(got, W[:3], rtol=0, atol=2e-6), (i, got, W[:3])
        # the quantised MeshInstance reproduces the matrix: R(q) * diag(s), T
        inst = sc.instances[i]
        q = inst["rotation"].astype(np.float64) / 65535.0 * 2 - 1
        q /= np.linalg.norm(q)
        s = inst["scale"].view(np.float16).astype(np.float64)
        back = trs(inst["translation"].astype(np.float64), q, s)
        assert np.allclose(back[:3], W[:3], atol=3e-3), i
        assert inst["prev_rotation"].tolist() == inst["rotation"].tolist() and (inst["d_translation"] == 0).all()
    # masks: MASK alpha mode -> non-opaque primary rays; the light is emissive
    assert sc.instance_mask.tolist() == [wire.SUBGROUP_NON_EMISSIVE | 0x80, wire.SUBGROUP_EMISSIVE, wire.SUBGROUP_NON_EMISSIVE]
    assert sc.instances["mat_idx"].tolist() == [1, 2, 3] and sc.instance_num_tris.tolist() == [2, 2, 2]
    # materials == the Python packer on the same numbers (index 0 = the default material)
    assert sc.materials[0].tobytes() == scene_io.pack_material(metallic=0.0, roughness=0.3).tobytes()
    assert sc.materials[1].tobytes() == scene_io.pack_material(base_color=(0.8, 0.2, 0.1, 0.6), metallic=1.0, roughness=0.35, alpha_mode=1, alpha_cutoff=0.4,
                                                               double_sided=True).tobytes()
    assert sc.materials[2].tobytes() == scene_io.pack_material(metallic=0.0, emissive_factor=(1.0, 0.5, 0.25), emissive_strength=12.5).tobytes()
    assert sc.materials[3].tobytes() == scene_io.pack_material(metallic=0.0, roughness=0.05, ior=1.33, transmission=1.0, coat_weight=0.7, coat_roughness=0.2).tobytes()
    assert int(sc.instances["alpha_factor_cutoff"][0]) == scene_io.unorm8(np.float32(scene_io.unorm8(0.6)) / np.float32(255)) | (scene_io.unorm8(np.float32(scene_io.unorm8(0.4)) / np.float32(255)) << 8)
    # emissive triangles: world-space vertices of instance 1, IDs = PCG3d(instance, 0, triangle).x, strength as half
    assert len(sc.emissives) == 2 and int(sc.instances["base_emissive_tri_offset"][1]) == 0 and int(sc.instances["base_emissive_tri_offset"][0]) == 0xFFFFFFFF
    assert [int(e["id"]) for e in sc.emissives] == [scene_io.pcg3d(1, 0, p)[0] for p in range(2)]
    W = want[1]
    v0 = W[:3, :3] @ (pos[0] * [1, 1, -1]) + W[:3, 3]
    assert np.allclose(sc.emissives[0]["vtx0"], v0, atol=1e-5)
    assert (int(sc.emissives[0]["packed_b"]) >> 16) == int(np.float16(12.5).view(np.uint16))
    # the scene is consumable: the oracle builds it and traces the G-buffer
    from oracle import zro
    o = zro.OracleScene(sc)
    cb = scene_io.make_frame_constants(32, 24, num_emissives=len(sc.emissives), cam_pos=(1.0, 6.0, -12.0))
    planes, _ = o.gbuffer(cb)
    assert ((np.asarray(planes[2]).reshape(-1) & 0xff) & 4 == 0).any()       # some pixel hit geometry


def test_native_loader_reports_errors(tmp_path):
    p = tmp_path / "bad.gltf"
    p.write_text('{"asset": {"version": "2.0"}, "buffers": []')
    with pytest.raises(RuntimeError, match="JSON"):
        scene_io.load_gltf_native(str(p))
    p.write_text(json.dumps({"asset": {"version": "2.0"}, "buffers": [], "meshes": [], "nodes": [], "scenes": [{"nodes": [0]}]}))
    with pytest.raises(RuntimeError):
        scene_io.load_gltf_native(str(p))
    with pytest.raises(RuntimeError, match="cannot open"):
        scene_io.load_gltf_native(str(tmp_path / "missing.gltf"))


def _dds(blocks, w, h, fmt):
    hdr = bytearray(148)
    hdr[0:4] = b"DDS "
    struct.pack_into("<7I", hdr, 4, 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, len(blocks), 0, 1)
    struct.pack_into("<2I", hdr, 76, 32, 0x4)
    hdr[84:88] = b"DX10"
    struct.pack_into("<I", hdr, 108, 0x1000)
    struct.pack_into("<5I", hdr, 128, fmt, 3, 0, 1, 0)
    return bytes(hdr) + bytes(blocks)


def test_block_decompression_matches_pillow():
    """4096 random BC7 blocks (512 per mode: every partition, rotation, index mode, p-bit variant shows up) and 4096 random BC5 blocks"""
    Image = pytest.importorskip("PIL.Image")
    import io
    L = sio()
    rng = np.random.default_rng(5)
    w = h = 256
    b = rng.integers(0, 256, ((w // 4) * (h // 4), 16), dtype=np.uint8)
    for i in range(len(b)):
        m = i % 8
        b[i, 0] = (int(b[i, 0]) & ~((1 << (m + 1)) - 1) & 0xFF) | (1 << m)
    blocks = np.ascontiguousarray(b.reshape(-1))
    ref = np.array(Image.open(io.BytesIO(_dds(blocks, w, h, 98))).convert("RGBA"))
    out = np.zeros((h, w, 4), np.uint8)
    L.zrh_bc7_decode(blocks.ctypes.data, w, h, out.ctypes.data)
    assert np.array_equal(out, ref)
    blocks5 = rng.integers(0, 256, (w // 4) * (h // 4) * 16, dtype=np.uint8)
    ref5 = np.array(Image.open(io.BytesIO(_dds(blocks5, w, h, 83))))
    out5 = np.zeros((h, w, 2), np.uint8)
    L.zrh_bc5_decode(blocks5.ctypes.data, w, h, out5.ctypes.data)
    assert np.array_equal(out5, ref5[..., :2])


CORNELL_TEX = os.path.join(ROOT, "tests", "golden", "cornell_textured.npz")
CORNELL_TEX_OFFSETS = dict(base_color=0, normal=1, metallic_roughness=1, emissive=1)


def _textured_cornell_frame(w, h, f=1):
    cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=0)
    scene_io.set_texture_heap_offsets(cb, CORNELL_TEX_OFFSETS)
    return cb


def test_cornell_with_the_reference_floor_texture_oracle():
    """tests/golden/cornell_textured.npz = cornell.gltf through the C++ loader, checkerboard.dds decoded: the oracle's G-buffer shows the
    two tones of the checkerboard on the floor and nothing else changes versus the untextured fixture"""
    from oracle import zro
    sc = scene_io.load_npz(CORNELL_TEX)
    plain = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell.npz"))
    w, h = 96, 54
    cb = _textured_cornell_frame(w, h)
    a, _ = zro.OracleScene(sc, cb=cb).gbuffer(cb)
    b, _ = zro.OracleScene(plain).gbuffer(scene_io.make_frame_constants(w, h, num_emissives=0))
    for plane in range(1, 10):
        if plane != 2:
            assert np.array_equal(np.asarray(a[plane]), np.asarray(b[plane])), wire.GB_PLANE_NAMES[plane]
    diff = np.asarray(a[0]).reshape(h, w) != np.asarray(b[0]).reshape(h, w)
    assert diff.any() and not diff[: h // 2].any()           # only the floor (lower half of the image) changes
    assert len(np.unique(np.asarray(a[0]).reshape(h, w)[diff])) >= 2


@pytest.mark.gpu
def test_cornell_with_the_reference_floor_texture_on_gpu():
    """... and the HIP passes render it bit-exactly like the oracle: K1 planes and 3 frames of ReSTIR PT (sun + sky, textured permutation)"""
    from oracle import zro
    from zetaray_amd import api
    sc = scene_io.load_npz(CORNELL_TEX)
    w, h = 128, 72
    prm = wire.default_params()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    osc = zro.OracleScene(sc, cb=_textured_cornell_frame(w, h))
    opt = zro.OracleRPT(osc, w, h)
    for f in range(1, 4):
        cb = _textured_cornell_frame(w, h, f)
        r.render_frame(cb)
        osc.sky_lut(cb, 256, 128)
        want = opt.render(cb, prm)
        got = r.final()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"frame {f}"
        if f == 1:
            planes, _ = r.gbuffer.download()
            oplanes, _ = osc.gbuffer(cb)
            for n, x, y in zip(wire.GB_PLANE_NAMES, planes, oplanes):
                assert np.array_equal(np.asarray(x).view(np.uint8).reshape(-1), np.asarray(y).view(np.uint8).reshape(-1)), n
    assert got[..., :3].max() > 0


@pytest.mark.skipif(not os.path.exists(ZREF), reason="needs oracle/_ref/libzref.so (built from /root/reference)")
def test_per_frame_scene_maintenance_matches_reference_code(tmp_path):
    """zrh_scene_data_begin_frame / _set_instance_world (the C++ host's per-frame scene update) against the reference's own code: MeshInstance
    records = TLAS::FillMeshInstanceData's !staticMesh branch (decomposeSRT of the current and the previous world matrix, dTranslation =
    half3(t - t_prev)), the moved light's EmissiveTriangle records = LoadVertices -> mul -> StoreVertices of the object-space ones; instances
    that do not move turn static (Prev* = current, dTranslation = 0)."""
    path, g, pos = _write_gltf(tmp_path)
    L, R = sio(), C.CDLL(ZREF)
    L.zrh_scene_data_begin_frame.argtypes = [C.c_void_p]
    L.zrh_scene_data_set_instance_world.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.zrh_scene_data_dirty_emissives.argtypes = [C.c_void_p] * 3
    L.zrh_scene_data_initial_emissives.restype = C.c_void_p
    L.zrh_scene_data_initial_emissives.argtypes = [C.c_void_p]
    R.zref_fill_mesh_instance.argtypes = [C.c_void_p] * 6
    R.zref_emissive_to_world.argtypes = [C.c_void_p] * 3
    rho, dim = scene_io.load_rho_default()
    rho = np.ascontiguousarray(rho, np.uint16)
    h = C.c_void_p()
    assert L.zrh_gltf_load(os.fsencode(path), rho.ctypes.data, (C.c_uint32 * 3)(*dim), C.byref(h)) == 0
    d = L.zrh_scene_data_desc(h).contents
    n, ne = d.num_instances, d.num_emissives
    inst = np.ctypeslib.as_array(C.cast(d.instances, C.POINTER(C.c_uint8)), (n * wire.MESH_INSTANCE.itemsize,)).view(wire.MESH_INSTANCE)
    world = np.ctypeslib.as_array(C.cast(d.instance_to_world, C.POINTER(C.c_float)), (n, 12))
    ems = np.ctypeslib.as_array(C.cast(d.emissives, C.POINTER(C.c_uint8)), (ne * 48,)).view(wire.EMISSIVE_TRI)
    init = np.ctypeslib.as_array(C.cast(L.zrh_scene_data_initial_emissives(h), C.POINTER(C.c_uint8)), (ne * 48,)).view(wire.EMISSIVE_TRI).copy()
    light = [i for i in range(n) if inst["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    plain = [i for i in range(n) if i != light][0]
    rng = np.random.default_rng(3)

    def ref_srt(M):      # the reference's decomposition + quantisation of a 3 x 4 column-vector matrix
        M4x3 = np.ascontiguousarray(np.vstack([M.reshape(3, 4)[:, :3].T, M.reshape(3, 4)[:, 3][None, :]]).astype(np.float32))
        s3, q4, t3, rot, scl = np.zeros(3, np.float32), np.zeros(4, np.float32), np.zeros(3, np.float32), np.zeros(4, np.uint16), np.zeros(3, np.uint16)
        R.zref_fill_mesh_instance(M4x3.ctypes.data, s3.ctypes.data, q4.ctypes.data, t3.ctypes.data, rot.ctypes.data, scl.ctypes.data)
        return t3, rot, scl, M4x3

    for frame in range(4):
        before_world, before_inst = world.copy(), inst.copy()
        L.zrh_scene_data_begin_frame(h)
        moved = {}
        for i in ((light, plain) if frame < 3 else (plain,)):      # last frame: the light rests
            a = rng.uniform(-0.6, 0.6)
            Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.diag([rng.uniform(0.5, 2)] * 3)
            M = np.zeros((3, 4), np.float32); M[:, :3] = Rm.astype(np.float32); M[:, 3] = rng.uniform(-3, 3, 3).astype(np.float32)
            assert L.zrh_scene_data_set_instance_world(h, i, M.ctypes.data) == 0
            moved[i] = M
        for i in range(n):
            if i in moved:
                t, rot, scl, M4x3 = ref_srt(moved[i])
                tp, rotp, sclp, _ = ref_srt(before_world[i])
                assert np.array_equal(world[i], moved[i].reshape(12))
                assert np.array_equal(inst["rotation"][i], rot) and np.array_equal(inst["scale"][i], scl) and np.array_equal(inst["translation"][i].view(np.uint32), t.view(np.uint32))
                assert np.array_equal(inst["prev_rotation"][i], rotp) and np.array_equal(inst["prev_scale"][i], sclp)
                assert np.array_equal(inst["d_translation"][i], (t - tp).astype(np.float16).view(np.uint16)), (frame, i)
            else:
                assert np.array_equal(inst["rotation"][i], before_inst["rotation"][i]) and np.array_equal(inst["prev_rotation"][i], inst["rotation"][i])
                assert np.array_equal(inst["prev_scale"][i], inst["scale"][i]) and not inst["d_translation"][i].any()
        first, count = C.c_uint32(), C.c_uint32()
        L.zrh_scene_data_dirty_emissives(h, C.byref(first), C.byref(count))
        if light in moved:
            b, k = int(inst["base_emissive_tri_offset"][light]), int(C.cast(d.instance_num_tris, C.POINTER(C.c_uint32))[light])
            assert (first.value, count.value) == (b, k)
            M4x3 = ref_srt(moved[light])[3]
            for j in range(b, b + k):
                want = np.zeros(1, wire.EMISSIVE_TRI)
                R.zref_emissive_to_world(init[j:j + 1].ctypes.data, M4x3.ctypes.data, want.ctypes.data)
                assert ems[j:j + 1].tobytes() == want.tobytes(), (frame, j)
        else:
            assert count.value == 0
    L.zrh_scene_data_destroy(h)


@pytest.mark.gpu
def test_cpp_scene_maintenance_drives_the_device_scene(tmp_path):
    """The C++ host's whole dynamic-scene path on the GPU: zrh_gltf_load -> per frame zrh_scene_data_begin_frame / _set_instance_world (a light-carrying
    instance and a plain one move) -> zrh_scene_apply_updates (zr_scene_update_emissives + zr_scene_update_instances: device refit, previous BVH kept)
    -> ReSTIR PT + ReSTIR DI.  The oracle is fed the same records; radiance of both bit-exact over 4 frames."""
    from oracle import zro
    from zetaray_amd import api
    path, g, pos = _write_gltf(tmp_path)
    sc, _ = scene_io.load_gltf_native(path)
    L = sio()
    H = C.CDLL(os.path.join(ROOT, "zetaray_amd", "libzetaray_host.so"))
    H.zrh_scene_apply_updates.argtypes = [C.c_void_p, C.c_void_p]
    L.zrh_scene_data_begin_frame.argtypes = [C.c_void_p]
    L.zrh_scene_data_set_instance_world.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rho, dim = scene_io.load_rho_default()
    rho = np.ascontiguousarray(rho, np.uint16)
    h = C.c_void_p()
    assert L.zrh_gltf_load(os.fsencode(path), rho.ctypes.data, (C.c_uint32 * 3)(*dim), C.byref(h)) == 0
    d = L.zrh_scene_data_desc(h).contents
    n, ne = d.num_instances, d.num_emissives
    inst = np.ctypeslib.as_array(C.cast(d.instances, C.POINTER(C.c_uint8)), (n * wire.MESH_INSTANCE.itemsize,)).view(wire.MESH_INSTANCE)
    world = np.ctypeslib.as_array(C.cast(d.instance_to_world, C.POINTER(C.c_float)), (n, 12))
    ems = np.ctypeslib.as_array(C.cast(d.emissives, C.POINTER(C.c_uint8)), (ne * 48,)).view(wire.EMISSIVE_TRI)
    light = [i for i in range(n) if inst["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    plain = [i for i in range(n) if i != light][0]
    w, hgt = 96, 64
    prm = wire.default_params()
    dprm = wire.default_params_di()
    r = api.Renderer(sc, w, hgt, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)      # ReSTIR DI: the moved light's records decide every pixel of it
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, hgt), zro.OracleRDI(osc, w, hgt)
    base = {i: world[i].reshape(3, 4).copy() for i in (light, plain)}
    prev, lit = None, 0.0
    for f in range(1, 5):
        if f >= 2:
            L.zrh_scene_data_begin_frame(h)
            for i in (light, plain):
                M = base[i].copy()
                M[:, 3] += np.float32([0.1 * (f - 1), 0.05 * (f - 1) * (1 if i == light else -1), 0.0])
                assert L.zrh_scene_data_set_instance_world(h, i, np.ascontiguousarray(M).ctypes.data) == 0
            assert H.zrh_scene_apply_updates(r.scene.h, h) == 0
            osc.update_emissives(ems.copy(), 0)
            osc.update_instances(inst.copy(), world.copy())
        # camera between the light (above) and the plain quad, looking down at the quad
        c = base[plain][:, 3]
        cb = scene_io.make_frame_constants(w, hgt, frame_num=f, num_emissives=ne, cam_pos=(float(c[0]), float(c[1]) + 3.0, float(c[2])), view_dir=(0, -1, 0), up=(0, 0, 1))
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb)
        want, want_di = opt.render(cb, prm), odi.render(cb, dprm)
        assert np.array_equal(r.final().view(np.uint32), want.view(np.uint32)), f"frame {f}: ReSTIR PT"
        got = di.download()
        assert np.array_equal(got.view(np.uint32), want_di.view(np.uint32)), f"frame {f}: ReSTIR DI"
        lit = max(lit, float(got[..., :3].max()))
    assert lit > 0      # the quad under the light is lit
    L.zrh_scene_data_destroy(h)
