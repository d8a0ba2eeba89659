"""Extracts the raw R16_UNORM payload of the reference's rho.dds (64x32x16 GGX dielectric reflectance LUT,
Assets/LUT/rho.dds, MIT) into zetaray_amd/assets/rho_lut_u16.bin so it travels to the GPU box (which has no
/root/reference), and converts the two Cornell glTF scenes (Assets/CornellBox, CC-BY-4.0, "Cornell Box- Original" by
t-ly, https://sketchfab.com/3d-models/cornell-box-original-0d18de8d108c4c9cab1a4405698cc6b6) into wire-format
fixtures tests/golden/cornell*.npz with zetaray_amd.scene_io.load_gltf.
Run once in the build container:  python tools/extract_assets.py
"""
import os
import shutil
import struct

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    raw = open(os.path.join(REF, "Assets/LUT/rho.dds"), "rb").read()
    assert raw[:4] == b"DDS "
    h = struct.unpack("<31I", raw[4:128])
    height, width, depth = h[2], h[3], h[5]
    assert (width, height, depth) == (64, 32, 16)
    payload = raw[128:128 + width * height * depth * 2]
    os.makedirs(os.path.join(ROOT, "zetaray_amd/assets"), exist_ok=True)
    open(os.path.join(ROOT, "zetaray_amd/assets/rho_lut_u16.bin"), "wb").write(payload)
    # 512-point spatial-search sample set (RP/IndirectLighting/ReSTIR_PT/SampleSet.hlsli: `static const half2 k_samples[512]`),
    # stored as 512 x 2 IEEE binary16 values; a data table, shipped like the LUT above
    import re
    import numpy as np
    txt = open(os.path.join(REF, "Source/ZetaRenderPass/IndirectLighting/ReSTIR_PT/SampleSet.hlsli")).read()
    pts = re.findall(r"half2\(([-0-9.e]+),\s*([-0-9.e]+)\)", txt)
    assert len(pts) == 512, len(pts)
    arr = np.array(pts, dtype=np.float64).astype(np.float32).astype(np.float16)
    arr.tofile(os.path.join(ROOT, "zetaray_amd/assets/rpt_sample_set_f16.bin"))
    # 32-point spatial sample set of ReSTIR DI (DirectLighting/Emissive/Resampling.hlsli: `static const half2 k_samples[32]`)
    txt = open(os.path.join(REF, "Source/ZetaRenderPass/DirectLighting/Emissive/Resampling.hlsli")).read()
    pts = re.findall(r"half2\(([-0-9.e]+),\s*([-0-9.e]+)\)", txt)
    assert len(pts) == 32, len(pts)
    np.array(pts, dtype=np.float64).astype(np.float32).astype(np.float16).tofile(os.path.join(ROOT, "zetaray_amd/assets/rdi_sample_set_f16.bin"))
    # Tony McMapface tone-mapping LUT (Assets/LUT/tony_mc_mapface.dds, MIT, https://github.com/h3r2tic/tony-mc-mapface): the 48^3
    # R9G9B9E5_SHAREDEXP payload behind the DDS + DX10 headers
    raw = open(os.path.join(REF, "Assets/LUT/tony_mc_mapface.dds"), "rb").read()
    assert raw[:4] == b"DDS " and raw[84:88] == b"DX10"
    h = struct.unpack("<31I", raw[4:128])
    assert (h[2], h[3], h[5]) == (48, 48, 48) and struct.unpack("<I", raw[128:132])[0] == 67      # DXGI_FORMAT_R9G9B9E5_SHAREDEXP
    open(os.path.join(ROOT, "zetaray_amd/assets/tony_mc_mapface_rgb9e5.bin"), "wb").write(raw[148:148 + 48 ** 3 * 4])
    import sys
    sys.path.insert(0, ROOT)
    from zetaray_amd import scene_io
    dst = os.path.join(ROOT, "tests/golden")
    os.makedirs(dst, exist_ok=True)
    for name in ("cornell", "cornell_emissive"):
        sc = scene_io.load_gltf(os.path.join(REF, "Assets/CornellBox", name + ".gltf"))
        scene_io.save_npz(sc, os.path.join(dst, name + ".npz"))
        print(name, "tris", sc.num_tris, "emissives", len(sc.emissives))
    # cornell.gltf through the C++ loader (zetaray_amd/host/zr_scene_io.cpp): the same scene with the reference's real floor, the BC7
    # checkerboard (Assets/CornellBox/compressed/checkerboard.dds, 1024^2, 11 mips) decoded into the texel heap
    sc, offs = scene_io.load_gltf_native(os.path.join(REF, "Assets/CornellBox", "cornell.gltf"))
    assert offs == dict(base_color=0, normal=1, metallic_roughness=1, emissive=1)
    scene_io.save_npz(sc, os.path.join(dst, "cornell_textured.npz"))
    print("cornell_textured: textures", len(sc.textures), "texel bytes", sc.texels.size)
    print("ok")


if __name__ == "__main__":
    main()
