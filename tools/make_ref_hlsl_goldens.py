"""Generates tests/golden/ref_hlsl_kat.npz + ref_layout.txt from the REFERENCE's own code compiled in place:

  oracle/_ref/libzref_hlsl.so   /root/reference/Source/ZetaRenderPass/Common/{Math,Sampling,RT,BSDF,BSDFSampling,...}.hlsli compiled as C++
                                (oracle/ref_hlsl/: hlsl2cpp.py + hlsl_shim.h; `make -C oracle -f _ref.mk`)
  oracle/_ref/libzref.so        offsetof / sizeof of the C++ side of RtCommon.h, Material.h, Vertex.h, FrameConstants.h

The inputs of every probe family are regenerated from seeds (tools/kat_inputs.py); only the reference's OUTPUTS are stored.  The
committed files let the pins be checked where /root/reference does not exist (GPU box); tests/test_ref_hlsl_pins.py also runs the
comparison live (more inputs) when oracle/_ref/ is present.  Run in the build container: python tools/make_ref_hlsl_goldens.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kat_inputs as K  # noqa: E402

N_GOLDEN = 4096


def ref_lib():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzref_hlsl.so"))
    rho = np.fromfile(os.path.join(ROOT, "zetaray_amd", "assets", "rho_lut_u16.bin"), np.uint16)
    L._rho = rho      # keep alive
    L.zrefh_bind_rho(C.c_void_p(rho.ctypes.data), 64, 32, 16)
    return L


def run_ref(L, fam, x):
    out = np.zeros((len(x), K.FAMILIES[fam][1]), np.float32)
    getattr(L, "zrefh_kat_" + fam)(C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), len(x))
    return out


def main():
    L = ref_lib()
    res = {}
    for fam, (gen, _) in K.FAMILIES.items():
        x = np.ascontiguousarray(gen(N_GOLDEN))
        res[fam] = run_ref(L, fam, x).view(np.uint32)      # bit patterns (NaN-safe)
        print(fam, res[fam].shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_hlsl_kat.npz"), **res)
    L2 = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzref.so"))
    buf = C.create_string_buffer(1 << 16)
    n = L2.zref_layout(buf, len(buf))
    assert n > 0
    open(os.path.join(ROOT, "tests", "golden", "ref_layout.txt"), "w").write(buf.value.decode())
    print("layout lines:", buf.value.decode().count("\n"))


if __name__ == "__main__":
    main()
