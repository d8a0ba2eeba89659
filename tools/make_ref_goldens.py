"""Generates tests/golden/ref_pins.npz from the REFERENCE's own compiled code (oracle/_ref/libzref.so, built by
`make -C oracle -f _ref.mk` from /root/reference/Source/ZetaCore/Math/{Common,Sampling}.cpp).  These vectors pin the
oracle's alias table / Kahan sum / oct32 / half conversions to the reference bit for bit; the committed .npz lets the
pin be checked where /root/reference does not exist (GPU box).  Run in the build container: python tools/make_ref_goldens.py
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def aligned(n, phase):
    """float32 array of n elements whose data pointer is `phase` floats short of the next 32-byte boundary."""
    raw = np.zeros(n + 16, np.float32)
    base = raw.ctypes.data
    off = ((32 - (base & 31)) & 31) // 4            # floats to reach alignment
    start = (off - phase) % 8
    a = raw[start:start + n]
    return a


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzref.so"))
    L.zref_kahan_sum.restype = C.c_float
    L.zref_kahan_sum.argtypes = [C.c_void_p, C.c_uint64]
    L.zref_alias_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zref_oct32_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.zref_oct32_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.zref_f32_to_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.zref_align_phase.restype = C.c_size_t
    L.zref_align_phase.argtypes = [C.c_void_p]

    rng = np.random.default_rng(20250925)
    out = {}
    cases = [("fixed6", np.array([1, 22, 4, 8, 3.5, 10], np.float32), 0)]
    for n in (1, 2, 3, 17, 50, 333, 1000, 4099):
        # phase > 0 only for n >= 17: the reference's Normalize/KahanSum index out of bounds when N < phase
        for phase in ((0, 3) if n >= 17 else (0,)):
            w = rng.uniform(0.01, 50.0, n).astype(np.float32) ** rng.choice([1, 2])
            cases.append((f"n{n}_p{phase}", w.astype(np.float32), phase))
    names = []
    for name, w, phase in cases:
        n = len(w)
        buf = aligned(n, phase)
        buf[:] = w
        assert L.zref_align_phase(buf.ctypes.data) == phase
        ks = L.zref_kahan_sum(buf.ctypes.data, n)
        p_curr = np.zeros(n, np.float32)
        p_orig = np.zeros(n, np.float32)
        alias = np.zeros(n, np.uint32)
        L.zref_alias_build(buf.ctypes.data, n, p_curr.ctypes.data, p_orig.ctypes.data, alias.ctypes.data)
        out[f"alias_{name}_w"] = w
        out[f"alias_{name}_phase"] = np.uint32(phase)
        out[f"alias_{name}_kahan"] = np.float32(ks)
        out[f"alias_{name}_p_curr"] = p_curr
        out[f"alias_{name}_p_orig"] = p_orig
        out[f"alias_{name}_alias"] = alias
        names.append(name)
    out["alias_cases"] = np.array(names)

    v = rng.normal(size=(2000, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    v = np.ascontiguousarray(v.astype(np.float32))
    enc = np.zeros((len(v), 2), np.uint16)
    L.zref_oct32_encode(v.ctypes.data, enc.ctypes.data, len(v))
    dec = np.zeros((len(v), 3), np.float32)
    L.zref_oct32_decode(enc.ctypes.data, dec.ctypes.data, len(v))
    out["oct_in"], out["oct_enc"], out["oct_dec"] = v, enc, dec

    x = np.concatenate([rng.normal(size=4000) * 10.0 ** rng.uniform(-9, 6, 4000),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 6.1e-5, 6.09e-5, np.inf, -np.inf]]).astype(np.float32)
    h = np.zeros(len(x), np.uint16)
    L.zref_f32_to_f16(x.ctypes.data, h.ctypes.data, len(x))
    out["half_in"], out["half_out"] = x, h

    L.zref_halton.restype = C.c_float
    L.zref_halton.argtypes = [C.c_int, C.c_int]
    out["halton"] = np.array([[L.zref_halton(i + 1, 2), L.zref_halton(i + 1, 3)] for i in range(64)], np.float32)

    L.zref_unorm4_from_normalized.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    q = rng.normal(size=(3000, 4)).astype(np.float32)
    q = np.ascontiguousarray((q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32))
    qu = np.zeros((len(q), 4), np.uint16)
    L.zref_unorm4_from_normalized(q.ctypes.data, qu.ctypes.data, len(q))
    out["unorm4_in"], out["unorm4_out"] = q, qu
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_pins.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
