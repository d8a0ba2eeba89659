"""Seeded random inputs of the function-level known-answer probes (oracle/zro_kat_layout.h).  Shared by tools/make_ref_goldens.py
(which runs them through the reference's own shader code compiled as C++, oracle/_ref/libzref_hlsl.so) and tests/test_ref_pins.py."""
import numpy as np

SAMPLING_IN, SAMPLING_OUT = 4, 32
MATH_IN, MATH_OUT = 21, 48
RT_IN, RT_OUT = 17, 12
BSDF_IN, BSDF_OUT = 26, 61


def _unit(rng, n, k=3):
    v = rng.normal(size=(n, k))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def _bits(u32):
    return np.asarray(u32, np.uint32).view(np.float32)


def sampling(n, seed=1):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, SAMPLING_IN), np.float32)
    x[:, 0:2] = rng.random((n, 2), np.float32)
    x[:8, 0:2] = [[0, 0], [0.5, 0.5], [0, 0.999999], [0.999999, 0], [0.25, 0.75], [0.5, 0.25], [1e-7, 1e-7], [0.75, 0.75]]
    x[:, 2] = rng.uniform(-0.5, 0.9999, n).astype(np.float32)
    x[:, 3] = _bits(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32))
    return x


def math(n, seed=2):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, MATH_IN), np.float32)
    x[:, 0:3] = _unit(rng, n)
    x[:6, 0:3] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, -1, 0], [0.6, 0, -0.8], [0, 0.6, 0.8]]
    x[:, 3:7] = _unit(rng, n, 4)
    x[:, 7:10] = rng.uniform(0.1, 4.0, (n, 3)).astype(np.float32)
    x[:, 10:13] = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    x[:, 13] = rng.uniform(-1, 1, n).astype(np.float32)
    x[:4, 13] = [0.0, -0.0, 1.0, -1.0]
    x[:, 14:16] = rng.uniform(-0.1, 1.1, (n, 2)).astype(np.float32)
    x[:, 16] = _bits(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32))
    x[:, 17] = _bits(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32))
    x[:, 18:21] = _unit(rng, n)
    return x


def rt(n, seed=3):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, RT_IN), np.float32)
    scale = 10.0 ** rng.uniform(-3, 2, (n, 1))
    x[:, 0:3] = (rng.uniform(-1, 1, (n, 3)) * scale).astype(np.float32)
    x[:, 3:6] = _unit(rng, n)
    x[:, 6:9] = _unit(rng, n)
    x[:, 9:12] = rng.uniform(0, 5, (n, 3)).astype(np.float32)
    x[:4, 9:12] = [[0, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]]
    x[:, 12:14] = rng.random((n, 2)).astype(np.float32) * 0.999
    x[:, 14] = np.tan(np.radians(rng.uniform(15, 50, n))).astype(np.float32)
    x[:, 15:17] = rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32)
    return x


def bsdf(n, seed=4):
    """material / direction configurations covering every lobe combination: metal, dielectric (diffuse + gloss), thin-walled,
    specular transmission (entering / exiting), with and without coat, rough and (near-)specular"""
    rng = np.random.default_rng(seed)
    x = np.zeros((n, BSDF_IN), np.float32)
    nrm = _unit(rng, n)
    x[:, 0:3] = nrm
    # wo mostly on the normal's side, wi anywhere (reflection and transmission configurations)
    wo = _unit(rng, n)
    flip = (np.sum(wo * nrm, axis=1) < 0) & (rng.random(n) < 0.9)
    wo[flip] = -wo[flip]
    x[:, 3:6] = wo
    x[:, 6:9] = _unit(rng, n)
    kind = rng.integers(0, 5, n)          # 0 metal, 1 dielectric opaque, 2 thin-walled, 3 specular transmission, 4 coated dielectric
    x[:, 9] = (kind == 0)
    rough = rng.uniform(0.0, 1.0, n)
    spec = rng.random(n) < 0.15
    rough[spec] = rng.uniform(0.0, 0.04, int(spec.sum()))
    x[:, 10] = rough
    x[:, 11:14] = rng.uniform(0.02, 1.0, (n, 3))
    x[:, 14] = (kind == 3)
    coated = (kind == 4) | (rng.random(n) < 0.25)
    x[:, 15] = np.where(coated, rng.uniform(0.05, 1.0, n), 0.0)
    x[:, 16:19] = rng.uniform(0.2, 1.0, (n, 3))
    x[:, 19] = np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0.0, 0.6, n))
    x[:, 20] = rng.uniform(1.1, 2.2, n)
    x[:, 21] = rng.uniform(1.05, 2.4, n)
    x[:, 22] = np.where((kind == 3) & (rng.random(n) < 0.5), rng.uniform(0.1, 2.0, n), 0.0)
    x[:, 23] = np.where(kind == 2, rng.uniform(0.1, 1.0, n), 0.0)
    x[:, 24] = (kind == 3) & (rng.random(n) < 0.4)
    x[:, 25] = _bits(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32))
    return x.astype(np.float32)


FAMILIES = {"sampling": (sampling, SAMPLING_OUT), "math": (math, MATH_OUT), "rt": (rt, RT_OUT), "bsdf": (bsdf, BSDF_OUT)}
