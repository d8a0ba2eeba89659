"""Regenerates tests/golden/render_emissive_64.npz: the oracle's G-buffer planes, path-traced FINAL and ray counters
for cornell_emissive at 64x64, frame 1 (regression pin of the oracle; run after an intentional oracle change)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zro  # noqa: E402
from zetaray_amd import scene_io, wire  # noqa: E402

sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
o = zro.OracleScene(sc)
cb = scene_io.make_frame_constants(64, 64, frame_num=1, num_emissives=len(sc.emissives))
arrays, planes = o.gbuffer(cb)
final, cnt = o.pathtrace(cb, planes, wire.default_params())
out = {"gb_" + n: a for n, a in zip(wire.GB_PLANE_NAMES, arrays)}
out["final"] = final
out["counters"] = np.array(cnt, np.uint64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "render_emissive_64.npz"), **out)
print("ok", final[..., :3].mean())


def config1(scene_name, out_name, w=256, h=256):
    """BASELINE config 1: the Cornell box at 256x256, camera (0, 1.2, -4.043) looking down +z, vfov 60 deg, jitter off,
    PATH_TRACING (K9) 1 spp, FrameNum 1, Accumulate 0, the reference's default sun / atmosphere.  Stores the oracle's G-buffer planes, the
    sky-view LUT (sun + sky scene only), FINAL and the ray counters."""
    scn = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", scene_name))
    osc = zro.OracleScene(scn)
    cbf = scene_io.make_frame_constants(w, h, frame_num=1, num_emissives=len(scn.emissives))
    res = {}
    if len(scn.emissives) == 0:
        res["sky_lut"] = osc.sky_lut(cbf, 256, 128)
    arr, pl = osc.gbuffer(cbf)
    fin, c = osc.pathtrace(cbf, pl, wire.default_params())
    res.update({"gb_" + n: a for n, a in zip(wire.GB_PLANE_NAMES, arr)})
    res["final"] = fin
    res["counters"] = np.array(c, np.uint64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", out_name), **res)
    print(out_name, "ok", fin[..., :3].mean(), c)


config1("cornell.npz", "config1_cornell_256.npz")
config1("cornell_emissive.npz", "config1_cornell_emissive_256.npz")
