import numpy as np, sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import zro
from zetaray_amd import scene_io, wire, api
import test_gpu_parity as T
sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=0, seed=11, open_top=True)
offs = scene_io.add_test_textures(sc)
w, h = 96, 64
prm = wire.default_params(); prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 5, 7
prm.flags &= ~(wire.IND_SPATIAL_RESAMPLE | wire.IND_TEMPORAL_RESAMPLE)
orc = zro.OracleScene(sc, force_bvh=True)
o = zro.OracleRPT(orc, w, h)
r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
for f, cb in T._textured_frames(sc, offs, w, h, 1, (0.0, 2.0, -3.5), (0.3, -0.8, 0.4)):
    r.render_frame(cb); got = r.final()
    orc.sky_lut(cb, 256, 128); want = o.render(cb, prm)
    A = o.plane("A").reshape(h, w, -1)[..., 0]
    k = (A & 0xf); lt_k = (A >> 14) & 3; lt_k1 = (A >> 16) & 3
    case = np.where(lt_k != 0, 3, np.where(lt_k1 != 0, 2, 1)); case = np.where(k == 15, 0, case)
    Dg, Do = r.p_indirect.download_plane("D").reshape(h, w, 4), o.plane("D").reshape(h, w, 4)
    Eg, Eo = r.p_indirect.download_plane("E").reshape(h, w), o.plane("E").reshape(h, w)
    bad = (Dg != Do).any(axis=2) | (Eg != Eo)
    print("all   :", sorted(collections.Counter(zip((k + 2).ravel().tolist(), case.ravel().tolist())).items()))
    print("bad   :", sorted(collections.Counter(zip((k + 2)[bad].tolist(), case[bad].tolist())).items()))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:6]:
        print((x, y), "k", int(k[y, x]) + 2, "case", int(case[y, x]), "gpu D.w %08x E %04x" % (Dg[y, x, 3], Eg[y, x]), "ora D.w %08x E %04x" % (Do[y, x, 3], Eo[y, x]),
              "final gpu", got[y, x, :3], "ora", want[y, x, :3])
