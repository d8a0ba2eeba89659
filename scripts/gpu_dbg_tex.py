import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import zro
from zetaray_amd import scene_io, wire, api
import test_gpu_parity as T
for tex in (False, True):
    for nb, gbn in ((3, 4), (5, 7)):
        sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=0, seed=11, open_top=True)
        offs = scene_io.add_test_textures(sc) if tex else dict(base_color=0, normal=0, metallic_roughness=0, emissive=0)
        w, h = 96, 64
        prm = wire.default_params(); prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gbn
        orc = zro.OracleScene(sc, force_bvh=True)
        o = zro.OracleRPT(orc, w, h)
        r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
        for f, cb in T._textured_frames(sc, offs, w, h, 4, (0.0, 2.0, -3.5), (0.3, -0.8, 0.4)):
            r.p_indirect.read_counters(reset=True)
            r.render_frame(cb)
            got = r.final()
            orc.sky_lut(cb, 256, 128)
            want = o.render(cb, prm)
            mism = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
            line = f"tex={tex} bounces={nb}/{gbn} frame {f}: radiance mism {mism} counters {tuple(r.p_indirect.read_counters()) == tuple(o.counters)}"
            for nm in "ABCDEFG":
                pa, pb = r.p_indirect.download_plane(nm), o.plane(nm)
                if nm == "A": pa, pb = pa & 0xffffff, pb & 0xffffff
                a8, b8 = pa.view(np.uint8).reshape(h, w, -1), pb.view(np.uint8).reshape(h, w, -1)
                d = (a8 != b8).any(axis=2)
                if d.any():
                    y, x = np.argwhere(d)[0]
                    line += f" | {nm}: {int(d.sum())} px, first ({x},{y}) gpu={pa.reshape(h, w, -1)[y, x]} ora={pb.reshape(h, w, -1)[y, x]}"
            print(line, flush=True)
