import sys, numpy as np
sys.path.insert(0, ".")
from zetaray_amd import api, scene_io, wire
from oracle import zro
from tests.hostexec import zhx
sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
osc = zro.OracleScene(sc, force_bvh=True)
hx = zhx.HostExecScene(sc, osc.alias)
W, H = 96, 64
prm = wire.default_params(); prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = 2, 2
prm.flags &= ~wire.IND_SPATIAL_RESAMPLE
r = api.Renderer(sc, W, H, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
h = zhx.HostExecRPT(hx, W, H)
for f in range(1, 3):
    cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
    r.render_frame(cb); h.render(cb, prm)
    for nm in ("ctn_A", "ctn_B", "ctn_C", "ctn_D", "ntc_A", "ntc_B", "ntc_C", "ntc_D", "A", "B", "C", "D", "E", "F", "G", "target"):
        a, b = r.p_indirect.download_plane(nm), h.plane(nm)
        if nm == "A": a, b = a & 0xffffff, b & 0xffffff
        d = (a != b).any(axis=2)
        if d.sum():
            ys, xs = np.nonzero(d)
            print("frame", f, "plane", nm, "differs at", d.sum(), "px; first:", [(int(x), int(y)) for y, x in zip(ys, xs)][:6])
            y, x = ys[0], xs[0]
            print("     gpu", a[y, x], "host", b[y, x])
