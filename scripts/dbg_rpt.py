import sys, numpy as np
sys.path.insert(0, ".")
from zetaray_amd import api, scene_io, wire
from oracle import zro
sc = scene_io.make_synthetic_scene(num_tris=3000, num_emissive=1500, seed=11)
osc = zro.OracleScene(sc, force_bvh=True)
W, H = 96, 64
def run(mode, nb=6, gbn=8, rr=True):
    prm = wire.default_params(); prm.max_non_tr_bounces, prm.max_glossy_tr_bounces = nb, gbn
    if mode == "temporal": prm.flags &= ~wire.IND_SPATIAL_RESAMPLE
    if mode == "noboil": prm.flags &= ~wire.IND_BOILING_SUPPRESSION
    if not rr: prm.flags &= ~wire.IND_RUSSIAN_ROULETTE
    r = api.Renderer(sc, W, H, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    r2 = api.Renderer(sc, W, H, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    o = zro.OracleRPT(osc, W, H)
    for f in range(1, 4):
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
        r.render_frame(cb); r2.render_frame(cb)
        got = r.final(); got2 = r2.final(); want = o.render(cb, prm)
        bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
        nd = (got.view(np.uint32) != got2.view(np.uint32)).any(axis=2).sum()
        print(mode, nb, gbn, "rr", rr, "frame", f, "bad px", bad.sum(), "gpu-vs-gpu diff", nd)
        for nm in ("A", "B", "C", "D", "E", "F", "G", "neighbor", "target"):
            a, b = r.p_indirect.download_plane(nm), o.plane(nm)
            if nm == "A": a, b = a & 0xffffff, b & 0xffffff
            d = (a != b).any(axis=2)
            if d.sum(): print("    plane", nm, "differs at", d.sum(), "px")
        if bad.sum():
            A = o.plane("A")[..., 0]; ys, xs = np.nonzero(bad)
            mr = o.prev[0][2]
            for y, x in list(zip(ys, xs))[:40]:
                a = int(A[y, x]); print("    px", x, y, "k", (a & 15), "M", (a >> 4) & 15, "lobes", (a >> 8) & 7, (a >> 11) & 7, "lt_k", (a >> 14) & 3, "lt_k1", (a >> 16) & 3,
                      "flags", int(mr[y, x] & 0xff), "got", got[y, x, :3], "want", want[y, x, :3])
            return
run("temporal", 3, 4); run("temporal", 2, 2); run("temporal", 1, 1)
