"""Where the host's time per submitted frame goes (Python binding; plain order): per C-ABI call, averaged over frames.  python tools/host_submit_breakdown.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zetaray_amd import api, scene_io, wire
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
W, H = 1920, 1080
r = api.Renderer(sc, W, H, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
for f in range(1, 40):
    r.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives)))
torch.cuda.synchronize()
acc = {}
def T(name, fn):
    t0 = time.perf_counter(); fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
N = 200
t00 = time.perf_counter()
for f in range(40, 40 + N):
    cb = None
    def mk():
        global cb
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives))
    T("make_frame_constants", mk)
    T("bind_post_inputs+sky", lambda: (r._bind_post_inputs(), r.render_sky(cb, None)))
    T("gbuffer.render", lambda: r.p_gbuffer.render(cb, r.scene, r.gbuffer, None))
    T("indirect.stage CANDIDATES", lambda: r.p_indirect.render_stage(cb, r.scene, r.gbuffer, api.STAGE_CANDIDATES, None))
    T("indirect.stage TEMPORAL_REUSE", lambda: r.p_indirect.render_stage(cb, r.scene, r.gbuffer, api.STAGE_TEMPORAL_REUSE, None))
    T("indirect.stage SPATIAL", lambda: r.p_indirect.render_stage(cb, r.scene, r.gbuffer, api.STAGE_SPATIAL | api.STAGE_SPATIAL2, None))
tot = time.perf_counter() - t00
torch.cuda.synchronize()
print(json.dumps({"host_ms_per_frame": round(tot / N * 1e3, 4), **{k: round(v / N * 1e3, 4) for k, v in acc.items()}}))
