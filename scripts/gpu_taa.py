import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zetaray_amd import scene_io, wire, api
sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
for (w, h) in ((1920, 1080), (3840, 2160)):
    r = api.Renderer(sc, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_PATH_TRACING)
    taa = r.enable_taa(0.1)
    taa.enable_timing(True)
    r.p_composit.enable_timing(True)
    acc = []
    for f in range(1, 12):
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), jitter=(0.1 * (f % 3 - 1), 0.1 * (f % 2)))
        r.render_frame(cb)
        t = taa.timings()
        if f > 3: acc.append(t["taa"][0])
    ms = float(np.median(acc))
    px = w * h
    print(f"{w}x{h}: taa {ms * 1e3:.1f} us = {px * (16 + 4 + 4 + 8 + 8) / ms / 1e6:.0f} GB/s algorithmic (40 B/px)", flush=True)
