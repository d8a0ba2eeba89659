"""The bandwidth-bound kernels of the frame against the HBM roofline: Compositing (+ firefly filter), TAA, the denoise pass and the G-buffer-sized
streaming kernels around them, at 3840 x 2160 on the Cornell box.  Per kernel: ms per launch (hipEvents inside the library), the algorithmic bytes
per pixel it must move, GB/s and the fraction of 8 TB/s.  Prints one JSON line.  GPU only."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zetaray_amd import api, scene_io, wire

# algorithmic bytes per pixel and launch: planes read once + planes written (neighbour taps are re-reads other pixels load once)
BYTES = {"compositing": 16 + 4 + 2 + 16,                 # indirect in, base colour + flags of the G-buffer, composited out (DI terms absent here)
         "firefly_filter": 16 + 16,                           # composited in, filtered out (5 x 5 taps from an LDS tile)
         "taa": 16 + 4 + 4 + 8 + 8,                    # signal, depth, motion, history in; RGBA16F out
         "denoise_temporal": 60 + 36, "denoise_variance": 32 + 16, "denoise_atrous": 20 + 16}      # definition 3 of zr_svgf.h (bench.py DENOISE_PIXEL_BYTES)
W, H = 3840, 2160
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
r = api.Renderer(sc, W, H, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
r.enable_compositing(firefly_filter=True)
taa = r.enable_taa(0.1)
dn = r.enable_denoise()
passes = [r.p_composit, taa, dn]
for p in passes:
    p.enable_timing(True)
acc = {}
for f in range(1, 17):
    r.render_frame(scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives)))
    torch.cuda.synchronize()
    if f > 6:
        for p in passes:
            for name, (ms, n) in p.timings().items():
                a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += n
out = {}
for name, (ms, n) in acc.items():
    per = ms / n
    b = BYTES.get(name)
    out[name] = {"ms_per_launch": round(per, 4), "launches_per_frame": n / 10,
                 "GBs": round(b * W * H / (per * 1e-3) / 1e9, 1) if b else None, "hbm_frac": round(b * W * H / (per * 1e-3) / 8e12, 3) if b else None, "bytes_per_px": b}
print(json.dumps({"size": [W, H], "kernels": out}))
