"""Cost of one fused halo exchange on ONE device: the C++ HaloExchange node (pack kernel -> grouped ncclSend / ncclRecv -> unpack kernel) run by a
world of one rank that sends the strips of an 8-way 1080p tile (three neighbours: 480 x 32, 32 x 544, 32 x 32 pixels of 62 B) to itself.  No xGMI
hop is in these numbers -- they are the fixed cost of the exchange (launches, RCCL's own kernel), which is what a small tile pays per frame.
Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zetaray_amd import api, scene_io, tiling, wire

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = scene_io.load_npz(os.path.join(root, "tests", "golden", "cornell_emissive.npz"))
w, h = 544, 608        # a 480 x 544 tile + its 32-px apron
r = api.Renderer(sc, w, h, params=wire.default_params(), integrator=api.INTEGRATOR_RESTIR_PT)
for f in (1, 2):
    r.render_frame(scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives)))
p = r.p_indirect
plan = [(0, (32, 32, 480, 32), (32, 0, 480, 32)), (0, (32, 32, 32, 544), (0, 32, 32, 544)), (0, (32, 32, 32, 32), (0, 0, 32, 32))]
nh = tiling.NativeHalo(p, r.gbuffer, 0, 1, 0, plan)
for _ in range(10):
    nh.run(api.HALO_FINAL)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
t0 = time.perf_counter(); e0.record()
for _ in range(n):
    nh.run(api.HALO_FINAL)
e1.record(); t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(json.dumps({"exchange": "8-way 1080p tile's three strips to itself over RCCL, one device", "bytes_sent": nh.send_bytes, "gpu_us_per_exchange": round(e0.elapsed_time(e1) / n * 1e3, 1),
                  "host_enqueue_us_per_exchange": round(t_enq / n * 1e6, 1)}))
nh.close()
