import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import zro
from zetaray_amd import api, scene_io, wire
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def frame(sc, w, h, f): return scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives))
def run(mode):
    sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
    w, h = 128, 96
    prm, dprm = wire.default_params(), wire.default_params_di()
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    di = r.enable_direct(dprm)
    osc = zro.OracleScene(sc)
    opt, odi = zro.OracleRPT(osc, w, h), zro.OracleRDI(osc, w, h)
    idx = [i for i in range(len(sc.instances)) if sc.instances["base_emissive_tri_offset"][i] != 0xFFFFFFFF][0]
    t0, xf = sc.instances["translation"][idx].copy(), {}
    s_upd, s_ren = torch.cuda.Stream(), torch.cuda.Stream()
    su = s_upd.cuda_stream if mode != "same" else s_ren.cuda_stream
    hip_path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    print("hip runtime:", sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)))
    hip = C.CDLL(hip_path)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    nbytes = w * h * 16
    frames, wants, prev = [], [], None
    for f in range(1, 7):
        if f >= 2:
            a = 0.2 * (f - 1)
            inst, xw, first, tris = scene_io.move_emissive_instance(sc, idx, translation=t0 + np.float32([0.05 * (f - 1), -0.02 * (f - 1), 0.03 * (f - 1)]),
                                                                     rotation=np.array([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], np.float32), xform_of=xf)
            r.scene.update_emissives(tris, first, stream=su)
            r.scene.update_instances(inst, xw, stream=su)
            osc.update_emissives(tris, first); osc.update_instances(inst, xw)
        cb = frame(sc, w, h, f)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        r.render_frame(cb, stream=s_ren.cuda_stream)
        snap = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
        torch.cuda.current_stream().synchronize()
        pt_ptr, di_ptr = r.p_indirect.output_ptr()[0], di.output_ptr()[0]
        e1 = hip.hipMemcpyAsync(snap.data_ptr(), pt_ptr, nbytes, 3, s_ren.cuda_stream)
        e2 = hip.hipMemcpyAsync(snap.data_ptr() + nbytes, di_ptr, nbytes, 3, s_ren.cuda_stream)
        if mode == "sync": torch.cuda.synchronize()
        frames.append(snap)
        wants.append((opt.render(cb, prm).copy(), odi.render(cb, dprm).copy()))
        if mode == "sync":
            direct = r.final()
            print(" frame", f, "hip rc", e1, e2, "direct download mism:", int((direct.view(np.uint32) != wants[-1][0].view(np.uint32)).any(axis=2).sum()))
    torch.cuda.synchronize()
    for f, (snap, (want, want_di)) in enumerate(zip(frames, wants), 1):
        got = snap.cpu().numpy().view(np.float32).reshape(2, h, w, 4)
        print(mode, "frame", f, "PT mism px", int((got[0].view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()), "DI mism px", int((got[1].view(np.uint32) != want_di.view(np.uint32)).any(axis=2).sum()),
              "got max", float(got[0][..., :3].max()), "want max", float(want[..., :3].max()))
for m in sys.argv[1:] or ["sync", "same", "cross"]:
    run(m)
