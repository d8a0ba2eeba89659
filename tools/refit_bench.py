"""Latency of zr_scene_update_instances on the BASELINE config-4 stand-in (262k-triangle atrium): device refit (default) next to the host rebuild
(ZR_SCENE_UPDATE=rebuild), and the ReSTIR PT frame time on the refit / rebuilt tree after the largest non-emissive clutter instance moved.
Prints one JSON line; scripts/gpu_refit.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab] runs it once per mode.  GPU only."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zetaray_amd import api, scene_io, wire


def main():
    w, h = 1920, 1080
    sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
    prm = wire.default_params()
    prm.presampling, prm.num_sample_sets, prm.sample_set_size = 1, 128, 512
    r = api.Renderer(sc, w, h, params=prm, integrator=api.INTEGRATOR_RESTIR_PT)
    cand = [i for i in range(1, len(sc.instances)) if sc.instance_mask[i] & wire.SUBGROUP_NON_EMISSIVE]
    idx = max(cand, key=lambda i: int(sc.instance_num_tris[i]))
    t0, xf = sc.instances["translation"][idx].copy(), {}
    import torch
    upd, frames, kern = [], [], {"static": {}, "moving": {}}
    n_static, n_moving = 24, int(os.environ.get("REFIT_MOVING_FRAMES", "16"))
    r.p_gbuffer.enable_timing(True); r.p_indirect.enable_timing(True)
    for f in range(1, n_static + n_moving + 1):
        moving = f > n_static
        if moving:
            k = f - n_static
            ang = 0.05 * k
            q = np.array([0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)], np.float32)
            scene_io.move_instance(sc, idx, translation=t0 + np.float32([0.02 * k, 0.0, 0.01 * k]), rotation=q, xform_of=xf)
            torch.cuda.synchronize(); a = time.perf_counter()
            r.scene.update_instances(sc.instances, sc.instance_to_world)
            torch.cuda.synchronize(); upd.append((time.perf_counter() - a) * 1e3)
        cb = scene_io.make_frame_constants(w, h, frame_num=f, num_emissives=len(sc.emissives), cam_pos=(0, 0, -3.5))
        torch.cuda.synchronize(); a = time.perf_counter()
        r.render_frame(cb)
        torch.cuda.synchronize(); frames.append((time.perf_counter() - a) * 1e3)
        if (not moving and f > 12) or (moving and f > n_static + 4):      # steady state of either phase
            for name, (ms, launches) in {**r.p_gbuffer.timings(), **r.p_indirect.timings()}.items():
                kern["moving" if moving else "static"].setdefault(name, []).append(ms)
    kms = {ph: {k: round(float(np.mean(v)), 3) for k, v in d.items() if np.mean(v) > 0.05} for ph, d in kern.items()}
    print(json.dumps({"mode": os.environ.get("ZR_SCENE_UPDATE", "refit"), "background_rebuilds": list(r.scene.background_rebuild_stats()), "instance": int(idx), "instance_tris": int(sc.instance_num_tris[idx]),
                      "bvh": list(r.scene.bvh_info()), "update_ms": [round(x, 3) for x in upd], "update_ms_median": round(float(np.median(upd)), 3), "update_ms_mean": round(float(np.mean(upd[4:])), 3),
                      # ZR_BVH_GROUP (and every other A/B switch) is read by the EXPERIMENTS build only (libzetaray_amd_exp.so, ZR_EXP_ENV); the product library groups always
                      "library": os.path.basename(api.LIB_PATH),
                      "group": (os.environ.get("ZR_BVH_GROUP", "1") if os.path.basename(api.LIB_PATH) == "libzetaray_amd_exp.so" else "1 (product library: the ZR_BVH_GROUP switch exists in libzetaray_amd_exp.so only)"), "frame_ms_moving_series": [round(x, 2) for x in frames[n_static:]],
                      "frame_ms_static": round(float(np.median(frames[12:n_static])), 3), "frame_ms_moving": round(float(np.median(frames[n_static + 4:])), 3),
                      "kernel_ms_static": kms["static"], "kernel_ms_moving": kms["moving"]}))


if __name__ == "__main__":
    main()
