"""Tree quality of the host BVH builder (zr_bvh.h), measured on the CPU: the builder's tree for a scene, traversed by the ordered stack traversal of
zr_dev_scene.h (tests/hostexec: the product's own code compiled for the host) over a fixed sample of rays of the benchmark's frame -- primary rays, a
diffuse bounce from their hits (closest-hit), and shadow segments between random hit points (any-hit) -- with the steps counted.  What the GPU kernels pay
per ray is proportional to these counts (the section profiler's nodes_per_ray / tris_per_ray are the same quantities measured in the kernels,
profiles/r06g_section_profile_atrium.json: 21.3 nodes and 2.7 triangles per ray in K11 on the atrium).  Builder knobs are environment variables read at
construction (ZR_BVH_*), so each variant runs in a process of its own:

    python tools/bvh_quality.py [--scene synthetic|cornell] [--rays 200000]          one JSON line
    python tools/bvh_quality.py --sweep                                              the table of variants (children processes)
    python tools/bvh_quality.py --scene cornell --dump                               the wide tree itself
"""
import argparse, ctypes as C, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample_rays(hx, cb, n, rng):
    """primary rays through random pixels, a uniformly random bounce direction from each hit, and segments between pairs of hit points"""
    from tests.hostexec import zhx
    w, h = int(cb["render_width"]), int(cb["render_height"])
    px, py = rng.integers(0, w, n), rng.integers(0, h, n)
    ndc_x, ndc_y = ((px + 0.5) / w) * 2 - 1, -(((py + 0.5) / h) * 2 - 1)
    d = np.stack([ndc_x * float(cb["aspect_ratio"]) * float(cb["tan_half_fov"]), ndc_y * float(cb["tan_half_fov"]), np.ones(n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o3 = np.tile(np.asarray(cb["camera_pos"], np.float64), (n, 1))
    prim = np.concatenate([o3, np.zeros((n, 1)), d, np.full((n, 1), 3.0e38)], 1).astype(np.float32)
    hits = hx.trace_closest(prim)
    ok = hits[:, 3] != 0xFFFFFFFF
    t = hits[ok, 0].view(np.float32)
    p = prim[ok, 0:3] + prim[ok, 4:7] * (t[:, None] * 0.999)
    d2 = rng.normal(size=p.shape)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    sec = np.concatenate([p, np.full((len(p), 1), 1e-4), d2, np.full((len(p), 1), 3.0e38)], 1).astype(np.float32)
    # third vertices: where the bounce rays land; shadow segments from the first hit to a random OTHER path's second hit
    h2 = hx.trace_closest(sec)
    ok2 = h2[:, 3] != 0xFFFFFFFF
    q = sec[ok2, 0:3] + sec[ok2, 4:7] * (h2[ok2, 0].view(np.float32)[:, None] * 0.999)
    a = p[rng.integers(0, len(p), len(q))]
    seg = q - a
    L = np.linalg.norm(seg, axis=1, keepdims=True)
    sh = np.concatenate([a, np.full((len(q), 1), 1e-4), seg / np.maximum(L, 1e-9), L * 0.999], 1).astype(np.float32)
    return prim, sec, sh


def wave_rays(hx, sc, cb, blocks, rng):
    """rays in the order a K11 wave issues them: `blocks` random 16 x 4 pixel blocks (one wave each), their primary rays, two generations of bounce rays from the
    hits (lanes whose path left the scene go idle) and a shadow segment from every first hit to a random point of a random emissive triangle"""
    w, h = int(cb["render_width"]), int(cb["render_height"])
    bx, by = rng.integers(0, w // 16, blocks), rng.integers(0, h // 4, blocks)
    lane = np.arange(64)
    px = (bx[:, None] * 16 + (lane & 15)[None, :]).ravel()
    py = (by[:, None] * 4 + (lane >> 4)[None, :]).ravel()
    n = len(px)
    ndc_x, ndc_y = ((px + 0.5) / w) * 2 - 1, -(((py + 0.5) / h) * 2 - 1)
    d = np.stack([ndc_x * float(cb["aspect_ratio"]) * float(cb["tan_half_fov"]), ndc_y * float(cb["tan_half_fov"]), np.ones(n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o3 = np.tile(np.asarray(cb["camera_pos"], np.float64), (n, 1))
    prim = np.concatenate([o3, np.zeros((n, 1)), d, np.full((n, 1), 3.0e38)], 1).astype(np.float32)
    sets = [("primary", prim, np.ones(n, np.uint8), 0)]
    rays, act = prim, np.ones(n, bool)
    first_hit = None
    for gen in (1, 2):
        hits = hx.trace_closest(rays)
        ok = act & (hits[:, 3] != 0xFFFFFFFF)
        t = hits[:, 0].view(np.float32)
        p = rays[:, 0:3] + rays[:, 4:7] * (np.where(ok, t, 0)[:, None] * 0.999)
        if first_hit is None:
            first_hit = (p.copy(), ok.copy())
        d2 = rng.normal(size=p.shape)
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        rays = np.concatenate([p, np.full((n, 1), 1e-4), d2, np.full((n, 1), 3.0e38)], 1).astype(np.float32)
        act = ok
        sets.append((f"bounce{gen}", rays, act.astype(np.uint8), 0))
    em = sc.emissives
    if len(em):
        v0 = np.asarray(em["vtx0"], np.float64)[rng.integers(0, len(em), n)]
        p, ok = first_hit
        seg = v0 - p
        L = np.linalg.norm(seg, axis=1, keepdims=True)
        sh = np.concatenate([p, np.full((n, 1), 1e-4), seg / np.maximum(L, 1e-9), L * 0.999], 1).astype(np.float32)
        # NEE reaches about half of the lanes (the light sample faces away from the rest)
        sets.append(("shadow", sh, (ok & (rng.random(n) < 0.5)).astype(np.uint8), 1))
    return sets


def vote_stats(hx, sets):
    from tests.hostexec import zhx
    L = zhx.lib()
    L.zhx_trace_vote_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    out, tot = {}, 0.0
    for name, rays, act, anyhit in sets:
        r, a = np.ascontiguousarray(rays, np.float32), np.ascontiguousarray(act, np.uint8)
        res = (C.c_uint64 * 5)()
        L.zhx_trace_vote_stats(hx.h, r.ctypes.data, a.ctypes.data, len(r), 0xFFFFFFFF, anyhit, res)
        calls = max(1, res[4])
        # VALU instructions per call: a node iteration ~ 192 + 25 of loop overhead, a leaf iteration (<= 2 triangles) ~ 140 + 25
        cost = (217.0 * res[0] + 165.0 * res[1]) / calls
        out[name] = {"calls": int(res[4]), "rays_per_call": round(float(a.sum()) / calls, 1), "node_iters_per_call": round(res[0] / calls, 2), "leaf_iters_per_call": round(res[1] / calls, 2),
                     "lane_util": round((res[2] + res[3]) / max(1, 64 * (res[0] + res[1])), 3), "valu_per_call": round(cost)}
        if name != "primary":
            tot += cost
    out["secondary_valu_per_wave"] = round(tot)
    return out


def measure(args):
    from zetaray_amd import scene_io
    from tests.hostexec import zhx
    if args.scene == "synthetic":
        sc = scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium")
        cam = dict(cam_pos=(0, 0, -3.5))
    else:
        sc = scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz"))
        cam = {}
    t0 = time.perf_counter()
    hx = zhx.HostExecScene(sc)
    build_s = time.perf_counter() - t0
    cb = scene_io.make_frame_constants(1920, 1080, frame_num=1, num_emissives=len(sc.emissives), **cam)
    prim, sec, sh = sample_rays(hx, cb, args.rays, np.random.default_rng(0x5EED))
    L = zhx.lib()
    L.zhx_trace_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    out = {"scene": args.scene, "knobs": {k: v for k, v in os.environ.items() if k.startswith("ZR_BVH_")}, "build_s": round(build_s, 2)}
    dg, nn, nt, st = hx.bvh_digest()
    out.update(nodes=nn, tris=nt, stack_need=st)
    tot_n = tot_t = tot_r = 0
    for name, rays, anyhit in (("primary", prim, 0), ("bounce", sec, 0), ("shadow", sh, 1)):
        r = np.ascontiguousarray(rays, np.float32)
        res = (C.c_uint64 * 4)()
        L.zhx_trace_stats(hx.h, r.ctypes.data, len(r), 0xFFFFFFFF, anyhit, res)
        out[name] = {"rays": len(r), "nodes_per_ray": round(res[0] / len(r), 3), "leaves_per_ray": round(res[1] / len(r), 3), "tris_per_ray": round(res[2] / len(r), 3),
                     "hit_frac": round(res[3] / len(r), 3)}
        if name != "primary":
            tot_n += res[0]; tot_t += res[2]; tot_r += len(r)
    # the figure of merit: secondary rays (what K11 / K13 / K14 / K16 trace), a node phase costing ~1.5 two-triangle leaf phases on the device (192 vs ~130 VALU instructions)
    out["secondary_nodes_per_ray"] = round(tot_n / tot_r, 3)
    out["secondary_tris_per_ray"] = round(tot_t / tot_r, 3)
    out["cost"] = round((tot_n * 1.5 + tot_t * 0.5) / tot_r, 3)
    # ... and as the device schedules it: waves of 64 rays voting for the node or the leaf phase (zhx_trace_vote_stats)
    out["voted"] = vote_stats(hx, wave_rays(hx, sc, cb, args.waves, np.random.default_rng(0xBEEF)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="synthetic")
    ap.add_argument("--rays", type=int, default=100000)
    ap.add_argument("--waves", type=int, default=400, help="16 x 4 pixel blocks whose rays are replayed through the voted schedule")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--dump", action="store_true", help="print the wide tree (small scenes): N<i> = inner node, L<n> = leaf of n triangles, - = empty slot")
    a = ap.parse_args()
    if a.dump:
        from zetaray_amd import scene_io
        from tests.hostexec import zhx
        sc = (scene_io.make_synthetic_scene(num_tris=262144, num_emissive=100000, layout="atrium") if a.scene == "synthetic"
              else scene_io.load_npz(os.path.join(ROOT, "tests", "golden", "cornell_emissive.npz")))
        hx = zhx.HostExecScene(sc)
        _, nn, nt, st = hx.bvh_digest()
        ch = np.zeros((nn, 4), np.uint32)
        zhx.lib().zhx_bvh_children.argtypes = [C.c_void_p, C.c_void_p]
        zhx.lib().zhx_bvh_children(hx.h, ch.ctypes.data)

        def show(i, d):
            print("  " * d + f"node {i}: " + " ".join("-" if c == 0xFFFFFFFF else (f"L{(c & 7) + 1}" if c & 0x80000000 else f"N{c}") for c in ch[i]))
            if d < 6:
                for c in ch[i]:
                    if c != 0xFFFFFFFF and not (c & 0x80000000):
                        show(int(c), d + 1)
        if nn <= 200:
            show(0, 0)
        flat = ch.ravel()
        leaf = (flat != 0xFFFFFFFF) & ((flat & 0x80000000) != 0)
        print(json.dumps({"nodes": nn, "tris": nt, "stack_need": st, "leaves": int(leaf.sum()), "leaf_sizes": np.bincount((flat[leaf] & 7) + 1).tolist(),
                          "children_per_node": round(float((flat != 0xFFFFFFFF).sum()) / max(nn, 1), 3)}))
        return
    if a.sweep:
        variants = [{}, {"ZR_BVH_SWEEP": "1000000000"}, {"ZR_BVH_MAX_LEAF": "4"}, {"ZR_BVH_SAH_LEAF": "1.0"}] + [dict(kv.split("=") for kv in v.split(",")) for v in os.environ.get("ZR_BVH_VARIANTS", "").split(";") if v]
        for v in variants:
            env = {k: val for k, val in os.environ.items() if not k.startswith("ZR_BVH_")}
            env.update(v)
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--scene", a.scene, "--rays", str(a.rays)], env=env, capture_output=True, text=True)
            line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps({"knobs": v, "error": cp.stderr[-300:]}), flush=True)
        return
    print(json.dumps(measure(a)))


if __name__ == "__main__":
    main()
