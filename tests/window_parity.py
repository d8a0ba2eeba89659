"""At-size parity without a full CPU frame: a full-resolution frame rendered by `full` (the GPU through the C-ABI in
tests/test_gpu_parity.py; a full-frame host executor in the CPU dry run of tests/test_rpt_cpu.py) is compared on scattered owned windows with the
product's stage functions executed on the host (tests/hostexec -- itself pinned to the oracle and to the reference's shaders at small sizes:
tests/test_cpu_parity.py, tests/test_ref_passes.py).

Each window is rendered the way a device of the tile split renders its tile (zetaray_amd/tiling.py): G-buffer, K11 and the temporal stage on
window + 32-px apron by the host executor; the apron's reservoirs -- post-temporal before the spatial stage, final after the frame -- come from the
full frame exactly as a neighbouring device would send them (zr_pass_halo_pack), so every cross-pixel read of an owned pixel sees what the full
frame sees.  Compared per window and frame, tolerance 0: the post-temporal planes, radiance, the seven final planes (all four bytes of plane A) and,
when the full frame reports them, the pass's BVH queries for the window's pixels.  Follows IndirectLighting.cpp:877-1004.  Test infrastructure.

Round 5: with `oracle=` (an oracle.zro.OracleScene) every window is ALSO rendered by the oracle itself -- oracle/zro_rpt.h on a
full-size frame with its loops restricted to the window (zro.OracleRPTWindows), G-buffer included -- driven exactly like the host executor, and the
full frame is compared with it too: at the quoted size GPU == host-executed stage functions == oracle, not GPU == its own code on the host."""
import numpy as np

HALO_PLANES = (("A", np.uint32, 1, 4), ("B", np.float32, 2, 8), ("C", np.uint32, 4, 16), ("D", np.uint32, 4, 16), ("E", np.uint16, 1, 2),
               ("F", np.float32, 2, 8), ("G", np.uint32, 2, 8))       # the block order of zr_pass_halo_pack (zr_api.hip HaloPlanes)


def apron_rects(ext, own):
    """ext minus own as up to four rects in ext-local coordinates: (x, y, w, h)"""
    ex, ey, ew, eh = ext
    ox, oy, ow, oh = own[0] - ex, own[1] - ey, own[2], own[3]
    rects = [(0, 0, ew, oy), (0, oy + oh, ew, eh - oy - oh), (0, oy, ox, oh), (ox + ow, oy, ew - ox - ow, oh)]
    return [r for r in rects if r[2] > 0 and r[3] > 0]


def windows_parity(full, sc, alias, W, H, windows, prm, cams, oracle=None):
    """full: .stage1(cb), .rect_planes(which, rect) -> {plane: array}, .stage2(cb) -> (radiance HxWx4, ray cells or None).  cams: one camera
    position per frame.  oracle: an OracleScene -> the windows are also rendered by the oracle and compared.  Returns the number of rays compared."""
    from tests.hostexec import zhx
    from zetaray_amd import scene_io, tiling
    hx = zhx.HostExecScene(sc, alias)
    wins = []
    for own in windows:
        assert own[0] % 32 == 0 and own[1] % 32 == 0 and ((own[0] + own[2]) % 32 == 0 or own[0] + own[2] == W) and ((own[1] + own[3]) % 32 == 0 or own[1] + own[3] == H), own
        ext = tiling.extended_rect(W, H, own)
        wins.append((own, ext, zhx.HostExecRPT(hx, ext[2], ext[3], ext=ext, owned=own)))
    if oracle is not None:
        from oracle import zro
        ow = zro.OracleRPTWindows(oracle, W, H, [(own, ext) for own, ext, _ in wins])
        # every (window, checker) pair below: the host executor first, then the oracle's view of the same window
        wins = [(own, ext, hr, ow.window(i)) for i, (own, ext, hr) in enumerate(wins)]
    else:
        wins = [(own, ext, hr, None) for own, ext, hr in wins]
    prev, checked_rays = None, 0
    for f, cam in enumerate(cams, 1):
        cb = scene_io.make_frame_constants(W, H, frame_num=f, num_emissives=len(sc.emissives), cam_pos=cam)
        if prev is not None:
            cb["prev_view"], cb["prev_view_inv"], cb["prev_camera_jitter"] = prev["curr_view"], prev["curr_view_inv"], prev["curr_camera_jitter"]
        prev = cb.copy()
        if prm.presampling:
            hx.presample(f, int(prm.num_sample_sets), int(prm.sample_set_size))
            if oracle is not None:
                oracle.presample(f, int(prm.num_sample_sets), int(prm.sample_set_size))
        full.stage1(cb)
        for own, ext, hr, orc in wins:
            post = full.rect_planes(1, ext)
            lx, ly = own[0] - ext[0], own[1] - ext[1]
            for who, chk in (("host executor", hr), ("oracle", orc)):
                if chk is None:
                    continue
                chk.render_stage(cb, prm, 1)
                for name, _, _, _ in HALO_PLANES:
                    mine = chk.plane(name, 1)[ly:ly + own[3], lx:lx + own[2]]
                    assert np.array_equal(mine.view(np.uint8), post[name][ly:ly + own[3], lx:lx + own[2]].view(np.uint8)), f"frame {f} window {own}: post-temporal plane {name} ({who})"
                    for rect in apron_rects(ext, own):
                        chk.write_plane_rect(name, 1, post[name], rect)
        got, cells = full.stage2(cb)
        for own, ext, hr, orc in wins:
            lx, ly = own[0] - ext[0], own[1] - ext[1]
            a = got[own[1]:own[1] + own[3], own[0]:own[0] + own[2]]
            fin = full.rect_planes(0, ext)
            cy0, cy1, cx0, cx1 = own[1] // 32, (own[1] + own[3] + 31) // 32, own[0] // 32, (own[0] + own[2] + 31) // 32
            for who, chk in (("host executor", hr), ("oracle", orc)):
                if chk is None:
                    continue
                want = chk.render_stage(cb, prm, 2)
                b = want[ly:ly + own[3], lx:lx + own[2]]
                mism = int((np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)).any(axis=2).sum())
                assert mism == 0, f"frame {f} window {own}: {mism} pixels of the radiance differ ({who})"
                for name, _, _, _ in HALO_PLANES:
                    mine = chk.plane(name, 0)[ly:ly + own[3], lx:lx + own[2]]
                    assert np.array_equal(mine.view(np.uint8), fin[name][ly:ly + own[3], lx:lx + own[2]].view(np.uint8)), f"frame {f} window {own}: final plane {name} ({who})"
                    for rect in apron_rects(ext, own):
                        chk.write_plane_rect(name, 0, fin[name], rect)
                if cells is not None:
                    # the window is a whole number of 32 x 32 cost-map cells (partial ones only at the frame's edge)
                    full_rays = int(cells[cy0:cy1, cx0:cx1].sum())
                    assert full_rays == sum(chk.counters), f"frame {f} window {own}: {full_rays} rays in the full frame, {sum(chk.counters)} by the {who}"
            checked_rays += int(cells[cy0:cy1, cx0:cx1].sum()) if cells is not None else sum(hr.counters)
        assert got[..., :3].max() > 0
    return checked_rays
