"""Screen-tile split of a frame over N devices and the reservoir halo exchange of the ReSTIR passes (SURVEY.md section
8(e)): host-side logic shared by bench.py, the GPU path (RCCL through torch.distributed) and the CPU tests (gloo + the
test-only host executor).

Tiles are 32-px aligned so thread groups, wave-reduction groups and RNG group ids are those of the single-device run; every
device renders the G-buffer of its tile plus a 32-px apron locally (geometry is replicated) and owns the reservoirs of its
tile; apron reservoirs come from the neighbours: post-temporal reservoirs before the spatial stage (the spatial passes
read a neighbour within 15 px) and final reservoirs after it (the next frame's temporal passes read the motion-shifted
pixel).  Point-to-point only: nothing is reduced, so a ring collective would only add per-link latency on xGMI.
"""
import numpy as np

APRON = 32


def tile_grid(n):
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}[n]


def tile_rect(w, h, n, rank, layout=None):
    """(x0, y0, tw, th) of rank's tile: 32-px aligned boundaries.  layout: the n rects of a cost-balanced split (balanced_layout), else the
    equal-area grid."""
    if layout is not None:
        assert len(layout) == n
        return tuple(int(v) for v in layout[rank])
    gx, gy = tile_grid(n)
    tx, ty = rank % gx, rank // gx

    def split(total, parts, i):
        edges = [min(total, ((total * k // parts) + 31) // 32 * 32) for k in range(parts + 1)]
        edges[-1] = total
        return edges[i], edges[i + 1] - edges[i]
    x0, tw = split(w, gx, tx)
    y0, th = split(h, gy, ty)
    return x0, y0, tw, th


def balanced_layout(w, h, n, cost, min_cells=2):
    """n 32-px-aligned rects that tile the w x h frame with (nearly) equal COST: `cost` is a (ceil(h / 32), ceil(w / 32)) array of work per
    32 x 32-px cell (zr_pass_read_cost_map: rays per cell of the previous frames).  Recursive kd-split: a region that has to feed k ranks
    is cut -- along the axis and at the cell boundary where the two sides' cost shares come closest to floor(k / 2) : ceil(k / 2) -- until
    every region feeds one rank.  Deterministic (every rank computes the same layout from the same map); a tile is at least `min_cells`
    cells wide and high so that the 32-px apron of its neighbours never spans a whole tile."""
    cost = np.asarray(cost, np.float64)
    gh, gw = (h + 31) // 32, (w + 31) // 32
    assert cost.shape == (gh, gw), (cost.shape, (gh, gw))
    cost = cost + cost.sum() * 1e-6 / cost.size + 1e-9          # empty regions still split by area
    out = []

    def rec(cx0, cy0, cx1, cy1, k):
        if k == 1:
            x0, y0 = cx0 * 32, cy0 * 32
            out.append((x0, y0, min(w, cx1 * 32) - x0, min(h, cy1 * 32) - y0))
            return
        k1 = k // 2
        want = k1 / k
        region = cost[cy0:cy1, cx0:cx1]
        total = region.sum()
        best = None
        for axis in (0, 1):             # 0: cut in x, 1: cut in y
            prof = np.cumsum(region.sum(axis=0 if axis == 0 else 1))
            length = len(prof)
            lo, hi = min_cells * k1, length - min_cells * (k - k1)         # each side keeps room for its ranks
            if hi < lo:
                continue
            for c in range(max(1, lo), min(length - 1, hi) + 1):
                err = abs(prof[c - 1] / total - want)
                # prefer the cut with the better balance; ties: the cut through the longer axis (squarer tiles, shorter borders)
                key = (round(err, 9), -(length))
                if best is None or key < best[0]:
                    best = (key, axis, c)
        assert best is not None, "region too small to split"
        _, axis, c = best
        if axis == 0:
            rec(cx0, cy0, cx0 + c, cy1, k1); rec(cx0 + c, cy0, cx1, cy1, k - k1)
        else:
            rec(cx0, cy0, cx1, cy0 + c, k1); rec(cx0, cy0 + c, cx1, cy1, k - k1)
    rec(0, 0, gw, gh, n)
    return out


def cost_share(cost, rect):
    x0, y0, tw, th = rect
    return float(np.asarray(cost)[y0 // 32:(y0 + th + 31) // 32, x0 // 32:(x0 + tw + 31) // 32].sum())


def choose_layout(w, h, n, cost, min_gain=1.15, uniform=0.3):
    """The cost-balanced layout if it is predicted to beat the equal-area grid by at least `min_gain` in the costliest tile, else None (= the
    grid).  Measured on the MI355X (profiles/r03_tile_balance.jsonl): on the Cornell box at 16:9, whose sides are empty, the kd-split brings the
    slowest of 8 tiles from 0.83 to 0.72 ms; on the uniformly dense atrium the grid is already within 8 % of perfect balance and unequal tile shapes
    only add apron and tail (3.60 -> 3.89 ms), so the grid stays.  uniform (round 6, tools/tile_balance.py --uniform, profiles/r06u_tile_uniform_term.txt): 0.2 - 0.4 brings
    the slowest of 8 Cornell tiles from 0.47 to 0.41 ms (bound 3.66 x -> 4.2 x with frame overlap), changes nothing for 2 / 4 tiles or the atrium."""
    total = float(np.asarray(cost).sum())
    if total <= 0 or n == 1:
        return None
    if uniform > 0:
        # the cost map holds the wave lifetimes of K11 / K14 / K16; the kernels that cost the same for every pixel whatever it shows (K1, the sorts, the
        # neighbour search) are a uniform term on top: `uniform` x the map's mean per cell
        cost = np.asarray(cost, np.float64)
        cost = cost + uniform * total / cost.size
    grid = max(cost_share(cost, tile_rect(w, h, n, r)) for r in range(n))
    lay = balanced_layout(w, h, n, cost)
    bal = max(cost_share(cost, t) for t in lay)
    return lay if grid > min_gain * bal else None


def extended_rect(w, h, rect, apron=APRON):
    x0, y0, tw, th = rect
    ex0, ey0 = max(0, x0 - apron), max(0, y0 - apron)
    ex1, ey1 = min(w, x0 + tw + apron), min(h, y0 + th + apron)
    return ex0, ey0, ex1 - ex0, ey1 - ey0


def intersect(a, b):
    x0, y0 = max(a[0], b[0]), max(a[1], b[1])
    x1, y1 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    if x1 <= x0 or y1 <= y0:
        return None
    return x0, y0, x1 - x0, y1 - y0


def halo_plan(w, h, n, rank, apron=APRON, layout=None):
    """[(peer, send_rect, recv_rect)]: send = my tile inside the peer's extended rect, recv = the peer's tile inside mine
    (global pixel coordinates).  Symmetric by construction, so both sides agree on sizes without a handshake."""
    mine = tile_rect(w, h, n, rank, layout)
    mine_ext = extended_rect(w, h, mine, apron)
    plan = []
    for peer in range(n):
        if peer == rank:
            continue
        theirs = tile_rect(w, h, n, peer, layout)
        send = intersect(mine, extended_rect(w, h, theirs, apron))
        recv = intersect(theirs, mine_ext)
        if send is not None or recv is not None:
            plan.append((peer, send, recv))
    return plan


def denoise_schedule(iterations, apron=APRON):
    """The denoise pass (ZR_PASS_DENOISE) on a tile + apron: which steps run between which halo exchanges so that every OWNED pixel equals the full
    frame's.  A step computed on the window is exact `reach` pixels inside what its inputs were exact on: the variance stage reads 3 px around a
    pixel, a-trous iteration i reads 2 * 2^i px (zr_svgf.h).  After an exchange the exchanged plane is exact on the whole apron again (its owners
    computed it).  Returns [("exchange", which) | ("steps", mask), ...]; for the default 5 iterations: exchange INPUT (signal + history), temporal +
    variance + a-trous 0..2 (32 - 3 - 2 - 4 - 8 = 15 px left), exchange ITER, a-trous 3 (16 px), exchange ITER, a-trous 4 (32 px)."""
    from . import api
    sched, steps = [("exchange", api.HALO_DENOISE_INPUT)], api.STAGE_DENOISE_TEMPORAL | api.STAGE_DENOISE_VARIANCE
    margin = apron - 3
    assert margin >= 0
    for i in range(iterations):
        need = 2 << i
        if need > apron:
            raise ValueError(f"denoise on tiles: a-trous iteration {i} reaches {need} px, beyond the {apron}-px apron (at most {apron.bit_length() - 1} iterations)")
        if margin < need:
            sched += [("steps", steps), ("exchange", api.HALO_DENOISE_ITER)]
            steps, margin = 0, apron
        steps |= api.stage_denoise_atrous(i)
        margin -= need
    sched.append(("steps", steps))
    return sched


def frame_reads_history_across_tiles(kind, cb, scene_changed, instances_in_motion=False):
    """Can this frame's temporal stage read a previous-frame reservoir that belongs to another tile?  ReSTIR PT reads exactly the reprojected
    pixel (FindTemporal), so with an unmoved camera (same view, same jitter), an unchanged scene AND a G-buffer whose motion vectors are all
    zero every pixel reads its own history and the apron's previous reservoirs are never touched.  The motion vector comes from the actual
    primary hit (zr_stages.h: GBuffer motion): with a thin-lens camera (cb.dof) the hit lies off the pinhole ray the reprojection assumes, and
    instance records that still carry a previous transform different from the current one (no new update_instances call is needed for that)
    move their pixels too -- both mean "yes".  The other passes pick temporal candidates in a neighbourhood: always yes."""
    if kind != "restir_pt" or scene_changed or instances_in_motion:
        return True
    if int(cb["dof"]) != 0:
        return True
    return not (np.array_equal(cb["curr_view"], cb["prev_view"]) and np.array_equal(cb["curr_camera_jitter"], cb["prev_camera_jitter"]))


class TiledRestirPT:
    """One rank of a tile-split renderer on a GPU: G-buffer + PreLighting + the pass with cross-pixel reuse (two stages) with
    the halo exchange in between, through torch.distributed P2P (backend nccl == RCCL over xGMI on ROCm).

    kind = "restir_pt" (default; Indirect, 62 B/px, exchanges post-temporal and final reservoirs), "restir_gi" (Indirect, 40 B/px,
    one stage, final exchange only), "di" (ReSTIR DI emissive, 24 B/px) or "sky_di" (sun + sky ReSTIR DI, 13 B/px): the DI
    passes exchange once, between their temporal and spatial stages."""

    def __init__(self, scene_host, width, height, world, rank, device=0, params=None, dist=None, kind="restir_pt", pass_params=None,
                 transport="torch_p2p", layout=None):
        import torch
        from . import api
        self.api, self.torch, self.dist = api, torch, dist
        self.kind = kind
        # ReSTIR PT with two spatial rounds (IndirectLighting::m_numSpatialPasses = 2): the second round reads the first one's outputs at neighbouring
        # pixels, so it is its own stage behind one more exchange of the set the next stage reads
        self.two_spatial_rounds = kind == "restir_pt" and params is not None and int(params.num_spatial_passes) == 2
        self.W, self.H, self.world, self.rank = width, height, world, rank
        self.layout = layout
        self.tile = tile_rect(width, height, world, rank, layout)
        self.ext = extended_rect(width, height, self.tile) if world > 1 else self.tile
        self.plan = halo_plan(width, height, world, rank, layout=layout) if world > 1 else []
        ex0, ey0, ew, eh = self.ext
        integ = {"restir_pt": api.INTEGRATOR_RESTIR_PT, "restir_gi": api.INTEGRATOR_RESTIR_GI}.get(kind, api.INTEGRATOR_PATH_TRACING)
        self.r = api.Renderer(scene_host, ew, eh, device=device, params=params, integrator=integ, tile_origin=(ex0, ey0))
        if kind == "di":
            self.hp = self.r.enable_direct(pass_params, device=device)
            self.r.skip_indirect = True
        elif kind == "sky_di":
            self.hp = self.r.enable_sky_direct(pass_params, device=device)
            self.r.skip_indirect = True
        else:
            self.hp = self.r.p_indirect          # the pass whose reservoirs cross tile borders
        if world > 1:
            self.hp.set_owned_rect(*self.tile)
        self.device = torch.device("cuda", device)
        self.bpp = self.hp.halo_bytes_per_pixel()
        self.bufs = {}
        for peer, send, recv in self.plan:
            sb = torch.empty(send[2] * send[3] * self.bpp, dtype=torch.uint8, device=self.device) if send else None
            rb = torch.empty(recv[2] * recv[3] * self.bpp, dtype=torch.uint8, device=self.device) if recv else None
            self.bufs[peer] = (sb, rb)
        self.halo_bytes = sum((sb.numel() if sb is not None else 0) for sb, _ in self.bufs.values())
        # transport="rccl_cpp": the C++ HaloExchange of libzetaray_host.so (zr_halo.cpp) -- one pack kernel, grouped ncclSend / ncclRecv issued
        # from C++ on the pass's stream, one unpack kernel, no host wait -- instead of per-plane copies + torch.distributed P2P ops
        self.transport = transport
        self.native = None
        self._frames_rendered = 0
        self._scene_version_seen = self.r.scene.version
        self.exchanges_done = 0      # (statistics: how many halo exchanges this object has run)
        if transport == "rccl_cpp" and world > 1:
            # every rank must take the same branch (communicator creation is collective): agree on success before using it
            try:
                native, ok = NativeHalo(self.hp, self.r.gbuffer, device, world, rank, self.plan, dist), 1
            except Exception as e:      # e.g. no librccl.so next to this build: the torch.distributed P2P path still works
                import sys
                print(f"[zetaray_amd.tiling] rank {rank}: C++ RCCL halo exchange unavailable ({e}); using torch.distributed P2P", file=sys.stderr)
                native, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                # first contact: ONE exchange of the planes as they stand (idempotent; nothing has rendered yet) before anything depends on the transport --
                # an error code from RCCL at run time (a peer-access or IPC refusal only a real multi-GPU node can produce) sends EVERY rank to the
                # torch.distributed path instead of ending the run in the first timed frame
                try:
                    post, _final = self.EXCHANGES[self.kind]
                    native.run(self.api.HALO_POST_TEMPORAL if post else self.api.HALO_FINAL)
                    torch.cuda.synchronize(self.device)
                    ok = 1
                except Exception as e:
                    import sys
                    print(f"[zetaray_amd.tiling] rank {rank}: the C++ RCCL halo exchange failed its first exchange ({e}); using torch.distributed P2P", file=sys.stderr)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.native = native
                self.halo_bytes = self.native.send_bytes
            else:
                self.transport = "torch_p2p"
                if native is not None:
                    try:
                        native.close()
                    except Exception:
                        pass

    def _xfer(self, which):
        """(pass, bytes per pixel) of an exchange: the reservoir exchanges belong to the halo pass, the ZR_HALO_DENOISE_* ones to the denoise pass"""
        api = self.api
        if which == api.HALO_DENOISE_INPUT:
            return self.p_denoise, 40
        if which == api.HALO_DENOISE_ITER:
            return self.p_denoise, 16
        return self.hp, self.bpp

    def pack(self, which):
        """stage 1 of an exchange: copy my border strips into the per-peer send buffers (device-to-device, on the stream)"""
        hp, bpp = self._xfer(which)
        for peer, send, recv in self.plan:
            if send:
                sb = self.bufs[peer][0]
                hp.halo_pack(self.r.gbuffer, which, send, sb.data_ptr(), send[2] * send[3] * bpp)

    def unpack(self, which):
        """stage 3: scatter the received strips into my apron"""
        hp, bpp = self._xfer(which)
        for peer, send, recv in self.plan:
            if recv:
                rb = self.bufs[peer][1]
                hp.halo_unpack(self.r.gbuffer, which, recv, rb.data_ptr(), recv[2] * recv[3] * bpp)

    def enable_denoise(self, params=None):
        """add the denoise pass on this tile: planes = tile + apron, steps and halo exchanges by denoise_schedule (every owned pixel == the full
        frame's); read the result with denoised_tile()"""
        from . import wire
        prm = params if params is not None else wire.default_params()
        self.p_denoise = self.r.enable_denoise(prm, device=self.device.index)      # the tile's device, like enable_direct / enable_sky_direct above
        self.r.p_denoise = None          # (Renderer.render_frame must not run it in one go: render_frame below drives the schedule)
        self._dn_sched = denoise_schedule(int(prm.svgf_iterations)) if self.world > 1 else [("steps", self.api.STAGE_DENOISE_MASK)]
        if self.native is not None:
            self.native_dn = {self.api.HALO_DENOISE_INPUT: NativeHalo(self.p_denoise, self.r.gbuffer, self.device.index, self.world, self.rank, self.plan, comm_of=self.native, bpp=40),
                              self.api.HALO_DENOISE_ITER: NativeHalo(self.p_denoise, self.r.gbuffer, self.device.index, self.world, self.rank, self.plan, comm_of=self.native, bpp=16)}
        return self.p_denoise

    def denoise(self, cb, exchange=None):
        """the denoise pass of this frame on the tile (after stage_spatial); exchange(which): how the halos move (default: this object's transport)"""
        api = self.api
        self.p_denoise.set_input(api.IN_DENOISE_SIGNAL, self.r.p_indirect.output_ptr()[0])
        for kind, v in self._dn_sched:
            if kind == "exchange":
                (exchange or self.exchange)(v)
            elif v:
                self.p_denoise.render_stage(cb, self.r.scene, self.r.gbuffer, v)

    def denoise_steps(self, cb, steps):
        """one group of steps of the schedule (tile objects sharing a process: tiling.render_frame_in_process)"""
        self.p_denoise.set_input(self.api.IN_DENOISE_SIGNAL, self.r.p_indirect.output_ptr()[0])
        if steps:
            self.p_denoise.render_stage(cb, self.r.scene, self.r.gbuffer, steps)

    def denoised_tile(self):
        full = self.p_denoise.download_plane("denoised")
        x0, y0, tw, th = self.tile
        ex0, ey0 = self.ext[0], self.ext[1]
        return self.tile, full[y0 - ey0:y0 - ey0 + th, x0 - ex0:x0 - ex0 + tw].copy()

    def exchange(self, which):
        if not self.plan:
            return
        self.exchanges_done += 1
        if self.native is not None:
            (self.native_dn[which] if which in getattr(self, "native_dn", {}) else self.native).run(which)
            return
        dist = self.dist
        hp, bpp = self._xfer(which)
        self.pack(which)
        if dist.get_backend() == "gloo":
            # test rig (bench.py with ZR_BENCH_SHARED_GPU=1: several ranks on one device, no RCCL between them): strips staged through the host
            self.torch.cuda.synchronize()
            reqs, landed = [], []
            for peer, send, recv in self.plan:
                sb, rb = self.bufs[peer]
                if send:
                    reqs.append(dist.isend(sb[:send[2] * send[3] * bpp].cpu(), peer))
                if recv:
                    hb = self.torch.empty(recv[2] * recv[3] * bpp, dtype=self.torch.uint8)
                    reqs.append(dist.irecv(hb, peer))
                    landed.append((rb, hb))
            for req in reqs:
                req.wait()
            for rb, hb in landed:
                rb[:hb.numel()].copy_(hb)
            self.unpack(which)
            return
        ops = []
        for peer, send, recv in self.plan:
            sb, rb = self.bufs[peer]
            if send:
                ops.append(dist.P2POp(dist.isend, sb[:send[2] * send[3] * bpp], peer))
            if recv:
                ops.append(dist.P2POp(dist.irecv, rb[:recv[2] * recv[3] * bpp], peer))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        self.unpack(which)

    def stage_temporal(self, cb, stream=None):
        """stream: a hipStream_t (integer handle) -- several tile objects of one device may run on streams of their own (tools/tile_balance.py --halves)"""
        api, r = self.api, self.r
        if r._overlap_stream is not None:
            # frame overlap (api.Renderer.enable_frame_overlap): G-buffer, PreLighting and K11 on the pass's own stream -- beside the previous frame's halo
            # exchange and spatial stage, which are still in `stream`'s queue -- and the temporal reuse behind them on `stream`
            assert self.kind == "restir_pt" and r.p_direct is None and r.p_sky_direct is None
            r._first_half(cb, stream, bool(len(r.scene_host.emissives) and (not r._alias_ready or r._presampling)))
            r.p_indirect.render_stage(cb, r.scene, r.gbuffer, api.STAGE_TEMPORAL_REUSE, stream)
            return
        r.render_sky(cb, stream)
        r.p_gbuffer.render(cb, r.scene, r.gbuffer, stream)
        if len(r.scene_host.emissives) and (not r._alias_ready or r._presampling):
            r.p_prelight.render(cb, r.scene, None, stream)
            r._alias_ready = True
        if self.kind in ("di", "sky_di"):
            self.hp.render_stage(cb, r.scene, r.gbuffer, api.STAGE_TEMPORAL, stream)
            return
        if r.p_direct is not None:
            r.p_direct.render(cb, r.scene, r.gbuffer, stream)
        if r.p_sky_direct is not None:
            r.p_sky_direct.render(cb, r.scene, r.gbuffer, stream)
        r.p_indirect.render_stage(cb, r.scene, r.gbuffer, api.STAGE_TEMPORAL, stream)

    def enable_frame_overlap(self, on=True, carry=False):
        """software-pipeline consecutive frames of this tile on two streams (api.Renderer.enable_frame_overlap); the stages and exchanges keep their order"""
        assert self.kind == "restir_pt"
        self.r.enable_frame_overlap(on, carry)

    def stage_spatial(self, cb, stream=None):
        self.hp.render_stage(cb, self.r.scene, self.r.gbuffer, self.api.STAGE_SPATIAL, stream)

    def stage_spatial2(self, cb):
        self.hp.render_stage(cb, self.r.scene, self.r.gbuffer, self.api.STAGE_SPATIAL2)

    # which exchanges a frame of this kind needs: (post-temporal, final)
    EXCHANGES = {"restir_pt": (True, True), "restir_gi": (False, True), "di": (True, False), "sky_di": (True, False)}

    def history_crosses_tiles(self, cb):
        return frame_reads_history_across_tiles(self.kind, cb, self.r.scene.version != self._scene_version_seen, self.r.scene.instances_in_motion)

    def render_frame(self, cb, exchange_final=True):
        """One frame of this rank's tile.  The FINAL halo (the reservoirs the temporal stage reads as "previous" in the apron) is exchanged at the
        START of the frame that needs it rather than at the end of the frame that produced it -- the planes are the same (nothing renders in
        between), and a frame whose reprojection cannot leave its tile (history_crosses_tiles) skips the exchange altogether: one exchange per
        frame instead of two while nothing moves.  exchange_final=False never exchanges it (the caller vouches for a static view)."""
        post, final = self.EXCHANGES[self.kind]
        if final and exchange_final and self._frames_rendered > 0 and self.history_crosses_tiles(cb):
            self.exchange(self.api.HALO_FINAL)
        self.stage_temporal(cb)
        if post:
            self.exchange(self.api.HALO_POST_TEMPORAL)
        self.stage_spatial(cb)
        if self.two_spatial_rounds:
            self.exchange(self.api.HALO_POST_TEMPORAL)
            self.stage_spatial2(cb)
        if getattr(self, "p_denoise", None) is not None:
            self.denoise(cb)
        self._frames_rendered += 1
        self._scene_version_seen = self.r.scene.version

    def owned_cost_cells(self):
        """this rank's contribution to the frame's cost map: (ceil(H / 32), ceil(W / 32)) float64, rays per cell of the OWNED tile since the
        last call (zero elsewhere); summing the ranks' arrays gives the whole frame's map"""
        gh, gw = (self.H + 31) // 32, (self.W + 31) // 32
        out = np.zeros((gh, gw), np.float64)
        cm = self.r.p_indirect.read_cost_map(reset=True)
        ex0, ey0 = self.ext[0] // 32, self.ext[1] // 32
        x0, y0, tw, th = self.tile
        c0x, c0y, c1x, c1y = x0 // 32, y0 // 32, (x0 + tw + 31) // 32, (y0 + th + 31) // 32
        out[c0y:c1y, c0x:c1x] = cm[c0y - ey0:c1y - ey0, c0x - ex0:c1x - ex0]
        return out

    def final_tile(self):
        """(tile rect, RGBA32F array of the owned tile)"""
        full = self.hp.download()
        x0, y0, tw, th = self.tile
        ex0, ey0 = self.ext[0], self.ext[1]
        return self.tile, full[y0 - ey0:y0 - ey0 + th, x0 - ex0:x0 - ex0 + tw].copy()


class NativeHalo:
    """ctypes face of the C++ halo exchange (zetaray_amd/host/zr_halo.cpp).  The RCCL unique id is created on rank 0 and handed to the other
    ranks through the process group that launched them (a 128-byte broadcast); world == 1 talks to itself (transport self-test)."""

    def __init__(self, halo_pass, gbuffer, device, world, rank, plan, dist=None, unique_id=None, comm_of=None, bpp=0):
        """comm_of: another NativeHalo whose communicator this object shares (creating one is collective); bpp: bytes per pixel of this object's
        exchanges when they differ from the pass's own figure (the denoise pass's two exchanges)"""
        import ctypes as C
        import os
        self.C = C
        L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzetaray_host.so"))
        L.zrh_halo_last_error.restype = C.c_char_p
        L.zrh_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.zrh_comm_destroy.argtypes = [C.c_void_p]
        L.zrh_halo_exchange_create_bpp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.zrh_halo_exchange_destroy.argtypes = [C.c_void_p]
        L.zrh_halo_exchange_send_bytes.restype = C.c_size_t
        L.zrh_halo_exchange_send_bytes.argtypes = [C.c_void_p]
        L.zrh_halo_exchange_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L = L
        self.owns_comm = comm_of is None
        idbuf = (C.c_uint8 * 128)()
        if comm_of is not None:
            pass
        elif unique_id is not None:
            C.memmove(idbuf, bytes(unique_id), 128)
        else:
            ok = 1
            if rank == 0:
                ok = 1 if L.zrh_rccl_unique_id(idbuf) == 0 else 0
            if world > 1:
                # rank 0 always broadcasts (status byte + id), so that a failure there cannot leave the others waiting
                import torch
                t = torch.tensor([ok] + list(bytes(idbuf)), dtype=torch.uint8, device=torch.device("cuda", device))
                dist.broadcast(t, src=0)
                raw = bytes(t.cpu().numpy().tobytes())
                ok = raw[0]
                C.memmove(idbuf, raw[1:], 128)
            if not ok:
                raise RuntimeError("halo exchange: " + (L.zrh_halo_last_error().decode() if rank == 0 else "rank 0 could not create the RCCL unique id"))
        if comm_of is not None:
            self.comm = comm_of.comm
        else:
            self.comm = C.c_void_p()
            self._check(L.zrh_comm_create(device, world, rank, idbuf, C.byref(self.comm)))

        class Peer(C.Structure):
            _fields_ = [("peer", C.c_int)] + [(n, C.c_uint32) for n in ("send_x0", "send_y0", "send_w", "send_h", "recv_x0", "recv_y0", "recv_w", "recv_h")]
        arr = (Peer * max(1, len(plan)))()
        for i, (peer, send, recv) in enumerate(plan):
            s, r = send or (0, 0, 0, 0), recv or (0, 0, 0, 0)
            arr[i] = Peer(peer, s[0], s[1], s[2], s[3], r[0], r[1], r[2], r[3])
        self.x = C.c_void_p()
        self._check(L.zrh_halo_exchange_create_bpp(halo_pass.h, gbuffer.h, self.comm, arr, len(plan), int(bpp), C.byref(self.x)))
        self.send_bytes = int(L.zrh_halo_exchange_send_bytes(self.x))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("halo exchange: " + self.L.zrh_halo_last_error().decode())

    def run(self, which, stream=None):
        self._check(self.L.zrh_halo_exchange_run(self.x, stream, which))

    def close(self):
        if self.x:
            self.L.zrh_halo_exchange_destroy(self.x)
            self.x = self.C.c_void_p()
        if self.comm and self.owns_comm:
            self.L.zrh_comm_destroy(self.comm)
        self.comm = self.C.c_void_p()


def render_frame_in_process(ranks, cb, exchange_final=True):
    """TiledRestirPT.render_frame for tile objects that live in ONE process (the single-GPU tests): the same exchange policy, the halos moved by
    exchange_in_process.  Returns the number of exchanges the frame took."""
    api = ranks[0].api
    post, final = ranks[0].EXCHANGES[ranks[0].kind]
    n = 0
    if final and exchange_final and ranks[0]._frames_rendered > 0 and ranks[0].history_crosses_tiles(cb):
        exchange_in_process(ranks, api.HALO_FINAL); n += 1
    for r in ranks:
        r.stage_temporal(cb)
    if post:
        exchange_in_process(ranks, api.HALO_POST_TEMPORAL); n += 1
    for r in ranks:
        r.stage_spatial(cb)
    if ranks[0].two_spatial_rounds:
        exchange_in_process(ranks, api.HALO_POST_TEMPORAL); n += 1
        for r in ranks:
            r.stage_spatial2(cb)
    if getattr(ranks[0], "p_denoise", None) is not None:
        # every tile runs the same schedule: the steps between two exchanges on every tile, then the exchange among them
        for k, (kind, v) in enumerate(ranks[0]._dn_sched):
            if kind == "exchange":
                if len(ranks) > 1:
                    exchange_in_process(ranks, v); n += 1
            else:
                for r in ranks:
                    r.denoise_steps(cb, v)
    for r in ranks:
        r._frames_rendered += 1
        r._scene_version_seen = r.r.scene.version
    return n


def exchange_in_process(ranks, which):
    """Halo exchange between TiledRestirPT objects living in ONE process (all tiles on one device): used by the GPU test
    on a single-GPU box; the data path (pack -> buffer -> unpack) is the one RCCL sees."""
    for r in ranks:
        r.pack(which)
    for r in ranks:
        for peer, send, recv in r.plan:
            if recv:
                r.bufs[peer][1].copy_(ranks[peer].bufs[r.rank][0])      # (whole buffers: the exchange's bytes are their head)
    for r in ranks:
        r.unpack(which)
