"""profiles/<name>.json <- the per-kernel summaries of separate rocprofv3 --pmc passes of one bench.py command (scripts/gpu.sh profiles):
FETCH_SIZE, WRITE_SIZE and the SQ passes A (waves / cycles / waits), B (active cycles per instruction type), C (instruction counts),
E (VALU classes).  Per kernel and launch:

  traffic_bytes   HBM-side bytes: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
                  (MI355X_MICROARCH.md, HBM section): fetch_bytes = 2 x 1024 x FETCH_SIZE, write_bytes = 1024 x WRITE_SIZE
  valu.busy_frac  share of a SIMD's cycles its VALU is busy, from the CALIBRATED issue cost of each instruction class (tools/valu_calib.hip,
                  profiles/r03_valu_calib.jsonl, measured at >= 2 waves per SIMD on this MI355X): v_mul / v_add / v_mov / v_and 2.5 cycles per
                  wave64 instruction, v_fma_f32 3.0 (2.3 - 3.7 with the register banks), 32-bit integer 3.5 (v_add_u32 2.7, v_mul_lo_u32 4.4),
                  v_cvt 4.2, transcendental 8.2, everything else (compares, v_cndmask, shifts, min / max, bit-field ops) 4.2.
                  The costs are upper bounds: a dense streaming kernel at high occupancy sustains more (the LDS-tiled a-trous iterations of the denoise
                  pass issue 0.40 VALU instructions per SIMD-cycle at 7.6 waves per SIMD, i.e. 2.5 cycles each where the model prices their mix at
                  3.1); busy_frac is clamped to 1 and the model's figure kept as busy_frac_model_raw.
                  SQ_ACTIVE_INST_VALU is NOT a cycle count: it reads 1 per instruction (2 per transcendental) whatever the instruction costs
                  -- round 2's "4 x SQ_ACTIVE_INST_VALU" therefore priced every instruction at 4 cycles.
  valu.lane_util  active lanes per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  wave            where a resident wave's time goes (SQ_WAVE_CYCLES = 1): issuing (SQ_ACTIVE_INST_ANY, split by type), waiting for memory /
                  barriers (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY); the three are disjoint and sum to ~1
  occupancy       average resident waves per SIMD: SQ_WAVE_CYCLES / (1024 SIMDs x launch quad-cycles)

  python tools/pmc_profile.py <dir> <prefix> <out.json>      reads <dir>/<prefix>_{fetch,write,sqA,sqB,sqC,sqE}.csv (missing passes are skipped)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_SE = 1024, 32
COST = {"SQ_INSTS_VALU_FMA_F32": 3.0, "SQ_INSTS_VALU_MUL_F32": 2.5, "SQ_INSTS_VALU_ADD_F32": 2.5, "SQ_INSTS_VALU_INT32": 3.5,
        "SQ_INSTS_VALU_CVT": 4.2, "SQ_INSTS_VALU_TRANS_F32": 8.2}
COST_OTHER = 4.2


def load(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path).read().splitlines()[1:]:
        # kernel names may contain commas (template arguments): split from the right
        name, ctr, launches, val, dur = line.rsplit(",", 4)
        out.setdefault(name.strip('"').replace("void ", ""), {})[ctr] = (float(val), int(launches), float(dur))
    return out


def main():
    d, prefix, dst = sys.argv[1:4]
    P = {k: load(os.path.join(d, f"{prefix}_{k}.csv")) for k in ("fetch", "write", "sqA", "sqB", "sqC", "sqE")}
    sys.path.insert(0, ROOT)
    import bench
    res = {"_meta": {"source_hash": bench.source_hash(), "command": f"bench.py workload '{prefix}'", "clock_note": "launch cycles = SQ_BUSY_CYCLES / 32 shader engines"}}
    kernels = sorted(set().union(*[set(p) for p in P.values()]))
    for k in kernels:
        if not k.startswith("k_"):
            continue
        f = P["fetch"].get(k, {}).get("FETCH_SIZE", (0.0, 0, 0.0))
        w = P["write"].get(k, {}).get("WRITE_SIZE", (0.0, 0, 0.0))
        r = {"fetch_bytes": 2.0 * 1024.0 * f[0], "write_bytes": 1024.0 * w[0], "launches_sampled": max(f[1], w[1]), "avg_us_under_pmc": max(f[2], w[2])}
        r["traffic_bytes"] = r["fetch_bytes"] + r["write_bytes"]
        a, b, c, e = P["sqA"].get(k), P["sqB"].get(k), P["sqC"].get(k), P["sqE"].get(k)
        if a and a.get("SQ_BUSY_CYCLES", (0,))[0] > 0 and a.get("SQ_WAVE_CYCLES", (0,))[0] > 0:
            cyc = a["SQ_BUSY_CYCLES"][0] / N_SE
            wc = a["SQ_WAVE_CYCLES"][0]
            us = a["SQ_INSTS_VALU"][2]
            insts = a["SQ_INSTS_VALU"][0]
            v = {"insts_per_launch": insts, "ginst_per_s": round(insts / (us * 1e-6) / 1e9, 2),
                 "lane_util": round(a["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * a["SQ_ACTIVE_INST_VALU"][0]), 4),
                 "insts_per_simd_cycle": round(insts / (N_SIMD * cyc), 4), "avg_us_under_pmc": us, "waves": a["SQ_WAVES"][0]}
            if e:
                classified = sum(e[n][0] for n in COST if n in e)
                # pass E ran as its own launch set: scale its class shares to this pass's instruction count
                tot_e = max(classified, 1.0)
                busy_cycles = sum(e[n][0] * COST[n] for n in COST if n in e)
                other = max(0.0, insts - classified)
                busy_cycles += other * COST_OTHER
                raw = busy_cycles / (N_SIMD * cyc)
                v["busy_frac"] = round(min(raw, 1.0), 4)
                if raw > 1.0:      # the calibrated costs are those of dependent streams at 2 - 8 waves; a dense kernel at high occupancy can beat them (see the docstring)
                    v["busy_frac_model_raw"] = round(raw, 4)
                v["avg_cycles_per_inst"] = round(busy_cycles / max(insts, 1.0), 3)
                v["mix"] = {n.replace("SQ_INSTS_VALU_", "").lower(): round(e[n][0] / max(insts, 1.0), 4) for n in COST if n in e}
                v["mix"]["other"] = round(other / max(insts, 1.0), 4)
            r["valu"] = v
            r["occupancy_waves_per_simd"] = round(wc / (N_SIMD * cyc / 4.0), 3)
            wave = {"wait_mem_or_barrier": round(a["SQ_WAIT_ANY"][0] / wc, 4) if "SQ_WAIT_ANY" in a else None,
                    "stalled_at_issue": round(a["SQ_WAIT_INST_ANY"][0] / wc, 4)}
            if b and b.get("SQ_WAVE_CYCLES", (0,))[0] > 0:
                wb = b["SQ_WAVE_CYCLES"][0]
                wave.update({"issuing": round(b["SQ_ACTIVE_INST_ANY"][0] / wb, 4), "issuing_valu": round(a["SQ_ACTIVE_INST_VALU"][0] / wc, 4),
                             "issuing_salu": round(b["SQ_ACTIVE_INST_SCA"][0] / wb, 4), "issuing_lds": round(b["SQ_ACTIVE_INST_LDS"][0] / wb, 4),
                             "issuing_vmem_flat": round((b["SQ_ACTIVE_INST_VMEM"][0] + b["SQ_ACTIVE_INST_FLAT"][0]) / wb, 4),
                             "issuing_branch_misc": round(b["SQ_ACTIVE_INST_MISC"][0] / wb, 4)})
                wave["sum"] = round((wave["issuing"] or 0) + (wave["wait_mem_or_barrier"] or 0) + wave["stalled_at_issue"], 4)
            r["wave_time_shares"] = wave
            if c:
                r["insts_per_launch"] = {n.replace("SQ_INSTS_", "").lower(): c[n][0] for n in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_INSTS_BRANCH") if n in c}
        res[k] = r
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        if k == "_meta":
            continue
        va, ws = v.get("valu", {}), v.get("wave_time_shares", {})
        print(f"{k:40s} {v['traffic_bytes'] / 1e6:9.1f} MB/launch  VALU busy {va.get('busy_frac', 0):.3f}  lanes {va.get('lane_util', 0):.3f}  occ {v.get('occupancy_waves_per_simd', 0):.2f}"
              f"  wave: issue {ws.get('issuing', 0)} wait {ws.get('wait_mem_or_barrier', 0)} stall {ws.get('stalled_at_issue', 0)}  {va.get('avg_us_under_pmc', 0):9.1f} us")


if __name__ == "__main__":
    main()
