"""profiles/<name>.json <- the per-kernel summaries of three separate rocprofv3 --pmc passes of one bench.py command
(scripts/gpu_r02_profiles.sh): FETCH_SIZE, WRITE_SIZE and the SQ set.  Per kernel and launch:

  traffic_bytes   HBM-side bytes: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
                  (MI355X_MICROARCH.md, HBM section): fetch_bytes = 2 x 1024 x FETCH_SIZE, write_bytes = 1024 x WRITE_SIZE
  valu.issue_frac share of the launch in which a SIMD's VALU is issuing: 4 cycles x SQ_ACTIVE_INST_VALU / (1024 SIMDs x launch cycles),
                  launch cycles = SQ_BUSY_CYCLES / 32 (the counter sums the 32 shader engines)
                  (numerator and denominator are different counters: short, VALU-saturated launches come out a few % above 1 and
                  are clamped; issue_frac_raw keeps the quotient)
  valu.lane_util  active lanes per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  valu.ginst_per_s VALU instructions issued per second (all SIMDs); the peak is 1024 SIMDs x 2.4 GHz / 4 = 614.4
  wait_frac       SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of resident-wave time spent waiting on an instruction's operands

  python tools/pmc_profile.py <fetch.csv> <write.csv> <sq.csv> profiles/r02_pmc_<workload>.json
"""
import json
import sys

N_SIMD, N_SE, CLOCK_GHZ = 1024, 32, 2.4


def load(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        # kernel names may contain commas (template arguments): split from the right
        name, ctr, launches, val, dur = line.rsplit(",", 4)
        out.setdefault(name.strip('"').replace("void ", ""), {})[ctr] = (float(val), int(launches), float(dur))
    return out


def main():
    fetch, write, sq, dst = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3]), sys.argv[4]
    res = {}
    for k in sorted(set(fetch) | set(write) | set(sq)):
        if not k.startswith("k_"):
            continue
        f = fetch.get(k, {}).get("FETCH_SIZE", (0.0, 0, 0.0))
        w = write.get(k, {}).get("WRITE_SIZE", (0.0, 0, 0.0))
        r = {"fetch_bytes": 2.0 * 1024.0 * f[0], "write_bytes": 1024.0 * w[0], "launches_sampled": max(f[1], w[1]),
             "avg_us_under_pmc": max(f[2], w[2])}
        r["traffic_bytes"] = r["fetch_bytes"] + r["write_bytes"]
        s = sq.get(k)
        if s and "SQ_ACTIVE_INST_VALU" in s and s["SQ_BUSY_CYCLES"][0] > 0:
            act, busy = s["SQ_ACTIVE_INST_VALU"][0], s["SQ_BUSY_CYCLES"][0]
            cycles = busy / N_SE
            us = s["SQ_ACTIVE_INST_VALU"][2]
            r["valu"] = {"issue_frac": round(min(1.0, 4.0 * act / (N_SIMD * cycles)), 4), "issue_frac_raw": round(4.0 * act / (N_SIMD * cycles), 4),
                         "lane_util": round(s["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * act), 4),
                         "insts_per_launch": s["SQ_INSTS_VALU"][0],
                         "ginst_per_s": round(s["SQ_INSTS_VALU"][0] / (us * 1e-6) / 1e9, 2), "peak_ginst_per_s": N_SIMD * CLOCK_GHZ / 4.0,
                         "waves": s.get("SQ_WAVES", (0,))[0], "avg_us_under_pmc": us}
            if "SQ_WAIT_INST_ANY" in s and s.get("SQ_WAVE_CYCLES", (0,))[0] > 0:
                r["valu"]["wait_frac"] = round(s["SQ_WAIT_INST_ANY"][0] / s["SQ_WAVE_CYCLES"][0], 4)
        res[k] = r
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        va = v.get("valu", {})
        print(f"{k:36s} {v['traffic_bytes'] / 1e6:10.1f} MB/launch  issue {va.get('issue_frac', 0):.3f}  lanes {va.get('lane_util', 0):.3f}  {va.get('avg_us_under_pmc', 0):9.1f} us")


if __name__ == "__main__":
    main()
