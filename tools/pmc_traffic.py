"""profiles/<name>.json <- gpurun_out/pmc_{fetch,write}_summary.csv: HBM-side bytes per launch of every kernel.
FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so
`fetch_bytes` = 2 x 1024 x FETCH_SIZE and `write_bytes` = 1024 x WRITE_SIZE (uncalibrated for narrow / scratch accesses).
  python tools/pmc_traffic.py gpurun_out/pmc_fetch_summary.csv gpurun_out/pmc_write_summary.csv profiles/r01_pmc_traffic_rpt1080p.json"""
import json
import sys


def load(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        # kernel names may contain commas (template arguments): split from the right
        name, _ctr, launches, val, dur = line.rsplit(",", 4)
        out[name.replace("void ", "")] = (float(val), int(launches), float(dur))
    return out


def main():
    fetch, write, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f = fetch.get(k, (0.0, 0, 0.0))
        w = write.get(k, (0.0, 0, 0.0))
        res[k] = {"fetch_bytes": 2.0 * 1024.0 * f[0], "write_bytes": 1024.0 * w[0], "launches_sampled": max(f[1], w[1]),
                  "avg_us_under_pmc": max(f[2], w[2])}
        res[k]["traffic_bytes"] = res[k]["fetch_bytes"] + res[k]["write_bytes"]
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:28s} {v['traffic_bytes'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
