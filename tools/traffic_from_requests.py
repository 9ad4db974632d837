"""HBM bytes per launch of every library kernel from two rocprofv3 --pmc passes over the L2's memory-side request
counters (tools/profile_round.sh: <tag>_pmc_rd, <tag>_pmc_wr):

    python tools/traffic_from_requests.py <tag> <pmc_rd_dir> <pmc_wr_dir> [bench.json]

    read bytes  = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B   (the three sizes add up to RDREQ; checked)
    write bytes = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B)

Writes profiles/<tag>_traffic.json and profiles/traffic.json (what bench.py's roofline.traffic reads).  With the
bench line of the same build, the dominant kernel's entry also gets its algorithmic bytes (inputs read once +
output written once, from the layer geometry of its launches) and the ratio measured / algorithmic."""

import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(pmc_dir):
    f = glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "unetmi::" not in name:
            continue
        name = name[name.index("unetmi::") + 8:]
        name = name[:name.index(">(") + 1] if ">(" in name else name[:name.index("(")]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())}
            for k, cs in acc.items()}


def main():
    tag, rd_dir, wr_dir = sys.argv[1:4]
    rd, wr = per_kernel(rd_dir), per_kernel(wr_dir)
    out = {}
    for k in sorted(set(rd) | set(wr)):
        r, w = rd.get(k, {}), wr.get(k, {})
        n32, n64, n128 = (r.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B"))
        total = r.get("TCC_EA0_RDREQ_sum", 0.0)
        rbytes = 32 * n32 + 64 * n64 + 128 * n128
        w64, wall = w.get("TCC_EA0_WRREQ_64B_sum", 0.0), w.get("TCC_EA0_WRREQ_sum", 0.0)
        wbytes = 64 * w64 + 32 * (wall - w64)
        out[k] = dict(read_requests=total, read_requests_32B=n32, read_requests_64B=n64, read_requests_128B=n128,
                      read_sizes_cover=(n32 + n64 + n128) / total if total else None,
                      write_requests=wall, write_requests_64B=w64, read_bytes_per_launch=rbytes,
                      write_bytes_per_launch=wbytes, hbm_bytes_per_launch=rbytes + wbytes,
                      launches_profiled=int(max(r.get("launches", 0), w.get("launches", 0))))
    if len(sys.argv) > 4:
        line = [l for l in open(sys.argv[4]) if l.startswith("{")][-1]
        roof = json.loads(line)["roofline"]
        k = roof["kernel"]
        if k in out and roof.get("algorithmic_bytes_per_launch"):
            out[k]["algorithmic_bytes_per_launch"] = roof["algorithmic_bytes_per_launch"]
            out[k]["measured_over_algorithmic"] = out[k]["hbm_bytes_per_launch"] / roof["algorithmic_bytes_per_launch"]
    prof = os.path.join(ROOT, "profiles")
    for name in (tag + "_traffic.json", "traffic.json"):
        json.dump(out, open(os.path.join(prof, name), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
