"""Turn rocprofv3 output merged under gpurun_out/ into the summaries kept in profiles/.

    python tools/summarise_profiles.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir>

Copies the --kernel-trace --stats CSV, and writes profiles/traffic.json: per kernel name,
HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) x 1024 (both counters are in KiB, collected
in two separate --pmc passes).  Calibration on this access pattern: direct_conv_kernel<8,1,7>
reads 96 x 64 x 257 x 509 fp32 = 3.21 GB per launch with one dword per lane and FETCH_SIZE
reports 3.19 GB, so dword-per-lane loads (which is what the conv kernels' input DMA uses) are
counted at face value; the factor-2 under-count documented for 16-byte-per-lane streaming
reads would only touch the (L2-resident) packed-weight stream and is not applied.
"""

import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(pmc_dir):
    f = glob.glob(os.path.join(pmc_dir, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name[name.index("unetmi::") + 8:name.index(">(") + 1] if "unetmi::" in name else name
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(prof, tag + "_kernel_stats.csv"))
    fetch, write = per_kernel(fetch_dir), per_kernel(write_dir)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "kernel" not in k:
            continue
        out[k] = dict(fetch_kib_per_launch=fetch.get(k, 0.0), write_kib_per_launch=write.get(k, 0.0),
                      hbm_bytes_per_launch=(fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0)
    json.dump(out, open(os.path.join(prof, "traffic.json"), "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join(prof, tag + "_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
