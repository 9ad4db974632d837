"""Turn rocprofv3 output merged under gpurun_out/ into the summaries kept in profiles/.

    python tools/summarise_profiles.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir>

Copies the --kernel-trace --stats CSV, and writes profiles/traffic.json: per kernel name,
HBM bytes per launch = (scale x FETCH_SIZE + WRITE_SIZE) x 1024 (both counters are in KiB, collected
in two separate --pmc passes; the scale of FETCH_SIZE is calibrated in the same run on a read of known
size, see main()).
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
    # FETCH_SIZE tallies every memory-side read request at 64 bytes, also the 128-byte ones, so it reports between
    # 1/2 and 1 of the bytes depending on how many of a kernel's requests cover a whole line.  Calibration in the
    # same run, on the same kind of access (one dword per lane): the last decoder of the headline UNet reads its
    # 96 x 64 x 257 x 509 fp32 input exactly once (plus a few per cent of halo rows), so scale = known bytes / counted
    # bytes, clamped to [1, 2], is applied to every kernel's FETCH_SIZE.  (Before the workgroup -> XCD tile order
    # that makes neighbouring tiles meet in one L2, the same kernel counted 3.19 GB for 3.21 GB: scale 1.)
    known = 96.0 * 64 * 257 * 509 * 4
    cal = [v for k, v in fetch.items() if k.startswith("direct_conv_kernel<") and ", 1, 7" in k]
    scale = min(2.0, max(1.0, known / (cal[0] * 1024.0))) if cal and cal[0] > 0 else 1.0
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "kernel" not in k or "at::native" in k:
            continue
        out[k] = dict(fetch_kib_per_launch=fetch.get(k, 0.0), write_kib_per_launch=write.get(k, 0.0), fetch_scale=scale,
                      hbm_bytes_per_launch=(scale * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0,
                      # what the counters cannot exclude: every read request 64 bytes ... every one 128 bytes
                      hbm_bytes_lower=(fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0,
                      hbm_bytes_upper=(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0)
    json.dump(out, open(os.path.join(prof, "traffic.json"), "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join(prof, tag + "_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
