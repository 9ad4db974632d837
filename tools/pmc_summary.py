"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV.

    python tools/pmc_summary.py <counter_collection.csv> [min_launches]

One line per kernel: launches and the mean of every counter over its launches, plus
MFMA utilisation when its counters are present: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's
1024 SIMDs (64 per fp32 32x32x2 instruction), GRBM_GUI_ACTIVE the kernel's cycles summed over the 8 XCDs, so
utilisation = MFMA_BUSY / (1024 x GUI_ACTIVE / 8)."""

import collections
import csv
import json
import sys


def short(name):
    for key in ("conv_dma_kernel", "direct_conv_kernel", "conv_mfma_kernel", "resblock_dma_kernel", "resblock_layer_kernel",
                "lstm_xcd_kernel", "conv_b16_kernel", "wgrad", "bn_"):
        if key in name:
            i = name.index(key)
            j = name.find(">", i)
            return name[i:j + 1] if j > 0 else name[i:i + 40]
    return name[:48]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    least = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = {}
    for k, c in acc.items():
        n = max(len(v) for v in c.values())
        if n < least:
            continue
        row = {name: sum(v) / len(v) for name, v in c.items()}
        row["launches"] = n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("GRBM_GUI_ACTIVE"):
            row["mfma_utilisation"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)
        out[k] = row
    order = sorted(out, key=lambda k: -out[k].get("SQ_BUSY_CYCLES", out[k]["launches"]) * out[k]["launches"])
    print(json.dumps({k: out[k] for k in order}, indent=1))


if __name__ == "__main__":
    main()
