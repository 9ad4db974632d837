"""Group a rocprofv3 kernel trace by (kernel, grid size): calls, mean duration, share of the total."""
import csv
import glob
import sys
from collections import defaultdict


def main(pattern, top=40):
    path = sorted(glob.glob(pattern))[-1]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[key][0] += 1
        acc[key][1] += d
    total = sum(v[1] for v in acc.values())
    print("total %.3f ms in %s" % (total / 1e3, path))
    for key, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-70s grid %6s %4s %3s lds %6s vgpr %4s/%4s  x%-4d %9.1f us each %5.1f%%" % (*key, n, t / n, 100 * t / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
