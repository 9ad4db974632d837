"""The dispatches of the LAST step in a rocprofv3 kernel trace, in launch order: kernel, grid, duration, gap.

    python tools/trace_last_step.py <kernel_trace.csv glob> <dispatches per step | 0 = guess from the repeat>"""
import csv
import glob
import sys


def short(n):
    n = n.replace("void ", "").replace("unetmi::", "")
    return n[:n.index("(")][:64] if "(" in n else n[:64]


def main(pattern, per_step=0):
    path = sorted(glob.glob(pattern, recursive=True))[-1]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    if per_step == 0:                       # smallest period of the tail of the name sequence
        for p in range(10, len(names) // 2):
            if names[-p:] == names[-2 * p:-p]:
                per_step = p
                break
    rows, names = rows[-per_step:], names[-per_step:]
    t_prev, total = None, 0.0
    for r, n in zip(rows, names):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if t_prev is None else (s - t_prev) / 1e3
        t_prev = e
        total += (e - s) / 1e3
        print("%-64s grid %8s  %9.1f us  gap %6.1f" % (n, r["Grid_Size_X"], (e - s) / 1e3, gap))
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print("%d dispatches, kernel time %.1f us, span %.1f us" % (per_step, total, span))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
