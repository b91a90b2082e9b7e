"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV (kernel_trace.csv).

    python scripts/trace_timeline.py <dir-or-csv> [step_index]

Prints, for one optimisation step (delimited by fx_step_begin launches), every kernel's start / end / duration relative
to the step's first launch, and a per-kernel summary over all steps (launches per step, us per step)."""
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:60]


def main():
    src = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if os.path.isdir(src):
        hits = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))
        if not hits:
            raise SystemExit("no *kernel_trace.csv under " + src)
        src = hits[-1]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    idx = [i for i, r in enumerate(rows) if "fx_step_begin" in r[0]]
    if len(idx) < which + 2:
        which = max(len(idx) - 3, 0)
    s, e = idx[which], idx[which + 1]
    t0 = rows[s][1]
    print(f"step {which}: start us | end us | dur us | kernel")
    for n, st, en in rows[s:e]:
        print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f}  {short(n)}")
    print("step span us", (rows[e][1] - t0) / 1e3)
    steps = len(idx) - 1
    agg = {}
    for n, st, en in rows[idx[0]:idx[-1]]:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += en - st
    print(f"\nper step over {steps} steps: launches | us | kernel")
    tot = 0.0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{a[0] / steps:6.2f} {a[1] / steps / 1e3:8.1f}  {k}")
        tot += a[1] / steps / 1e3
    print("sum of kernel time per step (us):", round(tot, 1), " mean step span (us):",
          round((rows[idx[-1]][1] - rows[idx[0]][1]) / steps / 1e3, 1))


if __name__ == "__main__":
    main()
