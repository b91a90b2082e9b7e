"""rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv -> markdown table (top N rows).

    python scripts/stats_to_md.py <dir-or-csv> [N]
"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if os.path.isdir(src):
        hits = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))
        if not hits:
            raise SystemExit("no *kernel_stats.csv under " + src)
        src = hits[-1]
    rows = list(csv.DictReader(open(src)))
    print("| kernel | calls | total ms | avg us | % | min us | max us |")
    print("|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        name = r["Name"]
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['Percentage']):.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")


if __name__ == "__main__":
    main()
