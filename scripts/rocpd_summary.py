"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats table (markdown/CSV)."""
import sqlite3, sys, re
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, (end-start) from kernels").fetchall()
agg = {}
for name, d in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0, 10**18, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print("| kernel | calls | total_ms | avg_us | min_us | max_us | pct |")
print("|---|---|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.2f} |")
