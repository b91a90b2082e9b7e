"""Mean per-launch value of every counter in a rocprofv3 --pmc run, per kernel.   python scripts/pmc_summary.py <dir> [substr]"""
import csv, glob, os, sys, collections, re
src = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "fx_"
hits = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for h in hits:
    for r in csv.DictReader(open(h)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if want in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} {sum(v)/len(v):16.1f}   (n={len(v)})")
