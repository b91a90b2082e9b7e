"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv (separate passes) -> profiles/<name>.json with per-kernel HBM bytes per
launch, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (128-B requests are counted as 64 B)."""
import csv, json, sys, collections
fetch_csv, write_csv, out = sys.argv[1:4]
def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
    return agg
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
kern = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("fx_") and "fx_" not in k:
        continue
    fm = sum(f[k]) / len(f[k]) if f.get(k) else 0.0
    wm = sum(w[k]) / len(w[k]) if w.get(k) else 0.0
    kern[k] = {"FETCH_SIZE_KB_mean": round(fm, 2), "WRITE_SIZE_KB_mean": round(wm, 2), "launches": len(f.get(k, w.get(k, []))),
               "hbm_bytes_per_launch_corrected": int(round((2.0 * fm + wm) * 1024))}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py "
           "--steps 6 --warmup 2 --no-cpu-baseline --no-graph`, 1x MI355X, cfg2. Units: rocprofv3 reports KB. gfx950 correction "
           "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads -> "
           "doubled. WRITE_SIZE is uncalibrated (it matches the algorithmic 12 B/param of the dW+Adam kernel).", "kernels": kern},
          open(out, "w"), indent=1)
for k, d in kern.items():
    print(f"{k[:60]:60s} {d['hbm_bytes_per_launch_corrected']/1e6:10.1f} MB/launch")
