import sys, time, torch
sys.path.insert(0, "/root/repo")
from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.models import DirectPred
from flexynesis_amd.fit import fit, split_indices
from flexynesis_amd import trials
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
dat = {k: torch.randn(2048, 20000, generator=g, device=dev) for k in ("gex", "cnv")}
ann = {"y": dat["gex"][:, :16].sum(1) / 4}
feats = {k: [f"{k}_{i}" for i in range(20000)] for k in dat}
ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(2048)], {})
p = trials.draw_search_space(4, seed=0, epochs=3)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = DirectPred(p[i], ds, ["y"], device_type="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m._bind("cuda")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    tr, va = split_indices(2048, 0.2, i)
    res = fit(m, ds, tr, va, batch_size=int(p[i]["batch_size"]), epochs=3, lr=p[i]["lr"], seed=i)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"trial {i}: ctor {t1-t0:.3f}s bind {t2-t1:.3f}s fit {t3-t2:.3f}s ({res.steps} steps) state copy {t4-t3:.3f}s", flush=True)
