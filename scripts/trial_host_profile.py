"""Where the HOST spends one HPO trial of the cfg5 sweep (eager launches, as on a worker thread): cProfile of run_trial, its wall time against the
time its wide weights need to stream.   python scripts/trial_host_profile.py [trial index]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from flexynesis_amd import trials
from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.fit import run_trial
from flexynesis_amd.models import DirectPred
from flexynesis_amd.sweep import _cohort
dev = torch.device("cuda:0")
tid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
layers = [("gex", 20000), ("cnv", 20000)]
dat, ann = _cohort(layers, 2048, dev, 1234)
feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(2048)], {})
plist = trials.draw_search_space(8, seed=0, epochs=3)
print(plist[tid])
for rep in range(2):            # first: warm-up (arena, caches)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    val, ep, model, info = run_trial(DirectPred, plist[tid], ds, ["y"], early_stop_patience=0, seed=tid, device=dev, use_graph=False)
    torch.cuda.synchronize(); print(f"trial wall {time.perf_counter() - t0:.3f} s, steps {info.get('steps')}, val {val:.4f}", {k: v for k, v in info.items() if k.endswith('_s')})
    del model
pr = cProfile.Profile(); pr.enable()
val, ep, model, info = run_trial(DirectPred, plist[tid], ds, ["y"], early_stop_patience=0, seed=tid, device=dev, use_graph=False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
