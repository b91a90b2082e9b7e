"""Two trials in flight on one GPU (continued): T worker threads, each with its own HIP stream, run engine
fits concurrently -- eager launches, so that one trial's latency-bound chain slips under the other's HBM-bound dW + Adam
launches -- against the same fits one after another with hipGraph replay.   python scripts/bench_two_trials.py [epochs]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.fit import fit, split_indices
from flexynesis_amd.models import DirectPred

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
dat = {k: torch.randn(2048, 20000, generator=g, device=dev) for k in ("gex", "cnv")}
ann = {"y": dat["gex"][:, :16].sum(1) / 4}
feats = {k: [f"{k}_{i}" for i in range(20000)] for k in dat}
ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(2048)], {})
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tr, va = split_indices(2048, 0.2, 0)
CFGS = [{"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "batch_size": b, "epochs": epochs}
        for b in (128, 64, 128, 32, 64, 128, 32, 64)]
lock = threading.Lock()


def one(i, use_graph):
    cfg = CFGS[i]
    with lock:
        torch.manual_seed(i)
        m = DirectPred(cfg, ds, ["y"], device_type="cuda")
    res = fit(m, ds, tr, va, batch_size=cfg["batch_size"], epochs=epochs, lr=1e-3, seed=i, use_graph=use_graph)
    return res.steps * cfg["batch_size"], res.val_loss


def sequential(use_graph):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = [one(i, use_graph) for i in range(len(CFGS))]
    torch.cuda.synchronize()
    return sum(s for s, _ in out) / (time.perf_counter() - t0), [v for _, v in out]


def threaded(T):
    todo, out = list(range(len(CFGS))), {}

    def work():
        with torch.cuda.stream(torch.cuda.Stream()):
            while True:
                with lock:
                    if not todo:
                        return
                    i = todo.pop(0)
                try:
                    out[i] = one(i, False)
                except Exception:
                    import traceback
                    traceback.print_exc()
                    out[i] = (0, float("nan"))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=work) for _ in range(T)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return sum(s for s, _ in out.values()) / (time.perf_counter() - t0), [out[i][1] for i in range(len(CFGS))]


one(0, True)                      # warm the process
for rep in range(2):
    sg, vg = sequential(True)
    se, ve = sequential(False)
    t2, v2 = threaded(2)
    t3, v3 = threaded(3)
    print(f"8 trials x {epochs} epochs (B mix): sequential graph {sg:9.0f} samples/s | sequential eager {se:9.0f} | 2 threads eager "
          f"{t2:9.0f} ({t2 / sg:.3f} x graph) | 3 threads {t3:9.0f} ({t3 / sg:.3f} x)", flush=True)
    print("   val losses agree (threads vs sequential eager):", all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(v2, ve)), flush=True)
