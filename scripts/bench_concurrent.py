"""Aggregate throughput of T independent trials interleaved on ONE GPU (each trial = its own parameters, cohort
replica, hipGraphs and stream): one trial's latency-bound head/backward chain overlaps another's HBM-bound kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[os.environ.get("CFG", "cfg2")]
spec = ArchSpec(cfg["model"], cfg["layers"], 64, 0.25, 16, cfg["variables"], cfg["surv"][0], cfg["surv"][1], True)
EAGER = bool(os.environ.get("EAGER"))
rows = 128 * (3 if cfg["model"] == "MultiTripletNetwork" else 1)
nb = max(int(cfg["n_samples"] * 0.8) // rows, 1)
for T in (1, 2, 3):
    trials = []
    for k in range(T):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1 + k)
            store = ParamStore(spec, dev, materialize_big_grads=False)
            pipe = PipelinedStep(store, 128, cohort=cohort, n_batches=nb, seed=3 + k)
            pipe.idx.copy_(torch.randperm(int(cfg["n_samples"] * 0.8), device=dev)[:nb * rows])
            pipe.prime(); pipe.step(1e-3); pipe.capture(1e-3)
        trials.append((s, pipe))
    torch.cuda.synchronize()
    def rounds(n):
        for _ in range(n):
            for s, pipe in trials:
                with torch.cuda.stream(s):
                    if EAGER:
                        pipe.step(1e-3)
                    else:
                        pipe.replay()
    rounds(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); rounds(40); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"{T} concurrent trial(s): {dt * 1e3:7.3f} ms/round  {T * 128 / dt:9.0f} samples/s aggregate", flush=True)
    del trials
