"""Step time of the FineTuner configurations at cfg2 shapes (frozen groups cost nothing on the engine)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["cfg2"]
spec = ArchSpec(cfg["model"], cfg["layers"], 64, 0.25, 16, cfg["variables"], None, None, True)
cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1)
for label, kw in (("HPO trainer (clip, nothing frozen)", dict(clip=True, frozen=())),
                  ("fine-tune, nothing frozen (no clip)", dict(clip=False, frozen=())),
                  ("fine-tune, supervisors frozen", dict(clip=False, frozen=("MLPs.",))),
                  ("fine-tune, encoders frozen", dict(clip=False, frozen=("encoders.",)))):
    store = ParamStore(spec, dev, materialize_big_grads=False)
    pipe = PipelinedStep(store, 128, cohort=cohort, n_batches=12, seed=3, **kw)
    pipe.idx.copy_(torch.randperm(1639, device=dev)[:12 * 128])
    pipe.prime(); pipe.step(1e-3); pipe.capture(1e-3)
    for _ in range(5): pipe.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): pipe.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print(f"{label:40s} {dt * 1e3:7.3f} ms/step  {128 / dt:9.0f} samples/s  launches {pipe.n_launches()}", flush=True)
    del pipe, store
