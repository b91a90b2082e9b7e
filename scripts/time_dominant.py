"""Why does the eager HIP-event timing of the dW+Adam kernel differ from its in-graph rocprof duration?
Times the kernel (a) inside the eager step, (b) inside the eager step with a sync before t_opt, (c) isolated."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flexynesis_amd import ops
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, StepPlan

dev = torch.device("cuda:0")
cfg = bench.CONFIGS["cfg2"]
spec = ArchSpec(cfg["model"], cfg["layers"], 64, 0.25, 16, cfg["variables"], None, None, True)
cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1)
store = ParamStore(spec, dev, materialize_big_grads=False)
plan = StepPlan(store, 128, train=True, fused=True, supplied_draws=False, seed=3, cohort=cohort, n_batches=12, epoch_acc=True)
plan.idx.copy_(torch.randperm(1639, device=dev)[:12 * 128])
name = {"fx_linear_dw_adam_bf16x3"}
def loop(sync):
    sink = []
    for i in range(12):
        ops.step_begin(ops.IMMEDIATE, store.ctrl, 1e-3, 12)
        plan.t_gather.run(); plan.t_fwd.run(); plan.t_bwd.run()
        if sync:
            torch.cuda.synchronize()
        plan.t_opt.run_timed(name, sink)
    torch.cuda.synchronize()
    return [round(1e3 * e0.elapsed_time(e1)) for _, e0, e1 in sink]
print("in-step      :", loop(False))
print("sync before  :", loop(True))
sink = []
for i in range(6):
    plan.t_opt.run_timed(name, sink)
torch.cuda.synchronize()
print("t_opt alone  :", [round(1e3 * e0.elapsed_time(e1)) for _, e0, e1 in sink])
