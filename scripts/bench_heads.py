"""Duration of the fused heads kernels in isolation (HIP events around each launch, eager)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore, StepPlan
dev = torch.device("cuda:0")
nh = int(os.environ.get("NH", "1"))
variables = [("y", "numerical", 1), ("c", "categorical", 5), ("z", "numerical", 1)][:nh]
spec = ArchSpec("DirectPred", [("gex", 300), ("cnv", 200)], 64, 0.3, 16, variables, None, None, True)
store = ParamStore(spec, dev, materialize_big_grads=True)
plan = StepPlan(store, 128, train=True, fused=False, seed=7)
g = torch.Generator().manual_seed(0)
plan.set_batch(x_list=[torch.randn(128, 300, generator=g).to(dev), torch.randn(128, 200, generator=g).to(dev)],
               y={"y": torch.randn(128).to(dev), "c": torch.randint(0, 5, (128,)).float().to(dev), "z": torch.randn(128).to(dev)})
for name, tape in (("fx_heads_fwd", plan.t_fwd), ("fx_heads_bwd", plan.t_bwd), ("fx_mse_masked", plan.t_fwd), ("fx_bn_act_fwd", plan.t_fwd)):
    sink = []
    for i in range(12):
        tape.run_timed({name}, sink)
    torch.cuda.synchronize()
    ts = [round(1e3 * e0.elapsed_time(e1), 1) for _, e0, e1 in sink]
    print(name, "n/iter", len(ts) // 12, "us:", ts[-8:])
