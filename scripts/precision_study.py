"""Per-step (re-synchronised) and free-running loss error of the engine vs the fp32 CPU oracle for
precision in {f32, bf16x3} at a cfg2-family shape.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore, StepPlan
from oracle import restate as O
dev = torch.device("cuda:0")
F = [int(a) for a in (sys.argv[1:3] or [8000, 6000])]
layers=[("gex",F[0]),("cnv",F[1])]; B=128
variables=[("y","numerical",1),("c","categorical",4)]
aspec=ArchSpec("DirectPred",layers,64,0.25,16,variables,None,None,True)
ospec=O.Spec("DirectPred",layers,64,0.25,16,variables,None,None,True)
dat, ann = O.synthetic_cohort(layers, 512, seed=1234)
st0 = O.init_state(ospec, seed=3)
for resync in (True, False):
  for prec in ("f32","bf16x3"):
    gen = torch.Generator().manual_seed(99)
    store = ParamStore(aspec, dev); store.load_state(st0)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True, precision=prec)
    st, opt = st0, {}
    errs=[]
    for step in range(5):
        idx = torch.randperm(512, generator=gen)[:B]
        y = {k: ann[k][idx] for k in plan.y}
        draws = {n: (torch.rand(t.shape, generator=gen) < 0.9).float() for n,t in plan.draws.items()}
        xs=[dat[n][idx] for n,_ in layers]
        if resync and step>0:
            store.load_state(st); store.reset_optimizer(); store.load_optimizer(opt["t"], opt["m"], opt["v"])
        plan.set_batch(x_list=[x.to(dev) for x in xs], y={k:v.to(dev) for k,v in y.items()})
        plan.set_draws({k:v.to(dev) for k,v in draws.items()})
        plan.train_step(1e-3)
        st, opt, info = O.train_step(ospec, st, opt, {"x":xs,"y":y}, draws, 1e-3)
        got=plan.losses()
        errs.append(max(abs(got[k]-float(v.reshape(-1)[0]))/abs(float(v.reshape(-1)[0])) for k,v in info["losses"].items()))
    print(f"F={F} resync={resync} precision={prec}: max rel loss err per step:", " ".join(f"{e:.2e}" for e in errs), flush=True)
