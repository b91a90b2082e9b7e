import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore, StepPlan
from oracle import restate as O
dev = torch.device("cuda:0")
layers=[(os.environ.get("L","gex"),int(os.environ.get("F","5000")))]; B=int(os.environ.get("B","32"))
variables=[("y","numerical",1)]
aspec=ArchSpec("DirectPred",layers,64,0.25,16,variables,None,None,True)
ospec=O.Spec("DirectPred",layers,64,0.25,16,variables,None,None,True)
dat, ann = O.synthetic_cohort(layers, 512, seed=1234)
st = O.init_state(ospec, seed=3)
gen = torch.Generator().manual_seed(99)
idx = torch.randperm(512, generator=gen)
for fused in (True, False):
    store = ParamStore(aspec, dev); store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=fused, supplied_draws=True)
    g2 = torch.Generator().manual_seed(5)
    y = {k: ann[k][idx[:B]] for k in plan.y}
    draws = {n: (torch.rand(t.shape, generator=g2) < 0.9).float() for n,t in plan.draws.items()}
    xs=[dat[n][idx[:B]] for n,_ in layers]
    plan.set_batch(x_list=[x.to(dev) for x in xs], y={k:v.to(dev) for k,v in y.items()})
    plan.set_draws({k:v.to(dev) for k,v in draws.items()})
    plan.train_step(1e-3)
    print("fused",fused,"engine gnorm",float(store.ctrl[5]), plan.losses())
    if not fused:
        tot=0.0
        for k in store.param_keys:
            gk=store.g(k)
            tot+=float((gk.double()**2).sum())
        print("  fp64 norm of engine grads", tot**0.5)
        eng={k:store.g(k).detach().cpu().clone() for k in store.param_keys}
for dt in (torch.float32, torch.float64):
    s2={k:(v.to(dt) if v.is_floating_point() else v) for k,v in st.items()}
    b={"x":[x.to(dt) for x in xs],"y":{k:v.to(dt) for k,v in y.items()}}
    d={k:v.to(dt) for k,v in draws.items()}
    _,_,info=O.train_step(ospec,s2,{},b,d,1e-3)
    print(dt,"oracle gnorm",float(info["grad_norm"]), {k:float(v.reshape(-1)[0]) for k,v in info["losses"].items()})
    if dt==torch.float64:
        for k,gv in info["grads"].items():
            e=(eng[k].double()-gv).abs().max().item(); r=gv.abs().max().item()
            print(f"   {k}: max abs err {e:.3e} ref max {r:.3e} | norm eng {eng[k].double().norm():.6e} ref {gv.norm():.6e}")
