import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore, StepPlan
from oracle import restate as O
dev = torch.device("cuda:0")
layers=[("gex",5000)]; B=32
variables=[("y","numerical",1)]
aspec=ArchSpec("DirectPred",layers,64,0.25,16,variables,None,None,True)
ospec=O.Spec("DirectPred",layers,64,0.25,16,variables,None,None,True)
dat, ann = O.synthetic_cohort(layers, 512, seed=1234)
st = O.init_state(ospec, seed=3)
gen = torch.Generator().manual_seed(99)
idx = torch.randperm(512, generator=gen)
g2 = torch.Generator().manual_seed(5)
k='encoders.0.layer_1.weight'
res={}
for prec in ("f32","bf16x3"):
    store = ParamStore(aspec, dev); store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=False, supplied_draws=True, precision=prec)
    if prec=="f32":
        y = {kk: ann[kk][idx[:B]] for kk in plan.y}
        draws = {n: (torch.rand(t.shape, generator=g2) < 0.9).float() for n,t in plan.draws.items()}
        xs=[dat[n][idx[:B]] for n,_ in layers]
    plan.set_batch(x_list=[x.to(dev) for x in xs], y={kk:v.to(dev) for kk,v in y.items()})
    plan.set_draws({kk:v.to(dev) for kk,v in draws.items()})
    plan.train_step(1e-3)
    res[prec]=(store.g(k).detach().cpu().clone(), store.p(k).detach().cpu().clone(), float(store.ctrl[4]))
st2,_,info=O.train_step(ospec,st,{},{"x":xs,"y":y},draws,1e-3)
go=info["grads"][k]; coef=float(info["clip_coef"])
for prec in res:
    ge,we,ce=res[prec]
    bad=((we-st2[k]).abs()>1e-3*st2[k].abs()+2e-5)
    print(prec,"coef",ce,coef,"n bad",int(bad.sum()),"max|g|",float(go.abs().max()))
    ii=bad.nonzero()[:8]
    for (r,c) in ii.tolist():
        print(f"   [{r},{c}] g_oracle={go[r,c]:.3e} g_engine={ge[r,c]:.3e} w0={st[k][r,c]:.6f} w_or={st2[k][r,c]:.6f} w_en={we[r,c]:.6f}")
    print("   |g_or| of bad: min/median/max", float(go[bad].abs().min()) if bad.any() else None, float(go[bad].abs().median()) if bad.any() else None, float(go[bad].abs().max()) if bad.any() else None)
    print("   max abs grad err", float((ge-go).abs().max()))
print("---- forward / gating comparison f32 vs bf16x3")
bufs={}
for prec in ("f32","bf16x3"):
    store = ParamStore(aspec, dev); store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=False, supplied_draws=True, precision=prec)
    plan.set_batch(x_list=[x.to(dev) for x in xs], y={kk:v.to(dev) for kk,v in y.items()})
    plan.set_draws({kk:v.to(dev) for kk,v in draws.items()})
    plan.train_step(1e-3)
    bufs[prec]={n:plan.buf[n].detach().cpu().clone() for n in ("encoders.0/y1","encoders.0/a1","encoders.0/da1")}
    bufs[prec]["g"]=store.g(k).detach().cpu().clone()
y1a,y1b=bufs["f32"]["encoders.0/y1"],bufs["bf16x3"]["encoders.0/y1"]
print("y1 max abs diff",float((y1a-y1b).abs().max()),"max|y1|",float(y1a.abs().max()), "rel fro", float((y1a-y1b).norm()/y1a.norm()))
a1a,a1b=bufs["f32"]["encoders.0/a1"],bufs["bf16x3"]["encoders.0/a1"]
print("gating flips (a1>0 differs):", int(((a1a>0)!=(a1b>0)).sum()), "of", a1a.numel())
d=(bufs["f32"]["g"]-bufs["bf16x3"]["g"]).abs()
rowmax=d.max(1).values
print("rows with grad err>1e-4:", int((rowmax>1e-4).sum()), "top rows", rowmax.topk(5))
da,db=bufs["f32"]["encoders.0/da1"],bufs["bf16x3"]["encoders.0/da1"]
print("dy1 max abs diff", float((da-db).abs().max()), "max|dy1|", float(da.abs().max()))
