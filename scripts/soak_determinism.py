"""Race detector: the pipelined, branch-parallel hipGraph step must be bit-reproducible.  Two runs of N replays from
the same state / tables / seeds at cfg2 scale; every parameter, moment and the loss curve must match bitwise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep
dev = torch.device("cuda:0")
name = os.environ.get("CFG", "cfg2")
cfg = bench.CONFIGS[name]
N = int(os.environ.get("N", "300"))
spec = ArchSpec(cfg["model"], cfg["layers"], 64, 0.25, 16, cfg["variables"], cfg["surv"][0], cfg["surv"][1], True)
cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1)
rows = 128 * (3 if cfg["model"] == "MultiTripletNetwork" else 1)
n_train = int(cfg["n_samples"] * 0.8)
nb = max(n_train // rows, 1)
torch.manual_seed(0)
init = ParamStore(spec, dev, materialize_big_grads=False).state_dict()
def run():
    store = ParamStore(spec, dev, materialize_big_grads=False)
    store.load_state(init)
    pipe = PipelinedStep(store, 128, cohort=cohort, n_batches=nb, seed=3)
    g = torch.Generator(device=dev); g.manual_seed(7)
    def table():
        pipe.idx.copy_(torch.randint(0, n_train, (nb * rows,), generator=g, device=dev))
    table(); pipe.prime(); pipe.step(1e-3); pipe.capture(1e-3)
    curve = []
    for i in range(N):
        if pipe.epoch_end_next():
            table()
        pipe.replay()
        curve.append(pipe.last.loss_vec.clone())
    torch.cuda.synchronize()
    return torch.stack(curve).cpu(), {k: v.clone() for k, v in store.state_dict().items()}, \
        {k: (store.m(k).clone(), store.v(k).clone()) for k in store.param_keys}
c1, s1, o1 = run()
c2, s2, o2 = run()
ok = torch.equal(c1, c2) and all(torch.equal(s1[k], s2[k]) for k in s1) and \
    all(torch.equal(o1[k][0], o2[k][0]) and torch.equal(o1[k][1], o2[k][1]) for k in o1)
print(f"{name}: {N} replays x 2 runs: bitwise identical = {ok}; finite = {bool(torch.isfinite(c1).all())}; "
      f"loss first/last = {float(c1[0, -1]):.5f} / {float(c1[-1, -1]):.5f}")
sys.exit(0 if ok else 1)
