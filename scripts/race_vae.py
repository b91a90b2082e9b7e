"""Determinism probe for the VAE schedules: the test_pipelined_graph_replay_is_bit_reproducible workload, eager or captured.
   python scripts/race_vae.py eager|graph [runs]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
layers = [("gex", 2600), ("cnv", 2100), ("meth", 1500)]
spec = ArchSpec("supervised_vae", layers, 32, 0.5, 16, [("c", "categorical", 4), ("event", "numerical", 1)], "event", "time", True)
cohort = synthetic_cohort(layers, 500, dev, seed=1)
torch.manual_seed(0)
init = ParamStore(spec, dev, materialize_big_grads=False).state_dict()


def run():
    global STORE
    store = STORE = ParamStore(spec, dev, materialize_big_grads=False)
    store.load_state(init)
    pipe = PipelinedStep(store, 64, cohort=cohort, n_batches=5, seed=3)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    pipe.idx.copy_(torch.randint(0, 500, (5 * 64,), generator=g, device=dev))
    pipe.prime()
    pipe.step(1e-3)
    if mode == "graph":
        pipe.capture(1e-3)
    curve, bufs = [], []
    for it in range(40):
        if pipe.epoch_end_next():
            pipe.idx.copy_(torch.randint(0, 500, (5 * 64,), generator=g, device=dev))
        if mode == "graph":
            pipe.replay()
        else:
            pipe.step(1e-3)
        p = pipe.last
        curve.append(p.loss_vec.clone())
        bufs.append((store.G.clone(), p.slots.clone(), store.ctrl.clone()))
    torch.cuda.synchronize()
    pipe.close()
    return torch.stack(curve).cpu(), bufs


ref = run()
for r in range(1, runs):
    c, b = run()
    same = torch.equal(c, ref[0])
    first = next((i for i in range(len(c)) if not torch.equal(c[i], ref[0][i])), None)
    print(f"{mode} run {r}: curves equal = {same}; first differing step = {first}")
    for it, (b0, b1) in enumerate(zip(ref[1], b)):
        if not torch.equal(b0[0], b1[0]) or not torch.equal(b0[1], b1[1]):
            bad = []
            for k in STORE.off:
                if k in STORE.big:
                    continue
                o, n = STORE.off[k]
                if not torch.equal(b0[0][o:o + n], b1[0][o:o + n]):
                    d = (b0[0][o:o + n] - b1[0][o:o + n]).abs()
                    bad.append((k, int((d > 0).sum()), n, float(d.max())))
            sl = (b0[1] != b1[1]).nonzero().flatten().tolist()
            print(f"   step {it}: G keys differing: {bad[:30]}; slots differing: {sl[:20]} of {b0[1].numel()}")
            break
