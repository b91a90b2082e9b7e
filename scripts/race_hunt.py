"""Which buffer diverges first when a trial runs beside an unrelated GPU load?  Runs the cfg2-shaped eager pipeline twice from
the same state -- alone, and with a noise thread hammering the GPU on another stream -- and compares a checksum of every plan
buffer after every step.   python scripts/race_hunt.py [steps] [features]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Fe = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
DETAIL = os.environ.get("DETAIL", "0") == "1"
layers = [("gex", Fe), ("cnv", Fe)]
spec = ArchSpec("DirectPred", layers, 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
cohort = synthetic_cohort(layers, 2048, dev, seed=1)
torch.manual_seed(3)
init = ParamStore(spec, dev, materialize_big_grads=False).state_dict()
table = torch.randperm(2048, device=dev)[: 12 * 128]


def checksums(pipe):
    out = {}
    for pi, p in enumerate(pipe.plans):
        import re
        for j, (k, t) in enumerate(p.buf.items()):
            if t.dtype == torch.float32 and t.numel() < (1 << 24):
                out[f"plan{pi}/{j}:" + re.sub(r"\d{7,}", "PTR", k)] = t.view(torch.int32).sum(dtype=torch.int64)
        for i, x in enumerate(p.X):
            out[f"plan{pi}/X{i}"] = x.view(torch.int32).sum(dtype=torch.int64)
        out[f"plan{pi}/loss_vec"] = p.loss_vec.view(torch.int32).sum(dtype=torch.int64)
    st = pipe.store
    out["G"] = st.G.view(torch.int32).sum(dtype=torch.int64)
    out["P"] = st.P.view(torch.int32).sum(dtype=torch.int64)
    out["ctrl"] = st.ctrl.view(torch.int32).sum(dtype=torch.int64)
    for k in st.big_keys:
        out["W/" + k] = st.big[k]["_W"].view(torch.int32).sum(dtype=torch.int64)
    return out


def run(noise):
    store = ParamStore(spec, dev, materialize_big_grads=False)
    store.load_state(init)
    stop = [False]
    th = None
    if noise:
        def hammer():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream()):
                st2 = ParamStore(spec, dev, materialize_big_grads=False)
                p2 = PipelinedStep(st2, 64, cohort=cohort, n_batches=12, seed=9)
                p2.idx.copy_(table[: 12 * 64])
                p2.prime()
                k = 0
                while not stop[0]:
                    p2.step(1e-3)
                    k += 1
                    if k % 16 == 0:
                        torch.cuda.current_stream().synchronize()
        th = threading.Thread(target=hammer)
        th.start()
        time.sleep(0.3)
    with torch.cuda.stream(torch.cuda.Stream()):
        pipe = PipelinedStep(store, 128, cohort=cohort, n_batches=12, seed=5)
        pipe.idx.copy_(table)
        pipe.prime()
        hist = []
        raw = []
        for s in range(steps):
            pipe.step(1e-3)
            if DETAIL:
                raw.append(checksums(pipe))
            else:
                raw.append({"loss_vec": pipe.last.loss_vec.clone(), "gnorm": store.ctrl[5:6].clone()})
        torch.cuda.current_stream().synchronize()
        for cs in raw:
            hist.append({k: (int(v) if v.dtype == torch.int64 else tuple(v.tolist())) for k, v in cs.items()})
    stop[0] = True
    if th:
        th.join()
    return hist


ref = run(False)
ref2 = run(False)
print("alone vs alone identical:", ref == ref2, flush=True)
for trial in range(3):
    got = run(True)
    first = next((s for s in range(steps) if got[s] != ref[s]), None)
    if first is None:
        print(f"noise run {trial}: identical over {steps} steps", flush=True)
        continue
    diff = [k for k in ref[first] if got[first][k] != ref[first][k]]
    print(f"noise run {trial}: first divergence at step {first}: {len(diff)} buffers:", diff[:40], flush=True)
