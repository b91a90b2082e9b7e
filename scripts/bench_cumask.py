"""two trials per GPU on disjoint CU sets.  The HBM-bound dW + Adam launches of a trial hold every CU's
register file, so nothing co-runs with them on ordinary streams (two interleaved trials take exactly 2x one, DESIGN.md
section 7).  Here each trial's dominant launches go to a stream restricted to a WIDE CU set and everything else (the
latency-bound chain, batch assembly) to a stream restricted to the remaining NARROW set (hipExtStreamCreateWithCUMask), so
that one trial's narrow chain runs under the other trial's dominant kernels.  Eager launches (a hipGraph runs on one
stream's CU mask).  Prints steps/s of: one trial alone (eager, unmasked), two trials interleaved unmasked, two trials with the
masks, for several splits.     python scripts/bench_cumask.py [steps]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from flexynesis_amd import ops
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.data import synthetic_cohort
from flexynesis_amd.engine import ParamStore, PipelinedStep

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
DOMINANT = "fx_linear_dw_adam_fwd_bf16x3"


def masked_stream(cus):
    words = (N_CU + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def make_trial(seed):
    layers = [("gex", 20000), ("cnv", 20000)]
    spec = ArchSpec("DirectPred", layers, 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
    cohort = synthetic_cohort(layers, 2048, dev, seed=seed)
    torch.manual_seed(seed)
    store = ParamStore(spec, dev, materialize_big_grads=False)
    pipe = PipelinedStep(store, 128, cohort=cohort, n_batches=12, seed=seed)
    pipe.idx.copy_(torch.randperm(2048, device=dev)[: 12 * 128])
    pipe.prime()
    pipe.step(1e-3)
    torch.cuda.synchronize()
    return pipe


def issue_split(pipe, narrow, wide, state):
    """One optimisation step of ``pipe``: dominant launches on ``wide``, everything else on ``narrow``."""
    k = pipe.k
    cur, nxt = pipe.plans[k], pipe.plans[1 - k]
    with torch.cuda.stream(narrow):
        if state.get("done") is not None:
            narrow.wait_event(state["done"])               # the previous step's dominant launches have updated the weights
        ops.step_begin(ops.IMMEDIATE, pipe.store.ctrl, 1e-3, pipe.n_batches)
        cur.t_fwd.run()
        nxt.t_gather.run()                                 # batch assembly of the next step, on the narrow set too
        cur.t_bwd.run()
        tail = [c for seg in cur.t_opt.segments for br in seg for c in br if c[0] is not None]
        for fn, name, args in tail:
            if name != DOMINANT:
                rc = fn(*args, narrow.cuda_stream)
                assert rc == 0, name
        ev = torch.cuda.Event()
        ev.record(narrow)
    with torch.cuda.stream(wide):
        wide.wait_event(ev)
        for fn, name, args in tail:
            if name == DOMINANT:
                rc = fn(*args, wide.cuda_stream)
                assert rc == 0, name
        done = torch.cuda.Event()
        done.record(wide)
    state["done"] = done
    pipe._advance()


def run(pipes, streams, steps):
    states = [dict() for _ in pipes]
    for _ in range(3):
        for p, (n, w), st in zip(pipes, streams, states):
            issue_split(p, n, w, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for p, (n, w), st in zip(pipes, streams, states):
            issue_split(p, n, w, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return len(pipes) * steps / dt


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    a, b = make_trial(1), make_trial(2)
    plain = [(torch.cuda.Stream(), torch.cuda.Stream()) for _ in range(2)]
    one = run([a], plain[:1], steps)
    two = run([a, b], plain, steps)
    print(f"CUs {N_CU}; eager, unmasked: one trial {one:8.1f} steps/s ({1e3 / one:.3f} ms/step); two trials interleaved {two:8.1f} steps/s "
          f"({two / one:.3f} x)", flush=True)
    # graph-replayed single trial for reference
    a.capture(1e-3)
    for _ in range(5):
        a.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        a.replay()
    torch.cuda.synchronize()
    g = steps / (time.perf_counter() - t0)
    print(f"hipGraph replay, one trial: {g:8.1f} steps/s ({1e3 / g:.3f} ms/step)", flush=True)
    # two graph-replayed trials, each on its own stream
    b.capture(1e-3)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(sa):
            a.replay()
        with torch.cuda.stream(sb):
            b.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(sa):
            a.replay()
        with torch.cuda.stream(sb):
            b.replay()
    torch.cuda.synchronize()
    g2 = 2 * steps / (time.perf_counter() - t0)
    print(f"hipGraph replay, two trials on two streams: {g2:8.1f} steps/s = {g2 / g:.3f} x one graph-replayed trial", flush=True)
    a.close()
    b.close()
    for n_narrow, pattern in ((32, "block"), (64, "block"), (96, "block")):
        if pattern == "stride":
            stride = N_CU // n_narrow
            narrow_set = [c for c in range(N_CU) if c % stride == stride - 1]
        else:
            narrow_set = list(range(N_CU - n_narrow, N_CU))
        wide_set = [c for c in range(N_CU) if c not in set(narrow_set)]
        streams = [(masked_stream(narrow_set), masked_stream(wide_set)) for _ in range(2)]
        m1 = run([a], streams[:1], steps)
        m2 = run([a, b], streams, steps)
        print(f"narrow {n_narrow:3d} CUs ({pattern:6s}) / wide {len(wide_set)}: one trial {m1:8.1f} steps/s; two trials {m2:8.1f} steps/s = "
              f"{m2 / one:.3f} x one eager trial, {m2 / g:.3f} x one graph-replayed trial", flush=True)


if __name__ == "__main__":
    main()
