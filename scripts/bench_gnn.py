"""GNN training step at a STRING-sized graph: samples/s through hipGraph replay, and the oracle beside it.

    python scripts/bench_gnn.py [--nodes 8000] [--deg 25] [--feat 2] [--emb 16] [--convs 2] [--batch 32] [--conv GC] [--cpu]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.arch import ArchSpec  # noqa: E402
from flexynesis_amd.data import DeviceCohort  # noqa: E402
from flexynesis_amd.engine import ParamStore, PipelinedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=8000)
    ap.add_argument("--deg", type=int, default=25)
    ap.add_argument("--feat", type=int, default=2)
    ap.add_argument("--emb", type=int, default=16)
    ap.add_argument("--convs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--conv", default="GC")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--big-threshold", type=int, default=1 << 20, help="weights with at least this many elements take the wide-layer path")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    E = a.nodes * a.deg
    # preferential-attachment-like degree skew: half of the endpoints drawn from a squared-uniform distribution
    src = rng.integers(0, a.nodes, E)
    dst = np.where(rng.random(E) < 0.5, (rng.random(E) ** 2 * a.nodes).astype(np.int64), rng.integers(0, a.nodes, E))
    ei = np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])])          # both directions, as STRING lists them
    variables = [("y", "numerical", 1), ("c", "categorical", 4)]
    gn = dict(nodes=a.nodes, node_features=a.feat, embedding_dim=a.emb, num_convs=a.convs, conv=a.conv, act="relu", edge_index=ei)
    spec = ArchSpec("GNN", [("nodes", a.nodes * a.feat)], 64, 0.0, 16, variables, gnn=gn)
    N = 1024
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(N, a.nodes * a.feat, generator=g, device=dev)
    ann = {"y": X[:, :16].sum(1) / 4, "c": torch.randint(0, 4, (N,), generator=g, device=dev).float()}
    cohort = DeviceCohort({"nodes": X}, ann, dev)
    store = ParamStore(spec, dev, big_threshold=a.big_threshold)
    nb = N // a.batch
    pipe = PipelinedStep(store, a.batch, cohort=cohort, n_batches=nb, seed=1, epoch_acc=True)
    def reshuffle():
        pipe.idx.copy_(torch.randperm(N, device=dev)[: nb * a.batch])

    reshuffle()
    pipe.prime()
    pipe.step(1e-3)
    pipe.capture(1e-3)

    def run(k):
        for _ in range(k):
            if pipe.epoch_end_next():
                reshuffle()
            pipe.replay()

    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    P = spec.param_count()
    acts = a.batch * a.nodes * a.emb * 4
    print(f"GNN {a.conv}: nodes {a.nodes}, edges {ei.shape[1]}, features {a.feat}, emb {a.emb}, convs {a.convs}, B {a.batch}, "
          f"params {P/1e6:.2f} M (fc {64*a.nodes*a.emb/1e6:.2f} M), activations {acts/1e6:.1f} MB/layer")
    print(f"step {dt*1e3:.3f} ms  ->  {a.batch/dt:,.0f} samples/s   launches/step {pipe.n_launches()}  losses {pipe.losses()}")
    gathered = a.batch * ei.shape[1] * 4 * (a.feat + a.emb * max(a.convs - 1, 0) + a.emb * a.convs)
    print(f"message-passing gather volume {gathered/1e9:.2f} GB/step (L2 side) -> {gathered/dt/1e12:.2f} TB/s if the step were only that")
    if a.cpu:
        from oracle import restate as O
        ospec = O.Spec("GNN", [("nodes", a.nodes * a.feat)], 64, 0.0, 16, variables, gnn=dict(gn, edge_index=torch.from_numpy(ei)))
        st = {k: v.cpu() for k, v in store.state_dict().items()}
        xb = X[: a.batch].cpu()
        yb = {k: v[: a.batch].cpu() for k, v in ann.items()}
        draws = {}
        O.train_step(ospec, st, {}, {"x": [xb], "y": yb}, draws, 1e-3)
        t0 = time.perf_counter()
        for _ in range(3):
            O.train_step(ospec, st, {}, {"x": [xb], "y": yb}, draws, 1e-3)
        tc = (time.perf_counter() - t0) / 3
        print(f"CPU oracle ({torch.get_num_threads()} threads): {tc*1e3:.0f} ms/step -> {a.batch/tc:,.0f} samples/s; engine {tc/dt:.0f}x")


if __name__ == "__main__":
    main()
