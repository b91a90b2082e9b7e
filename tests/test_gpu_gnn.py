"""GNN (flexGCN encoder + supervisor heads; reference models/gnn_early.py, modules.py:153-262) on the engine against the
CPU oracle: one optimisation step at a time from the oracle's state, for every conv type the CLI offers, several
activations, wide fc weights on the split-bf16 path, partial-width channels and a survival head."""
import numpy as np
import pytest
import torch

from test_gpu_parity import close, noise_atol

pytestmark = pytest.mark.gpu


def make_specs(conv, act, nodes, nf, C, K, variables, surv=(None, None), L=16, E=None, seed=0):
    from flexynesis_amd.arch import ArchSpec
    from oracle import restate as O
    g = torch.Generator().manual_seed(seed)
    E = E or nodes * 6
    ei = torch.randint(0, nodes, (2, E), generator=g)
    ei[1, : E // 6] = 3                                                # a hub
    ei[:, E // 6: E // 6 + 4] = ei[:, :4]                              # duplicate edges
    ei[0, -3:] = ei[1, -3:]                                            # self loops
    gn = dict(nodes=nodes, node_features=nf, embedding_dim=C, num_convs=K, conv=conv, act=act)
    ospec = O.Spec("GNN", [("nodes", nodes * nf)], L, 0.0, 8, variables, surv[0], surv[1], True, gnn=dict(gn, edge_index=ei))
    aspec = ArchSpec("GNN", [("nodes", nodes * nf)], L, 0.0, 8, variables, surv[0], surv[1], True, gnn=dict(gn, edge_index=ei.numpy()))
    return aspec, ospec


CASES = [
    ("GC", "relu", 300, 2, 16, 2, 64),
    ("SAGE", "relu", 257, 1, 8, 3, 32),
    ("GCN", "relu", 200, 3, 32, 1, 50),
    ("GC", "gelu", 120, 2, 7, 2, 37),          # width not a power of two, odd batch
    ("SAGE", "tanh", 90, 3, 5, 2, 16),
    ("GCN", "leakyrelu", 150, 1, 12, 4, 24),
    ("GC", "sigmoid", 64, 2, 4, 1, 8),
]


@pytest.mark.parametrize("conv,act,nodes,nf,C,K,B", CASES)
@pytest.mark.parametrize("wide", [False, True])
def test_gnn_engine_vs_oracle_steps(conv, act, nodes, nf, C, K, B, wide):
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = torch.device("cuda:0")
    variables = [("y", "numerical", 1), ("c", "categorical", 3), ("event", "numerical", 1)]
    aspec, ospec = make_specs(conv, act, nodes, nf, C, K, variables, ("event", "time"))
    st = O.init_state(ospec, seed=2)
    # wide = the fc weight [L, nodes*C] goes through the split-bf16 MFMA kernels and the fused dW+clip+Adam epilogue
    store = ParamStore(aspec, dev, big_threshold=(1 << 10) if wide else (1 << 30))
    assert bool(store.big_keys) == wide
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    gen = torch.Generator().manual_seed(7)
    N = 128
    X = torch.randn(N, nodes * nf, generator=gen)
    ann = {"y": torch.randn(N, generator=gen), "c": torch.randint(0, 3, (N,), generator=gen).float(),
           "time": torch.rand(N, generator=gen) * 10, "event": (torch.rand(N, generator=gen) < 0.5).float()}
    ann["y"][::9] = float("nan")
    opt, lr = {}, 1e-3
    for step in range(2):
        if step > 0:
            store.load_state(st)
            store.reset_optimizer()
            store.load_optimizer(opt["t"], opt["m"], opt["v"])
        idx = torch.randperm(N, generator=gen)[:B]
        y = {k: ann[k][idx] for k in plan.y}
        draws = {name: (torch.rand(t.shape, generator=gen) < (0.8 if ".drop." in name else 0.9)).float()
                 for name, t in plan.draws.items()}
        assert any(".drop." in k for k in draws)
        batch = {"x": [X[idx]], "y": y}
        plan.set_batch(x_list=[X[idx].to(dev)], y={k: v.to(dev) for k, v in y.items()})
        plan.set_draws({k: v.to(dev) for k, v in draws.items()})
        plan.train_step(lr)
        st_prev = st
        st, opt, info = O.train_step(ospec, st, opt, batch, draws, lr)
        got = plan.losses()
        for k, v in info["losses"].items():
            close(got[k], v, 2e-5, 1e-6, f"{conv}/{act} step{step} loss {k}")          # gate 1e-4
        exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
        close(store.ctrl[5], exact, 1e-4, 1e-7, "grad_norm")
        sd = store.state_dict()
        for k in st:
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(st[k]), k
                continue
            if k in store.big_keys:
                a, b_ = sd[k].double(), st[k].double()
                bad = (a - b_).abs() > 2e-5 + 1e-3 * b_.abs()
                assert float(bad.double().mean()) <= 2e-3, f"{k}: {int(bad.sum())} elements differ"
                assert float((a - b_).norm() / (b_ - st_prev[k].double()).norm()) <= 2e-2, k
                continue
            # conv biases (and head layer_1 biases) sit in front of a BatchNorm: their true gradient is exactly zero and
            # what any implementation computes is rounding noise that Adam turns into +-lr (see noise_atol)
            atol = noise_atol(info["grads"].get(k), info["grad_norm"], lr, 3e-6)
            close(sd[k], st[k], 2e-4, atol, f"{conv}/{act} step{step} {k}")


def test_gnn_eval_forward_and_graph_replay():
    """validation-mode forward (running statistics, no dropout) equals the oracle; a captured training graph replays
    bit-identically from the same state."""
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = torch.device("cuda:0")
    variables = [("y", "numerical", 1), ("c", "categorical", 4)]
    aspec, ospec = make_specs("GC", "relu", 180, 2, 16, 2, variables)
    st = O.init_state(ospec, seed=5)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = torch.randn_like(st[k]) * 0.1
        if k.endswith("running_var"):
            st[k] = torch.rand_like(st[k]) + 0.5
    B = 20
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, 360, generator=gen)
    y = {"y": torch.randn(B, generator=gen), "c": torch.randint(0, 4, (B,), generator=gen).float()}
    store = ParamStore(aspec, dev)
    store.load_state(st)
    ev = StepPlan(store, B, train=False)
    ev.set_batch(x_list=[x.to(dev)], y={k: v.to(dev) for k, v in y.items()})
    ev.forward()
    losses, aux = O.eval_losses(ospec, st, {"x": [x], "y": y})
    got = ev.losses()
    for k, v in losses.items():
        close(got[k], v, 2e-5, 1e-6, f"eval loss {k}")
    close(ev.embeddings, aux["embeddings"], 1e-4, 1e-5, "embeddings")
    outs = []
    for _ in range(2):
        store.load_state(st)
        store.reset_optimizer()
        plan = StepPlan(store, B, train=True, fused=True, seed=3)
        plan.set_batch(x_list=[x.to(dev)], y={k: v.to(dev) for k, v in y.items()})
        plan.capture(1e-3, gather=False)
        for _ in range(3):
            plan.replay()
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in store.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def _nw_dataset(n=96, genes=120, seed=0):
    import pandas as pd
    from flexynesis_amd.data import MultiOmicDataset, MultiOmicDatasetNW
    g = torch.Generator().manual_seed(seed)
    names = [f"G{i}" for i in range(genes)]
    dat = {"gex": torch.randn(n, genes, generator=g), "cnv": torch.randn(n, genes - 20, generator=g)}
    feats = {"gex": names, "cnv": names[10:genes - 10]}                # cnv lacks 20 genes -> per-sample median fill
    y = dat["gex"][:, :8].sum(1) / 3 + 0.1 * torch.randn(n, generator=g)
    c = (dat["gex"][:, 8] > 0).float()
    ds = MultiOmicDataset(dat, {"y": y, "c": c}, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)], {})
    rng = np.random.default_rng(seed)
    a, b = rng.integers(0, genes + 15, 700), rng.integers(0, genes + 15, 700)          # some proteins are not features
    inter = pd.DataFrame({"protein1": [f"G{i}" for i in a], "protein2": [f"G{i}" for i in b]})
    return ds, MultiOmicDatasetNW(ds, inter)


def test_nw_dataset_semantics():
    ds, nw = _nw_dataset()
    x0, y0, s0 = nw[0]
    nodes = len(nw.common_features)
    assert x0.shape == (nodes, 2) and s0 == "s0" and set(y0) == {"y", "c"}
    assert nw.edge_index.shape[0] == 2 and int(nw.edge_index.max()) < nodes
    assert nw.common_features == sorted(nw.common_features)
    gi = nw.gene_to_index
    gname = nw.common_features[5]
    assert float(x0[gi[gname], 1]) == float(ds.dat["gex"][0, int(gname[1:])])           # sorted layers: cnv = 0, gex = 1
    missing = [g_ for g_ in nw.common_features if g_ not in set(ds.features["cnv"])]
    assert missing
    col = nw.node_features_tensor[0, :, 0]
    have = torch.tensor([g_ not in set(missing) for g_ in nw.common_features])
    assert torch.allclose(col[~have], torch.nanmedian(torch.where(have, col, torch.tensor(float("nan")))))
    assert nw.dat["nodes"].shape == (len(nw), nodes * 2)
    sub = nw.subset([3, 5, 8])
    assert len(sub) == 3 and torch.equal(sub.edge_index, nw.edge_index)


@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
def test_gnn_model_class_protocol_and_fit(conv):
    from flexynesis_amd.models import GNN
    from flexynesis_amd.fit import fit
    ds, nw = _nw_dataset()
    cfg = {"latent_dim": 16, "node_embedding_dim": 8, "num_convs": 2, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 3,
           "batch_size": 32, "activation": "relu"}
    torch.manual_seed(0)
    model = GNN(cfg, nw, ["y", "c"], device_type="cuda", gnn_conv_type=conv)
    keys = list(model.state_dict().keys())
    first = {"GC": "encoders.0.convs.0.lin_rel.weight", "SAGE": "encoders.0.convs.0.lin_l.weight", "GCN": "encoders.0.convs.0.lin.weight"}[conv]
    assert first in keys and "encoders.0.bns.1.running_var" in keys and "encoders.0.fc.weight" in keys
    assert tuple(model.state_dict()["encoders.0.fc.weight"].shape) == (16, 8 * len(nw.common_features))
    # Lightning protocol: training_step -> backward -> Adam, validation_step
    loader = torch.utils.data.DataLoader(nw, batch_size=32, shuffle=False, drop_last=True)
    opt = model.configure_optimizers()
    batch = next(iter(loader))
    model.train()
    l0 = None
    for _ in range(4):
        opt.zero_grad()
        loss = model.training_step(batch, 0, log=False)
        loss.sum().backward()
        opt.step()
        l0 = l0 if l0 is not None else float(loss.sum())
    assert float(loss.sum()) < l0
    v = model.validation_step(batch, 0, log=False)
    assert np.isfinite(float(v))
    # engine loop + inference API
    r = fit(model, nw, np.arange(0, 64), np.arange(64, 96), batch_size=32, epochs=3, lr=1e-3, patience=5, seed=0)
    assert np.isfinite(r.val_loss) and r.steps == 6
    pred = model.predict(nw)
    assert pred["y"].shape == (96, 1) and pred["c"].shape == (96, 2) and np.allclose(pred["c"].sum(1), 1, atol=1e-5)
    emb = model.transform(nw)
    assert emb.shape == (96, 16) and list(emb.index[:2]) == ["s0", "s1"]
    model.eval()
    out = model(nw.node_features_tensor[:5], nw.edge_index)
    assert np.allclose(out["y"].cpu().numpy(), pred["y"][:5], atol=1e-5)
    import copy
    m2 = copy.deepcopy(model)
    m2.load_state_dict(model.state_dict())
    assert np.allclose(m2.predict(nw)["y"], pred["y"], atol=1e-6)


@pytest.mark.parametrize("frozen", [("encoders.",), ("MLPs.",)])
def test_gnn_finetune_step_frozen_groups_vs_oracle(frozen):
    """FineTuner configuration (reference main.py:530-539,562-566): requires_grad=False groups, no clipping."""
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = torch.device("cuda:0")
    variables = [("y", "numerical", 1), ("c", "categorical", 3)]
    aspec, ospec = make_specs("GC", "relu", 150, 2, 8, 2, variables)
    st = O.init_state(ospec, seed=6)
    B = 24
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, 300, generator=gen)
    y = {"y": torch.randn(B, generator=gen), "c": torch.randint(0, 3, (B,), generator=gen).float()}
    store = ParamStore(aspec, dev, big_threshold=1 << 10)
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True, clip=False, frozen=frozen)
    draws = {n: (torch.rand(t.shape, generator=gen) < (0.8 if ".drop." in n else 0.9)).float() for n, t in plan.draws.items()}
    plan.set_batch(x_list=[x.to(dev)], y={k: v.to(dev) for k, v in y.items()})
    plan.set_draws({k: v.to(dev) for k, v in draws.items()})
    plan.train_step(1e-3)
    st2, _, info = O.train_step(ospec, st, {}, {"x": [x], "y": y}, draws, 1e-3, clip=False, frozen=frozen)
    got = plan.losses()
    for k, v in info["losses"].items():
        close(got[k], v, 2e-5, 1e-6, f"loss {k}")
    sd = store.state_dict()
    for k in st:
        if k.endswith("num_batches_tracked"):
            continue
        if k.startswith(frozen) and "running" not in k:
            assert torch.equal(sd[k], st[k]), f"frozen parameter {k} moved"
        elif k in store.big_keys:
            a, b_ = sd[k].double(), st2[k].double()
            assert float((a - b_).norm() / (b_ - st[k].double()).norm()) <= 2e-2, k
        else:
            atol = noise_atol(info["grads"].get(k), info["grad_norm"], 1e-3, 3e-6)
            close(sd[k], st2[k], 2e-4, atol, k)


def test_gnn_run_trial_and_fine_tune():
    from flexynesis_amd.models import GNN
    from flexynesis_amd.fit import run_trial, fine_tune
    ds, nw = _nw_dataset(n=80)
    params = {"latent_dim": 16, "node_embedding_dim": 6, "num_convs": 2, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 2,
              "batch_size": 16, "activation": "relu"}
    val, epochs, model, info = run_trial(GNN, params, nw, ["y", "c"], seed=1, device="cuda:0", gnn_conv_type="SAGE")
    assert np.isfinite(val) and epochs == 2 and model.spec.gnn["conv"] == "SAGE", info
    final, best, results = fine_tune(model, nw, n_splits=2, batch_size=16, max_epoch=2, learning_rates=[1e-3],
                                     freeze_configs=[{"encoders": True, "supervisors": False}, {"encoders": False, "supervisors": False}])
    assert len(results) == 2 and np.isfinite(best["average_val_loss"])
    assert np.isfinite(final.predict(nw)["y"]).all()


def test_gnn_and_ingest_bad_arguments_raise():
    from flexynesis_amd import ops
    from flexynesis_amd.ops import FxError
    rec = ops.ImmediateRecorder()
    x = torch.zeros(2, 10, 40, device="cuda")                      # C = 40 > 32
    rp = torch.zeros(11, dtype=torch.int32, device="cuda")
    e = torch.zeros(1, dtype=torch.int32, device="cuda")
    w = torch.zeros(1, device="cuda")
    with pytest.raises(FxError):
        ops.spmm_rows(rec, torch.zeros_like(x), x, rp, e, w)
    x8 = torch.zeros(2, 10, 8, device="cuda")
    with pytest.raises(FxError):
        ops.spmm_rows(rec, x8, x8, rp, e, w)                        # in place
    with pytest.raises(FxError):
        ops.rowlin2(rec, torch.zeros(20, 8, device="cuda"), x8, torch.zeros(4, 4, device="cuda"))
    with pytest.raises(FxError):
        ops.bn_rows_fwd(rec, x8.clone(), x8, w, w, w, w, None, None, 9, False, 0.0, ops.gnn_scratch(20, 32, "cuda"))   # unknown act
    with pytest.raises(FxError):
        ops.col_moments(rec, torch.zeros(4, 4, dtype=torch.float16, device="cuda"))
    with pytest.raises(FxError):
        ops.ingest_transform(rec, torch.zeros(4, 4, device="cuda"), torch.zeros(3, 4, device="cuda"))
    from flexynesis_amd.arch import gnn_conv_keys
    with pytest.raises(ValueError):
        gnn_conv_keys("x", "GAT")


@pytest.mark.parametrize("seed", list(range(16)))
def test_gnn_random_configurations_one_step_vs_oracle(seed):
    """Randomised shapes: any width 1..32 (vector and scalar kernel layouts), 1-3 node features, 1-4 conv layers, odd node
    counts, sparse to dense graphs with isolated nodes, batch 2..70, all conv types and activations."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    rng = np.random.default_rng(1000 + seed)
    conv = ["GC", "SAGE", "GCN"][seed % 3]
    act = ["relu", "sigmoid", "leakyrelu", "tanh", "gelu"][int(rng.integers(0, 5))]
    nodes = int(rng.integers(20, 400))
    nf, C, K = int(rng.integers(1, 4)), int(rng.integers(1, 33)), int(rng.integers(1, 5))
    B, L = int(rng.integers(2, 71)), int(rng.integers(4, 40))
    E = int(nodes * rng.uniform(0.5, 12))
    ei = torch.from_numpy(rng.integers(0, max(nodes - 5, 2), size=(2, E)))            # the last nodes stay isolated
    variables = [("y", "numerical", 1), ("c", "categorical", int(rng.integers(2, 6)))]
    gn = dict(nodes=nodes, node_features=nf, embedding_dim=C, num_convs=K, conv=conv, act=act)
    ospec = O.Spec("GNN", [("nodes", nodes * nf)], L, 0.0, 8, variables, gnn=dict(gn, edge_index=ei))
    aspec = ArchSpec("GNN", [("nodes", nodes * nf)], L, 0.0, 8, variables, gnn=dict(gn, edge_index=ei.numpy()))
    st = O.init_state(ospec, seed=seed)
    dev = torch.device("cuda:0")
    store = ParamStore(aspec, dev, big_threshold=(1 << 12) if seed % 2 else (1 << 30))
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, nodes * nf, generator=gen)
    y = {"y": torch.randn(B, generator=gen), "c": torch.randint(0, variables[1][2], (B,), generator=gen).float()}
    draws = {n: (torch.rand(t.shape, generator=gen) < (0.8 if ".drop." in n else 0.9)).float() for n, t in plan.draws.items()}
    plan.set_batch(x_list=[x.to(dev)], y={k: v.to(dev) for k, v in y.items()})
    plan.set_draws({k: v.to(dev) for k, v in draws.items()})
    plan.train_step(1e-3)
    st2, _, info = O.train_step(ospec, st, {}, {"x": [x], "y": y}, draws, 1e-3)
    got = plan.losses()
    tag = f"seed {seed}: {conv}/{act} nodes {nodes} nf {nf} C {C} K {K} B {B} E {E}"
    for k, v in info["losses"].items():
        close(got[k], v, 5e-5, 1e-6, f"{tag} loss {k}")                         # gate 1e-4
    exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
    close(store.ctrl[5], exact, 2e-4, 1e-7, f"{tag} grad_norm")


def test_nw_dataset_from_device_resident_layers_trains():
    """Layers that live in HBM (what DeviceImporter / to_dataset produce) give a device-resident node tensor equal to the
    host-built one, and the GNN trains on it without a host round trip."""
    from flexynesis_amd.data import MultiOmicDataset, MultiOmicDatasetNW
    from flexynesis_amd.models import GNN
    from flexynesis_amd.fit import fit
    ds, nw = _nw_dataset(n=64, genes=60)
    ds_dev = MultiOmicDataset({k: v.cuda() for k, v in ds.dat.items()}, ds.ann, ds.variable_types, ds.features, ds.samples, {})
    nw_dev = MultiOmicDatasetNW(ds_dev, nw.interaction_df)
    assert nw_dev.node_features_tensor.is_cuda
    assert torch.equal(nw_dev.node_features_tensor.cpu(), nw.node_features_tensor) and torch.equal(nw_dev.edge_index, nw.edge_index)
    cfg = {"latent_dim": 8, "node_embedding_dim": 4, "num_convs": 1, "lr": 1e-3, "supervisor_hidden_dim": 4, "epochs": 2,
           "batch_size": 16, "activation": "relu"}
    model = GNN(cfg, nw_dev, ["y"], device_type="cuda", gnn_conv_type="GCN")
    r = fit(model, nw_dev, np.arange(0, 48), np.arange(48, 64), batch_size=16, epochs=2, lr=1e-3, patience=5, seed=0)
    assert np.isfinite(r.val_loss) and r.steps == 6
