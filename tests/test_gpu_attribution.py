"""compute_feature_importance on the engine (eval forward + input-gradient tapes on the HIP kernels) against the restated
reference computation (oracle/attribution.py; reference models/direct_pred.py:418-590).  GPU, -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(model_name, layers, n=150, seed=0, big=False):
    from flexynesis_amd import models as M
    from flexynesis_amd.data import MultiOmicDataset
    from oracle import restate as O
    g = torch.Generator().manual_seed(seed)
    dat = {k: torch.randn(n, F, generator=g) for k, F in layers}
    ann = {"y": torch.randn(n, generator=g), "c": torch.randint(0, 3, (n,), generator=g).float()}
    feats = {k: [f"{k}_{j}" for j in range(F)] for k, F in layers}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)],
                          {"c": {0: "zero", 1: "one", 2: "two"}})
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 32}
    cls = getattr(M, model_name)
    torch.manual_seed(seed + 1)
    m = cls(cfg, ds, ["c", "y"] if model_name == "MultiTripletNetwork" else ["y", "c"], device_type="cuda")
    # non-trivial running statistics (a freshly initialised BatchNorm is the identity in eval mode)
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    m.load_state_dict(sd)
    variables = [("c", "categorical", 3), ("y", "numerical", 1)] if model_name == "MultiTripletNetwork" else \
        [("y", "numerical", 1), ("c", "categorical", 3)]
    ospec = O.Spec(model_name, layers, 16, 0.5, 8, variables)
    st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return m, ds, ospec, st, dat


@pytest.mark.parametrize("model_name,layers", [
    ("DirectPred", [("gex", 300), ("cnv", 220)]),
    ("DirectPred", [("all", 2500)]),                       # early fusion; hidden 1250 x 2500 weight -> the wide (split-bf16) dX path
    ("MultiTripletNetwork", [("gex", 260), ("cnv", 180)]),
])
@pytest.mark.parametrize("method", ["IntegratedGradients", "GradientShap"])
def test_feature_importance_matches_restated_reference(model_name, layers, method):
    from oracle import attribution as A
    m, ds, ospec, st, dat = _setup(model_name, layers)
    alphas = [0.13, 0.42, 0.58, 0.77, 0.91]
    for var, kind, C in (("y", "numerical", 1), ("c", "categorical", 3)):
        df = m.compute_feature_importance(ds, var, method=method, steps_or_samples=5, batch_size=64, alphas=alphas)
        assert df is m.feature_importances[var]
        assert list(df.columns) == ["target_variable", "target_class", "target_class_label", "layer", "name", "importance"]
        assert len(df) == C * sum(F for _, F in layers)
        ref = A.feature_importance(ospec, st, dat, var, kind, C, method, 5, batch_size=64, alphas=alphas)
        for c in range(C):
            for j, (lname, F) in enumerate(layers):
                got = df[(df.target_class == c) & (df.layer == lname)]
                assert list(got.name) == ds.features[lname]
                if kind == "categorical":
                    assert set(got.target_class_label) == {ds.label_mappings["c"][c]}
                a, b = torch.as_tensor(got.importance.to_numpy()).double(), ref[c][j]
                scale = float(b.abs().max())
                # (the wide layer's input gradient runs on the split-bf16 MFMA path: ~3e-5 of the tensor's scale per element)
                assert float((a - b).abs().max()) <= 1e-3 * scale + 1e-9, (var, c, lname, float((a - b).abs().max()), scale)
                assert float((a - b).norm() / b.norm()) <= 3e-4


def test_feature_importance_argument_checks_and_batching():
    m, ds, _, _, _ = _setup("DirectPred", [("gex", 120), ("cnv", 90)], n=70)
    with pytest.raises(ValueError):
        m.compute_feature_importance(ds, "y", method="Saliency")
    with pytest.raises(KeyError):
        m.compute_feature_importance(ds, "nope")
    # IntegratedGradients is deterministic and independent of how the samples are batched (eval mode: samples independent)
    a = m.compute_feature_importance(ds, "y", steps_or_samples=4, batch_size=512).importance.to_numpy()
    b = m.compute_feature_importance(ds, "y", steps_or_samples=4, batch_size=17).importance.to_numpy()
    assert np.allclose(a, b, rtol=1e-5, atol=1e-9) and a.min() >= 0 and a.max() > 0
    # the model is back in a usable state for training / prediction
    assert set(m.predict(ds)) == {"y", "c"}


@pytest.mark.parametrize("model_name", ["supervised_vae", "CrossModalPred"])
@pytest.mark.parametrize("method", ["IntegratedGradients", "GradientShap"])
def test_vae_family_feature_importance_matches_restated_reference(model_name, method):
    """The VAE family differentiates its heads through the SAMPLED latent z = mean + log_var * eps (reference
    supervised_vae.py:553-563): same eps draws on both sides.  CrossModalPred attributes its input layers."""
    from flexynesis_amd import models as M
    from flexynesis_amd.data import MultiOmicDataset
    from oracle import attribution as A
    from oracle import restate as O
    g = torch.Generator().manual_seed(5)
    n = 90
    layers = [("gex", 2400), ("cnv", 300), ("meth", 200)]            # gex: 1200 x 2400 weight -> the wide (split-bf16) dX path
    dat = {k: torch.randn(n, F, generator=g) for k, F in layers}
    ann = {"y": torch.randn(n, generator=g), "c": torch.randint(0, 3, (n,), generator=g).float()}
    feats = {k: [f"{k}_{j}" for j in range(F)] for k, F in layers}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)], {})
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 32}
    kw = dict(input_layers=["gex", "meth"], output_layers=["cnv", "gex"]) if model_name == "CrossModalPred" else {}
    torch.manual_seed(2)
    m = getattr(M, model_name)(cfg, ds, ["y", "c"], device_type="cuda", **kw)
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    m.load_state_dict(sd)
    ospec = O.Spec(model_name, layers, 16, 0.5, 8, [("y", "numerical", 1), ("c", "categorical", 3)],
                   input_layers=kw.get("input_layers"), output_layers=kw.get("output_layers"))
    st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    alphas = [0.2, 0.5, 0.9]
    draws = {}

    def eps_of(b, di, rows):
        key = (b, di)
        if key not in draws:
            draws[key] = torch.randn(rows, 16, generator=torch.Generator().manual_seed(100 * b + di))
        return draws[key]

    in_layers = kw.get("input_layers", ["gex", "cnv", "meth"])
    for var, kind, C in (("y", "numerical", 1), ("c", "categorical", 3)):
        df = m.compute_feature_importance(ds, var, method=method, steps_or_samples=3, batch_size=64, alphas=alphas,
                                          eps=lambda b, c0, di, rows: eps_of(b, di, rows))
        assert sorted(set(df.layer)) == sorted(in_layers)
        ref = A.feature_importance(ospec, st, dat, var, kind, C, method, 3, batch_size=64, alphas=alphas, eps=eps_of)
        names = [ospec.layers[i][0] for i in ospec.enc_idx]
        for c in range(C):
            for j, lname in enumerate(names):
                a = torch.as_tensor(df[(df.target_class == c) & (df.layer == lname)].importance.to_numpy()).double()
                b = ref[c][j]
                # (a LeakyReLU input within rounding of zero switches its slope on one side only: isolated elements)
                assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-9, (var, c, lname)
                assert float((a - b).norm() / b.norm()) <= 1e-3
    # production mode draws its own eps per forward and still returns finite, non-negative importances
    df = m.compute_feature_importance(ds, "y", steps_or_samples=2, batch_size=64)
    assert np.isfinite(df.importance.to_numpy()).all() and df.importance.min() >= 0


@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
@pytest.mark.parametrize("method", ["IntegratedGradients", "GradientShap"])
def test_gnn_feature_importance_matches_restated_reference(conv, method):
    """GNN (reference gnn_early.py:427-631): attributions of the node features through heads, fc and both graph convolutions
    (BatchNorm on its running statistics, ReLU gates), reported per omics layer and node."""
    import pandas as pd
    from flexynesis_amd.data import MultiOmicDataset, MultiOmicDatasetNW
    from flexynesis_amd.models import GNN
    from oracle import attribution as A
    from oracle import restate as O
    n, genes = 70, 90
    g = torch.Generator().manual_seed(3)
    names = [f"G{i}" for i in range(genes)]
    dat = {"gex": torch.randn(n, genes, generator=g), "cnv": torch.randn(n, genes - 12, generator=g)}
    feats = {"gex": names, "cnv": names[6:genes - 6]}
    ann = {"y": torch.randn(n, generator=g), "c": torch.randint(0, 3, (n,), generator=g).float()}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)],
                          {"c": {0: "zero", 1: "one", 2: "two"}})
    rng = np.random.default_rng(1)
    a, b = rng.integers(0, genes + 8, 500), rng.integers(0, genes + 8, 500)
    nw = MultiOmicDatasetNW(ds, pd.DataFrame({"protein1": [f"G{i}" for i in a], "protein2": [f"G{i}" for i in b]}))
    nodes, nf = len(nw.common_features), 2
    cfg = {"latent_dim": 12, "node_embedding_dim": 6, "num_convs": 2, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1,
           "batch_size": 32, "activation": "relu"}
    torch.manual_seed(4)
    m = GNN(cfg, nw, ["y", "c"], device_type="cuda", gnn_conv_type=conv)
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    m.load_state_dict(sd)
    gn = dict(nodes=nodes, node_features=nf, embedding_dim=6, num_convs=2, conv=conv, act="relu", edge_index=nw.edge_index)
    ospec = O.Spec("GNN", [("nodes", nodes * nf)], 12, 0.0, 8, [("y", "numerical", 1), ("c", "categorical", 3)], gnn=gn)
    st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    flat = {"nodes": nw.dat["nodes"].cpu()}
    alphas = [0.21, 0.48, 0.66, 0.93]
    for var, kind, C in (("y", "numerical", 1), ("c", "categorical", 3)):
        df = m.compute_feature_importance(nw, var, method=method, steps_or_samples=4, batch_size=32, alphas=alphas)
        assert list(df.columns) == ["target_variable", "target_class", "target_class_label", "layer", "name", "importance"]
        assert len(df) == C * nf * nodes and df is m.feature_importances[var]
        ref = A.feature_importance(ospec, st, flat, var, kind, C, method, 4, batch_size=32, alphas=alphas)
        for c in range(C):
            want = ref[c][0].reshape(nodes, nf)
            for li, lname in enumerate(ds.dat.keys()):
                got = df[(df.target_class == c) & (df.layer == lname)]
                assert list(got.name) == nw.common_features
                x, y = torch.as_tensor(got.importance.to_numpy()).double(), want[:, li]
                assert float((x - y).abs().max()) <= 1e-4 * float(y.abs().max()) + 1e-9, (conv, var, c, lname)
