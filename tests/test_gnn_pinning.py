"""What of the GNN path can be pinned against the reference without torch_geometric (not installed here).

PINNED against the reference's own code, run live from /root/reference:
  * MultiOmicDatasetNW (data.py:1153-1266): node set, edge_index, node feature tensor with the per-sample median fill;
  * flexGCN.forward's wrapper (modules.py:251-262) and GNN.training_step (gnn_early.py:160-198): BatchNorm over the
    batch*nodes rows, the activation table, Dropout(0.2), flatten, fc, heads, losses, uncertainty weighting -- with the
    three convolution classes replaced by STAND-INS that evaluate the oracle's restated aggregation.
NOT pinned: the arithmetic of GraphConv / SAGEConv / GCNConv themselves (oracle/restate.py gnn_edges / gnn_conv restate
torch_geometric's published definitions) -- parity unpinned for those, as DESIGN.md states.  CPU only."""
import numpy as np
import pytest
import torch
from torch import nn

from oracle import ref_shim
from oracle import restate as O

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference not present (GPU box)")


class _StandIn(nn.Module):
    """torch_geometric-shaped conv (same parameter names) whose forward is the oracle's restated arithmetic."""
    CONV = None

    def __init__(self, cin, cout):
        super().__init__()
        if self.CONV == "GC":
            self.lin_rel, self.lin_root = nn.Linear(cin, cout), nn.Linear(cin, cout, bias=False)
        elif self.CONV == "SAGE":
            self.lin_l, self.lin_r = nn.Linear(cin, cout), nn.Linear(cin, cout, bias=False)
        else:
            self.bias = nn.Parameter(torch.zeros(cout))
            self.lin = nn.Linear(cin, cout, bias=False)

    def forward(self, x, edge_index):
        st = {"p." + k: v for k, v in self.named_parameters()}
        return O.gnn_conv(st, "p", x, O.gnn_edges(edge_index, x.shape[1], self.CONV), self.CONV)


def _standins():
    return {c: type("StandIn" + c, (_StandIn,), {"CONV": c}) for c in ("GC", "SAGE", "GCN")}


def _patched_modules():
    ref_shim.install()
    import flexynesis.modules as M
    s = _standins()
    M.GraphConv, M.SAGEConv, M.GCNConv = s["GC"], s["SAGE"], s["GCN"]
    return M


def _graph(nodes, E, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, nodes, (2, E), generator=g)
    ei[0, -2:] = ei[1, -2:]
    return ei


@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
@pytest.mark.parametrize("act", ["relu", "sigmoid", "leakyrelu", "tanh", "gelu"])
def test_flexgcn_wrapper_matches_reference(conv, act):
    from oracle.ref_capture import capture_rng
    M = _patched_modules()
    nodes, nf, C, K, L, B = 23, 2, 6, 2, 5, 9
    ei = _graph(nodes, 80, 1)
    spec = O.Spec("GNN", [("nodes", nodes * nf)], L, 0.0, 4, [("y", "numerical", 1)],
                  gnn=dict(nodes=nodes, node_features=nf, embedding_dim=C, num_convs=K, conv=conv, act=act, edge_index=ei))
    st = O.init_state(spec, seed=4)
    ref = M.flexGCN(nodes, nf, C, L, num_convs=K, conv=conv, act=act)
    assert ref.dropout.p == O.GNN_DROPOUT_P
    sd = {k[len("encoders.0."):]: v for k, v in st.items() if k.startswith("encoders.0.")}
    assert sorted(sd) == sorted(ref.state_dict().keys())                   # parameter / buffer names and shapes of the wrapper
    ref.load_state_dict(sd)
    x = torch.randn(B, nodes, nf, generator=torch.Generator().manual_seed(2))
    ref.train()
    with capture_rng() as cap:
        want = ref(x, ei)
    draws = {f"encoders.0.drop.{k}": m for k, m in enumerate(cap.dropout_masks)}
    nb = {}
    got = O.flexgcn_forward(spec, st, "encoders.0", x, True, draws, nb)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    for k, v in ref.state_dict().items():
        if "running" in k:
            torch.testing.assert_close(nb["encoders.0." + k], v, rtol=1e-6, atol=1e-7)
    ref.eval()
    torch.testing.assert_close(O.flexgcn_forward(spec, {**st, **nb}, "encoders.0", x, False, {}, None), ref(x, ei),
                               rtol=1e-5, atol=1e-6)


def test_nw_dataset_matches_reference():
    import pandas as pd
    ref_shim.install()
    from flexynesis.data import MultiOmicDataset as RefDS, MultiOmicDatasetNW as RefNW
    from flexynesis_amd.data import MultiOmicDataset, MultiOmicDatasetNW
    g = torch.Generator().manual_seed(0)
    n, genes = 12, 40
    names = [f"G{i}" for i in range(genes)]
    dat = {"gex": torch.randn(n, genes, generator=g), "cnv": torch.randn(n, genes - 9, generator=g),
           "mut": torch.randn(n, 7, generator=g)}
    feats = {"gex": pd.Index(names), "cnv": pd.Index(names[4:genes - 5]), "mut": pd.Index(names[30:37])}
    ann = {"y": torch.randn(n, generator=g)}
    rng = np.random.default_rng(0)
    inter = pd.DataFrame({"protein1": [f"G{i}" for i in rng.integers(0, genes + 8, 150)],
                          "protein2": [f"G{i}" for i in rng.integers(0, genes + 8, 150)]})
    samples = [f"s{i}" for i in range(n)]
    want = RefNW(RefDS(dat, ann, {"y": "numerical"}, feats, samples, {}), inter)
    got = MultiOmicDatasetNW(MultiOmicDataset(dat, ann, {"y": "numerical"}, {k: list(v) for k, v in feats.items()}, samples, {}), inter)
    assert got.common_features == want.common_features
    assert torch.equal(got.edge_index, want.edge_index)
    assert torch.equal(got.node_features_tensor, want.node_features_tensor)
    x, y, s = got[3]
    xr, yr, sr = want[3]
    assert torch.equal(x, xr) and s == sr and torch.equal(y["y"], yr["y"])
    assert torch.equal(got.subset([1, 5]).node_features_tensor, want.subset([1, 5]).node_features_tensor)


@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
def test_gnn_training_step_matches_reference(conv):
    """The reference's GNN class (stand-in convs) through zero_grad -> training_step -> backward -> clip -> Adam."""
    import pandas as pd
    from oracle.ref_capture import capture_rng
    _patched_modules()
    import flexynesis.models.gnn_early as GE
    s = _standins()
    GE.flexGCN.__init__.__globals__.update(GraphConv=s["GC"], SAGEConv=s["SAGE"], GCNConv=s["GCN"])
    from flexynesis.data import MultiOmicDataset as RefDS, MultiOmicDatasetNW as RefNW
    g = torch.Generator().manual_seed(3)
    n, genes = 10, 18
    names = [f"G{i}" for i in range(genes)]
    dat = {"gex": torch.randn(n, genes, generator=g), "cnv": torch.randn(n, genes, generator=g)}
    ann = {"y": torch.randn(n, generator=g), "c": torch.randint(0, 3, (n,), generator=g).float()}
    inter = pd.DataFrame({"protein1": [f"G{i}" for i in range(genes)] * 3,
                          "protein2": [f"G{(i * 7 + 3) % genes}" for i in range(genes * 3)]})
    nw = RefNW(RefDS(dat, ann, {"y": "numerical", "c": "categorical"}, {k: pd.Index(names) for k in dat}, [f"s{i}" for i in range(n)], {}), inter)
    cfg = {"latent_dim": 6, "node_embedding_dim": 5, "num_convs": 2, "lr": 1e-3, "supervisor_hidden_dim": 4, "activation": "relu"}
    model = GE.GNN(cfg, nw, ["y", "c"], device_type="cpu", gnn_conv_type=conv)
    nodes = nw.node_features_tensor.shape[1]
    spec = O.Spec("GNN", [("nodes", nodes * 2)], 6, 0.0, 4, [("y", "numerical", 1), ("c", "categorical", 3)],
                  gnn=dict(nodes=nodes, node_features=2, embedding_dim=5, num_convs=2, conv=conv, act="relu", edge_index=nw.edge_index))
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert sorted(st) == sorted(O.state_manifest(spec))
    x = nw.node_features_tensor
    batch = (x, {k: v for k, v in ann.items()}, nw.samples)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()
    opt.zero_grad()
    with capture_rng() as cap:
        loss = model.training_step(batch, 0, log=False)
    loss.sum().backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    masks = list(cap.dropout_masks)
    draws = {"encoders.0.drop.0": masks[0], "encoders.0.drop.1": masks[1], "MLPs.y": masks[2], "MLPs.c": masks[3]}
    st2, _, info = O.train_step(spec, st, {}, {"x": [x.reshape(n, -1)], "y": ann}, draws, 1e-3)
    torch.testing.assert_close(info["losses"]["total"].reshape(-1), loss.detach().reshape(-1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(info["grad_norm"], gn, rtol=1e-4, atol=1e-6)
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(st2[k]) == int(v)
        else:
            torch.testing.assert_close(st2[k], v.detach(), rtol=2e-4, atol=2.2e-3 if k.endswith("bias") else 2e-5)


# ---- independent dense-adjacency formulation of the three graph convolutions -------------------------------------
@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
def test_conv_restatement_matches_dense_adjacency_formulation(conv):
    """torch_geometric is absent, so oracle.restate.gnn_edges / gnn_conv (edge-list form, restated from PyG's docs) cannot
    be pinned to PyG itself.  This is a second, independently written statement of the same layers straight from the
    papers, as dense matrix algebra in fp64 -- Kipf & Welling 2017 (GCN: A_hat = D~^-1/2 (A + I) D~^-1/2), Hamilton et al.
    2017 (GraphSAGE mean aggregator), Morris et al. 2019 (GraphConv: W_root x_i + W_rel sum_j x_j) -- on multigraphs with
    duplicate edges, self loops, hubs and isolated nodes.  Agreement removes the single-author risk, not the PyG gap."""
    g = torch.Generator().manual_seed({"GC": 1, "SAGE": 2, "GCN": 3}[conv])
    n, cin, cout, B, E = 23, 3, 5, 4, 90
    src = torch.randint(0, n - 2, (E,), generator=g)           # the last two nodes stay isolated
    dst = torch.randint(0, n - 2, (E,), generator=g)
    src[:6], dst[:6] = torch.tensor([0, 0, 1, 1, 2, 5]), torch.tensor([0, 0, 1, 3, 2, 5])      # self loops (one doubled)
    src[6:10], dst[6:10] = torch.tensor([4, 4, 4, 7]), torch.tensor([9, 9, 9, 9])               # duplicate edges, a hub
    ei = torch.stack([src, dst])
    x = torch.randn(B, n, cin, generator=g, dtype=torch.float64)
    W1 = torch.randn(cout, cin, generator=g, dtype=torch.float64)
    W2 = torch.randn(cout, cin, generator=g, dtype=torch.float64)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    A = torch.zeros(n, n, dtype=torch.float64)                 # A[i, j] = number of edges j -> i (messages flow source -> target)
    for s_, d_ in zip(src.tolist(), dst.tolist()):
        A[d_, s_] += 1.0
    if conv == "GC":
        st = {"c.lin_rel.weight": W1, "c.lin_rel.bias": b, "c.lin_root.weight": W2}
        want = torch.einsum("ij,bjc->bic", A, x) @ W1.t() + b + x @ W2.t()
    elif conv == "SAGE":
        st = {"c.lin_l.weight": W1, "c.lin_l.bias": b, "c.lin_r.weight": W2}
        deg = A.sum(1)
        Dinv = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg))
        want = torch.einsum("ij,bjc->bic", Dinv[:, None] * A, x) @ W1.t() + b + x @ W2.t()
    else:
        st = {"c.lin.weight": W1, "c.bias": b}
        At = A.clone()
        At.fill_diagonal_(0.0)                                 # renormalisation trick: A~ = A + I (existing self loops are replaced)
        At = At + torch.eye(n, dtype=torch.float64)
        d = At.sum(1)
        Ahat = d.pow(-0.5)[:, None] * At * d.pow(-0.5)[None, :]
        want = torch.einsum("ij,bjc->bic", Ahat, x @ W1.t()) + b
    got = O.gnn_conv(st, "c", x, O.gnn_edges(ei, n, conv), conv)
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12), float((got - want).abs().max())
    assert float(want[:, -2:].abs().sum()) > 0 or conv == "GC"           # isolated nodes still get the root / bias / self-loop term
