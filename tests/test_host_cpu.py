"""CPU-only tests (no GPU, no compute calls into the HIP kernels): the C-ABI library loads and exports
every symbol include/fxhip.h declares, the host-side mirror of the reference interface (architecture
derivation, state_dict layout, dataset contract, model-class attribute surface, triplet sampling rules)
and the multi-process trial sharding on the gloo backend (world_size 2)."""
import copy
import os
import pickle
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_io import Golden, MODEL_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    import ctypes
    from flexynesis_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fxhip.h")).read()
    declared = set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fxhip.h but not exported by libfxhip.so"
    assert _lib.lib.fx_version() >= 100
    # argument validation happens before any launch, so it is testable without a GPU
    assert _lib.lib.fx_gemm_f32(7, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, None, 0, None) == -22
    assert b"layout" in _lib.lib.fx_last_error_string()
    assert _lib.lib.fx_cox_ph(None, None, None, None, None, 5, 1, 1, None, 1.0, None) == -22
    assert _lib.lib.fx_gemm_workspace_bytes(128, 5000, 20000) > 0
    assert _lib.lib.fx_gemm_workspace_bytes(5000, 20000, 128) == 0


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "flexynesis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src, f


@pytest.mark.parametrize("case", MODEL_CASES)
def test_arch_state_layout_matches_reference_state_dict(case):
    from flexynesis_amd.arch import ArchSpec
    g = Golden(case)
    s = g.spec
    a = ArchSpec(s.model, list(s.layers), s.latent_dim, s.hidden_dim_factor, s.supervisor_hidden_dim,
                 list(s.variables), s.surv_event_var, s.surv_time_var, s.use_loss_weighting,
                    s.input_layers, s.output_layers)
    ref = {k: tuple(v.shape) for k, v in g.state0().items()}
    assert a.state_shapes() == ref
    assert a.loss_names() == [k for k in _loss_order(g)]


def _loss_order(g):
    names = [k for k in g.exp(0, "loss") if k != "total"]
    spec = g.spec
    return spec.loss_names() if set(names) == set(spec.loss_names()) else names


def _toy_dataset(n=24, missing_label=False):
    from flexynesis_amd.data import MultiOmicDataset
    g = torch.Generator().manual_seed(0)
    dat = {"gex": torch.randn(n, 40, generator=g), "cnv": torch.randn(n, 24, generator=g)}
    c = (torch.arange(n) % 3).float()
    if missing_label:
        c[5] = float("nan")
    ann = {"y": torch.randn(n, generator=g), "c": c, "event": (torch.rand(n, generator=g) < 0.5).float(),
           "time": torch.rand(n, generator=g) * 5}
    vt = {"y": "numerical", "c": "categorical", "event": "numerical", "time": "numerical"}
    feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    return MultiOmicDataset(dat, ann, vt, feats, [f"s{i}" for i in range(n)], {"c": {0: "a", 1: "b", 2: "c"}})


def test_dataset_contract():
    ds = _toy_dataset()
    dat, ann, sid = ds[3]
    assert set(dat) == {"gex", "cnv"} and dat["gex"].shape == (40,) and sid == "s3"
    assert set(ann) == {"y", "c", "event", "time"}
    sub = ds.subset([1, 5, 7])
    assert len(sub) == 3 and sub.samples == ["s1", "s5", "s7"] and torch.equal(sub.dat["cnv"][1], ds.dat["cnv"][5])
    assert ds.get_dataset_stats()["sample_count"] == 24
    from torch.utils.data import DataLoader
    b = next(iter(DataLoader(ds, batch_size=8)))
    assert b[0]["gex"].shape == (8, 40) and b[1]["y"].shape == (8,) and len(b[2]) == 8


def test_triplet_dataset_index_semantics_match_reference_golden():
    """label -> indices map incl. the "NA" group, valid anchors (reference data.py:1102-1151; golden from
    the reference's own TripletMultiOmicDataset)."""
    from flexynesis_amd.data import MultiOmicDataset, TripletMultiOmicDataset
    g = Golden("functions")
    lab = g.get("tripletds/labels")
    ds = MultiOmicDataset({"gex": torch.randn(len(lab), 4)}, {"c": lab}, {"c": "categorical"},
                          {"gex": list("abcd")}, [f"s{i}" for i in range(len(lab))], {})
    t = TripletMultiOmicDataset(ds, "c")
    assert t.valid_indices == g.get("tripletds/valid_indices").tolist()
    ref = g.sub("tripletds/idx")
    assert {str(k) for k in t.label_to_indices} == set(ref)
    for k, idx in t.label_to_indices.items():
        assert idx.tolist() == ref[str(k)].tolist()
    np.random.seed(0)
    import random
    random.seed(0)
    for i in range(len(t)):
        a, p, n = t.sample_indices(i)
        la, lp, ln = float(lab[a]), float(lab[p]), float(lab[n])
        assert a != p and la == lp and (np.isnan(ln) or ln != la)
        anchor, pos, neg, y = t[i]
        assert set(anchor) == {"gex"} and "c" in y


@pytest.mark.parametrize("name,targets", [("DirectPred", ["y", "c"]), ("supervised_vae", ["c"]),
                                          ("MultiTripletNetwork", ["c", "y"])])
def test_model_class_api_surface(name, targets):
    import flexynesis_amd.models as M
    cls = getattr(M, name)
    ds = _toy_dataset(missing_label=True)
    cfg = {"latent_dim": 8, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 4, "epochs": 2,
           "batch_size": 8}
    m = cls(cfg, ds, targets, surv_event_var="event", surv_time_var="time")
    assert cls.__name__ == name
    assert m.target_variables == targets + ["event"] and m.variables == m.target_variables
    assert m.config is cfg and m.surv_event_var == "event" and m.feature_importances == {}
    assert m.layers == ["gex", "cnv"] and m.input_dims == [40, 24]
    assert len(list(m.encoders)) == 2 and set(m.MLPs.keys()) == set(m.variables)
    # NaN counts as a class in len(np.unique(ann)) (reference direct_pred.py:100)
    assert m.MLPs["c"].layer_out.weight.shape[0] == 4
    assert m.MLPs["event"].layer_out.bias is None and m.MLPs["event"].layer_out.weight.shape == (1, 4)
    if name == "MultiTripletNetwork":
        assert m.main_var == "c" and "log_vars.triplet_loss" in m.state_dict()
    if name == "supervised_vae":
        assert "log_vars.mmd_loss" in m.state_dict() and len(m.decoders) == 2
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == m.spec.state_shapes()
    m2 = copy.deepcopy(m)
    m3 = pickle.loads(pickle.dumps(m))
    for other in (m2, m3):
        for k, v in other.state_dict().items():
            assert torch.equal(v, sd[k])
    m2.load_state_dict(sd)
    opt = m.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-3
    # freezing as FineTuner does (reference main.py:532-539)
    for p in m.encoders.parameters():
        p.requires_grad = False
    assert not any(p.requires_grad for p in m.encoders.parameters())
    with pytest.raises(RuntimeError):        # no CPU execution path: fails loudly instead of falling back
        m.training_step(({"gex": ds.dat["gex"][:8], "cnv": ds.dat["cnv"][:8]}, {k: v[:8] for k, v in ds.ann.items()},
                         ds.samples[:8]), 0) if name != "MultiTripletNetwork" else m._bind()


def test_triplet_requires_categorical_main_variable():
    import flexynesis_amd.models as M
    cfg = {"latent_dim": 8, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 4, "epochs": 2, "batch_size": 8}
    with pytest.raises(ValueError):
        M.MultiTripletNetwork(cfg, _toy_dataset(), ["y"])


def test_split_and_search_space_are_deterministic():
    from flexynesis_amd.fit import split_indices
    from flexynesis_amd.trials import assign_trials, draw_search_space, trial_cost
    tr, va = split_indices(2048, 0.2, 7)
    assert len(va) == 409 and len(tr) == 1639 and sorted(tr + va) == list(range(2048))
    assert split_indices(2048, 0.2, 7) == (tr, va)
    ps = draw_search_space(64, seed=0)
    assert ps == draw_search_space(64, seed=0) and len(ps) == 64
    assert all(16 <= p["latent_dim"] <= 128 and 0.2 <= p["hidden_dim_factor"] <= 0.5 and p["batch_size"] in (32, 64, 128)
               and 1e-4 <= p["lr"] <= 1e-2 for p in ps)
    costs = [trial_cost(p, 40000, 1639) for p in ps]
    a = assign_trials(costs, 8)
    assert sorted(sum(a, [])) == list(range(64))
    loads = [sum(costs[i] for i in r) for r in a]
    assert max(loads) / (sum(loads) / 8) < 1.15          # LPT keeps the tail imbalance small


# ---- multi-process (gloo, world_size 2) -----------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flexynesis_amd import trials
    dat = ann = None
    if rank == 0:
        g = torch.Generator().manual_seed(1)
        dat = {"zeta": torch.randn(16, 6, generator=g), "alpha": torch.randn(16, 3, generator=g)}   # non-sorted order
        ann = {"y": torch.randn(16, generator=g)}
    dat, ann = trials.broadcast_cohort(dat, ann, "cpu")
    params = trials.draw_search_space(7, seed=3)
    shapes = {"w": (2, 3), "bn.num_batches_tracked": ()}

    def trial_fn(tid, p):
        if tid == 4:
            raise RuntimeError("boom")            # a failing trial must not hang the gather
        if tid == 5:
            return float("nan"), 1, None           # non-finite -> +inf
        val = 10.0 - tid if tid != 2 else 0.5      # trial 2 wins
        state = {"w": torch.full((2, 3), float(tid)), "bn.num_batches_tracked": torch.tensor(tid * 3)}
        return val, p["epochs"], state

    # the winner's layout is derived from ITS parameters on every rank (HPO trials differ in architecture)
    table, best, state = trials.run_sweep(params, trial_fn, costs=[1.0] * 7, device="cpu",
                                          state_shapes=lambda p: dict(shapes) if "epochs" in p else None)
    # a winner whose owner holds no weights must degrade to best_state=None on EVERY rank (never a hang in the broadcast)
    def stateless(tid, p):
        return (0.25 if tid == 3 else 5.0 + tid), 1, (None if tid == 3 else {"w": torch.zeros(2, 3), "bn.num_batches_tracked": torch.tensor(0)})
    _, best2, state2 = trials.run_sweep(params, stateless, costs=[1.0] * 7, device="cpu", state_shapes=shapes)
    # ranks that disagree about the sweep (here: its costs) fail TOGETHER before any unit runs -- no silently split queue
    try:
        trials.run_units(3, lambda u: (1.0, 1, None), costs=[1.0, 2.0, 3.0 + rank], device="cpu")
        mismatch = "no error"
    except RuntimeError as e:
        mismatch = str(e)
    # ... and the next, consistent sweep works again (fresh queue id from rank 0)
    t3, _ = trials.run_units(4, lambda u: (float(u), 1, None), costs=[4.0, 3.0, 2.0, 1.0], device="cpu")
    out.put((rank, list(dat.keys()), float(dat["zeta"].sum()), table.tolist(), best,
             state["w"].tolist(), int(state["bn.num_batches_tracked"]), best2, state2 is None, mismatch, t3[:, 1].tolist(),
             sorted(set(t3[:, 4].tolist()))))
    dist.destroy_process_group()


def test_trial_sharding_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1] == r1[1] == ["zeta", "alpha"]            # rank 0's layer ORDER, not a re-sorted one
    assert r0[2] == r1[2]                                  # same cohort everywhere
    assert r0[3] == r1[3]                                  # identical result table on every rank
    table = np.array(r0[3])
    assert table[:, 0].tolist() == list(range(7))
    assert np.isinf(table[4, 1]) and table[4, 3] == 1.0 and np.isinf(table[5, 1])
    assert r0[4] == r1[4] == 2 and table[2, 1] == 0.5
    assert r0[5] == r1[5] == [[2.0] * 3] * 2 and r0[6] == r1[6] == 6   # winner's weights reached both ranks
    assert r0[7] == r1[7] == 3 and r0[8] and r1[8]                      # stateless winner: None everywhere, no hang
    assert "do not agree" in r0[9] and "do not agree" in r1[9]
    assert r0[10] == r1[10] == [0.0, 1.0, 2.0, 3.0] and set(r0[11]) <= {0.0, 1.0}


def test_kfold_indices_are_a_partition_with_sklearn_fold_sizes():
    from flexynesis_amd.fit import kfold_indices
    folds = kfold_indices(23, 5, seed=3)
    assert [len(v) for _, v in folds] == [5, 5, 5, 4, 4]            # sklearn KFold: first n % k folds get one extra
    allv = sorted(i for _, v in folds for i in v)
    assert allv == list(range(23))
    for tr, va in folds:
        assert sorted(tr + va) == list(range(23)) and not set(tr) & set(va)
    assert kfold_indices(23, 5, seed=3) == folds and kfold_indices(23, 5, seed=4) != folds


def test_gat_is_refused_at_construction_with_the_reason():
    """flexGCN's table lists GATConv (reference modules.py:221-226) but its forward hands every conv the batched
    [B, nodes, C] tensor (modules.py:251-262), which torch_geometric's GATConv rejects ("Static graphs not supported"), and the
    CLI offers GC / GCN / SAGE only (__main__.py:536-540): the engine refuses the choice up front and says why."""
    from flexynesis_amd.arch import GNN_CONVS, gnn_conv_keys
    assert GNN_CONVS == ("GC", "SAGE", "GCN")
    with pytest.raises(ValueError, match="GATConv does not accept the batched"):
        gnn_conv_keys("encoders.0.convs.0", "GAT")
    with pytest.raises(ValueError, match="Unknown convolution type"):
        gnn_conv_keys("encoders.0.convs.0", "GIN")


@pytest.mark.parametrize("conv", ["GC", "SAGE", "GCN"])
def test_graph_operator_matches_oracle_edge_weights(conv):
    """flexynesis_amd/graph.py (host side of the GNN path) against the oracle's restatement of the torch_geometric
    aggregation weights: duplicates, self loops, isolated nodes, a hub."""
    from flexynesis_amd import graph as G
    from oracle import restate as O
    rng = np.random.default_rng(3)
    n = 37
    ei = rng.integers(0, n - 3, size=(2, 160))            # nodes n-3 .. n-1 are isolated
    ei[1, :30] = 5
    ei[:, 40:44] = ei[:, :4]
    ei[0, 50:55] = ei[1, 50:55]
    src, dst, w = G.edge_weights(ei, n, conv)
    osrc, odst, ow = O.gnn_edges(torch.from_numpy(ei), n, conv)
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), w)
    Ao = np.zeros((n, n))
    np.add.at(Ao, (odst.numpy(), osrc.numpy()), ow.numpy())
    np.testing.assert_allclose(A, Ao, rtol=1e-12, atol=1e-15)
    if conv == "SAGE":
        rows = A.sum(1)
        assert np.allclose(rows[rows > 0], 1.0)               # a mean over the in-neighbours
    if conv == "GCN":
        assert (np.diag(A) > 0).all()                         # every node has its self loop


def test_graph_csr_build_needs_no_gpu_for_weights():
    from flexynesis_amd import graph as G
    src, dst, w = G.edge_weights(np.array([[0, 1, 1], [1, 0, 1]]), 3, "GCN")
    assert sorted(zip(src.tolist(), dst.tolist())) == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 2)]
    assert w[(src == 2) & (dst == 2)][0] == 1.0               # isolated node: degree 1 from its self loop


# ---- device triplet sampler (fit.TripletSampler; reference data.py:1106-1131) ------------------------------------
def _triplet_labels():
    # groups: 0 -> 3 members, 1 -> 2, 2 -> 4, NaN ("NA" group) -> 2; scattered positions
    return torch.tensor([2., 0., float("nan"), 1., 2., 0., 2., float("nan"), 1., 0., 2.])


def test_triplet_sampler_semantics():
    """positive = another sample with the anchor's label (never the anchor); negative = member of another label group,
    the NaN samples forming one extra group (reference data.py:1118-1127); anchors = samples with a label (:1102-1104)."""
    from flexynesis_amd.fit import TripletSampler
    lab = _triplet_labels()
    sm = TripletSampler(lab)
    assert sm.valid.tolist() == [i for i, v in enumerate(lab.tolist()) if v == v]
    assert sm.n_groups == 4
    gen = torch.Generator().manual_seed(0)
    anchors = sm.valid.repeat(4000)
    pos, neg = sm.sample(anchors, gen)
    la, lp, ln = lab[anchors], lab[pos], lab[neg]
    assert bool((pos != anchors).all()) and bool((lp == la).all())
    assert bool(((ln != la) | torch.isnan(ln)).all())
    assert bool(torch.isnan(ln).any()), "the NA group must be drawn as a negative label"

    # uniformity: every admissible positive / negative label / member of the drawn group equally likely (4 sigma)
    def check(counts, p, what):
        n = int(counts.sum())
        sigma = (p * (1 - p) / n) ** 0.5
        assert bool(((counts / n - p).abs() <= 4 * sigma + 1e-9).all()), (what, (counts / n).tolist(), p)

    members = {0.0: [1, 5, 9], 1.0: [3, 8], 2.0: [0, 4, 6, 10], "NA": [2, 7]}
    for a in sm.valid.tolist():
        sel = anchors == a
        key = float(lab[a])
        others = [i for i in members[key] if i != a]
        check(torch.bincount(pos[sel], minlength=11)[others].double(), 1.0 / len(others), f"positives of {a}")
        neg_groups = [k for k in members if k != key]
        gcount = torch.tensor([float(sum(int((neg[sel] == i).sum()) for i in members[k])) for k in neg_groups])
        check(gcount, 1.0 / len(neg_groups), f"negative label of {a}")
        for k in neg_groups:
            check(torch.bincount(neg[sel], minlength=11)[members[k]].double() / 1.0, gcount[neg_groups.index(k)].item() / int(sel.sum()) / len(members[k]),
                  f"negatives of {a} inside group {k}") if False else None
            inside = torch.bincount(neg[sel], minlength=11)[members[k]].double()
            check(inside, 1.0 / len(members[k]), f"negatives of {a} inside group {k}")


def test_triplet_sampler_rejects_what_the_reference_cannot_sample():
    from flexynesis_amd.fit import TripletSampler
    gen = torch.Generator().manual_seed(0)
    with pytest.raises(ValueError):          # one label group only: random.choice(list(set())) raises in the reference
        TripletSampler(torch.tensor([1., 1., 1.]))
    sm = TripletSampler(torch.tensor([0., 0., 1., 2., 2.]))
    with pytest.raises(ValueError):          # a class with a single member: the reference's rejection loop never ends
        sm.sample(torch.tensor([2]), gen)
    sm.sample(torch.tensor([0, 1, 3, 4]), gen)
    # NaN-only "other group": two groups = one label + NA is a legal configuration
    sm = TripletSampler(torch.tensor([3., 3., float("nan")]))
    pos, neg = sm.sample(torch.tensor([0, 1, 0, 1]), gen)
    assert pos.tolist() == [1, 0, 1, 0] and neg.tolist() == [2, 2, 2, 2]


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="reference only exists in the build container")
def test_triplet_sampler_matches_the_reference_dataset_distribution():
    """Live: the reference's TripletMultiOmicDataset.__getitem__ (numpy / random RNG) and the device sampler draw from the
    same distribution -- per anchor, the empirical frequencies of every (positive, negative) index agree within 5 sigma."""
    import random
    from oracle import ref_shim
    from flexynesis_amd.fit import TripletSampler
    R = ref_shim.load()
    lab = _triplet_labels()
    n = lab.numel()
    dat = {"a": torch.arange(n, dtype=torch.float32).reshape(n, 1)}           # the feature IS the sample index
    ds = R.MultiOmicDataset(dat, {"c": lab}, {"c": "categorical"}, {"a": ["f"]}, [f"s{i}" for i in range(n)], {})
    tds = R.TripletMultiOmicDataset(ds, "c")
    sm = TripletSampler(lab)
    assert tds.valid_indices == sm.valid.tolist()
    np.random.seed(0)
    random.seed(0)
    reps = 3000
    gen = torch.Generator().manual_seed(1)
    for ai, a in enumerate(tds.valid_indices):
        ref_pos, ref_neg = torch.zeros(n), torch.zeros(n)
        for _ in range(reps):
            anc, p, ng, y = tds[ai]
            assert int(anc["a"].item()) == a
            ref_pos[int(p["a"].item())] += 1
            ref_neg[int(ng["a"].item())] += 1
        p2, n2 = sm.sample(torch.full((reps,), a), gen)
        got_pos, got_neg = torch.bincount(p2, minlength=n).float(), torch.bincount(n2, minlength=n).float()
        for ref, got in ((ref_pos, got_pos), (ref_neg, got_neg)):
            assert bool(((ref > 0) == (got > 0)).all()), (a, ref.tolist(), got.tolist())       # same support
            pr = (ref + got) / (2 * reps)
            sigma = (pr * (1 - pr) * 2 / reps).sqrt()
            assert bool((((ref - got) / reps).abs() <= 5 * sigma + 1e-9).all()), (a, ref.tolist(), got.tolist())


def test_library_carries_the_hash_of_its_sources():
    """csrc/build.py rebuilds on a SOURCE-HASH mismatch, not on file times: the shipped libfxhip.so must have been built from
    exactly the sources in the tree."""
    from flexynesis_amd import _lib
    from flexynesis_amd.csrc import build
    assert not build.needs_build()
    assert _lib.lib.fx_source_hash().decode() == build.source_hash() == build.built_hash()


def test_model_builds_from_the_inference_namespace():
    """reference inference.py:116-122 rebuilds a model from saved artefacts with a SimpleNamespace in place of the dataset:
    ``dat`` values are None, ``ann[var]`` is the category LIST (or a dummy array for numerical targets)."""
    from types import SimpleNamespace
    import flexynesis_amd.models as M
    ns = SimpleNamespace(layers=["gex", "cnv"], features={"gex": [f"g{i}" for i in range(40)], "cnv": [f"c{i}" for i in range(24)]},
                         dat={"gex": None, "cnv": None}, variable_types={"c": "categorical", "y": "numerical"},
                         ann={"c": ["A", "B", "C"], "y": np.array([0.0])})
    cfg = {"latent_dim": 6, "hidden_dim_factor": 0.5, "lr": 1e-3, "supervisor_hidden_dim": 4, "epochs": 1, "batch_size": 8}
    for cls in (M.DirectPred, M.supervised_vae):
        m = cls(cfg, ns, ["y", "c"], device_type="cpu")
        sd = m.state_dict()
        assert sd["MLPs.c.layer_out.weight"].shape == (3, 4) and sd["MLPs.y.layer_out.weight"].shape == (1, 4)
        assert m.layers == ["gex", "cnv"] and m.input_dims == [40, 24]
        first = "encoders.0.layer_1.weight" if cls is M.DirectPred else "encoders.0.hidden_layers.0.weight"
        assert sd[first].shape == (20, 40)


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="reference only exists in the build container")
@pytest.mark.parametrize("name", ["DirectPred", "supervised_vae", "MultiTripletNetwork", "CrossModalPred"])
def test_state_dict_loads_into_the_reference_model(name):
    """The other direction of the checkpoint ABI: a state_dict produced by the engine's model class loads STRICTLY into
    the reference's class (same keys, shapes and dtypes, incl. the int64 num_batches_tracked buffers), so weights trained
    here can be handed to the reference's inference / attribution code."""
    import flexynesis_amd.models as M
    from oracle import ref_capture, ref_shim
    from oracle import restate as O
    from oracle.gen_goldens import make_cohort
    R = ref_shim.load()
    kw = dict(input_layers=["a"], output_layers=["b", "a"]) if name == "CrossModalPred" else {}
    spec = O.Spec(name, [("a", 30), ("b", 18)], 6, 0.4, 3, [("c", "categorical", 3), ("y", "numerical", 1)], **kw)
    dat, ann, vt = make_cohort(spec, 30, seed=2, missing=False)
    cfg = {"latent_dim": 6, "hidden_dim_factor": 0.4, "lr": 1e-3, "supervisor_hidden_dim": 3, "epochs": 1, "batch_size": 6}
    ref = ref_capture.build_reference_model(R, spec, ref_capture.make_dataset(R, dat, ann, vt), cfg)
    from flexynesis_amd.data import MultiOmicDataset
    ds = MultiOmicDataset(dat, ann, vt, {k: [f"{k}_{j}" for j in range(v.shape[1])] for k, v in dat.items()},
                          [f"s{i}" for i in range(30)], {})
    torch.manual_seed(5)
    mine = getattr(M, name)(cfg, ds, ["c", "y"], device_type="cpu", **kw)
    sd = mine.state_dict()
    sd["encoders.0." + ("layer_1" if name in ("DirectPred", "MultiTripletNetwork") else "hidden_layers.0") + ".weight"] += 0.25
    res = ref.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ref.state_dict().items():
        assert v.dtype == sd[k].dtype and torch.equal(v, sd[k]), k


def test_fused_level1_mode_is_chosen_only_under_a_suitable_trainer():
    """configure_optimizers() picks the fused FxAdam on its own only when a Lightning Trainer drives the model with one
    backward per step, full precision and norm clipping (it then clips through the configure_gradient_clipping hook);
    hand-written loops keep materialised gradients unless they opt in."""
    import types
    from flexynesis_amd.models.base import FxModel
    probe = types.SimpleNamespace(__dict__={}, automatic_optimization=True)
    f = FxModel._trainer_allows_fused
    assert f(probe) is False                                                       # no trainer
    ok = types.SimpleNamespace(accumulate_grad_batches=1, precision="32-true", gradient_clip_algorithm=None)
    for tr, want in ((ok, True),
                     (types.SimpleNamespace(accumulate_grad_batches=4, precision="32-true", gradient_clip_algorithm="norm"), False),
                     (types.SimpleNamespace(accumulate_grad_batches=1, precision="16-mixed", gradient_clip_algorithm="norm"), False),
                     (types.SimpleNamespace(accumulate_grad_batches=1, precision="bf16-mixed", gradient_clip_algorithm=None), False),
                     (types.SimpleNamespace(accumulate_grad_batches=1, precision=32, gradient_clip_algorithm="value"), False)):
        p = types.SimpleNamespace(automatic_optimization=True)
        p.__dict__["_trainer"] = tr
        assert f(p) is want, tr
    p = types.SimpleNamespace(automatic_optimization=False)
    p.__dict__["_trainer"] = ok
    assert f(p) is False


# ---- sharded cross-validation / fine-tuning protocol (gloo, world_size 2; the engine's fit() replaced by a stand-in) ----
def _fake_fit_factory(log):
    """Stand-in for flexynesis_amd.fit.fit: a deterministic 'training' whose outcome depends only on its arguments, so the
    sharded and the sequential drivers must agree exactly whatever rank runs which unit."""
    from flexynesis_amd.fit import FitResult

    def fake_fit(model, dataset, train_idx, val_idx=None, *, batch_size, epochs, lr, patience=0, seed=0, frozen=(), **kw):
        tag = (round(float(lr), 9), tuple(frozen), int(seed), len(train_idx), 0 if val_idx is None else len(val_idx))
        log.append(tag)
        import zlib
        h = zlib.crc32(repr(tag).encode()) % 1000            # (hash() is salted per process)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1e-3 * (1 + h % 7))          # "training" moves the weights in an argument-dependent way
        val = 0.1 + (h % 97) / 100.0 + (0.0 if frozen else 0.001)
        return FitResult(val_loss=val, epochs_run=int(epochs), stopped_epoch=(h % 4), history=[], steps=3)
    return fake_fit


def _toy_model():
    from flexynesis_amd import models as M
    cfg = {"latent_dim": 4, "hidden_dim_factor": 0.5, "lr": 1e-2, "supervisor_hidden_dim": 3, "batch_size": 8, "epochs": 2}
    torch.manual_seed(5)
    return M.DirectPred(cfg, _toy_dataset(), ["y"])


def _ft_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import flexynesis_amd.fit as F
    from flexynesis_amd import trials
    log = []
    F.fit = _fake_fit_factory(log)
    model = _toy_model()
    if rank == 1:                                   # rank 0's weights must win
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    res = {}
    for schedule in ("queue", "static"):
        final, best, results = F.fine_tune(model, _toy_dataset(), n_splits=3, batch_size=8, learning_rates=[1e-2, 1e-3],
                                           max_epoch=3, seed=2, sharded=True, schedule=schedule, comm_device="cpu")
        res[schedule] = (best, results, {k: v.double().sum().item() for k, v in final.state_dict().items()})
    # queue mode: every unit claimed exactly once across the ranks
    n_units, claimed = 11, []
    t, held = trials.run_units(n_units, lambda u: (float(u), 1, {"w": torch.full((2,), float(u))}), [float(1 + u % 3) for u in range(n_units)],
                               "cpu", keep=[4], schedule="queue")
    out.put((rank, res, len(log), t.tolist(), sorted(held.keys())))
    dist.destroy_process_group()


def test_sharded_fine_tune_matches_sequential_gloo_world2():
    """fine_tune(sharded=True) over two gloo ranks == the sequential driver: same per-configuration means, same best
    configuration, same final weights on both ranks (reference FineTuner.run_experiments, main.py:575-659), with the
    work-queue and with the static LPT schedule."""
    import flexynesis_amd.fit as F
    real_fit = F.fit
    log = []
    F.fit = _fake_fit_factory(log)
    try:
        final, best, results = F.fine_tune(_toy_model(), _toy_dataset(), n_splits=3, batch_size=8, learning_rates=[1e-2, 1e-3],
                                           max_epoch=3, seed=2)
    finally:
        F.fit = real_fit
    want = {k: v.double().sum().item() for k, v in final.state_dict().items()}
    n_fits = 2 * 3 * 3
    assert len(log) == n_fits + (1 if best["epochs"] > 0 else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ft_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for schedule in ("queue", "static"):
        for r in got:
            b, res, sums = r[1][schedule]
            assert b == best and res == results                         # identical records on every rank, equal to the sequential run
            for k in want:
                assert abs(sums[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), (schedule, k)
    # the 18 fits of each sharded run were split between the ranks (2 runs per worker; + the final fit on the owner of the last unit)
    total = got[0][2] + got[1][2]
    assert 2 * n_fits <= total <= 2 * (n_fits + 1) and min(got[0][2], got[1][2]) >= 4
    t0, t1 = np.array(got[0][3]), np.array(got[1][3])
    assert np.array_equal(t0, t1) and t0[:, 0].tolist() == list(range(11)) and (t0[:, 3] == 0).all()
    assert set(t0[:, 4].tolist()) <= {0.0, 1.0}
    owner = int(t0[4, 4])
    assert got[owner][4] == [4] and got[1 - owner][4] == []           # only the requested unit's state is kept, on the rank that ran it


def test_run_trial_cv_branch_and_full_train_follow_objective():
    """run_trial(use_cv=True) = objective()'s KFold branch (main.py:267-269, :327-333): one fit per fold on a new model,
    (mean of the folds' val losses, int(mean epochs), last model); full_train = objective(full_train=True)."""
    import flexynesis_amd.fit as F
    from flexynesis_amd import models as M
    real_fit = F.fit
    log = []
    F.fit = _fake_fit_factory(log)
    try:
        cfg = {"latent_dim": 4, "hidden_dim_factor": 0.5, "lr": 1e-2, "supervisor_hidden_dim": 3, "batch_size": 8, "epochs": 5}
        ds = _toy_dataset()
        val, ep, model, info = F.run_trial(M.DirectPred, cfg, ds, ["y"], seed=3, use_cv=True, n_splits=4, early_stop_patience=2)
        folds = F.kfold_indices(len(ds), 4, 3)
        assert [(t[3], t[4]) for t in log] == [(len(tr), len(va)) for tr, va in folds]
        assert len(info["fold_val_losses"]) == 4 and abs(val - float(np.mean(info["fold_val_losses"]))) < 1e-12
        # epochs per fold = stopped_epoch or max_epochs (main.py:319-322); the trial reports int(mean)
        assert ep == int(np.mean(info["fold_epochs"])) and all(e in (1, 2, 3, 5) for e in info["fold_epochs"])
        assert isinstance(model, M.DirectPred)
        seeds = [t[2] for t in log]
        assert len(set(seeds)) == 4 and seeds[0] == 3                      # every fold: a new model and new shuffles
        log.clear()
        v1, e1, _, _ = F.run_trial(M.DirectPred, cfg, ds, ["y"], seed=3, use_cv=False)
        assert len(log) == 1 and (log[0][3], log[0][4]) == tuple(len(x) for x in F.split_indices(len(ds), 0.2, 3))
        log.clear()
        m, info = F.full_train(M.DirectPred, dict(cfg, epochs=2), ds, ["y"], seed=4)
        assert len(log) == 1 and log[0][3] == len(ds) and log[0][4] == 0    # all samples, no validation split
    finally:
        F.fit = real_fit


def test_dropout_seed_stream_restarts_with_the_global_seed():
    """modules._next_seed: calling torch.manual_seed(s) again in the same process reproduces the dropout seeds."""
    from flexynesis_amd import modules
    torch.manual_seed(123)
    a = [modules._next_seed() for _ in range(4)]
    torch.manual_seed(123)
    b = [modules._next_seed() for _ in range(4)]
    torch.manual_seed(124)
    c = [modules._next_seed() for _ in range(4)]
    assert a == b and a != c and len(set(a)) == 4


def test_run_units_in_flight_threads_match_sequential():
    """run_units(in_flight=k): k host threads claim units from the same queue; the table and the held states equal the
    sequential run's whatever order the units finish in."""
    import time
    from flexynesis_amd import trials

    def unit(u):
        time.sleep(0.002 * ((u * 7) % 5))
        if u == 6:
            raise RuntimeError("a broken unit")
        return float((u * 37) % 11) + 0.25, 3 + u, {"w": torch.full((3,), float(u))}

    base_t, base_h = trials.run_units(13, unit, None, "cpu")
    for k in (2, 4):
        t, h = trials.run_units(13, unit, None, "cpu", in_flight=k)
        assert np.array_equal(t, base_t)
        assert sorted(h) == sorted(base_h) and all(torch.equal(h[u]["w"], base_h[u]["w"]) for u in h)
    t, h = trials.run_units(13, unit, None, "cpu", keep=[2, 9], in_flight=3)
    assert sorted(h) == [2, 9] and np.array_equal(t, base_t)
    assert base_t[6, 3] == trials.STATUS_FAILED and np.isinf(base_t[6, 1])


def test_library_has_no_packed_fp32_valu_ops():
    """The build turns the packed fp32 VALU instructions off (build.py FLAGS): v_pk_fma_f32 loses a term in lanes 48..63 when
    a wave of another kernel on the same SIMD streams ds_read_b128 data into MFMAs (scripts/pkfma_hazard_probe.hip).  Guard
    the flag: no v_pk_*_f32 in any code object of the shipped library."""
    import glob, shutil, subprocess, tempfile
    from flexynesis_amd.csrc import build as B
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    assert "-packed-fp32-ops" in B.FLAGS
    so = os.path.join(os.path.dirname(B.__file__), "libfxhip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(so, td)
        subprocess.run([objdump, "--offloading", "libfxhip.so"], cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = glob.glob(os.path.join(td, "libfxhip.so.*gfx950"))
        assert objs
        n_kernels = 0
        for o in objs:
            dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            n_kernels += dis.count("s_endpgm")
            bad = [ln for ln in dis.splitlines() if "v_pk_" in ln and "_f32" in ln]
            assert not bad, bad[:3]
        assert n_kernels > 50


def test_lease_returns_when_the_last_view_has_gone():
    """ops.LEASES (csrc/fx_runtime.hip): a range of long-lived memory handed out as a torch tensor comes back exactly when the LAST view
    of that tensor's storage is gone -- not when the object that took it is dropped (parameters that outlived their
    ParamStore as views of memory the next trial was given) -- with the events its owner attached.  Host memory here (DLPack device
    type 1); the GPU tests run the same path on arena ranges."""
    import gc
    from flexynesis_amd import ops
    base = torch.arange(64, dtype=torch.float32)
    back = []
    before = ops.LEASES.outstanding()
    t, lid = ops.LEASES.wrap(base.data_ptr() + 16 * 4, 32, "cpu", lambda evs, unsynced: back.append((list(evs), unsynced)))
    assert t.shape == (32,) and t.dtype == torch.float32 and t.data_ptr() == base.data_ptr() + 64
    assert torch.equal(t, base[16:48]) and ops.LEASES.outstanding() == before + 1
    w = t.view(4, 8)[:, :6]                  # what ParamStore hands out: a strided view (the "W" of a padded buffer)
    p = torch.nn.Parameter(torch.empty(0))
    p.data = w                               # ... and what a model's nn.Parameter keeps
    sd = {"w": p.detach()}                   # ... and a state_dict tensor
    t.mul_(2.0)
    assert float(base[16]) == 32.0           # the lease IS the memory, not a copy
    ops.LEASES.add_events(lid, ["ev0"], False)
    del t, w
    gc.collect()
    assert ops.LEASES.drain() == 0 and not back          # the parameter and the state_dict still view it
    del p
    assert ops.LEASES.drain() == 0 and not back
    del sd
    assert ops.LEASES.drain() == 1 and back == [(["ev0"], False)]
    assert ops.LEASES.outstanding() == before
    assert ops.LEASES.drain() == 0 and len(back) == 1    # exactly once
