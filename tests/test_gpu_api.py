"""GPU tests of the drop-in surface: model classes (reference LightningModule-style API), the nn.Module
blocks, fit()/run_trial(), hipGraph replay vs eager tapes.  Run with -m gpu on the MI355X box."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from golden_io import Golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dataset_from_golden(g, n_rows=None):
    from flexynesis_amd.data import MultiOmicDataset
    spec = g.spec
    cohort = g.sub("cohort")
    dat = {name: cohort[name] for name, _ in spec.layers}
    n = next(iter(dat.values())).shape[0]
    ann, vt = {}, {}
    gen = torch.Generator().manual_seed(0)
    for (v, kind, C) in spec.variables:
        if kind == "categorical":
            ann[v] = (torch.arange(n) % C).float()
            vt[v] = "categorical"
        else:
            ann[v] = torch.rand(n, generator=gen)
            vt[v] = "numerical"
    if spec.surv_time_var:
        ann[spec.surv_time_var] = torch.rand(n, generator=gen) * 10
        vt[spec.surv_time_var] = "numerical"
    feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    return MultiOmicDataset(dat, ann, vt, feats, [f"s{i}" for i in range(n)], {})


def _model_from_golden(g, cls):
    spec = g.spec
    ds = _dataset_from_golden(g)
    cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": g.lr,
           "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": 1, "batch_size": 8}
    targets = [v[0] for v in spec.variables if v[0] != spec.surv_event_var]
    extra = dict(input_layers=spec.input_layers, output_layers=spec.output_layers) if spec.model == "CrossModalPred" else {}
    m = cls(cfg, ds, targets, surv_event_var=spec.surv_event_var, surv_time_var=spec.surv_time_var,
            use_loss_weighting=spec.use_loss_weighting, device_type="cuda", **extra)
    return m, ds


def test_directpred_predict_transform_validation_match_reference_golden():
    from flexynesis_amd.models import DirectPred
    g = Golden("directpred_2omics_multitask")
    m, ds = _model_from_golden(g, DirectPred)
    m.load_state_dict(g.exp(g.n_steps - 1, "state"))
    m.to(DEV)
    emb = m.transform(ds)
    assert list(emb.columns) == [f"E{i}" for i in range(g.spec.latent_dim)] and list(emb.index) == ds.samples
    np.testing.assert_allclose(emb.values, g.get("exp/transform").numpy(), rtol=2e-4, atol=2e-5)
    pred = m.predict(ds)
    for k, v in g.sub("exp/predict").items():
        np.testing.assert_allclose(pred[k], v.numpy(), rtol=2e-4, atol=2e-5)
    b = g.batch(0)
    batch = ({n: x for (n, _), x in zip(g.spec.layers, b["x"])}, b["y"], tuple(f"s{i}" for i in range(8)))
    val = m.validation_step(batch, 0)
    ref = g.get("exp/val/loss/total")
    assert abs(float(val) - float(ref)) <= 1e-4 * abs(float(ref))
    assert m._logged["val_loss"] == pytest.approx(float(ref), rel=1e-4)


def test_svae_eval_forward_matches_golden_given_eps():
    """supervised_vae's predict/transform are stochastic in the reference (z is sampled in eval too); check the
    deterministic part through the plan with supplied eps."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    g = Golden("supervised_vae_2omics")
    s = g.spec
    a = ArchSpec(s.model, list(s.layers), s.latent_dim, s.hidden_dim_factor, s.supervisor_hidden_dim, list(s.variables),
                 s.surv_event_var, s.surv_time_var, s.use_loss_weighting)
    store = ParamStore(a, DEV, big_threshold=512)
    store.load_state(g.exp(g.n_steps - 1, "state"))
    cohort = g.sub("cohort")
    n = next(iter(cohort.values())).shape[0]
    plan = StepPlan(store, n, train=False, supplied_draws=True)
    plan.set_batch(x_list=[cohort[name].to(DEV) for name, _ in s.layers], y=None)
    for t in plan.y.values():
        t.fill_(float("nan"))
    plan.set_draws({"eps": g.get("draws/transform/eps").to(DEV)})
    plan.forward()
    np.testing.assert_allclose(plan.embeddings.cpu().numpy(), g.get("exp/transform").numpy(), rtol=2e-4, atol=2e-5)
    plan.set_draws({"eps": g.get("draws/predict/eps").to(DEV)})
    plan.forward()
    for k, v in g.sub("exp/predict").items():
        o = plan.buf[f"MLPs.{k}/out"]
        o = torch.softmax(o, 1) if dict((x[0], x[1]) for x in s.variables)[k] == "categorical" else o
        np.testing.assert_allclose(o.cpu().numpy(), v.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("name,case", [("DirectPred", "directpred_2omics_multitask"),
                                       ("supervised_vae", "supervised_vae_2omics"),
                                       ("CrossModalPred", "crossmodal_2in_2out")])
def test_training_step_backward_and_torch_adam_drop_in(name, case):
    """Lightning-style external loop: training_step -> loss.backward() -> clip_grad_norm_ -> torch Adam."""
    import flexynesis_amd.models as M
    g = Golden(case)
    m, ds = _model_from_golden(g, getattr(M, name))
    m.load_state_dict(g.state0())
    m.to(DEV)
    m.train()
    opt = m.configure_optimizers()
    b = g.batch(0)
    batch = ({n: x.to(DEV) for (n, _), x in zip(g.spec.layers, b["x"])}, {k: v.to(DEV) for k, v in b["y"].items()},
             tuple(f"s{i}" for i in range(8)))
    first = None
    for it in range(25):
        opt.zero_grad()
        loss = m.training_step(batch, it)
        assert loss.shape == ((1,) if m.spec.weighted else ())
        loss.backward()
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        first = float(loss) if first is None else first
    assert float(loss) < first, (first, float(loss))
    assert "train_loss" in m._logged
    sd = m.state_dict()
    assert int(sd["MLPs." + m.variables[0] + ".batchnorm.num_batches_tracked"]) == 25
    # frozen encoders (FineTuner, reference main.py:532-539) get no gradient
    for p in m.encoders.parameters():
        p.requires_grad = False
    opt.zero_grad()
    m.training_step(batch, 0).backward()
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in m.encoders.parameters())
    # deepcopy is independent of the original's arenas
    m2 = copy.deepcopy(m)
    w0 = m.state_dict()["MLPs." + m.variables[0] + ".layer_1.weight"].clone()
    m2.training_step(batch, 0)
    for p in m2.parameters():
        p.data.add_(1.0)
    assert torch.equal(m.state_dict()["MLPs." + m.variables[0] + ".layer_1.weight"], w0)


def test_blocks_match_torch_modules():
    """MLP / Encoder / Decoder forward+backward (HIP via autograd.Function) vs the same stack of stock torch
    layers on the GPU in fp64."""
    from flexynesis_amd.modules import MLP, Decoder, Encoder
    torch.manual_seed(0)
    B = 24
    x = torch.randn(B, 40, device=DEV)
    mask = (torch.rand(B, 10, device=DEV) < 0.9).float()
    mlp = MLP(40, 10, 5).to(DEV).train()
    xr = x.clone().requires_grad_(True)
    out = mlp(xr, dropout_mask=mask)
    out.sum().backward()
    ref = copy.deepcopy(mlp).double()
    for p in ref.parameters():
        p.grad = None
    ref.batchnorm.running_mean.zero_(); ref.batchnorm.running_var.fill_(1.0)
    xd = x.double().requires_grad_(True)
    h = ref.batchnorm(ref.layer_1(xd))
    o2 = ref.layer_out(torch.relu(h) * (mask.double() / 0.9))
    o2.sum().backward()
    assert torch.allclose(out.double(), o2, rtol=1e-4, atol=1e-5)
    assert torch.allclose(xr.grad.double(), xd.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(mlp.layer_1.weight.grad.double(), ref.layer_1.weight.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(mlp.batchnorm.running_var.double(), ref.batchnorm.running_var, rtol=1e-4, atol=1e-6)
    enc = Encoder(40, [12], 6).to(DEV).train()
    dec = Decoder(6, [12], 40).to(DEV).train()
    mean, logv = enc(x)
    xh = dec(mean)
    (xh.sum() + logv.sum()).backward()
    e64, d64 = copy.deepcopy(enc).double(), copy.deepcopy(dec).double()
    for mod in (e64, d64):
        for p in mod.parameters():
            p.grad = None
        for bn in [m_ for m_ in mod.modules() if isinstance(m_, nn.BatchNorm1d)]:
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
    hh = e64.hidden_layers(x.double())
    m2, v2 = e64.FC_mean(hh), e64.FC_var(hh)
    xh2 = torch.sigmoid(d64.FC_output(d64.hidden_layers(m2)))
    (xh2.sum() + v2.sum()).backward()
    assert torch.allclose(xh.double(), xh2, rtol=1e-4, atol=1e-5)
    assert torch.allclose(enc.hidden_layers[0].weight.grad.double(), e64.hidden_layers[0].weight.grad, rtol=2e-3, atol=1e-5)
    assert torch.allclose(dec.FC_output.weight.grad.double(), d64.FC_output.weight.grad, rtol=2e-3, atol=1e-5)
    with pytest.raises(RuntimeError):
        MLP(4, 4, 2)(torch.randn(3, 4))          # CPU tensors are refused, not silently computed elsewhere


def _synthetic_ds(n=600, F=(300, 200), seed=0):
    from flexynesis_amd.data import MultiOmicDataset
    g = torch.Generator().manual_seed(seed)
    dat = {"gex": torch.randn(n, F[0], generator=g), "cnv": torch.randn(n, F[1], generator=g)}
    w = torch.randn(F[0], generator=g) / F[0] ** 0.5
    y = dat["gex"] @ w + 0.05 * torch.randn(n, generator=g)
    c = (dat["cnv"][:, :3].sum(1) > 0).float() + (dat["gex"][:, 0] > 1).float()
    ann = {"y": y, "c": c, "event": (torch.rand(n, generator=g) < 0.6).float(), "time": torch.rand(n, generator=g) * 9}
    vt = {"y": "numerical", "c": "categorical", "event": "numerical", "time": "numerical"}
    feats = {k: [f"{k}{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    return MultiOmicDataset(dat, ann, vt, feats, [f"s{i}" for i in range(n)], {})


@pytest.mark.parametrize("name,targets,surv", [("DirectPred", ["y", "c"], True), ("supervised_vae", ["c"], False),
                                               ("MultiTripletNetwork", ["c", "y"], False)])
def test_fit_learns_and_graph_replay_equals_eager(name, targets, surv):
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import fit, split_indices
    ds = _synthetic_ds()
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.25, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 6,
           "batch_size": 64}
    kw = dict(surv_event_var="event", surv_time_var="time") if surv else {}
    tr, va = split_indices(len(ds), 0.2, 1)
    finals = []
    for use_graph in (True, False):
        torch.manual_seed(5)
        m = getattr(M, name)(cfg, ds, targets, device_type="cuda", **kw)
        res = fit(m, ds, tr, va, batch_size=64, epochs=6, lr=3e-3, patience=0, seed=11, use_graph=use_graph)
        assert res.epochs_run == 6 and res.steps == 6 * (len(tr) // 64)
        h = res.history
        assert h[-1]["train_loss"] < h[0]["train_loss"], h
        assert np.isfinite(res.val_loss)
        finals.append({k: v.clone() for k, v in m.state_dict().items()})
    for k in finals[0]:
        assert torch.allclose(finals[0][k].float(), finals[1][k].float(), rtol=1e-5, atol=1e-6), k


def test_early_stopping_and_run_trial():
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import run_trial
    ds = _synthetic_ds(n=400)
    ds.ann["y"] = torch.randn(400)                     # pure-noise target: validation loss cannot keep improving
    params = {"latent_dim": 16, "hidden_dim_factor": 0.3, "lr": 1e-2, "supervisor_hidden_dim": 8, "epochs": 60,
              "batch_size": 32}
    val, epochs, model, info = run_trial(M.DirectPred, params, ds, ["y"], early_stop_patience=3, seed=3, device="cuda")
    assert np.isfinite(val) and 0 < epochs < 60, (val, epochs)
    assert len(info["history"]) == epochs + 1          # stopped_epoch is 0-based (Lightning)
    # a trial that cannot run reports +inf instead of raising
    bad = dict(params, batch_size=100000)
    val2, _, _, info2 = run_trial(M.DirectPred, bad, ds, ["y"], seed=3, device="cuda")
    assert val2 == float("inf") and "error" in info2


@pytest.mark.parametrize("widths,B", [((1300, 1100), 32),
                                      ((2051, 1537), 100),       # odd widths: scalar gather path, no next-step fusion (k_in % 4 != 0); ragged batch
                                      ((4100, 2052), 64)])
@pytest.mark.parametrize("model_name", ["DirectPred", "MultiTripletNetwork"])
def test_pipelined_step_equals_plain_step(model_name, widths, B):
    """Double-buffered batch assembly (PipelinedStep: the batch of step t+1 is gathered during step t, hipGraph
    replay) walks the same index tables as the plain one-plan step and must produce the identical trajectory."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep, StepPlan
    dev = torch.device("cuda:0")
    layers = [("gex", widths[0]), ("cnv", widths[1])]
    trip = model_name == "MultiTripletNetwork"
    variables = [("c", "categorical", 4)] if trip else [("y", "numerical", 1)]
    spec = ArchSpec(model_name, layers, 32, 0.9, 16, variables, None, None, True)      # hidden >= 2^20/F: wide path
    cohort = synthetic_cohort(layers, 400, dev, seed=5)
    nb, steps, lr = 3, 8, 1e-3
    rows = B * (3 if trip else 1)
    g = torch.Generator().manual_seed(0)
    tables = [torch.randint(0, 400, (nb * rows,), generator=g).to(dev) for _ in range(steps // nb + 2)]
    torch.manual_seed(11)
    init = ParamStore(spec, dev).state_dict()

    def plain():
        store = ParamStore(spec, dev)
        store.load_state(init)
        plan = StepPlan(store, B, train=True, fused=True, seed=9, cohort=cohort, n_batches=nb, epoch_acc=True)
        store.ctrl[8] = -1.0            # fx_step_begin advances the cursor before the gather: row 0 first
        out = []
        for s in range(steps):
            if s % nb == 0:
                plan.idx.copy_(tables[s // nb])
            plan.train_step(lr, gather=True)
            out.append(plan.losses()["total"])
        return out, store.state_dict()

    def piped(graph, fuse):
        store = ParamStore(spec, dev)
        store.load_state(init)
        pipe = PipelinedStep(store, B, cohort=cohort, n_batches=nb, seed=9, fuse_next_fwd=fuse)
        # (more than one M-tile: the separate forward is faster; ANY feature count takes the fused kernel: the MLP family's input
        # widths are rounded up to 4 inside the engine, ArchSpec.engine_shapes)
        assert bool(pipe.plans[0]._next_fwd) == (fuse and rows <= 128)
        pipe.idx.copy_(tables[0])
        pipe.prime()
        out, e = [], 0
        for s in range(steps):
            if pipe.epoch_end_next():
                e += 1
                pipe.idx.copy_(tables[e])
            if graph and pipe.graphs[0] is not None:
                pipe.replay()
            else:
                pipe.step(lr)
                if graph:
                    pipe.capture(lr)
            out.append(pipe.losses()["total"])
        return out, store.state_dict()

    l0, s0 = plain()
    # separate forward kernel: the pipelined schedule is the plain step, bit for bit
    for graph in (False, True):
        l1, s1 = piped(graph, False)
        assert l1 == l0, (graph, l0, l1)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), (graph, k)
    # next-step forward fused into dW+Adam: another (fixed) summation order in the wide forward -> rounding-level
    # differences from the plain step, but graph replay and eager launches of the SAME schedule agree bit for bit
    lf, sf = piped(False, True)
    lg, sg = piped(True, True)
    assert lf == lg
    for k in sf:
        assert torch.equal(sf[k], sg[k]), k
    for a, b in zip(lf, l0):
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-6, (lf, l0)
    # Biases that reach the loss only through a BatchNorm have a true gradient of exactly zero: what any implementation
    # holds there is rounding noise that Adam turns into +-lr steps -- only bounded, not compared.
    noise = (".layer_1.bias", ".layer_out.bias", "fusion_block.bias", ".running_mean")     # (running means track those biases)
    for k in s0:
        if s0[k].dtype.is_floating_point:
            d = (sf[k].double() - s0[k].double()).abs()
            assert float(d.max()) <= 2.1 * lr * steps, k
            if not k.endswith(noise):
                assert float((d > 1e-5 + 1e-3 * s0[k].double().abs()).double().mean()) <= 5e-3, k


def test_crossmodal_validation_decode_and_fit():
    """CrossModalPred: validation_step vs the reference golden, decode() layout
    (reference crossmodal_pred.py:467-481) and the engine fit loop on separate input / output layer lists."""
    from flexynesis_amd.fit import fit
    from flexynesis_amd.models import CrossModalPred
    g = Golden("crossmodal_2in_2out")
    m, ds = _model_from_golden(g, CrossModalPred)
    assert m.input_layers == ["gex", "cnv"] and m.output_layers == ["meth", "gex"]
    assert len(m.encoders) == 2 and len(m.decoders) == 2
    m.load_state_dict(g.exp(g.n_steps - 1, "state"))
    m.to(DEV)
    dec = m.decode(ds)
    assert list(dec) == ["meth", "gex"]
    for l in dec:
        assert list(dec[l].index) == ds.features[l] and list(dec[l].columns) == ds.samples
        v = dec[l].values
        assert np.isfinite(v).all() and v.min() >= 0.0 and v.max() <= 1.0          # Decoder ends in a sigmoid
    pred = m.predict(ds)
    assert set(pred) == {"y", "c"} and pred["c"].shape == (len(ds), 3)
    np.testing.assert_allclose(pred["c"].sum(1), 1.0, rtol=1e-5)
    m.load_state_dict(g.state0())
    n = len(ds)
    res = fit(m, ds, list(range(0, n - 8)), list(range(n - 8, n)), batch_size=8, epochs=6, lr=3e-3, seed=1, device="cuda")
    assert res.epochs_run == 6 and np.isfinite(res.val_loss)
    assert res.history[-1]["train_loss"] < res.history[0]["train_loss"]
    assert set(res.history[0]) >= {"mmd_loss", "y", "c", "train_loss", "val_loss"}


def test_fit_keeps_partial_batch_and_fine_tune_runs():
    """fit(drop_last=False) uses the partial last batch like the FineTuner's DataLoader (reference main.py:541-545);
    fine_tune = FineTuner.run_experiments (main.py:575-659) on the engine."""
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import fit, fine_tune
    ds = _synthetic_ds(n=75)
    torch.manual_seed(0)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 16}
    m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    res = fit(m, ds, list(range(60)), list(range(60, 75)), batch_size=16, epochs=3, lr=3e-3, seed=2, device="cuda",
              clip=False, frozen=("encoders.",), drop_last=False, fresh_optimizer=True)
    assert res.steps == 3 * (60 // 16 + 1)                      # 3 full batches + the 12-sample tail, per epoch
    after = m.state_dict()
    for k in before:
        if k.startswith("encoders.") and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert torch.equal(before[k].to(after[k].device), after[k]), k
    assert not torch.equal(before["MLPs.y.layer_1.weight"].to(DEV), after["MLPs.y.layer_1.weight"])
    assert not torch.equal(before["encoders.0.batchnorm.running_mean"].to(DEV), after["encoders.0.batchnorm.running_mean"])
    assert int(after["encoders.0.batchnorm.num_batches_tracked"]) == res.steps
    final, best, results = fine_tune(m, ds, n_splits=2, batch_size=16, learning_rates=[3e-3, 3e-4], max_epoch=4,
                                     freeze_configs=[{"encoders": True, "supervisors": False},
                                                     {"encoders": False, "supervisors": True}], seed=1, device="cuda")
    assert len(results) == 4 and best in results and all(np.isfinite(r["average_val_loss"]) for r in results)
    assert set(results[0]) == {"learning_rate", "average_val_loss", "freeze", "epochs"}
    assert final is not m and set(final.state_dict()) == set(m.state_dict())
    assert set(final.predict(ds)) == {"y", "c"}


def test_pipelined_graph_replay_is_bit_reproducible():
    """Race detector for the branch-parallel, double-buffered hipGraphs: two runs from the same state, tables and
    seeds give bit-identical parameters, moments and losses (scripts/soak_determinism.py does this at cfg2-4 scale)."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    dev = torch.device("cuda:0")
    layers = [("gex", 2600), ("cnv", 2100), ("meth", 1500)]
    spec = ArchSpec("supervised_vae", layers, 32, 0.5, 16, [("c", "categorical", 4), ("event", "numerical", 1)], "event", "time", True)
    cohort = synthetic_cohort(layers, 500, dev, seed=1)
    torch.manual_seed(0)
    init = ParamStore(spec, dev, materialize_big_grads=False).state_dict()

    def run():
        store = ParamStore(spec, dev, materialize_big_grads=False)
        store.load_state(init)
        pipe = PipelinedStep(store, 64, cohort=cohort, n_batches=5, seed=3)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        pipe.idx.copy_(torch.randint(0, 500, (5 * 64,), generator=g, device=dev))
        pipe.prime()
        pipe.step(1e-3)
        pipe.capture(1e-3)
        curve = []
        for _ in range(40):
            if pipe.epoch_end_next():
                pipe.idx.copy_(torch.randint(0, 500, (5 * 64,), generator=g, device=dev))
            pipe.replay()
            curve.append(pipe.last.loss_vec.clone())
        return torch.stack(curve).cpu(), store.state_dict()

    c1, s1 = run()
    c2, s2 = run()
    assert torch.isfinite(c1).all() and torch.equal(c1, c2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


@pytest.mark.parametrize("name,targets,frozen", [("MultiTripletNetwork", ["c", "y"], ("encoders.",)), ("MultiTripletNetwork", ["c", "y"], ("MLPs.",)),
                                                 ("supervised_vae", ["c"], ("encoders.",)), ("supervised_vae", ["c"], ("MLPs.",))])
def test_frozen_groups_fit_every_model_family(name, targets, frozen):
    """FineTuner freeze configurations (reference main.py:530-539) on the triplet and VAE schedules: frozen groups stay
    bit-identical, the rest trains, BatchNorm buffers of frozen blocks still move."""
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import fit
    ds = _synthetic_ds(n=260)
    torch.manual_seed(3)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.25, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 2, "batch_size": 32}
    m = getattr(M, name)(cfg, ds, targets, device_type="cuda")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    res = fit(m, ds, list(range(200)), list(range(200, 260)), batch_size=32, epochs=2, lr=3e-3, seed=5, device="cuda",
              clip=False, frozen=frozen, drop_last=False, fresh_optimizer=True)
    assert np.isfinite(res.val_loss) and res.steps == 2 * 7
    after = m.state_dict()
    moved = [k for k in before if not torch.equal(before[k].to(after[k].device), after[k])]
    for k in before:
        if k.startswith(frozen) and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert k not in moved, k
    assert any(k.startswith(frozen) and k.endswith("running_mean") for k in moved)
    assert any(not k.startswith(frozen) and k.endswith(".weight") for k in moved)


def test_run_trial_crossmodal_takes_layer_lists():
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import run_trial
    ds = _synthetic_ds(n=200)
    params = {"latent_dim": 16, "hidden_dim_factor": 0.3, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 32}
    val, epochs, model, info = run_trial(M.CrossModalPred, params, ds, ["y"], early_stop_patience=0, seed=2, device="cuda",
                                         input_layers=["cnv"], output_layers=["gex"])
    assert "error" not in info and np.isfinite(val) and epochs == 3
    assert model.input_layers == ["cnv"] and model.output_layers == ["gex"] and len(model.encoders) == 1


def test_drop_in_fx_adam_equals_torch_adam_and_masks_change():
    """Level-1 drop-in (Lightning protocol): configure_optimizers() returns FxAdam, whose step() runs fx_adam_flat on the
    arenas that param.grad already views; it must follow torch.optim.Adam on the same gradients.  The dropout masks of
    training_step are keyed on a step counter that training_step itself advances: they change from step to step with
    either optimiser (an external optimiser never touches the engine's control block)."""
    import flexynesis_amd.models as M
    from flexynesis_amd.models.base import FxAdam
    g = Golden("directpred_2omics_multitask")
    m, ds = _model_from_golden(g, M.DirectPred)
    m.load_state_dict(g.state0())
    m.to(DEV)
    ma, mb = copy.deepcopy(m), copy.deepcopy(m)
    oa, ob = ma.configure_optimizers(), torch.optim.Adam(mb.parameters(), lr=ma.config["lr"])
    assert isinstance(oa, FxAdam) and isinstance(oa, torch.optim.Optimizer)
    b = g.batch(0)
    batch = ({n: x.to(DEV) for (n, _), x in zip(g.spec.layers, b["x"])}, {k: v.to(DEV) for k, v in b["y"].items()},
             tuple(f"s{i}" for i in range(8)))
    patterns = {0: [], 1: []}
    for it in range(6):
        for j, (mm, oo) in enumerate(((ma, oa), (mb, ob))):
            mm.train()
            oo.zero_grad()
            loss = mm.training_step(batch, it)
            loss.backward()
            plan = mm._plans[(8, True, False)]
            patterns[j].append((plan.buf["encoders.0/a1"] == 0).clone())
            # param.grad is the arena itself: no copy was made on the way to autograd
            pk = "encoders.0.layer_out.weight"
            assert dict(mm.named_parameters())[pk].grad.data_ptr() == mm._store.g(pk).data_ptr()
            torch.nn.utils.clip_grad_norm_(mm.parameters(), 1.0)
            oo.step()
    for j in (0, 1):
        assert not torch.equal(patterns[j][0], patterns[j][1]) and not torch.equal(patterns[j][1], patterns[j][2])
        assert torch.equal(patterns[0][j], patterns[1][j])            # same seeds, same counters: both models drew the same masks
    sa, sb = ma.state_dict(), mb.state_dict()
    noise = (".layer_1.bias", ".layer_out.bias", "fusion_block.bias", ".running_mean")   # zero-gradient biases: Adam on rounding noise
    for k in sa:
        if sa[k].dtype.is_floating_point and not k.endswith(noise):
            assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 + 1e-5 * float(sb[k].abs().max()), k
    assert float((sa["encoders.0.layer_1.weight"] - g.state0()["encoders.0.layer_1.weight"].to(DEV)).abs().max()) > 1e-4


def test_level1_fused_optimizer_follows_materialised_path():
    """Level-1 fast path: model.fused_optimizer = True makes configure_optimizers() return a fused FxAdam -- backward forms
    no wide gradients (their .grad stays None), the trainer's gradient_clip_val arrives through the
    configure_gradient_clipping hook, and step() runs the engine's clip + dW+Adam launches.  It must follow the default
    level-1 path (materialised gradients, torch's clip_grad_norm_, FxAdam on the arenas) to rounding."""
    import flexynesis_amd.models as M
    from flexynesis_amd.models.base import FxAdam
    torch.manual_seed(5)
    ds = _synthetic_ds(n=256, F=(8192, 4100), seed=3)
    cfg = {"latent_dim": 32, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 64}
    m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    m.to(DEV)
    ma, mb = copy.deepcopy(m), copy.deepcopy(m)
    ma.fused_optimizer = True
    oa, ob = ma.configure_optimizers(), mb.configure_optimizers()
    assert isinstance(oa, FxAdam) and oa.fused and not ob.fused
    # Three steps: t = 2, 3 exercise non-zero moments and bias corrections.  Beyond that the comparison measures Adam, not the code:
    # entries at the noise floor of their gradient step +lr in one path and -lr in the other, and a seed sweep
    # of this very test (6 initialisations x 3 shapes, aligned or not, engine widths padded or not) has 1 trajectory in 6 whose
    # small-parameter gradients are 5-30 % apart by step 4-6 while the others stay at 1e-3.
    lr, steps, B = cfg["lr"], 3, 64
    losses = {0: [], 1: []}
    for it in range(steps):
        idx = torch.arange(it * 32, it * 32 + B) % 256
        batch = ({k: v[idx].to(DEV) for k, v in ds.dat.items()}, {k: torch.as_tensor(v)[idx].to(DEV) for k, v in ds.ann.items()}, None)
        for j, (mm, oo) in enumerate(((ma, oa), (mb, ob))):
            mm.train()
            oo.zero_grad()
            loss = mm.training_step(batch, it, log=False)
            loss.backward()
            big = [k for k in mm._store.big_keys]
            assert len(big) == 2
            grads = dict(mm.named_parameters())
            if j == 0:
                assert all(grads[k].grad is None for k in big)                 # never formed
                assert grads["encoders.0.layer_out.weight"].grad is not None
            else:
                assert all(grads[k].grad is not None for k in big)
            mm.configure_gradient_clipping(oo, 1.0, "norm")                     # what Lightning's Trainer calls
            oo.step()
            losses[j].append(float(loss))
    assert oa.max_norm == 1.0
    for it, (a, b) in enumerate(zip(losses[0], losses[1])):
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-6, (it, losses[0], losses[1])
    assert losses[0][-1] < losses[0][0]
    sa, sb = ma.state_dict(), mb.state_dict()
    noise = (".layer_1.bias", ".layer_out.bias", "fusion_block.bias", ".running_mean")
    moved = 0.0
    for k in sa:
        if sa[k].dtype.is_floating_point:
            d = (sa[k].double() - sb[k].double()).abs()
            assert float(d.max()) <= 2.1 * lr * steps, k
            if not k.endswith(noise):
                assert float((d > 1e-5 + 1e-3 * sb[k].double().abs()).double().mean()) <= 5e-3, k
    w0 = m.state_dict()["encoders.0.layer_1.weight"]
    assert float((sa["encoders.0.layer_1.weight"] - w0).abs().max()) > 1e-4       # the wide weights did move
    # a second step() without a backward in between applies nothing
    before = sa["encoders.0.layer_1.weight"].clone()
    oa.step()
    assert torch.equal(ma.state_dict()["encoders.0.layer_1.weight"], before)
    # copies and pickles of a model that has trained in this mode carry no run-time state (plans, events, the optimiser)
    import io
    mc = copy.deepcopy(ma)
    assert mc._fused_ready is None and mc._fx_optimizer is None and mc.fused_optimizer
    buf = io.BytesIO()
    torch.save(ma, buf)
    buf.seek(0)
    md = torch.load(buf, weights_only=False)
    assert torch.equal(md.state_dict()["encoders.0.layer_1.weight"].cpu(), before.cpu()) and md._fx_optimizer is None
    oc = mc.configure_optimizers()
    mc.train()
    oc.zero_grad()
    mc.training_step(batch, 7, log=False).backward()
    mc.configure_gradient_clipping(oc, 1.0, "norm")
    oc.step()
    assert not torch.equal(mc.state_dict()["encoders.0.layer_1.weight"], before)
    # an upstream gradient other than 1 is refused (detected one backward late, without a host sync in the step)
    ma.train()
    loss = ma.training_step(batch, 99, log=False)
    (2.0 * loss).backward()
    torch.cuda.synchronize()
    loss = ma.training_step(batch, 100, log=False)
    with pytest.raises(RuntimeError, match="upstream gradient"):
        loss.backward()


@pytest.mark.parametrize("name,targets,surv", [("supervised_vae", ["c"], False), ("MultiTripletNetwork", ["c", "y"], False),
                                               ("CrossModalPred", ["y"], True)])
def test_level1_fused_optimizer_other_families(name, targets, surv):
    """The fused level-1 mode on the VAE family and the triplet network: wide weights (encoders and decoders) move without
    ever having a .grad, and the run follows the default level-1 mode (same seeds -> same in-kernel draws) to rounding."""
    import random
    import flexynesis_amd.models as M
    from flexynesis_amd.data import TripletMultiOmicDataset
    torch.manual_seed(2)
    random.seed(2)                      # TripletMultiOmicDataset draws its positives / negatives from the global generators
    np.random.seed(2)
    ds = _synthetic_ds(n=192, F=(8192, 4100), seed=4)
    cfg = {"latent_dim": 24, "hidden_dim_factor": 0.25, "lr": 2e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 32}
    kw = dict(surv_event_var="event", surv_time_var="time") if surv else {}
    if name == "CrossModalPred":
        kw.update(input_layers=["gex", "cnv"], output_layers=["cnv"])
    m = getattr(M, name)(cfg, ds, targets, device_type="cuda", **kw)
    m.to(DEV)
    w0 = {k: v.clone() for k, v in m.state_dict().items()}
    ma, mb = copy.deepcopy(m), copy.deepcopy(m)
    ma.fused_optimizer = True
    oa, ob = ma.configure_optimizers(), mb.configure_optimizers()
    tds = TripletMultiOmicDataset(ds, "c") if name == "MultiTripletNetwork" else None
    g = torch.Generator().manual_seed(0)
    losses = {0: [], 1: []}
    steps, lr = 4, cfg["lr"]
    for it in range(steps):
        idx = torch.randperm(192, generator=g)[:32].tolist()
        if tds is not None:
            items = [tds[i] for i in idx]
            col = lambda j: {k: torch.stack([torch.as_tensor(t[j][k]) for t in items]).to(DEV) for k in items[0][j]}
            batch = (col(0), col(1), col(2), col(3))
        else:
            batch = ({k: v[idx].to(DEV) for k, v in ds.dat.items()}, {k: torch.as_tensor(v)[idx].to(DEV) for k, v in ds.ann.items()}, None)
        for j, (mm, oo) in enumerate(((ma, oa), (mb, ob))):
            mm.train()
            oo.zero_grad()
            loss = mm.training_step(batch, it, log=False)
            loss.backward()
            mm.configure_gradient_clipping(oo, 1.0, "norm")
            oo.step()
            losses[j].append(float(loss.detach()))
    big = list(ma._store.big_keys)
    assert len(big) >= 2
    params = dict(ma.named_parameters())
    assert all(params[k].grad is None for k in big)
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in big:
        assert float((sa[k] - w0[k]).abs().max()) > 1e-4, k
    for a, b in zip(losses[0], losses[1]):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (losses[0], losses[1])
    noise = (".layer_1.bias", ".layer_out.bias", "fusion_block.bias", ".running_mean", ".bias")
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert bool(torch.isfinite(sa[k]).all()), k
            d = (sa[k].double() - sb[k].double()).abs()
            if "running_" in k:            # BatchNorm statistics follow the activations, not the learning rate
                assert float(d.max()) <= 5e-2 * float(sb[k].double().abs().max()) + 1e-3, k
                continue
            assert float(d.max()) <= 2.1 * lr * steps, k
            if not k.endswith(noise):
                assert float((d > 1e-5 + 2e-3 * sb[k].double().abs()).double().mean()) <= 1e-2, k


def test_level1_tape_graphs_equal_eager_launches():
    """FX_LEVEL1_GRAPHS=1: the level-1 plans replay their forward / backward / optimiser tapes as hipGraphs from the third use
    on; the same run with eager launches must give the same bits.  Runs in its own process (tests/_level1_graphs_check.py):
    the switch is opt-in because graphs captured in the middle of a long-lived process made later, unrelated graph
    launches crash in one of three runs of this whole suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FX_LEVEL1_GRAPHS="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_level1_graphs_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "LEVEL1_GRAPHS_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_fxadam_under_a_lightning_style_closure():
    """Lightning's automatic optimisation calls ``optimizer.step(closure=...)``: the closure runs training_step, zero_grad,
    backward and -- after the backward -- the gradient-clipping hook; only then does the optimiser update.  The fused
    FxAdam must find the backward's plan and the clip value that the closure left behind, and configure_optimizers() must
    pick the fused mode on its own when such a trainer is attached."""
    import types
    import flexynesis_amd.models as M
    from flexynesis_amd.models.base import FxAdam
    torch.manual_seed(11)
    ds = _synthetic_ds(n=256, F=(8192, 4100), seed=8)
    cfg = {"latent_dim": 32, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 64}
    m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    m.to(DEV)
    ma, mb = copy.deepcopy(m), copy.deepcopy(m)
    # a trainer as the reference configures it (main.py:212-225): gradient_clip_val=1.0, no accumulation, fp32
    ma.__dict__["_trainer"] = types.SimpleNamespace(accumulate_grad_batches=1, precision="32-true", gradient_clip_algorithm=None)
    oa, ob = ma.configure_optimizers(), mb.configure_optimizers()
    assert isinstance(oa, FxAdam) and oa.fused and ma.fused_optimizer is True
    assert isinstance(ob, FxAdam) and not ob.fused                      # no trainer attached: materialised gradients
    for it in range(4):
        idx = torch.arange(it * 48, it * 48 + 64) % 256
        batch = ({k: v[idx].to(DEV) for k, v in ds.dat.items()}, {k: torch.as_tensor(v)[idx].to(DEV) for k, v in ds.ann.items()}, None)
        for mm, oo in ((ma, oa), (mb, ob)):
            mm.train()

            def closure(mm=mm, oo=oo):
                loss = mm.training_step(batch, it, log=False)
                oo.zero_grad()
                loss.backward()
                mm.configure_gradient_clipping(oo, 1.0, None)          # precision plugin's _after_closure
                return loss
            out = oo.step(closure=closure)
            assert out is not None and bool(torch.isfinite(out.detach()).all())
    sa, sb = ma.state_dict(), mb.state_dict()
    noise = (".layer_1.bias", ".layer_out.bias", "fusion_block.bias", ".running_mean")
    for k in sa:
        if sa[k].dtype.is_floating_point and "running_" not in k:
            d = (sa[k].double() - sb[k].double()).abs()
            assert float(d.max()) <= 2.1 * 1e-3 * 4, k
            if not k.endswith(noise):
                assert float((d > 1e-5 + 1e-3 * sb[k].double().abs()).double().mean()) <= 5e-3, k
    assert float((sa["encoders.0.layer_1.weight"] - m.state_dict()["encoders.0.layer_1.weight"]).abs().max()) > 1e-4


def test_fit_and_predict_edge_cases():
    """What the loop does at the edges the reference has too: no validation set, a batch larger than the training split
    (drop_last leaves nothing), BatchNorm's batch-size-1 error (surfacing as a failed trial, +inf, in run_trial), batch size 2,
    a head whose labels are all missing (zero-loss leaf, reference direct_pred.py:167-177), validation sets smaller than a
    batch down to one row, predict on one sample."""
    import flexynesis_amd.models as M
    from flexynesis_amd._lib import FxError
    from flexynesis_amd.fit import fit, run_trial, split_indices
    ds = _synthetic_ds(n=300, F=(500, 300), seed=12)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 2, "batch_size": 32}
    tr, va = split_indices(len(ds), 0.2, 0)
    new = lambda cls=M.DirectPred, t=("y",): cls(cfg, ds, list(t), device_type="cuda")
    assert np.isnan(fit(new(t=("y", "c")), ds, tr, None, batch_size=32, epochs=2, lr=1e-3).val_loss)
    with pytest.raises(ValueError, match="exceeds the training split"):
        fit(new(), ds, tr[:20], va, batch_size=32, epochs=2, lr=1e-3)
    with pytest.raises(FxError, match="more than 1 value per channel"):
        fit(new(), ds, tr, va, batch_size=1, epochs=1, lr=1e-3)
    val, ep = run_trial(M.DirectPred, dict(cfg, batch_size=1), ds, ["y"], early_stop_patience=0, seed=0, device=torch.device(DEV))[:2]
    assert val == float("inf") and ep == 0                                   # a failed trial does not abort the sweep
    assert np.isfinite(fit(new(), ds, tr[:40], va, batch_size=2, epochs=1, lr=1e-3).val_loss)
    import copy as _copy
    dn = _copy.copy(ds)
    dn.ann = dict(ds.ann, y=torch.full((len(ds),), float("nan")))
    r = fit(M.DirectPred(cfg, dn, ["y", "c"], device_type="cuda"), dn, tr, va, batch_size=32, epochs=2, lr=1e-3)
    assert np.isfinite(r.val_loss) and all(np.isfinite(h["train_loss"]) for h in r.history)
    assert np.isfinite(fit(new(), ds, tr, va[:5], batch_size=32, epochs=1, lr=1e-3).val_loss)
    assert np.isfinite(fit(new(M.supervised_vae, ("c",)), ds, tr, va[:1], batch_size=32, epochs=1, lr=1e-3).val_loss)
    out = new(t=("y", "c")).predict(ds.subset([3]))
    assert out["y"].shape == (1, 1) and out["c"].shape[0] == 1


def test_gradient_accumulation_two_backwards_before_zero_grad():
    """Lightning's accumulate_grad_batches / a hand-written accumulation loop: two loss.backward() calls before zero_grad
    must leave g1 + g2 in EVERY param.grad -- small parameters (whose .grad is a zero-copy view of the gradient arena after
    the first backward) and wide weights alike.  (the arena view used to be overwritten by the second backward and
    then added to itself: 2 g2.)"""
    import flexynesis_amd.models as M
    g = Golden("directpred_2omics_multitask")
    m, ds = _model_from_golden(g, M.DirectPred)
    m.load_state_dict(g.state0())
    m.to(DEV)
    m.fused_optimizer = False
    m.train()
    m._bind()
    batches = []
    for i in (0, 1):
        b = g.batch(i)
        batches.append(({n: x.to(DEV) for (n, _), x in zip(g.spec.layers, b["x"])}, {k: v.to(DEV) for k, v in b["y"].items()},
                        tuple(f"s{j}" for j in range(8))))
    params = dict(m.named_parameters())

    def grads_of(bs, step0):
        """gradients after backward over the batches ``bs`` without zero_grad in between (training_step index fixed per
        batch so that the dropout masks of a batch are the same in both runs)"""
        for p in params.values():
            p.grad = None
        for j, b in enumerate(bs):
            m._store.ctrl[0] = float(step0 + j)           # the dropout stream is keyed on the step counter
            loss = m.training_step(b, step0 + j)
            loss.backward()
        return {k: p.grad.detach().clone() for k, p in params.items() if p.grad is not None}

    g1 = grads_of(batches[:1], 10)
    g2 = grads_of(batches[1:], 11)
    g12 = grads_of(batches, 10)
    assert set(g12) == set(g1) == set(g2)
    wide = "encoders.0.layer_1.weight"
    assert wide in g12
    for k in g12:
        want = g1[k] + g2[k]
        scale = float(want.abs().max()) + 1e-12
        assert float((g12[k] - want).abs().max()) <= 1e-6 * scale + 1e-9, (k, float((g12[k] - want).abs().max()), scale)
        # and it is NOT twice the second gradient (the failure mode) wherever the two batches' gradients differ
        if float((g1[k] - g2[k]).abs().max()) > 1e-3 * scale:
            assert float((g12[k] - 2 * g2[k]).abs().max()) > 1e-4 * scale, k
    # the first backward after zero_grad(set_to_none) is still zero-copy
    for p in params.values():
        p.grad = None
    m.training_step(batches[0], 20).backward()
    pk = "encoders.0.layer_out.weight"
    assert params[pk].grad.data_ptr() == m._store.g(pk).data_ptr()


def test_run_trial_cv_and_sharded_fine_tune_on_the_engine():
    """The cross-validated branch of objective() (main.py:267-269, :327-333) and the sharded FineTuner driver on real engine
    fits.  With one process the sharded driver claims every unit itself, in longest-first order instead of loop order: the
    fits are independent and deterministic, so its records, its best configuration and its final weights equal the
    sequential driver's bit for bit."""
    import flexynesis_amd.models as M
    from flexynesis_amd.fit import fine_tune, full_train, kfold_indices, run_trial
    ds = _synthetic_ds(n=90)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 16}
    val, ep, model, info = run_trial(M.DirectPred, cfg, ds, ["y", "c"], seed=4, device="cuda", use_cv=True, n_splits=3,
                                     early_stop_patience=0)
    assert len(info["fold_val_losses"]) == 3 and all(np.isfinite(v) for v in info["fold_val_losses"])
    assert abs(val - float(np.mean(info["fold_val_losses"]))) < 1e-12 and ep == 3
    folds = kfold_indices(90, 3, 4)
    assert info["steps"] == sum(3 * (len(tr) // 16) for tr, _ in folds)
    val2, _, _, info2 = run_trial(M.DirectPred, cfg, ds, ["y", "c"], seed=4, device="cuda", use_cv=True, n_splits=3,
                                  early_stop_patience=0)
    assert val2 == val and info2["fold_val_losses"] == info["fold_val_losses"]          # seeded: reproducible
    final, finfo = full_train(M.DirectPred, dict(cfg, epochs=2), ds, ["y", "c"], seed=5, device="cuda")
    assert finfo["steps"] == 2 * (90 // 16) and set(final.predict(ds)) == {"y", "c"}
    torch.manual_seed(1)
    m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    kw = dict(n_splits=2, batch_size=16, learning_rates=[3e-3, 3e-4], max_epoch=3, seed=1, device="cuda",
              freeze_configs=[{"encoders": True, "supervisors": False}, {"encoders": False, "supervisors": False}])
    f_seq, b_seq, r_seq = fine_tune(copy.deepcopy(m), ds, **kw)
    f_sh, b_sh, r_sh = fine_tune(copy.deepcopy(m), ds, sharded=True, **kw)
    assert r_seq == r_sh and b_seq == b_sh
    sa, sb = f_seq.state_dict(), f_sh.state_dict()
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k


@pytest.mark.parametrize("fused", [False, True])
def test_fx_adam_state_dict_round_trip_resumes_adam(fused):
    """optimizer.state_dict() carries the Adam moments and the step count out of the engine's arenas: a model +
    optimiser restored from checkpoints continue exactly like the uninterrupted pair (torch.optim.Adam's contract, which a
    Lightning checkpoint relies on)."""
    import flexynesis_amd.models as M
    g = Golden("directpred_2omics_multitask")
    m, ds = _model_from_golden(g, M.DirectPred)
    m.load_state_dict(g.state0())
    m.to(DEV)
    m.fused_optimizer = fused
    batches = []
    for i in (0, 1):
        b = g.batch(i)
        batches.append(({n: x.to(DEV) for (n, _), x in zip(g.spec.layers, b["x"])}, {k: v.to(DEV) for k, v in b["y"].items()},
                        tuple(f"s{j}" for j in range(8))))

    def steps(model, opt, first, n):
        for it in range(first, first + n):
            model.train()
            opt.zero_grad()
            model._bind().ctrl[0] = float(it)                     # same dropout stream position in both runs
            loss = model.training_step(batches[it % 2], it)
            loss.backward()
            model.configure_gradient_clipping(opt, 1.0, "norm")
            opt.step()

    ma = copy.deepcopy(m)
    ma.fused_optimizer = fused
    oa = ma.configure_optimizers()
    steps(ma, oa, 0, 3)
    ck_model = {k: v.detach().cpu().clone() for k, v in ma.state_dict().items()}
    ck_opt = oa.state_dict()
    assert ck_opt["fx"]["step"] == 3 and float(ck_opt["fx"]["exp_avg_sq"]["encoders.0.layer_1.weight"].abs().sum()) > 0
    steps(ma, oa, 3, 2)                                           # the uninterrupted run
    mb = copy.deepcopy(m)
    mb.fused_optimizer = fused
    mb.load_state_dict(ck_model)
    ob = mb.configure_optimizers()
    ob.load_state_dict(ck_opt)
    steps(mb, ob, 3, 2)                                           # the resumed run
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert float((sa[k] - sb[k]).abs().max()) <= 1e-7 + 1e-6 * float(sa[k].abs().max()), k
    assert ob.state_dict()["fx"]["step"] == 5
