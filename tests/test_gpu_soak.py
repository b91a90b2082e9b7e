"""hipGraph lifetime.  Root cause of round 2's intermittent crashes inside hipGraphLaunch, reproduced in
isolation by scripts/graph_lifetime_probe.py: on ROCm torch's ~CUDAGraph synchronises the device, which is not permitted while
any stream is capturing -- a graph object that dies during ANOTHER graph's capture terminates the process.  The engine therefore
(1) keeps the cyclic collector off while capturing, (2) releases graphs explicitly (StepPlan.close / PipelinedStep.close at the
end of fit(), FxModel.close), (3) parks a release requested during a capture until the capture has ended (ops.retire_graph).
These tests exercise (2) and (3) directly and then soak the process the way an HPO sweep / FineTuner run does, with per-epoch
validation graphs and level-1 tape graphs ON (their defaults)."""
import gc

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ds(n, F, seed):
    from flexynesis_amd.data import MultiOmicDataset
    g = torch.Generator().manual_seed(seed)
    dat = {"gex": torch.randn(n, F[0], generator=g), "cnv": torch.randn(n, F[1], generator=g)}
    ann = {"y": dat["gex"][:, :8].sum(1) / 3, "c": (dat["cnv"][:, 0] > 0).float()}
    feats = {k: [f"{k}{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    return MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)], {})


def _pipe(B=32, seed=0):
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    layers = [("gex", 1100), ("cnv", 900)]
    spec = ArchSpec("DirectPred", layers, 16, 0.25, 8, [("y", "numerical", 1)], None, None, True)
    cohort = synthetic_cohort(layers, 256, DEV, seed=seed)
    store = ParamStore(spec, DEV, materialize_big_grads=False, big_threshold=1 << 16)
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=4, seed=seed)
    pipe.idx.copy_(torch.randperm(256, device=DEV)[: 4 * B])
    pipe.prime()
    pipe.step(1e-3)
    pipe.capture(1e-3)
    return pipe


def test_graph_released_during_a_capture_is_parked_until_it_ends():
    """The crash, made deterministic and defused: closing a plan (= giving up its hipGraphs) while another graph is being
    captured.  Without ops.retire_graph the CUDAGraph destructor would run inside the capture and abort the process."""
    from flexynesis_amd import ops
    a, b = _pipe(seed=1), _pipe(seed=2)
    for _ in range(3):
        a.replay(); b.replay()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    x = torch.ones(1024, device=DEV)
    with ops.graph_capture(g):
        x.mul_(2.0)
        assert ops.capturing()
        a.close()                                   # two step graphs given up in the middle of a capture
        assert len(ops._GRAVEYARD) == 2 and a.graphs == [None, None]
        x.add_(1.0)
    assert not ops.capturing() and not ops._GRAVEYARD          # released when the capture ended
    g.replay()
    for _ in range(3):
        b.replay()                                   # unrelated graphs keep working
    torch.cuda.synchronize()
    assert float(x[0]) == 3.0 and all(v == v for v in b.losses().values())
    a.close(); b.close()                             # idempotent
    del g


def test_fit_releases_its_graphs_when_it_returns():
    import flexynesis_amd.models as M
    from flexynesis_amd import engine, fit as F
    made = []
    orig = engine.PipelinedStep.capture

    def spy(self, lr):
        made.append(self)
        return orig(self, lr)
    engine.PipelinedStep.capture = spy
    try:
        ds = _ds(300, (300, 200), 3)
        cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 32}
        m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
        tr, va = F.split_indices(300, 0.2, 0)
        res = F.fit(m, ds, tr, va, batch_size=32, epochs=3, lr=1e-3, seed=1)
    finally:
        engine.PipelinedStep.capture = orig
    assert res.steps == 3 * (len(tr) // 32) and len(made) == 1
    assert made[0].graphs == [None, None] and all(p.graph is None and "_eval_graph" not in p.__dict__ for p in made[0].plans)


def test_soak_fits_and_level1_models_with_every_graph_feature_on(monkeypatch):
    """~60 engine fits (training graphs, per-epoch validation graphs captured in the middle of a fit) interleaved with
    short-lived level-1 models whose tapes run as hipGraphs, models dying by reference count, by the cyclic collector and
    by close(): the lifetime pattern of an HPO sweep followed by the FineTuner, in one process."""
    import flexynesis_amd.models as M
    from flexynesis_amd import fit as F
    monkeypatch.setenv("FX_LEVEL1_GRAPHS", "1")
    assert F.EVAL_GRAPHS
    wide, small = _ds(256, (4100, 2052), 1), _ds(400, (300, 200), 2)
    cfg = {"latent_dim": 32, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 2, "batch_size": 64}
    tr, va = F.split_indices(len(small), 0.2, 1)
    keep, tape_graphs = [], 0
    for it in range(60):
        m = M.DirectPred(cfg, wide, ["y", "c"], device_type="cuda")
        m.fused_optimizer = bool(it & 1)
        opt = m.configure_optimizers()
        for s in range(4):
            idx = torch.arange(s * 32, s * 32 + 64) % 256
            batch = ({k: v[idx].to(DEV) for k, v in wide.dat.items()}, {k: torch.as_tensor(v)[idx].to(DEV) for k, v in wide.ann.items()}, None)
            m.train()
            opt.zero_grad()
            loss = m.training_step(batch, s, log=False)
            loss.backward()
            m.configure_gradient_clipping(opt, 1.0, "norm")
            opt.step()
        tape_graphs += sum(len(p.__dict__.get("_tape_graph", {})) for p in m._plans.values())
        assert torch.isfinite(loss.detach()).all()
        if it % 3 == 0:
            m.close()                                   # explicit release
        elif it % 3 == 1:
            keep.append((m, opt))                       # stays alive for a while, dies in a batch below
        del m, opt
        m2 = M.DirectPred(cfg, small, ["y", "c"], device_type="cuda")
        res = F.fit(m2, small, tr, va, batch_size=64, epochs=3, lr=3e-3, seed=it)
        assert res.steps == 3 * (len(tr) // 64) and res.val_loss == res.val_loss
        if it % 8 == 7:
            keep.clear()
            gc.collect()
    torch.cuda.synchronize()
    assert tape_graphs >= 60                            # the level-1 models really ran from captured tapes


def test_small_fp32_kernels_beside_an_mfma_kernel_are_exact():
    """Regression for the packed-fp32 hazard (scripts/pkfma_hazard_probe.hip): with v_pk_fma_f32 in its inner loop
    fx_small_linear_fwd dropped single terms in lanes 48..63 (40 launches in 60000) whenever waves of an MFMA kernel
    (ds_read_b128 -> MFMA chains: fx_gram_kb_group, the dW + Adam kernel) shared its SIMDs -- which the step's own side branch
    and trials in flight make routine.  The library is built without packed fp32 ops; the small kernels must now be bit-stable
    beside that noise."""
    import threading, time
    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    g = torch.Generator(device=DEV); g.manual_seed(0)
    B, L, n = 128, 85, 2
    ecat = torch.randn(B, n * L, device=DEV, generator=g)
    W = torch.randn(L, n * L, device=DEV, generator=g) * 0.1
    b = torch.randn(L, device=DEV, generator=g)
    dy = torch.randn(B, L, device=DEV, generator=g)
    layers = [("gex", 4000), ("cnv", 4000)]
    spec = ArchSpec("DirectPred", layers, 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
    cohort = synthetic_cohort(layers, 2048, DEV, seed=1)
    stop, started, err = [False], threading.Event(), []

    def hammer():
        try:
            torch.cuda.set_device(DEV)
            with torch.cuda.stream(torch.cuda.Stream()):
                st2 = ParamStore(spec, DEV, materialize_big_grads=False)
                p2 = PipelinedStep(st2, 64, cohort=cohort, n_batches=12, seed=9)
                p2.idx.copy_(torch.randperm(2048, device=DEV)[: 12 * 64]); p2.prime()
                p2.step(1e-3); p2.step(1e-3)
                calls = [c for nm in ("gather", "fwd", "bwd", "opt") for seg in getattr(p2.plans[0], "t_" + nm).segments for br in seg
                         for c in br if c[0] is not None and c[1] in ("fx_gram_kb_group", "fx_linear_dw_adam_fwd_bf16x3")]
                assert {c[1] for c in calls} == {"fx_gram_kb_group", "fx_linear_dw_adam_fwd_bf16x3"}
                s_ = torch.cuda.current_stream().cuda_stream
                k = 0
                started.set()
                while not stop[0]:
                    for fn, name, args in calls:
                        fn(*args, s_)
                    k += 1
                    if k % 16 == 0:
                        torch.cuda.current_stream().synchronize()
                torch.cuda.current_stream().synchronize()
                p2.close()
        except Exception as e:          # pragma: no cover
            err.append(e); started.set()

    th = threading.Thread(target=hammer)
    th.start()
    try:
        assert started.wait(120) and not err, err
        time.sleep(0.2)
        with torch.cuda.stream(torch.cuda.Stream()):
            ref_y = torch.empty(B, L, device=DEV)
            ops.small_linear_fwd(ops.IMMEDIATE, ref_y, ecat, W, b)
            ref_dx, ref_gW, ref_gb = torch.empty(B, n * L, device=DEV), torch.empty_like(W), torch.empty_like(b)
            ops.small_linear_bwd(ops.IMMEDIATE, ref_dx, ref_gW, ref_gb, dy, ecat, W)
            torch.cuda.current_stream().synchronize()
            bad = 0
            for chunk in range(100):
                outs = []
                for it in range(200):
                    y = torch.empty(B, L, device=DEV)
                    ops.small_linear_fwd(ops.IMMEDIATE, y, ecat, W, b)
                    outs.append((y, ref_y))
                    if it % 4 == 0:
                        dx, gW = torch.empty(B, n * L, device=DEV), torch.empty_like(W)
                        ops.small_linear_bwd(ops.IMMEDIATE, dx, gW, torch.empty_like(b), dy, ecat, W)
                        outs += [(dx, ref_dx), (gW, ref_gW)]
                bad += sum(0 if torch.equal(o, r) else 1 for o, r in outs)
            assert bad == 0, f"{bad} launches differ from the quiet result"
    finally:
        stop[0] = True
        th.join(120)
    assert not err, err


def test_trials_in_flight_match_one_at_a_time():
    """run_cfg5(in_flight=2): two trials at a time on host threads with their own streams -- every trial's validation loss,
    the winner and its weights equal the one-at-a-time sweep's (hipGraph replay or eager)."""
    from flexynesis_amd.sweep import run_cfg5
    kw = dict(n_trials=6, epochs=2, features=3000, samples=512, seed=3)
    a = run_cfg5(DEV, in_flight=1, use_graph=True, **kw)
    b = run_cfg5(DEV, in_flight=1, use_graph=False, **kw)
    c = run_cfg5(DEV, in_flight=2, **kw)
    d = run_cfg5(DEV, in_flight=3, **kw)
    for o in (b, c, d):
        assert o["trial_val_losses"] == a["trial_val_losses"]
        assert o["best_trial"] == a["best_trial"] and o["winner_state_tensors"] == a["winner_state_tensors"]
    assert c["trials_in_flight_per_gpu"] == 2 and not c["hipgraph_replay"]
    with pytest.raises(ValueError):
        run_cfg5(DEV, in_flight=2, use_graph=True, **kw)


def test_units_in_flight_never_capture_and_share_one_cohort():
    """a user trial_fn that calls the fit entry points with their DEFAULT use_graph=True from
    trials.run_units(in_flight=2).  A hipGraph capture is process-wide on ROCm, so such a fit is downgraded to eager launches
    (with a RuntimeWarning) instead of capturing beside a neighbour that synchronises; the resident cohort is built once, under
    a lock, and published only after its upload stream has finished; results equal the one-at-a-time run with graphs; the
    worker threads' stream pools are gone afterwards; an explicit capture on such a thread is refused."""
    import warnings
    import flexynesis_amd.models as M
    from flexynesis_amd import ops, trials
    from flexynesis_amd._lib import FxError
    from flexynesis_amd.data import MultiOmicDataset
    from flexynesis_amd.fit import run_trial
    g = torch.Generator().manual_seed(5)
    n = 320
    dat = {"gex": torch.randn(n, 1200, generator=g), "cnv": torch.randn(n, 800, generator=g)}
    ann = {"y": dat["gex"][:, :8].sum(1) + 0.1 * torch.randn(n, generator=g)}
    feats = {k: [f"{k}{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    plist = trials.draw_search_space(6, seed=11, epochs=2)

    def sweep(in_flight):
        ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(n)], {})     # fresh: no cached cohort
        refused = []

        def unit(uid):
            if in_flight > 1 and not refused:
                try:
                    with ops.graph_capture(torch.cuda.CUDAGraph()):
                        pass
                except FxError as e:
                    refused.append(str(e))
            val, ep, model, info = run_trial(M.DirectPred, plist[uid], ds, ["y"], early_stop_patience=0, seed=100 + uid, device="cuda")
            assert "error" not in info, info
            del model
            return val, ep, None
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            table, _ = trials.run_units(len(plist), unit, device=DEV, in_flight=in_flight)
        return table[:, 1].tolist(), [x for x in w if issubclass(x.category, RuntimeWarning) and "launching eagerly" in str(x.message)], refused, ds

    before = set(ops._POOLS)
    one, w1, _, _ = sweep(1)
    two, w2, refused, ds2 = sweep(2)
    assert one == two and all(np.isfinite(v) for v in one)
    assert not w1 and len(w2) >= 1                         # downgraded (and said so) only where fits run side by side
    assert refused and "capture" in refused[0]
    assert getattr(ds2, "_fx_cohort", None) is not None    # built once, shared by both threads
    assert set(ops._POOLS) <= before | {k for k in ops._POOLS if k[1] == __import__("threading").get_ident()}, "worker pools leaked"
    assert not ops._WORKER_STREAMS
