"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against (a) the committed golden vectors generated from the reference and (b) the CPU oracle
(oracle/restate.py) on seeded inputs at larger shapes.  Nothing here reads /root/reference."""
import numpy as np
import pytest
import torch

from golden_io import Golden, MODEL_CASES

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4        # BASELINE.json north_star: losses within 1e-4 relative on identical inputs


def _dev():
    return torch.device("cuda:0")


def close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = (torch.as_tensor(atol).double().reshape(-1) if torch.is_tensor(atol) else atol) + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{a.numel()} off, max err {err.max().item():.3e} "
                                 f"(ref max {b.abs().max().item():.3e})")


def noise_atol(grad, grad_norm, lr, default):
    """See tests/test_oracle_pinning.py::shadow_atol: exactly-zero-gradient entries are rounding noise
    that Adam amplifies to +-lr; they are implementation-defined in the reference too."""
    if grad is None:
        return default
    noise = grad.abs() < 1e-6 * float(grad_norm)
    return torch.where(noise, torch.tensor(2.1 * lr), torch.tensor(float(default)))


def arch_from_golden(g):
    from flexynesis_amd.arch import ArchSpec
    s = g.spec
    return ArchSpec(s.model, list(s.layers), s.latent_dim, s.hidden_dim_factor, s.supervisor_hidden_dim,
                    list(s.variables), s.surv_event_var, s.surv_time_var, s.use_loss_weighting,
                    s.input_layers, s.output_layers)


def feed(plan, spec, batch, draws):
    dev = plan.dev
    if spec.model == "MultiTripletNetwork":
        plan.set_batch(parts=[[x.to(dev) for x in batch[k]] for k in ("anchor", "positive", "negative")],
                       y={k: v.to(dev) for k, v in batch["y"].items()})
    else:
        plan.set_batch(x_list=[x.to(dev) for x in batch["x"]], y={k: v.to(dev) for k, v in batch["y"].items()})
    plan.set_draws({k: v.to(dev) for k, v in draws.items()})


def golden_opt(g, s):
    return (0, {}, {}) if s < 0 else (s + 1, g.exp(s, "m"), g.exp(s, "v"))


# ---------------------------------------------------------------------------------------------------
# kernels vs plain fp32/fp64 torch references at awkward shapes
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 5000, 2000), (128, 64, 5000), (7, 3, 5), (33, 130, 70), (384, 100, 96),
                                   (128, 1, 16), (1, 16, 64), (130, 257, 1000)])
def test_gemm_layouts(M, N, K):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).to(dev)
    Bt = torch.randn(N, K, generator=g).to(dev)      # NT operand
    bias = torch.randn(N, generator=g).to(dev)
    ws = ops.Workspace(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, C, A, Bt, bias, ws)
    ref = (A.double() @ Bt.double().t() + bias.double())
    scale = (A.double().abs() @ Bt.double().abs().t()).max().item()
    assert (C.double() - ref).abs().max().item() <= 2e-6 * scale + 1e-6, "NT"
    Bn = Bt.t().contiguous()                          # [K,N]
    C2 = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NN, C2, A, Bn, None, ws)
    assert (C2.double() - A.double() @ Bn.double()).abs().max().item() <= 2e-6 * scale + 1e-6, "NN"
    At = A.t().contiguous()                           # [K,M]
    C3 = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_TN, C3, At, Bn, None, ws)
    assert (C3.double() - A.double() @ Bn.double()).abs().max().item() <= 2e-6 * scale + 1e-6, "TN"
    # accumulate + strided output view
    big = torch.zeros(M, N + 9, device=dev)
    view = big[:, 4:4 + N]
    view.fill_(1.0)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, view, A, Bt, None, ws, accumulate=True)
    assert (view.double() - (A.double() @ Bt.double().t() + 1.0)).abs().max().item() <= 2e-6 * scale + 1e-5, "acc"
    assert float(big[:, :4].abs().max()) == 0.0 and float(big[:, 4 + N:].abs().max()) == 0.0, "out-of-view write"


def test_gemm_transpose_detecting():
    """A = I with an asymmetric B catches row/column swaps in the MFMA C-layout (guide rule 16)."""
    from flexynesis_amd import ops
    dev = _dev()
    n = 96
    A = torch.eye(n, device=dev)
    Bm = (torch.arange(n * n, device=dev, dtype=torch.float32).reshape(n, n) / 7.0)
    C = torch.zeros(n, n, device=dev)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NN, C, A, Bm, None, ops.Workspace(dev))
    assert torch.equal(C, Bm)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, C, A, Bm, None, ops.Workspace(dev))
    assert torch.equal(C, Bm.t())


@pytest.mark.parametrize("B,C,pre,post", [(8, 16, 0, 2), (128, 5000, 0, 2), (10, 33, 1, 0), (128, 1250, 1, 0)])
def test_bn_act_fwd_bwd_vs_torch(B, C, pre, post):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, C, generator=g, dtype=torch.float64)
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rm0 = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rv0 = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    mask = (torch.rand(B, C, generator=g) < 0.9).double()
    dout = torch.randn(B, C, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    h = torch.where(xr > 0, xr, 0.2 * xr) if pre == 1 else xr
    mean, var = h.mean(0), h.var(0, unbiased=False)
    yb = (h - mean) / torch.sqrt(var + 1e-5) * gr + br
    out_ref = torch.relu(yb) * (mask / 0.9) if post == 2 else yb
    out_ref.backward(dout)
    f32 = lambda t: t.float().to(dev).contiguous()
    out, dx = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    rm, rv = f32(rm0), f32(rv0)
    sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dg, db, dbias = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    xd, gd, bd = f32(x), f32(gamma), f32(beta)
    ops.bn_act_fwd(ops.IMMEDIATE, out, xd, gd, bd, rm, rv, sm, si, pre, post, True, 0.1 if post == 2 else 0.0,
                   mask=f32(mask) if post == 2 else None)
    close(out, out_ref, 1e-4, 1e-5, "bn fwd")
    close(rm, 0.9 * rm0 + 0.1 * mean, 1e-5, 1e-6, "running_mean")
    close(rv, 0.9 * rv0 + 0.1 * h.var(0, unbiased=True), 1e-5, 1e-6, "running_var")
    ops.bn_act_bwd(ops.IMMEDIATE, dx, dg, db, dbias, f32(dout), xd, out if post == 2 else None, gd, sm, si, pre, post,
                   0.1 if post == 2 else 0.0)
    close(dx, xr.grad, 2e-4, 2e-5, "bn dx")
    close(dg, gr.grad, 2e-4, 2e-4, "bn dgamma")
    close(db, br.grad, 2e-4, 2e-4, "bn dbeta")
    close(dbias, xr.grad.sum(0), 1e-3, 2e-4, "dbias")
    # eval mode
    ops.bn_act_fwd(ops.IMMEDIATE, out, xd, gd, bd, f32(rm0), f32(rv0), None, None, pre, post, False)
    ev = (h.detach() - rm0) / torch.sqrt(rv0 + 1e-5) * gamma + beta
    close(out, torch.relu(ev) if post == 2 else ev, 1e-4, 1e-5, "bn eval")


def test_loss_kernels_vs_function_goldens():
    """mse / ce / cox / triplet / mmd / total against values produced by the reference's own functions."""
    from flexynesis_amd import ops
    dev = _dev()
    g = Golden("functions")
    d = lambda t: t.float().to(dev).contiguous()
    one = lambda: torch.zeros(1, device=dev)
    for tag in ("plain", "with_nan", "all_censored", "none_valid", "single_valid", "large_batch"):
        c = g.sub(f"cox/{tag}")
        o = d(c["outputs"])
        lo, do = one(), torch.full_like(o, float("nan"))
        ops.cox_ph(ops.IMMEDIATE, lo, do, o, d(c["durations"]), d(c["events"]))
        close(lo, c["loss"], 1e-5, 1e-6, f"cox {tag}")
        close(do, c["grad"], 1e-4, 1e-7, f"cox grad {tag}")
    for tag in ("mse/plain", "mse/with_nan", "mse/all_missing"):
        c = g.sub(tag)
        yh = d(c["yhat"])
        lo, dy = one(), torch.full_like(yh, float("nan"))
        ops.mse_masked(ops.IMMEDIATE, lo, dy, yh, d(c["y"]))
        close(lo, c["loss"], 1e-5, 1e-7, tag)
        close(dy, c["grad"], 1e-4, 1e-7, tag + " grad")
    for tag in ("ce/plain", "ce/with_missing", "ce/all_missing"):
        c = g.sub(tag)
        lg = d(c["yhat"])
        lo, dl = one(), torch.full_like(lg, float("nan"))
        ops.ce_masked(ops.IMMEDIATE, lo, dl, lg, d(c["y"]))
        close(lo, c["loss"], 1e-5, 1e-7, tag)
        close(dl, c["grad"], 1e-4, 1e-7, tag + " grad")
    t = g.sub("triplet")
    a, p, n = d(t["a"]), d(t["p"]), d(t["n"])
    lo, da, dp, dn = one(), torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    ops.triplet(ops.IMMEDIATE, lo, da, dp, dn, a, p, n)
    close(lo, t["loss"], 1e-5, 1e-7, "triplet")
    close(da, t["grad_a"], 1e-4, 1e-7); close(dp, t["grad_p"], 1e-4, 1e-7); close(dn, t["grad_n"], 1e-4, 1e-7)
    m = g.sub("mmd")
    z, x, xh, prior = d(m["z"]), d(m["x"]), m["xhat"], d(m["prior"])
    logits = d(torch.log(xh / (1 - xh)))           # xhat = sigmoid(logits)
    P, B = prior.shape[0], z.shape[0]
    rows, dz = torch.zeros(2 * (P + B), device=dev), torch.zeros_like(z)
    part, dlg, lo = torch.zeros(1024, device=dev), torch.empty_like(logits), one()
    ops.mmd_rows(ops.IMMEDIATE, rows, dz, prior, z)
    nblk = int(ops.lib.fx_recon_blocks(logits.numel()))
    ops.recon_sigmoid(ops.IMMEDIATE, part, dlg, None, logits, x)
    ops.mmd_finalize(ops.IMMEDIATE, lo, rows, P, B, part, nblk, float(logits.numel()), 1.0, False)
    close(lo, m["loss"], 1e-5, 1e-6, "mmd loss")
    close(dz, m["grad_z"], 1e-4, 1e-8, "mmd dz")
    close(dlg, m["grad_xhat"] * (xh * (1 - xh)), 2e-4, 1e-8, "recon dlogits")
    tot = g.sub("total")
    l1, l2 = d(tot["l1"].reshape(1)), d(tot["l2"].reshape(1))
    s1, s2 = torch.tensor([0.3], device=dev), torch.tensor([-0.2], device=dev)
    d1, d2, out = one(), one(), one()
    ops.total_loss(ops.IMMEDIATE, out, [l1, l2], [s1, s2], [d1, d2], True)
    close(out, tot["weighted"], 1e-6, 1e-7, "weighted total")
    close(d1, 1 - np.exp(-0.3) * float(tot["l1"]), 1e-5, 1e-7, "dlogvar")
    ops.total_loss(ops.IMMEDIATE, out, [l1], [], [], False)
    close(out, tot["single"], 1e-6, 1e-7, "single total")


def test_fused_dw_adam_matches_materialised():
    """fx_linear_dw_adam_f32 == (dW GEMM, then flat Adam) on the same inputs."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, n_out, k_in = 128, 300, 1100
    dy, x = torch.randn(B, n_out, generator=g).to(dev), torch.randn(B, k_in, generator=g).to(dev)
    W0 = torch.randn(n_out, k_in, generator=g).to(dev)
    m0, v0 = torch.randn(n_out, k_in, generator=g).to(dev) * 0.01, torch.rand(n_out, k_in, generator=g).to(dev) * 1e-3
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 4.0
    ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
    ctrl[4] = 0.37
    W1, m1, v1 = W0.clone(), m0.clone(), v0.clone()
    ops.linear_dw_adam(ops.IMMEDIATE, W1, m1, v1, dy, x, ctrl)
    dW = torch.empty_like(W0)
    ops.linear_bwd_w(ops.IMMEDIATE, dW, dy, x, ops.Workspace(dev))
    W2, m2, v2 = W0.clone(), m0.clone(), v0.clone()
    ops.adam_flat(ops.IMMEDIATE, W2.view(-1), dW.view(-1), m2.view(-1), v2.view(-1), ctrl)
    assert torch.equal(W1, W2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    # and both equal the textbook update in fp64
    gd = (dy.double().t() @ x.double()) * 0.37
    t = 5
    mm = 0.9 * m0.double() + 0.1 * gd
    vv = 0.999 * v0.double() + 0.001 * gd * gd
    ref = W0.double() - (1e-3 / (1 - 0.9 ** t)) * mm / (vv.sqrt() / np.sqrt(1 - 0.999 ** t) + 1e-8)
    close(W1, ref, 1e-5, 1e-6, "fused adam vs fp64")


def test_gram_norm_identity():
    """|dY^T X|_F^2 via the Gram identity equals the norm of the materialised gradient."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    B, H, F = 128, 700, 3000
    dy, x = (torch.randn(B, H, generator=g) * 0.01).to(dev), torch.randn(B, F, generator=g).to(dev)
    ws = ops.Workspace(dev)
    gx, gd = torch.empty(B, B, device=dev), torch.empty(B, B, device=dev)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, gx, x, x, None, ws)
    ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, gd, dy, dy, None, ws)
    slot = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.hadamard_sum(ops.IMMEDIATE, slot, gx, gd)
    ref = float(((dy.double().t() @ x.double()) ** 2).sum())
    assert abs(float(slot) - ref) <= 2e-6 * ref


def test_gather_rows_and_cursor():
    from flexynesis_amd import ops
    dev = _dev()
    src = torch.randn(50, 36, device=dev)
    idx = torch.randint(0, 50, (3 * 8,), device=dev)
    dst = torch.empty(8, 36, device=dev)
    ops.gather_rows(ops.IMMEDIATE, dst, src, idx)
    assert torch.equal(dst, src[idx[:8]])
    ctrl = torch.zeros(64, device=dev)
    ctrl[8] = 2.0
    ops.gather_rows(ops.IMMEDIATE, dst, src, idx, ctrl, 8)
    assert torch.equal(dst, src[idx[16:24]])
    lab = torch.randn(50, device=dev)
    out = torch.empty(8, device=dev)
    ops.gather_rows(ops.IMMEDIATE, out, lab, idx)
    assert torch.equal(out, lab[idx[:8]])


def test_bad_arguments_raise():
    from flexynesis_amd import ops
    from flexynesis_amd._lib import FxError
    dev = _dev()
    with pytest.raises(FxError):
        ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, torch.zeros(4, 4, device=dev), torch.zeros(4, 5, device=dev),
                 torch.zeros(3, 5, device=dev), None, None)
    with pytest.raises(FxError):
        ops.gemm(ops.IMMEDIATE, ops.GEMM_NT, torch.zeros(4, 4), torch.zeros(4, 5), torch.zeros(4, 5), None, None)
    with pytest.raises(FxError):
        ops.cox_ph(ops.IMMEDIATE, torch.zeros(1, device=dev), torch.zeros(2000, 1, device=dev),
                   torch.zeros(2000, 1, device=dev), torch.zeros(2000, device=dev), torch.zeros(2000, device=dev))


# ---------------------------------------------------------------------------------------------------
# whole training steps vs the reference goldens
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case", MODEL_CASES)
def test_train_step_matches_reference_golden(case, fused):
    """Per-step parity: every step starts from the reference's previous state + Adam moments."""
    from flexynesis_amd.engine import ParamStore, StepPlan
    g = Golden(case)
    spec = arch_from_golden(g)
    B = next(iter(g.batch(0)["y"].values())).shape[0]
    # big_threshold small enough that the wide first layers take the fused dW+Adam path
    store = ParamStore(spec, _dev(), big_threshold=512)
    plan = StepPlan(store, B, train=True, fused=fused, supplied_draws=True)
    for s in range(g.n_steps):
        store.load_state(g.state0() if s == 0 else g.exp(s - 1, "state"))
        store.reset_optimizer()
        t, m, v = golden_opt(g, s - 1)
        store.load_optimizer(t, m, v)
        feed(plan, g.spec, g.batch(s), g.draws(s))
        plan.train_step(g.lr)
        got = plan.losses()
        for k, val in g.exp(s, "loss").items():
            close(got[k], val, LOSS_RTOL, 1e-6, f"{case} step{s} loss {k}")
        exp_grads, gn = g.exp(s, "grad"), g.get(f"exp/{s}/grad_norm")
        close(store.ctrl[5], gn, 1e-4, 1e-7, "grad norm")
        if not fused:
            for k, val in exp_grads.items():
                # split-bf16 forward/backward of the wide layers: ~1e-5 of the tensor's scale per element
                close(store.g(k), val, 1e-3, 2e-6 + 3e-5 * float(val.abs().max()), f"{case} step{s} grad {k}")
        st = store.state_dict()
        for k, val in g.exp(s, "state").items():
            close(st[k], val, 2e-4, noise_atol(exp_grads.get(k), gn, g.lr, 3e-6), f"{case} step{s} state {k}")
        for k, val in g.exp(s, "m").items():
            close(store.m(k), val, 1e-3, 3e-7, f"{case} step{s} exp_avg {k}")
        for k, val in g.exp(s, "v").items():
            close(store.v(k), val, 2e-3, 1e-9, f"{case} step{s} exp_avg_sq {k}")


@pytest.mark.parametrize("case", MODEL_CASES)
def test_free_running_trajectory_and_validation(case):
    from flexynesis_amd.engine import ParamStore, StepPlan
    g = Golden(case)
    spec = arch_from_golden(g)
    B = next(iter(g.batch(0)["y"].values())).shape[0]
    store = ParamStore(spec, _dev(), big_threshold=512)
    store.load_state(g.state0())
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    for s in range(g.n_steps):
        feed(plan, g.spec, g.batch(s), g.draws(s))
        plan.train_step(g.lr)
        got = plan.losses()
        for k, val in g.exp(s, "loss").items():
            close(got[k], val, 1e-4, 2e-6, f"{case} free-run step{s} loss {k}")       # the north-star gate, free-running over the golden's 2-3 steps
    # num_batches_tracked bookkeeping (triplet encoders see 3 BN passes per step, triplet_encoder.py:153-155)
    final = g.exp(g.n_steps - 1, "state")
    sd = store.state_dict()
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(final[k]), k
    # validation_step arithmetic (eval-mode BN/dropout, unweighted sum) from the reference's final state
    store.load_state(final)
    ev = StepPlan(store, B, train=False, supplied_draws=True)
    feed(ev, g.spec, g.batch(0), g.sub("draws/val"))
    ev.forward()
    got = ev.losses()
    for k, val in g.sub("exp/val/loss").items():
        close(got[k], val, LOSS_RTOL, 1e-6, f"{case} val {k}")


# ---------------------------------------------------------------------------------------------------
# HIP engine vs the CPU oracle at larger, BASELINE-shaped sizes
# ---------------------------------------------------------------------------------------------------
def _oracle_spec(aspec):
    from oracle.restate import Spec
    return Spec(aspec.model, list(aspec.layers), aspec.latent_dim, aspec.hidden_dim_factor,
                aspec.supervisor_hidden_dim, list(aspec.variables), aspec.surv_event_var, aspec.surv_time_var,
                aspec.use_loss_weighting, aspec.input_layers, aspec.output_layers)


@pytest.mark.parametrize("model,layers,B", [
    ("DirectPred", [("gex", 5000)], 32),                                   # BASELINE cfg1 shape
    ("DirectPred", [("gex", 4000), ("cnv", 3000)], 128),                   # cfg2 family (scaled to oracle-seconds)
    ("DirectPred", [("all", 7000)], 100),                                  # --fusion_type early: one layer "all", B % 32 != 0
    ("DirectPred", [("gex", 3000), ("covariates", 6)], 64),                # covariates modality: hidden = max(int(6*.25), 2)
    ("DirectPred", [("gex", 4099), ("cnv", 3001)], 37),                    # odd feature counts / batch: unaligned rows everywhere
    ("DirectPred", [("gex", 2500), ("cnv", 1800)], 200),                   # B > 128: per-layer heads / loop BatchNorm fallbacks
    ("supervised_vae", [("gex", 2051), ("cnv", 1403)], 50),                # same for the VAE family (decoder outputs of odd width)
    ("supervised_vae", [("gex", 2000), ("cnv", 1600)], 64),                # cfg3 family
    ("MultiTripletNetwork", [("gex", 1500), ("cnv", 1200), ("meth", 900)], 32),   # cfg4 family
    ("CrossModalPred", [("gex", 2000), ("cnv", 1600), ("meth", 1200)], 64),      # section 8(f): encode gex+cnv, decode meth+gex
])
def test_engine_vs_oracle_three_steps(model, layers, B):
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = _dev()
    if model == "DirectPred" and len(layers) == 1:
        variables = [("y", "numerical", 1)]
        surv = (None, None)
    elif model == "DirectPred":
        variables = [("y", "numerical", 1), ("c", "categorical", 4)]
        surv = (None, None)
    elif model == "supervised_vae":
        variables = [("c", "categorical", 4), ("event", "numerical", 1)]
        surv = ("event", "time")
    elif model == "CrossModalPred":
        variables = [("y", "numerical", 1), ("c", "categorical", 4)]
        surv = (None, None)
    else:
        variables = [("c", "categorical", 4), ("y", "numerical", 1)]
        surv = (None, None)
    io = (["gex", "cnv"], ["meth", "gex"]) if model == "CrossModalPred" else (None, None)
    aspec = ArchSpec(model, layers, 64, 0.25, 16, variables, surv[0], surv[1], True, io[0], io[1])
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, 512, seed=1234)
    st = O.init_state(ospec, seed=3)
    store = ParamStore(aspec, dev)
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    gen = torch.Generator().manual_seed(99)
    opt, lr = {}, 1e-3
    for step in range(3):
        if step > 0:
            # per-step parity: every step starts from the oracle's state and Adam
            # moments.  Free-running trajectories diverge chaotically through Adam in ANY implementation
            # (fp32 engine vs fp32 oracle: 2e-5 after 5 steps; the reference 1 vs 8 threads: 4.6e-5 after 7).
            store.load_state(st)
            store.reset_optimizer()
            store.load_optimizer(opt["t"], opt["m"], opt["v"])
        idx = torch.randperm(512, generator=gen)
        y = {k: ann[k][idx[:B]] for k in plan.y}
        draws = {}
        for name, t in plan.draws.items():
            if name == "eps" or name.startswith("prior."):
                draws[name] = torch.randn(t.shape, generator=gen)
            else:
                draws[name] = (torch.rand(t.shape, generator=gen) < 0.9).float()
        if model == "MultiTripletNetwork":
            parts = [[dat[n][idx[j * B:(j + 1) * B]] for n, _ in layers] for j in range(3)]
            batch = {"anchor": parts[0], "positive": parts[1], "negative": parts[2], "y": y}
            plan.set_batch(parts=[[x.to(dev) for x in p] for p in parts], y={k: v.to(dev) for k, v in y.items()})
        else:
            xs = [dat[n][idx[:B]] for n, _ in layers]
            batch = {"x": xs, "y": y}
            plan.set_batch(x_list=[x.to(dev) for x in xs], y={k: v.to(dev) for k, v in y.items()})
        plan.set_draws({k: v.to(dev) for k, v in draws.items()})
        plan.train_step(lr)
        st_prev = st
        st, opt, info = O.train_step(ospec, st, opt, batch, draws, lr)
        got = plan.losses()
        for k, v in info["losses"].items():
            close(got[k], v, 2e-5, 1e-6, f"{model} step{step} loss {k}")      # gate is 1e-4; observed <= 5e-7
        # torch-CPU's fp32 vector_norm over millions of elements is itself 2e-4..5e-4 off the exact value
        # (measured at [1750,7000]: engine 20.532915 == fp64 oracle 20.532911, fp32 oracle 20.52206;
        # DESIGN.md section 3), so the global norm is checked tightly against the fp64 norm of the oracle's
        # fp32 gradients and only loosely against the oracle's own fp32 reduction.
        exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
        close(store.ctrl[5], exact, 1e-4, 1e-7, "grad_norm vs fp64 norm of the oracle's gradients")    # observed <= 3e-5
        close(store.ctrl[5], info["grad_norm"], 1e-3, 1e-7, "grad_norm")
        sd = store.state_dict()
        for k in store.big_keys:      # wide weights after this step, element-wise
            # Adam's update is ~lr * sign(g) while v is young, so an element whose gradient is below the
            # arithmetic noise floor (~3e-5 of the tensor's scale for split-bf16) may legitimately move by
            # +-lr in either implementation; everything else must agree tightly.
            # Two legitimate sources of isolated element-level differences (both exist between any two
            # implementations, incl. the reference at 1 vs 8 threads, and neither moves a loss):
            #  (a) gradients below the arithmetic noise floor take +-lr Adam steps of either sign;
            #  (b) a pre-activation within ~1e-5 of zero can switch its ReLU gate, which changes ONE row of dW
            #      discretely (measured: 1 gate of 40000 differs between the f32 and split-bf16 forward).
            # So: >= 99.9 % of the elements must agree tightly and the whole update must agree in norm.
            a, b_ = sd[k].double(), st[k].double()
            bad = (a - b_).abs() > 2e-5 + 1e-3 * b_.abs()
            assert float(bad.double().mean()) <= 1e-3, f"{model} {k} step{step}: {int(bad.sum())} elements differ"
            upd_ref = b_ - st_prev[k].double()
            assert float((a - b_).norm() / upd_ref.norm()) <= 2e-2, f"{model} {k} step{step}: update norm mismatch"


@pytest.mark.parametrize("n_out,k_in,B", [(1100, 2500, 96), (2600, 700, 64), (130, 2100, 32), (2050, 2050, 128)])
def test_dw_adam_xcd_tile_order_is_bit_identical(n_out, k_in, B):
    """The XCD-partitioned, L2-blocked tile order of the fused dW+Adam kernel (used when an operand outgrows one L2) only
    changes WHICH workgroup computes a tile: forced on (tile_order = 2) it must reproduce the linear order bit for bit,
    including ragged edges and the padded part of the index space."""
    import os
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(n_out + k_in)
    dy = (torch.randn(B, n_out, generator=g) * 1e-2).to(dev)
    x = torch.randn(B, k_in, generator=g).to(dev)
    ldw = (k_in + 31) // 32 * 32
    W0 = torch.randn(n_out, ldw, generator=g).to(dev)
    m0, v0 = torch.randn(n_out, ldw, generator=g).to(dev) * 1e-3, torch.rand(n_out, ldw, generator=g).to(dev) * 1e-5
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 3.0
    ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
    ctrl[4] = 0.8
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    outs = []
    for order in (1, 2):              # 1 = linear, 2 = XCD-partitioned: explicit arguments of fx_linear_dw_adam_bf16x3_ex
        W, m, v = W0.clone(), m0.clone(), v0.clone()
        ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl,
                                  tile_order=order)
        torch.cuda.synchronize()
        outs.append((W, m, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not torch.equal(outs[0][0][:, :k_in], W0[:, :k_in])            # the step did something
    assert torch.equal(outs[0][0][:, k_in:], W0[:, k_in:])                # the row padding is untouched


# ---------------------------------------------------------------------------------------------------
# split-bf16 (bf16x3) wide-layer kernels
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 5000, 20000), (128, 300, 1000), (384, 130, 2500), (26, 70, 100), (8, 16, 64),
                                   # the stacked-rows kernel (X fragments in registers; K blocks in threes, surplus blocks multiply zeros): 2 and
                                   # 3 M tiles, ragged rows / columns, 1 .. 5 blocks per slice, more than one row group, a K that is no multiple of 4
                                   (200, 300, 1000), (256, 129, 96), (300, 257, 4099), (500, 130, 2500), (384, 64, 32),
                                   (384, 200, 64), (129, 128, 160), (384, 1500, 6000),
                                   # its four-wave form for at most 64 rows
                                   (64, 257, 4099), (33, 5000, 20000), (1, 300, 1000), (48, 129, 96), (64, 1500, 6000), (17, 128, 32)])
def test_linear_fwd_bf16x3_vs_fp64(M, N, K):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    hi, lo = ops.new_split_kb(M, K, dev)
    ops.split_bf16(ops.IMMEDIATE, hi, lo, x)
    Kp = ops.pad32(K)
    back = ops.unblock(hi, M, Kp).float() + ops.unblock(lo, M, Kp).float()       # K-blocked [Kp/32, pad128(M), 32]
    assert float(back[:, :K].sub(x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    assert float(back[:, K:].abs().sum()) == 0.0 and float(hi[:, M:].float().abs().sum()) == 0.0
    y = torch.full((M, N), float("nan"), device=dev)
    ops.linear_fwd_bf16x3(ops.IMMEDIATE, y, hi, lo, W, b, ops.Workspace(dev))
    ref = x.double() @ W.double().t() + b.double()
    scale = (x.double().abs() @ W.double().abs().t())
    err = ((y.double() - ref).abs() / scale).max().item()
    assert err <= 2e-5, err                      # per-product error <= ~2^-16, far smaller after averaging
    rel = ((y.double() - ref).norm() / ref.norm()).item()
    assert rel <= 1e-5, rel


def test_linear_dw_adam_bf16x3_vs_fp32_path():
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B, n_out, k_in = 100, 300, 1100                 # batch not a multiple of 32: zero padding path
    dy = (torch.randn(B, n_out, generator=g) * 1e-2).to(dev)
    x = torch.randn(B, k_in, generator=g).to(dev)
    W0 = torch.randn(n_out, k_in, generator=g).to(dev)
    m0, v0 = torch.randn(n_out, k_in, generator=g).to(dev) * 1e-3, torch.rand(n_out, k_in, generator=g).to(dev) * 1e-5
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 9.0
    ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
    ctrl[4] = 0.5
    W1, m1, v1 = W0.clone(), m0.clone(), v0.clone()
    ops.linear_dw_adam(ops.IMMEDIATE, W1, m1, v1, dy, x, ctrl)
    W2, m2, v2 = W0.clone(), m0.clone(), v0.clone()
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    assert torch.allclose((dyt[0].float() + dyt[1].float())[:, :B], dy.t(), rtol=0, atol=2.0 ** -16 * float(dy.abs().max()))
    ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W2, m2, v2, dyt[0], dyt[1], xt[0], xt[1], ctrl)
    gref = (dy.double().t() @ x.double()) * 0.5
    close(m2, 0.9 * m0.double() + 0.1 * gref, 1e-4, 2e-7, "bf16x3 exp_avg")
    close(m2, m1, 1e-4, 2e-7, "bf16x3 vs f32 exp_avg")
    close(W2, W1, 1e-5, 2e-6, "bf16x3 vs f32 weights")


@pytest.mark.parametrize("case", MODEL_CASES)
def test_train_step_exact_fp32_mode(case):
    """precision='f32' keeps every contraction on the exact-fp32 MFMA (the round-1 first path)."""
    from flexynesis_amd.engine import ParamStore, StepPlan
    g = Golden(case)
    spec = arch_from_golden(g)
    B = next(iter(g.batch(0)["y"].values())).shape[0]
    store = ParamStore(spec, _dev(), big_threshold=512)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True, precision="f32")
    store.load_state(g.state0())
    feed(plan, g.spec, g.batch(0), g.draws(0))
    plan.train_step(g.lr)
    got = plan.losses()
    for k, val in g.exp(0, "loss").items():
        close(got[k], val, 2e-5, 1e-6, f"{case} f32 loss {k}")
    st = store.state_dict()
    exp_grads, gn = g.exp(0, "grad"), g.get("exp/0/grad_norm")
    for k, val in g.exp(0, "state").items():
        close(st[k], val, 2e-4, noise_atol(exp_grads.get(k), gn, g.lr, 3e-6), f"{case} f32 state {k}")


# ---------------------------------------------------------------------------------------------------
# launch-fusion kernels (fx_fused_small.hip)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,F,src_cols", [(100, 1000, 1000),      # R not a multiple of 32: padded rows must be zero
                                          (384, 4099, 4099),      # stacked triplet rows; F % 4 != 0: the scalar path, ragged last tile
                                          (7, 30, 30),            # smaller than one tile
                                          (128, 996, 1001),       # F % 4 == 0 but the source row pitch is not: scalar loads, vector stores off
                                          (33, 4224, 4224)])      # F a multiple of the 128-column tile
def test_gather_split_matches_gather_plus_splits(R, F, src_cols):
    from flexynesis_amd import ops
    dev = _dev()
    src = torch.randn(300, src_cols, device=dev)[:, :F]
    idx = torch.randint(0, 300, (3 * R,), device=dev)
    ctrl = torch.zeros(64, device=dev)
    ctrl[8] = 1.0
    x = torch.empty(R, F, device=dev)
    sp, spt = ops.new_split_kb(R, F, dev), ops.new_split(F, R, dev)
    for t in spt:
        t.fill_(7.0)
    ops.gather_split(ops.IMMEDIATE, x, sp[0], sp[1], spt[0], spt[1], src, idx, ctrl, R)
    ref = src[idx[R:2 * R]]
    assert torch.equal(x, ref)
    hi, lo = ops.new_split_kb(R, F, dev)
    ops.split_bf16(ops.IMMEDIATE, hi, lo, ref)
    assert torch.equal(sp[0], hi) and torch.equal(sp[1], lo)
    # K-blocked layout: element (r, c) at [c // 32, r, c % 32]; x == hi + lo to 2^-16; padding stays zero
    rec = ops.unblock(hi, R, F).float() + ops.unblock(lo, R, F).float()
    assert float((rec - ref).abs().max()) <= 2.0 ** -15 * float(ref.abs().max())
    assert torch.equal(ops.unblock(hi, R, F), ref.to(torch.bfloat16))
    assert float(hi[:, R:].float().abs().sum()) == 0.0 and float(ops.unblock(hi, R, ops.pad32(F))[:, F:].float().abs().sum()) == 0.0
    hit, lot = ops.new_split(F, R, dev)
    ops.split_bf16_t(ops.IMMEDIATE, hit, lot, ref)
    assert torch.equal(spt[0], hit) and torch.equal(spt[1], lot)
    assert float(spt[0][:, R:].float().abs().sum()) == 0.0


def test_gram_hadamard_from_slabs():
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, H, F = 128, 900, 4000
    dy, x = (torch.randn(B, H, generator=g) * 1e-2).to(dev), torch.randn(B, F, generator=g).to(dev)
    nx, nd = int(ops.lib.fx_gemm_splitk(B, B, F)), int(ops.lib.fx_gemm_splitk(B, B, H))
    sx, sd = torch.empty(nx, B * B, device=dev), torch.empty(nd, B * B, device=dev)
    assert ops.gemm_slabs(ops.IMMEDIATE, ops.GEMM_NT, sx, x, x, B, B) == nx
    ops.gemm_slabs(ops.IMMEDIATE, ops.GEMM_NT, sd, dy, dy, B, B)
    nb = ops.gram_hadamard_blocks(B * B)
    slots = torch.zeros(nb, dtype=torch.float64, device=dev)
    ops.gram_hadamard(ops.IMMEDIATE, slots, sx, nx, sd, nd, B * B)
    ref = float(((dy.double().t() @ x.double()) ** 2).sum())
    assert abs(float(slots.sum()) - ref) <= 3e-6 * ref


@pytest.mark.parametrize("B,C,pre,post", [(128, 5000, 0, 2), (100, 700, 1, 0)])
def test_bn_fed_by_splitk_slabs(B, C, pre, post):
    """fx_bn_act_fwd_slabs == reduce(slabs)+bias followed by fx_bn_act_fwd (bitwise: same arithmetic order)."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + C)
    ns = 5
    slabs = torch.randn(ns, B * C, generator=g).to(dev)
    bias, gamma, beta = (torch.randn(C, generator=g).to(dev) for _ in range(3))
    mask = (torch.rand(B, C, generator=g) < 0.9).float().to(dev)
    rm1, rv1, rm2, rv2 = (torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev),
                          torch.ones(C, device=dev))
    x_ref = bias.clone().expand(B, C).clone()
    for z in range(ns):
        x_ref = x_ref + slabs[z].view(B, C)
    out1, out2, x_out = (torch.empty(B, C, device=dev) for _ in range(3))
    sm1, si1, sm2, si2 = (torch.empty(C, device=dev) for _ in range(4))
    drop = 0.1 if post == 2 else 0.0
    ops.bn_act_fwd(ops.IMMEDIATE, out1, x_ref, gamma, beta, rm1, rv1, sm1, si1, pre, post, True, drop,
                   mask=mask if post == 2 else None)
    ops.bn_act_fwd_slabs(ops.IMMEDIATE, out2, x_out, slabs, ns, B * C, bias, gamma, beta, rm2, rv2, sm2, si2, pre, post,
                         True, drop, mask=mask if post == 2 else None)
    close(x_out, x_ref, 1e-6, 1e-6, "slab sum")
    close(out2, out1, 1e-5, 1e-5, "bn from slabs")
    close(rv2, rv1, 1e-5, 1e-6, "running_var")


@pytest.mark.parametrize("B,L", [(128, 64), (100, 128), (37, 16)])
def test_fused_heads_match_per_layer_kernels(B, L):
    """fx_heads_fwd / fx_heads_bwd (all supervisor heads in one launch each way) against the per-layer GEMM +
    BatchNorm kernels: same Philox dropout stream, so masks are identical and everything else agrees to rounding."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    dev = _dev()
    variables = [("y", "numerical", 1), ("c", "categorical", 5), ("event", "numerical", 1)]
    spec = ArchSpec("DirectPred", [("gex", 300), ("cnv", 200)], L, 0.3, 24, variables, "event", "time", True)
    torch.manual_seed(B + L)
    init = ParamStore(spec, dev).state_dict()
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(B, 300, generator=g), torch.randn(B, 200, generator=g)]
    y = {"y": torch.randn(B, generator=g), "c": torch.randint(0, 5, (B,), generator=g).float(),
         "event": (torch.rand(B, generator=g) < 0.5).float(), "time": torch.rand(B, generator=g) * 10}
    y["y"][3] = float("nan")
    y["c"][5] = -1.0
    res = []
    for fuse in (True, False):
        store = ParamStore(spec, dev, materialize_big_grads=True)
        store.load_state(init)
        plan = StepPlan(store, B, train=True, fused=False, seed=77, fuse_heads=fuse)
        assert plan._heads_fusable(plan.embeddings) == fuse
        plan.set_batch(x_list=[x.to(dev) for x in xs], y={k: v.to(dev) for k, v in y.items()})
        plan.train_step(1e-3)
        res.append((plan.losses(), {k: store.g(k).clone() for k in store.small_keys},
                    store.state_dict(), plan.buf["MLPs.c/a1"].clone()))
    (l1, g1, s1, a1), (l0, g0, s0, a0) = res
    assert torch.equal(a1 == 0, a0 == 0), "dropout/ReLU pattern differs: the fused kernel is not on the same Philox stream"
    for k in l0:
        close(l1[k], l0[k], 2e-6, 1e-7, f"loss {k}")
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:      # biases in front of a BatchNorm have a true gradient of 0: both paths hold rounding noise there
        scale = float(g0[k].abs().max())
        close(g1[k], g0[k], 1e-4, 2e-6 * scale + 1e-6 * gmax, f"grad {k}")
    for k in s0:
        if k.endswith("running_mean") or k.endswith("running_var"):
            close(s1[k], s0[k], 1e-5, 1e-7, k)


@pytest.mark.parametrize("model,frozen", [("DirectPred", ("encoders.",)), ("DirectPred", ("MLPs.",)),
                                          ("supervised_vae", ("encoders.",)), ("supervised_vae", ("MLPs.",))])
def test_finetune_step_frozen_groups_no_clip_vs_oracle(model, frozen):
    """FineTuner step (reference main.py:530-539, :562-566, :591-600; pinned live in test_oracle_pinning.py): frozen
    groups are bit-identical after the step, everything else matches the oracle's unclipped Adam step, BatchNorm
    buffers of frozen blocks still move."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = _dev()
    layers = [("gex", 2400), ("cnv", 1800)]
    variables = [("y", "numerical", 1), ("c", "categorical", 4)]
    aspec = ArchSpec(model, layers, 32, 0.5, 16, variables, None, None, True)
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, 256, seed=7)
    st0 = O.init_state(ospec, seed=5)
    B, lr = 64, 1e-3
    store = ParamStore(aspec, dev)
    store.load_state(st0)
    with pytest.raises(ValueError):
        StepPlan(store, B, train=True, fused=True, supplied_draws=True, frozen=frozen)        # frozen + clip: not a reference setup
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True, clip=False, frozen=frozen)
    names = {c[1] for c in plan.t_bwd.calls} | {c[1] for c in plan.t_opt.calls}
    if frozen == ("encoders.",):
        assert "fx_linear_dw_adam_bf16x3" not in names or model == "supervised_vae"      # svae decoders are still trained
    gen = torch.Generator().manual_seed(3)
    xs = [dat[n][:B] for n, _ in layers]
    y = {k: ann[k][:B] for k in plan.y}
    draws = {}
    for name, t in plan.draws.items():
        draws[name] = torch.randn(t.shape, generator=gen) if (name == "eps" or name.startswith("prior.")) \
            else (torch.rand(t.shape, generator=gen) < 0.9).float()
    plan.set_batch(x_list=[x.to(dev) for x in xs], y={k: v.to(dev) for k, v in y.items()})
    plan.set_draws({k: v.to(dev) for k, v in draws.items()})
    plan.train_step(lr)
    st1, _, info = O.train_step(ospec, st0, {}, {"x": xs, "y": y}, draws, lr, clip=False, frozen=frozen)
    got = plan.losses()
    for k, v in info["losses"].items():
        close(got[k], v, 2e-5, 1e-6, f"loss {k}")
    sd = store.state_dict()
    gn = float(info["grad_norm"])
    moved_buffers = 0
    for k in sd:
        if k.endswith("num_batches_tracked"):
            continue
        if k.startswith(frozen) and not O.is_buffer(k):
            assert torch.equal(sd[k].cpu(), st0[k]), f"frozen {k} changed"
        elif O.is_buffer(k):
            close(sd[k], st1[k], 1e-4, 1e-6, k)
            moved_buffers += int(k.startswith(frozen) and not torch.equal(st1[k], st0[k]))
        else:
            g = info["grads"].get(k)
            if k in store.big_keys:
                bad = (sd[k].cpu().double() - st1[k].double()).abs() > 2e-5 + 1e-3 * st1[k].double().abs()
                assert float(bad.double().mean()) <= 1e-3, k
            else:
                close(sd[k], st1[k], 1e-4, noise_atol(g, gn, lr, 2e-6), k)
    assert moved_buffers > 0, "BatchNorm running statistics of the frozen blocks must still update (train mode)"


@pytest.mark.parametrize("model", ["DirectPred", "supervised_vae"])
@pytest.mark.parametrize("freeze", ["enc_frozen", "sup_frozen"])
@pytest.mark.parametrize("fused", [True, False])
def test_finetune_step_matches_reference_fixture(model, freeze, fused):
    """The engine's frozen plans against the FineTuner steps recorded from the reference itself (tests/golden/finetune_step.npz;
    main.py:530-539, :562-566, :591-600): two consecutive unclipped steps per freeze configuration, losses, grad norm, frozen
    groups bit-identical, trainable parameters and BatchNorm buffers after the step."""
    from golden_io import FinetuneGolden
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    G = FinetuneGolden(model)
    gs, frozen, dev = G.spec, FinetuneGolden.FREEZES[freeze], _dev()
    aspec = arch_from_golden(G)
    st0 = G.state0()
    store = ParamStore(aspec, dev, big_threshold=512)
    store.load_state(st0)
    B = G.batch(0)["x"][0].shape[0]
    plan = StepPlan(store, B, train=True, fused=fused, supplied_draws=True, clip=False, frozen=frozen)
    for s in range(2):
        exp = G.step(freeze, s)
        feed(plan, gs, G.batch(s), exp["draws"])
        plan.train_step(G.lr)
        close(plan.losses()["total"], exp["total"], LOSS_RTOL, 1e-6, f"{model} {freeze} step{s} total")
        gn = float(exp["grad_norm"])
        sd = store.state_dict()
        for k, v in exp["state"].items():
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(v), k
            elif k.startswith(frozen) and not O.is_buffer(k):
                assert torch.equal(sd[k].cpu(), st0[k]), f"frozen {k} changed"
            elif s == 0:                       # (free-running from here on: the second step checks the loss and the frozen groups)
                close(sd[k], v, 2e-4, noise_atol(exp["grads"].get(k), gn, G.lr, 3e-6), f"{model} {freeze} state {k}")


@pytest.mark.parametrize("M,N,K", [(128, 5000, 20000), (100, 330, 1000), (37, 64, 96), (128, 1250, 5000)])
def test_linear_bwd_x_bf16x3_vs_fp64(M, N, K):
    """dX = dY . W with W stored [K, N] (the wide data-gradient contraction): split-bf16 kernel vs fp64."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    hi, lo = ops.new_split_kb(M, K, dev)
    ops.split_bf16(ops.IMMEDIATE, hi, lo, dy)
    dx = torch.full((M, N), float("nan"), device=dev)
    ops.linear_bwd_x_bf16x3(ops.IMMEDIATE, dx, hi, lo, W, ops.Workspace(dev))
    ref = dy.double() @ W.double()
    scale = dy.double().abs() @ W.double().abs()
    assert ((dx.double() - ref).abs() / scale).max().item() <= 2e-5
    assert ((dx.double() - ref).norm() / ref.norm()).item() <= 1e-5


def _random_config(seed):
    """A random point of the reference's search space (config.py:7-15) x model class x layer/head layout."""
    rng = np.random.default_rng(seed)
    model = ["DirectPred", "supervised_vae", "MultiTripletNetwork", "CrossModalPred"][seed % 4]
    n_layers = int(rng.integers(1, 4)) if model != "CrossModalPred" else 3
    names = ["gex", "cnv", "meth"][:n_layers]
    layers = [(n, int(rng.integers(40, 900))) for n in names]
    heads = []
    if model == "MultiTripletNetwork" or rng.random() < 0.6:
        heads.append(("c", "categorical", int(rng.integers(2, 6))))
    if rng.random() < 0.7 or not heads:
        heads.append(("y", "numerical", 1))
    surv = (None, None)
    if model != "MultiTripletNetwork" and rng.random() < 0.4:
        heads.append(("event", "numerical", 1))
        surv = ("event", "time")
    io = (None, None)
    if model == "CrossModalPred":
        io = (list(rng.permutation(names)[: int(rng.integers(1, 4))]), list(rng.permutation(names)[: int(rng.integers(1, 4))]))
    return dict(model=model, layers=layers, latent=int(rng.integers(4, 129)), factor=float(rng.uniform(0.2, 0.5)),
                sup=int(rng.integers(2, 33)), heads=heads, surv=surv, io=io, B=int(rng.choice([8, 17, 32, 64, 128])),
                weighting=bool(rng.random() < 0.8))


@pytest.mark.parametrize("seed", list(range(32)))
def test_random_configurations_one_step_vs_oracle(seed):
    """Seeded sweep (32 committed seeds; 96 more were run once, all green) over model class x layer count x widths x head layout x batch size: one optimisation step of the
    HIP engine against the CPU oracle from identical state, inputs and random draws (everything is in the small-
    parameter regime here, so all tensors are compared element-wise)."""
    _one_step_vs_oracle(_random_config(seed), seed)


_EXTRA_WIDE = [int(v) for v in __import__("os").environ.get("FX_TEST_EXTRA_SEEDS", "").split(",") if v]      # widen the sweep by hand


@pytest.mark.parametrize("seed", list(range(100, 112)) + _EXTRA_WIDE)
def test_random_wide_configurations_one_step_vs_oracle(seed):
    """(12 committed seeds; 100 more -- FX_TEST_EXTRA_SEEDS=200..259,300..339 -- were run once, all green.)
    The same sweep with WIDE layers (2049 .. 9000 features of any residue mod 4 / 32, hidden_dim_factor up to 0.5): the
    first layers exceed 2^20 elements and take the split-bf16 kernels -- wide forward (1-3 M tiles, the stacked triplet
    rows through the LDS-DMA kernel), fused dW + clip + Adam, the block backward per BatchNorm pass, the Gram norm, the
    decoders' data gradient -- at batch sizes that are not multiples of 32."""
    c = _random_config(seed)
    rng = np.random.default_rng(seed + 7)
    c["layers"] = [(n, int(rng.integers(2049, 9001))) for n, _ in c["layers"]]
    c["factor"] = float(rng.uniform(0.3, 0.5))
    c["B"] = int(rng.choice([8, 17, 32, 64, 100, 128]))
    _one_step_vs_oracle(c, seed, expect_big=True)


def _one_step_vs_oracle(c, seed, expect_big=False):
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = _dev()
    aspec = ArchSpec(c["model"], c["layers"], c["latent"], c["factor"], c["sup"], c["heads"], c["surv"][0], c["surv"][1],
                     c["weighting"], c["io"][0], c["io"][1])
    ospec = _oracle_spec(aspec)
    B, lr = c["B"], 1e-3
    rows = 3 * B if c["model"] == "MultiTripletNetwork" else B
    dat, ann = O.synthetic_cohort(c["layers"], max(rows, 16), seed=seed)
    ann["event"] = (torch.rand(max(rows, 16), generator=torch.Generator().manual_seed(seed)) < 0.6).float()
    st0 = O.init_state(ospec, seed=seed + 1)
    store = ParamStore(aspec, dev)
    store.load_state(st0)
    if expect_big:
        assert len(store.big_keys) >= 1, "the wide sweep must reach the split-bf16 kernels"
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    # which schedule ran: every one-pass model of <= 4 modalities and latent <= 128 takes the grouped encoder tails whatever its
    # widths are mod 4 (hidden / input widths are rounded up inside the engine, the tail Linears' rows are allocated up to 4)
    n_enc = len(aspec.enc_idx) if aspec.is_vae else len(c["layers"])
    assert plan.path["grouped_tails"] == (c["model"] != "MultiTripletNetwork" and B <= 128 and n_enc * c["latent"] <= 512), (plan.path, c)
    gen = torch.Generator().manual_seed(seed + 2)
    y = {k: ann[k][:B].clone() for k in plan.y}
    if "y" in y and B > 4:
        y["y"][1] = float("nan")                       # a missing numerical label
    if "c" in y and B > 4:
        y["c"] = torch.minimum(y["c"], torch.tensor(float(dict((h[0], h[2]) for h in c["heads"])["c"] - 1)))
        y["c"][2] = -1.0                               # a missing categorical label
    draws = {}
    for name, t in plan.draws.items():
        draws[name] = torch.randn(t.shape, generator=gen) if (name == "eps" or name.startswith("prior.")) \
            else (torch.rand(t.shape, generator=gen) < 0.9).float()
    if c["model"] == "MultiTripletNetwork":
        parts = [[dat[n][j * B:(j + 1) * B] for n, _ in c["layers"]] for j in range(3)]
        batch = {"anchor": parts[0], "positive": parts[1], "negative": parts[2], "y": y}
        plan.set_batch(parts=[[x.to(dev) for x in p] for p in parts], y={k: v.to(dev) for k, v in y.items()})
    else:
        xs = [dat[n][:B] for n, _ in c["layers"]]
        batch = {"x": xs, "y": y}
        plan.set_batch(x_list=[x.to(dev) for x in xs], y={k: v.to(dev) for k, v in y.items()})
    plan.set_draws({k: v.to(dev) for k, v in draws.items()})
    plan.train_step(lr)
    st1, _, info = O.train_step(ospec, st0, {}, batch, draws, lr)
    got = plan.losses()
    for k, v in info["losses"].items():
        close(got[k], v, LOSS_RTOL, 1e-6, f"{c['model']} seed {seed} loss {k}")
    exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
    close(store.ctrl[5], exact, 2e-4, 1e-7, "grad_norm")
    sd = store.state_dict()
    gn = float(info["grad_norm"])
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(st1[k]), k
        elif O.is_buffer(k):
            close(sd[k], st1[k], 1e-4, 1e-6, k)
        elif k in store.big_keys:
            bad = (sd[k].cpu().double() - st1[k].double()).abs() > 2e-5 + 1e-3 * st1[k].double().abs()
            assert float(bad.double().mean()) <= (2e-3 if expect_big else 1e-3), k
        elif expect_big:
            # Adam's first step moves every entry by lr * sign(g): where |g| is at the rounding level of ITS computation
            # (split-bf16 contractions upstream: ~1e-5 of the tensor's gradient scale, far above 1e-6 of the global norm)
            # the sign is implementation-defined -- a few entries per tensor land 2 lr apart; everything else must agree
            a, b = sd[k].detach().double().cpu().reshape(-1), st1[k].double().reshape(-1)
            err = (a - b).abs()
            tol = noise_atol(info["grads"].get(k), gn, lr, 2e-6)
            tol = (tol.double().reshape(-1) if torch.is_tensor(tol) else tol) + 1e-4 * b.abs()
            bad = err > tol
            assert int(bad.sum()) <= max(1, int(2e-3 * a.numel())) and float(err.max()) <= 2.1 * lr + 1e-4 * float(b.abs().max()), \
                (f"{c['model']} seed {seed} {k}: {int(bad.sum())}/{a.numel()} off, max err {float(err.max()):.3e}")
        else:
            close(sd[k], st1[k], 1e-4, noise_atol(info["grads"].get(k), gn, lr, 2e-6), f"{c['model']} seed {seed} {k}")


@pytest.mark.parametrize("model", ["DirectPred", "supervised_vae"])
def test_free_running_training_curve_tracks_oracle(model):
    """40 un-resynchronised optimisation steps from the same initial state, batches and random draws: element-level
    trajectories diverge chaotically through Adam in ANY two implementations, but the
    training curve must not: every step's total loss stays within 2 % of the oracle's and both learn."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = _dev()
    layers = [("gex", 1500), ("cnv", 1100)]
    variables = [("y", "numerical", 1), ("c", "categorical", 4)]
    aspec = ArchSpec(model, layers, 32, 0.8, 16, variables, None, None, True)        # hidden 1200 x 1500: wide path
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, 512, seed=11)
    st = O.init_state(ospec, seed=2)
    store = ParamStore(aspec, dev)
    store.load_state(st)
    B, lr, steps = 64, 2e-3, 40
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    assert store.big_keys, "the wide (split-bf16, fused dW+Adam) path must be exercised"
    gen = torch.Generator().manual_seed(5)
    opt, ref_curve, got_curve = {}, [], []
    for s in range(steps):
        idx = torch.randperm(512, generator=gen)[:B]
        xs = [dat[n][idx] for n, _ in layers]
        y = {k: ann[k][idx] for k in plan.y}
        draws = {}
        for name, t in plan.draws.items():
            draws[name] = torch.randn(t.shape, generator=gen) if (name == "eps" or name.startswith("prior.")) \
                else (torch.rand(t.shape, generator=gen) < 0.9).float()
        plan.set_batch(x_list=[x.to(dev) for x in xs], y={k: v.to(dev) for k, v in y.items()})
        plan.set_draws({k: v.to(dev) for k, v in draws.items()})
        plan.train_step(lr)
        got_curve.append(plan.losses()["total"])
        st, opt, info = O.train_step(ospec, st, opt, {"x": xs, "y": y}, draws, lr)
        ref_curve.append(float(info["losses"]["total"].reshape(-1)[0]))
    got, ref = np.array(got_curve), np.array(ref_curve)
    assert np.all(np.abs(got - ref) <= 2e-2 * np.abs(ref) + 1e-3), (np.abs(got - ref) / np.abs(ref)).max()
    assert abs(got[0] - ref[0]) <= 1e-4 * abs(ref[0])
    assert ref[-5:].mean() < 0.8 * ref[:5].mean() and got[-5:].mean() < 0.8 * got[:5].mean()


@pytest.mark.parametrize("Bp,passes,C", [(128, 3, 7500), (64, 3, 4100), (128, 1, 300), (96, 2, 1031)])
def test_block_bwd_k_blocked_split_equals_split_bf16(Bp, passes, C):
    """fx_block_bwd_ex: the K-blocked split of dY that every pass of stacked rows writes for its own rows is, bit for bit, what
    fx_split_bf16 makes of the fp32 dY the same launches wrote (the triplet network's dY dY^T product reads it: engine._mlp_bwd),
    pad rows and pad columns zero; every other output is untouched by the extra store."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(Bp + C)
    rows, L = Bp * passes, 32
    x = torch.randn(rows, C, device=dev, generator=g)
    out = torch.relu(torch.randn(rows, C, device=dev, generator=g))
    gamma = torch.rand(C, device=dev, generator=g) + 0.5
    W = torch.randn(L, C, device=dev, generator=g) / C ** 0.5
    dE = torch.randn(rows, L, device=dev, generator=g)
    sm = torch.randn(passes, C, device=dev, generator=g) * 0.1
    si = torch.rand(passes, C, device=dev, generator=g) + 0.5
    res = []
    for with_kb in (False, True):
        gW, gb = torch.zeros(L, C, device=dev), torch.zeros(L, device=dev)
        dg, db, dbias = (torch.zeros(C, device=dev) for _ in range(3))
        dy = torch.full((rows, C), float("nan"), device=dev)
        dyT = ops.new_split(C, rows, dev)
        kb = ops.new_split_kb(rows, C, dev)
        for p_ in range(passes):
            sl = slice(p_ * Bp, (p_ + 1) * Bp)
            ops.block_bwd(ops.IMMEDIATE, [(dE[sl], W, gW, gb)], x[sl], out[sl], gamma, sm[p_], si[p_], dg, db, dbias, 0, 2, 0.1, dy=dy[sl],
                          dyT=(dyT[0][:, p_ * Bp:], dyT[1][:, p_ * Bp:]) if Bp % 32 == 0 else None, accumulate=p_ > 0,
                          dy_kb=(kb[0], kb[1], p_ * Bp) if with_kb else None)
        torch.cuda.synchronize()
        res.append((dy, gW, gb, dg, db, dbias, dyT, kb))
    for a, b in zip(res[0][:6], res[1][:6]):
        assert torch.equal(a, b)
    if Bp % 32 == 0:
        assert torch.equal(res[0][6][0].view(torch.int16), res[1][6][0].view(torch.int16)) and torch.equal(res[0][6][1].view(torch.int16), res[1][6][1].view(torch.int16))
    ref = ops.new_split_kb(rows, C, dev)
    ops.split_bf16(ops.IMMEDIATE, ref[0], ref[1], res[1][0])
    torch.cuda.synchronize()
    kb = res[1][7]
    assert torch.equal(kb[0].view(torch.int16), ref[0].view(torch.int16)) and torch.equal(kb[1].view(torch.int16), ref[1].view(torch.int16))
    assert float(res[0][7][0].float().abs().sum()) == 0.0          # (without dy_kb nothing is written there)
    with pytest.raises(ops.FxError):
        ops.block_bwd(ops.IMMEDIATE, [(dE[:Bp], W, gW, gb)], x[:Bp], out[:Bp], gamma, sm[0], si[0], dg, db, dbias, 0, 2, 0.1,
                      dy_kb=(kb[0][:1], kb[1][:1], 0))


@pytest.mark.parametrize("B,C,Ls,pre,post", [(128, 5000, (64,), 0, 2), (100, 700, (64, 64), 1, 0), (37, 95, (17,), 0, 2),
                                              (128, 1250, (100, 33), 1, 0)])
def test_block_bwd_vs_autograd_fp64(B, C, Ls, pre, post):
    """fx_block_bwd (whole encoder-tail backward in one launch) against torch autograd in fp64: data gradient through
    the small Linears, BatchNorm(+ReLU+dropout | LeakyReLU) backward, the small Linears' weight/bias gradients, the
    transposed split-bf16 dY and the Gram share of the wide layer's gradient norm."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, C, generator=g, dtype=torch.float64)
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    mask = (torch.rand(B, C, generator=g) < 0.9).double()
    Ws = [torch.randn(L, C, generator=g, dtype=torch.float64) / C ** 0.5 for L in Ls]
    dEs = [torch.randn(B, L, generator=g, dtype=torch.float64) for L in Ls]
    X = torch.randn(B, 300, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    h = torch.where(xr > 0, xr, 0.2 * xr) if pre == 1 else xr
    mean, var = h.mean(0), h.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    yb = (h - mean) * invstd * gr + br
    out = torch.relu(yb) * (mask / 0.9) if post == 2 else yb
    sum(((out @ w.t()) * d).sum() for w, d in zip(Wr, dEs)).backward()
    f32 = lambda t: t.float().to(dev).contiguous()
    xd, od = f32(x), f32(out.detach())
    ups = [(f32(d), f32(w), torch.full(w.shape, float("nan"), device=dev), torch.full((w.shape[0],), float("nan"), device=dev))
           for w, d in zip(Ws, dEs)]
    dg, db, dbias = (torch.full((C,), float("nan"), device=dev) for _ in range(3))
    dy = torch.full((B, C), float("nan"), device=dev)
    dyT = ops.new_split(C, B, dev)
    for t in dyT:
        t.fill_(5.0)
    gx = f32(X @ X.t())
    nb = ops.block_bwd_blocks(C)
    slots = torch.full((nb,), float("nan"), dtype=torch.float64, device=dev)
    ops.block_bwd(ops.IMMEDIATE, ups, xd, od, f32(gamma), f32(mean.detach()), f32(invstd.detach()), dg, db, dbias, pre, post,
                  0.1 if post == 2 else 0.0, dy=dy, dyT=dyT, gram_x=gx, slots=slots)
    close(dy, xr.grad, 2e-4, 2e-5 * float(xr.grad.abs().max()), "dy")
    close(dg, gr.grad, 2e-4, 2e-4 * float(gr.grad.abs().max()), "dgamma")
    close(db, br.grad, 2e-4, 2e-4 * float(br.grad.abs().max()), "dbeta")
    close(dbias, xr.grad.sum(0), 1e-3, 2e-4 * float(xr.grad.abs().max()) * B ** 0.5, "dbias")
    for (d, w, gW, gb), wr in zip(ups, Wr):
        close(gW, wr.grad, 2e-4, 2e-5 * float(wr.grad.abs().max()), "gW")
        close(gb, d.double().sum(0), 2e-4, 1e-5, "gb")
    back = dyT[0].float() + dyT[1].float()                    # [C, pad32(B)] = dy^T, zero padded
    close(back[:, :B].t(), dy, 1e-4, 2.0 ** -15 * float(dy.abs().max()), "dyT")
    assert float(back[:, B:].abs().sum()) == 0.0
    ref_norm2 = float(((xr.grad.t() @ X) ** 2).sum())
    assert abs(float(slots.sum()) - ref_norm2) <= 2e-5 * ref_norm2, (float(slots.sum()), ref_norm2)
