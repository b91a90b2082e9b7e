"""GPU parity tests of EXACTLY what bench.py times (run with -m gpu):

  * one / two optimisation steps at the full BASELINE shapes -- cfg2 (2 x 20000, B = 128), cfg3 (supervised_vae,
    2 x 20000) and cfg4 (MultiTripletNetwork 3 x 30000, B = 128 -> 384 stacked rows: the multi-M-tile forward and the
    XCD-partitioned dW+Adam tile order engage on their own) -- against the CPU oracle with supplied draws;
  * the dominant kernel (fused dW + clip + Adam) against an fp64 reference at 5000 x 20000 / K = 128 and
    7500 x 30000 / K = 384;
  * the production RNG mode (in-kernel Philox: what fit() / bench.py run, supplied_draws=False): keep rates, moments of
    the normal draws, independence across steps / layers / anchor-positive-negative passes, forward mask == backward gate;

(The pipelined / hipGraph-replayed schedule itself is pinned to the plain step in tests/test_gpu_api.py.)
Nothing here reads /root/reference."""
import math

import numpy as np
import pytest
import torch

from test_gpu_parity import _dev, _oracle_spec, close

pytestmark = pytest.mark.gpu

FULL = {
    # bench.py CONFIGS / SURVEY.md section 8(d)
    "cfg2": dict(model="DirectPred", layers=[("gex", 20000), ("cnv", 20000)], variables=[("y", "numerical", 1)],
                 surv=(None, None), steps=2),
    "cfg3": dict(model="supervised_vae", layers=[("gex", 20000), ("cnv", 20000)],
                 variables=[("c", "categorical", 4), ("event", "numerical", 1)], surv=("event", "time"), steps=2),
    "cfg4": dict(model="MultiTripletNetwork", layers=[("gex", 30000), ("cnv", 30000), ("meth", 30000)],
                 variables=[("c", "categorical", 4)], surv=(None, None), steps=2),
    # cfg2 under --fusion_type early (reference data.py:234-257: the layers are concatenated into one, "all"): a single
    # [10000, 40000] weight of 1.6 GB (157 row blocks x 313 column tiles, 3-4 runs per row block)
    "cfg2_early": dict(model="DirectPred", layers=[("all", 40000)], variables=[("y", "numerical", 1)], surv=(None, None), steps=1),
    # what the reference's search space actually draws (config.py:7-15: latent_dim any integer in 16..128, hidden = int(F * U[0.2, 0.5]))
    # on a cohort whose feature counts are not multiples of anything: H = 8823 / 8880, every width odd mod 4
    "cfg2_odd": dict(model="DirectPred", layers=[("gex", 19873), ("cnv", 20001)], variables=[("y", "numerical", 1)], surv=(None, None),
                     steps=2, latent=17, factor=0.444, sup=11),
}


def _arch(cfg):
    from flexynesis_amd.arch import ArchSpec
    return ArchSpec(cfg["model"], cfg["layers"], cfg.get("latent", 64), cfg.get("factor", 0.25), cfg.get("sup", 16), cfg["variables"],
                    cfg["surv"][0], cfg["surv"][1], True)


def _assert_no_bad_tile(bad, what):
    """The 0.1 % of elements a wide weight may miss are SCATTERED (entries at the split-bf16 noise floor whose Adam step flips
    sign).  A localised defect -- one 64 x 128 tile of the dW + Adam kernels, or one wave's 32 x 32 block of it -- would put
    thousands of misses into one tile and still pass a global 0.1 % count (12 whole tiles at cfg2): bound the misses per tile
    (<= 3 %) and per 32 x 32 block (<= 12.5 %) as well."""
    H, F = bad.shape
    Hp, Fp = -(-H // 64) * 64, -(-F // 128) * 128
    pad = torch.zeros(Hp, Fp, dtype=torch.float32, device=bad.device)
    pad[:H, :F] = bad.float()
    per_tile = pad.view(Hp // 64, 64, Fp // 128, 128).sum(dim=(1, 3))
    per_blk = pad.view(Hp // 32, 32, Fp // 32, 32).sum(dim=(1, 3))
    assert float(per_tile.max()) <= 256, f"{what}: {int(per_tile.max())} of 8192 elements of one 64 x 128 tile differ (tile {divmod(int(per_tile.argmax()), Fp // 128)})"
    assert float(per_blk.max()) <= 128, f"{what}: {int(per_blk.max())} of 1024 elements of one 32 x 32 block differ"


def _state_close(got, ref, g, gnorm, lr, what):
    """A parameter after one Adam step vs the oracle's.  Adam's normalised update m / (sqrt(v) + eps) is O(1) whatever the
    gradient's size, so a RELATIVE error delta of a gradient element moves the parameter by ~lr * delta.  The split-bf16
    contractions (and any fp32 reduction order) leave an absolute error of ~3e-5 of the tensor's scale per element, i.e.
    delta ~ 3e-5 * max|g| / |g|: tight for the elements that carry the gradient, up to a full +-lr step of either sign
    (2.1 lr apart) for entries at the noise floor -- in particular the biases in front of a BatchNorm, whose true gradient
    is exactly zero.  Hard bound for every element: two opposite full steps.  Beyond the
    element-wise rule a tensor may hold max(1, 0.2 %) outliers: a column sum that cancels heavily (BatchNorm / bias
    gradients) or a ReLU gate within rounding of zero puts single elements at the noise floor without a small |g|."""
    got, ref = torch.as_tensor(got).double().cpu().reshape(-1), torch.as_tensor(ref).double().reshape(-1)
    err = (got - ref).abs()
    assert float(err.max()) <= 2.1 * lr + 1e-6 + 2e-4 * float(ref.abs().max()), f"{what}: max err {float(err.max()):.3e}"
    if g is None:
        atol = torch.full_like(ref, 3e-6)
    else:
        ga = g.double().abs().reshape(-1)
        delta = 1e-4 * float(ga.max()) / (ga + 1e-30)
        delta = torch.where(ga < 2e-6 * gnorm, torch.full_like(delta, 2.1), delta)
        # floor of 5 % of a step: column-sum gradients (BatchNorm / bias) cancel heavily, so their relative error is set by
        # the size of the summed terms, not by |g|
        atol = 3e-6 + lr * torch.clamp(delta, min=0.05, max=2.1)
    bad = err > atol + 2e-4 * ref.abs()
    allowed = max(1, int(2e-3 * ref.numel()))
    assert int(bad.sum()) <= allowed, (f"{what}: {int(bad.sum())}/{ref.numel()} off (allowed {allowed}), max err "
                                       f"{float(err.max()):.3e}")


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg2_early", "cfg2_odd"])
def test_fullsize_step_vs_oracle(name):
    """The shapes bench.py times, with supplied draws: named losses <= 1e-4 relative (the north-star gate), grad norm vs
    the fp64 norm of the oracle's gradients, >= 99.9 % of every wide weight's elements tight after the step, update
    norm.  Every step starts from the oracle's state (per-step parity)."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    cfg, dev, B, N = FULL[name], _dev(), 128, 512
    layers, model = cfg["layers"], cfg["model"]
    aspec = _arch(cfg)
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, N, seed=1234)
    st = O.init_state(ospec, seed=5)
    store = ParamStore(aspec, dev, materialize_big_grads=False)
    assert len(store.big_keys) >= len(layers), store.big_keys          # the wide weights take the fused dW+Adam path
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    gen = torch.Generator().manual_seed(2024)
    opt, lr = {}, 1e-3
    for step in range(cfg["steps"]):
        if step > 0:
            store.load_state(st)
            store.reset_optimizer()
            store.load_optimizer(opt["t"], opt["m"], opt["v"])
        idx = torch.randperm(N, generator=gen)
        y = {k: ann[k][idx[:B]] for k in plan.y}
        draws = {}
        for dn, t in plan.draws.items():
            if dn == "eps" or dn.startswith("prior."):
                draws[dn] = torch.randn(t.shape, generator=gen)
            else:
                draws[dn] = (torch.rand(t.shape, generator=gen) < 0.9).float()
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
            close(got[k], v, 1e-4, 1e-6, f"{name} step{step} loss {k}")
        exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
        close(store.ctrl[5], exact, 1e-4, 1e-7, f"{name} grad norm vs the fp64 norm of the oracle's gradients")
        sd = store.state_dict()
        for k in store.big_keys:
            a, b_ = sd[k].double(), st[k].double()
            bad = (a - b_).abs() > 2e-5 + 1e-3 * b_.abs()
            assert float(bad.double().mean()) <= 1e-3, f"{name} {k} step{step}: {int(bad.sum())} of {bad.numel()} elements differ"
            _assert_no_bad_tile(bad, f"{name} {k} step{step}")
            upd_ref = b_ - st_prev[k].double()
            assert float((a - b_).norm() / upd_ref.norm()) <= 2e-2, f"{name} {k} step{step}: update norm mismatch"
            del a, b_, bad, upd_ref
        for k in store.small_keys:      # every small parameter too
            _state_close(sd[k], st[k], info["grads"].get(k), exact, lr, f"{name} step{step} state {k}")


@pytest.mark.parametrize("name", ["cfg2", "cfg2_odd"])
def test_timed_schedule_vs_oracle_cfg2(name):
    """EXACTLY what bench.py times, against the oracle: PipelinedStep with the next step's wide forward fused into the dW + Adam
    launches, batch assembly one step ahead from the resident cohort, one eager step, then hipGraph replay -- five
    consecutive optimisation steps at 2 x 20000 / B = 128 with supplied draws.  Every step is compared with the oracle's
    step from the ENGINE's own previous state (weights, BatchNorm buffers, Adam moments and step count read back), so the
    pipeline is never reloaded or refreshed: step t's wide forward really is the partial sums that step t-1's dW + Adam
    launches left behind, and t > 1 exercises non-zero moments and bias corrections."""
    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import DeviceCohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    from oracle import restate as O
    cfg, dev, B, N, nb = FULL[name], _dev(), 128, 1024, 8
    layers = cfg["layers"]
    aspec = _arch(cfg)
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, N, seed=77)
    cohort = DeviceCohort(dat, ann, dev)
    store = ParamStore(aspec, dev, materialize_big_grads=False)
    store.load_state(O.init_state(ospec, seed=9))
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=nb, seed=3, supplied_draws=True)
    assert sorted(pipe.plans[0]._next_fwd) == sorted(store.big_keys) and len(store.big_keys) == 2     # the fused schedule ...
    # ... in its 14 launches (gather + split, Gram, reduce, labels | tails, fusion, heads step | fusion backward, tails backward |
    # clip + Adam of the small parameters, 2 x dW + Adam + next forward | step begin), whatever the widths are mod 4
    assert pipe.n_launches() == 14 and pipe.plans[0].path == {"grouped_tails": True, "heads_step": True, "grouped_bwd": True}, \
        (pipe.n_launches(), pipe.plans[0].path)
    gen = torch.Generator().manual_seed(99)
    table = torch.randperm(N, generator=gen)[: nb * B]
    pipe.idx.copy_(table.to(dev))
    pipe.prime()
    lr = 1e-3

    def engine_state():
        sd = store.state_dict()
        t = int(store.ctrl[ops.CTRL_STEP]) + (int(store.ctrl[ops.CTRL_STEP_HI]) << 24)
        opt = {"t": t, "m": {k: store.m(k).detach().cpu().clone() for k in store.param_keys},
               "v": {k: store.v(k).detach().cpu().clone() for k in store.param_keys}} if t > 0 else {}
        return sd, opt

    for step in range(5):
        st_prev, opt_prev = engine_state()
        rows = table[step * B:(step + 1) * B]
        xs = [dat[n][rows] for n, _ in layers]
        y = {k: ann[k][rows] for k in pipe.pending.y}
        draws = {dn: (torch.rand(t.shape, generator=gen) < 0.9).float() for dn, t in pipe.pending.draws.items()}
        pipe.pending.set_draws({k: v.to(dev) for k, v in draws.items()})
        if pipe.graphs[0] is not None:
            pipe.replay()
        else:
            pipe.step(lr)                    # bench.py: one eager step, then capture, then replay
            pipe.capture(lr)
        got = pipe.losses()
        gnorm = float(store.ctrl[ops.CTRL_GNORM])
        st_ref, _, info = O.train_step(ospec, st_prev, opt_prev, {"x": xs, "y": y}, draws, lr)
        for k, v in info["losses"].items():
            close(got[k], v, 1e-4, 1e-6, f"step{step} loss {k}")
        exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
        close(gnorm, exact, 1e-4, 1e-7, f"step{step} grad norm vs the fp64 norm of the oracle's gradients")
        sd = store.state_dict()
        for k in store.big_keys:
            a, b_ = sd[k].double(), st_ref[k].double()
            bad = (a - b_).abs() > 2e-5 + 1e-3 * b_.abs()
            assert float(bad.double().mean()) <= 1e-3, f"{k} step{step}: {int(bad.sum())} of {bad.numel()} elements differ"
            _assert_no_bad_tile(bad, f"{k} step{step}")
            upd_ref = b_ - st_prev[k].double()
            assert float((a - b_).norm() / upd_ref.norm()) <= 2e-2, f"{k} step{step}: update norm mismatch"
            del a, b_, bad, upd_ref
        for k in store.small_keys:
            _state_close(sd[k], st_ref[k], info["grads"].get(k), exact, lr, f"step{step} state {k}")
    assert pipe.graphs[0] is not None and pipe.done == 5


@pytest.mark.parametrize("n_out,k_in,K", [(5000, 20000, 128), (7500, 30000, 384),
                                          (40000, 20000, 128)])      # hidden_dim_factor 2 (examples/configs/hpo_configuration.yaml:7): 3.2 GB per array
def test_dominant_kernel_fullsize_vs_fp64(n_out, k_in, K):
    """fx_linear_dw_adam_bf16x3 at the cfg2 and cfg4 weight shapes (cfg4: K = 3B = 384 and an 11.5 MB dY^T operand, so
    the XCD-partitioned tile order engages by itself) against dW, m, v, W computed in fp64."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(n_out + K)
    dy = torch.randn(K, n_out, generator=g, device=dev) * 1e-3
    x = torch.randn(K, k_in, generator=g, device=dev)
    ldw = ops.pad32(k_in)
    W = torch.randn(n_out, ldw, generator=g, device=dev) * 0.01
    m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-4
    v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-7
    W0, m0, v0 = W[:, :k_in].double(), m[:, :k_in].double(), v[:, :k_in].double()
    pad0 = W[:, k_in:].clone()
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 4.0
    lr, coef = 1e-3, 0.37
    ops.step_begin(ops.IMMEDIATE, ctrl, lr)           # t = 5
    ctrl[4] = coef
    dyt, xt = ops.new_split(n_out, K, dev), ops.new_split(k_in, K, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl)
    torch.cuda.synchronize()
    gr = (dy.double().t() @ x.double()) * coef
    t = 5
    m_ref = 0.9 * m0 + 0.1 * gr
    v_ref = 0.999 * v0 + 0.001 * gr * gr
    denom = v_ref.sqrt() / math.sqrt(1 - 0.999 ** t) + 1e-8
    W_ref = W0 - (lr / (1 - 0.9 ** t)) * m_ref / denom
    gscale = float(gr.abs().max())
    em = (m[:, :k_in].double() - m_ref).abs().max().item()
    assert em <= 0.1 * 3e-5 * gscale + 1e-9, (em, gscale)          # split-bf16 products: ~2^-16 relative per term
    eg = 3e-5 * gscale                                             # bound on the error of one dW element (as for m above)
    ev = ((v[:, :k_in].double() - v_ref).abs() - 1e-5 * v_ref - 0.001 * (2 * gr.abs() * eg + eg * eg)).max().item()
    assert ev <= 0, ev
    bad = (W[:, :k_in].double() - W_ref).abs() > 2e-6 + 1e-4 * (W_ref - W0).abs()
    assert float(bad.double().mean()) <= 1e-4, int(bad.sum())
    rel = ((W[:, :k_in].double() - W_ref).norm() / (W_ref - W0).norm()).item()
    assert rel <= 1e-4, rel
    assert torch.equal(W[:, k_in:], pad0)                          # row padding untouched


# ---------------------------------------------------------------------------------------------------------------
# production RNG mode (Philox in the kernels): what fit() and bench.py actually run
# ---------------------------------------------------------------------------------------------------------------
def _bn_out(plan, prefix, rows, affine):
    """Recompute the BatchNorm output (before ReLU / dropout) of an MLP block from the saved tensors of the plan;
    ``affine`` = the block's (gamma, beta) as they were BEFORE the step (Adam has moved them since)."""
    H = affine[prefix][0].shape[0]          # the reference's hidden width (the engine's buffers may be wider: inert zero columns)
    y1 = plan.buf[prefix + "/y1"][rows, :H].double()
    p = rows.start // plan.B if prefix.startswith("encoders.") else 0
    sm, si = plan.buf[prefix + "/save_mean"][p, :H].double(), plan.buf[prefix + "/save_invstd"][p, :H].double()
    return (y1 - sm) * si * affine[prefix][0] + affine[prefix][1]


def _affine_snapshot(store):
    return {k[:-len(".batchnorm.weight")]: (store.p(k).double().clone(), store.p(k[:-6] + "bias").double().clone())
            for k in store.param_keys if k.endswith(".batchnorm.weight")}


def _mask_of(plan, prefix, rows, affine):
    """(kept, defined): the dropout decision can be read off wherever the ReLU output is safely positive."""
    bn = _bn_out(plan, prefix, rows, affine)
    a1 = plan.buf[prefix + "/a1"][rows, :bn.shape[1]]
    assert not bool(plan.buf[prefix + "/a1"][rows, bn.shape[1]:].any()), f"{prefix}: the engine's pad columns must stay zero"
    defined = bn > 1e-3
    kept = a1 != 0
    # kept elements carry exactly bn / 0.9
    ok = (a1.double() - bn / 0.9).abs() <= 1e-4 * bn.abs() + 1e-5
    assert bool(ok[defined & kept].all()), f"{prefix}: kept activations are not scaled by 1 / (1 - p)"
    return kept, defined


def _rate_ok(k, n, p, what):
    sigma = math.sqrt(p * (1 - p) / n)
    assert abs(k / n - p) <= 4 * sigma + 1e-9, f"{what}: rate {k / n:.5f}, expected {p} +- {4 * sigma:.5f} (n = {n})"


@pytest.mark.parametrize("model", ["DirectPred", "MultiTripletNetwork"])
def test_philox_dropout_statistics_and_independence(model):
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    dev = _dev()
    trip = model == "MultiTripletNetwork"
    layers = [("gex", 3000), ("cnv", 2000)]
    variables = [("c", "categorical", 4), ("y", "numerical", 1)]
    spec = ArchSpec(model, layers, 64, 0.25, 16, variables, None, None, True)
    B = 96
    torch.manual_seed(3)
    store = ParamStore(spec, dev)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=False, seed=1234)
    assert not plan.draws, "production mode must not expose supplied-draw slots"
    g = torch.Generator().manual_seed(8)
    R = plan.R
    xs = [torch.randn(R, F, generator=g).to(dev) for _, F in layers]
    y = {"c": torch.randint(0, 4, (B,), generator=g).float().to(dev), "y": torch.randn(B, generator=g).to(dev)}
    if trip:
        plan.set_batch(parts=[[x[j * B:(j + 1) * B] for x in xs] for j in range(3)], y=y)
    else:
        plan.set_batch(x_list=xs, y=y)
    masks = {}
    for step in range(2):
        affine = _affine_snapshot(store)
        plan.train_step(1e-3)
        torch.cuda.synchronize()
        for i in range(len(layers)):
            for p in range(plan.passes):
                kept, defined = _mask_of(plan, f"encoders.{i}", slice(p * B, (p + 1) * B), affine)
                n, k = int(defined.sum()), int((kept & defined).sum())
                _rate_ok(k, n, 0.9, f"encoders.{i} pass {p} step {step}")
                masks[(step, f"enc{i}", p)] = (kept.clone(), defined.clone())
        for (v, _, _) in variables:
            kept, defined = _mask_of(plan, f"MLPs.{v}", slice(0, B), affine)
            masks[(step, f"head.{v}", 0)] = (kept.clone(), defined.clone())
        kh = sum(int((masks[(step, f"head.{v}", 0)][0] & masks[(step, f"head.{v}", 0)][1]).sum()) for (v, _, _) in variables)
        nh = sum(int(masks[(step, f"head.{v}", 0)][1].sum()) for (v, _, _) in variables)
        _rate_ok(kh, nh, 0.9, f"heads step {step}")

    def agreement(a, b):
        (ka, da), (kb, db) = masks[a], masks[b]
        both = da & db
        return float((ka == kb)[both].double().mean()), int(both.sum())

    # independent Bernoulli(0.9) masks agree with probability 0.9^2 + 0.1^2 = 0.82; identical streams would give 1.0
    pairs = [((0, "enc0", 0), (1, "enc0", 0)), ((0, "enc1", 0), (1, "enc1", 0))]         # step t vs t + 1
    if trip:
        pairs += [((0, "enc0", 0), (0, "enc0", 1)), ((0, "enc0", 1), (0, "enc0", 2)), ((0, "enc1", 0), (0, "enc1", 2))]
    for a, b in pairs:
        agr, n = agreement(a, b)
        assert abs(agr - 0.82) <= 4 * math.sqrt(0.82 * 0.18 / n) + 5e-3, (a, b, agr)
    # two layers of equal shape must not share a stream either: compare the common [B, min(H)] corner
    (k0, d0), (k1, d1) = masks[(0, "enc0", 0)], masks[(0, "enc1", 0)]
    h = min(k0.shape[1], k1.shape[1])
    both = d0[:, :h] & d1[:, :h]
    agr = float((k0[:, :h] == k1[:, :h])[both].double().mean())
    assert abs(agr - 0.82) <= 0.02, agr
    # a second plan with another seed draws another stream; the same seed reproduces it bit for bit
    res = []
    for seed in (1234, 1234, 99):
        torch.manual_seed(3)
        st2 = ParamStore(spec, dev)
        p2 = StepPlan(st2, B, train=True, fused=True, supplied_draws=False, seed=seed)
        if trip:
            p2.set_batch(parts=[[x[j * B:(j + 1) * B] for x in xs] for j in range(3)], y=y)
        else:
            p2.set_batch(x_list=xs, y=y)
        p2.train_step(1e-3)
        torch.cuda.synchronize()
        res.append(p2.buf["encoders.0/a1"].clone())
    assert torch.equal(res[0], res[1])
    assert not torch.equal(res[0] == 0, res[2] == 0)


@pytest.mark.parametrize("B,C", [(128, 5000), (100, 333)])
def test_philox_forward_mask_is_backward_gate(B, C):
    """MLP block in Philox mode: the mask the forward drew (recorded through mask_out) is exactly the gate the backward
    applies (it gates from the saved output, no mask tensor): dx / dgamma / dbeta vs fp64 autograd with that mask."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, C, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
    out, mask = torch.empty(B, C, device=dev), torch.full((B, C), -1.0, device=dev)
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 7.0
    ops.bn_act_fwd(ops.IMMEDIATE, out, x, gamma, beta, rm, rv, sm, si, ops.ACT_NONE, ops.ACT_RELU, True, drop_p=0.1,
                   mask_out=mask, seed=42, offset=5 << 32, ctrl=ctrl)
    torch.cuda.synchronize()
    assert bool(((mask == 0) | (mask == 1)).all())
    _rate_ok(int(mask.sum()), mask.numel(), 0.9, "recorded Philox mask")
    dout = torch.randn(B, C, generator=g).to(dev)
    dx = torch.empty(B, C, device=dev)
    dgm, dbt, dbias = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_act_bwd(ops.IMMEDIATE, dx, dgm, dbt, dbias, dout, x, out, gamma, sm, si, ops.ACT_NONE, ops.ACT_RELU, 0.1)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    bn = torch.nn.functional.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5)
    ref = torch.relu(bn) * (mask.double() / 0.9)
    close(out, ref, 1e-5, 1e-5, "forward with the recorded mask")
    ref.backward(dout.double())
    close(dx, xd.grad, 1e-3, 2e-5 * float(xd.grad.abs().max()), "dx")
    close(dgm, gd.grad, 1e-3, 1e-4 * float(gd.grad.abs().max()), "dgamma")
    close(dbt, bd.grad, 1e-3, 1e-4 * float(bd.grad.abs().max()), "dbeta")
    # a different step count / offset gives a different mask, the same one reproduces it
    m2, m3 = torch.empty_like(mask), torch.empty_like(mask)
    ops.bn_act_fwd(ops.IMMEDIATE, out, x, gamma, beta, rm, rv, sm, si, ops.ACT_NONE, ops.ACT_RELU, True, drop_p=0.1,
                   mask_out=m2, seed=42, offset=5 << 32, ctrl=ctrl)
    ctrl[0] = 8.0
    ops.bn_act_fwd(ops.IMMEDIATE, out, x, gamma, beta, rm, rv, sm, si, ops.ACT_NONE, ops.ACT_RELU, True, drop_p=0.1,
                   mask_out=m3, seed=42, offset=5 << 32, ctrl=ctrl)
    torch.cuda.synchronize()
    assert torch.equal(m2, mask) and not torch.equal(m3, mask)
    assert abs(float((m3 == mask).float().mean()) - 0.82) < 0.01


def _moments_ok(t, what):
    x = t.double().reshape(-1)
    n = x.numel()
    m1, m2, m4 = float(x.mean()), float((x ** 2).mean()), float((x ** 4).mean())
    assert abs(m1) <= 4 / math.sqrt(n), (what, "mean", m1)
    assert abs(m2 - 1) <= 4 * math.sqrt(2 / n), (what, "variance", m2)
    assert abs(m4 - 3) <= 4 * math.sqrt(96 / n), (what, "4th moment", m4)
    assert float(x.abs().max()) < 7.0, (what, "range")


def test_philox_normal_draws_moments_and_independence():
    """eps of the reparameterisation and the 200 x L MMD priors (supervised_vae.py:198,545) in production mode."""
    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    dev = _dev()
    big = torch.empty(1 << 20, device=dev)
    ops.fill_normal(ops.IMMEDIATE, big, 7, 3 << 32)
    torch.cuda.synchronize()
    _moments_ok(big, "fx_fill_normal")
    c = float(torch.corrcoef(torch.stack([big[:-1], big[1:]]))[0, 1])
    assert abs(c) < 5e-3, c
    layers = [("gex", 1500), ("cnv", 1100)]
    spec = ArchSpec("supervised_vae", layers, 64, 0.25, 16, [("c", "categorical", 4)], None, None, True)
    B = 128
    torch.manual_seed(1)
    store = ParamStore(spec, dev)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=False, seed=5)
    assert not plan.draws
    g = torch.Generator().manual_seed(0)
    plan.set_batch(x_list=[torch.randn(B, F, generator=g).to(dev) for _, F in layers],
                   y={"c": torch.randint(0, 4, (B,), generator=g).float().to(dev)})
    eps, pri = [], []
    for step in range(12):
        plan.train_step(1e-3)
        torch.cuda.synchronize()
        eps.append(plan.buf["eps_used"].clone())
        pri.append(torch.stack([plan.buf["prior.0"].clone(), plan.buf["prior.1"].clone()]))
        z = plan.buf["mean"] + plan.buf["log_var"] * plan.buf["eps_used"]     # the reference's z = mean + log_var * eps
        close(plan.buf["z"], z, 1e-6, 1e-6, "reparameterisation uses the recorded eps")
    _moments_ok(torch.stack(eps), "eps")
    _moments_ok(torch.stack(pri), "MMD prior")
    for a, b in ((eps[0], eps[1]), (pri[0][0], pri[0][1]), (pri[0][0], pri[1][0]), (eps[0][:, :64], pri[0][0][:B])):
        a, b = a.reshape(-1), b.reshape(-1)
        n = min(a.numel(), b.numel())
        assert not torch.equal(a[:n], b[:n])
        c = float(torch.corrcoef(torch.stack([a[:n], b[:n]]))[0, 1])
        assert abs(c) <= 4 / math.sqrt(n) + 1e-3, c


# ---------------------------------------------------------------------------------------------------------------
# unsupervised VAE family (no supervisor heads): the latent gradient must be rebuilt from zero every step
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["supervised_vae", "CrossModalPred"])
def test_unsupervised_vae_steps_vs_oracle(model):
    """The reference CLI accepts supervised_vae / CrossModalPred without target variables (__main__.py:997).  With no
    head nothing overwrites dz at the start of the backward, so a stale dz would leak step t's gradient into step
    t + 1: three consecutive steps on the same plan, each compared with the oracle from identical state."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, StepPlan
    from oracle import restate as O
    dev = _dev()
    layers = [("gex", 1800), ("cnv", 1300), ("meth", 900)] if model == "CrossModalPred" else [("gex", 1800), ("cnv", 1300)]
    io = (["gex", "cnv"], ["meth", "gex"]) if model == "CrossModalPred" else (None, None)
    aspec = ArchSpec(model, layers, 32, 0.6, 16, [], None, None, True, io[0], io[1])
    ospec = _oracle_spec(aspec)
    dat, _ = O.synthetic_cohort(layers, 256, seed=4)
    st = O.init_state(ospec, seed=2)
    B = 64
    store = ParamStore(aspec, dev)
    store.load_state(st)
    plan = StepPlan(store, B, train=True, fused=True, supplied_draws=True)
    assert plan.spec.loss_names() == ["mmd_loss"] and not plan.spec.weighted
    gen = torch.Generator().manual_seed(6)
    opt, lr = {}, 1e-3
    for step in range(3):
        if step > 0:
            store.load_state(st)
            store.reset_optimizer()
            store.load_optimizer(opt["t"], opt["m"], opt["v"])
        idx = torch.randperm(256, generator=gen)[:B]
        xs = [dat[n][idx] for n, _ in layers]
        draws = {n: torch.randn(t.shape, generator=gen) for n, t in plan.draws.items()}
        assert set(draws) >= {"eps", "prior.0"}
        plan.set_batch(x_list=[x.to(dev) for x in xs], y={})
        plan.set_draws({k: v.to(dev) for k, v in draws.items()})
        plan.train_step(lr)
        st, opt, info = O.train_step(ospec, st, opt, {"x": xs, "y": {}}, draws, lr)
        got = plan.losses()
        for k, v in info["losses"].items():
            close(got[k], v, 2e-5, 1e-6, f"{model} step{step} loss {k}")
        exact = sum(float((gv.double() ** 2).sum()) for gv in info["grads"].values()) ** 0.5
        close(store.ctrl[5], exact, 1e-4, 1e-7, f"{model} step{step} grad norm")       # a stale dz shows up here first
        sd = store.state_dict()
        for k in store.small_keys:
            _state_close(sd[k], st[k], info["grads"].get(k), exact, lr, f"{model} step{step} {k}")
        assert float(sd["log_vars.mmd_loss"]) == 0.0                                    # single loss term: no gradient


# ---------------------------------------------------------------------------------------------------------------
# cfg5 on hardware: the sweep's collectives on the RCCL backend (world size 1 on the single-GPU box)
# ---------------------------------------------------------------------------------------------------------------
def test_sweep_collectives_on_rccl_world1():
    """broadcast_cohort -> run_sweep (real engine trials through run_trial, LPT assignment, all_gather of the records)
    -> broadcast_state on torch.distributed's "nccl" (= RCCL) backend.  A 1-rank group still creates the communicator
    and runs every collective call; the 8-GPU shape of the same code is covered by the gloo world-2 test."""
    import os
    import socket
    import torch.distributed as dist
    from flexynesis_amd import trials
    from flexynesis_amd.sweep import run_cfg5
    dev = _dev()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                   # the communicator exists and works
        assert float(t.sum()) == 4.0
        # force_collectives: the one-rank group takes the very broadcast / all_gather / all_reduce calls an 8-GPU job makes
        # (without it world_size == 1 returns early and the RCCL lines would run for the first time on the 8-GPU node)
        calls = {"broadcast": 0, "all_gather": 0, "all_reduce": 0}
        real = {k: getattr(dist, k) for k in calls}

        def counted(name):
            def f(*a, **k):
                calls[name] += 1
                return real[name](*a, **k)
            return f
        for k in calls:
            setattr(dist, k, counted(k))
        try:
            out = run_cfg5(dev, n_trials=5, epochs=2, features=1500, samples=320, seed=3, force_collectives=True)
            assert calls["broadcast"] >= 3 + 1 and calls["all_gather"] >= 2 and calls["all_reduce"] >= 1, calls
            assert out["trials_ok"] == 5 and out["n_gpus"] == 1
            assert np.isfinite(out["best_val_loss"]) and out["winner_state_tensors"] > 10
            assert out["aggregate_samples_per_s"] > 0 and len(out["rank_busy_s"]) == 1
            # the cross-validated search: units = trial x fold, mean over the folds, final model on all samples, broadcast
            n0 = dict(calls)
            cv = run_cfg5(dev, n_trials=3, epochs=2, features=1500, samples=320, seed=4, force_collectives=True, use_cv=True, n_splits=3)
            assert cv["trials_ok"] == 3 and np.isfinite(cv["best_val_loss"]) and cv["winner_state_tensors"] > 10
            assert calls["broadcast"] > n0["broadcast"] and calls["all_gather"] > n0["all_gather"]
            # the pieces, with their results checked: cohort order, result table, winner weights
            dat = {"zeta": torch.randn(64, 300), "alpha": torch.randn(64, 200)}
            ann = {"y": torch.randn(64)}
            d2, a2 = trials.broadcast_cohort(dat, ann, dev, force_collectives=True)
            assert list(d2) == ["zeta", "alpha"] and d2["zeta"].is_cuda and torch.equal(d2["alpha"].cpu(), dat["alpha"])
            table = trials.gather_results([(0, 0.5, 3, trials.STATUS_OK), (2, float("inf"), 0, trials.STATUS_FAILED)], 3, dev,
                                          force_collectives=True)
            assert table[0, 1] == 0.5 and np.isinf(table[1, 1]) and table[1, 3] == trials.STATUS_FAILED and table[0, 4] == 0
            shapes = {"w": (3, 4), "bn.num_batches_tracked": ()}
            st = trials.broadcast_state({"w": torch.arange(12.).reshape(3, 4), "bn.num_batches_tracked": torch.tensor(7)},
                                        shapes, 0, dev, force_collectives=True)
        finally:
            for k in calls:
                setattr(dist, k, real[k])
        assert st["w"].is_cuda and st["w"].cpu().tolist() == torch.arange(12.).reshape(3, 4).tolist()
        assert int(st["bn.num_batches_tracked"]) == 7
    finally:
        dist.destroy_process_group()


def test_bench_with_forced_process_group_reports_sweep_object(tmp_path):
    """`FX_BENCH_FORCE_PG=1 python bench.py --gpus 1`: the bench creates the RCCL group on one GPU, runs the timed cfg2
    steps and then the cfg5 leg with its collectives; stdout is exactly one JSON line carrying roofline + sweep."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FX_BENCH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--features", "3000", "--no-cpu-baseline", "--sweep-trials-per-gpu", "3"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["loss_finite"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["device_copy_GBps_this_box"] > 1000
    sw = out["sweep"]
    assert "error" not in sw, sw
    assert sw["trials"] == 3 and sw["trials_ok"] == 3 and sw["winner_state_tensors"] > 10 and sw["aggregate_samples_per_s"] > 0


def test_bench_dry_mode_runs_every_collective_once_and_reports_phases():
    """`bench.py --gpus N --dry` (here N = 1 with the RCCL group forced): cohort broadcast, one 1-epoch trial per rank, all_gather of the
    records, winner broadcast; one JSON line with the seconds of every phase and each rank's placement of the headline weights."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FX_BENCH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry", "--features", "3000"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry"] and out["n_gpus"] == 1 and out["error"] is None and out["trials_ok"] == 1 and out["winner_state_tensors"] > 10
    ph = out["phases_s"]
    assert all(ph[k] is not None and ph[k] >= 0 for k in ("cohort_generate", "cohort_broadcast", "units_s", "gather_s", "winner_broadcast_s", "sweep_wall"))
    assert len(out["ranks"]) == 1 and out["ranks"][0]["rank"] == 0 and "placement" in out["ranks"][0]


def test_integration_md_ctypes_binding_runs_as_written():
    """INTEGRATION.md section 3 shows the ctypes stub a flexynesis maintainer would add to call one kernel (fx_cox_ph) from the
    reference's own PyTorch code.  The block is executed verbatim (only the library path is made absolute) and checked
    against the oracle's restatement of cox_ph_loss and its autograd gradient."""
    import os
    import re
    from oracle import restate as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = [b for b in blocks if "def cox_ph_loss_hip" in b]
    assert len(code) == 1
    src = code[0].replace('"flexynesis_amd/csrc/libfxhip.so"', repr(os.path.join(root, "flexynesis_amd", "csrc", "libfxhip.so")))
    ns = {}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B = 96
    out = torch.randn(B, 1, generator=g)
    dur = torch.rand(B, generator=g) * 10
    ev = (torch.rand(B, generator=g) < 0.6).float()
    ev[5] = float("nan")                                  # a missing label
    loss, grad = ns["cox_ph_loss_hip"](out.to(dev), dur.to(dev), ev.to(dev))
    o = out.double().clone().requires_grad_(True)
    ref = O.cox_ph(o, dur.double(), ev.double())
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())) + 1e-7
    assert torch.allclose(grad.cpu().double(), o.grad, rtol=1e-4, atol=1e-7)


# ---- the plain-bf16 THROUGHPUT mode (precision="bf16"): a labelled second mode beside the split-bf16 parity mode ---------------------
# Documented tolerance: (a) the kernels compute exactly "operands rounded to bf16, products exact, fp32
# accumulation" -- checked against fp64 contractions of the ROUNDED operands at the split-bf16 kernels' own tolerances; (b) a training
# step's named losses stay within 5e-3 relative of the fp32 oracle (measured ~1e-3: bf16 carries 8 significant bits, a contraction over
# K = 5000 .. 20000 averages the roundings); (c) a free-running 5-step trajectory stays inside the band the reference itself shows
# between torch's "medium" (what main.py:24 sets) and "highest" matmul precision: 2.7e-2 relative.
BF16_STEP_RTOL = 5e-3
BF16_TRAJ_RTOL = 2.7e-2


def test_bf16_mode_kernels_are_exact_products_of_rounded_operands():
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    n_out, k_in, K = 2048, 4096, 128
    rnd = lambda t: t.bfloat16().double()
    # -- forward: y = bf16(x) . bf16(W)^T, and its data gradient dx = bf16(dy) . bf16(W)
    x = torch.randn(K, k_in, generator=g, device=dev)
    W = torch.randn(n_out, k_in, generator=g, device=dev) * 0.02
    b = torch.randn(n_out, generator=g, device=dev)
    y = torch.empty(K, n_out, device=dev)
    sp = ops.new_split_kb(K, k_in, dev)
    ops.split_bf16(ops.IMMEDIATE, sp[0], sp[1], x)
    ws = ops.Workspace(dev)
    for products, op_x, op_w in ((1, rnd(x), rnd(W)), (3, x.double(), W.double())):
        rec = ops.TapeRecorder(products=products)
        ops.linear_fwd_bf16x3(rec, y, sp[0], sp[1], W, b, ws)
        rec.run()
        ref = op_x @ op_w.t() + b.double()
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-5, (products, err)
    plain_vs_full = float((rnd(x) @ rnd(W).t() - x.double() @ W.double().t()).abs().max() / (x.double() @ W.double().t()).abs().max())
    assert plain_vs_full > 1e-4                                    # (the two modes really differ: the check above tells them apart)
    dy = torch.randn(K, n_out, generator=g, device=dev) * 1e-3
    dsp = ops.new_split_kb(K, n_out, dev)
    ops.split_bf16(ops.IMMEDIATE, dsp[0], dsp[1], dy)
    dx = torch.empty(K, k_in, device=dev)
    rec = ops.TapeRecorder(products=1)
    ops.linear_bwd_x_bf16x3(rec, dx, dsp[0], dsp[1], W, ws)
    rec.run()
    ref = rnd(dy) @ rnd(W)
    assert float((dx.double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    # -- dW + clip + Adam (+ the next step's forward): dW = bf16(dy)^T . bf16(x); the next forward multiplies bf16(W_new)
    ldw = ops.pad32(k_in)
    m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-4
    v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-7
    Wp = torch.zeros(n_out, ldw, device=dev)
    Wp[:, :k_in] = W
    ctrl = torch.zeros(64, device=dev)
    lr, coef, t = 1e-3, 0.37, 5
    dyt, xt = ops.new_split(n_out, K, dev), ops.new_split(k_in, K, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    gr = (rnd(dy).t() @ rnd(x)) * coef
    m_ref = 0.9 * m[:, :k_in].double() + 0.1 * gr
    v_ref = 0.999 * v[:, :k_in].double() + 0.001 * gr * gr
    W_ref = W.double() - (lr / (1 - 0.9 ** t)) * m_ref / (v_ref.sqrt() / math.sqrt(1 - 0.999 ** t) + 1e-8)
    xn = torch.randn(K, k_in, generator=g, device=dev)
    nsp = ops.new_split_kb(K, k_in, dev)
    ops.split_bf16(ops.IMMEDIATE, nsp[0], nsp[1], xn)
    for fused in (False, True):
        W1, m1, v1 = Wp.clone(), m.clone(), v.clone()
        ctrl.zero_()
        ctrl[0] = 4.0
        ops.step_begin(ops.IMMEDIATE, ctrl, lr)           # t = 5
        ctrl[4] = coef
        rec = ops.TapeRecorder(products=1)
        if fused:
            S = ops.dw_adam_fwd_slabs(n_out, k_in, 128)
            slabs = torch.empty(S, K, n_out, device=dev)
            ops.linear_dw_adam_fwd_bf16x3(rec, W1[:, :k_in], m1[:, :k_in], v1[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, nsp[0], nsp[1],
                                          K, slabs)
        else:
            ops.linear_dw_adam_bf16x3(rec, W1[:, :k_in], m1[:, :k_in], v1[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl)
        rec.run()
        torch.cuda.synchronize()
        gscale = float(gr.abs().max())
        assert float((m1[:, :k_in].double() - m_ref).abs().max()) <= 0.1 * 3e-5 * gscale + 1e-9, fused
        rel = float((W1[:, :k_in].double() - W_ref).norm() / (W_ref - W.double()).norm())
        assert rel <= 1e-4, (fused, rel)
        if fused:
            yn = slabs.sum(0).double()
            ref = rnd(xn) @ rnd(W1[:, :k_in]).t()
            assert float((yn - ref).abs().max() / ref.abs().max()) <= 2e-5


@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_train_step_bf16_mode_within_documented_band(name):
    """precision="bf16" on the schedule bench.py times (PipelinedStep, hipGraph replay), free-running for 5 steps from the oracle's
    initial state on the oracle's batches and draws: every step's named losses within BF16_STEP_RTOL of the fp32 oracle's step from
    ITS OWN trajectory while both are close (step 0 exactly comparable), the 5-step trajectory within BF16_TRAJ_RTOL -- and the same
    run in the parity mode (bf16x3) within 1e-4, so the band is about the mode and not about the harness."""
    from flexynesis_amd import ops
    from flexynesis_amd.data import DeviceCohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    from oracle import restate as O
    cfg, dev, B, N, nb = FULL[name], _dev(), 128, 768, 6
    layers = cfg["layers"]
    aspec = _arch(cfg)
    ospec = _oracle_spec(aspec)
    dat, ann = O.synthetic_cohort(layers, N, seed=78)
    cohort = DeviceCohort(dat, ann, dev)
    gen = torch.Generator().manual_seed(5)
    table = torch.randperm(N, generator=gen)[: nb * B]
    lr, steps = 1e-3, 5
    runs = {}
    draws_log = None
    for precision in ("bf16", "bf16x3"):
        store = ParamStore(aspec, dev, materialize_big_grads=False)
        store.load_state(O.init_state(ospec, seed=9))
        pipe = PipelinedStep(store, B, cohort=cohort, n_batches=nb, seed=3, supplied_draws=True, precision=precision)
        assert pipe.plans[0].plain_bf16 == (precision == "bf16")
        if name == "cfg2":           # the SAME schedule in both modes: the next step's wide forward rides on the dW + Adam launches
            assert sorted(pipe.plans[0]._next_fwd) == sorted(store.big_keys) and pipe.n_launches() == 14
        pipe.idx.copy_(table.to(dev))
        pipe.prime()
        if draws_log is None:
            dgen = torch.Generator().manual_seed(17)
            shapes = {dn: t.shape for dn, t in pipe.pending.draws.items()}
            draws_log = []
            for _ in range(steps):
                d = {}
                for dn, shp in shapes.items():
                    normal = dn == "eps" or dn.startswith("prior.")
                    d[dn] = torch.randn(shp, generator=dgen) if normal else (torch.rand(shp, generator=dgen) < 0.9).float()
                draws_log.append(d)
        got = []
        for step in range(steps):
            pipe.pending.set_draws({k: v.to(dev) for k, v in draws_log[step].items()})
            if pipe.graphs[0] is not None:
                pipe.replay()
            else:
                pipe.step(lr)
                pipe.capture(lr)
            got.append(pipe.losses())
        runs[precision] = got
        y_keys = list(pipe.pending.y)
        pipe.close()
        del pipe, store
    st, opt = O.init_state(ospec, seed=9), {}
    ref = []
    for step in range(steps):
        rows = table[step * B:(step + 1) * B]
        batch = {"x": [dat[n][rows] for n, _ in layers], "y": {k: ann[k][rows] for k in y_keys}}
        st, opt, info = O.train_step(ospec, st, opt, batch, draws_log[step], lr)
        ref.append({k: float(v.reshape(-1)[0]) for k, v in info["losses"].items()})
    worst = {"bf16": 0.0, "bf16x3": 0.0}
    # (free-running, the parity mode's own deviation grows with the steps too -- chaotic amplification of 1e-5-level differences through
    # Adam's normalised updates: 2e-4 at step 3 of the full-size models; the per-step parity tests above resynchronise every step)
    table_ = {p: [max(abs(runs[p][s_][k] - r) / max(abs(r), 1e-6) for k, r in ref[s_].items()) for s_ in range(steps)] for p in runs}
    print(f"[bf16 band] {name}: worst relative loss deviation per free-running step: " + "; ".join(f"{p} " + " ".join(f"{v:.1e}" for v in t) for p, t in table_.items()))
    for precision, tol_step, tol_traj in (("bf16x3", 1e-4, 1e-3), ("bf16", BF16_STEP_RTOL, BF16_TRAJ_RTOL)):
        for step in range(steps):
            for k, r in ref[step].items():
                rel = abs(runs[precision][step][k] - r) / max(abs(r), 1e-6)
                worst[precision] = max(worst[precision], rel)
                tol = tol_step if step == 0 else tol_traj
                assert rel <= tol + 1e-6, f"{name} {precision} step {step} loss {k}: {runs[precision][step][k]} vs {r} (rel {rel:.2e} > {tol})"
    print(f"[bf16 band] {name}: worst relative loss deviation over {steps} free-running steps: bf16 {worst['bf16']:.2e}, bf16x3 {worst['bf16x3']:.2e}")
    assert worst["bf16"] > worst["bf16x3"]            # (the throughput mode IS the less exact one; it ran)
