"""fx_linear_dw_adam_fwd_bf16x3: the fused dW + clip + Adam kernel that also computes the NEXT step's wide forward from
the weight tile it has just updated (GPU, -m gpu).  Kernel level: against the two kernels it replaces; engine level
(tests/test_gpu_api.py, tests/test_gpu_production.py): pipelined steps with and without the fusion."""
import pytest
import torch

from test_gpu_parity import _dev, close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_out,k_in,B,Bn", [
    (300, 1100, 100, 100),        # ragged everywhere: partial row block, partial column tile, batch padded to 128
    (130, 2100, 32, 17),          # K = 32 (one K-step), tiny next batch
    (64, 128, 32, 128),           # exactly one tile
    (1250, 5000, 64, 64),         # cfg1 shape
    (2050, 2052, 128, 37),        # row-block and column remainders of 2 / 4
    (5000, 20000, 128, 128),      # cfg2 shape (what bench.py runs)
    (700, 1500, 96, 200),         # two M-tiles of the next batch (rows padded to 256), K = 96
    (1875, 3000, 384, 384),       # triplet: 3 B stacked rows = three M-tiles, K = 384 (XCD-grouped row blocks)
    (7500, 30000, 384, 384),      # cfg4 shape
    (33000, 512, 64, 64),         # > 32768 rows: an XCD's share of row blocks exceeds its 64 slots (the XCD-contiguous mapping must stand down)
    (40100, 260, 32, 200),        # the same with ragged edges and two M-tiles of the next batch
    (40000, 20000, 128, 128),     # hidden_dim_factor 2 on a 20 k-feature layer (the upper end of the shipped HPO configuration): 3.2 GB per array
])
def test_dw_adam_fwd_matches_the_kernels_it_replaces(n_out, k_in, B, Bn):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(n_out * 7 + k_in)
    dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
    x = torch.randn(B, k_in, generator=g, device=dev)
    xn = torch.randn(Bn, k_in, generator=g, device=dev)
    bias = torch.randn(n_out, generator=g, device=dev)
    ldw = ops.pad32(k_in)
    W0 = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
    m0 = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
    v0 = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 6.0
    ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
    ctrl[4] = 0.6
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    xnh, xnl = ops.new_split_kb(Bn, k_in, dev)
    ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
    # the kernels it replaces
    W1, m1, v1 = W0.clone(), m0.clone(), v0.clone()
    ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W1[:, :k_in], m1[:, :k_in], v1[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl)
    y1 = torch.empty(Bn, n_out, device=dev)
    ops.linear_fwd_bf16x3(ops.IMMEDIATE, y1, xnh, xnl, W1[:, :k_in], bias, ops.Workspace(dev))
    # fused
    W2, m2, v2 = W0.clone(), m0.clone(), v0.clone()
    Bp = ops.pad32(B)
    S = ops.dw_adam_fwd_slabs(n_out, k_in, Bp)
    assert 1 <= S <= (k_in + 127) // 128
    slabs = torch.full((S, Bn, n_out), float("nan"), device=dev)
    ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W2[:, :k_in], m2[:, :k_in], v2[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl,
                                  xnh, xnl, Bn, slabs)
    # every workgroup mapping (1 plain: the last row blocks get one run more so that all 512 slots are taken; 2 row blocks
    # interleaved over the XCDs; 3 contiguous row-block ranges per XCD) gives the same W / m / v bits and the same forward up to
    # the order in which a row block's column tiles are summed; a slab buffer larger than needed is zero-filled
    for mapping in (1, 2, 3):
        W3, m3, v3 = W0.clone(), m0.clone(), v0.clone()
        S3 = ops.dw_adam_fwd_slabs(n_out, k_in, Bp, mapping)
        slabs3 = torch.full((S3 + 1, Bn, n_out), float("nan"), device=dev)
        ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W3[:, :k_in], m3[:, :k_in], v3[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl,
                                      xnh, xnl, Bn, slabs3, mapping=mapping)
        torch.cuda.synchronize()
        assert torch.equal(W3, W2) and torch.equal(m3, m2) and torch.equal(v3, v2), mapping
        assert not bool(torch.isnan(slabs3).any()), f"mapping {mapping}: a slab element was never written"
        assert not bool(slabs3[S3].any()), "the surplus slab must be zero"
        y3, y2s = slabs3.sum(0), slabs.sum(0)
        assert float((y3.double() - y2s.double()).norm() / y2s.double().norm()) <= 2e-6, mapping
    y2 = torch.empty(Bn, n_out, device=dev)
    ops.reduce_slabs(ops.IMMEDIATE, y2, slabs, bias, S)
    torch.cuda.synchronize()
    assert not torch.equal(W2[:, :k_in], W0[:, :k_in])
    # same contraction order and the same Adam arithmetic: bit-identical optimiser results, padding untouched
    assert torch.equal(W2, W1) and torch.equal(m2, m1) and torch.equal(v2, v1)
    assert not bool(torch.isnan(slabs).any()), "a slab element was never written"
    # the forward on the UPDATED weight: vs fp64 (as test_linear_fwd_bf16x3_vs_fp64) and vs the stand-alone kernel
    ref = xn.double() @ W2[:, :k_in].double().t() + bias.double()
    scale = xn.double().abs() @ W2[:, :k_in].double().abs().t()
    assert float(((y2.double() - ref).abs() / scale).max()) <= 2e-5
    assert float((y2.double() - ref).norm() / ref.norm()) <= 1e-5
    close(y2, y1, 1e-5, 2e-6 * float(scale.max()), "fused forward vs stand-alone forward")


def test_dw_adam_fwd_argument_checks():
    from flexynesis_amd import ops
    from flexynesis_amd._lib import FxError
    dev = _dev()
    n_out, k_in, B = 64, 130, 32            # k_in % 4 != 0
    W = torch.zeros(n_out, ops.pad32(k_in), device=dev)
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    xn = ops.new_split_kb(B, k_in, dev)
    ctrl = torch.zeros(64, device=dev)
    slabs = torch.zeros(4, B, n_out, device=dev)
    with pytest.raises(FxError):
        ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], W.clone()[:, :k_in], W.clone()[:, :k_in], dyt[0], dyt[1],
                                      xt[0], xt[1], ctrl, xn[0], xn[1], B, slabs)


@pytest.mark.parametrize("model,layers,B", [
    ("DirectPred", [("gex", 2600), ("cnv", 2200)], 128),
    ("DirectPred", [("gex", 4100), ("cnv", 3000)], 37),              # ragged batch: rows padded to 128 / 64
    ("supervised_vae", [("gex", 2600), ("cnv", 2200)], 64),          # encoders fused, decoders (activations as input) not
    ("supervised_vae", [("gex", 2601), ("cnv", 2200)], 64),          # an odd feature count: encoder input padded inside the engine, target kept at F
                                                                     # ((2601, 2203) is bit-reproducible either way and agrees to 1e-7 for two steps, then its
                                                                     # fused / unfused trajectories separate to 8e-4 by step 9: Adam)
    ("DirectPred", [("gex", 2601), ("cnv", 2203)], 100),             # ... and for the MLP family (hidden 650 / 550 -> 652 / 552)
    ("DirectPred", [("gex", 20000), ("cnv", 20000)], 128),           # cfg2: the shape bench.py times
    ("MultiTripletNetwork", [("gex", 2600), ("cnv", 2200)], 96),     # 3 B = 288 stacked rows: three M-tiles
])
def test_pipeline_fused_forward_is_the_forward_of_the_next_batch(model, layers, B, monkeypatch):
    """Engine wiring: after step t the slabs of the pending plan, summed, must equal a stand-alone wide forward of the
    pending batch with the weights as they are NOW (t_boot recomputes exactly that); then the trajectory with the fusion
    tracks the one without it, and hipGraph replay reproduces eager launches bit for bit."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    dev = _dev()
    monkeypatch.setenv("FX_FUSE_NEXT_MT", "1")          # exercise the multi-M-tile path too (off by default: slower)
    trip = model == "MultiTripletNetwork"
    variables = [("c", "categorical", 4), ("y", "numerical", 1)] if trip else [("y", "numerical", 1), ("c", "categorical", 4)]
    spec = ArchSpec(model, layers, 32, 0.25, 16, variables, None, None, True)
    cohort = synthetic_cohort(layers, 700, dev, seed=3)
    nb = 4
    g = torch.Generator().manual_seed(1)
    tables = [torch.randint(0, 700, (nb * B * (3 if trip else 1),), generator=g).to(dev) for _ in range(4)]
    torch.manual_seed(5)
    init = ParamStore(spec, dev, materialize_big_grads=False).state_dict()

    def run(fuse, graph, check):
        store = ParamStore(spec, dev, materialize_big_grads=False)
        store.load_state(init)
        pipe = PipelinedStep(store, B, cohort=cohort, n_batches=nb, seed=4, fuse_next_fwd=fuse)
        fused_keys = sorted(pipe.plans[0]._next_fwd)
        wide_inputs = sorted(k for k in store.big_keys if k.startswith("encoders."))     # the wide layers fed by the batch itself
        assert fused_keys == (wide_inputs if fuse else []) and (wide_inputs or not fuse), (fused_keys, store.big_keys)
        pipe.idx.copy_(tables[0])
        pipe.prime()
        losses, e = [], 0
        for s in range(2 * nb + 1):
            if pipe.epoch_end_next():
                e += 1
                pipe.idx.copy_(tables[e])
            if graph and pipe.graphs[0] is not None:
                pipe.replay()
            else:
                pipe.step(1e-3)
                if graph:
                    pipe.capture(1e-3)
            losses.append(pipe.last.loss_vec.clone())
            if check:
                pend = pipe.plans[pipe.k]
                got = {k: pend._next_fwd[k][0].sum(0).clone() for k in fused_keys}
                pipe.refresh()                                   # stand-alone forward of the pending batch, current weights
                for k in fused_keys:
                    ref = pend._next_fwd[k][0].sum(0)
                    assert float((got[k] - ref).norm() / ref.norm()) <= 2e-6, (s, k)
                    assert float((got[k] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (s, k)
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), store.state_dict()

    l_check, _ = run(True, False, True)
    l_f, s_f = run(True, False, False)
    l_g, s_g = run(True, True, False)
    l_u, _ = run(False, False, False)
    assert torch.equal(l_f, l_g)                                   # graph replay == eager, bit for bit
    for k in s_f:
        assert torch.equal(s_f[k], s_g[k]), k
    for a in (l_f, l_check):
        assert float(((a - l_u).abs() / (l_u.abs() + 1e-6)).max()) <= 1e-4, (a, l_u)


def test_wide_weight_beyond_4GiB():
    """A layer_1.weight larger than 4 GiB (reference config.py:7-15 allows hidden_dim_factor 0.5: 50000 features -> [25000,
    50000] = 5 GB): every wide kernel addresses W / m / v through descriptors rebased per row block, so the 32-bit buffer
    offsets never span the whole weight.  The rows past the 4 GiB mark are the ones an un-rebased kernel would drop."""
    import math
    from flexynesis_amd import ops
    dev = _dev()
    n_out, k_in, B = 22000, 50000, 64
    ldw = ops.pad32(k_in)
    assert n_out * ldw * 4 > (1 << 32)
    first_beyond = (1 << 32) // (ldw * 4) + 1
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
    x = torch.randn(B, k_in, generator=g, device=dev)
    xn = torch.randn(B, k_in, generator=g, device=dev)
    W0 = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
    ctrl = torch.zeros(64, device=dev)
    lr = 1e-3
    ops.step_begin(ops.IMMEDIATE, ctrl, lr)          # t = 1: m = 0.1 g, v = 0.001 g^2, update = lr * sign-ish
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    xnh, xnl = ops.new_split_kb(B, k_in, dev)
    ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
    res = []
    for fused in (False, True):
        W, m, v = W0.clone(), torch.zeros_like(W0), torch.zeros_like(W0)
        y = torch.empty(B, n_out, device=dev)
        if fused:
            S = ops.dw_adam_fwd_slabs(n_out, k_in)
            slabs = torch.zeros(S, B, n_out, device=dev)
            ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl,
                                          xnh, xnl, B, slabs)
            ops.reduce_slabs(ops.IMMEDIATE, y, slabs, None, S)
        else:
            ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl)
            ops.linear_fwd_bf16x3(ops.IMMEDIATE, y, xnh, xnl, W[:, :k_in], None, ops.Workspace(dev))
        torch.cuda.synchronize()
        res.append((W, m, v, y))
    (W1, m1, v1, y1), (W2, m2, v2, y2) = res
    assert torch.equal(W1, W2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    for rows in (slice(0, 70), slice(first_beyond - 3, first_beyond + 70), slice(n_out - 130, n_out)):
        gr = dy[:, rows].double().t() @ x.double()                       # clip coefficient 1
        assert float(gr.abs().max()) > 0
        mref = 0.1 * gr
        err = (m1[rows, :k_in].double() - mref).abs().max().item()
        assert err <= 0.1 * 3e-5 * float(gr.abs().max()), (rows, err)
        assert bool((m1[rows, :k_in] != 0).any(dim=1).all()), f"rows {rows} were not updated"
        Wn = W1[rows, :k_in].double()
        ref = xn.double() @ Wn.t()
        for y in (y1, y2):
            assert float((y[:, rows].double() - ref).norm() / ref.norm()) <= 1e-5, rows
    assert float((W1[:, :k_in] - W0[:, :k_in]).abs().max()) <= 1.001 * lr and torch.equal(W1[:, k_in:], W0[:, k_in:])


@pytest.mark.parametrize("R,O,K", [(128, 64, 128), (37, 5, 3), (384, 128, 384), (100, 70, 130), (1, 1, 1), (200, 16, 17)])
def test_small_linear_kernels_vs_fp64(R, O, K):
    """fx_small_linear_fwd / _bwd (fusion layer, VAE FC_mean / FC_log_var) against fp64, strided operands included."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(R * 7 + O * 3 + K)
    xbig = torch.randn(R, K + 5, generator=g).to(dev)
    x = xbig[:, 2:2 + K]                                   # a strided view (like ecat slices)
    W = torch.randn(O, K, generator=g).to(dev)
    b = torch.randn(O, generator=g).to(dev)
    ybig = torch.full((R, O + 3), float("nan"), device=dev)
    y = ybig[:, 1:1 + O]
    ops.small_linear_fwd(ops.IMMEDIATE, y, x, W, b)
    ref = x.double() @ W.double().t() + b.double()
    close(y, ref, 1e-5, 1e-5 * K ** 0.5, "small_linear_fwd")
    assert bool(torch.isnan(ybig[:, 0]).all()) and bool(torch.isnan(ybig[:, 1 + O:]).all())
    dy = torch.randn(R, O, generator=g).to(dev)
    dx = torch.full((R, K), float("nan"), device=dev)
    gW = torch.full((O, K), float("nan"), device=dev)
    gb = torch.full((O,), float("nan"), device=dev)
    ops.small_linear_bwd(ops.IMMEDIATE, dx, gW, gb, dy, x, W)
    close(dx, dy.double() @ W.double(), 1e-5, 2e-5, "dx")
    close(gW, dy.double().t() @ x.double(), 1e-5, 2e-5 * R ** 0.5, "gW")
    close(gb, dy.double().sum(0), 1e-5, 2e-5 * R ** 0.5, "gb")
    gW2 = torch.empty_like(gW)
    ops.small_linear_bwd(ops.IMMEDIATE, None, gW2, None, dy, x, W)        # frozen upstream / bias-free layer
    assert torch.equal(gW2, gW)
    dx2 = torch.ones(R, K, device=dev)
    ops.small_linear_bwd(ops.IMMEDIATE, dx2, gW2, gb, dy, x, W, dx_accumulate=True)
    close(dx2, 1.0 + dy.double() @ W.double(), 1e-5, 2e-5, "dx accumulate")


def test_placement_probe_leaves_contents_alone_and_store_search_is_transparent(monkeypatch):
    """fx_placement_probe walks W / m / v in the fused kernel's pattern and writes back what it read; ParamStore's search over
    candidate placements changes WHERE the wide weights live, never what they hold."""
    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import POOL, ParamStore, placement_tries
    dev = _dev()
    POOL.clear(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for (n_out, k_in) in ((300, 1100), (5000, 20000), (64, 128)):
        ld = ops.pad32(k_in)
        bufs = [torch.randn(n_out, ld, generator=g, device=dev) for _ in range(3)]
        ref = [b.clone() for b in bufs]
        us = ops.placement_probe_us(*[b[:, :k_in] for b in bufs])
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(bufs, ref))
        assert 0.5 < us < 5000
        if (n_out, k_in) == (5000, 20000):
            assert 24.0 * n_out * k_in / (us * 1e-6) > 3.0e12, us          # streams at HBM rates (observed 4.9-6.1 TB/s)
    with pytest.raises(ops.FxError):
        ops.placement_probe_us(torch.zeros(8, 6, device=dev), torch.zeros(8, 6, device=dev), torch.zeros(8, 6, device=dev))   # k_in % 4
    spec = ArchSpec("DirectPred", [("gex", 20000)], 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
    monkeypatch.delenv("FX_PLACEMENT_TRIES", raising=False)
    monkeypatch.setenv("FX_PARTITION_ARENA", "0")        # this test is about the per-weight search and the pool behind the arena (its fallback)
    states = []
    for tries in (1, 4):
        with placement_tries(tries):
            st = ParamStore(spec, dev, materialize_big_grads=False)
        st.reset_parameters(seed=5)
        info = st.placement.get("encoders.0.layer_1.weight")
        if tries == 1:
            assert info is None
        else:
            # (arrays are placed one by one: at least three candidates, at most 8 x the asked-for number, inside the time budget)
            assert info is not None and info["from_pool"] == 0 and 3 <= info["probed"] <= 32 and len(info["kept_TBps"]) == 3
            assert max(info["kept_TBps"]) == max(info["probed_TBps"]) and info["search_s"] < 5.0
        states.append(st.state_dict())
        assert st.p("encoders.0.layer_1.weight").data_ptr() % 16 == 0
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k
    # The process-level pool: a dropped store hands its RATED fast arrays back, the next model of the same shape takes them (no search
    # of its own, even with tries = 1: the k folds of a trial, the FineTuner's 45 fits), a model of another shape does not
    kept = [r for r in info["kept_TBps"] if r is not None and r >= ParamStore.PLACE_GOOD_TBS]
    ptrs = {st.big["encoders.0.layer_1.weight"][n].data_ptr() for n in ("_W", "_M", "_V")}
    del st
    import gc
    gc.collect()
    assert len(POOL.held(dev)) == len(kept) and all(shp == (5000, 20000) for shp, _ in POOL.held(dev))
    POOL.give(dev, (6100, 20000), [(torch.zeros(6100 * 20000, device=dev), 6.0)])      # an entry of another shape, ahead in the list
    POOL._free[dev.index].insert(0, POOL._free[dev.index].pop())
    with placement_tries(1):
        other = ParamStore(ArchSpec("DirectPred", [("gex", 20000)], 64, 0.3, 16, [("y", "numerical", 1)], None, None, True), dev,
                           materialize_big_grads=False)
        assert other.placement.get("encoders.0.layer_1.weight") is None and len(POOL.held(dev)) == len(kept) + 1   # [6000, 20000]: not served
        st2 = ParamStore(spec, dev, materialize_big_grads=False)
    if kept:
        info2 = st2.placement["encoders.0.layer_1.weight"]
        assert info2["from_pool"] == len(kept) and "probed" not in info2 and [shp for shp, _ in POOL.held(dev)] == [(6100, 20000)]
        assert {st2.big["encoders.0.layer_1.weight"][n].data_ptr() for n in ("_W", "_M", "_V")} & ptrs
    st2.reset_parameters(seed=5)
    s2 = st2.state_dict()
    for k in states[0]:
        assert torch.equal(states[0][k], s2[k]), k                  # pooled arrays arrive zeroed: the same model
    assert not bool(st2.m("encoders.0.layer_1.weight").any()) and not bool(st2.v("encoders.0.layer_1.weight").any())
    del st2, other
    gc.collect()
    POOL.clear(dev)


def test_partition_arena_serves_every_shape_at_the_fast_rate(monkeypatch):
    """engine.PartitionArena: W of a wide weight from one pool, m and v from a pool behind a partition boundary (profiles/r05_partitions.txt):
    the fused kernel's traffic pattern runs at the two-partition rate for any shape, with no search -- also for a short fit (tries = 1) --,
    the model is the same as through the allocator, and a dropped store gives its ranges back."""
    import gc
    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, PartitionArena, placement_tries
    dev = _dev()
    monkeypatch.delenv("FX_PLACEMENT_TRIES", raising=False)
    monkeypatch.delenv("FX_PARTITION_ARENA", raising=False)
    ar = PartitionArena.get(dev)
    if ar is None:
        pytest.skip(f"no arena on this device: {PartitionArena._arenas[dev.index].info}")
    assert ar.info["pool_A_ends_TBps"] < PartitionArena.FAST_TBS <= min(r for r in ar.info["candidate_TBps"] if r >= PartitionArena.FAST_TBS)
    gc.collect()
    free0 = ar.free_bytes()
    key = "encoders.0.layer_1.weight"
    spec = ArchSpec("DirectPred", [("gex", 20000)], 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
    odd = ArchSpec("DirectPred", [("gex", 19873)], 61, 0.444, 13, [("y", "numerical", 1)], None, None, True)
    stores = []
    for sp in (spec, odd):
        with placement_tries(1):                        # what fit() sets for a short fit: no per-weight search -- the arena serves it all the same
            st = ParamStore(sp, dev, materialize_big_grads=False)
        info = st.placement[key]
        out, fin = st.eshapes[key]
        assert info["arena"] is True and info["search_s"] < 5.0
        assert 24.0 * out * fin / (info["kept_us"] * 1e-6) >= 5.5e12, info           # one partition: 4.9-5.0 TB/s; two: 5.9-6.1
        big = st.big[key]
        a0, a1 = ar.chunks[0].data_ptr(), ar.chunks[0].data_ptr() + ar.chunks[0].numel()
        assert a0 <= big["_W"].data_ptr() < a1 and not (a0 <= big["_M"].data_ptr() < a1) and not (a0 <= big["_V"].data_ptr() < a1)
        assert big["_W"].data_ptr() % 16 == 0 and not bool(big["_M"].any()) and not bool(big["_V"].any())
        stores.append(st)
    assert ar.free_bytes()[0] < free0[0] and ar.free_bytes()[1] < free0[1]
    st = stores[0]
    st.reset_parameters(seed=5)
    sd = st.state_dict()
    monkeypatch.setenv("FX_PARTITION_ARENA", "0")
    with placement_tries(1):
        ref = ParamStore(spec, dev, materialize_big_grads=False)
    monkeypatch.delenv("FX_PARTITION_ARENA")
    assert ref.placement.get(key) is None
    ref.reset_parameters(seed=5)
    for k, v in ref.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert ar.take3(64 << 30) is None                    # more than a pool holds: the caller falls back to the allocator
    del st, stores, ref, big
    gc.collect()
    assert ar.free_bytes() == free0                      # every range came back and merged


def test_partition_arena_gives_up_within_its_time_budget(monkeypatch):
    """A build that cannot afford its allocations (FX_ARENA_BUDGET_S) leaves the arena off and ParamStore on the bounded search; the next
    process-level build (reset) works again."""
    import gc
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.engine import ParamStore, PartitionArena, placement_tries
    dev = _dev()
    monkeypatch.delenv("FX_PLACEMENT_TRIES", raising=False)
    monkeypatch.delenv("FX_PARTITION_ARENA", raising=False)
    gc.collect()
    PartitionArena.reset(dev)
    monkeypatch.setenv("FX_ARENA_BUDGET_S", "0.000001")
    assert PartitionArena.get(dev) is None
    info = PartitionArena._arenas[dev.index].info
    assert "skipped" in info and info["build_s"] < 5.0, info
    spec = ArchSpec("DirectPred", [("gex", 20000)], 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
    with placement_tries(1):
        st = ParamStore(spec, dev, materialize_big_grads=False)
    assert st.placement.get("encoders.0.layer_1.weight") is None          # first placement, as a short fit asks for
    st.reset_parameters(seed=1)
    assert bool(torch.isfinite(st.p("encoders.0.layer_1.weight")).all())
    del st
    gc.collect()
    monkeypatch.delenv("FX_ARENA_BUDGET_S")
    PartitionArena.reset(dev)                                              # later tests build their arena afresh


@pytest.mark.parametrize("M,K,N", [(128, 5000, 20000), (100, 4100, 1000), (37, 2080, 300)])
def test_forward_tile_variants_compute_the_same_contraction(M, K, N):
    """fx_linear_fwd_bf16x3_ex's kernel choices -- the 128 x 128 tile, the register-fragment kernel (no_mt 4) and round 6's 128 x 256
    tile (wave_cols 8: the activation tile crosses L2 -> LDS once per 32 KB of W instead of once per 16 KB) -- against an fp64
    contraction, in the parity mode and (hi operands only) in the plain-bf16 mode; equal split-K -> bit-identical sums."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(M + N)
    x = torch.randn(M, K, generator=g, device=dev)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    b = torch.randn(N, generator=g, device=dev)
    sp = ops.new_split_kb(M, K, dev)
    ops.split_bf16(ops.IMMEDIATE, sp[0], sp[1], x)
    ws = ops.Workspace(dev)
    ws.reserve(8 * M * N * 4)
    for products, ref in ((3, x.double() @ W.double().t() + b.double()),
                          (1, x.bfloat16().double() @ W.bfloat16().double().t() + b.double())):
        outs = {}
        for name, (wc, no_mt) in {"128x128": (0, 0), "128x256": (8, 0), "reg": (0, 4)}.items():
            y = torch.full((M, N), float("nan"), device=dev)
            rec = ops.TapeRecorder(products=products)
            rec.emit("fx_linear_fwd_bf16x3_ex", y.data_ptr(), sp[0].data_ptr(), ops._lo(rec, sp[1]), W.data_ptr(), b.data_ptr(), M, N, K,
                     sp[0].shape[1], W.stride(0), y.stride(0), ws.buf.data_ptr(), ws.nbytes, 2, wc, no_mt, 0)
            rec.run()
            torch.cuda.synchronize()
            err = float((y.double() - ref).abs().max() / ref.abs().max())
            assert err <= 2e-5, (products, name, err)
            outs[name] = y
        if M > 64:      # (at most 64 rows: "128x128" is served by the register-fragment kernel, whose 16 x 16 x 32 MFMAs sum in another order)
            assert torch.equal(outs["128x128"], outs["128x256"]), products  # same K slices, same summation order within a slice
