"""Kernel-level parity of the fused narrow-chain kernels (fx_enc_tail.hip, fx_mid.hip) against fp64 torch on the GPU box:
every saved tensor, statistic and partial product of the launches they replace.  Called through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


def _dev():
    return torch.device("cuda:0")


def _close(a, b, rtol, atol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{a.numel()} off, max err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e})"


@pytest.mark.parametrize("B,Hs,Ls,n_slabs,pre,post,use_slabs", [
    (128, [5000, 5000], [[64], [64]], 6, ACT_NONE, ACT_RELU, True),        # cfg2: two MLP encoders, six partial-sum slabs each
    (100, [1500, 644, 2052], [[48], [48], [48]], 3, ACT_NONE, ACT_RELU, True),   # ragged: B < 128, H not a multiple of 64, 3 modalities
    (128, [1024, 768], [[32, 32], [32, 32]], 5, ACT_LEAKY, ACT_NONE, True),  # VAE encoders: LeakyReLU -> BN -> FC_mean, FC_var
    (37, [260], [[128]], 1, ACT_NONE, ACT_RELU, False),                      # one modality, latent 128, x given directly, odd batch
    (2, [64], [[16]], 2, ACT_NONE, ACT_RELU, True),                          # smallest train batch
    (10, [12, 8], [[8], [8]], 1, ACT_NONE, ACT_RELU, False),                 # the fine-tune golden's widths: blocks far narrower than a tile
    (30, [12, 8], [[8], [8]], 9, ACT_NONE, ACT_RELU, True),                  # ... and more than eight slabs (two request rounds)
])
@pytest.mark.parametrize("train", [True, False])
def test_enc_tail_fwd_vs_fp64(B, Hs, Ls, n_slabs, pre, post, use_slabs, train):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(B * 7 + len(Hs))
    drop_p = 0.1 if post == ACT_RELU else 0.0
    descs, refs = [], []
    for H, Lk in zip(Hs, Ls):
        slabs = torch.randn(n_slabs, B, H, device=dev, generator=g)
        bias = torch.randn(H, device=dev, generator=g) * 0.3
        x = torch.zeros(B, H, device=dev)
        if not use_slabs:
            x.copy_(slabs.sum(0) + bias)
        out = torch.full((B, H), float("nan"), device=dev)
        gamma = torch.rand(H, device=dev, generator=g) + 0.5
        beta = torch.randn(H, device=dev, generator=g) * 0.2
        rm, rv = torch.randn(H, device=dev, generator=g) * 0.1, torch.rand(H, device=dev, generator=g) + 0.5
        rm0, rv0 = rm.clone(), rv.clone()
        sm, si = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
        mask = (torch.rand(B, H, device=dev, generator=g) < 0.9).float()
        Ws = [torch.randn(L, H, device=dev, generator=g) / H ** 0.5 for L in Lk]
        nb = ops.enc_tail_blocks(H)
        parts = [torch.full((nb, B, L), float("nan"), device=dev) for L in Lk]
        descs.append(ops.enc_tail_desc(slabs=slabs if use_slabs else None, n_slabs=n_slabs, slab_stride=B * H,
                                       lin_bias=bias if use_slabs else None, x=x, out=out, gamma=gamma, beta=beta, running_mean=rm,
                                       running_var=rv, save_mean=sm, save_invstd=si, mask=mask, ups=list(zip(Ws, parts)), seed=0, offset=0))
        refs.append((slabs, bias, x, out, gamma, beta, rm, rv, rm0, rv0, sm, si, mask, Ws, parts, nb))
    ops.enc_tail_fwd(ops.IMMEDIATE, descs, B, pre, post, train, drop_p)
    torch.cuda.synchronize()
    for i, (slabs, bias, x, out, gamma, beta, rm, rv, rm0, rv0, sm, si, mask, Ws, parts, nb) in enumerate(refs):
        xr = slabs.double().sum(0) + bias.double()
        _close(x, xr, 2e-6, 2e-6, f"modality {i}: x")
        xa = torch.where(xr > 0, xr, 0.2 * xr) if pre == ACT_LEAKY else xr
        if train:
            mean, var = xa.mean(0), xa.var(0, unbiased=False)
            _close(sm, mean, 1e-5, 1e-6, f"modality {i}: save_mean")
            _close(si, 1.0 / torch.sqrt(var + 1e-5), 1e-5, 1e-6, f"modality {i}: save_invstd")
            _close(rm, 0.9 * rm0.double() + 0.1 * mean, 1e-5, 1e-6, f"modality {i}: running_mean")
            _close(rv, 0.9 * rv0.double() + 0.1 * var * (B / (B - 1)), 1e-5, 1e-6, f"modality {i}: running_var")
        else:
            mean, var = rm0.double(), rv0.double()
            assert torch.equal(rm, rm0) and torch.equal(rv, rv0)
        y = (xa - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
        if post == ACT_RELU:
            y = y.clamp_min(0)
        if train and drop_p > 0:
            y = y * (mask.double() / (1 - drop_p))
        _close(out, y, 2e-5, 2e-5, f"modality {i}: block output")
        for k, (W, part) in enumerate(zip(Ws, parts)):
            assert not torch.isnan(part).any(), f"modality {i}: partial product {k} not fully written"
            _close(part.double().sum(0), out.double() @ W.double().t(), 1e-5, 1e-5, f"modality {i}: following Linear {k}")


def test_enc_tail_philox_stream_is_bn_act_fwds():
    """Production mode: the in-kernel Philox dropout mask of fx_enc_tail_fwd is the stream fx_bn_act_fwd draws for the same
    (seed, offset, step) -- supplying the mask recorded by fx_bn_act_fwd reproduces the Philox run bit for bit."""
    from flexynesis_amd import ops
    dev = _dev()
    B, H, L = 128, 1000, 32
    g = torch.Generator(device=dev).manual_seed(5)
    slabs = torch.randn(2, B, H, device=dev, generator=g)
    gamma, beta = torch.rand(H, device=dev, generator=g) + 0.5, torch.randn(H, device=dev, generator=g)
    W = torch.randn(L, H, device=dev, generator=g)
    ctrl = torch.zeros(64, device=dev)
    ctrl[0] = 7.0                                       # Adam step 7 folds into the counter
    seed, offset = 1234, 5 << 32
    rec = torch.zeros(B, H, device=dev)
    tmp = torch.zeros(B, H, device=dev)
    xsum = slabs.sum(0)
    ops.bn_act_fwd(ops.IMMEDIATE, tmp, xsum, gamma, beta, torch.zeros(H, device=dev), torch.ones(H, device=dev), torch.zeros(H, device=dev),
                   torch.zeros(H, device=dev), ACT_NONE, ACT_RELU, True, 0.1, mask_out=rec, seed=seed, offset=offset, ctrl=ctrl)
    outs = []
    for mask in (None, rec):
        x, out = torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)
        part = torch.zeros(ops.enc_tail_blocks(H), B, L, device=dev)
        d = ops.enc_tail_desc(slabs=slabs, n_slabs=2, slab_stride=B * H, lin_bias=None, x=x, out=out, gamma=gamma, beta=beta,
                              running_mean=torch.zeros(H, device=dev), running_var=torch.ones(H, device=dev),
                              save_mean=torch.zeros(H, device=dev), save_invstd=torch.zeros(H, device=dev), mask=mask,
                              ups=[(W, part)], seed=seed, offset=offset)
        ops.enc_tail_fwd(ops.IMMEDIATE, [d], B, ACT_NONE, ACT_RELU, True, 0.1, ctrl=ctrl)
        outs.append((out.clone(), part.clone()))
    torch.cuda.synchronize()
    keep = float(rec.mean())
    assert 0.88 < keep < 0.92
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,widths,blocks,L", [(128, [64, 64], [79, 79], 64), (100, [48, 48, 48], [24, 11, 33], 100),
                                               (37, [128], [5], 0), (128, [128, 128, 128, 128], [3, 9, 2, 1], 128), (5, [16, 20], [1, 2], 17)])
def test_fusion_fwd_vs_fp64(B, widths, blocks, L):
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(B + L)
    parts = [torch.randn(nb, B, w, device=dev, generator=g) for w, nb in zip(widths, blocks)]
    biases = [torch.randn(w, device=dev, generator=g) if i % 2 == 0 else None for i, w in enumerate(widths)]
    Kf = sum(widths)
    ecat = torch.full((B, Kf), float("nan"), device=dev)
    W = torch.randn(L, Kf, device=dev, generator=g) / Kf ** 0.5 if L else None
    b = torch.randn(L, device=dev, generator=g) if L else None
    emb = torch.full((B, L), float("nan"), device=dev) if L else None
    ops.fusion_fwd(ops.IMMEDIATE, emb, ecat, list(zip(parts, blocks)), biases, W, b)
    torch.cuda.synchronize()
    ref = torch.cat([p.double().sum(0) + (bb.double() if bb is not None else 0.0) for p, bb in zip(parts, biases)], dim=1)
    _close(ecat, ref, 2e-6, 2e-5, "ecat")
    if L:
        _close(emb, ref @ W.double().t() + b.double(), 1e-5, 1e-5, "emb")


@pytest.mark.parametrize("B,width,blocks,L", [(10, 6, [1, 1], 6), (30, 6, [1, 1], 6), (128, 61, [78, 79], 61), (7, 17, [12], 0)])
def test_fusion_fwd_unaligned_width_vs_fp64(B, width, blocks, L):
    """Latent widths that are not multiples of 4 (the engine's layout: a part's row pitch = the width rounded up to 4, pad columns
    zero, biases allocated up to the pitch) -- the fine-tune golden's latent 6 and the bench's cfg2_odd latent 61."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(B + L + width)
    pw = (width + 3) // 4 * 4
    parts = []
    for nb in blocks:
        p_ = torch.zeros(nb, B, pw, device=dev)
        p_[:, :, :width] = torch.randn(nb, B, width, device=dev, generator=g)
        parts.append(p_)
    biases = []
    for i in range(len(blocks)):
        bb = torch.zeros(pw, device=dev)
        bb[:width] = torch.randn(width, device=dev, generator=g)
        biases.append(bb if i % 2 == 0 else None)
    Kf = width * len(blocks)
    ecat = torch.full((B, Kf), float("nan"), device=dev)
    W = torch.randn(L, Kf, device=dev, generator=g) / Kf ** 0.5 if L else None
    b = torch.randn(L, device=dev, generator=g) if L else None
    emb = torch.full((B, L), float("nan"), device=dev) if L else None
    ops.fusion_fwd(ops.IMMEDIATE, emb, ecat, list(zip(parts, blocks)), biases, W, b, width=width)
    torch.cuda.synchronize()
    ref = torch.cat([p_[:, :, :width].double().sum(0) + (bb[:width].double() if bb is not None else 0.0) for p_, bb in zip(parts, biases)], dim=1)
    _close(ecat, ref, 2e-6, 2e-5, "ecat")
    if L:
        _close(emb, ref @ W.double().t() + b.double(), 1e-5, 1e-5, "emb")


@pytest.mark.parametrize("R,Fs", [(128, [20000, 20000]), (100, [1500, 644, 2052]), (37, [260]), (128, [4100, 33, 96, 5000])])
def test_grouped_assembly_kernels(R, Fs):
    """fx_gather_split_group == fx_gather_split per layer (bit for bit), fx_gram_kb_group + fx_reduce_group == X X^T (fp64)
    to the split-bf16 accuracy."""
    from flexynesis_amd import ops
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(R + len(Fs))
    N = 300
    srcs = [torch.randn(N, F, device=dev, generator=g) for F in Fs]
    idx = torch.randint(0, N, (R,), device=dev, generator=g)
    single, items = [], []
    for src in srcs:
        F = src.shape[1]
        bufs = []
        for _ in range(2):
            x = torch.full((R, F), float("nan"), device=dev)
            sp, spt = ops.new_split_kb(R, F, dev), ops.new_split(F, R, dev)
            bufs.append((x, sp[0], sp[1], spt[0], spt[1]))
        ops.gather_split(ops.IMMEDIATE, *bufs[0], src, idx, n_rows=R)
        single.append(bufs[0])
        items.append((*bufs[1], src))
    ops.gather_split_group(ops.IMMEDIATE, items, idx, None, 0, R)
    torch.cuda.synchronize()
    for a, b in zip(single, items):
        for ta, tb in zip(a, b[:5]):
            assert torch.equal(ta.view(torch.int16) if ta.dtype == torch.bfloat16 else ta, tb.view(torch.int16) if tb.dtype == torch.bfloat16 else tb)
        assert torch.equal(a[0], b[5][idx])
    slabs = [torch.full((ops.gram_kb_slices(F), R * R), float("nan"), device=dev) for F in Fs]
    ops.gram_kb_group(ops.IMMEDIATE, [(it[1], it[2]) for it in items], slabs, Fs, R)
    torch.cuda.synchronize()
    for sl, it in zip(slabs, items):
        assert not torch.isnan(sl).any()
        ref = it[0].double() @ it[0].double().t()
        got = sl.double().sum(0).view(R, R)
        _close(got, ref, 1e-4, 2e-5 * float(ref.abs().max()), "X X^T")
    if (R * R) % 4 == 0:
        outs = [torch.empty(R, R, device=dev) for _ in Fs]
        ops.reduce_group(ops.IMMEDIATE, [(o, sl, sl.shape[0], None) for o, sl in zip(outs, slabs)])
        torch.cuda.synchronize()
        for o, sl in zip(outs, slabs):
            ref = torch.zeros(R * R, device=dev)
            for z in range(sl.shape[0]):
                ref += sl[z]
            assert torch.equal(o.view(-1), ref)                     # ordered sum: bit-identical to the sequential one
