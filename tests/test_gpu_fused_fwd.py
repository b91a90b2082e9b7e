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
    S = ops.dw_adam_fwd_slabs(n_out, k_in)
    assert 1 <= S <= (k_in + 127) // 128
    slabs = torch.full((S, Bn, n_out), float("nan"), device=dev)
    ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W2[:, :k_in], m2[:, :k_in], v2[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl,
                                  xnh, xnl, Bn, slabs)
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
