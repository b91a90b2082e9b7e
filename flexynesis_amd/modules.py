"""Network blocks with the reference's interface (reference flexynesis/modules.py: ``MLP`` :106-150,
``Encoder`` :10-57, ``Decoder`` :60-103, ``cox_ph_loss`` :265-305) whose arithmetic runs on the HIP
kernels of libfxhip through ``torch.autograd.Function`` wrappers (forward AND backward are hand-written
kernels; autograd only routes tensors).  Parameter containers are torch's own ``nn.Linear`` /
``nn.BatchNorm1d`` so that ``state_dict`` keys, shapes, default initialisation, ``requires_grad``
freezing and (de)serialisation are identical to the reference's.  GPU tensors only -- no CPU path.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, IMMEDIATE, Workspace

__all__ = ["Encoder", "Decoder", "MLP", "cox_ph_loss"]

_WS = {}
def _next_seed() -> int:
    """Philox seed of one dropout call, drawn from torch's CPU default generator: reproducible under torch.manual_seed -- also
    when the same seed is set again inside one process -- like torch's own dropout, and free of any device work or host
    synchronisation (a CPU scalar: ~3 us; the round-1 torch.randint(...).item() on the device generator cost ~10 us and a sync)."""
    return int(torch.randint(0, 1 << 62, (), dtype=torch.int64, device="cpu"))


def _ws(device) -> Workspace:
    key = (device.type, device.index)
    if key not in _WS:
        _WS[key] = Workspace(device)
    return _WS[key]


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b  (fx_gemm_f32 NT forward; NN / TN + fx_colsum backward)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x, W = _f32c(x), _f32c(W)
        y = torch.empty(x.shape[0], W.shape[0], device=x.device, dtype=torch.float32)
        ops.linear_fwd(IMMEDIATE, y, x, W, None if b is None else _f32c(b), _ws(x.device))
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f32c(dy)
        dx = dW = db = None
        ws = _ws(x.device)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.linear_bwd_x(IMMEDIATE, dx, dy, W, ws)
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            ops.linear_bwd_w(IMMEDIATE, dW, dy, x, ws)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(W.shape[0], device=x.device, dtype=torch.float32)
            ops.colsum(IMMEDIATE, db, dy)
        return dx, dW, db


class _SigmoidFn(torch.autograd.Function):
    """y = sigmoid(x) (fx_sigmoid); backward dy * y * (1 - y) from the saved output (fx_sigmoid_bwd)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        ops.sigmoid(IMMEDIATE, y, x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.sigmoid_bwd(IMMEDIATE, dx, _f32c(dy), y)
        return dx


class _BnActFn(torch.autograd.Function):
    """[LeakyReLU ->] BatchNorm1d [-> ReLU -> Dropout] in one kernel each way (fx_bn_act_fwd/bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn: nn.BatchNorm1d, pre_act, post_act, training, drop_p, mask):
        x = _f32c(x)
        B, C = x.shape
        out = torch.empty_like(x)
        sm = torch.empty(C, device=x.device)
        si = torch.empty(C, device=x.device)
        seed = _next_seed() if (training and drop_p > 0 and mask is None) else 0
        ops.bn_act_fwd(IMMEDIATE, out, x, _f32c(gamma), _f32c(beta), bn.running_mean, bn.running_var, sm, si, pre_act,
                       post_act, training, drop_p if training else 0.0, mask=mask, seed=seed, offset=0)
        if training:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, out, gamma, sm, si)
        ctx.cfg = (pre_act, post_act, drop_p if training else 0.0, training)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gamma, sm, si = ctx.saved_tensors
        pre_act, post_act, drop_p, training = ctx.cfg
        if not training:
            raise RuntimeError("backward through an eval-mode flexynesis_amd BatchNorm block is not supported")
        dout = _f32c(dout)
        C = x.shape[1]
        dx = torch.empty_like(x)
        dg, db = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
        ops.bn_act_bwd(IMMEDIATE, dx, dg, db, None, dout, x, out if post_act == ACT_RELU else None, _f32c(gamma), sm, si,
                       pre_act, post_act, drop_p)
        return dx, dg, db, None, None, None, None, None, None


def _require_gpu(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: flexynesis_amd blocks run on the GPU only (got a {x.device} tensor); "
                           "move the module and its inputs to 'cuda'")


class MLP(nn.Module):
    """Linear -> BatchNorm1d -> ReLU -> Dropout(0.1) -> Linear; ``output_dim == 1`` gives a bias-free
    scalar head (reference modules.py:106-150)."""

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        hidden_dim = max(hidden_dim, 2)
        self.layer_1 = nn.Linear(input_dim, hidden_dim)
        self.layer_out = nn.Linear(hidden_dim, output_dim) if output_dim > 1 else nn.Linear(hidden_dim, 1, bias=False)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=0.1)
        self.batchnorm = nn.BatchNorm1d(hidden_dim)

    def forward(self, x, dropout_mask=None):
        _require_gpu(x, "MLP")
        h = _LinearFn.apply(x, self.layer_1.weight, self.layer_1.bias)
        h = _BnActFn.apply(h, self.batchnorm.weight, self.batchnorm.bias, self.batchnorm, ACT_NONE, ACT_RELU,
                           self.training, self.dropout.p, dropout_mask)
        return _LinearFn.apply(h, self.layer_out.weight, self.layer_out.bias)


def _hidden_stack(input_dim, hidden_dims):
    layers, d = [], input_dim
    for h in hidden_dims:
        lin = nn.Linear(d, h)
        nn.init.xavier_uniform_(lin.weight)
        layers += [lin, nn.LeakyReLU(0.2), nn.BatchNorm1d(h)]
        d = h
    return nn.Sequential(*layers)


def _run_hidden(seq: nn.Sequential, x, training):
    h = x
    for i in range(0, len(seq), 3):
        lin, bn = seq[i], seq[i + 2]
        h = _LinearFn.apply(h, lin.weight, lin.bias)
        h = _BnActFn.apply(h, bn.weight, bn.bias, bn, ACT_LEAKY, ACT_NONE, training, 0.0, None)
    return h


class Encoder(nn.Module):
    """(Linear -> LeakyReLU(0.2) -> BatchNorm1d)* then FC_mean / FC_var (reference modules.py:10-57)."""

    def __init__(self, input_dim, hidden_dims, latent_dim):
        super().__init__()
        self.act = nn.LeakyReLU(0.2)
        self.hidden_layers = _hidden_stack(input_dim, hidden_dims)
        self.FC_mean = nn.Linear(hidden_dims[-1], latent_dim)
        nn.init.xavier_uniform_(self.FC_mean.weight)
        self.FC_var = nn.Linear(hidden_dims[-1], latent_dim)
        nn.init.xavier_uniform_(self.FC_var.weight)

    def forward(self, x):
        _require_gpu(x, "Encoder")
        h = _run_hidden(self.hidden_layers, x, self.training)
        return (_LinearFn.apply(h, self.FC_mean.weight, self.FC_mean.bias),
                _LinearFn.apply(h, self.FC_var.weight, self.FC_var.bias))


class Decoder(nn.Module):
    """(Linear -> LeakyReLU(0.2) -> BatchNorm1d)* -> FC_output -> sigmoid (reference modules.py:60-103)."""

    def __init__(self, latent_dim, hidden_dims, output_dim):
        super().__init__()
        self.act = nn.LeakyReLU(0.2)
        self.hidden_layers = _hidden_stack(latent_dim, hidden_dims)
        self.FC_output = nn.Linear(hidden_dims[-1], output_dim)
        nn.init.xavier_uniform_(self.FC_output.weight)

    def forward(self, x):
        _require_gpu(x, "Decoder")
        h = _run_hidden(self.hidden_layers, x, self.training)
        return _SigmoidFn.apply(_LinearFn.apply(h, self.FC_output.weight, self.FC_output.bias))


class _CoxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, durations, events):
        o = _f32c(outputs).reshape(-1, 1)
        loss = torch.zeros(1, device=o.device)
        do = torch.empty_like(o)
        ops.cox_ph(IMMEDIATE, loss, do, o, _f32c(durations), _f32c(events))
        ctx.save_for_backward(do)
        ctx.shape = outputs.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (do,) = ctx.saved_tensors
        return (do * g).reshape(ctx.shape), None, None


def cox_ph_loss(outputs, durations, events):
    """Cox partial-likelihood loss (reference modules.py:265-305): NaN rows dropped, 0 when nothing is
    valid or the value is non-finite (e.g. no events)."""
    _require_gpu(outputs, "cox_ph_loss")
    return _CoxFn.apply(outputs, durations, events)
