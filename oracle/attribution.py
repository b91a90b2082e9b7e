"""CPU restatement of the reference's gradient attributions -- test infrastructure (see oracle/restate.py's header).

Reference: ``compute_feature_importance`` / ``forward_target`` (models/direct_pred.py:418-590): Captum's
IntegratedGradients or GradientShap of the head output, all-zero baselines, the whole DataLoader batch folded into ONE
Captum example ([1, batch, F] inputs), |attribution| summed over the batch, divided by the number of samples.

Captum is un-vendored and not installed here: its two rules are restated from its documentation -- PARITY UNPINNED for
  * IntegratedGradients(method="gausslegendre", n_steps): attr = (x - x0) * sum_i w_i dF(x0 + a_i (x - x0))/dx with
    (a_i, w_i) the n-point Gauss-Legendre rule on [0, 1];
  * GradientShap(n_samples, stdevs=0): NoiseTunnel("smoothgrad") over InputBaselineXGradient: one alpha ~ U(0, 1) per
    expanded example -- i.e. per draw, shared by the folded batch -- a baseline drawn from the (identical, zero)
    baselines; attr = mean_i (x - x0) * dF(x0 + a_i (x - x0))/dx.
The differentiated function is the pinned oracle's eval-mode forward (restate.py) under torch autograd."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import restate as O


def head_output(spec: O.Spec, st, x_list: List[torch.Tensor], var: str, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
    """forward(x_list)[var] in eval mode, differentiable in x_list.  MLP family: direct_pred.py:107-133.  VAE family
    (supervised_vae.py:132-200, crossmodal_pred.py:79-132): x_list = the encoder inputs; the heads see the SAMPLED latent
    z = mean + log_var * eps even in eval mode (``eps`` supplied)."""
    if spec.is_vae:
        means, lvs = [], []
        for j in range(len(spec.enc_idx)):
            m, lv = O.encoder_forward(st, f"encoders.{j}", x_list[j], False, None)
            means.append(m)
            lvs.append(lv)
        mean = O.linear(torch.cat(means, 1), st["FC_mean.weight"], st["FC_mean.bias"])
        log_var = O.linear(torch.cat(lvs, 1), st["FC_log_var.weight"], st["FC_log_var.bias"])
        emb = mean + log_var * eps
    elif spec.model == "GNN":
        # gnn_early.py:427-438 (forward_target) -> :142-158: the single pseudo-layer holds [B, nodes * node_features]
        g = spec.gnn
        xg = x_list[0].reshape(-1, int(g["nodes"]), int(g["node_features"]))
        emb = O.flexgcn_forward(spec, st, "encoders.0", xg, False, {}, None)
    else:
        emb = O.directpred_embed(spec, st, x_list, False, {}, None)
    return O.mlp_forward(st, "MLPs." + var, emb, False, None, None)


def quadrature(n_steps: int):
    xs, ws = np.polynomial.legendre.leggauss(int(n_steps))
    return (0.5 * (1.0 + xs)).tolist(), (0.5 * ws).tolist()


def feature_importance(spec: O.Spec, st, dat: Dict[str, torch.Tensor], var: str, kind: str, num_class: int, method: str,
                       n: int, batch_size: int = 512, alphas: Optional[Sequence[float]] = None, dtype=torch.float64,
                       eps=None):
    """{class: [importance vector per differentiated layer]} = mean over samples of |attribution|.  VAE family: the
    differentiated layers are the encoder inputs; ``eps(batch_index, draw, rows)`` supplies each forward's draw."""
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    names = [spec.layers[i][0] for i in spec.enc_idx] if spec.is_vae else [nm for nm, _ in spec.layers]
    N = dat[names[0]].shape[0]
    out = {c: [torch.zeros(dat[nm].shape[1], dtype=dtype) for nm in names] for c in range(num_class)}
    for s0 in range(0, N, batch_size):
        xs = [dat[nm][s0:s0 + batch_size].to(dtype) for nm in names]
        if method == "IntegratedGradients":
            al, wt = quadrature(n)
        else:
            al, wt = list(alphas), [1.0 / n] * n
        for c in range(num_class):
            acc = [torch.zeros_like(x) for x in xs]
            for di, (a, w) in enumerate(zip(al, wt)):
                pts = [(x * a).requires_grad_(True) for x in xs]
                e = torch.as_tensor(eps(s0 // batch_size, di, xs[0].shape[0])).to(dtype) if eps is not None else None
                o = head_output(spec, st, pts, var, e)
                grads = torch.autograd.grad(o[:, c if num_class > 1 else 0].sum(), pts)
                for j, g in enumerate(grads):
                    acc[j] += w * g
            for j in range(len(names)):
                out[c][j] += (acc[j] * xs[j]).abs().sum(0)
    return {c: [v / N for v in out[c]] for c in out}
