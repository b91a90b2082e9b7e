"""DirectPred: per-omics MLP encoders -> concatenation -> fusion Linear -> supervisor MLP heads
(reference flexynesis/models/direct_pred.py:15-415)."""
import torch
from torch import nn

from ..modules import MLP, _LinearFn
from .base import FxModel


class DirectPred(FxModel):
    MODEL = "DirectPred"

    def _build_modules(self):
        spec = self.spec
        L = spec.latent_dim
        # hidden_dim = int(F * hidden_dim_factor) (reference direct_pred.py:78-80); MLP clamps it to >= 2
        self.encoders = nn.ModuleList([MLP(F, int(F * spec.hidden_dim_factor), L) for _, F in spec.layers])
        self.fusion_block = nn.Linear(L * spec.n_layers, L) if spec.n_layers > 1 else None
        self.MLPs = nn.ModuleDict({v: MLP(L, spec.supervisor_hidden_dim, C) for (v, _, C) in spec.variables})

    def embed(self, x_list):
        cat = torch.cat([enc(x) for enc, x in zip(self.encoders, x_list)], dim=1)
        if self.fusion_block is not None:
            return _LinearFn.apply(cat, self.fusion_block.weight, self.fusion_block.bias)
        return cat

    def forward(self, x_list):
        """{var: head output} (reference direct_pred.py:107-133); differentiable, HIP kernels underneath."""
        emb = self.embed(x_list)
        return {var: mlp(emb) for var, mlp in self.MLPs.items()}
