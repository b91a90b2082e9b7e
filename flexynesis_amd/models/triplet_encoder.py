"""MultiTripletNetwork: DirectPred-style encoders applied to (anchor, positive, negative), squared-L2
triplet loss + supervisor heads on the anchor embedding (reference models/triplet_encoder.py:18-445)."""
import torch
from torch import nn

from ..modules import MLP, _LinearFn
from .base import FxModel


class MultiTripletNetwork(FxModel):
    MODEL = "MultiTripletNetwork"

    def __init__(self, config, dataset, target_variables, batch_variables=None, surv_event_var=None,
                 surv_time_var=None, use_loss_weighting=True, device_type=None):
        super().__init__(config, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                         use_loss_weighting, device_type)
        self.main_var = self.target_variables[0]      # dictates the triplets; must be categorical (:69-75)

    def _build_modules(self):
        spec = self.spec
        L = spec.latent_dim
        self.encoders = nn.ModuleList([MLP(F, int(F * spec.hidden_dim_factor), L) for _, F in spec.layers])
        self.fusion_block = nn.Linear(L * spec.n_layers, L) if spec.n_layers > 1 else None
        self.MLPs = nn.ModuleDict({v: MLP(L, spec.supervisor_hidden_dim, C) for (v, _, C) in spec.variables})

    def concat_embeddings(self, dat):
        cat = torch.cat([enc(dat[l]) for enc, l in zip(self.encoders, dat.keys())], dim=1)
        if self.fusion_block is not None:
            return _LinearFn.apply(cat, self.fusion_block.weight, self.fusion_block.bias)
        return cat

    def forward(self, anchor, positive, negative):
        a, p, n = (self.concat_embeddings(d) for d in (anchor, positive, negative))
        return a, p, n, {var: mlp(a) for var, mlp in self.MLPs.items()}
