"""supervised_vae: per-omics Encoder -> fused mean/log_var -> z = mean + log_var*eps -> per-omics Decoder,
MMD(+reconstruction) regulariser and supervisor heads on z (reference models/supervised_vae.py:21-550)."""
import torch
from torch import nn

from ..modules import MLP, Decoder, Encoder, _LinearFn
from .base import FxModel


class supervised_vae(FxModel):
    MODEL = "supervised_vae"

    def __init__(self, config, dataset, target_variables, batch_variables=None, surv_event_var=None,
                 surv_time_var=None, use_loss_weighting=True, device_type=None):
        super().__init__(config, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                         use_loss_weighting, device_type)
        self.dataset = dataset
        self.nan_detected = False

    def _build_modules(self):
        spec = self.spec
        L, n = spec.latent_dim, spec.n_layers
        self.encoders = nn.ModuleList([Encoder(F, [spec.hidden(i)], L) for i, (_, F) in enumerate(spec.layers)])
        self.FC_mean = nn.Linear(n * L, L)
        self.FC_log_var = nn.Linear(n * L, L)
        self.decoders = nn.ModuleList([Decoder(L, [spec.hidden(i)], F) for i, (_, F) in enumerate(spec.layers)])
        self.MLPs = nn.ModuleDict({v: MLP(L, spec.supervisor_hidden_dim, C) for (v, _, C) in spec.variables})

    def multi_encoder(self, x_list):
        pairs = [enc(x) for enc, x in zip(self.encoders, x_list)]
        mean = _LinearFn.apply(torch.cat([m for m, _ in pairs], 1), self.FC_mean.weight, self.FC_mean.bias)
        log_var = _LinearFn.apply(torch.cat([v for _, v in pairs], 1), self.FC_log_var.weight, self.FC_log_var.bias)
        return mean, log_var

    def reparameterization(self, mean, var):
        # the reference uses log_var directly as the scale (supervised_vae.py:187-200)
        return mean + var * torch.randn_like(var)

    def forward(self, x_list):
        mean, log_var = self.multi_encoder(x_list)
        z = self.reparameterization(mean, log_var)
        x_hat_list = [dec(z) for dec in self.decoders]
        return x_hat_list, z, mean, log_var, {var: mlp(z) for var, mlp in self.MLPs.items()}
