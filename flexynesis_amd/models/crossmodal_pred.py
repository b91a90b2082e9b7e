"""CrossModalPred: encode ``input_layers``, reconstruct ``output_layers`` (an auto-encoder when they coincide), MMD
regulariser on the latent and supervisor heads on z (reference models/crossmodal_pred.py:16-540).  The network is
supervised_vae's with separate encoder / decoder layer lists, so it runs on the same engine schedule and kernels
(SURVEY.md section 8(f), rank 1)."""
import torch
from torch import nn

from ..modules import MLP, Decoder, Encoder, _LinearFn
from .base import FxModel


class CrossModalPred(FxModel):
    MODEL = "CrossModalPred"

    def __init__(self, config, dataset, target_variables=None, batch_variables=None, surv_event_var=None,
                 surv_time_var=None, input_layers=None, output_layers=None, use_loss_weighting=True, device_type=None):
        # reference crossmodal_pred.py:62-65: default = every layer of the dataset, in dataset.dat order
        self_in = list(input_layers) if input_layers else list(dataset.dat.keys())
        self_out = list(output_layers) if output_layers else list(dataset.dat.keys())
        super().__init__(config, dataset, target_variables or [], batch_variables, surv_event_var, surv_time_var,
                         use_loss_weighting, device_type, input_layers=self_in, output_layers=self_out)
        self.input_layers, self.output_layers = self_in, self_out

    def _build_modules(self):
        spec = self.spec
        L, n = spec.latent_dim, len(spec.enc_idx)
        self.encoders = nn.ModuleList([Encoder(spec.layers[i][1], [spec.hidden(i)], L) for i in spec.enc_idx])
        self.FC_mean = nn.Linear(n * L, L)
        self.FC_log_var = nn.Linear(n * L, L)
        self.decoders = nn.ModuleList([Decoder(L, [spec.hidden(i)], spec.layers[i][1]) for i in spec.dec_idx])
        self.MLPs = nn.ModuleDict({v: MLP(L, spec.supervisor_hidden_dim, C) for (v, _, C) in spec.variables})

    def multi_encoder(self, x_list):
        pairs = [enc(x) for enc, x in zip(self.encoders, x_list)]
        mean = _LinearFn.apply(torch.cat([m for m, _ in pairs], 1), self.FC_mean.weight, self.FC_mean.bias)
        log_var = _LinearFn.apply(torch.cat([v for _, v in pairs], 1), self.FC_log_var.weight, self.FC_log_var.bias)
        return mean, log_var

    def reparameterization(self, mean, var):
        # the reference uses log_var directly as the scale (crossmodal_pred.py:189-202)
        return mean + var * torch.randn_like(var)

    def forward(self, x_list_input):
        """x_list_input: the INPUT layers only, in ``input_layers`` order (crossmodal_pred.py:159-187)."""
        mean, log_var = self.multi_encoder(x_list_input)
        z = self.reparameterization(mean, log_var)
        x_hat_list = [dec(z) for dec in self.decoders]
        return x_hat_list, z, mean, log_var, {var: mlp(z) for var, mlp in self.MLPs.items()}

    def decode(self, dataset):
        """{output layer: DataFrame [features x samples]} of the reconstructions (crossmodal_pred.py:467-481)."""
        import pandas as pd
        self.eval()
        cols = {l: [] for l in self.output_layers}
        for _, dat in self._eval_batches(dataset, 64):
            plan = self._run_eval(dat)
            for j, l in enumerate(self.output_layers):
                cols[l].append(plan.xhat[j].detach().cpu().clone())
        out = {}
        for l in self.output_layers:
            x = pd.DataFrame(torch.cat(cols[l], 0).numpy()).transpose()
            x.columns = dataset.samples
            x.index = dataset.features[l]
            out[l] = x
        return out
