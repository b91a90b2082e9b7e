"""GNN: a flexGCN graph-convolution encoder over one shared interaction graph -> supervisor MLP heads
(reference flexynesis/models/gnn_early.py:15-633; flexGCN, modules.py:153-262).

The convolutions are torch_geometric's GraphConv / SAGEConv / GCNConv (``gnn_conv_type`` GC | SAGE | GCN, reference
__main__.py:536); their parameters keep torch_geometric's names so a ``state_dict`` moves between the two
implementations.  Training and inference run through the engine's plans (csrc/fx_gnn.hip kernels)."""
import numpy as np
import torch
from torch import nn

from ..arch import gnn_conv_keys
from ..modules import MLP
from .base import FxModel


class _Params(nn.Module):
    """Parameter holder with torch_geometric's layout for one conv layer."""

    def __init__(self, conv, cin, cout):
        super().__init__()
        def lin(bias):
            m = nn.Linear(cin, cout, bias=bias)
            return m
        if conv == "GC":
            self.lin_rel, self.lin_root = lin(True), lin(False)
        elif conv == "SAGE":
            self.lin_l, self.lin_r = lin(True), lin(False)
        else:
            self.bias = nn.Parameter(torch.zeros(cout))
            self.lin = lin(False)
        for n, p in self.named_parameters():
            if n.endswith("bias"):
                nn.init.zeros_(p)


class FlexGCN(nn.Module):
    def __init__(self, node_count, node_feature_count, node_embedding_dim, output_dim, num_convs=2, conv="GC"):
        super().__init__()
        self.convs = nn.ModuleList([_Params(conv, node_feature_count if k == 0 else node_embedding_dim, node_embedding_dim)
                                    for k in range(num_convs)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(node_embedding_dim) for _ in range(num_convs)])
        self.fc = nn.Linear(node_embedding_dim * node_count, output_dim)


class GNN(FxModel):
    MODEL = "GNN"

    def __init__(self, config, dataset, target_variables, batch_variables=None, surv_event_var=None, surv_time_var=None,
                 use_loss_weighting=True, device_type=None, gnn_conv_type=None):
        self.gnn_conv_type = gnn_conv_type
        super().__init__(config, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                         use_loss_weighting, device_type, gnn_conv_type=gnn_conv_type)
        self.edge_index = torch.as_tensor(self.spec.gnn["edge_index"])

    def _build_modules(self):
        spec, g = self.spec, self.spec.gnn
        self.encoders = nn.ModuleList([FlexGCN(g["nodes"], g["node_features"], g["embedding_dim"], spec.latent_dim,
                                               g["num_convs"], g["conv"])])
        self.MLPs = nn.ModuleDict({v: MLP(spec.latent_dim, spec.supervisor_hidden_dim, C) for (v, _, C) in spec.variables})

    # batches are (x [B, nodes, node_features], y_dict, samples) -- reference data.py:1254-1263
    @staticmethod
    def _batch_size(batch):
        return int(batch[0].shape[0])

    def _feed(self, plan, batch):
        x, y = batch[0], batch[1]
        dev = plan.dev
        plan.set_batch(x_list=[torch.as_tensor(x).to(dev, torch.float32).reshape(x.shape[0], -1)],
                       y={k: torch.as_tensor(v).to(dev) for k, v in y.items() if k in plan.y})

    def _eval_batches(self, dataset, batch_size=64):
        n = len(dataset)
        flat = dataset.dat["nodes"]
        for s in range(0, n, batch_size):
            idx = list(range(s, min(s + batch_size, n)))
            yield idx, {"nodes": flat[idx]}

    def forward(self, x, edge_index=None):
        """{var: head output} for node features ``x`` [B, nodes, node_features] (reference gnn_early.py:142-158), in the
        module's current train/eval mode; the graph is the one the model was built with."""
        x = torch.as_tensor(x)
        plan = self._plan(int(x.shape[0]), train=False) if not self.training else self._plan(int(x.shape[0]), train=True)
        plan.set_batch(x_list=[x.to(plan.dev, torch.float32).reshape(x.shape[0], -1)], y=None)
        for t in plan.y.values():
            t.fill_(float("nan"))
        plan.forward()
        return {v: plan.buf[f"MLPs.{v}/out"].detach().clone() for v in self.variables}
