"""Architecture description derived from (config, dataset) exactly the way the reference model
constructors derive it (reference models/direct_pred.py:30-105, models/supervised_vae.py:42-130,
models/triplet_encoder.py:36-123, models/crossmodal_pred.py:31-132) plus the reference ``state_dict`` layout (the on-disk ABI read by
reference inference.py:378-379)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

import os as _os
_ENGINE_PAD = _os.environ.get("FX_ENGINE_PAD", "1") != "0"      # A/B switch: 0 = the engine computes with the reference's widths (rounds 1-4)

MODEL_KINDS = ("DirectPred", "supervised_vae", "MultiTripletNetwork", "CrossModalPred", "GNN")
GNN_CONVS = ("GC", "SAGE", "GCN")        # --gnn_conv_type choices (reference __main__.py:536); GAT is not offered by the CLI


@dataclass
class ArchSpec:
    model: str
    layers: List[Tuple[str, int]]            # (layer name, n_features) in dataset.dat.keys() order
    latent_dim: int
    hidden_dim_factor: float
    supervisor_hidden_dim: int
    variables: List[Tuple[str, str, int]]    # (name, "numerical"|"categorical", n_out)
    surv_event_var: Optional[str] = None
    surv_time_var: Optional[str] = None
    use_loss_weighting: bool = True
    # CrossModalPred (crossmodal_pred.py:62-65): layers that are encoded / reconstructed; None = every layer
    input_layers: Optional[List[str]] = None
    output_layers: Optional[List[str]] = None
    # GNN (models/gnn_early.py:103-118): one flexGCN encoder over a graph shared by all samples.  ``layers`` then holds the
    # single pseudo-layer ("nodes", n_nodes * n_node_features) = the flattened [n_nodes, n_node_features] sample.
    # keys: nodes, node_features, embedding_dim, num_convs, conv (GC|SAGE|GCN), act, edge_index (int64 ndarray [2, E])
    gnn: Optional[dict] = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_graph_ops", None)        # device-side CSR cache of the engine (engine._build_gnn): never copied / pickled
        return d

    # -- derived sizes ---------------------------------------------------------------------------
    @property
    def is_vae(self) -> bool:
        return self.model in ("supervised_vae", "CrossModalPred")

    def _idx(self, names) -> List[int]:
        order = [n for n, _ in self.layers]
        if names is None or self.model != "CrossModalPred":
            return list(range(len(order)))
        return [order.index(n) for n in names]

    @property
    def enc_idx(self) -> List[int]:
        """encoders.j reads layers[enc_idx[j]] (all layers except for CrossModalPred's input_layers)"""
        return self._idx(self.input_layers)

    @property
    def dec_idx(self) -> List[int]:
        """decoders.j reconstructs layers[dec_idx[j]]"""
        return self._idx(self.output_layers)

    def hidden(self, i: int) -> int:
        # int(F*factor) (direct_pred.py:78-80) then max(.,2) (modules.py:124 / supervised_vae.py:92)
        return max(int(self.layers[i][1] * self.hidden_dim_factor), 2)

    @property
    def sup_hidden(self) -> int:
        return max(int(self.supervisor_hidden_dim), 2)

    @property
    def n_layers(self) -> int:
        return len(self.layers)

    @property
    def extra_loss(self) -> Optional[str]:
        return {"DirectPred": None, "supervised_vae": "mmd_loss", "MultiTripletNetwork": "triplet_loss",
                "CrossModalPred": "mmd_loss", "GNN": None}[self.model]

    def loss_names(self) -> List[str]:
        """Insertion order of the reference's ``losses`` dict in training_step
        (supervised_vae.py:318, triplet_encoder.py:309, direct_pred.py:241-253)."""
        names = [v[0] for v in self.variables]
        return ([self.extra_loss] if self.extra_loss else []) + names

    def logvar_names(self) -> List[str]:
        """Order of the ``log_vars`` ParameterDict (direct_pred.py:60-64, supervised_vae.py:78-82)."""
        if not self.use_loss_weighting:
            return []
        names = [v[0] for v in self.variables]
        return names + ([self.extra_loss] if self.extra_loss else [])

    @property
    def weighted(self) -> bool:
        return self.use_loss_weighting and len(self.loss_names()) > 1   # direct_pred.py:213

    def param_count(self) -> int:
        return sum(int(np.prod(s)) if s else 1 for k, s in self.state_shapes().items() if not is_buffer_key(k))

    # -- engine layout -----------------------------------------------------------------------------
    # The engine's kernels move rows of activations and weights in 16-byte units, so it runs every model with its HIDDEN widths
    # (int(F * hidden_dim_factor): any integer, direct_pred.py:78-80) and the INPUT widths of its encoders (the feature count of a
    # cohort layer: any integer, data.py:358-503) rounded up to multiples of 4 (a layer the VAE family reconstructs keeps its width
    # as a TARGET: rows of FC_output; the engine holds a second, unpadded copy of such a batch).  The extra hidden units / input
    # columns are inert: their weights, biases and BatchNorm affine parameters are zero, so they output zero, receive exactly
    # zero gradients (every product that reaches them has a zero factor) and Adam leaves them at zero; nothing of them is
    # visible in ``state_dict`` (``state_shapes`` stays the reference's ABI, ParamStore exposes logical views).
    @property
    def pads_features(self) -> bool:
        return self.model != "GNN"

    def engine_hidden(self, i: int) -> int:
        return pad4(self.hidden(i)) if (self.model != "GNN" and _ENGINE_PAD) else self.hidden(i)

    def engine_features(self, i: int) -> int:
        """Width of cohort layer i as an ENCODER input (a layer that is only reconstructed keeps its width)."""
        if not (self.pads_features and _ENGINE_PAD) or (self.is_vae and i not in self.enc_idx):
            return self.layers[i][1]
        return pad4(self.layers[i][1])

    def engine_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """The shapes the engine computes with: ``state_shapes`` with hidden / encoder-input widths rounded up to 4."""
        return self._shapes(True)

    # -- state_dict manifest ---------------------------------------------------------------------
    def state_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return self._shapes(False)

    def _shapes(self, engine: bool) -> Dict[str, Tuple[int, ...]]:
        out: Dict[str, Tuple[int, ...]] = {}
        n, L, S = self.n_layers, self.latent_dim, self.sup_hidden
        hidden = self.engine_hidden if engine else self.hidden
        feats = self.engine_features if engine else (lambda i: self.layers[i][1])

        def bn(prefix, c):
            out[prefix + ".weight"] = (c,)
            out[prefix + ".bias"] = (c,)
            out[prefix + ".running_mean"] = (c,)
            out[prefix + ".running_var"] = (c,)
            out[prefix + ".num_batches_tracked"] = ()

        def mlp(prefix, fin, h, o):
            out[prefix + ".layer_1.weight"] = (h, fin)
            out[prefix + ".layer_1.bias"] = (h,)
            out[prefix + ".layer_out.weight"] = (o, h)
            if o > 1:                                   # modules.py:126-130: scalar heads are bias-free
                out[prefix + ".layer_out.bias"] = (o,)
            bn(prefix + ".batchnorm", h)

        for name in self.logvar_names():
            out["log_vars." + name] = (1,)
        if self.model in ("DirectPred", "MultiTripletNetwork"):
            for i in range(len(self.layers)):
                mlp(f"encoders.{i}", feats(i), hidden(i), L)
            if n > 1:
                out["fusion_block.weight"] = (L, n * L)
                out["fusion_block.bias"] = (L,)
        elif self.model == "GNN":
            # flexGCN (modules.py:197-249) with torch_geometric's parameter names per conv type
            g = self.gnn
            C, cin = int(g["embedding_dim"]), int(g["node_features"])
            for k in range(int(g["num_convs"])):
                wa, ba, wr = gnn_conv_keys(f"encoders.0.convs.{k}", g["conv"])
                out[wa] = (C, cin)
                out[ba] = (C,)
                if wr:
                    out[wr] = (C, cin)
                cin = C
            for k in range(int(g["num_convs"])):
                bn(f"encoders.0.bns.{k}", C)
            out["encoders.0.fc.weight"] = (L, C * int(g["nodes"]))
            out["encoders.0.fc.bias"] = (L,)
        else:
            n = len(self.enc_idx)
            for j, i in enumerate(self.enc_idx):
                F, H, p = feats(i), hidden(i), f"encoders.{j}"
                out[p + ".hidden_layers.0.weight"] = (H, F)
                out[p + ".hidden_layers.0.bias"] = (H,)
                bn(p + ".hidden_layers.2", H)
                for fc in ("FC_mean", "FC_var"):
                    out[f"{p}.{fc}.weight"] = (L, H)
                    out[f"{p}.{fc}.bias"] = (L,)
            for fc in ("FC_mean", "FC_log_var"):
                out[fc + ".weight"] = (L, n * L)
                out[fc + ".bias"] = (L,)
            for j, i in enumerate(self.dec_idx):
                F, H, p = self.layers[i][1], hidden(i), f"decoders.{j}"       # (a reconstructed layer keeps its width: rows of FC_output)
                out[p + ".hidden_layers.0.weight"] = (H, L)
                out[p + ".hidden_layers.0.bias"] = (H,)
                bn(p + ".hidden_layers.2", H)
                out[p + ".FC_output.weight"] = (F, H)
                out[p + ".FC_output.bias"] = (F,)
        for (v, _, C) in self.variables:
            mlp("MLPs." + v, L, S, C)
        return out


def pad4(n: int) -> int:
    return (int(n) + 3) // 4 * 4


def is_tail_key(key: str) -> bool:
    """Small Linears fed by an encoder's BatchNorm block ([latent, hidden]): the engine allocates their rows rounded up to 4
    (zero rows), so that the grouped tail kernel can produce aligned partial products for any latent size."""
    return key.startswith("encoders.") and key.endswith((".layer_out.weight", ".layer_out.bias", ".FC_mean.weight", ".FC_mean.bias",
                                                         ".FC_var.weight", ".FC_var.bias"))


def gnn_conv_keys(prefix: str, conv: str):
    """(weight applied to the aggregate, bias, weight applied to the node itself or None) of one conv layer:
    GraphConv lin_rel(+bias) / lin_root, SAGEConv lin_l(+bias) / lin_r, GCNConv lin + bias (torch_geometric names)."""
    if conv == "GC":
        return prefix + ".lin_rel.weight", prefix + ".lin_rel.bias", prefix + ".lin_root.weight"
    if conv == "SAGE":
        return prefix + ".lin_l.weight", prefix + ".lin_l.bias", prefix + ".lin_r.weight"
    if conv == "GCN":
        return prefix + ".lin.weight", prefix + ".bias", None
    if conv == "GAT":
        # The reference lists GATConv in flexGCN's table (modules.py:221-226) but cannot run it: flexGCN.forward feeds every conv the
        # batched node features [B, nodes, C] with one shared edge_index (modules.py:251-262, a "static graph" in torch_geometric's
        # terms), which GraphConv / SAGEConv / GCNConv broadcast over and GATConv refuses (its forward asserts ``x.dim() == 2``:
        # "Static graphs not supported in 'GATConv'"); accordingly the CLI does not offer it (__main__.py:536-540).  There is no
        # reference behaviour to reproduce, so the engine refuses at construction instead of at the first forward.
        raise ValueError("gnn_conv_type 'GAT': torch_geometric's GATConv does not accept the batched [B, nodes, C] node features that "
                         f"the reference's flexGCN feeds its convolutions (static graphs unsupported). Choose one of: {list(GNN_CONVS)}")
    raise ValueError(f"Unknown convolution type {conv!r}. Choose one of: {list(GNN_CONVS)}")


def is_buffer_key(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def spec_from_dataset(model: str, config: dict, dataset, target_variables, batch_variables=None,
                      surv_event_var=None, surv_time_var=None, use_loss_weighting=True, input_layers=None,
                      output_layers=None, gnn_conv_type=None) -> ArchSpec:
    """Same derivations as the reference constructors.  ``dataset`` is only read for ``.dat.keys()``,
    ``.features[layer]``, ``.ann[var]`` and ``.variable_types`` (also satisfied by the SimpleNamespace of
    reference inference.py:116-122)."""
    if model not in MODEL_KINDS:
        raise ValueError(f"unknown model class {model!r}")
    if model == "GNN":
        return _gnn_spec(config, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                         use_loss_weighting, gnn_conv_type)
    targets = list(target_variables)
    if surv_event_var is not None and surv_time_var is not None:
        targets = targets + [surv_event_var]          # direct_pred.py:48-49
    variables = targets + list(batch_variables) if batch_variables else targets
    layers = [(name, len(dataset.features[name])) for name in dataset.dat.keys()]   # direct_pred.py:68-71
    var_specs = []
    for v in variables:
        if dataset.variable_types[v] == "numerical":
            var_specs.append((v, "numerical", 1))
        else:
            ann = dataset.ann[v]
            arr = ann.detach().cpu().numpy() if hasattr(ann, "detach") else np.asarray(ann)
            var_specs.append((v, "categorical", int(len(np.unique(arr)))))   # direct_pred.py:100 (NaN counts)
    if model == "MultiTripletNetwork" and dataset.variable_types[targets[0]] == "numerical":
        raise ValueError("The first target variable", targets[0], " must be a categorical variable")
    names = [n for n, _ in layers]
    for lst in (input_layers, output_layers):
        for n in (lst or []):
            if n not in names:
                raise KeyError(f"layer {n!r} is not in the dataset ({names})")
    return ArchSpec(model, layers, int(config["latent_dim"]), float(config["hidden_dim_factor"]),
                    int(config["supervisor_hidden_dim"]), var_specs, surv_event_var, surv_time_var,
                    bool(use_loss_weighting),
                    list(input_layers) if (model == "CrossModalPred" and input_layers) else None,
                    list(output_layers) if (model == "CrossModalPred" and output_layers) else None)


def _var_specs(dataset, variables):
    out = []
    for v in variables:
        if dataset.variable_types[v] == "numerical":
            out.append((v, "numerical", 1))
        else:
            ann = dataset.ann[v]
            arr = ann.detach().cpu().numpy() if hasattr(ann, "detach") else np.asarray(ann)
            out.append((v, "categorical", int(len(np.unique(arr)))))
    return out


def _gnn_spec(config, dataset, target_variables, batch_variables, surv_event_var, surv_time_var, use_loss_weighting,
              gnn_conv_type) -> ArchSpec:
    """GNN.__init__ (reference models/gnn_early.py:58-130): ``dataset`` is a MultiOmicDatasetNW -- node feature
    tensor [n_samples, n_nodes, n_node_features] and one edge_index for all samples."""
    conv = gnn_conv_type if gnn_conv_type is not None else "GC"       # flexGCN's default conv (modules.py:205)
    gnn_conv_keys("x", conv)                                           # validates the choice
    act = str(config.get("activation", "relu"))
    if act not in ("relu", "sigmoid", "leakyrelu", "tanh", "gelu"):
        raise ValueError(f"Invalid activation function string {act!r}")
    targets = list(target_variables)
    if surv_event_var is not None and surv_time_var is not None:
        targets = targets + [surv_event_var]
    variables = targets + list(batch_variables) if batch_variables else targets
    src = getattr(dataset, "multiomic_dataset", dataset)
    x0 = dataset[0][0]
    nodes, nf = int(x0.shape[0]), int(x0.shape[1])
    ei = dataset.edge_index
    ei = ei.detach().cpu().numpy() if hasattr(ei, "detach") else np.asarray(ei)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, n_edges], got {ei.shape}")
    if ei.size and (ei.min() < 0 or ei.max() >= nodes):
        raise ValueError("edge_index refers to nodes outside the node feature tensor")
    gnn = {"nodes": nodes, "node_features": nf, "embedding_dim": int(config["node_embedding_dim"]),
           "num_convs": int(config["num_convs"]), "conv": conv, "act": act, "edge_index": ei.astype(np.int64)}
    if not (1 <= gnn["embedding_dim"] <= 32 and 1 <= nf <= 32):
        raise ValueError("the graph kernels cover node widths 1..32 (reference search space: 4..32, config.py:45)")
    return ArchSpec("GNN", [("nodes", nodes * nf)], int(config["latent_dim"]), 0.0, int(config["supervisor_hidden_dim"]),
                    _var_specs(src, variables), surv_event_var, surv_time_var, bool(use_loss_weighting), gnn=gnn)
