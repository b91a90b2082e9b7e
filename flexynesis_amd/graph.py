"""The message-passing operator of one graph as two CSR matrices (host side, built once per model).

flexGCN (reference modules.py:153-262) applies torch_geometric's GraphConv / SAGEConv / GCNConv to node features
``[B, nodes, C]`` with ONE ``edge_index`` shared by all samples (models/gnn_early.py:93-96).  Each of the three
aggregations is ``out[b, i] = sum_{e: dst(e) = i} w_e * x[b, src(e)]`` (flow source_to_target: ``edge_index[0]`` is the
source j, ``edge_index[1]`` the target i) with
  GC    w = 1                                    (GraphConv, aggr='add')
  SAGE  w = 1 / in_degree(i)                     (SAGEConv, aggr='mean')
  GCN   self loops replaced by exactly one self loop of weight 1 per node, then deg^-1/2[j] * deg^-1/2[i] with deg the
        in-degree including the self loop        (gcn_norm + add_remaining_self_loops)
Duplicate edges count as often as they are listed.  The forward kernel walks the edges grouped by target, the
backward (u = A^T dOut) grouped by source.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class GraphOp:
    t_rowptr: torch.Tensor   # int32 [nodes + 1]  edges grouped by target
    t_idx: torch.Tensor      # int32 [E']         source of each edge
    t_w: torch.Tensor        # fp32  [E']
    s_rowptr: torch.Tensor   # int32 [nodes + 1]  edges grouped by source
    s_idx: torch.Tensor      # int32 [E']         target of each edge
    s_w: torch.Tensor        # fp32  [E']
    n_edges: int


def _csr(key: np.ndarray, other: np.ndarray, w: np.ndarray, n: int, device):
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(key, minlength=n), out=rowptr[1:])
    if rowptr[-1] >= 2 ** 31:
        raise ValueError("too many edges for int32 CSR offsets")
    dev = torch.device(device)
    return (torch.from_numpy(rowptr.astype(np.int32)).to(dev), torch.from_numpy(other[order].astype(np.int32)).to(dev),
            torch.from_numpy(w[order].astype(np.float32)).to(dev))


def edge_weights(edge_index: np.ndarray, n_nodes: int, conv: str):
    """(src, dst, w) of the messages ``conv`` sums at dst; weights in fp64."""
    src, dst = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    if conv == "GC":
        w = np.ones(src.size, dtype=np.float64)
    elif conv == "SAGE":
        deg = np.bincount(dst, minlength=n_nodes).astype(np.float64)
        w = 1.0 / deg[dst]
    elif conv == "GCN":
        keep = src != dst
        loops = np.arange(n_nodes, dtype=np.int64)
        src, dst = np.concatenate([src[keep], loops]), np.concatenate([dst[keep], loops])
        deg = np.bincount(dst, minlength=n_nodes).astype(np.float64)
        with np.errstate(divide="ignore"):
            dis = deg ** -0.5
        dis[np.isinf(dis)] = 0.0
        w = dis[src] * dis[dst]
    else:
        raise ValueError(f"Unknown convolution type {conv!r}")
    return src, dst, w


def build(edge_index, n_nodes: int, conv: str, device) -> GraphOp:
    src, dst, w = edge_weights(np.asarray(edge_index), int(n_nodes), conv)
    t = _csr(dst, src, w, n_nodes, device)
    s = _csr(src, dst, w, n_nodes, device)
    return GraphOp(*t, *s, int(src.size))
