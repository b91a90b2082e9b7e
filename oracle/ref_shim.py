"""Import shim for the REAL reference (flexynesis @ /root/reference) -- test infrastructure only.

This module is used ONLY in the build container, ONLY by ``oracle/gen_goldens.py`` and the
``-m "not gpu"`` pinning tests, to import the reference's own hot-path classes so that
  (1) golden vectors can be generated from the reference itself, and
  (2) the CPU restatement in ``oracle/restate.py`` can be pinned against it.

``/root/reference`` does not exist on the GPU box; nothing that runs there imports this file
(``available()`` returns False and callers skip).  No reference source is copied: the shim only
fabricates empty stand-ins for third-party packages the image lacks (lightning, captum,
torch_geometric, skopt, ...) so that ``flexynesis.modules`` / ``flexynesis.data`` /
``flexynesis.models.*`` import; none of those packages' code is on the hot path
(SURVEY.md section 8c).  ``flexynesis.main`` is never imported: it sets
``torch.set_float32_matmul_precision("medium")`` at import (reference main.py:24), which would
silently turn the fp32 oracle into a bf16 one on AMX CPUs.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = (
    "lightning", "captum", "torch_geometric", "skopt", "seaborn", "sksurv", "umap", "xgboost",
    "community", "ot", "lifelines", "plotnine", "geomloss", "IPython", "h5py", "papermill",
    "louvain", "leidenalg", "igraph", "statsmodels", "adjustText", "mplcursors",
)


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "flexynesis"))


class _AnyAttr(types.ModuleType):
    """Module whose every attribute is a harmless placeholder class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AnyAttr(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Install the stub finder + a LightningModule stand-in and put the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference not present at %s (this is expected on the GPU box)" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference mount is read-only by policy
    import torch
    from torch import nn

    for root in _STUB_ROOTS:
        try:
            __import__(root)
        except Exception:
            pass
    sys.meta_path.insert(0, _StubFinder())

    import lightning  # noqa: F401  (stub unless really installed)

    class LightningModule(nn.Module):
        """Minimal stand-in: nn.Module + no-op logging + a device property
        (needed by reference supervised_vae.py:545)."""

        def log_dict(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    lightning.LightningModule = LightningModule
    lightning.seed_everything = lambda *a, **k: None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def load():
    """Return a namespace with the reference's hot-path symbols."""
    install()
    import torch

    torch.set_float32_matmul_precision("highest")
    from flexynesis.modules import MLP, Encoder, Decoder, cox_ph_loss
    from flexynesis.data import MultiOmicDataset, TripletMultiOmicDataset
    from flexynesis.models.direct_pred import DirectPred
    from flexynesis.models.supervised_vae import supervised_vae
    from flexynesis.models.triplet_encoder import MultiTripletNetwork
    from flexynesis.models.crossmodal_pred import CrossModalPred

    torch.set_float32_matmul_precision("highest")
    return types.SimpleNamespace(
        MLP=MLP, Encoder=Encoder, Decoder=Decoder, cox_ph_loss=cox_ph_loss,
        MultiOmicDataset=MultiOmicDataset, TripletMultiOmicDataset=TripletMultiOmicDataset,
        DirectPred=DirectPred, supervised_vae=supervised_vae, MultiTripletNetwork=MultiTripletNetwork,
        CrossModalPred=CrossModalPred,
    )
