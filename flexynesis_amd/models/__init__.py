"""Model classes of the hot path, importable by the names the reference's callers check
(reference flexynesis/models/__init__.py:1-13, main.py:159,241,526)."""
from .direct_pred import DirectPred
from .supervised_vae import supervised_vae
from .triplet_encoder import MultiTripletNetwork
from .crossmodal_pred import CrossModalPred
from .gnn import GNN

__all__ = ["DirectPred", "supervised_vae", "MultiTripletNetwork", "CrossModalPred", "GNN"]
