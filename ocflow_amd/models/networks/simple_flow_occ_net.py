"""Module path kept for drop-in imports (reference: models/networks/simple_flow_occ_net.py)."""
from .simple_flow_net import SimpleFlowOccNet

__all__ = ["SimpleFlowOccNet"]
