"""Same module path and names as the reference's models/networks/correlation_layer.py:7,42 -- the two functions are
the HIP operators."""
from ocflow_amd.ops import compute_cost_volume, normalize_features

__all__ = ["compute_cost_volume", "normalize_features"]
