"""`CostVolumeLayer` -- the module the reference imports from models/networks/cost_volume_net.py but does not ship
(SURVEY F1).  Contract taken from its call sites: `CostVolumeLayer()` / `CostVolumeLayer(10)` constructed once
(flow_occ_net.py:73, cost_volume_flow_occ_net.py:53, flow_occ_net_c.py:26) and called as `layer(f1, f2)` returning
(2d+1)^2 channels (cost_volume_flow_occ_net.py:56 expects 81 for the default d=4)."""
import torch.nn as nn

from ocflow_amd import ops


class CostVolumeLayer(nn.Module):
    def __init__(self, max_displacement=4):
        super().__init__()
        self.max_displacement = int(max_displacement)

    def forward(self, features1, features2, flow=None, occ=None, flow_scale=1.0):
        """`layer(f1, f2)` is the reference contract; the optional flow / occ arguments select the fused
        warp(+mask)+correlation kernel (ops.corr_warp) so callers need not materialise the warped features."""
        return ops.corr_warp(features1, features2, flow, occ, self.max_displacement, flow_scale, False)

    def extra_repr(self):
        return f"max_displacement={self.max_displacement}"
