"""`FlowNet` (model 'flownet'): FPN encoder + per-level {warp -> cost volume -> estimator} + context network
(reference: models/networks/flow_net.py:11-131).  On MI355X the warp and the cost volume of a level are ONE kernel
(ops.corr_warp): the reference's `warped2` tensor (flow_net.py:110-112) is never written to HBM."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import ops
from ._epilogue import cat_aligned, conv_bias, conv_lrelu, conv_lrelu_chain, deconv_bias
from .context_net import ContextNetwork
from .feature_pyramid_net import FeaturePyramidNet

LEVELS = (6, 5, 4, 3, 2)
FLOW_IN_CHANNELS = (277, 213, 181, 149, 117)  # 81 + C_level (+ 2 flow + 2 flow-feature below level 6)


def kaiming_init_(module):
    """flow_net.py:51-55: kaiming-normal (fan_in) weights and zero bias on every conv / deconv"""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(m.weight.data, mode="fan_in")
            if m.bias is not None:
                m.bias.data.zero_()


class OpticalFlowEstimator(nn.Module):
    """conv1..conv6 (3x3) -> flow; below the finest level also upconv1 (flow) / upconv2 (features) to the next size"""

    _WIDTHS = (128, 128, 96, 64, 32)

    def __init__(self, input_channels, level, highest_resolution=False):
        super().__init__()
        self.highest_res = highest_resolution
        cin = input_channels
        for i, cout in enumerate(self._WIDTHS, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, 3, 1, 1))
            cin = cout
        self.conv6 = nn.Conv2d(cin, 2, 3, 1, 1)
        self.upconv1 = nn.ConvTranspose2d(2, 2, 3, 2, 1)
        self.upconv2 = nn.ConvTranspose2d(32, 2, 3, 2, 1)

    def forward(self, x, h_up=None, w_up=None):
        x = conv_lrelu_chain([getattr(self, f"conv{i}") for i in range(1, 6)], x, 0.1)
        flow = conv_bias(self.conv6, x)
        if self.highest_res:
            return flow, x
        size = (h_up, w_up)
        return flow, deconv_bias(self.upconv1, flow, size), deconv_bias(self.upconv2, x, size)


class FlowNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.correlation_layer = ops.compute_cost_volume  # attribute kept for parity with flow_net.py:43
        self.feature_pyramid_network = FeaturePyramidNet()
        self.opticalflow_estimators = nn.ModuleList(
            OpticalFlowEstimator(c, level=l, highest_resolution=(l == 2)) for c, l in zip(FLOW_IN_CHANNELS, LEVELS))
        self.context_network = ContextNetwork(34)
        self.upsample = nn.Upsample(scale_factor=4, mode="bilinear")
        kaiming_init_(self)

    def warp(self, img, flow):
        """feature warp, align_corners=False variant (flow_net.py:57-87)"""
        return ops.warp(img, flow, align_corners=False)

    def forward(self, inputs):
        pyr1, pyr2 = self.feature_pyramid_network.forward_pair(inputs[:, 0:3], inputs[:, 3:6])
        return self.decode(pyr1, pyr2)

    def forward_two_way(self, inputs):
        """(forward(inputs), forward(cat(inputs[:, 3:], inputs[:, :3])) without gradient) -- what the occlusion-aware step asks for in two
        calls (models/model.py:376,384) -- from ONE encoder pass where that is exact: with the paired encoder every image is normalised
        by its own statistics group, so the swapped call's pyramids ARE these, swapped; what it would have done to the BatchNorm buffers
        (one more momentum update per group, the second image's group first; num_batches_tracked += 2) is replayed from the statistics
        this pass saved (ops.bn_replay_running_update).  Otherwise (unpaired encoder, layouts the BatchNorm kernels do not take): two
        forwards, as the reference."""
        fpn = self.feature_pyramid_network
        x1, x2 = inputs[:, 0:3], inputs[:, 3:6]
        if not (fpn.pairs(x1, x2) and os.environ.get("OCFLOW_SHARED_ENCODER", "1") != "0"):
            out = self.forward(inputs)
            with torch.no_grad():
                return out, self.forward(torch.cat((x2, x1), dim=1))
        with ops.bn_record_stats() as rec:
            pyr1, pyr2 = fpn.forward_pair(x1, x2)
        sites = sum(1 for m in fpn.modules() if isinstance(m, nn.BatchNorm2d))
        if fpn.training and len(rec) != sites:  # a BatchNorm site went through the framework's own operator: its update cannot be replayed
            out = self.decode(pyr1, pyr2)
            with torch.no_grad():
                return out, self.forward(torch.cat((x2, x1), dim=1))
        out = self.decode(pyr1, pyr2)
        with torch.no_grad():
            if fpn.training:
                ops.bn_replay_running_update(rec, order=(1, 0))
            back = self.decode([t.detach() for t in pyr2], [t.detach() for t in pyr1])
        return out, back

    def decode(self, pyr1, pyr2):
        flow_up = feat_up = flow = feat = None
        for i, level in enumerate(LEVELS):
            f1, f2 = pyr1[i], pyr2[i]
            if flow_up is None:
                corr = ops.compute_cost_volume(f1, f2)
                x = cat_aligned([corr.to(f1.dtype), f1])
            else:
                # warp(feat2, flow_up * 20 / 2^level) fused into the correlation's window gather
                corr = ops.corr_warp(f1, f2, flow_up, None, 4, 20.0 / (2.0 ** level), False)
                x = cat_aligned([corr.to(f1.dtype), f1, flow_up.to(f1.dtype), feat_up.to(f1.dtype)])
            est = self.opticalflow_estimators[i]
            if level == 2:
                flow, feat = est(x)
            else:
                flow, flow_up, feat_up = est(x, pyr1[i + 1].size(2), pyr1[i + 1].size(3))
        refined = flow + self.context_network(cat_aligned([feat, flow.to(feat.dtype)]))
        return self.upsample(refined)  # no *20: flow_net.py:127-128
