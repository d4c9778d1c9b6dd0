"""`FlowOccNet` (model 'flowoccnet', the "woc" network of image/woc.PNG): per level  warp -> occlusion estimator ->
mask -> cost volume -> flow estimator  (reference: models/networks/flow_occ_net.py:11-189).

MI355X mapping of one level: ops.warp (HIP, the warped features feed the occlusion estimator so they must exist as a
tensor), then ONE correlation kernel that applies the mask while staging the search window (ops.corr_warp with
occ=...), replacing the reference's in-place `warped2 *= occ` + CostVolumeLayer (flow_occ_net.py:167-168)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import ops
from ._epilogue import cat_aligned, conv_bias, conv_lrelu, conv_lrelu_chain, deconv_bias
from .context_net import ContextNetwork
from .cost_volume_net import CostVolumeLayer
from .feature_pyramid_net import FeaturePyramidNet
from .flow_net import FLOW_IN_CHANNELS, LEVELS, OpticalFlowEstimator, kaiming_init_

OCC_IN_CHANNELS = (392, 258, 194, 130, 66)  # 2*C_level (+ 1 occ-feature + 1 occ below level 6)


class OcclusionEstimator(nn.Module):
    _WIDTHS = (128, 96, 64, 32)

    def __init__(self, input_channels, level, highest_resolution=False):
        super().__init__()
        self.highest_res = highest_resolution
        cin = input_channels
        for i, cout in enumerate(self._WIDTHS, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, 3, 1, 1))
            cin = cout
        self.feat_layer = nn.Conv2d(32, 16, 3, 1, 1)
        self.mask_layer = nn.Conv2d(16, 1, 3, 1, 1)
        self.upconv1 = nn.ConvTranspose2d(16, 1, 3, 2, 1)
        self.upconv2 = nn.ConvTranspose2d(1, 1, 3, 2, 1)

    def forward(self, features, h_up=None, w_up=None):
        feat = conv_lrelu_chain([getattr(self, f"conv{i}") for i in range(1, 5)] + [self.feat_layer], features, 0.1)
        logit = conv_bias(self.mask_layer, feat)
        if self.highest_res:
            return torch.sigmoid(10 * logit)
        size = (h_up, w_up)
        occ = torch.sigmoid(logit)
        return occ, deconv_bias(self.upconv2, occ, size, sigmoid=True), deconv_bias(self.upconv1, feat, size, sigmoid=True)


class FlowOccNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.correlation_layer = CostVolumeLayer()
        self.feature_pyramid_network = FeaturePyramidNet()
        self.opticalflow_estimators = nn.ModuleList(
            OpticalFlowEstimator(c, level=l, highest_resolution=(l == 2)) for c, l in zip(FLOW_IN_CHANNELS, LEVELS))
        self.occlusion_estimators = nn.ModuleList(
            OcclusionEstimator(c, level=l, highest_resolution=(l == 2)) for c, l in zip(OCC_IN_CHANNELS, LEVELS))
        self.context_network = ContextNetwork(34)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.upsample2 = nn.Upsample(scale_factor=4, mode="bilinear")
        kaiming_init_(self)

    def warp(self, img, flow):
        """feature warp, align_corners=False variant (flow_occ_net.py:96-126)"""
        return ops.warp(img, flow, align_corners=False)

    lowres_head = True  # forward(..., upsample=False) returns the quarter-resolution (flow, occlusion) the two nn.Upsample modules would take

    def forward(self, inputs, training=False, mask=None, upsample=True):
        pyr1, pyr2 = self.feature_pyramid_network.forward_pair(inputs[:, 0:3], inputs[:, 3:6])
        flow_up = flow_feat_up = occ_up = occ_feat_up = None
        flow = flow_feat = occ = None
        for i, level in enumerate(LEVELS):
            f1, f2 = pyr1[i], pyr2[i]
            last = level == 2
            size = () if last else (pyr1[i + 1].size(2), pyr1[i + 1].size(3))
            if flow_up is None:
                warped2 = f2
                occ_in = cat_aligned([f1, warped2])
            else:
                warped2 = ops.warp(f2, flow_up, False, flow_scale=20.0 / (2.0 ** level))
                occ_in = cat_aligned([f1, warped2.to(f1.dtype), occ_feat_up.to(f1.dtype), occ_up.to(f1.dtype)])
            if last:
                occ = self.occlusion_estimators[i](occ_in)
            else:
                occ, occ_up, occ_feat_up = self.occlusion_estimators[i](occ_in, *size)
            # corr(feat1, warped2 * occ): the mask multiply happens while the window is staged into LDS
            corr = self.correlation_layer(f1, warped2, None, occ)
            parts = [corr.to(f1.dtype), f1]
            if flow_up is not None:
                parts += [flow_up.to(f1.dtype), flow_feat_up.to(f1.dtype)]
            x = cat_aligned(parts)
            if last:
                flow, flow_feat = self.opticalflow_estimators[i](x)
            else:
                flow, flow_up, flow_feat_up = self.opticalflow_estimators[i](x, *size)
        refined = flow + self.context_network(cat_aligned([flow_feat, flow.to(flow_feat.dtype)]))
        if not upsample:
            return refined, occ
        return self.upsample1(refined), self.upsample2(occ)
