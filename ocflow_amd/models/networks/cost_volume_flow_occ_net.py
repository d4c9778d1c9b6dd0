"""`FlowOccNetCV` (model 'pwoc'): FlowNetCV + per-level occlusion head; the warped second feature map is multiplied
by the up-sampled occlusion map before the correlation (reference:
models/networks/cost_volume_flow_occ_net.py:26-271).  On MI355X warp * occ -> cost volume is ONE kernel
(ops.corr_warp); neither `warp5` nor `warp5 * up_occ6` (:204-206) is written to HBM."""
import os

import torch
import torch.nn as nn

from ._epilogue import cat_aligned, conv_bias, deconv_bias, deconv_bias_range
from .cost_volume_flow_net import (DENSE_TOTAL, DENSE_WIDTHS, ENC_WIDTHS, WARP_SCALE, FlowNetCV, build_context, build_encoder, context,
                                   conv, deconv, dense_block, encode, level_features, predict_flow)
from .cost_volume_net import CostVolumeLayer
from .flow_net import kaiming_init_
from ocflow_amd import conv as nconv
from ocflow_amd import ops


def predict_occlusion(in_planes):
    return nn.Sequential(nn.Conv2d(in_planes, 1, kernel_size=3, stride=1, padding=1, bias=True), nn.Sigmoid())


class FlowOccNetCV(nn.Module):
    def __init__(self):
        super().__init__()
        build_encoder(self)
        self.corr = CostVolumeLayer()
        self.leakyRELU = nn.LeakyReLU(0.1)
        nd = (2 * 4 + 1) ** 2
        for lvl in range(6, 1, -1):
            od = nd if lvl == 6 else nd + ENC_WIDTHS[lvl] + 4 + 1
            cin = od
            for j, w in enumerate(DENSE_WIDTHS):
                setattr(self, f"conv{lvl}_{j}", conv(cin, w))
                cin += w
            setattr(self, f"predict_flow{lvl}", predict_flow(cin))
            setattr(self, f"predict_occ{lvl}", predict_occlusion(cin))
            if lvl > 2:
                setattr(self, f"upflow{lvl}", deconv(2, 2))
                setattr(self, f"upocc{lvl}", deconv(1, 1))
                setattr(self, f"upfeat{lvl}", deconv(cin, 2))
        build_context(self, nd + ENC_WIDTHS[2] + 4 + 1 + DENSE_TOTAL)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.upsample2 = nn.Upsample(scale_factor=4, mode="bilinear")
        kaiming_init_(self)

    def warp(self, img, flow):
        """feature warp, align_corners=False variant (cost_volume_flow_occ_net.py:137-167)"""
        return ops.warp(img, flow, align_corners=False)

    lowres_head = True  # forward(x, upsample=False) returns the quarter-resolution (flow, occlusion)

    def forward(self, x, upsample=True):
        # aligned concatenations + the repo's convolution kernels in the bf16 channels_last configuration, exactly as FlowNetCV.forward
        # (cost_volume_flow_net.py): the library's transposed convolution `upfeat` is handed the unpadded channel range
        native = (os.environ.get("OCFLOW_PAD_PWC", "auto") == "auto" and nconv.enabled() and x.is_cuda and torch.is_autocast_enabled()
                  and not torch.compiler.is_compiling())
        c1, c2 = FlowNetCV.pyramids(self, x)  # (both images through the encoder as one batch: no batch-coupled layer)
        up_flow = up_occ = up_feat = flow = occ = feats = None
        for lvl in range(6, 1, -1):
            f1, f2 = c1[lvl - 1], c2[lvl - 1]
            # LeakyReLU(cost volume of f1 against warp(f2) * up_occ) [| f1 | up_flow | up_occ | up_feat] through the dense block
            # (cost_volume_flow_occ_net.py:188-213); see cost_volume_flow_net.level_features
            d = self.corr.max_displacement
            if lvl == 6:
                corr_fn = lambda leaky_slope, grad_premasked, a=f1, b=f2: ops.corr_warp(
                    a, b, None, None, d, leaky_slope=leaky_slope, grad_premasked=grad_premasked)
                others = []
            else:
                corr_fn = lambda leaky_slope, grad_premasked, a=f1, b=f2, fl=up_flow, oc=up_occ, sc=WARP_SCALE[lvl]: ops.corr_warp(
                    a, b, fl, oc, d, sc, False, leaky_slope=leaky_slope, grad_premasked=grad_premasked)
                others = [f1, up_flow, up_occ, up_feat]
            feats = level_features(self, lvl, corr_fn, others, native, f1)
            if native:
                flow = conv_bias(getattr(self, f"predict_flow{lvl}"), feats)
                po = getattr(self, f"predict_occ{lvl}")
                occ = po[1](conv_bias(po[0], feats))
            else:
                flow = getattr(self, f"predict_flow{lvl}")(feats)
                occ = getattr(self, f"predict_occ{lvl}")(feats)
            if lvl > 2:
                up_flow = getattr(self, f"upflow{lvl}")(flow)
                up_occ = getattr(self, f"upocc{lvl}")(occ)
                upf = getattr(self, f"upfeat{lvl}")
                up_feat = deconv_bias_range(upf, feats) if native else upf(feats)
        flow2 = flow + context(self, feats)
        if not upsample:
            return flow2, occ
        return self.upsample1(flow2), self.upsample2(occ)
