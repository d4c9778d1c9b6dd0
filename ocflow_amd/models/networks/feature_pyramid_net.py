"""Feature pyramid encoder (reference: models/networks/feature_pyramid_net.py:5-95).  Convolutions stay in PyTorch-ROCm
(MIOpen -> MFMA); BatchNorm + LeakyReLU run as the fused HIP passes of ops.batchnorm_lrelu (csrc/bn.hip) in the channels_last
configuration.  Module structure, parameter and buffer names are the reference's (state_dict compatibility)."""
import os

import torch
import torch.nn as nn

from ocflow_amd import conv as nconv
from ocflow_amd import ops
from ._epilogue import deconv_bias


def _conv3x3_same(conv: nn.Conv2d, x):
    """the stride-1 3x3 convolution of a DoubleConv block (bias-free, BatchNorm follows): on the repo's implicit-GEMM kernels where they
    take the operands (bf16 channels_last, channel counts that are multiples of 8 -- every level but the 196-channel top)"""
    if (conv.bias is None and nconv.enabled() and os.environ.get("OCFLOW_NATIVE_FPN", "1") != "0" and not torch.compiler.is_compiling()
            and conv.weight.shape[0] % 8 == 0 and conv.weight.shape[1] >= 32  # (16 channels: half of every staged 64-byte pixel row is padding)
            and nconv.eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)):
        return nconv.conv3x3_bias_act(x, conv.weight, None, 1.0, 1)
    if not torch.compiler.is_compiling() and os.environ.get("OCFLOW_NATIVE_FPN", "1") != "0" and nconv.c16_eligible(x, conv):
        return nconv.conv3x3_c16(x, conv.weight)  # the full-resolution 16 -> 16 layer: forward + input gradient (csrc/conv_c16.hip)
    return conv(x)


def _bn_act(bn: nn.BatchNorm2d, act: nn.LeakyReLU, x, groups: int = 1):
    # OCFLOW_UNFUSED_BN=1: the reference's own op chain (A/B switch for measurements and bisecting, not a fallback)
    if (os.environ.get("OCFLOW_UNFUSED_BN", "0") == "1" or torch.compiler.is_compiling()) and groups == 1:
        return act(bn(x))
    return ops.batchnorm_lrelu(x, bn, act.negative_slope, groups)

_SLOPE = 0.1
_WIDTHS = (3, 16, 32, 64, 96, 128, 196)


def _conv_bn_act(cin, cout, k, stride):
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout),
            nn.LeakyReLU(_SLOPE)]


class DoubleConv(nn.Module):
    """stride-2 conv + BN + LeakyReLU, then stride-1 conv + BN + LeakyReLU (keys double_conv.{0,1,3,4}.*)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.double_conv = nn.Sequential(*_conv_bn_act(in_channels, out_channels, 3, 2),
                                         *_conv_bn_act(out_channels, out_channels, 3, 1))

    def forward(self, x, groups: int = 1):
        s = self.double_conv
        if not torch.compiler.is_compiling() and os.environ.get("OCFLOW_NATIVE_FPN", "1") != "0" and nconv.c3_eligible(x, s[0]):
            h = nconv.conv3x3s2_c3(x, s[0].weight)  # the first layer (3 -> 16, stride 2) on the direct matrix-core kernel
        else:
            h = s[0](x)
        x = _bn_act(s[1], s[2], h, groups)
        return _bn_act(s[4], s[5], _conv3x3_same(s[3], x), groups)


class Upsample(nn.Module):
    """transposed conv to the skip tensor's size + BN + LeakyReLU + skip add (keys deconv.*, batchnorm.*)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.deconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1)
        self.batchnorm = nn.BatchNorm2d(out_channels)
        self.leaky = nn.LeakyReLU(negative_slope=_SLOPE)

    def forward(self, x, y, groups: int = 1):
        return _bn_act(self.batchnorm, self.leaky, deconv_bias(self.deconv, x, y.size()), groups) + y


class FeaturePyramidNet(nn.Module):
    def __init__(self):
        super().__init__()
        for lvl in range(1, 7):
            setattr(self, f"layer{lvl}", DoubleConv(_WIDTHS[lvl - 1], _WIDTHS[lvl]))
        self.pyr_top = nn.Sequential(nn.Conv2d(196, 196, kernel_size=1, bias=False), nn.BatchNorm2d(196), nn.LeakyReLU(_SLOPE))
        for lvl in range(5, 1, -1):
            setattr(self, f"upsample{lvl}", Upsample(_WIDTHS[lvl + 1], _WIDTHS[lvl]))

    def forward(self, x, groups: int = 1):
        skips = []
        for lvl in range(1, 7):
            x = getattr(self, f"layer{lvl}")(x, groups)
            skips.append(x)
        p = _bn_act(self.pyr_top[1], self.pyr_top[2], self.pyr_top[0](skips[5]), groups)
        pyramid = [p]
        for lvl in range(5, 1, -1):
            p = getattr(self, f"upsample{lvl}")(p, skips[lvl - 1], groups)
            pyramid.append(p)
        return pyramid  # [p6, p5, p4, p3, p2] with 196, 128, 96, 64, 32 channels

    def pairs(self, x1, x2) -> bool:
        """does `forward_pair` take the two images as one batch?"""
        return (x1.is_cuda and x1.shape == x2.shape and x1.dim() == 4 and os.environ.get("OCFLOW_UNPAIRED_FPN", "0") != "1"
                and not torch.compiler.is_compiling()
                and os.environ.get("OCFLOW_UNFUSED_BN", "0") != "1"
                and (not self.training or x1.is_contiguous(memory_format=torch.channels_last) or x1.shape[1] == 3))

    def forward_pair(self, x1, x2):
        """the pyramids of both images of a pair: `(self(x1), self(x2))` (flow_net.py:100-101, flow_occ_net.py:149-150), computed as ONE
        batch of 2B images on the GPU.  The convolutions do not couple samples; every BatchNorm runs with two statistics groups
        (ops.batchnorm_lrelu(groups=2): per-half statistics, running buffers updated half after half, as the two calls would), so
        the values are those of the two calls while the 17 convolutions / BatchNorm sites per image -- most of them launch-bound at
        the coarse levels -- are launched once instead of twice.  OCFLOW_UNPAIRED_FPN=1, CPU tensors and NCHW layouts: two calls."""
        if not self.pairs(x1, x2):
            return self(x1), self(x2)
        B = x1.shape[0]
        both = ops.pair_input(x1, x2) if ops.pair_input_eligible(x1, x2) else torch.cat([x1, x2], 0)
        pyr = self(both, 2 if self.training else 1)
        halves = [p.view(2, B, *p.shape[1:]).unbind(0) for p in pyr]
        return [h[0] for h in halves], [h[1] for h in halves]
