"""conv -> bias -> LeakyReLU blocks of the estimator nets with the bias add and the activation in one HIP pass.

The reference writes either `F.leaky_relu(self.convN(x), 0.1)` or `nn.Sequential(nn.Conv2d(..., bias=True), nn.LeakyReLU(0.1))`
(cost_volume_flow_net.py:19-21 `conv`, simple_flow_net.py:55-57).  Module structure, parameter names and values are unchanged:
the convolution runs without its bias and `ops.bias_leaky_relu` applies bias + activation in place on its output."""
import os

import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import ops


def _conv_nobias(conv: nn.Conv2d, x):
    """the convolution proper.  3x3 / stride-1 / same layers in the bf16 channels_last throughput configuration take their weight
    gradient from the hand-written MFMA kernel (ops.conv3x3: ocf_conv3x3_wgrad); forward and input gradient stay MIOpen's.
    Opt-in (OCFLOW_NATIVE_WGRAD=1): the first version of that kernel is parity-green and level with MIOpen's kernel + helper launches
    on the large layers (profiles/r2_*_wgrad.txt) but not yet a net win for the whole step."""
    if _native_wgrad() and ops.conv3x3_wgrad_eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
        return ops.conv3x3(x, conv.weight, conv.dilation[0])
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _native_wgrad() -> bool:
    return os.environ.get("OCFLOW_NATIVE_WGRAD", "0") == "1"


def conv_lrelu(conv: nn.Conv2d, x, negative_slope: float = 0.1):
    # OCFLOW_UNFUSED_EPILOGUE=1: the reference's own op chain (A/B switch for measurements and bisecting, not a fallback)
    if conv.bias is None or conv.padding_mode != "zeros" or _unfused():
        return F.leaky_relu(conv(x), negative_slope)
    return ops.bias_leaky_relu(_conv_nobias(conv, x), conv.bias, negative_slope)


def _unfused() -> bool:
    return os.environ.get("OCFLOW_UNFUSED_EPILOGUE", "0") == "1"


def conv_bias(conv: nn.Conv2d, x):
    """nn.Conv2d(bias=True) without activation (prediction layers): bias-free convolution + in-place bias kernel"""
    if conv.bias is None or conv.padding_mode != "zeros" or _unfused():
        return conv(x)
    return ops.bias_add(_conv_nobias(conv, x), conv.bias)


def deconv_bias(deconv: nn.ConvTranspose2d, x, output_size=None):
    """nn.ConvTranspose2d(bias=True)(x, output_size=...) as a bias-free transposed convolution + in-place bias kernel"""
    if deconv.bias is None or deconv.padding_mode != "zeros" or _unfused():
        return deconv(x, output_size=output_size)
    output_padding = deconv._output_padding(x, output_size, deconv.stride, deconv.padding, deconv.kernel_size, 2, deconv.dilation)
    y = F.conv_transpose2d(x, deconv.weight, None, deconv.stride, deconv.padding, output_padding, deconv.groups, deconv.dilation)
    return ops.bias_add(y, deconv.bias)


class ConvLReLU(nn.Sequential):
    """nn.Sequential(nn.Conv2d, nn.LeakyReLU) with the same children / state_dict keys ('0.weight', '0.bias')"""

    def forward(self, x):
        return conv_lrelu(self[0], x, self[1].negative_slope)
