"""conv -> bias -> LeakyReLU blocks of the estimator nets with the bias add and the activation in one HIP pass.

The reference writes either `F.leaky_relu(self.convN(x), 0.1)` or `nn.Sequential(nn.Conv2d(..., bias=True), nn.LeakyReLU(0.1))`
(cost_volume_flow_net.py:19-21 `conv`, simple_flow_net.py:55-57).  Module structure, parameter names and values are unchanged:
the convolution runs without its bias and `ops.bias_leaky_relu` applies bias + activation in place on its output."""
import os

import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import ops


def conv_lrelu(conv: nn.Conv2d, x, negative_slope: float = 0.1):
    # OCFLOW_UNFUSED_EPILOGUE=1: the reference's own op chain (A/B switch for measurements and bisecting, not a fallback)
    if conv.bias is None or conv.padding_mode != "zeros" or os.environ.get("OCFLOW_UNFUSED_EPILOGUE", "0") == "1":
        return F.leaky_relu(conv(x), negative_slope)
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return ops.bias_leaky_relu(y, conv.bias, negative_slope)


class ConvLReLU(nn.Sequential):
    """nn.Sequential(nn.Conv2d, nn.LeakyReLU) with the same children / state_dict keys ('0.weight', '0.bias')"""

    def forward(self, x):
        return conv_lrelu(self[0], x, self[1].negative_slope)
