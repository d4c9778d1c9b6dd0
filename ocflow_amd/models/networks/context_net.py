"""Dilated-conv flow refinement (reference: models/networks/context_net.py:5-28).  PyTorch-ROCm convs."""
import torch.nn as nn
import torch.nn.functional as F

from ._epilogue import conv_bias, conv_lrelu_chain

_PLAN = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (2, 1))  # (out channels, dilation)


class ContextNetwork(nn.Module):
    def __init__(self, input_channels):
        super().__init__()
        cin = input_channels
        for i, (cout, dil) in enumerate(_PLAN, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, 3, 1, dil, dil))
            cin = cout

    def forward(self, x):
        x = conv_lrelu_chain([getattr(self, f"conv{i}") for i in range(1, len(_PLAN))], x, 0.1)
        return conv_bias(getattr(self, f"conv{len(_PLAN)}"), x)
