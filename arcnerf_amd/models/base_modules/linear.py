"""DenseLayer / SirenLayer (arcnerf/models/base_modules/linear.py:11-71): nn.Linear + activation.  Same parameters and state_dict
keys as torch.nn.Linear; on fp32 CUDA tensors the product, its gradients and their gradients run on the hand-written f32-MFMA kernels
of csrc/gemm.hip (ops.autograd.linear), the activation stays a torch module (differentiable to any order)."""
import math

import torch
import torch.nn as nn

from .activation import Sine


class Linear(nn.Linear):
    """torch.nn.Linear whose forward runs ops.autograd.linear (the last layer of GeoNet, the tone mappers of HDR-NeRF)"""

    def forward(self, x):
        from ...ops.autograd import linear
        return linear(x, self.weight, self.bias)


class DenseLayer(Linear):
    def __init__(self, input_dim, out_dim, activation=None, bias=True):
        super().__init__(input_dim, out_dim, bias=bias)
        self.activation = activation if activation is not None else nn.ReLU(inplace=True)

    def forward(self, x):
        if type(self.activation) is nn.ReLU:
            from ...ops.autograd import linear_relu
            return linear_relu(x, self.weight, self.bias)      # bias + ReLU in the product's epilogue, mask folded into its backward
        if type(self.activation) is nn.Softplus and self.activation.threshold == 20:
            from ...ops.autograd import linear_act_nograd
            y = linear_act_nograd(x, self.weight, self.bias, 'softplus', float(self.activation.beta))   # graph-free passes only
            if y is not None:
                return y
            from ...ops.autograd import linear_softplus, softplus
            y = linear_softplus(x, self.weight, self.bias, float(self.activation.beta))   # activation in the epilogue, twice differentiable
            if y is not None:
                return y
            return softplus(super().forward(x), float(self.activation.beta))
        return self.activation(super().forward(x))


class SirenLayer(Linear):
    def __init__(self, input_dim, out_dim, is_first=False, bias=True):
        self.is_first, self.input_dim, self.w0, self.c = is_first, input_dim, 30, 6
        super().__init__(input_dim, out_dim, bias=bias)
        self.activation = Sine(self.w0)

    def reset_parameters(self):
        super().reset_parameters()
        with torch.no_grad():
            std = (1 / self.input_dim) if self.is_first else (math.sqrt(self.c / self.input_dim) / self.w0)
            self.weight.uniform_(-std, std)

    def forward(self, x):
        return self.activation(super().forward(x))
