"""Latent connectors with the reference's names and signatures (models/nn_lib.py:136-286) on HIP kernels."""
import torch
import torch.nn as nn

from . import ops


class LinearConnector(nn.Module):
    """nn_lib.py:136-162: latent -> decoder initial state [1,B,out] (LSTM: (h, c))."""

    def __init__(self, input_size, output_size, is_lstm, has_bias=True):
        super().__init__()
        if is_lstm:
            self.linear_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.linear_c = nn.Linear(input_size, output_size, bias=has_bias)
        else:
            self.linear = nn.Linear(input_size, output_size, bias=has_bias)
        self.is_lstm = is_lstm

    def forward(self, inputs):
        if self.is_lstm:
            h = ops.linear(inputs, self.linear_h.weight, self.linear_h.bias).unsqueeze(0)
            c = ops.linear(inputs, self.linear_c.weight, self.linear_c.bias).unsqueeze(0)
            return (h, c)
        return ops.linear(inputs, self.linear.weight, self.linear.bias).unsqueeze(0)

    def get_w(self):
        return self.linear_h.weight if self.is_lstm else self.linear.weight


class Hidden2Gaussian(nn.Module):
    """nn_lib.py:188-223: hidden state -> (mu, logvar)."""

    def __init__(self, input_size, output_size, is_lstm=False, has_bias=True):
        super().__init__()
        if is_lstm:
            self.mu_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.mu_c = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar_c = nn.Linear(input_size, output_size, bias=has_bias)
        else:
            self.mu = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar = nn.Linear(input_size, output_size, bias=has_bias)
        self.is_lstm = is_lstm

    def forward(self, inputs):
        if self.is_lstm:
            h, c = inputs
            if h.dim() == 3:
                h, c = h.squeeze(0), c.squeeze(0)
            mu = ops.linear(h, self.mu_h.weight, self.mu_h.bias) + ops.linear(c, self.mu_c.weight, self.mu_c.bias)
            lv = ops.linear(h, self.logvar_h.weight, self.logvar_h.bias) + \
                ops.linear(c, self.logvar_c.weight, self.logvar_c.bias)
            return mu, lv
        if inputs.dim() == 3:
            inputs = inputs.squeeze(0)
        x = inputs.squeeze(0)
        return ops.linear(x, self.mu.weight, self.mu.bias), ops.linear(x, self.logvar.weight, self.logvar.bias)


class GumbelConnector(nn.Module):
    """nn_lib.py:226-265.  The reference draws the uniform noise with CPU `torch.rand(logits.size())` and copies
    it to the device every call (nn_lib.py:231-233); `u=None` keeps exactly that draw (so CPU seeds reproduce the
    reference), `u=<device tensor>` injects the noise without the H2D copy."""

    def sample_uniform(self, logits):
        return torch.rand(logits.size()).to(logits.device)

    def forward(self, logits, temperature, use_gpu, hard=False, return_max_id=False, straight_through=False, u=None):
        if u is None:
            u = self.sample_uniform(logits)
        y, y_hard = ops.GumbelSoftmax.apply(logits, u, float(temperature), bool(straight_through or hard))
        return (y, y_hard) if return_max_id else y


class GreedyConnector(nn.Module):
    """nn_lib.py:268-286: one-hot of the arg-max (index work only; no gradient)."""

    def forward(self, logits, use_gpu, return_max_id=False):
        _, y_hard = torch.max(logits, dim=1, keepdim=True)
        y = torch.zeros_like(logits).scatter_(1, y_hard, 1.0)
        return (y, y_hard) if return_max_id else y
