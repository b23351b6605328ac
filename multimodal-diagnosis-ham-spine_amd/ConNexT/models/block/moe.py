"""Sparsely-gated mixture of KAN experts (reference ConNexT/models/block/moe.py:171-291) on the hamspine kernels.

Gating (noisy top-k, load / importance cv^2 loss) is one fused node; experts are evaluated densely and combined with
their gate columns -- rows whose gate is zero contribute exactly nothing, so outputs and gradients equal the reference's
gather/scatter dispatch (KAN experts have no cross-sample coupling) without the host sync of `.tolist()`.
Training-mode noise comes from hamspine's counter RNG (cannot equal torch.randn_like bit-for-bit)."""
import torch
import torch.nn as nn

from hamspine import kan as K

from .kan1 import KAN1


class SparseDispatcher(object):
    """API of moe.py:17-112 on top of the dense gate matrix: dispatch() hands every expert the full batch, combine() weights
    by the gates (zero for rows the reference would not have dispatched)."""

    def __init__(self, num_experts, gates):
        self._gates = gates
        self._num_experts = num_experts

    def dispatch(self, inp):
        return [inp for _ in range(self._num_experts)]

    def combine(self, expert_out, multiply_by_gates=True):
        if not multiply_by_gates:
            raise NotImplementedError("combine(multiply_by_gates=False) is not implemented")
        return K.MoECombineFn.apply(self._gates, *expert_out)

    def expert_to_gates(self):
        return [self._gates[:, e] for e in range(self._num_experts)]


class MoE(nn.Module):
    def __init__(self, input_size, output_size, num_experts, hidden_size, noisy_gating=True, k=4, layers_hidden=None,
                 grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0, scale_spline=1.0):
        super().__init__()
        self.noisy_gating, self.num_experts = noisy_gating, num_experts
        self.output_size, self.input_size, self.hidden_size, self.k = output_size, input_size, hidden_size, k
        layers = [input_size, 512, 128, 32, output_size] if layers_hidden is None else layers_hidden
        self.experts = nn.ModuleList(KAN1(layers_hidden=layers, grid_size=grid_size, spline_order=spline_order,
                                          scale_noise=scale_noise, scale_base=scale_base, scale_spline=scale_spline)
                                     for _ in range(num_experts))
        self.w_gate = nn.Parameter(torch.zeros(input_size, num_experts), requires_grad=True)
        self.w_noise = nn.Parameter(torch.zeros(input_size, num_experts), requires_grad=True)
        self.softplus = nn.Softplus()
        self.softmax = nn.Softmax(1)
        self.register_buffer("mean", torch.tensor([0.0]))
        self.register_buffer("std", torch.tensor([1.0]))
        assert self.k <= self.num_experts

    def forward(self, x, loss_coef=1e-2):
        if x.dtype != torch.float32:
            x = x.float()
        noisy = bool(self.noisy_gating and self.training)
        gates, loss = K.MoEGateFn.apply(x, self.w_gate, self.w_noise, self.k, noisy, float(loss_coef))
        dispatcher = SparseDispatcher(self.num_experts, gates)
        expert_inputs = dispatcher.dispatch(x)
        outs = [self.experts[i](expert_inputs[i]) for i in range(self.num_experts)]
        return dispatcher.combine(outs), loss
