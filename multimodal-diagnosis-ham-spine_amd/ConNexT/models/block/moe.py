"""Sparsely-gated mixture of KAN experts (reference ConNexT/models/block/moe.py:171-291) on the hamspine kernels.

Gating (noisy top-k, load / importance cv^2 loss) is one fused node (hs_moe_gate_fwd / _bwd).  Dispatch is SPARSE as in the
reference (moe.py:48-112): expert e receives only the rows whose gate is > 0 (hs_moe_dispatch_index builds the per-expert
row lists on the device, hs_rows_gather collects them), runs its KAN stack on that k/E share of the batch, and
hs_rows_scatter_add stitches the gate-weighted outputs back.  Like the reference, the dispatcher reads the experts' batch
sizes back to the host (`.tolist()`, moe.py:60).  Training-mode noise comes from hamspine's counter RNG (it cannot equal
torch.randn_like bit for bit); a caller that needs the reference's draw reproduced sets `moe.gating_noise` to the recorded
(batch, experts) standard-normal tensor before the call (consumed once)."""
import torch
import torch.nn as nn

from hamspine import kan as K

from .kan1 import KAN1


class SparseDispatcher(object):
    """moe.py:17-112: `dispatch` -> one input tensor per expert (the rows with gates[b, e] > 0, ascending b), `combine` ->
    the gate-weighted sum back in batch order, `expert_to_gates` -> the nonzero gate values per expert."""

    def __init__(self, num_experts, gates):
        self._gates = gates
        self._num_experts = num_experts
        self._idx, self._part_sizes = K.moe_dispatch_index(gates)

    def dispatch(self, inp):
        return [K.RowsGatherFn.apply(inp, self._idx[e], n) for e, n in enumerate(self._part_sizes)]

    def combine(self, expert_out, multiply_by_gates=True):
        if not multiply_by_gates:
            raise NotImplementedError("combine(multiply_by_gates=False) is not implemented")
        return K.MoESparseCombineFn.apply(self._gates, self._idx, self._part_sizes, self._gates.shape[0], *expert_out)

    def expert_to_gates(self):
        return [self._gates[self._idx[e, :n].long(), e] for e, n in enumerate(self._part_sizes)]


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
        self.gating_noise = None
        assert self.k <= self.num_experts

    def forward(self, x, loss_coef=1e-2):
        if x.dtype != torch.float32:
            x = x.float()
        noisy = bool(self.noisy_gating and self.training)
        noise, self.gating_noise = self.gating_noise, None
        gates, loss = K.MoEGateFn.apply(x, self.w_gate, self.w_noise, self.k, noisy, float(loss_coef), noise if noisy else None)
        dispatcher = SparseDispatcher(self.num_experts, gates)
        expert_inputs = dispatcher.dispatch(x)
        outs = [self.experts[i](expert_inputs[i]) for i in range(self.num_experts)]
        return dispatcher.combine(outs), loss
