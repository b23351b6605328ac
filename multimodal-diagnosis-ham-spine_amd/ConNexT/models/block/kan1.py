"""KANLinear / KAN1 (reference ConNexT/models/block/kan1.py) with the same parameters (base_weight, spline_weight,
spline_scaler), buffer (grid) and initialisation; the forward/backward run on the hamspine kernels in f32."""
import math
import types

import torch

from hamspine import kan as K


# `base_activation` classes the feature kernel implements (hs_kan_features_*: base_act)
_BASE_ACT_CODES = {torch.nn.SiLU: "silu", torch.nn.GELU: "gelu", torch.nn.ReLU: "relu", torch.nn.Identity: "identity"}


class KANLinear(torch.nn.Module):
    def __init__(self, in_features, out_features, grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0,
                 scale_spline=1.0, enable_standalone_scale_spline=True, base_activation=torch.nn.SiLU, grid_eps=0.02,
                 grid_range=[-1, 1]):
        super().__init__()
        if base_activation not in _BASE_ACT_CODES:
            raise NotImplementedError(f"base activation {base_activation!r}: the kernels implement "
                                      f"{[c.__name__ for c in _BASE_ACT_CODES]}")
        self._base_act = _BASE_ACT_CODES[base_activation]
        self.in_features, self.out_features = in_features, out_features
        self.grid_size, self.spline_order = grid_size, spline_order
        h = (grid_range[1] - grid_range[0]) / grid_size
        knots = torch.arange(-spline_order, grid_size + spline_order + 1) * h + grid_range[0]
        self.register_buffer("grid", knots.expand(in_features, -1).contiguous())
        self.base_weight = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.spline_weight = torch.nn.Parameter(torch.empty(out_features, in_features, grid_size + spline_order))
        if enable_standalone_scale_spline:
            self.spline_scaler = torch.nn.Parameter(torch.empty(out_features, in_features))
        self.scale_noise, self.scale_base, self.scale_spline = scale_noise, scale_base, scale_spline
        self.enable_standalone_scale_spline = enable_standalone_scale_spline
        self.base_activation = base_activation()
        self.grid_eps = grid_eps
        self.reset_parameters()

    # -- initialisation: host-side (runs once, on the CPU, before .to(device)), follows kan1.py:55-75 --------------
    def _b_splines_host(self, x):
        g = self.grid
        x = x.unsqueeze(-1)
        bases = ((x >= g[:, :-1]) & (x < g[:, 1:])).to(x.dtype)
        for k in range(1, self.spline_order + 1):
            left = (x - g[:, :-(k + 1)]) / (g[:, k:-1] - g[:, :-(k + 1)]) * bases[:, :, :-1]
            right = (g[:, k + 1:] - x) / (g[:, k + 1:] - g[:, 1:(-k)]) * bases[:, :, 1:]
            bases = left + right
        return bases.contiguous()

    def curve2coeff(self, x, y):
        A = self._b_splines_host(x).transpose(0, 1)
        sol = torch.linalg.lstsq(A, y.transpose(0, 1)).solution
        return sol.permute(2, 0, 1).contiguous()

    def reset_parameters(self):
        torch.nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5) * self.scale_base)
        with torch.no_grad():
            noise = (torch.rand(self.grid_size + 1, self.in_features, self.out_features) - 0.5) * self.scale_noise / self.grid_size
            coeff = self.curve2coeff(self.grid.T[self.spline_order:-self.spline_order], noise)
            self.spline_weight.data.copy_((self.scale_spline if not self.enable_standalone_scale_spline else 1.0) * coeff)
            if self.enable_standalone_scale_spline:
                torch.nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5) * self.scale_spline)

    @property
    def scaled_spline_weight(self):
        return self.spline_weight * (self.spline_scaler.unsqueeze(-1) if self.enable_standalone_scale_spline else 1.0)

    def forward(self, x):
        assert x.size(-1) == self.in_features
        shape = x.shape
        flat = x.reshape(-1, self.in_features)
        if flat.dtype != torch.float32:
            flat = flat.float()
        scaler = self.spline_scaler if self.enable_standalone_scale_spline else None
        out = K.kan_linear(flat, self.grid, self.base_weight, self.spline_weight, scaler, self.grid_size, self.spline_order,
                           self._base_act)
        return out.reshape(*shape[:-1], self.out_features)

    @torch.no_grad()
    def update_grid(self, x, margin=0.01):
        """Data-dependent re-gridding (reference kan1.py:167-212): the knots move to the batch's per-feature quantiles
        (blended with a uniform grid by grid_eps) and the spline coefficients are re-fitted by least squares so that the
        spline part reproduces its old outputs on x.  A maintenance step between batches, not part of the per-batch path:
        like reset_parameters it runs on the host in f32 (sort + a batched least-squares solve of `in_features` small systems)
        and writes the results back into the device-resident buffers the kernels read."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        dev = self.grid.device
        xh = x.detach().float().cpu()
        batch = xh.size(0)
        host = types.SimpleNamespace(grid=self.grid.detach().float().cpu(), spline_order=self.spline_order)   # for _b_splines_host
        splines = KANLinear._b_splines_host(host, xh).permute(1, 0, 2)                      # (in, batch, coeff)
        coeff = self.scaled_spline_weight.detach().float().cpu().permute(1, 2, 0)           # (in, coeff, out)
        y = torch.bmm(splines, coeff).permute(1, 0, 2)                                      # (batch, in, out)
        xs = torch.sort(xh, dim=0)[0]
        adaptive = xs[torch.linspace(0, batch - 1, self.grid_size + 1, dtype=torch.int64)]
        step = (xs[-1] - xs[0] + 2 * margin) / self.grid_size
        uniform = torch.arange(self.grid_size + 1, dtype=torch.float32).unsqueeze(1) * step + xs[0] - margin
        grid = self.grid_eps * uniform + (1 - self.grid_eps) * adaptive
        k = self.spline_order
        grid = torch.cat([grid[:1] - step * torch.arange(k, 0, -1).unsqueeze(1), grid,
                          grid[-1:] + step * torch.arange(1, k + 1).unsqueeze(1)], dim=0)
        host.grid = grid.T.contiguous()
        A = KANLinear._b_splines_host(host, xh).transpose(0, 1)
        sol = torch.linalg.lstsq(A, y.transpose(0, 1)).solution
        self.grid.copy_(host.grid.to(dev))
        self.spline_weight.data.copy_(sol.permute(2, 0, 1).contiguous().to(dev))

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        return K.kan_regularization(self.spline_weight, regularize_activation, regularize_entropy)


class KAN1(torch.nn.Module):
    def __init__(self, layers_hidden=[768, 512, 256], grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0,
                 scale_spline=1.0, base_activation=torch.nn.SiLU, grid_eps=0.02, grid_range=[-1, 1]):
        super().__init__()
        self.grid_size, self.spline_order = grid_size, spline_order
        self.output_dim = layers_hidden[-1]
        self.layers = torch.nn.ModuleList(
            KANLinear(i, o, grid_size=grid_size, spline_order=spline_order, scale_noise=scale_noise, scale_base=scale_base,
                      scale_spline=scale_spline, base_activation=base_activation, grid_eps=grid_eps, grid_range=grid_range)
            for i, o in zip(layers_hidden, layers_hidden[1:]))

    def forward(self, x, update_grid=False):
        if x.numel() == 0:
            # an expert that received no rows (reference: the empty batch flows through every op and its parameters get
            # zero gradients, not None): an empty output that still hangs on the parameters
            y = torch.zeros((*x.shape[:-1], self.output_dim), device=x.device)
            if torch.is_grad_enabled():
                y = y + sum((p.reshape(-1)[:1].sum() * 0.0 for p in self.parameters() if p.requires_grad), torch.zeros((), device=x.device))
            return y
        for layer in self.layers:
            if update_grid:
                layer.update_grid(x)
            x = layer(x)
        return x

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        return sum(l.regularization_loss(regularize_activation, regularize_entropy) for l in self.layers)
