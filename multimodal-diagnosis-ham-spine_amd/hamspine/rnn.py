"""Recurrent cells of the slice-sequence encoder (reference modules/sequence_blocks.py -> torch.nn.LSTM / nn.GRU).
The matmuls run through hamspine.functional.linear (x W_ih^T + b_ih once for all steps, h W_hh^T + b_hh per step);
the gate arithmetic of one step is one HIP kernel per direction (hs_lstm_cell_* / hs_gru_cell_*).  f32 throughout: the
sequence encoder sits behind the pooled (f32) image features."""
import torch
from torch.autograd import Function

from . import _lib as L
from . import rt
from . import functional as F
from . import small as S


def _pitch(t):
    """row pitch (elements) of a (B, K) tensor whose rows are contiguous"""
    if t.stride(1) != 1:
        raise L.HamspineError("rnn cell inputs need contiguous rows")
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def _f32(t):
    rt.need_gpu(t)
    if t.dtype != torch.float32:
        raise L.HamspineError(f"rnn cells run in f32, got {t.dtype}")
    return t


class LSTMCellFn(Function):
    @staticmethod
    def forward(ctx, gx, gh, c_prev):
        gx, gh, c_prev = _f32(gx), _f32(gh), _f32(c_prev).contiguous()
        B, H = c_prev.shape
        h = torch.empty_like(c_prev)
        c = torch.empty_like(c_prev)
        act = torch.empty((B, 4 * H), dtype=torch.float32, device=c_prev.device)
        L.check(L.lib().hs_lstm_cell_fwd(rt.p(gx), _pitch(gx), rt.p(gh), _pitch(gh), rt.p(c_prev), rt.p(h), rt.p(c),
                                         rt.p(act), B, H, rt.stream()), "hs_lstm_cell_fwd")
        ctx.save_for_backward(act, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        act, c_prev, c = ctx.saved_tensors
        B, H = c_prev.shape
        dh = dh.contiguous() if dh is not None else None
        dc = dc.contiguous() if dc is not None else None
        dg = torch.empty_like(act)
        dcp = torch.empty_like(c_prev)
        L.check(L.lib().hs_lstm_cell_bwd(rt.p(dh), rt.p(dc), rt.p(act), rt.p(c_prev), rt.p(c), rt.p(dg), rt.p(dcp), B, H,
                                         rt.stream()), "hs_lstm_cell_bwd")
        return dg, dg, dcp


class GRUCellFn(Function):
    @staticmethod
    def forward(ctx, gx, gh, h_prev):
        gx, gh, h_prev = _f32(gx), _f32(gh), _f32(h_prev).contiguous()
        B, H = h_prev.shape
        h = torch.empty_like(h_prev)
        act = torch.empty((B, 4 * H), dtype=torch.float32, device=h_prev.device)
        L.check(L.lib().hs_gru_cell_fwd(rt.p(gx), _pitch(gx), rt.p(gh), _pitch(gh), rt.p(h_prev), rt.p(h), rt.p(act), B, H,
                                        rt.stream()), "hs_gru_cell_fwd")
        ctx.save_for_backward(act, h_prev)
        return h

    @staticmethod
    def backward(ctx, dh):
        act, h_prev = ctx.saved_tensors
        B, H = h_prev.shape
        dh = dh.contiguous()
        dgx = torch.empty((B, 3 * H), dtype=torch.float32, device=dh.device)
        dgh = torch.empty_like(dgx)
        dhp = torch.empty_like(h_prev)
        L.check(L.lib().hs_gru_cell_bwd(rt.p(dh), rt.p(act), rt.p(h_prev), rt.p(dgx), rt.p(dgh), rt.p(dhp), B, H,
                                        rt.stream()), "hs_gru_cell_bwd")
        return dgx, dgh, dhp


def run_direction(kind, x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x (B, T, D) f32 -> list of T hidden states (B, H), indexed by time step"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih)                                  # (B, T, gates*H), all steps in one GEMM
    h = torch.zeros((B, H), dtype=torch.float32, device=x.device)
    c = torch.zeros_like(h) if kind == "lstm" else None
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        gh = F.linear(h, w_hh, b_hh)
        if kind == "lstm":
            h, c = LSTMCellFn.apply(gx[:, t], gh, c)
        else:
            h = GRUCellFn.apply(gx[:, t], gh, h)
        outs[t] = h
    return outs


def stack_steps(fwd, bwd):
    """per-step (B, H) states of one or two directions -> (B, T, H or 2H)"""
    steps = [S.concat2(f, b) for f, b in zip(fwd, bwd)] if bwd is not None else fwd
    return torch.stack(steps, dim=1)
