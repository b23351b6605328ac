"""Thin torch-tensor wrappers over the C ABI (include/hamspine.h).  torch is plumbing here: device
memory (tensors), the current HIP stream, nothing else.  No function in this file computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib as L


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def hs_dtype(t):
    if t.dtype == torch.bfloat16:
        return L.HS_BF16
    if t.dtype == torch.float32:
        return L.HS_F32
    raise L.HamspineError(f"unsupported dtype {t.dtype}")


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.HamspineError(
                "hamspine kernels only run on the HIP device (tensor on %s); there is no CPU fallback" % t.device)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def conv_geom(N, H, W, Cin, Kout, R, S, stride, pad, row_pitch=None, img_pitch=None, qstep=None,
              no_bounds=0, P=None, Q=None):
    g = L.ConvGeom()
    g.N, g.H, g.W, g.C = N, H, W, Cin
    g.P = (H + 2 * pad - R) // stride + 1 if P is None else P
    g.Q = (W + 2 * pad - S) // stride + 1 if Q is None else Q
    g.K, g.R, g.S, g.stride, g.pad = Kout, R, S, stride, pad
    g.row_pitch = W * Cin if row_pitch is None else row_pitch
    g.img_pitch = H * g.row_pitch if img_pitch is None else img_pitch
    g.qstep = stride * Cin if qstep is None else qstep
    g.no_bounds = no_bounds
    return g


def gemm(A, B, D, M, N, K, a_kind=L.A_KC, b_kind=L.B_KC, lda=0, ldb=0, ldd=None, geom=None,
         batch=1, batch_inner=1, a_bs=(0, 0), b_bs=(0, 0), d_bs=(0, 0), split_k=1,
         alpha=1.0, bias=None, act=L.ACT_NONE, preact=None, residual=None, ldr=None,
         dropout_p=0.0, dropout_seed=0, mul_mode=L.MUL_NONE, mul_src=None, ldm=None, accumulate=False, rowsum_a=None, bnb=None,
         bn_finish=None, bnb_finish=None):
    """D[m][n] = epi(alpha * sum_k A(m,k) B(n,k)); see include/hamspine.h for the operand kinds.
    bn_finish = dict(gamma, beta, running_mean, running_var, eps, momentum): the launch also produces the train-mode
    BatchNorm statistics of the result (hs_gemm_params.colstats + .bn_finish); returns (D, mean, invstd, scale, shift).
    bnb = (c, scale, shift, mean, invstd): also return the BatchNorm-backward partial sums of the result
    (hs_gemm_params.bnb_*) as a float tensor [tile rows][N][2].
    bnb_finish = dict(gamma, training) (with bnb): the launch also finishes those sums (hs_gemm_params.bnb_finish); returns
    (D, partial rows, dgamma, dbeta, coef [4][N])."""
    need_gpu(A, B, D, bias, preact, residual, mul_src, rowsum_a)
    p = L.GemmParams()
    p.dtype = hs_dtype(A)
    if hs_dtype(B) != p.dtype:
        raise L.HamspineError("gemm: A and B dtypes differ")
    p.a_kind, p.b_kind = a_kind, b_kind
    p.M, p.N, p.K = M, N, K
    p.A, p.B = ptr(A), ptr(B)
    p.a_elems, p.b_elems = A.numel(), B.numel()
    p.lda, p.ldb = lda, ldb
    if geom is not None:
        p.g = geom
    p.batch, p.batch_inner = batch, batch_inner
    p.a_bs0, p.a_bs1 = a_bs
    p.b_bs0, p.b_bs1 = b_bs
    p.d_bs0, p.d_bs1 = d_bs
    p.split_k = split_k
    ws = None
    if split_k > 1:
        ws = torch.empty(int(L.lib().hs_gemm_splitk_ws_bytes(C.byref(p))) // 4, dtype=torch.float32, device=A.device)
        p.splitk_ws = ptr(ws)
    p.D = ptr(D)
    p.ldd = N if ldd is None else ldd
    p.out_dtype = hs_dtype(D)
    p.alpha = alpha
    if bias is not None and bias.dtype != torch.float32:
        raise L.HamspineError("gemm: bias must be f32")
    p.bias = ptr(bias)
    p.act = act
    p.D_preact = ptr(preact)
    p.residual = ptr(residual)
    p.ldr = p.ldd if ldr is None else ldr
    p.dropout_p = dropout_p
    p.dropout_seed = dropout_seed
    p.mul_mode = mul_mode
    p.mul_src = ptr(mul_src)
    p.ldm = p.ldd if ldm is None else ldm
    p.accumulate = 1 if accumulate else 0
    p.rowsum_a = ptr(rowsum_a)
    partials = None
    if bnb is not None:
        c, sc, sh, mu, inv = bnb[:5]
        need_gpu(c, sc, sh, mu, inv)
        p.bnb_x, p.bnb_scale, p.bnb_shift, p.bnb_mean, p.bnb_invstd = ptr(c), ptr(sc), ptr(sh), ptr(mu), ptr(inv)
        if len(bnb) > 5:                 # block-output form: mask from the saved output y (hs_gemm_params.bnb_y)
            need_gpu(bnb[5])
            p.bnb_y = ptr(bnb[5])
        p.bnb_partials = 16              # any non-null value: hs_gemm_tile_rows only looks at the configuration
        rows = int(L.lib().hs_gemm_tile_rows(C.byref(p)))
        if rows <= 0:
            raise L.HamspineError("gemm: bnb is not available for this configuration")
        fin = None
        if bnb_finish is not None:
            frows = int(L.lib().hs_gemm_bnb_finish_rows(C.byref(p)))
            if frows <= 0:
                raise L.HamspineError("gemm: bnb_finish is not available for this configuration")
            buf = torch.full((frows * N * 2 + 4 * N,), float("nan"), dtype=torch.float32, device=A.device)
            partials = buf[:rows * N * 2].view(rows, N, 2)
            dgamma, dbeta = (torch.full((N,), float("nan"), dtype=torch.float32, device=A.device) for _ in range(2))
            need_gpu(bnb_finish.get("gamma"))
            fq = L.BnBwdParams()
            fq.dtype, fq.C, fq.M, fq.training = p.dtype, N, M, int(bnb_finish.get("training", 1))
            fq.gamma, fq.dgamma, fq.dbeta = ptr(bnb_finish.get("gamma")), ptr(dgamma), ptr(dbeta)
            p.bnb_partials = ptr(buf)
            p.bnb_finish = C.addressof(fq)
            fin = (fq, dgamma, dbeta, buf[frows * N * 2:].view(4, N))
        else:
            partials = torch.empty((rows, N, 2), dtype=torch.float32, device=A.device)
            p.bnb_partials = ptr(partials)
    if bn_finish is not None:
        f = bn_finish
        p.colstats = 16                  # any non-null value: the row query only looks at the configuration
        rows = int(L.lib().hs_gemm_bn_finish_rows(C.byref(p)))
        if rows <= 0:
            raise L.HamspineError("gemm: bn_finish is not available for this configuration")
        stats_ws = torch.empty((rows, N, 3), dtype=torch.float32, device=A.device)
        out = [torch.empty(N, dtype=torch.float32, device=A.device) for _ in range(4)]
        need_gpu(f.get("gamma"), f.get("beta"), f.get("running_mean"), f.get("running_var"))
        bp = L.BnParams()
        bp.dtype, bp.C, bp.M, bp.training = p.dtype, N, M, 1
        bp.eps, bp.momentum = f.get("eps", 1e-5), f.get("momentum", 0.1)
        bp.gamma, bp.beta = ptr(f.get("gamma")), ptr(f.get("beta"))
        bp.running_mean, bp.running_var = ptr(f.get("running_mean")), ptr(f.get("running_var"))
        bp.save_mean, bp.save_invstd, bp.scale, bp.shift = (ptr(t) for t in out)
        bp.ws, bp.ws_bytes = ptr(stats_ws), stats_ws.numel() * 4
        p.colstats = ptr(stats_ws)
        p.bn_finish = C.addressof(bp)
        L.check(L.lib().hs_gemm(C.byref(p), stream_ptr()), "hs_gemm")
        return (D, *out)
    L.check(L.lib().hs_gemm(C.byref(p), stream_ptr()), "hs_gemm")
    if bnb is not None and bnb_finish is not None:
        return (D, partials, fin[1], fin[2], fin[3])
    return (D, partials) if bnb is not None else D


def suggest_split(M, N, K, dtype):
    return int(L.lib().hs_gemm_suggest_split(M, N, K, dtype))


def pointwise_fwd(x, w, want_stats=True):
    """streaming 1x1 convolution (hs_pointwise_fwd): x [M][K] bf16, w [N][K] bf16 -> (y [M][N] bf16, stats [(rows)][N][3] or None)"""
    need_gpu(x, w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    rows = int(L.lib().hs_pointwise_stat_rows(M, N, K)) if want_stats else 0
    stats = torch.full((max(rows, 1), N, 3), float("nan"), dtype=torch.float32, device=x.device) if want_stats else None
    L.check(L.lib().hs_pointwise_fwd(ptr(x), M, K, x.stride(0), ptr(w), N, ptr(y), N, ptr(stats), stream_ptr()), "hs_pointwise_fwd")
    return y, (stats[:rows] if want_stats else None)
