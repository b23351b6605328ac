"""Parity of the MFMA GEMM / implicit-GEMM conv core (C ABI hs_gemm) against float64 torch on the CPU.

bf16 cases are fed integer-valued data where exactness is checked (catches layout / transpose bugs
bit-exactly) and random data with a tolerance that is stated next to each assert."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import hamspine._lib as L  # noqa: E402
from hamspine import raw  # noqa: E402

DEV = "cuda"


def _rand(shape, dtype, ints=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    if ints:
        x = torch.randint(-3, 4, shape, generator=g).float()
    else:
        x = torch.randn(shape, generator=g)
    return x.to(dtype)


def _tol(dtype, K):
    # f32: exact fmaf chain, only summation order differs from the CPU -> 1e-5 relative to |a|.|b|
    # bf16: inputs are rounded to bf16 on both sides (the reference uses the rounded values), f32
    # accumulation -> same bound; outputs stored as bf16 add 2^-8 relative.
    return 2e-5 if dtype == torch.float32 else 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("combo", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 136, 72), (4096, 768, 768), (64, 8, 256), (33, 48, 40)])
def test_gemm_layouts(dtype, combo, shape):
    M, N, K = shape
    for ints in (True, False):
        if combo == "nt":
            A = _rand((M, K), dtype, ints, 1)
            B = _rand((N, K), dtype, ints, 2)
            ref = A.double() @ B.double().t()
            kinds = (L.A_KC, L.B_KC, K, K)
        elif combo == "nn":
            A = _rand((M, K), dtype, ints, 1)
            B = _rand((K, N), dtype, ints, 2)
            ref = A.double() @ B.double()
            kinds = (L.A_KC, L.B_RC, K, N)
        else:
            A = _rand((K, M), dtype, ints, 1)
            B = _rand((K, N), dtype, ints, 2)
            ref = A.double().t() @ B.double()
            kinds = (L.A_RC, L.B_RC, M, N)
        if dtype == torch.bfloat16 and (kinds[2] % 8 or kinds[3] % 8):
            pytest.skip("bf16 path requires 16-byte aligned rows")
        D = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        raw.gemm(A.to(DEV), B.to(DEV), D, M, N, K, a_kind=kinds[0], b_kind=kinds[1], lda=kinds[2], ldb=kinds[3])
        out = D.cpu().double()
        if ints:
            assert torch.equal(out, ref), f"{combo} {shape} {dtype}: integer data must be exact"
        else:
            scale = (A.double().abs().max() * B.double().abs().max() * K)
            assert (out - ref).abs().max() <= _tol(dtype, K) * scale


@pytest.mark.parametrize("cfg", [0, 1, 2, 4, 5, 6])   # 128x128, 128x64, 64x64, 256x128 (8 waves), 128x128x32, 256x128x32
@pytest.mark.parametrize("combo", ["nt", "nn", "tn"])
def test_gemm_every_tile_config_exact(cfg, combo):
    """each bf16 tile configuration forced on ragged and multi-tile shapes: integer data must come out exact"""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    try:
        lib.hs_gemm_debug(cfg, 0)
        for M, N, K in ((520, 264, 200), (256, 128, 64), (1000, 136, 1032), (72, 8, 40)):
            if combo == "nt":
                A, B = _rand((M, K), torch.bfloat16, True, 3), _rand((N, K), torch.bfloat16, True, 4)
                ref, kinds = A.double() @ B.double().t(), (L.A_KC, L.B_KC, K, K)
            elif combo == "nn":
                A, B = _rand((M, K), torch.bfloat16, True, 3), _rand((K, N), torch.bfloat16, True, 4)
                ref, kinds = A.double() @ B.double(), (L.A_KC, L.B_RC, K, N)
            else:
                A, B = _rand((K, M), torch.bfloat16, True, 3), _rand((K, N), torch.bfloat16, True, 4)
                ref, kinds = A.double().t() @ B.double(), (L.A_RC, L.B_RC, M, N)
            D = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
            bias = torch.arange(N, dtype=torch.float32, device=DEV)
            raw.gemm(A.to(DEV), B.to(DEV), D, M, N, K, a_kind=kinds[0], b_kind=kinds[1], lda=kinds[2], ldb=kinds[3], bias=bias)
            assert torch.equal(D.cpu().double(), ref + bias.cpu().double()), f"cfg {cfg} {combo} {(M, N, K)}"
    finally:
        lib.hs_gemm_debug(-1, 0)


@pytest.mark.parametrize("cfg", [-1, 1, 2])
def test_wgrad_gemm_row_sums_are_the_bias_gradient(cfg):
    """rowsum_a: the weight-gradient GEMM dY^T X also returns colsum(dY) (one extra MFMA per A fragment in the first tile
    column).  Integer data: exact, on ragged sizes, several tile configurations and a strided dY (ld > out features)."""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    try:
        lib.hs_gemm_debug(cfg, 0)
        for rows, out_f, in_f, ld in ((300, 72, 40, 72), (4096, 768, 768, 768), (130, 200, 264, 208), (64, 8, 8, 24)):
            dY = _rand((rows, ld), torch.bfloat16, True, 5)
            X = _rand((rows, in_f), torch.bfloat16, True, 6)
            dW = torch.full((out_f, in_f), float("nan"), dtype=torch.float32, device=DEV)
            db = torch.full((out_f,), float("nan"), dtype=torch.float32, device=DEV)
            raw.gemm(dY.to(DEV), X.to(DEV), dW, out_f, in_f, rows, a_kind=L.A_RC, b_kind=L.B_RC, lda=ld, ldb=in_f, rowsum_a=db)
            assert torch.equal(dW.cpu().double(), dY[:, :out_f].double().t() @ X.double()), (cfg, rows, out_f, in_f)
            assert torch.equal(db.cpu().double(), dY[:, :out_f].double().sum(0)), (cfg, rows, out_f, in_f)
    finally:
        lib.hs_gemm_debug(-1, 0)
    # K-contiguous operands (dY^T, X^T: the transposed-operand weight gradient of the BERT layers) produce the row sums too
    for rows, out_f, in_f in ((320, 72, 40), (4096, 768, 768), (136, 200, 264)):
        dYt = _rand((out_f, rows), torch.bfloat16, True, 7)
        Xt = _rand((in_f, rows), torch.bfloat16, True, 8)
        dW = torch.full((out_f, in_f), float("nan"), dtype=torch.float32, device=DEV)
        db = torch.full((out_f,), float("nan"), dtype=torch.float32, device=DEV)
        raw.gemm(dYt.to(DEV), Xt.to(DEV), dW, out_f, in_f, rows, a_kind=L.A_KC, b_kind=L.B_KC, lda=rows, ldb=rows, rowsum_a=db)
        assert torch.equal(dW.cpu().double(), dYt.double() @ Xt.double().t()), (rows, out_f, in_f)
        assert torch.equal(db.cpu().double(), dYt.double().sum(1)), (rows, out_f, in_f)
    with pytest.raises(L.HamspineError, match="rowsum_a"):     # mixed layouts have no row-sum variant
        A = _rand((64, 64), torch.bfloat16, True, 1).to(DEV)
        raw.gemm(A, A, torch.empty(64, 64, device=DEV), 64, 64, 64, a_kind=L.A_KC, b_kind=L.B_RC, lda=64, ldb=64,
                 rowsum_a=torch.empty(64, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_and_splitk(dtype):
    M, N, K = 256, 192, 512
    A = _rand((M, K), dtype, False, 3)
    B = _rand((N, K), dtype, False, 4)
    bias = _rand((N,), torch.float32, False, 5)
    res = _rand((M, N), dtype, False, 6)
    pre_ref = 0.5 * (A.double() @ B.double().t()) + bias.double()
    ref = torch.nn.functional.gelu(pre_ref) + res.double()
    D = torch.empty((M, N), dtype=dtype, device=DEV)
    P = torch.empty((M, N), dtype=dtype, device=DEV)
    raw.gemm(A.to(DEV), B.to(DEV), D, M, N, K, lda=K, ldb=K, alpha=0.5, bias=bias.to(DEV), act=L.ACT_GELU,
             preact=P, residual=res.to(DEV))
    rel = 1e-5 if dtype == torch.float32 else 2.0 ** -7   # bf16 storage rounding of output
    assert (D.cpu().double() - ref).abs().max() <= rel * ref.abs().max() + 1e-5
    assert (P.cpu().double() - pre_ref).abs().max() <= rel * pre_ref.abs().max() + 1e-5
    # split-K (deterministic slabs) == single pass up to f32 summation order
    D1 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    D4 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    raw.gemm(A.to(DEV), B.to(DEV), D1, M, N, K, lda=K, ldb=K, bias=bias.to(DEV))
    raw.gemm(A.to(DEV), B.to(DEV), D4, M, N, K, lda=K, ldb=K, bias=bias.to(DEV), split_k=4)
    assert (D1 - D4).abs().max().item() <= 1e-4 * D1.abs().max().item()
    # relu mask + gelu grad multipliers
    u = _rand((M, N), dtype, False, 7)
    Dm = torch.empty((M, N), dtype=torch.float32, device=DEV)
    raw.gemm(A.to(DEV), B.to(DEV), Dm, M, N, K, lda=K, ldb=K, mul_mode=L.MUL_RELU_MASK, mul_src=u.to(DEV))
    ref_m = (A.double() @ B.double().t()) * (u.double() > 0)
    assert (Dm.cpu().double() - ref_m).abs().max() <= 2e-5 * K * 16
    raw.gemm(A.to(DEV), B.to(DEV), Dm, M, N, K, lda=K, ldb=K, mul_mode=L.MUL_GELU_GRAD, mul_src=u.to(DEV))
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    ref_g = (A.double() @ B.double().t()) * ud.grad
    assert (Dm.cpu().double() - ref_g).abs().max() <= 1e-4 * ref_g.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_batched_and_dropout(dtype):
    Bt, H, Lq, Lk, d = 3, 4, 49, 128, 32
    q = _rand((Bt, Lq, H * d), dtype, False, 1)
    k = _rand((Bt, Lk, H * d), dtype, False, 2)
    S = torch.empty((Bt, H, Lq, Lk), dtype=torch.float32, device=DEV)
    raw.gemm(q.to(DEV), k.to(DEV), S, Lq, Lk, d, lda=H * d, ldb=H * d, ldd=Lk, batch=Bt * H, batch_inner=H,
             a_bs=(Lq * H * d, d), b_bs=(Lk * H * d, d), d_bs=(H * Lq * Lk, Lq * Lk), alpha=0.25)
    ref = 0.25 * torch.einsum("bqhd,bkhd->bhqk", q.double().view(Bt, Lq, H, d), k.double().view(Bt, Lk, H, d))
    assert (S.cpu().double() - ref).abs().max() <= 1e-4 * ref.abs().max()
    # dropout: deterministic in (seed, index); keep-rate ~ 1-p; kept values scaled by 1/(1-p)
    M, N, K = 512, 256, 64
    A = _rand((M, K), dtype, False, 3)
    Bm = _rand((N, K), dtype, False, 4)
    D0 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    D1 = torch.empty_like(D0)
    D2 = torch.empty_like(D0)
    raw.gemm(A.to(DEV), Bm.to(DEV), D0, M, N, K, lda=K, ldb=K)
    raw.gemm(A.to(DEV), Bm.to(DEV), D1, M, N, K, lda=K, ldb=K, dropout_p=0.1, dropout_seed=1234)
    raw.gemm(A.to(DEV), Bm.to(DEV), D2, M, N, K, lda=K, ldb=K, dropout_p=0.1, dropout_seed=1234)
    assert torch.equal(D1, D2)
    kept = D1 != 0
    frac = kept.float().mean().item()
    assert abs(frac - 0.9) < 0.01
    assert torch.allclose(D1[kept], D0[kept] / 0.9, rtol=1e-6, atol=1e-6)


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad
    (2, 64, 14, 14, 64, 3, 1, 1),
    (2, 64, 15, 13, 128, 3, 2, 1),
    (3, 128, 8, 8, 256, 1, 1, 0),
    (2, 256, 14, 14, 512, 1, 2, 0),
    (4, 64, 56, 56, 64, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dtype, case):
    N, Cin, H, W, K, R, stride, pad = case
    x = _rand((N, Cin, H, W), dtype, False, 1)
    w = (_rand((K, Cin, R, R), dtype, False, 2) * 0.1).to(dtype)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    yref = torch.nn.functional.conv2d(xd, wd, stride=stride, padding=pad)
    P, Q = yref.shape[2:]
    dy = _rand(tuple(yref.shape), dtype, False, 3)
    yref.backward(dy.double())
    g = raw.conv_geom(N, H, W, Cin, K, R, R, stride, pad)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wg = w.to(DEV).contiguous(memory_format=torch.channels_last)
    dyg = dy.to(DEV).contiguous(memory_format=torch.channels_last)
    RSC = R * R * Cin
    tol = 2e-5 if dtype == torch.float32 else 2e-5
    # forward
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=DEV).contiguous(memory_format=torch.channels_last)
    raw.gemm(xg, wg, y, N * P * Q, K, RSC, a_kind=L.A_CONV, b_kind=L.B_KC, ldb=RSC, ldd=K, geom=g)
    assert (y.cpu().double() - yref.detach()).abs().max() <= tol * RSC * x.abs().max().item() * w.abs().max().item()
    # dgrad
    dx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=DEV).contiguous(memory_format=torch.channels_last)
    raw.gemm(dyg, wg, dx, N * H * W, Cin, R * R * K, a_kind=L.A_DGRAD, b_kind=L.B_WDGRAD, ldd=Cin, geom=g)
    assert (dx.cpu().double() - xd.grad).abs().max() <= tol * R * R * K * dy.abs().max().item() * w.abs().max().item()
    # wgrad (with and without split-K)
    for split in (1, 3):
        dw = torch.empty((K, Cin, R, R), dtype=torch.float32, device=DEV).contiguous(memory_format=torch.channels_last)
        raw.gemm(dyg, xg, dw, K, RSC, N * P * Q, a_kind=L.A_RC, b_kind=L.B_CONV, lda=K, ldd=RSC, geom=g,
                 split_k=split)
        assert (dw.cpu().double() - wd.grad).abs().max() <= tol * N * P * Q * dy.abs().max().item() * x.abs().max().item()


@pytest.mark.parametrize("kind,M,N,K,split", [("tn", 768, 768, 4096, 4), ("tn", 256, 64, 12800, 16), ("nt", 512, 768, 3072, 3),
                                               ("nn", 320, 200, 2048, 2), ("tn", 104, 72, 5000, 5),
                                               # two-level hand-off: groups of 8 slices (full, ragged last group, 128x64 tiles)
                                               ("tn", 256, 64, 100352, 64), ("tn", 200, 72, 20000, 13), ("nt", 256, 1024, 8192, 20),
                                               ("nn", 1568, 512, 2048, 9)])
def test_splitk_in_launch_reduction(kind, M, N, K, split):
    """bf16 split-K reduces inside the launch (slices hand their slabs off in groups of 8: the last slice of a group sums the
    group in slice order, the last group of a tile sums the group slabs and runs the epilogue; no second pass): equal to the
    unsplit GEMM up to f32 summation order, bit-identical from launch to launch
    (the sum order does not depend on which slice arrives last), repeatable back to back on one stream (the arrival
    counters re-zero themselves), and on two streams at once (each stream has its own counters)."""
    g = torch.Generator().manual_seed(M + N + K)
    BF = torch.bfloat16
    if kind == "nt":
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_KC, lda=K, ldb=K)
        ref = A.to(BF).float() @ B.to(BF).float().T
    elif kind == "nn":
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N)
        ref = A.to(BF).float() @ B.to(BF).float()
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        kw = dict(a_kind=L.A_RC, b_kind=L.B_RC, lda=M, ldb=N)
        ref = A.to(BF).float().T @ B.to(BF).float()
    bias = torch.randn(N, generator=g)
    Ad, Bd, bd = A.to(DEV).to(BF), B.to(DEV).to(BF), bias.to(DEV)
    outs = []
    for _ in range(3):
        D = torch.full((M, N), float("nan"), device=DEV)
        raw.gemm(Ad, Bd, D, M, N, K, bias=bd, split_k=split, **kw)
        outs.append(D.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = ref + bias
    err = (outs[0] - want).abs().max().item()
    assert err <= 2e-3 * want.abs().max().item(), err
    D1 = torch.empty((M, N), device=DEV)
    raw.gemm(Ad, Bd, D1, M, N, K, bias=bd, **kw)
    assert (D1.cpu() - outs[0]).abs().max().item() <= 1e-4 * want.abs().max().item()
    # two streams at once
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    Da, Db = torch.empty((M, N), device=DEV), torch.empty((M, N), device=DEV)
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            raw.gemm(Ad, Bd, Da, M, N, K, bias=bd, split_k=split, **kw)
        with torch.cuda.stream(s2):
            raw.gemm(Ad, Bd, Db, M, N, K, bias=bd, split_k=split, **kw)
    torch.cuda.synchronize()
    assert torch.equal(Da.cpu(), outs[0]) and torch.equal(Db.cpu(), outs[0])


@pytest.mark.parametrize("kind,M,N,K", [("nt", 100352, 64, 64), ("nt", 16384, 512, 256), ("nn", 12800, 320, 192), ("nt", 4096, 3072, 768),
                                        ("nn", 100352, 256, 64)])
def test_persistent_launch_equals_one_tile_per_workgroup(kind, M, N, K):
    """The persistent variant of the GEMM kernel (a workgroup walks several tiles and prefetches the next tile's operands
    under the current epilogue; off by default -- measured neutral to slower -- on with HAMSPINE_PERSISTENT=1 or hs_gemm_debug
    ablation bit 64) is bit-identical to one workgroup per tile, with bias + GELU + pre-activation copy and with a residual,
    on ragged tile counts."""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    g = torch.Generator().manual_seed(M + N)
    BF = torch.bfloat16
    A = torch.randn(M, K, generator=g).to(BF).to(DEV)
    if kind == "nt":
        B = torch.randn(N, K, generator=g).to(BF).to(DEV)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_KC, lda=K, ldb=K)
        ref = A.float() @ B.float().T
    else:
        B = torch.randn(K, N, generator=g).to(BF).to(DEV)
        kw = dict(a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N)
        ref = A.float() @ B.float()
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(BF).to(DEV)
    outs = []
    for ablate in (0, 64):
        lib.hs_gemm_debug(-1, ablate)
        try:
            D1 = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
            P1 = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
            raw.gemm(A, B, D1, M, N, K, bias=bias, act=L.ACT_GELU, preact=P1, **kw)
            D2 = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
            raw.gemm(A, B, D2, M, N, K, residual=res, **kw)
            D3 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
            raw.gemm(A, B, D3, M, N, K, **kw)
        finally:
            lib.hs_gemm_debug(-1, 0)
        outs.append((D1.cpu(), P1.cpu(), D2.cpu(), D3.cpu()))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    assert (outs[0][3] - ref.cpu()).abs().max().item() <= 2e-5 * K * 16


@pytest.mark.parametrize("M,N,K,split", [(6272, 256, 1024, 1), (1568, 512, 2048, 3), (100352, 64, 256, 1), (1000, 72, 320, 1)])
def test_data_gradient_gemm_takes_the_batchnorm_backward_sums(M, N, K, split):
    """hs_gemm_params.bnb_*: a data-gradient GEMM (dx = dy W) whose result is the gradient of relu(bn(c)) also returns that
    BatchNorm's backward sums per row tile -- sum dz and sum dz * xhat with dz = dx * (fma(c, scale, shift) > 0) taken on the
    bf16-rounded dx the epilogue stores.  Checked against the same sums of the stored result in float64 (the order of the
    f32 additions differs: 1e-4 relative to the sum of magnitudes), with and without split-K, ragged tiles included."""
    g = torch.Generator().manual_seed(M + N)
    BF = torch.bfloat16
    dy = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(DEV)
    W = (torch.randn(K, N, generator=g) * 0.1).to(BF).to(DEV)
    c = torch.randn(M, N, generator=g).to(BF).to(DEV)
    gamma = torch.rand(N, generator=g) + 0.5
    beta = torch.randn(N, generator=g) * 0.3
    mean, invstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    scale = (gamma * invstd)
    shift = beta - mean * scale
    dev = lambda t: t.float().to(DEV)
    D = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    D, part = raw.gemm(dy, W, D, M, N, K, a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N, split_k=split,
                       bnb=(c, dev(scale), dev(shift), dev(mean), dev(invstd)))
    ref = (dy.float() @ W.float())
    assert (D.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6
    cd, dd = c.double().cpu(), D.double().cpu()
    mask = (torch.addcmul(dev(shift).cpu().float(), c.float().cpu(), dev(scale).cpu().float()) > 0).double()   # f32 fma, as the kernel
    dz = dd * mask
    xhat = (cd - mean.float().double()) * invstd.float().double()
    s = part.double().cpu().sum(0)
    want1, want2 = dz.sum(0), (dz * xhat).sum(0)
    tol1 = 1e-4 * dz.abs().sum(0) + 1e-6
    tol2 = 1e-4 * (dz * xhat).abs().sum(0) + 1e-6
    assert ((s[:, 0] - want1).abs() <= tol1).all()
    assert ((s[:, 1] - want2).abs() <= tol2).all()


@pytest.mark.parametrize("M,N,K,block_form", [(6272, 256, 1024, False), (100352, 64, 256, False), (1000, 72, 320, False), (6272, 1024, 256, True),
                                              (25088 + 8, 128, 512, True)])
def test_data_gradient_gemm_finishes_the_batchnorm_backward_sums(M, N, K, block_form):
    """hs_gemm_params.bnb_finish: the launch's last workgroups add the tile rows' sums and write dbeta, dgamma and the apply
    pass's coefficient vectors [g*is | -g*is^2*dgamma/M | -g*is*dbeta/M | mean] (what bn_bwd_final_kernel writes): one merge level,
    two levels (1568 tile rows), ragged tiles, both forms of the rider, twice (the arrival counters re-arm)"""
    g = torch.Generator().manual_seed(M + N + 11)
    BF = torch.bfloat16
    dy = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(DEV)
    W = (torch.randn(K, N, generator=g) * 0.1).to(BF).to(DEV)
    c = torch.randn(M, N, generator=g).to(BF).to(DEV)
    y = torch.relu(torch.randn(M, N, generator=g)).to(BF).to(DEV)
    gamma = torch.rand(N, generator=g) + 0.5
    beta = torch.randn(N, generator=g) * 0.3
    mean, invstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    scale = gamma * invstd
    shift = beta - mean * scale
    dev = lambda t: t.float().to(DEV)
    for rep in range(2):
        D = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
        bnb = (c, dev(scale), dev(shift), dev(mean), dev(invstd)) + ((y,) if block_form else ())
        D, part, dgamma, dbeta, coef = raw.gemm(dy, W, D, M, N, K, a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N, bnb=bnb,
                                                bnb_finish=dict(gamma=dev(gamma), training=1))
        dd, cd = D.double().cpu(), c.double().cpu()
        if block_form:
            mask = (y.double().cpu() > 0).double()
        else:
            mask = (torch.addcmul(dev(shift).cpu().float(), c.float().cpu(), dev(scale).cpu().float()) > 0).double()
        dz = dd * mask
        xhat = (cd - mean.float().double()) * invstd.float().double()
        want1, want2 = dz.sum(0), (dz * xhat).sum(0)
        tol1 = 1e-4 * dz.abs().sum(0) + 1e-6
        tol2 = 1e-4 * (dz * xhat).abs().sum(0) + 1e-6
        assert ((part.double().cpu().sum(0)[:, 0] - want1).abs() <= tol1).all(), rep
        assert ((dbeta.double().cpu() - want1).abs() <= tol1).all(), rep
        assert ((dgamma.double().cpu() - want2).abs() <= tol2).all(), rep
        gi = (gamma.float() * invstd.float()).double()
        co = coef.double().cpu()
        assert ((co[0] - gi).abs() <= 1e-6 * gi.abs()).all()
        cb_ref, cc_ref = -gi * invstd.float().double() * want2 / M, -gi * want1 / M
        assert ((co[1] - cb_ref).abs() <= 1e-4 * (gi * invstd.double() * (dz * xhat).abs().sum(0) / M) + 1e-9).all()
        assert ((co[2] - cc_ref).abs() <= 1e-4 * (gi * dz.abs().sum(0) / M) + 1e-9).all()
        assert torch.equal(co[3].float(), mean.float())


@pytest.mark.parametrize("M,N,K,with_res", [(6272, 1024, 256, True), (100352, 256, 64, True), (1000, 72 + 8, 320, True), (25088, 512, 128, False)])
def test_data_gradient_gemm_takes_the_block_final_batchnorm_sums(M, N, K, with_res):
    """block-output form of the rider (hs_gemm_params.bnb_y): dx = dy W + identity-path gradient is the gradient of
    relu(bn(c) + identity); the mask is y > 0 on the saved block output, the residual is added before the sums and before
    the (single) bf16 rounding of the stored result"""
    g = torch.Generator().manual_seed(M + N + 3)
    BF = torch.bfloat16
    dy = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(DEV)
    W = (torch.randn(K, N, generator=g) * 0.1).to(BF).to(DEV)
    c = torch.randn(M, N, generator=g).to(BF).to(DEV)
    y = torch.relu(torch.randn(M, N, generator=g)).to(BF).to(DEV)
    res = (torch.randn(M, N, generator=g) * 0.3).to(BF).to(DEV) if with_res else None
    mean, invstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    dev = lambda t: t.float().to(DEV)
    D = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    D, part = raw.gemm(dy, W, D, M, N, K, a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N, residual=res,
                       bnb=(c, dev(mean), dev(mean), dev(mean), dev(invstd), y))
    ref = dy.float() @ W.float() + (res.float() if with_res else 0.0)
    assert (D.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6
    # without the rider the same launch stores the same values (the residual is added exactly once either way)
    D0 = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    raw.gemm(dy, W, D0, M, N, K, a_kind=L.A_KC, b_kind=L.B_RC, lda=K, ldb=N, residual=res)
    assert torch.equal(D0, D)
    dd, cd = D.double().cpu(), c.double().cpu()
    dz = dd * (y.double().cpu() > 0)
    xhat = (cd - mean.float().double()) * invstd.float().double()
    s = part.double().cpu().sum(0)
    assert ((s[:, 0] - dz.sum(0)).abs() <= 1e-4 * dz.abs().sum(0) + 1e-6).all()
    assert ((s[:, 1] - (dz * xhat).sum(0)).abs() <= 1e-4 * (dz * xhat).abs().sum(0) + 1e-6).all()


@pytest.mark.parametrize("case", [
    # (M, N, K) of 1x1 convolutions (K-contiguous operands): one merge level, two levels (ragged last group), ragged tiles
    ("pw", 1568, 2048, 512), ("pw", 100352, 64, 64), ("pw", 25088, 512, 128), ("pw", 1000, 72, 320), ("pw", 6272 + 40, 256, 1024),
    # (N, Cin, H, W, Cout, R, stride, pad) of implicit-GEMM convolutions (the 7x7 stem's launch: tests/test_tower_gpu.py)
    ("conv", 32, 64, 56, 56, 64, 3, 1, 1), ("conv", 4, 128, 28, 28, 128, 3, 2, 1), ("conv", 2, 64, 14, 14, 512, 3, 1, 1)])
def test_forward_gemm_finishes_the_batchnorm_statistics(case):
    """hs_gemm_params.bn_finish: the forward (1x1 / implicit-GEMM) convolution also finishes the train-mode statistics of the
    BatchNorm that follows -- mean, 1/sqrt(var + eps), scale = gamma * invstd, shift = beta - mean * scale and the running
    statistics (biased variance for normalisation, unbiased for running_var, reference torch.nn.BatchNorm2d inside
    torchvision's Bottleneck) -- in the same launch.  Checked against float64 statistics of the f32 accumulators' product
    (the kernel takes them before the bf16 rounding of the stored result) and run twice: the arrival counters re-arm."""
    g = torch.Generator().manual_seed(len(case) * 7 + case[1])
    BF = torch.bfloat16
    if case[0] == "pw":
        _, M, N, K = case
        A = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(DEV)
        W = (torch.randn(N, K, generator=g) * 0.1).to(BF).to(DEV)
        ref = A.double() @ W.double().t()
        kw = dict(a_kind=L.A_KC, b_kind=L.B_KC, lda=K, ldb=K)
        args = (A, W)
    else:
        _, Nb, Cin, H, Wd, N, R, stride, pad = case
        x = (torch.randn(Nb, Cin, H, Wd, generator=g) * 0.5).to(BF)
        w = (torch.randn(N, Cin, R, R, generator=g) * 0.1).to(BF)
        ref4 = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride, padding=pad)
        P, Q = ref4.shape[2:]
        M, K = Nb * P * Q, R * R * Cin
        ref = ref4.permute(0, 2, 3, 1).reshape(M, N).to(DEV)
        geom = raw.conv_geom(Nb, H, Wd, Cin, N, R, R, stride, pad)
        args = (x.to(DEV).contiguous(memory_format=torch.channels_last), w.to(DEV).contiguous(memory_format=torch.channels_last))
        kw = dict(a_kind=L.A_CONV, b_kind=L.B_KC, ldb=K, geom=geom)
    gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(N, generator=g) * 0.3).to(DEV)
    rm0, rv0 = (torch.randn(N, generator=g) * 0.1).to(DEV), (torch.rand(N, generator=g) + 0.5).to(DEV)
    eps, mom = 1e-5, 0.1
    mean_ref = ref.mean(0)
    var_ref = ref.var(0, unbiased=False)
    for rep in range(2):
        rm, rv = rm0.clone(), rv0.clone()
        D = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
        D, mean, invstd, scale, shift = raw.gemm(*args, D, M, N, K, ldd=N, **kw,
                                                 bn_finish=dict(gamma=gamma, beta=beta, running_mean=rm, running_var=rv, eps=eps, momentum=mom))
        assert (D.double() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6
        sd = var_ref.sqrt()
        assert ((mean.double() - mean_ref).abs() <= 1e-5 * (mean_ref.abs() + sd)).all(), rep
        invstd_ref = 1.0 / (var_ref + eps).sqrt()
        assert ((invstd.double() - invstd_ref).abs() <= 1e-4 * invstd_ref).all(), rep
        assert ((scale.double() - gamma.double() * invstd_ref).abs() <= 1e-4 * (gamma.double() * invstd_ref).abs()).all()
        shift_ref = beta.double() - mean_ref * gamma.double() * invstd_ref
        assert ((shift.double() - shift_ref).abs() <= 1e-4 * (shift_ref.abs() + beta.double().abs() + 1.0)).all()
        rm_ref = (1 - mom) * rm0.double() + mom * mean_ref
        rv_ref = (1 - mom) * rv0.double() + mom * var_ref * M / (M - 1)
        assert ((rm.double() - rm_ref).abs() <= 1e-5 * (rm_ref.abs() + sd)).all()
        assert ((rv.double() - rv_ref).abs() <= 1e-4 * rv_ref.abs()).all()


# ---- the 3x3 halo-patch convolution kernel (csrc/conv3.hip) ---------------------------------------------------------------------
C3_CASES = [
    # Nb, Cin, H, W, Cout: rows per tile / tiles per image / row fragments, channel chunks
    (3, 64, 56, 56, 64),       # 2 rows x 56, 28 tiles per image, 7 fragments, one chunk (ResNet50 layer1)
    (2, 128, 28, 28, 128),     # 4 rows x 28, two chunks
    (2, 256, 14, 14, 256),     # 7 x 14 = 98 pixels (14 padding rows in the last fragment), four chunks, four column tiles
    (5, 512, 7, 7, 128),       # whole 7 x 7 image per tile, 4 fragments, eight chunks
    (2, 64, 16, 16, 192),      # 6 + 6 + 4 rows: ragged last tile of every image
    (3, 192, 10, 12, 64),      # 5 rows x 12 = 60 pixels: 4 fragments, three chunks (odd count: both loop parities end the walk)
    (1, 320, 8, 8, 64),        # 64 pixels exactly, five chunks
    (2, 64, 9, 80, 64),        # one image row per tile (80 pixels: 5 of the 7 fragments carry rows)
]


@pytest.mark.parametrize("with_stats", [False, True])
@pytest.mark.parametrize("case", C3_CASES)
def test_conv3x3_halo_kernel_matches_float64(case, with_stats):
    """3x3 stride-1 pad-1 forward convolutions with a plain bf16 result take the halo-patch kernel (hs_gemm picks it; the same
    call with an f32 result keeps the generic implicit GEMM and serves as a second reference): result within bf16 rounding of
    the float64 convolution of the same bf16 operands, BatchNorm statistics (when asked for) as in
    test_forward_gemm_finishes_the_batchnorm_statistics, zero padding exact, twice (the arrival counters re-arm)"""
    Nb, Cin, H, Wd, N = case
    g = torch.Generator().manual_seed(Nb * 1000 + Cin + H)
    BF = torch.bfloat16
    x = (torch.randn(Nb, Cin, H, Wd, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, Cin, 3, 3, generator=g) * 0.1).to(BF)
    ref4 = torch.nn.functional.conv2d(x.double(), w.double(), stride=1, padding=1)
    M, K = Nb * H * Wd, 9 * Cin
    ref = ref4.permute(0, 2, 3, 1).reshape(M, N).to(DEV)
    geom = raw.conv_geom(Nb, H, Wd, Cin, N, 3, 3, 1, 1)
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wg = w.to(DEV).contiguous(memory_format=torch.channels_last)
    kw = dict(a_kind=L.A_CONV, b_kind=L.B_KC, ldb=K, geom=geom)
    # the generic body, f32 result
    D32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    raw.gemm(xg, wg, D32, M, N, K, ldd=N, **kw)
    assert (D32.double() - ref).abs().max().item() <= 2e-5 * K * x.abs().max().item() * w.abs().max().item()
    for rep in range(2):
        D = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
        if not with_stats:
            raw.gemm(xg, wg, D, M, N, K, ldd=N, **kw)
        else:
            gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
            beta = (torch.randn(N, generator=g) * 0.3).to(DEV)
            rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
            D, mean, invstd, scale, shift = raw.gemm(xg, wg, D, M, N, K, ldd=N, **kw,
                                                     bn_finish=dict(gamma=gamma, beta=beta, running_mean=rm, running_var=rv, eps=1e-5, momentum=0.1))
            mean_ref, var_ref = ref.mean(0), ref.var(0, unbiased=False)
            sd = var_ref.sqrt()
            invstd_ref = 1.0 / (var_ref + 1e-5).sqrt()
            assert ((mean.double() - mean_ref).abs() <= 1e-5 * (mean_ref.abs() + sd)).all(), rep
            assert ((invstd.double() - invstd_ref).abs() <= 1e-4 * invstd_ref).all(), rep
            assert ((scale.double() - gamma.double() * invstd_ref).abs() <= 1e-4 * (gamma.double() * invstd_ref).abs()).all()
            assert ((rm.double() - 0.1 * mean_ref).abs() <= 1e-5 * (mean_ref.abs() + sd)).all()
            assert ((rv.double() - (0.9 + 0.1 * var_ref * M / (M - 1))).abs() <= 1e-4 * (0.9 + 0.1 * var_ref)).all()
        assert not torch.isnan(D.float()).any()
        # the stored value is the bf16 rounding of an f32 accumulation: half an ulp of the result + the accumulation's own noise
        assert ((D.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 2e-5 * K * 0.05).all(), rep
        assert (D.double() - D32.double()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()


# ---- the phase-pipelined body (csrc/gemm_p8.h): 256x256 / 256x128 / 128x128 tiles, full tiles of nt GEMMs --------------------
P8_CFGS = {7: (256, 256), 8: (256, 128), 9: (128, 128)}


@pytest.mark.parametrize("cfg", [7, 8, 9])
@pytest.mark.parametrize("K", [128, 192, 768])      # 2 K tiles (prologue + tail only), 3 (one steady iteration), 12 (BERT)
def test_p8_body_is_exact_on_integer_data(cfg, K):
    """Integer-valued bf16 operands: every product and partial sum is exact in f32, so ANY wrong fragment, slot, swizzle or
    stale / early LDS read shows as a mismatch.  Asymmetric operands (guide: A = I checks with a symmetric B hide transposes).
    Several output tiles per dimension so the XCD / L2-group tile map is exercised too."""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    bm, bn = P8_CFGS[cfg]
    M, N = 3 * bm, 5 * bn
    A, B = _rand((M, K), torch.bfloat16, True, 11), _rand((N, K), torch.bfloat16, True, 12)
    ref = A.double() @ B.double().t()
    bias = torch.arange(N, dtype=torch.float32) - 7.0
    try:
        lib.hs_gemm_debug(cfg, 0)
        for rep in range(3):       # a race would not show every time: the same launch three times
            D = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
            raw.gemm(A.to(DEV), B.to(DEV), D, M, N, K, lda=K, ldb=K)
            assert torch.equal(D.cpu().double(), ref), f"cfg {cfg} K {K} rep {rep}: f32 result"
        Db = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        raw.gemm(A.to(DEV), B.to(DEV), Db, M, N, K, lda=K, ldb=K, bias=bias.to(DEV))
        want = (ref + bias.double()).to(torch.bfloat16)
        assert torch.equal(Db.cpu(), want), f"cfg {cfg} K {K}: bf16 result + bias"
    finally:
        lib.hs_gemm_debug(-1, 0)


def test_p8_body_epilogues_match_the_generic_body():
    """every epilogue feature set the pipelined body has code for, against the generic body on the same operands (random
    data: both accumulate in f32 and walk K in the same order, so the results agree to the last bit)"""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    M, N, K = 512, 768, 320
    bf = torch.bfloat16
    A, B = _rand((M, K), bf, False, 21).to(DEV), _rand((N, K), bf, False, 22).to(DEV)
    bias = _rand((N,), torch.float32, False, 23).to(DEV)
    res = _rand((M, N), bf, False, 24).to(DEV)
    u = _rand((M, N), bf, False, 25).to(DEV)

    def run(cfg, **kw):
        lib.hs_gemm_debug(cfg, 0)
        try:
            outs = {}
            D = torch.full((M, N), float("nan"), dtype=kw.pop("out", bf), device=DEV)
            pre = torch.full((M, N), float("nan"), dtype=bf, device=DEV) if kw.pop("want_pre", False) else None
            raw.gemm(A, B, D, M, N, K, lda=K, ldb=K, preact=pre, **kw)
            outs["D"] = D.float().cpu()
            if pre is not None:
                outs["pre"] = pre.float().cpu()
            return outs
        finally:
            lib.hs_gemm_debug(-1, 0)
    cases = {
        "plain": {},
        "bias": dict(bias=bias),
        "bias+gelu+preact": dict(bias=bias, act=L.ACT_GELU, want_pre=True),
        "bias+residual": dict(bias=bias, residual=res),
        "bias+dropout+residual": dict(bias=bias, residual=res, dropout_p=0.1, dropout_seed=1234),
        "gelu-grad multiplier": dict(mul_mode=L.MUL_GELU_GRAD, mul_src=u),
        "residual": dict(residual=res),
        "f32 out": dict(out=torch.float32),
    }
    for name, kw in cases.items():
        want = run(2, **dict(kw))        # 64x64 tiles of the generic body
        for cfg in (7, 8, 9):
            if cfg == 7 and M % 256 == 0 and N % 256 == 0 or cfg == 8 and N % 128 == 0 or cfg == 9:
                got = run(cfg, **dict(kw))
                for k in want:
                    assert torch.equal(got[k], want[k]), f"{name}: cfg {cfg} {k} differs from the generic body"


def test_p8_body_row_sums_and_segments():
    """the K-contiguous weight gradient of a BertLayer on the pipelined body: dW = dY^T X with the bias gradient as row sums,
    Q/K/V in one launch as three output segments (reference: transformers BertSelfAttention's three Linear layers)"""
    import ctypes as C
    lib = L.lib()
    lib.hs_gemm_debug.argtypes = [C.c_int32, C.c_int32]
    rows, out_f, in_f = 640, 768, 512
    dYt = _rand((out_f, rows), torch.bfloat16, True, 31)
    Xt = _rand((in_f, rows), torch.bfloat16, True, 32)
    for cfg in (7, 8):
        try:
            lib.hs_gemm_debug(cfg, 0)
            dW = torch.full((out_f, in_f), float("nan"), dtype=torch.float32, device=DEV)
            db = torch.full((out_f,), float("nan"), dtype=torch.float32, device=DEV)
            raw.gemm(dYt.to(DEV), Xt.to(DEV), dW, out_f, in_f, rows, lda=rows, ldb=rows, rowsum_a=db)
            assert torch.equal(dW.cpu().double(), dYt.double() @ Xt.double().t()), cfg
            assert torch.equal(db.cpu().double(), dYt.double().sum(1)), cfg
        finally:
            lib.hs_gemm_debug(-1, 0)


def test_p8_body_is_what_bert_sized_projections_run():
    """the launcher's own choice (no override) on the fused Q/K/V projection of BERT-base at 4096 tokens takes the pipelined
    body (HAMSPINE_P8 default on) and agrees with float64"""
    M, N, K = 4096, 2304, 768
    A, B = _rand((M, K), torch.bfloat16, False, 41), _rand((N, K), torch.bfloat16, False, 42)
    bias = _rand((N,), torch.float32, False, 43)
    D = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    raw.gemm(A.to(DEV), B.to(DEV), D, M, N, K, lda=K, ldb=K, bias=bias.to(DEV))
    ref = A.double() @ B.double().t() + bias.double()
    assert (D.cpu().double() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()


# ---- streaming 1x1 convolution (csrc/pw_stream.hip) ---------------------------------------------------------------------------
def _merge_stats(st):
    """(count, mean, M2) partial rows -> per-channel mean and biased variance (Chan), in float64"""
    st = st.double().cpu()
    n = st[:, :, 0]
    tot = n.sum(0)
    mean = (n * st[:, :, 1]).sum(0) / tot
    m2 = (st[:, :, 2] + n * (st[:, :, 1] - mean) ** 2).sum(0)
    return tot, mean, m2 / tot


PW_SHAPES = [(100352, 256, 64), (100352, 64, 256), (100352, 64, 64), (100352, 128, 256), (25088, 512, 128), (6272, 1024, 256),
             (1000, 64, 64), (200, 128, 192), (1568, 2048, 256), (4160, 256, 128)]


@pytest.mark.parametrize("M,N,K", PW_SHAPES)
def test_streaming_pointwise_conv_matches_float64(M, N, K):
    """every 1x1 shape class of ResNet50 at batch 32 (rows x out x in), plus ragged row counts (tail block, fewer row blocks
    than workgroups): integer data -> the bf16 result and the column statistics are exact"""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randint(-3, 4, (M, K), generator=g).float().bfloat16()
    w = torch.randint(-2, 3, (N, K), generator=g).float().bfloat16()
    ref = x.double() @ w.double().t()
    assert ref.abs().max() < 256 * 8          # exactly representable after the bf16 rounding of the result? no: compare rounded
    y, st = raw.pointwise_fwd(x.to(DEV), w.to(DEV))
    assert torch.equal(y.cpu().double(), ref.bfloat16().double()), "result"
    assert st.shape[0] == int(L.lib().hs_pointwise_stat_rows(M, N, K)) and torch.isfinite(st).all()
    tot, mean, var = _merge_stats(st)
    assert torch.equal(tot, torch.full((N,), float(M), dtype=torch.float64)), "every row counted once"
    assert (mean - ref.mean(0)).abs().max() <= 1e-5 * max(ref.abs().max().item(), 1.0)
    assert (var - ref.var(0, unbiased=False)).abs().max() <= 1e-4 * max(ref.var(0, unbiased=False).max().item(), 1.0)
    # random data: agreement with the tiled GEMM body to bf16 rounding of the result
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    y, st = raw.pointwise_fwd(x.to(DEV), w.to(DEV))
    ref = x.double() @ w.double().t()
    assert (y.cpu().double() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()
    tot, mean, var = _merge_stats(st)
    assert (var - ref.var(0, unbiased=False)).abs().max() <= 1e-3 * ref.var(0, unbiased=False).max()
