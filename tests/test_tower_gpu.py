"""Whole-tower executors (hs_resnet_* / hs_bert_*: one C call per tower and direction) against the per-block autograd
nodes they replace, the transposed-operand weight gradients against the row-major ones, and the bf16 transpose kernel.
The per-block path is itself held against the reference's vectors and the CPU oracle in test_product_gpu.py."""
import os

import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu

import hamspine  # noqa: E402
from hamspine import _lib as L  # noqa: E402
from hamspine import rt  # noqa: E402
from oracle.procedural import load_procedural  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _restore():
    yield
    os.environ.pop("HAMSPINE_TOWER_EXEC", None)
    os.environ.pop("HAMSPINE_XBLOCK_BN", None)
    hamspine.set_compute_dtype("bf16")
    L.lib().hs_set_wgrad_nt(1)


def _run_e2e(name, tmp_path, tower, mode):
    import model as product_model
    from hamspine import functional as F
    os.environ["HAMSPINE_TOWER_EXEC"] = "1" if tower else "0"
    hamspine.set_compute_dtype(mode)
    seed, kw = gc.E2E_CASES[name]
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / f"bert{int(tower)}"))
    m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                              **gc.E2E_COMMON, **kw)
    load_procedural(m, seed)
    m = m.to(DEV).train()
    images, ids, mask, labels, tab = [t.to(DEV) for t in gc.e2e_inputs(kw)]
    outs = []
    for _ in range(2):                 # two steps: the second one reuses the cached descriptors and gradient buffers
        m.zero_grad(set_to_none=True)
        logits = gc.e2e_forward(m, name, kw, images, ids, mask, tab)
        loss = F.cross_entropy(logits, labels, label_smoothing=0.02)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((logits.detach().float().cpu(), {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()
                                                     if p.grad is not None}))
    bufs = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    return outs, bufs


@pytest.mark.parametrize("name", ["e2e_basic_mlp", "e2e_multiscale_residual", "e2e_gate_globallocal"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_tower_executor_equals_per_block_nodes(name, mode, tmp_path):
    """same kernels in the same order on the same values: logits, every gradient and the BatchNorm buffers must be
    bit-identical between the one-call tower path and the per-block path (multi-scale taps, two tower passes per step
    with gate / global-local, both compute dtypes), on the first step and on the second (cached) one"""
    os.environ["HAMSPINE_XBLOCK_BN"] = "0"      # (the cross-block BatchNorm sums exist only in the tower path: next test)
    a, abuf = _run_e2e(name, tmp_path, True, mode)
    b, bbuf = _run_e2e(name, tmp_path, False, mode)
    for step in range(2):
        assert torch.equal(a[step][0], b[step][0]), f"step {step}: logits differ by {(a[step][0] - b[step][0]).abs().max():.3e}"
        assert a[step][1].keys() == b[step][1].keys()
        for k in a[step][1]:
            assert torch.equal(a[step][1][k], b[step][1][k]), f"step {step}: gradient {k} differs"
    assert abuf.keys() == bbuf.keys()
    for k in abuf:
        assert torch.equal(abuf[k], bbuf[k]), f"buffer {k} differs"


def test_cross_block_batchnorm_sums_match_the_separate_pass():
    """ResNet50 tower, bf16: a bottleneck block's first data-gradient GEMM also takes the previous block-final BatchNorm's
    backward sums (default where the block input is small; HAMSPINE_XBLOCK_BN=2: everywhere) against the separate
    bn_bwd_partial pass (=0): the same values summed in another order -> forward identical, gradients within bf16-path
    rounding of each other"""
    from hamspine.nn import resnet50
    hamspine.set_compute_dtype("bf16")
    x = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(DEV)
    cot = torch.randn(4, 10, generator=torch.Generator().manual_seed(2)).to(DEV)
    res = {}
    for setting in ("2", "1", "0"):
        os.environ["HAMSPINE_XBLOCK_BN"] = setting
        p = load_procedural(resnet50(num_classes=10), 5).to(DEV).train()
        y = p(x)
        (y.float() * cot).sum().backward()
        torch.cuda.synchronize()
        res[setting] = (y.detach().float().cpu(), {k: q.grad.detach().float().cpu() for k, q in p.named_parameters()})
    for setting in ("2", "1"):
        assert torch.equal(res[setting][0], res["0"][0])
        differing = 0
        for k, gb in res["0"][1].items():
            ga = res[setting][1][k]
            differing += int(not torch.equal(ga, gb))
            err = (ga - gb).norm().item() / max(gb.norm().item(), 1e-12)
            # the sums themselves (first consumers: the affine gradients of layer4's block-final BatchNorms) agree to f32
            # rounding; below them every bf16 re-rounding of a data gradient turns a 1e-7 difference into ulp flips, a few
            # BatchNorms later the two runs differ by bf16 noise (measured: 1e-4 at layer4.0.conv2, 4e-3 at layer3.5, 2e-2 in
            # layer1, 7e-2 at the stem's bias) -- the same floor the bf16 mode has against the oracle
            tol = 1e-5 if k in ("layer4.1.bn3.weight", "layer4.1.bn3.bias", "layer4.0.bn3.weight", "layer4.0.bn3.bias") else 0.3
            assert err <= tol, f"HAMSPINE_XBLOCK_BN={setting} {k}: relative L2 {err:.3e}"
        assert differing > 0, "the cross-block path did not run (bit-identical gradients)"


def test_tower_falls_back_when_a_hook_is_registered(tmp_path):
    """Grad-CAM style hooks on a stage boundary (reference scripts/run_analysis.py:126-133) must still fire"""
    import model as product_model
    seed, kw = gc.E2E_CASES["e2e_basic_mlp"]
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                              **gc.E2E_COMMON, **kw)
    m = load_procedural(m, seed).to(DEV).eval()
    images, ids, mask, labels, tab = [t.to(DEV) for t in gc.e2e_inputs(kw)]
    seen = {}
    h = m.image_encoder.layer4[-1].register_forward_hook(lambda mod, i, out: seen.__setitem__("l4", out.shape))
    ref = m(images, ids, mask)
    assert seen["l4"] == (4, 512, 2, 2)
    h.remove()
    again = m(images, ids, mask)          # tower path now
    assert torch.equal(ref, again)


def test_tower_gradient_accumulation_without_zero_grad(tmp_path):
    """two backwards without zero_grad: the second must ADD to the first (the tower's cached gradient buffer may not be
    overwritten while a parameter still holds it)"""
    import model as product_model
    from hamspine import functional as F
    hamspine.set_compute_dtype("f32")
    seed, kw = gc.E2E_CASES["e2e_basic_mlp"]
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                              **gc.E2E_COMMON, **kw)
    m = load_procedural(m, seed).to(DEV).train()
    images, ids, mask, labels, tab = [t.to(DEV) for t in gc.e2e_inputs(kw)]

    def bwd():
        F.cross_entropy(m.classifier(m.forward_features(images, ids, mask)), labels).backward()
    bwd()
    g1 = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    bwd()                                 # accumulates (BatchNorm batch statistics do not depend on the running buffers)
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        want = 2.0 * g1[k]
        err = (p.grad - want).abs().max().item()
        assert err <= 1e-5 * max(want.abs().max().item(), 1e-6) + 1e-9, f"{k}: accumulated gradient off by {err:.3e}"


@pytest.mark.parametrize("R,Cc", [(4096, 768), (256, 3072), (72, 40), (64, 64)])
def test_transpose_bf16(R, Cc):
    x = torch.randn(R, Cc, generator=torch.Generator().manual_seed(R + Cc)).bfloat16().to(DEV)
    y = torch.empty(Cc, R, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().hs_transpose_bf16(x.data_ptr(), y.data_ptr(), R, Cc, Cc, R, rt.stream()), "hs_transpose_bf16")
    assert torch.equal(y.cpu(), x.cpu().T.contiguous())


def test_transpose_bf16_multi():
    """several matrices per launch (ragged 64-tile edges, strided sources, more entries than one launch holds)"""
    import ctypes as C
    shapes = [(4096, 3072), (768, 3072), (3072, 768), (200, 72), (8, 8), (136, 1000), (2304, 768), (64, 64), (72, 200), (40, 24)]
    xs, ys = [], []
    for i, (R, Cc) in enumerate(shapes):
        ld = Cc + (16 if i % 3 == 1 else 0)
        full = torch.randn(R, ld, generator=torch.Generator().manual_seed(i)).bfloat16().to(DEV)
        xs.append(full)
        ys.append(torch.full((Cc, R), float("nan"), dtype=torch.bfloat16, device=DEV))
    n = len(shapes)
    src = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    dst = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
    Rs = (C.c_int32 * n)(*[s[0] for s in shapes])
    Cs = (C.c_int32 * n)(*[s[1] for s in shapes])
    lds = (C.c_int64 * n)(*[x.shape[1] for x in xs])
    ldd = (C.c_int64 * n)(*[s[0] for s in shapes])
    L.check(L.lib().hs_transpose_bf16_multi(n, src, dst, Rs, Cs, lds, ldd, rt.stream()), "hs_transpose_bf16_multi")
    for (R, Cc), x, y in zip(shapes, xs, ys):
        assert torch.equal(y.cpu(), x[:, :Cc].cpu().T.contiguous()), (R, Cc)


def test_bert_weight_gradients_transposed_operands_equal_row_major():
    """BertLayer backward in bf16: dW from transposed (K-contiguous) copies of dY and X, with the bias gradients as row sums
    of dY^T, against the row-major form: f32 accumulation in both, the K order inside a tile differs -> 1e-5 relative"""
    from hamspine.nn import BertConfig, BertModel
    hamspine.set_compute_dtype("bf16")
    cfg = BertConfig(vocab_size=500, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = load_procedural(BertModel(cfg), 4).to(DEV).train()
    ids = torch.randint(1, 500, (8, 128), generator=torch.Generator().manual_seed(1)).to(DEV)
    mask = torch.ones(8, 128, dtype=torch.long, device=DEV)
    mask[3, 70:] = 0
    cot = torch.randn(8, 128, 768, generator=torch.Generator().manual_seed(2)).to(DEV)
    grads = []
    for nt in (1, 0):
        L.lib().hs_set_wgrad_nt(nt)
        m.zero_grad(set_to_none=True)
        (m(input_ids=ids, attention_mask=mask).last_hidden_state.float() * cot).sum().backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        if k.endswith("key.bias"):
            continue                      # analytically zero gradient: rounding noise on both sides
        err = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        assert err <= 1e-5, f"{k}: transposed-operand gradient differs from the row-major one by {err:.3e}"


def test_weight_shadows_follow_every_kind_of_parameter_update(tmp_path):
    """bf16 weight shadows (hamspine.rt.ensure_shadows; on by default since round 3, HAMSPINE_WEIGHT_SHADOWS=0 turns them off): the
    towers read persistent bf16 copies instead of casting every weight in every forward.  A training run with them is
    bit-identical to one without, the fused AdamW step keeps them current without casts, and anything else that touches a
    parameter -- an in-place torch op, load_state_dict, FusedSGD -- is picked up by the next forward."""
    import model as product_model
    from hamspine import functional as F
    from hamspine.optim import FusedAdamW, FusedSGD
    name = "e2e_basic_mlp"
    seed, kw = gc.E2E_CASES[name]
    images, ids, mask, labels, tab = gc.e2e_inputs(kw)
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    hamspine.set_compute_dtype("bf16")

    def build():
        m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                                  **gc.E2E_COMMON, **kw)
        return load_procedural(m, seed).to(DEV).train()

    def run(m, opt, steps):
        out = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            logits = gc.e2e_forward(m, name, kw, images.to(DEV), ids.to(DEV), mask.to(DEV), tab.to(DEV))
            loss = F.cross_entropy(logits, labels.to(DEV), label_smoothing=0.02)
            loss.backward()
            opt.step()
            out.append(loss.item())
        return out

    from hamspine.nn.bert import BertModel
    rt.clear_shadows()
    os.environ["HAMSPINE_WEIGHT_SHADOWS"] = "0"
    try:
        m0 = build()
        l0 = run(m0, FusedAdamW(m0.parameters(), lr=1e-3, weight_decay=0.01), 3)     # without shadows: every forward casts
        assert not rt._shadows
    finally:
        os.environ.pop("HAMSPINE_WEIGHT_SHADOWS", None)
    try:
        _shadow_checks(build, run, m0, l0, BertModel, name, kw, images, ids, mask, tab)      # the default
    finally:
        rt.clear_shadows()


def _shadow_checks(build, run, m0, l0, BertModel, name, kw, images, ids, mask, tab):
    from hamspine.optim import FusedAdamW, FusedSGD
    m1 = build()
    opt1 = FusedAdamW(m1.parameters(), lr=1e-3, weight_decay=0.01)
    l1 = run(m1, opt1, 3)
    w = m1.image_encoder.model.layer1[0].conv1.weight
    bert = next(mod for mod in m1.text_encoder.modules() if isinstance(mod, BertModel))
    qw = bert.encoder.layer[0].attention.self.query.weight
    assert rt.shadow_ptr_of(w) is not None and rt.shadow_ptr_of(qw) is not None, "the towers registered no shadows"
    assert l0 == l1, (l0, l1)
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k

    def shadow_equals_weight(p):
        rec = rt._shadows[id(p)]
        return torch.equal(rec.view.cpu(), p.detach().reshape(-1).to(torch.bfloat16).cpu()
                           if p.dim() != 4 else p.detach().permute(0, 2, 3, 1).reshape(-1).to(torch.bfloat16).cpu())
    torch.cuda.synchronize()
    assert shadow_equals_weight(w) and shadow_equals_weight(qw), "AdamW did not keep the shadows current"
    # an in-place torch op: picked up through _version by the next forward
    with torch.no_grad():
        w.mul_(0.5)
    assert not shadow_equals_weight(w)
    gc.e2e_forward(m1, name, kw, images.to(DEV), ids.to(DEV), mask.to(DEV), tab.to(DEV))
    torch.cuda.synchronize()
    assert shadow_equals_weight(w)
    # load_state_dict
    m1.load_state_dict(m0.state_dict())
    gc.e2e_forward(m1, name, kw, images.to(DEV), ids.to(DEV), mask.to(DEV), tab.to(DEV))
    torch.cuda.synchronize()
    assert shadow_equals_weight(w) and shadow_equals_weight(qw)
    # a raw-pointer writer that does not write shadows
    run(m1, FusedSGD(m1.parameters(), lr=1e-2), 1)
    gc.e2e_forward(m1, name, kw, images.to(DEV), ids.to(DEV), mask.to(DEV), tab.to(DEV))
    torch.cuda.synchronize()
    assert shadow_equals_weight(w) and shadow_equals_weight(qw)


def test_grouped_weight_gradients_of_an_mlp_equal_separate_launches():
    """hs_wgrad_group_begin / _end (the fused MLP node groups its two K-contiguous weight-gradient GEMMs into one grid of
    256x128 tiles when the outputs are wide enough): same gradients as two Linear nodes (f32 accumulation in both; tile shape
    and the fused bias row sums change the summation order -> 2e-5 relative), data gradient included."""
    from hamspine import functional as F
    hamspine.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(11)
    M, d, hid = 4096, 1024, 4096
    x = (torch.randn(M, d, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(hid, d, generator=g) * 0.03).to(DEV)
    b1 = (torch.randn(hid, generator=g) * 0.1).to(DEV)
    w2 = (torch.randn(d, hid, generator=g) * 0.03).to(DEV)
    b2 = (torch.randn(d, generator=g) * 0.1).to(DEV)
    cot = torch.randn(M, d, generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    for fused in (True, False):
        ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        xi = x.clone().requires_grad_(True)
        if fused:
            y = F.mlp_gelu(xi, *ps)
        else:
            y = F.linear(F.linear(xi, ps[0], ps[1], act="gelu"), ps[2], ps[3])
        (y.float() * cot.float()).sum().backward()
        torch.cuda.synchronize()
        outs.append([y.detach().float().cpu(), xi.grad.float().cpu()] + [p.grad.float().cpu() for p in ps])
    names = ["y", "dx", "dw1", "db1", "dw2", "db2"]
    for nme, a, b in zip(names, outs[0], outs[1]):
        # dw2 / db2 read the same operands in both forms (only the summation order differs); everything downstream of the
        # hidden gradient differs by its bf16 rounding (the fused node rounds gelu' * (dy W2) once, the two nodes twice)
        tol = 2e-5 if nme in ("dw2", "db2") else 2e-2
        assert (a - b).abs().max().item() <= tol * b.abs().max().item() + 1e-6, nme


def test_tower_outputs_are_guarded_against_in_place_edits(tmp_path):
    """the tower's outputs are views into the arena its backward reads (round-2 advisor finding): a consumer that modifies
    one in place must get autograd's version-counter error instead of silently corrupted gradients; a second backward
    through the same graph is refused with a clear message"""
    from hamspine.nn import BertConfig, BertModel
    cfg = BertConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = load_procedural(BertModel(cfg), 3).to(DEV).train()
    ids = torch.randint(1, 100, (2, 16), generator=torch.Generator().manual_seed(1)).to(DEV)
    mask = torch.ones(2, 16, dtype=torch.long, device=DEV)
    h = m(input_ids=ids, attention_mask=mask).last_hidden_state
    # in place, on the view of the saved arena: autograd refuses either at the edit (the output is a view made inside a
    # custom Function) or, for a non-view output, at backward through the version counter
    with pytest.raises(RuntimeError, match="inplace"):
        h.mul_(2.0)
        h.float().sum().backward()
    h = m(input_ids=ids, attention_mask=mask).last_hidden_state
    loss = h.float().sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
