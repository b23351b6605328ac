"""Is every difference between the product's f32-mode gradients and an exact (f64) evaluation a ReLU / max-pool DECISION
flip?  (the claim was made in a test comment and never demonstrated.)

The product's own decisions are read out of the image tower's saved arena (every BatchNorm+ReLU output and the stem's
max-pool arg-max: hs_resnet_debug_offsets), the CPU oracle is then evaluated in f64 with THOSE decisions forced
(y = x * product_mask, max-pool = the product's arg-max) -- same function the product differentiated, exact arithmetic --
and the product's gradients must match it to f32 rounding (1e-4 relative L2 per tensor), on the end-to-end fixtures whose
free-running comparison needed a 2e-2 bound.  The number of decisions that differ from the free f64 run is reported.
"""
import ctypes as C

import pytest
import torch

import golden_cases as gc
from decisions import Decisions, decisions

pytestmark = pytest.mark.gpu

import hamspine  # noqa: E402
from hamspine import _lib as L  # noqa: E402
from hamspine import tower  # noqa: E402
from oracle import models as om  # noqa: E402
from oracle.procedural import load_procedural  # noqa: E402

DEV = "cuda"
TOL = 1e-4


@pytest.fixture(autouse=True)
def _modes():
    hamspine.set_compute_dtype("f32")
    yield
    hamspine.set_compute_dtype("bf16")
    tower.DEBUG_KEEP = None


def _product_decisions(ent, saved):
    """-> (list of NCHW bool masks in the oracle's ReLU call order, max-pool indices in torch's convention)"""
    d = ent.desc
    offs = (C.c_int64 * 128)()
    n = L.lib().hs_resnet_debug_offsets(C.byref(d), offs, 128)
    assert n > 0
    N = d.stem.N
    P, Q = (d.stem.H + 6 - 7) // 2 + 1, (d.stem.W + 6 - 7) // 2 + 1
    P2, Q2 = (P + 2 - 3) // 2 + 1, (Q + 2 - 3) // 2 + 1
    Cs = d.stem.cb.Cout

    def act(off, n_, h, w, c):
        t = saved[off:off + n_ * h * w * c * 4].view(torch.float32).view(n_, h, w, c)
        return (t > 0).permute(0, 3, 1, 2).cpu()
    masks = [act(offs[0], N, P, Q, Cs)]
    taps = saved[offs[1]:offs[1] + N * P2 * Q2 * Cs].view(N, P2, Q2, Cs).permute(0, 3, 1, 2).cpu().long()
    p2 = torch.arange(P2).view(1, 1, P2, 1)
    q2 = torch.arange(Q2).view(1, 1, 1, Q2)
    pool_idx = (2 * p2 - 1 + taps // 3) * Q + (2 * q2 - 1 + taps % 3)
    k = 2
    H, W = P2, Q2
    for i in range(d.n_blocks):
        b = d.blocks[i]
        for j in range(b.n_main):
            cb = b.main[j]
            H = (H + 2 * cb.pad - cb.R) // cb.stride + 1
            W = (W + 2 * cb.pad - cb.R) // cb.stride + 1
            masks.append(act(offs[k], N, H, W, cb.Cout))      # inner stage outputs, then (j = n_main - 1) the block output
            k += 1
    assert k == n
    return masks, pool_idx


@pytest.mark.parametrize("name", ["e2e_basic_mlp", "e2e_multiscale_residual"])
def test_product_gradients_equal_f64_with_the_products_decisions(name, tmp_path):
    import model as product_model
    from hamspine import functional as F
    seed, kw = gc.E2E_CASES[name]
    images, ids, mask, labels, tab = gc.e2e_inputs(kw)
    d = gc.save_bert_dir(gc.TINY_BERT, str(tmp_path / "bert"))
    m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d,
                                              **gc.E2E_COMMON, **kw)
    m = load_procedural(m, seed).to(DEV).train()
    tower.DEBUG_KEEP = []
    logits = gc.e2e_forward(m, name, kw, images.to(DEV), ids.to(DEV), mask.to(DEV), tab.to(DEV))
    loss = F.cross_entropy(logits, labels.to(DEV), label_smoothing=0.02)
    loss.backward()
    torch.cuda.synchronize()
    assert len(tower.DEBUG_KEEP) == 1, "the image tower must have run once, through the tower executor"
    masks, pool_idx = _product_decisions(*tower.DEBUG_KEEP[0])
    gp = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}

    o = load_procedural(om.OMultimodalBaselineModel(bert_cfg=gc.TINY_BERT, **gc.E2E_COMMON, **kw), seed).double().train()

    def run(store, mode):
        o.zero_grad(set_to_none=True)
        with decisions(store, mode):
            lg = gc.e2e_forward(o, name, kw, images.double(), ids, mask, tab.double())
            torch.nn.functional.cross_entropy(lg, labels, label_smoothing=0.02).backward()
        return {k: p.grad.detach().clone() for k, p in o.named_parameters() if p.grad is not None}

    free = Decisions()
    g_free = run(free, "record")
    n_tower = len(masks)
    assert [tuple(a.shape) for a in free.relu[:n_tower]] == [tuple(a.shape) for a in masks], "ReLU call order / shapes"
    flips = sum(int((a != b).sum()) for a, b in zip(free.relu[:n_tower], masks))
    flips_pool = int((free.pool[0] != pool_idx).sum())
    total = sum(a.numel() for a in masks)
    forced = Decisions()
    forced.relu, forced.pool = list(masks), [pool_idx]          # tower decisions from the product; the rest decided by the run
    g_forced = run(forced, "force")
    worst, worst_free = 0.0, 0.0
    for k, g in g_forced.items():
        if k.endswith("key.bias"):
            continue                     # analytically zero (softmax shift invariance): rounding noise on every side
        e = (gp[k] - g).norm().item() / max(g.norm().item(), 1e-30)
        ef = (gp[k] - g_free[k]).norm().item() / max(g_free[k].norm().item(), 1e-30)
        worst, worst_free = max(worst, e), max(worst_free, ef)
        assert e <= TOL, f"{name}: grad {k}: {e:.3e} off the f64 evaluation with the product's own decisions (free f64: {ef:.3e})"
    print(f"{name}: {flips} of {total} ReLU decisions and {flips_pool} max-pool arg-maxima of the product differ from the free "
          f"f64 run; worst gradient error vs free f64 {worst_free:.2e}, vs f64 with the product's decisions {worst:.2e}")
