import sys, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "multimodal-diagnosis-ham-spine_amd")
import torch
import golden_cases as gc
import hamspine
from hamspine import functional as F
from oracle import models as om
from oracle.procedural import load_procedural
import model as product_model
hamspine.set_compute_dtype("f32")
name = sys.argv[1] if len(sys.argv) > 1 else "e2e_sequence_lstm"
seed, kw = gc.E2E_CASES[name]
images, ids, mask, labels, tab = gc.e2e_inputs(kw)
with tempfile.TemporaryDirectory() as tmp:
    d = gc.save_bert_dir(gc.TINY_BERT, tmp + "/bert")
    m = product_model.MultimodalBaselineModel(pretrained_image=False, image_weights_path=None, text_model_name=d, **gc.E2E_COMMON, **kw)
load_procedural(m, seed)
m = m.cuda().train()
o = load_procedural(om.OMultimodalBaselineModel(bert_cfg=gc.TINY_BERT, **gc.E2E_COMMON, **kw), seed).double().train()
seen = {}
for nm in ("layer1", "layer2", "layer3", "layer4"):
    for bi in range(2):
        getattr(m.image_encoder, nm)[bi].register_forward_hook(lambda mod, i, out, k=f"{nm}.{bi}": seen.__setitem__("p" + k, out.detach().float().cpu()))
        getattr(o.image_encoder.model, nm)[bi].register_forward_hook(lambda mod, i, out, k=f"{nm}.{bi}": seen.__setitem__("o" + k, out.detach()))
lg = gc.e2e_forward(m, name, kw, images.cuda(), ids.cuda(), mask.cuda(), tab.cuda())
lo = gc.e2e_forward(o, name, kw, images.double(), ids, mask, tab.double())
for nm in ("layer1", "layer2", "layer3", "layer4"):
    for bi in range(2):
        a, b = seen[f"p{nm}.{bi}"], seen[f"o{nm}.{bi}"]
        flips = ((a > 0) != (b > 0)).sum().item()
        near = (b.abs() < 1e-6).sum().item()
        print(nm, bi, "relu sign flips", flips, "of", a.numel(), " |ref|<1e-6:", near, " max err", (a.double() - b).abs().max().item())
print("logit err", (lg.cpu().double() - lo).abs().max().item())
F.cross_entropy(lg, labels.cuda(), label_smoothing=0.02).backward()
torch.nn.functional.cross_entropy(lo, labels, label_smoothing=0.02).backward()
po = dict(o.named_parameters())
rows = []
for k, p in m.named_parameters():
    if p.grad is None or po[k].grad is None: continue
    ref = po[k].grad
    rows.append(((p.grad.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-30), k, ref.norm().item()))
rows.sort(reverse=True)
import re
for r in rows:
    if not r[1].startswith("image_encoder.model.layer") and "key.bias" not in r[1]: print("%.3e  %-60s |ref| %.3e" % r)
print("median", sorted(r[0] for r in rows)[len(rows)//2])
