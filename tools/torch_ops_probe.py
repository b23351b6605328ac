"""Which ATen kernels (fills, copies, casts) does a C2 training step still launch, and from where?  torch.profiler with
stacks over two steps; prints every non-hamspine device kernel with its count and the Python frames that issued it."""
import collections
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import bench  # noqa: E402


def main():
    import hamspine
    from hamspine import functional as F
    from hamspine.optim import FusedAdamW
    import model as product_model
    hamspine.set_compute_dtype("bf16")
    os.environ["HAMSPINE_BERT_RANDOM_INIT"] = "1"
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as tmp:
        net = product_model.MultimodalBaselineModel(
            num_classes=bench.CLASSES, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
            text_model_name=bench.bert_base_dir(tmp), num_heads=8, image_backbone="resnet50", classifier_type="mlp",
            fusion_type="basic")
    net = net.to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=0.01)
    images, ids, mask, labels = bench.synthetic(0, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net.classifier(net.forward_features(images, ids, mask))
        loss = F.cross_entropy(logits, labels, label_smoothing=0.02)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    ops = collections.Counter()
    where = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        if ev.device_type.name == "CPU" and ev.name.startswith("aten::") and ev.name in (
                "aten::fill_", "aten::zero_", "aten::copy_", "aten::_to_copy", "aten::ones_like", "aten::zeros", "aten::add_",
                "aten::add", "aten::mul", "aten::contiguous", "aten::clone", "aten::sum", "aten::div"):
            shapes = str(ev.input_shapes)[:60]
            ops[(ev.name, shapes)] += 1
            frames = [f for f in (ev.stack or []) if "hamspine" in f or "model.py" in f or "modules/" in f or "bench" in f][:3]
            where[(ev.name, shapes)][" <- ".join(frames) or "(autograd engine / no python frame)"] += 1
    for k, n in ops.most_common(40):
        print(f"{n / 2:6.1f}/step  {k[0]:18s} {k[1]}")
        for w, c in where[k].most_common(3):
            print(f"          {c / 2:5.1f}  {w[:200]}")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))


if __name__ == "__main__":
    main()
