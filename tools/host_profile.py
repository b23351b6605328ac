"""cProfile of the host side of C2 training steps (where does the enqueue time go: ctypes calls vs Python)."""
import cProfile
import os
import pstats
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

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
