"""Feasibility probe: capture the whole C2 training step (forward, backward, fused AdamW) in one HIP graph through
torch.cuda.graph and time replays against eager execution.  Measurement helper, not part of the product."""
import os
import sys
import tempfile
import time

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
    torch.manual_seed(1234)
    with tempfile.TemporaryDirectory() as tmp:
        net = product_model.MultimodalBaselineModel(
            num_classes=bench.CLASSES, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
            text_model_name=bench.bert_base_dir(tmp), num_heads=8, image_backbone="resnet50", classifier_type="mlp",
            fusion_type="basic")
    net = net.to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=0.01)
    images, ids, mask, labels = bench.synthetic(0, dev)

    side = torch.cuda.Stream()
    overlap = "--towers" in sys.argv

    def features():
        if not overlap:
            return net.forward_features(images, ids, mask)
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            text_tokens = net.text_encoder(ids, mask)
        image_tokens, _ = net._encode_image_tokens(images)
        cur.wait_stream(side)
        return net.fusion(image_tokens, text_tokens, mask)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net.classifier(features())
        loss = F.cross_entropy(logits, labels, label_smoothing=0.02)
        loss.backward()
        opt.step()
        return loss

    def timed(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        step()
    print(f"eager: {timed(step):.2f} ms/step", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {timed(g.replay):.2f} ms/step, loss {loss.item():.4f}", flush=True)


if __name__ == "__main__":
    main()
