#!/usr/bin/env python
"""Inference-side measurements on the C2 shape (ResNet50 + BERT-base, 224 px, batch 32, bf16):
  * eval forward with BatchNorm folded into the convolutions (no_grad) vs the autograd-capable eval forward;
  * test-time augmentation ["hflip"] / ["hflip","vflip","rot90"]: one fused batch vs the reference's per-variant loop;
  * input staging: pageable `.to(device)` per tensor vs BatchStager (pinned + copy stream + prefetch), f32 and u8 images.
Prints one JSON object; run on the GPU box:  python tools/inference_bench.py > gpurun_out/inference_bench.json
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd")]

import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    import hamspine
    from hamspine import inference as inf
    from hamspine import staging
    import model as product_model

    hamspine.require_device()
    hamspine.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    os.environ["HAMSPINE_BERT_RANDOM_INIT"] = "1"
    torch.manual_seed(1)
    with tempfile.TemporaryDirectory() as tmp:
        net = product_model.MultimodalBaselineModel(
            num_classes=bench.CLASSES, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
            text_model_name=bench.bert_base_dir(tmp), num_heads=8, image_backbone="resnet50", classifier_type="mlp",
            fusion_type="basic")
    net = net.to(dev).eval()
    images, ids, mask, labels = bench.synthetic(0, dev)
    B = images.shape[0]
    out = {"shape": "resnet50+bert-base, 224px, batch 32, bf16"}

    def fwd_nograd():
        with torch.no_grad():
            return net(images, ids, mask)

    def fwd_grad():
        return net(images, ids, mask)

    ms = timed(fwd_nograd, 30)
    out["eval_forward_folded_ms"] = round(ms, 3)
    out["eval_forward_folded_images_per_s"] = round(B / ms * 1e3, 1)
    ms = timed(fwd_grad, 30)
    out["eval_forward_autograd_ms"] = round(ms, 3)

    for names in (["hflip"], ["hflip", "vflip", "rot90"]):
        def fused():
            with torch.no_grad():
                return inf.predict_tta(net, images, ids, mask, transforms=names)

        def loop():
            with torch.no_grad():
                variants = [images, images.flip(-1)] if names == ["hflip"] else \
                    [images, images.flip(-1), images.flip(-2), torch.rot90(images, 1, (-2, -1))]
                return torch.stack([net(v.contiguous(), ids, mask) for v in variants], 0).mean(0)
        a, b = fused(), loop()
        key = "tta_" + "_".join(names)
        out[key + "_max_diff"] = float((a - b).abs().max())
        out[key + "_fused_ms"] = round(timed(fused, 20), 3)
        out[key + "_loop_ms"] = round(timed(loop, 20), 3)

    # staging: 8 host batches; consumer = the folded eval forward (so copies can hide behind compute)
    host = [(torch.randn(B, 3, bench.HW, bench.HW), torch.randint(0, bench.VOCAB, (B, bench.SEQ)),
             torch.ones(B, bench.SEQ, dtype=torch.int64)) for _ in range(8)]
    host_u8 = [(torch.randint(0, 256, (B, bench.HW, bench.HW, 3), dtype=torch.uint8), b[1], b[2]) for b in host]

    def naive():
        for im, i, m in host:
            with torch.no_grad():
                net(im.to(dev), i.to(dev), m.to(dev))

    stager = staging.BatchStager(host, dev, prefetch=2)            # built once, iterated once per epoch
    stager_u8 = staging.BatchStager(host_u8, dev, prefetch=2, u8_images=0)

    def staged():
        for im, i, m in stager:
            with torch.no_grad():
                net(im, i, m)

    def staged_u8():
        for im, i, m in stager_u8:
            with torch.no_grad():
                net(im, i, m)

    out["epoch8_naive_to_device_ms_per_batch"] = round(timed(naive, 5, 2) / 8, 3)
    out["epoch8_batch_stager_ms_per_batch"] = round(timed(staged, 5, 2) / 8, 3)
    out["epoch8_batch_stager_u8_ms_per_batch"] = round(timed(staged_u8, 5, 2) / 8, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
