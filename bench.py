#!/usr/bin/env python
"""Headline benchmark: training images/s of the multimodal hot path (BASELINE.json metric).

Workload C2: ResNet50 + BERT-base (L=128) + cross-attention FusionModule + MLP head,
224x224 images, batch 32 PER GPU (weak scaling, as reference mibf_net/train_resnet.py:111-119), bf16
activations with f32 accumulation, synthetic data, random-init weights of the real architecture.

One step = zero_grad -> forward_features -> classifier -> CrossEntropy(label_smoothing=0.02) -> backward
(-> gradient all-reduce over RCCL when N > 1) -> fused AdamW step   (mirrors reference scripts/train.py:362-385).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     : the dominant kernel family (bf16 MFMA GEMM / implicit-GEMM conv core, csrc/gemm_core.h):
                 algorithmic FLOPs of its launches / their summed HIP-event durations, measured in two extra
                 instrumented steps after the timed region (events perturb timing, so they are not in `value`).
  cpu_baseline : the CPU oracle (oracle/, a plain-PyTorch port of the same step) on the host cores, bounded
                 sample; a reported baseline, not a target.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BATCH, HW, SEQ, CLASSES, VOCAB = 32, 224, 128, 7, 30522
TRAIN_TFLOP_PER_STEP = 2.954          # SURVEY.md section 8(d), case C2 at B=32 (fwd+dgrad+wgrad, 1 MAC = 2 FLOP)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # /opt/skills/guides/MI355X_MICROARCH.md, dense bf16 MFMA


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def bert_base_dir(tmp):
    d = os.path.join(tmp, "bert-base-config")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(vocab_size=VOCAB, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                       intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                       attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                       layer_norm_eps=1e-12, pad_token_id=0), f)
    return d


def synthetic(rank, device):
    from oracle.procedural import synthetic_batch
    images, ids, mask, labels = synthetic_batch(BATCH, HW, SEQ, VOCAB, CLASSES, seed=1234 + rank, min_len=16)
    return [t.to(device) for t in (images, ids, mask, labels)]


def gpu_leg(args, rank, world, local_rank):
    import hamspine
    from hamspine import _lib as L
    from hamspine import functional as F
    from hamspine.optim import FusedAdamW
    import model as product_model

    hamspine.require_device()
    hamspine.set_compute_dtype("bf16")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    os.environ["HAMSPINE_BERT_RANDOM_INIT"] = "1"
    torch.manual_seed(1234)
    with tempfile.TemporaryDirectory() as tmp:
        net = product_model.MultimodalBaselineModel(
            num_classes=CLASSES, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
            text_model_name=bert_base_dir(tmp), num_heads=8, image_backbone="resnet50", classifier_type="mlp",
            fusion_type="basic")
    net = net.to(device).train()
    nparams = sum(p.numel() for p in net.parameters())
    ddp = None
    if world > 1 or os.environ.get("HAMSPINE_FORCE_DDP") == "1":   # the latter: hook / bucket overhead without collectives
        from hamspine.ddp import DataParallel
        ddp = DataParallel(net)
    # overlap_backward (updates enqueued while backward runs) measured 17.0 vs 16.7 ms/step here: the HBM-bound update
    # slows the GEMMs it runs beside by as much as it saves, so the optimizer steps after backward
    opt = FusedAdamW(net.parameters(), lr=1e-4, weight_decay=0.01)
    images, ids, mask, labels = synthetic(rank, device)

    def step():
        opt.zero_grad(set_to_none=True)
        feats = net.forward_features(images, ids, mask)
        logits = net.classifier(feats)
        loss = F.cross_entropy(logits, labels, label_smoothing=0.02)
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}: model ready ({nparams / 1e6:.1f} M params), warming up {args.warmup} steps")
    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_issue = time.perf_counter() - t0     # host time to enqueue the timed steps (GPU-bound when well below dt)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    final_loss = loss.item()
    log(f"rank {rank}: {args.steps} steps in {dt:.3f} s -> {dt / args.steps * 1e3:.2f} ms/step "
        f"(host enqueue {t_issue / args.steps * 1e3:.2f} ms/step), loss {final_loss:.4f}")

    # ---- roofline leg: per-launch HIP events around the dominant kernel family, two extra steps ----------
    lib = L.lib()
    lib.hs_prof_enable.argtypes = [C.c_int32]
    lib.hs_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    fl, ms, cnt = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int64 * 4)()
    # one kernel at a time for this leg (no weight-gradient side stream, towers back to back): the events then bracket
    # the kernel alone, which is what a per-kernel roofline means; the timed region above ran with the default streams
    # (text tower on its own stream; weight-gradient side stream only if HAMSPINE_OVERLAP=1)
    lib.hs_set_overlap(0)
    tower_default = os.environ.get("HAMSPINE_TOWER_OVERLAP", "1")
    os.environ["HAMSPINE_TOWER_OVERLAP"] = "0"
    step()
    fence()
    lib.hs_prof_enable(1)
    prof_steps = 2
    for _ in range(prof_steps):
        step()
    L.check(lib.hs_prof_collect(fl, ms, cnt), "hs_prof_collect")
    lib.hs_prof_enable(0)
    fence()
    bracket_us = C.c_float(0.0)   # what the event bracket costs around an empty kernel (reported, not subtracted)
    L.check(lib.hs_prof_calibrate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200, C.byref(bracket_us)),
            "hs_prof_calibrate")
    fam_flops = fl[0] + fl[1]
    fam_ms = ms[0] + ms[1]
    fam_launches = cnt[0] + cnt[1]
    achieved = fam_flops / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
    # HBM traffic per launch of the same kernel family: PMC counters need their own rocprofv3 passes (FETCH_SIZE and
    # WRITE_SIZE cannot share one on gfx950), so the figure is read from the committed summary of those passes
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = round(tj["traffic_bytes_per_launch"])
        traffic_src = "profiles/round1_pmc_traffic.json: " + tj["method"]
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
        "kernel": "gemm_bf16_kernel<BM,BN,BK,A,B> (bf16 MFMA GEMM + implicit-GEMM conv core)",
        "launches_per_step": fam_launches // prof_steps,
        "avg_launch_us": round(fam_ms * 1e3 / max(fam_launches, 1), 2),
        "kernel_ms_per_step": round(fam_ms / prof_steps, 3),
        "algorithmic_tflop_per_step": round(fam_flops / prof_steps / 1e12, 3),
        "traffic_unit": "bytes per launch (PMC, separate passes)", "traffic_source": traffic_src,
        "timing": "HIP events around every launch of the family on its own stream, 2 steps run without stream overlap; "
                  "the same bracket around an empty kernel reads event_bracket_us (not subtracted from avg_launch_us)",
        "event_bracket_us": round(bracket_us.value, 2),
        "split": {"gemm": {"tflops": round(fl[0] / max(ms[0], 1e-9) / 1e9, 1), "ms_per_step": round(ms[0] / prof_steps, 3)},
                  "conv": {"tflops": round(fl[1] / max(ms[1], 1e-9) / 1e9, 1), "ms_per_step": round(ms[1] / prof_steps, 3)}},
    }
    if args.gemm_log and rank == 0:   # per-launch table of one step (analysis aid, outside every timed region)
        lib.hs_prof_enable(1)
        step()
        L.check(lib.hs_prof_dump(args.gemm_log.encode()), "hs_prof_dump")
        lib.hs_prof_enable(0)
    lib.hs_set_overlap(1 if os.environ.get("HAMSPINE_OVERLAP") == "1" else 0)
    os.environ["HAMSPINE_TOWER_OVERLAP"] = tower_default
    return dt, final_loss, roofline, nparams


def cpu_leg(sample_batch=4, steps=24):
    """The oracle's restatement of the same step on the host cores (kind = "port"), bounded sample."""
    from oracle import models as om
    from oracle.procedural import synthetic_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))      # the GPU box grants 16 host cores per GPU
    torch.set_num_threads(cores)
    bert_cfg = dict(vocab_size=VOCAB, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                    hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    net = om.OMultimodalBaselineModel(num_classes=CLASSES, bert_cfg=bert_cfg, hidden_dim=256, dropout=0.2, num_heads=8,
                                      image_backbone="resnet50", classifier_type="mlp", fusion_type="basic").train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0.01)
    images, ids, mask, labels = synthetic_batch(sample_batch, HW, SEQ, VOCAB, CLASSES, seed=1234, min_len=16)

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net.classifier(net.forward_features(images, ids, mask))
        loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=0.02)
        loss.backward()
        opt.step()
    t0 = time.perf_counter()
    step()   # warm-up
    log(f"cpu oracle warm-up step: {time.perf_counter() - t0:.1f} s on {cores} threads")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    log(f"cpu oracle: {steps} steps of batch {sample_batch} in {dt:.1f} s")
    return {"value": round(sample_batch * steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the same C2 train step at batch {sample_batch}, fp32, "
                      f"torch {torch.__version__} CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--gemm-log", default=None, help="append a per-launch CSV of one extra step to this file")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (not used by the driver): several ranks on one card over gloo exercise the N>1 code path on a
    # one-GPU box -- HAMSPINE_DIST_BACKEND=gloo HAMSPINE_BENCH_DEVICE=0
    backend = os.environ.get("HAMSPINE_DIST_BACKEND", "nccl")
    if "HAMSPINE_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HAMSPINE_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    dt, final_loss, roofline, nparams = gpu_leg(args, rank, world, local_rank)
    if rank == 0:
        value = BATCH * world * args.steps / dt
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "training images/sec, ResNet50+BERT-base 224px bs32",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init weights of the real architecture)",
            "config": {"workload": "C2: ResNet50 + BERT-base(L=128) + cross-attention FusionModule + MLP head, "
                                   "224x224, batch 32 per GPU, CE(label_smoothing=0.02), fused AdamW, train-mode "
                                   "BatchNorm and dropout", "per_gpu_batch": BATCH, "global_batch": BATCH * world,
                       "seq_len": SEQ, "params": nparams, "parallelism": f"dp{world}"},
            "step_mfma_frac": round(TRAIN_TFLOP_PER_STEP / (ms_step * 1e-3) / MFMA_BF16_PEAK_TFLOPS, 4),
            "final_loss": round(final_loss, 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle (bounded sample) ...")
            out["cpu_baseline"] = cpu_leg()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
