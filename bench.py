#!/usr/bin/env python
"""Headline benchmark: training images/s of the multimodal hot path (BASELINE.json metric).

Workloads (`--workload`, default c2 = the configuration BASELINE.json's metric is quoted on):
  c2  ResNet50 + BERT-base (L=128) + cross-attention FusionModule + MLP head, batch 32 per GPU        (configs[1])
  c3  MIBF-Net: ResNet50 + fc768, BERT-base CLS, IBFA both ways, three heads, MP-Loss, batch 32 per GPU (configs[2])
  c4  ConNeXT: ConvNeXt-base + BERT-base CLS + conv cross-attention, batch 64 per GPU                  (configs[3])
  c5  ResNet18 multi-scale taps x BERT through 3 CrossAttentionBlocks + global-local + gate + KAN head  (configs[4])
224x224 synthetic images, 128-token synthetic captions, bf16 activations with f32 accumulation, random-init weights of
the real architecture, train-mode BatchNorm and dropout, weak scaling (per-GPU batch fixed, as reference
mibf_net/train_resnet.py:111-119).

One step = zero_grad -> forward -> loss -> backward (-> gradient all-reduce over RCCL when N > 1) -> fused AdamW step
(mirrors reference scripts/train.py:362-385 / mibf_net/train_resnet.py:29-33).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     : the dominant kernel family (bf16 MFMA GEMM / implicit-GEMM conv core, csrc/gemm_core.h): algorithmic
                 FLOPs of its launches / their summed HIP-event durations, measured live in two extra instrumented steps
                 after the timed region (events perturb timing, so they are not in `value`); `by_class` splits the family
                 by shape class and prices the memory-bound members (1x1 convolutions with K <= 128, < 64 FLOP/B) against
                 HBM bytes instead of MFMA FLOPs.
  f32_mode     : the same step in exact-f32 mode (the mode that meets the 1e-4 parity bound): images/s and the fraction
                 of the 157.3 TFLOP/s f32 MFMA peak (N=1 only).
  cpu_baseline : the CPU oracle (oracle/, a plain-PyTorch port of the same step) on the host cores, bounded sample;
                 a reported baseline, not a target.  The product legs never import oracle/.
"""
import argparse
import collections
import csv
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

HW, SEQ, VOCAB = 224, 128, 30522
BATCH, CLASSES = 32, 7                 # workload c2 (tools/ import these)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # /opt/skills/guides/MI355X_MICROARCH.md, dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3          # same guide, f32-in MFMA (= the f32 vector rate)
HBM_PEAK_GBS = 8000.0                 # same guide, HBM3E spec
WORKLOADS = {
    # name: (per-GPU batch, classes, train TFLOP per step (fwd + dgrad + wgrad, 1 MAC = 2 FLOP), description)
    "c2": (32, 7, 2.954, "C2: ResNet50 + BERT-base(L=128) + cross-attention FusionModule + MLP head, 224x224, batch 32 per GPU, "
                         "CE(label_smoothing=0.02), fused AdamW, train-mode BatchNorm and dropout"),
    "c3": (32, 6, 2.932, "C3: MIBF-Net (ResNet50+fc768, BERT-base CLS, IBFA x2, 3 heads, MP-Loss), 224x224, L=128, batch 32 per "
                         "GPU, fused Adam, train-mode BatchNorm and dropout"),
    "c4": (64, 7, 10.27, "C4: ConNeXT (ConvNeXt-base + BERT-base CLS + conv cross-attention), 224x224, L=128, batch 64 per GPU, "
                         "CE, fused Adam, train-mode dropout / stochastic depth"),
    "c5": (32, 7, 2.922, "C5: hierarchical fusion -- ResNet18 layer2/3/4 taps x 3 CrossAttentionBlocks (MultiScaleFusionModule) + "
                         "global-local dual stream (two image-tower passes: full image and 0.6 centre crop) + BERT-base(L=128) + "
                         "dual-expert gate + KAN head, 224x224, batch 32 per GPU, CE(label_smoothing=0.02), fused AdamW"),
}
C5_KW = dict(fusion_type="multiscale", classifier_type="kan", kan_num_groups=8, kan_act_mode="gelu", gate_enabled=True,
             global_local_enabled=True, global_local_crop_ratio=0.6)      # reference model.py:22-58; configs[4] of BASELINE.json


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


def synthetic(rank, device, batch=32, classes=7):
    from hamspine.synthetic import synthetic_batch
    images, ids, mask, labels = synthetic_batch(batch, HW, SEQ, VOCAB, classes, seed=1234 + rank, min_len=16)
    return [t.to(device) for t in (images, ids, mask, labels)]


def build_workload(name, device, rank):
    """-> (module, step_fn factory(opt, ddp), optimizer class kwargs)"""
    import model as product_model
    from hamspine import functional as F
    from hamspine.optim import FusedAdam, FusedAdamW
    batch, classes, _, _ = WORKLOADS[name]
    os.environ["HAMSPINE_BERT_RANDOM_INIT"] = "1"
    torch.manual_seed(1234)
    with tempfile.TemporaryDirectory() as tmp:
        bdir = bert_base_dir(tmp)
        if name == "c2":
            net = product_model.MultimodalBaselineModel(
                num_classes=classes, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
                text_model_name=bdir, num_heads=8, image_backbone="resnet50", classifier_type="mlp", fusion_type="basic")
        elif name == "c5":
            net = product_model.MultimodalBaselineModel(
                num_classes=classes, hidden_dim=256, dropout=0.2, pretrained_image=False, image_weights_path=None,
                text_model_name=bdir, num_heads=8, image_backbone="resnet18", **C5_KW)
        elif name == "c3":
            from mibf_net.model_resnet import Resnet50WithOurs
            net = Resnet50WithOurs(num_labels=classes, loss_class="KL_loss", bert_path=bdir)
        else:
            from ConNexT.models.ourmodel import OurClassfierConvnextV2
            net = OurClassfierConvnextV2(num_labels=classes, pretrained=False, bert_path=bdir)
    net = net.to(device).train()
    images, ids, mask, labels = synthetic(rank, device, batch, classes)
    if name in ("c2", "c5"):
        def fwd_loss():                                                                  # scripts/train.py:373-381
            logits = net.classifier(net.forward_features(images, ids, mask))
            return F.cross_entropy(logits, labels, label_smoothing=0.02)
        make_opt = lambda: FusedAdamW(net.parameters(), lr=1e-4, weight_decay=0.01)      # scripts/train.py:257-261
    elif name == "c3":
        bd = {"input_ids": ids, "attention_mask": mask, "transformed_image": images}

        def fwd_loss():
            return net.cal_loss(net(bd), labels)                                         # mibf_net/train_resnet.py:30-31
        make_opt = lambda: FusedAdam(net.parameters(), lr=1e-4)                          # :137
    else:
        bd = {"input_ids": ids, "attention_mask": mask, "transformed_image": images}

        def fwd_loss():
            return F.cross_entropy(net(bd), labels)                                      # pl_model_MOE2.py:104-113
        make_opt = lambda: FusedAdam(net.parameters(), lr=1e-4)
    return net, fwd_loss, make_opt


RIDGE_FLOP_PER_BYTE = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)     # 312: below it a launch is HBM-bound at peak


def shape_class(combo, M, N, K, R):
    """shape classes of the GEMM family.  Plain GEMMs by operand layout (combo 0 nt: forward GEMMs and -- on transposed
    operand / weight copies -- the BERT weight and data gradients; 1 nn: data gradients on row-major weights; 2 tn: weight
    gradients on row-major operands); convolutions by filter and role (3 forward, 4 data gradient, 5 weight gradient)."""
    if combo < 3:
        bert = (M == 4096 or K == 4096) and min(M, N, K) >= 768 and max(M, N) <= 4096
        return ("bert_" if bert else "pointwise_") + ("nt", "nn", "tn")[combo]   # pointwise = 1x1 convolutions and small Linear layers
    role = ("fwd", "dgrad", "wgrad")[combo % 3]
    return ("conv3x3_" if R == 3 else "conv7x7_" if R == 7 else "conv1x1s2_") + role


def roofline_by_class(path, steps):
    """per shape class and bound: launches whose algorithmic intensity (FLOP per byte of operands + result, each moved
    once) is below the chip's ridge point (312 FLOP/B) cannot reach the MFMA peak however good the kernel -- they are
    priced against HBM bytes, everything else against bf16 MFMA FLOPs"""
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.reader(f):
            cls, combo, cfg, M, N, K, batch, split, R, stride = map(int, r[:10])
            if cls > 1:
                continue
            ms, flop = float(r[10]), float(r[11])     # FLOPs as the launcher counted them (strided dgrad: executed taps)
            nbytes = 2.0 * batch * (M * K + N * K) + (4.0 if combo in (2, 5) else 2.0) * batch * M * N
            if combo == 3:
                nbytes = 2.0 * (M * K / (R * R) * stride * stride + N * K) + 2.0 * M * N     # the image is read once, not R*R times
            elif combo == 4:
                nbytes = 2.0 * (M * N + N * K) + 2.0 * M * K / (R * R) / (stride * stride)    # dy read once
            elif combo == 5:
                nbytes = 2.0 * (M * K + N * K / (R * R) * stride * stride) + 4.0 * M * N
            name = shape_class(combo, M, N, K, R)
            if combo >= 6:                            # grouped weight gradients of the image tower: M problems in one grid
                nbytes = float(r[12])
                name = {6: "pointwise_wgrad_grouped", 7: "conv_wgrad_grouped", 8: "bert_wgrad_grouped"}[combo]
            bound = "mfma" if flop / nbytes >= RIDGE_FLOP_PER_BYTE else "hbm"
            a = agg.setdefault((name, bound), {"launches": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0})
            a["launches"] += 1
            a["ms"] += ms
            a["flop"] += flop
            a["bytes"] += nbytes
    out = {}
    for (name, bound), a in agg.items():
        tf = a["flop"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
        gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
        frac = tf / MFMA_BF16_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS
        out[f"{name}/{bound}"] = {"launches_per_step": a["launches"] // steps, "ms_per_step": round(a["ms"] / steps, 3),
                                  "tflops": round(tf, 1), "algorithmic_gbs": round(gbs, 1), "frac_of_bound": round(frac, 4)}
    return out


def gpu_leg(args, rank, world, local_rank):
    import hamspine
    from hamspine import _lib as L

    hamspine.require_device()
    hamspine.set_compute_dtype("bf16")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    batch, classes, train_tflop, _ = WORKLOADS[args.workload]
    net, fwd_loss, make_opt = build_workload(args.workload, device, rank)
    nparams = sum(p.numel() for p in net.parameters())
    ddp = None
    if world > 1 or os.environ.get("HAMSPINE_FORCE_DDP") == "1":   # the latter: hook / bucket overhead without collectives
        from hamspine.ddp import DataParallel
        ddp = DataParallel(net)
    opt = make_opt()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = fwd_loss()
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}: {args.workload} ready ({nparams / 1e6:.1f} M params), warming up {args.warmup} steps")
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
    # host cost of enqueuing ONE step into an empty queue (in the timed loop above the host runs at most a queue's depth
    # ahead of the GPU and then blocks inside the launch call, so its per-step time there only mirrors the GPU's)
    host_samples = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        step()
        host_samples.append(time.perf_counter() - t1)
    fence()
    host_ms = sorted(host_samples)[1] * 1e3
    log(f"rank {rank}: {args.steps} steps in {dt:.3f} s -> {dt / args.steps * 1e3:.2f} ms/step "
        f"(host enqueue: {t_issue / args.steps * 1e3:.2f} ms/step inside the timed loop, {host_ms:.2f} ms for one step into an "
        f"empty queue), loss {final_loss:.4f}")
    if ddp is not None:
        log(f"rank {rank}: bucket collectives launched during backward / in finish(): {ddp.stats}")

    # ---- roofline leg: per-launch HIP events around the dominant kernel family, two extra steps ----------
    lib = L.lib()
    lib.hs_prof_enable.argtypes = [C.c_int32]
    lib.hs_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    # one kernel at a time for this leg (no weight-gradient side stream, towers back to back): the events then bracket
    # the kernel alone, which is what a per-kernel roofline means; the timed region above ran with the default streams
    # (text tower on its own stream; weight-gradient side stream only if HAMSPINE_OVERLAP=1)
    lib.hs_set_overlap(0)
    tower_default = os.environ.get("HAMSPINE_TOWER_OVERLAP", "1")
    os.environ["HAMSPINE_TOWER_OVERLAP"] = "0"
    step()
    fence()
    prof_steps = 2
    lib.hs_prof_enable(1)
    for _ in range(prof_steps):
        step()
    with tempfile.TemporaryDirectory() as tmp:
        rec = os.path.join(tmp, "launches.csv")
        L.check(lib.hs_prof_dump(rec.encode()), "hs_prof_dump")       # synchronises, writes one line per launch
        lib.hs_prof_enable(0)
        by_class = roofline_by_class(rec, prof_steps)
        fl = [0.0, 0.0]
        ms = [0.0, 0.0]
        cnt = [0, 0]
        with open(rec) as f:
            for r in csv.reader(f):
                c = int(r[0])
                if c <= 1:
                    fl[c] += float(r[11])
                    ms[c] += float(r[10])
                    cnt[c] += 1
        if args.gemm_log and rank == 0:
            import shutil
            shutil.copy(rec, args.gemm_log)
    fence()
    bracket_us = C.c_float(0.0)   # what the event bracket costs around an empty kernel (reported, not subtracted)
    L.check(lib.hs_prof_calibrate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200, C.byref(bracket_us)),
            "hs_prof_calibrate")
    fam_flops, fam_ms, fam_launches = fl[0] + fl[1], ms[0] + ms[1], cnt[0] + cnt[1]
    achieved = fam_flops / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
    # HBM traffic per launch of the same kernel family: PMC counters need their own rocprofv3 passes (FETCH_SIZE and
    # WRITE_SIZE cannot share one on gfx950), so the figure is read from the committed summary of those passes
    # The file carries the build id (hash of the kernel sources) it was measured on: a figure from another build is not
    # reported (null) instead of going stale silently.
    traffic, traffic_src, traffic_by_class = None, None, None
    tpath = os.path.join(ROOT, "profiles", "round3_pmc_traffic.json")
    if args.workload == "c2" and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            from pmc_traffic import build_id
            here = build_id()
        except Exception:      # noqa: BLE001
            here = None
        if here is not None and tj.get("build_id") == here:
            traffic = round(tj["traffic_bytes_per_launch"])
            traffic_src = f"profiles/round3_pmc_traffic.json (build {here}): " + tj["method"]
            traffic_by_class = {k: v["ratio"] for k, v in tj.get("by_class", {}).items()} or None
        else:
            traffic_src = (f"profiles/round3_pmc_traffic.json was measured on build {tj.get('build_id')}, this is build {here}: "
                           "not reported (re-run tools/final_profiles.sh)")
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
        "kernel": "gemm_bf16_* (bf16 MFMA GEMM / implicit-GEMM convolution family: tiled, grouped and phase-pipelined bodies, + conv3_bf16_kernel)",
        "launches_per_step": fam_launches // prof_steps,
        "avg_launch_us": round(fam_ms * 1e3 / max(fam_launches, 1), 2),
        "kernel_ms_per_step": round(fam_ms / prof_steps, 3),
        "algorithmic_tflop_per_step": round(fam_flops / prof_steps / 1e12, 3),
        "traffic_unit": "bytes per launch (PMC, separate passes)", "traffic_source": traffic_src,
        "traffic_over_algorithmic_by_class": traffic_by_class,
        "timing": "HIP events around every launch of the family on its own stream, 2 steps run without stream overlap; "
                  "the same bracket around an empty kernel reads event_bracket_us (not subtracted from avg_launch_us)",
        "event_bracket_us": round(bracket_us.value, 2),
        "split": {"gemm": {"tflops": round(fl[0] / max(ms[0], 1e-9) / 1e9, 1), "ms_per_step": round(ms[0] / prof_steps, 3)},
                  "conv": {"tflops": round(fl[1] / max(ms[1], 1e-9) / 1e9, 1), "ms_per_step": round(ms[1] / prof_steps, 3)}},
        "by_class": by_class,
    }
    lib.hs_set_overlap(1 if os.environ.get("HAMSPINE_OVERLAP") == "1" else 0)
    os.environ["HAMSPINE_TOWER_OVERLAP"] = tower_default

    # ---- exact-f32 mode (the mode that meets the 1e-4 parity bound), N = 1 only ------------------------------------
    f32_mode = None
    if world == 1 and not args.no_f32:
        hamspine.set_compute_dtype("f32")
        for _ in range(2):
            step()
        fence()
        n32 = max(3, min(args.steps, 8))
        t0 = time.perf_counter()
        for _ in range(n32):
            step()
        fence()
        d32 = (time.perf_counter() - t0) / n32
        f32_mode = {"value": round(batch / d32, 2), "unit": "images/s", "ms_per_step": round(d32 * 1e3, 3), "steps": n32,
                    "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact fmaf chain)",
                    "step_mfma_frac": round(train_tflop / d32 / MFMA_F32_PEAK_TFLOPS, 4), "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                    "parity": "logits <= 1e-4 * max|ref| of the CPU reference at this size (tests/test_fullsize_gpu.py)"}
        hamspine.set_compute_dtype("bf16")
        log(f"f32 mode: {d32 * 1e3:.2f} ms/step")
    # ---- the N>1 configuration on this one GPU (N = 1 only): hamspine.ddp around the same model with a one-rank RCCL group,
    # so the bucket hooks, the tower milestones, the side stream and a real `all_reduce(AVG, async_op=True)` per bucket all
    # run; the step time is the per-rank cost of the data-parallel step minus the wire time --------------------------------
    ddp_cfg = None
    if world == 1 and ddp is None and not args.no_ddp_config:
        try:
            ddp_cfg = ddp_config_leg(net, fwd_loss, make_opt, device, batch, min(args.steps, 10))
        except Exception as e:      # noqa: BLE001  (reported in the line, never fatal for the headline)
            ddp_cfg = {"error": repr(e)}
            log(f"ddp-config leg failed: {e!r}")
    return dt, final_loss, roofline, nparams, host_ms, f32_mode, ddp_cfg


def ddp_config_leg(net, fwd_loss, make_opt, device, batch, steps):
    from hamspine.ddp import DataParallel
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29731")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        created = True
    os.environ["HAMSPINE_DDP_SINGLE"] = "1"
    try:
        out = {}
        for algo in ("allreduce", "direct_bf16"):
            ddp = DataParallel(net, algo=algo)
            opt = make_opt()

            def step():
                opt.zero_grad(set_to_none=True)
                loss = fwd_loss()
                loss.backward()
                ddp.finish()
                opt.step()
                return loss
            for _ in range(6):                       # (the wrapper learns its unused set on step 1 and its launch order on step 3)
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / steps
            out[algo] = {"ms_per_step": round(d * 1e3, 3), "images_per_s": round(batch / d, 1), "loss": round(loss.item(), 4),
                         "bucket_exchanges_in_backward": ddp.stats["launched_in_backward"],
                         "bucket_exchanges_in_finish": ddp.stats["launched_in_finish"], "buckets": len(ddp.buckets),
                         "bucket_mb": [round(b.flat.numel() * 4 / 2**20, 1) for b in ddp.buckets]}
            log(f"ddp-config [{algo}]: {d * 1e3:.2f} ms/step, exchanges in backward / in finish(): {ddp.stats}")
            ddp.detach()                             # hooks and bucket slots of this wrapper go away before the next one
            del ddp, opt
        out["backend"] = "nccl (RCCL), world size 1: every collective of the N>1 step is issued; wire time is not represented"
        out["steps"] = steps
        return out
    finally:
        os.environ.pop("HAMSPINE_DDP_SINGLE", None)
        if created:
            dist.destroy_process_group()


def cpu_leg(workload, steps=3):
    """The oracle's restatement of the same step on the host cores (kind = "port"), bounded sample: the SAME batch size
    as the GPU leg, `steps` timed steps (+1 warm-up)."""
    from oracle import models as om
    from oracle.procedural import synthetic_batch
    batch, classes, _, _ = WORKLOADS[workload]
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
    images, ids, mask, labels = synthetic_batch(batch, HW, SEQ, VOCAB, classes, seed=1234, min_len=16)
    if workload in ("c2", "c5"):
        kw = dict(image_backbone="resnet50", classifier_type="mlp", fusion_type="basic") if workload == "c2" else \
            dict(image_backbone="resnet18", **C5_KW)
        net = om.OMultimodalBaselineModel(num_classes=classes, bert_cfg=bert_cfg, hidden_dim=256, dropout=0.2, num_heads=8, **kw).train()
        opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0.01)

        def fwd_loss():
            return torch.nn.functional.cross_entropy(net.classifier(net.forward_features(images, ids, mask)), labels,
                                                     label_smoothing=0.02)
    elif workload == "c3":
        net = om.OResnet50WithOurs(classes, bert_cfg, "KL_loss").train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        bd = {"input_ids": ids, "attention_mask": mask, "transformed_image": images}

        def fwd_loss():
            return net.cal_loss(net(bd), labels)
    else:
        net = om.OConNeXT(classes, bert_cfg, {}).train()            # {} = ConvNeXt-base defaults
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        bd = {"input_ids": ids, "attention_mask": mask, "transformed_image": images}

        def fwd_loss():
            return torch.nn.functional.cross_entropy(net(bd), labels)

    def step():
        opt.zero_grad(set_to_none=True)
        fwd_loss().backward()
        opt.step()
    t0 = time.perf_counter()
    step()   # warm-up
    log(f"cpu oracle warm-up step: {time.perf_counter() - t0:.1f} s on {cores} threads")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    log(f"cpu oracle: {steps} steps of batch {batch} in {dt:.1f} s")
    return {"value": round(batch * steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the same {workload.upper()} train step at batch {batch}, fp32, "
                      f"torch {torch.__version__} CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--seq-len", type=int, default=128, help="caption length (the reference's MIBF loader pads to 256, "
                    "mibf_net/dataset_spine.py:88; the metric of BASELINE.json is quoted at 128)")
    ap.add_argument("--gemm-log", default=None, help="copy the per-launch CSV of the roofline leg to this file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-f32 mode leg")
    ap.add_argument("--no-ddp-config", action="store_true", help="skip the N=1 leg that runs the data-parallel configuration")
    args = ap.parse_args()
    global SEQ
    SEQ = args.seq_len
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
    batch, classes, train_tflop, desc = WORKLOADS[args.workload]
    if SEQ != 128:
        # SURVEY 8d counts at L = 128; BERT-base is 10.874 GMAC/sample of projections + FFN (linear in L) and 0.300 of attention
        # (quadratic): 22.951 GMAC at L = 256.  Train FLOPs = 3 x forward, 1 MAC = 2 FLOP.
        gmac = lambda L: 10.874 * (L / 128.0) + 0.300 * (L / 128.0) ** 2
        train_tflop = train_tflop + 3 * 2 * (gmac(SEQ) - gmac(128)) * batch / 1000.0
        WORKLOADS[args.workload] = (batch, classes, train_tflop, desc.replace("L=128", f"L={SEQ}") + f" [caption length {SEQ}]")
        desc = WORKLOADS[args.workload][3]
    # a line measured with work left out is invalid: the work-skipping switch of tools/knockout.sh exists only in
    # -DHS_MEASURE builds of the library, and this script refuses both the variable and such a build
    if os.environ.get("HAMSPINE_KNOCKOUT", "0") not in ("", "0"):
        sys.exit("bench.py: HAMSPINE_KNOCKOUT is set -- it removes launches from the step; refusing to produce a bench line "
                 "(use tools/knockout.sh, which times such steps without printing one)")
    from hamspine import _lib as _L
    if _L.lib().hs_measure_build() and os.environ.get("HAMSPINE_ALLOW_MEASURE_BUILD") != "1":
        sys.exit("bench.py: libhamspine_hip.so was built with -DHS_MEASURE (knockout switches compiled in); rebuild without it")
    dt, final_loss, roofline, nparams, host_ms, f32_mode, ddp_cfg = gpu_leg(args, rank, world, local_rank)
    if rank == 0:
        value = batch * world * args.steps / dt
        ms_step = dt / args.steps * 1e3
        metric = {"c2": "training images/sec, ResNet50+BERT-base 224px bs32",
                  "c3": "training images/sec, MIBF-Net (ResNet50+BERT-base, IBFA, MP-Loss) 224px bs32",
                  "c4": "training images/sec, ConNeXT (ConvNeXt-base+BERT-base) 224px bs64",
                  "c5": "training images/sec, hierarchical fusion (ResNet18 multi-scale + global-local + gate + KAN head) 224px bs32"}[args.workload]
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init weights of the real architecture)",
            "config": {"workload": desc, "per_gpu_batch": batch, "global_batch": batch * world,
                       "seq_len": SEQ, "params": nparams, "parallelism": f"dp{world}"},
            "step_mfma_frac": round(train_tflop / (ms_step * 1e-3) / MFMA_BF16_PEAK_TFLOPS, 4),
            "host_enqueue_ms_per_step": round(host_ms, 3),
            "final_loss": round(final_loss, 4),
            "roofline": roofline,
        }
        if f32_mode is not None:
            out["f32_mode"] = f32_mode
        if ddp_cfg is not None:
            out["ddp_config"] = ddp_cfg
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle (bounded sample) ...")
            out["cpu_baseline"] = cpu_leg(args.workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
