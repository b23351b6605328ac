"""Where the towers sit in an un-profiled C2 step: HIP events recorded on the launching stream around the four tower calls
(hs_resnet_fwd / _bwd, hs_bert_fwd / _bwd) and the optimizer step, times relative to the step's first event.
python tools/tower_timeline.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    import hamspine
    from hamspine import _lib as L
    hamspine.require_device()
    hamspine.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    net, fwd_loss, make_opt = bench.build_workload(args[0] if args else "c2", dev, 0)
    ddp = None
    if "--pre" in sys.argv:          # like bench.py: the model trains without a wrapper first
        opt0 = make_opt()
        for _ in range(10):
            opt0.zero_grad(set_to_none=True)
            fwd_loss().backward()
            opt0.step()
        torch.cuda.synchronize()
        del opt0
    if "--ddp" in sys.argv:          # the N > 1 configuration on a one-rank RCCL group (bench.py's ddp_config leg)
        import torch.distributed as dist
        from hamspine.ddp import DataParallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29733")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        os.environ["HAMSPINE_DDP_SINGLE"] = "1"
        ddp = DataParallel(net, algo=os.environ.get("HAMSPINE_DDP_ALGO", "allreduce"))
    opt = make_opt()
    lib = L.lib()
    marks = []

    def wrap(name):
        fn = getattr(lib, name)

        def call(*a):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = fn(*a)
            e1.record(s)
            marks.append((name, e0, e1))
            return r
        setattr(lib, name, call)
    for n in ("hs_resnet_fwd", "hs_resnet_bwd", "hs_bert_fwd", "hs_bert_bwd"):
        wrap(n)

    import time
    host = {"zero": 0.0, "fwd": 0.0, "bwd": 0.0, "finish": 0.0, "opt": 0.0}

    def step():
        h0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        host["zero"] += time.perf_counter() - h0
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        h1 = time.perf_counter()
        loss = fwd_loss()
        a0 = torch.cuda.Event(enable_timing=True); a0.record()
        h2 = time.perf_counter()
        loss.backward()
        h3 = time.perf_counter()
        f0 = torch.cuda.Event(enable_timing=True); f0.record()
        if ddp is not None:
            ddp.finish()
        h4 = time.perf_counter()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        opt.step()
        b1.record()
        h5 = time.perf_counter()
        host["fwd"] += h2 - h1; host["bwd"] += h3 - h2; host["finish"] += h4 - h3; host["opt"] += h5 - h4
        return t0, a0, b0, b1, f0
    nsteps = int(os.environ.get("HS_TL_STEPS", "6"))
    for it in range(nsteps):
        if it == nsteps // 2:
            torch.cuda.synchronize()
            for k in host:
                host[k] = 0.0
            w0 = time.perf_counter()
        marks.clear()
        t0, a0, b0, b1, f0 = step()
    torch.cuda.synchronize()
    nw = nsteps - nsteps // 2
    print(f"wall clock over the last {nw} steps: {(time.perf_counter() - w0) / nw * 1e3:.3f} ms/step; host time per step (ms): " +
          ", ".join(f"{k} {v / nw * 1e3:.2f}" for k, v in host.items()))
    rows = [(n, t0.elapsed_time(e0), t0.elapsed_time(e1)) for n, e0, e1 in marks]
    rows.append(("forward enqueued -> loss ready (main stream)", 0.0, t0.elapsed_time(a0)))
    rows.append(("backward enqueued (main stream) -> finish() done", t0.elapsed_time(f0), t0.elapsed_time(b0)))
    rows.append(("optimizer step", t0.elapsed_time(b0), t0.elapsed_time(b1)))
    if ddp is not None:
        print("buckets (MB, launch order):", [round(b.flat.numel() * 4 / 2**20, 1) for b in ddp.buckets], "exchanges started in backward / in finish():", ddp.stats)
    for n, s, e in sorted(rows, key=lambda r: r[1]):
        print(f"{n:48s} {s:7.3f} -> {e:7.3f} ms  ({e - s:6.3f})")


if __name__ == "__main__":
    main()
