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

    def step():
        opt.zero_grad(set_to_none=True)
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        loss = fwd_loss()
        a0 = torch.cuda.Event(enable_timing=True); a0.record()
        loss.backward()
        f0 = torch.cuda.Event(enable_timing=True); f0.record()
        if ddp is not None:
            ddp.finish()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        opt.step()
        b1.record()
        return t0, a0, b0, b1, f0
    for _ in range(6):
        marks.clear()
        t0, a0, b0, b1, f0 = step()
    torch.cuda.synchronize()
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
