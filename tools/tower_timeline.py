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
    net, fwd_loss, make_opt = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c2", dev, 0)
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
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        opt.step()
        b1.record()
        return t0, a0, b0, b1
    for _ in range(6):
        marks.clear()
        t0, a0, b0, b1 = step()
    torch.cuda.synchronize()
    rows = [(n, t0.elapsed_time(e0), t0.elapsed_time(e1)) for n, e0, e1 in marks]
    rows.append(("forward enqueued -> loss ready (main stream)", 0.0, t0.elapsed_time(a0)))
    rows.append(("optimizer step", t0.elapsed_time(b0), t0.elapsed_time(b1)))
    for n, s, e in sorted(rows, key=lambda r: r[1]):
        print(f"{n:48s} {s:7.3f} -> {e:7.3f} ms  ({e - s:6.3f})")


if __name__ == "__main__":
    main()
