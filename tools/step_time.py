#!/usr/bin/env python
"""ms per step of a bench.py workload, nothing else (no JSON line): the timing leg that measurement scripts use
(tools/knockout.sh runs it on a -DHS_MEASURE build with HAMSPINE_KNOCKOUT set, where bench.py itself refuses to run)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2")
    args = ap.parse_args()
    import hamspine
    hamspine.require_device()
    hamspine.set_compute_dtype("bf16")
    dev = torch.device("cuda:0")
    net, fwd_loss, make_opt = bench.build_workload(args.workload, dev, 0)
    opt = make_opt()

    def step():
        opt.zero_grad(set_to_none=True)
        fwd_loss().backward()
        opt.step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / args.steps * 1e3:.3f}")


if __name__ == "__main__":
    main()
