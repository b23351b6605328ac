"""world_size-2 gloo test of hamspine.ddp.DataParallel (the N>1 path of bench.py): rank-0 broadcast at
construction, bucketed gradient averaging, grad=None for unused parameters, buffers following rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.bn = nn.BatchNorm1d(32)
        self.b = nn.Linear(32, 8)
        self.unused = nn.Linear(8, 8)     # never called: must keep grad None (find_unused_parameters semantics)

    def forward(self, x):
        return self.b(torch.relu(self.bn(self.a(x))))


class _Branchy(nn.Module):
    """`extra` takes part only when the caller asks for it: the set of used parameters changes from step to step.  It sits
    at the END of the parameter list = the front of the (reverse-order) bucket, and its gradient is produced FIRST in
    backward (it is applied to the output), so a re-learn meets a bucket that has not been launched yet."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)
        self.extra = nn.Linear(8, 8)

    def forward(self, x, branch=False):
        y = self.b(torch.relu(self.a(x)))
        return self.extra(y) if branch else y


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "multimodal-diagnosis-ham-spine_amd"))
        from hamspine.ddp import DataParallel
        torch.manual_seed(100 + rank)            # different init per rank: the constructor must equalise it
        net = _Net()
        ddp = DataParallel(net, bucket_mb=0.001)  # tiny buckets -> several collectives
        assert len(ddp.buckets) > 1
        w = [torch.empty_like(net.a.weight) for _ in range(world)]
        dist.all_gather(w, net.a.weight.data)
        assert all(torch.equal(w[0], t) for t in w), "parameters must follow rank 0"
        for step in range(2):
            x = torch.randn(6, 16, generator=torch.Generator().manual_seed(7 * step + rank))
            # reference: local gradients averaged by hand
            ref = _Net()
            ref.load_state_dict(net.state_dict())
            ref(x).square().sum().backward()
            ddp.zero_grad()
            ddp(x).square().sum().backward()
            ddp.finish()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                if n.startswith("unused"):
                    assert p.grad is None, f"{n}: unused parameter must keep grad None"
                    continue
                g = [torch.empty_like(q.grad) for _ in range(world)]
                dist.all_gather(g, q.grad)
                want = sum(g) / world
                assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), f"{n} step {step}"
            with torch.no_grad():
                for p in net.parameters():
                    if p.grad is not None:
                        p -= 0.01 * p.grad
        # parameters stay identical across ranks after synchronous steps
        dist.all_gather(w, net.a.weight.data)
        assert all(torch.allclose(w[0], t) for t in w)
        # one big bucket that holds the never-used parameter: the first step learns the unused set (bucket reduced in
        # finish()), from the second step on the bucket completes -- and launches -- during backward
        net2 = _Net()
        big = DataParallel(net2, bucket_mb=128)
        assert len(big.buckets) == 1
        for step in range(3):
            x = torch.randn(6, 16, generator=torch.Generator().manual_seed(50 + 7 * step + rank))
            ref = _Net()
            ref.load_state_dict(net2.state_dict())
            ref(x).square().sum().backward()
            big.zero_grad()
            big(x).square().sum().backward()
            big.finish()
            assert net2.unused.weight.grad is None
            g = [torch.empty_like(ref.a.weight.grad) for _ in range(world)]
            dist.all_gather(g, ref.a.weight.grad)
            assert torch.allclose(net2.a.weight.grad, sum(g) / world, rtol=1e-5, atol=1e-6), f"big bucket step {step}"
            assert big.stats == {"launched_in_backward": step, "launched_in_finish": 1}, (step, big.stats)
        # ---- the xGMI-native exchange algorithms: all-to-all of shards + f32 sum at the owner + all-gather --------------
        for algo, tol in (("direct", 1e-6), ("direct_bf16", 1e-2)):
            net3 = _Net()
            d3 = DataParallel(net3, bucket_mb=0.001, algo=algo)
            for step in range(2):
                x = torch.randn(6, 16, generator=torch.Generator().manual_seed(90 + 7 * step + rank))
                ref = _Net()
                ref.load_state_dict(net3.state_dict())
                ref(x).square().sum().backward()
                d3.zero_grad()
                d3(x).square().sum().backward()
                d3.finish()
                assert net3.unused.weight.grad is None
                for (n, p), (_, q) in zip(net3.named_parameters(), ref.named_parameters()):
                    if n.startswith("unused"):
                        continue
                    g = [torch.empty_like(q.grad) for _ in range(world)]
                    dist.all_gather(g, q.grad)
                    want = sum(g) / world
                    err = (p.grad - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
                    assert err <= tol, f"{algo}: {n} step {step}: rel err {err:.2e}"
            # every rank ends with the same averaged gradient (bitwise: they all received the same reduced shards)
            g = [torch.empty_like(net3.a.weight.grad) for _ in range(world)]
            dist.all_gather(g, net3.a.weight.grad)
            assert all(torch.equal(g[0], t) for t in g), f"{algo}: ranks disagree on the averaged gradient"
        # ---- the used set changes after the first step (round-2 advisor finding) ---------------------------------------
        # rank 1 alone runs the optional branch in step 0: the learnt set is the UNION over ranks, so rank 0 keeps
        # contributing zeros for it and both ranks launch the same buckets at the same points
        net4 = _Branchy()
        d4 = DataParallel(net4, bucket_mb=128)
        for step, branch in enumerate([rank == 1, False, True, False]):
            x = torch.randn(6, 16, generator=torch.Generator().manual_seed(130 + 7 * step + rank))
            ref = _Branchy()
            ref.load_state_dict(net4.state_dict())
            ref(x, branch).square().sum().backward()
            d4.zero_grad()
            d4(x, branch).square().sum().backward()
            d4.finish()
            flags = [torch.zeros(1) for _ in range(world)]
            dist.all_gather(flags, torch.tensor([1.0 if branch else 0.0]))
            for (n, p), (_, q) in zip(net4.named_parameters(), ref.named_parameters()):
                loc = q.grad if q.grad is not None else torch.zeros_like(q)
                g = [torch.empty_like(loc) for _ in range(world)]
                dist.all_gather(g, loc)
                if p.grad is None:
                    assert q.grad is None, f"branchy step {step}: {n} lost its gradient"
                    continue
                assert torch.allclose(p.grad, sum(g) / world, rtol=1e-5, atol=1e-6), f"branchy {n} step {step}"
        # a model whose optional branch first runs AFTER the first step: the wrapper re-learns instead of raising as long as
        # the bucket has not been launched; with static_unused=False that is guaranteed (buckets holding such parameters
        # wait for the end of backward)
        for static in (False, True):
            net5 = _Branchy()
            d5 = DataParallel(net5, bucket_mb=128, static_unused=static)
            for step, branch in enumerate([False, True, False, True]):
                x = torch.randn(6, 16, generator=torch.Generator().manual_seed(170 + 7 * step + rank))
                ref = _Branchy()
                ref.load_state_dict(net5.state_dict())
                ref(x, branch).square().sum().backward()
                d5.zero_grad()
                d5(x, branch).square().sum().backward()
                d5.finish()
                for (n, p), (_, q) in zip(net5.named_parameters(), ref.named_parameters()):
                    if q.grad is None:
                        assert p.grad is None, f"static={static} step {step}: {n} must keep grad None"
                        continue
                    g = [torch.empty_like(q.grad) for _ in range(world)]
                    dist.all_gather(g, q.grad)
                    assert torch.allclose(p.grad, sum(g) / world, rtol=1e-5, atol=1e-6), f"static={static} {n} step {step}"
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, repr(e) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
