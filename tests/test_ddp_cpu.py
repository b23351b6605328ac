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
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, repr(e)))
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
