import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-diagnosis-ham-spine_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29741")
dev=torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.randn(64<<20, device=dev)          # 256 MB
big=torch.randn(8192,8192,device=dev)
s=torch.cuda.Stream()
dist.all_reduce(x); torch.cuda.synchronize()
for mode in ("idle","busy"):
    torch.cuda.synchronize()
    if mode=="busy":
        for _ in range(20): big @ big          # ~ tens of ms of queued work on the current stream
    ev=torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s):
        s.wait_event(ev)
        t0=time.perf_counter()
        w=dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True)
        t1=time.perf_counter()
    print(mode, "host time of all_reduce(async_op=True) behind pending work: %.3f ms" % ((t1-t0)*1e3))
    torch.cuda.synchronize()
