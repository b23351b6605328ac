"""Host -> device input staging, the step right before the path.

The reference's loop (scripts/train.py:353-360,387-389) issues five synchronous `.to(device)` copies from pageable
memory per step and reads `loss.item()` twice, so the host stalls on the GPU every batch.  `BatchStager` wraps any
iterable of batches: each tensor goes through a reusable pinned buffer and an asynchronous copy on a dedicated copy
stream, one batch ahead of the consumer, and the consumer's stream waits on an event instead of the host.  Decoded
`uint8` HWC images are normalised on the device (`hs_stage_images_u8`: ToTensor + Normalize fused with the NHWC->NCHW
transpose) so only 1/4 of the bytes cross PCIe.  `LossMeter` keeps the running loss on the device and synchronises
once per epoch.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import rt

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_u8(images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """(B, H, W, 3) uint8 on the device -> (B, 3, H, W) f32, (v/255 - mean) / std per channel."""
    rt.need_gpu(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise TypeError("normalize_u8: expected a (B, H, W, 3) uint8 tensor")
    x = images_u8.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    L.check(L.lib().hs_stage_images_u8(rt.p(x), rt.p(out), B, H, W, m, s, rt.stream()), "hs_stage_images_u8")
    return out


def _host_copy(dst, src):
    """one plain memcpy into the pinned buffer.  torch's `copy_` goes through the intra-op thread pool; on a host that
    shows 128 hardware threads to a 16-core share, waking that pool between steps cost 20-35 ms per 19 MB batch
    (measured, tools/inference_bench.py), against 0.4 ms for the memcpy."""
    np.copyto(dst.reshape(-1).view(torch.uint8).numpy(), src.contiguous().reshape(-1).view(torch.uint8).numpy())


class BatchStager:
    """Iterate `loader`, yielding batches whose tensors already live on `device`.

    * tensors keep their position in the batch tuple / list / dict; non-tensors (image ids, None) pass through;
    * `prefetch` batches are in flight on the copy stream while the previous one is consumed;
    * a uint8 (B, H, W, 3) tensor at position / key `u8_images` is normalised on the device into f32 NCHW.
    """

    def __init__(self, loader, device="cuda", prefetch=1, u8_images=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.loader = loader
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.HamspineError("BatchStager stages onto an MI355X; there is no host path")
        self.prefetch = max(1, int(prefetch))
        self.u8_images = u8_images
        self.mean, self.std = tuple(mean), tuple(std)
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = {}      # (slot, key) -> pinned host buffer, reused while the shape / dtype repeat

    def __len__(self):
        return len(self.loader)

    def _stage_tensor(self, slot, key, t):
        if t.is_cuda:
            return t
        if not t.is_pinned():
            buf = self._pinned.get((slot, key))
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                self._pinned[(slot, key)] = buf
            _host_copy(buf, t)
            t = buf
        return t.to(self.device, non_blocking=True)

    def _stage(self, slot, batch):
        items = batch.items() if isinstance(batch, dict) else enumerate(batch)
        with torch.cuda.stream(self._copy_stream):
            out = {k: (self._stage_tensor(slot, k, v) if torch.is_tensor(v) else v) for k, v in items}
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        return out, ready, type(batch)

    def _finish(self, staged):
        out, ready, kind = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for k, v in out.items():
            if torch.is_tensor(v):
                v.record_stream(cur)
        if self.u8_images is not None and self.u8_images in out:
            out[self.u8_images] = normalize_u8(out[self.u8_images], self.mean, self.std)
        if kind is dict:
            return out
        seq = [out[i] for i in range(len(out))]
        return tuple(seq) if kind is tuple else seq

    def __iter__(self):
        it = iter(self.loader)
        inflight = []
        slot = 0
        # a pinned buffer is rewritten only after the batch that used it was handed over `prefetch + 1` turns ago and
        # its copy event was waited on, so prefetch + 1 buffer slots are enough
        nslots = self.prefetch + 1
        try:
            for batch in it:
                if len(inflight) == self.prefetch:
                    head = inflight.pop(0)
                    head[1].synchronize()          # the copy (not the compute) of the oldest batch: its slot is reused next
                    yield self._finish(head)
                inflight.append(self._stage(slot, batch))
                slot = (slot + 1) % nslots
            while inflight:
                yield self._finish(inflight.pop(0))
        finally:
            inflight.clear()


class LossMeter:
    """Running sum of scalar losses kept on the device: `add` enqueues one tiny kernel and never synchronises;
    `mean()` / `total()` read back once (the reference calls loss.item() twice per step, scripts/train.py:387-389)."""

    def __init__(self, device="cuda"):
        self._acc = torch.zeros(1, dtype=torch.float32, device=device)
        self.count = 0

    def add(self, loss):
        x = loss.detach().reshape(1)
        if x.dtype != torch.float32:
            raise TypeError("LossMeter.add: losses are f32 on this path")
        rt.need_gpu(x)
        L.check(L.lib().hs_axpby(L.HS_F32, L.HS_F32, rt.p(x), rt.p(self._acc), rt.p(self._acc), 1, 1.0, 1.0, rt.stream()),
                "hs_axpby")
        self.count += 1

    def total(self):
        return float(self._acc.item())

    def mean(self):
        return self.total() / max(self.count, 1)

    def reset(self):
        self._acc.zero_()
        self.count = 0
