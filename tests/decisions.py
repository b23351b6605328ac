"""Recording / forcing the ReLU and max-pool DECISIONS of a forward pass (test infrastructure).

A ReLU output that is within f32 rounding of zero, or two max-pool candidates that tie to rounding, can come out differently
in two evaluations of the same network; the element's gradient is then switched on in one and off in the other.  Forcing
the decisions of one evaluation into another removes exactly that effect, so what remains is plain rounding.
"""
import contextlib

import torch  # noqa: F401


class Decisions:
    """ReLU masks and max-pool arg-max indices of one forward, in call order."""

    def __init__(self):
        self.relu, self.pool = [], []


@contextlib.contextmanager
def decisions(store, mode):
    """mode 'record': run normally and store every ReLU mask / max-pool index; mode 'force': ignore the signs / maxima of
    this run and apply the stored decisions (y = x * mask, y = x[argmax])."""
    import torch.nn.functional as F
    orig_relu, orig_pool = F.relu, F.max_pool2d
    pos = {"r": 0, "p": 0}

    def relu(x, inplace=False):
        if mode == "record":
            y = orig_relu(x)
            store.relu.append(y > 0)
            return y
        m = store.relu[pos["r"]] if pos["r"] < len(store.relu) else None
        pos["r"] += 1
        if m is None:                       # no decision stored for this call: decide from this run's own values
            return orig_relu(x)
        return x * m.to(x.dtype)

    def pool(x, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False, return_indices=False):
        if mode == "record":
            y, idx = orig_pool(x, kernel_size, stride, padding, dilation, ceil_mode, True)
            store.pool.append(idx)
            return y
        idx = store.pool[pos["p"]] if pos["p"] < len(store.pool) else None
        pos["p"] += 1
        if idx is None:
            return orig_pool(x, kernel_size, stride, padding, dilation, ceil_mode, False)
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)

    F.relu, F.max_pool2d = relu, pool
    try:
        yield
    finally:
        F.relu, F.max_pool2d = orig_relu, orig_pool


