"""CPU tests of the host-side policies around the path: checkpoint loading tolerance and the top-k
policy of the reference's training driver (scripts/train.py:411-428, mibf_net/predict_resnet.py:13-23)."""
import os

import torch
import torch.nn as nn

from hamspine import checkpoint as ck


class _Wrapped(nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(4, 3), nn.ReLU(), nn.Linear(3, 2))


def test_load_checkpoint_tolerates_wrappers_and_prefixes(tmp_path, capsys):
    src = _net(1)
    dp_state = _Wrapped(src).state_dict()                       # "module.0.weight", ... as DataParallel / DDP save them
    assert all(k.startswith("module.") for k in dp_state)
    for name, payload in (("plain.pth", src.state_dict()), ("dp.pth", dp_state), ("wrapped.pth", {"state_dict": dp_state})):
        path = str(tmp_path / name)
        torch.save(payload, path)
        dst = _net(2)
        missing, unexpected = ck.load_checkpoint(dst, path)
        assert missing == [] and unexpected == []
        for a, b in zip(dst.parameters(), src.parameters()):
            assert torch.equal(a, b)
    # non-strict: reports instead of raising (reference predict_resnet.py:19-23)
    partial = {k: v for k, v in src.state_dict().items() if not k.startswith("2.")}
    partial["extra.weight"] = torch.zeros(1)
    torch.save(partial, str(tmp_path / "partial.pth"))
    missing, unexpected = ck.load_checkpoint(_net(3), str(tmp_path / "partial.pth"))
    assert missing == ["2.weight", "2.bias"] and unexpected == ["extra.weight"]
    out = capsys.readouterr().out
    assert "missing keys" in out and "unexpected keys" in out


def test_top3_checkpoint_policy(tmp_path):
    out = str(tmp_path / "ckpt")
    top = ck.TopKCheckpoints(out, k=3)
    net = _net(0)
    accs = [50.0, 61.25, 55.5, 55.5, 70.126, 40.0, 61.25]
    for epoch, acc in enumerate(accs):
        model = _Wrapped(net) if epoch % 2 else net             # wrapped models are saved without the prefix
        path = top.update(model, epoch, acc)
        if path is not None:
            assert os.path.basename(path) == f"epoch_{epoch + 1}_val_acc_{acc:.2f}.pth"
            assert list(torch.load(path).keys()) == list(net.state_dict().keys())
    # the policy walked by hand:
    #   [50] -> [61.25, 50] -> [61.25, 55.5, 50] -> 55.5 > 50: evict 50 -> [61.25, 55.5, 55.5]
    #   70.126 > 55.5: evict one 55.5 (the first minimum) -> [70.126, 61.25, 55.5]; 40 rejected;
    #   61.25 > 55.5: evict 55.5 -> [70.126, 61.25, 61.25]
    assert top.accuracies() == [70.126, 61.25, 61.25]
    files = sorted(os.listdir(out))
    assert files == ["epoch_2_val_acc_61.25.pth", "epoch_5_val_acc_70.13.pth", "epoch_7_val_acc_61.25.pth"]
    assert top.update(net, 7, 61.25) is None                    # equal to the worst: strict '>' keeps the old one
