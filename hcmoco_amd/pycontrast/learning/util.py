"""Meters (reference: /root/reference/pycontrast/learning/util.py:6-38)."""
import torch


class AverageMeter(object):
    """Running average.  Accepts python numbers or 0-dim tensors; tensors are accumulated on the
    device and only synchronised when ``avg``/``val`` is read, so a training step issues no
    device->host copy unless something is printed (the reference syncs 13x per step)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._val = 0
        self._sum = 0
        self.count = 0

    def update(self, val, n=1):
        if isinstance(val, torch.Tensor):
            val = val.detach()
        self._val = val
        self._sum = self._sum + val * n
        self.count += n

    @staticmethod
    def _num(v):
        return float(v.item()) if isinstance(v, torch.Tensor) else float(v)

    @property
    def val(self):
        return self._num(self._val)

    @property
    def sum(self):
        return self._num(self._sum)

    @property
    def avg(self):
        return self._num(self._sum) / max(self.count, 1)


def accuracy(output, target, topk=(1,)):
    """top-k accuracy in percent (API-mode logits only; the fused kernel reports it itself)."""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]
