"""Small shared layers (reference: networks/util.py:74-80)."""
import torch.nn as nn
import torch.nn.functional as F


class Normalize(nn.Module):
    def __init__(self, p=2):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F.normalize(x, p=self.p, dim=1)
