"""Seeded synthetic inputs of the HRNetPN fixtures (no reference code): used by gen_golden.py when it records the
reference's outputs and by the tests when they replay them."""
import torch


def pn_inputs(B, size, J, seed, empty=()):
    """Synthetic NTU-shaped HRNetPN inputs: rgb ~ N(0,1); depth = 0.3 * N(0,1) metres about the
    per-sample mean on a centred disc, 0 elsewhere; grid_xy = pixel numbers of a centred 600-pixel box of a 1080 x 1920
    frame; samples listed in ``empty`` have no depth at all (use_depth = 0: zero depth, zero mask, dataset.py:574-575)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing='ij')
    disc = (((yy - size / 2) ** 2 + (xx - size / 2) ** 2) < (0.375 * size) ** 2).float()
    mask = disc.unsqueeze(0).repeat(B, 1, 1)
    for b in empty:
        mask[b] = 0
    rgb = torch.randn(B, 3, size, size, generator=g)
    depth = (0.3 * torch.randn(B, 1, size, size, generator=g) * mask.unsqueeze(1)).expand(B, 3, size, size)
    x = torch.cat([rgb, depth], 1).contiguous()
    s = torch.rand(B, J, 2, generator=g) * 2 - 1
    ys = torch.linspace(240, 840, size).round().int()
    xs = torch.linspace(660, 1260, size).round().int()
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    grid_xy = torch.stack([gy, gx], -1).unsqueeze(0).expand(B, size, size, 2).contiguous()
    mean = torch.rand(B, generator=g) * 2 + 2
    return x, s, mask, grid_xy, 1080, 1920, mean
