"""hcm_conv3x3_forward / hcm_conv3x3_backward_data (direct fp32 MFMA convolution of the two high-resolution
HRNet branches) against torch's conv2d in float64 -- what networks/official_hrnet.py:40-70 `conv3x3` computes.
A floating-point kernel: tolerance 2e-5 of the output's scale (fp32 sums of 162 / 324 products)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _call(name, a, w, N, Cc, K, H, W):
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    out = torch.full((N, K if name.endswith('forward') else Cc, H, W), float('nan'), device=a.device)
    check(getattr(_lib.lib(), name)(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()),
                                    N, Cc, K, H, W, C.c_void_p(torch.cuda.current_stream().cuda_stream)), name)
    return out


@pytest.mark.parametrize('N,Cc,H,W', [(32, 18, 64, 64), (32, 36, 32, 32), (3, 18, 8, 64), (2, 20, 12, 64), (1, 17, 4, 64),
                                      (5, 36, 4, 32), (2, 33, 20, 32), (1, 34, 32, 32)])
def test_conv3x3_forward_and_data_gradient_match_conv2d(N, Cc, H, W):
    from hcmoco_amd import _lib
    assert _lib.lib().hcm_conv3x3_supported(Cc, Cc, H, W) == 1
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N * 1000 + Cc)
    x = torch.randn(N, Cc, H, W, generator=g).to(dev)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.1).to(dev)
    gy = torch.randn(N, Cc, H, W, generator=g).to(dev)
    y = _call('hcm_conv3x3_forward', x, w, N, Cc, Cc, H, W)
    dx = _call('hcm_conv3x3_backward_data', gy, w, N, Cc, Cc, H, W)
    xd = x.double().requires_grad_()
    yd = F.conv2d(xd, w.double(), padding=1)
    yd.backward(gy.double())
    for got, ref in ((y, yd.detach()), (dx, xd.grad)):
        assert torch.isfinite(got).all()
        scale = ref.abs().max().item()
        assert (got.double() - ref).abs().max().item() <= 2e-5 * scale


def test_conv3x3_is_deterministic_and_refuses_other_shapes():
    from hcmoco_amd import _lib
    dev = torch.device('cuda:0')
    x, w = torch.randn(4, 18, 64, 64, device=dev), torch.randn(18, 18, 3, 3, device=dev)
    a = _call('hcm_conv3x3_forward', x, w, 4, 18, 18, 64, 64)
    b = _call('hcm_conv3x3_forward', x, w, 4, 18, 18, 64, 64)
    assert torch.equal(a, b)
    L = _lib.lib()
    for shape in [(72, 72, 16, 16), (18, 36, 64, 64), (18, 18, 62, 64), (18, 18, 64, 48), (16, 16, 64, 64), (32, 32, 64, 64)]:
        assert L.hcm_conv3x3_supported(*shape) == 0
    with pytest.raises(_lib.HipError):
        _call('hcm_conv3x3_forward', torch.randn(1, 72, 16, 16, device=dev), torch.randn(72, 72, 3, 3, device=dev),
              1, 72, 72, 16, 16)
