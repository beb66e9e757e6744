"""hcm_conv3x3_wgrad against ATen's convolution_backward (weight gradient of a 3x3/s1/p1 convolution).

Floating point: fp32 MFMA partial sums reduced in a fixed order; bound 2e-5 of the gradient's scale
(reduction lengths up to 131072).  Deterministic: two runs are bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(32, 18, 18, 64, 64), (8, 36, 36, 32, 32), (4, 72, 72, 16, 16), (4, 144, 144, 8, 8), (2, 32, 64, 16, 16),
          (5, 7, 7, 12, 20), (3, 18, 36, 9, 12), (2, 256, 256, 8, 8), (1, 3, 5, 4, 4)]


@pytest.mark.parametrize('shape', SHAPES)
def test_wgrad_matches_aten(shape):
    from hcmoco_amd import hip_ops
    N, C, K, H, W = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    dy = torch.randn(N, K, H, W, generator=g).to(dev)
    w = torch.zeros(K, C, 3, 3, device=dev)
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False,
                                              [0, 0], 1, [False, True, False])[1]
    got = hip_ops.conv3x3_wgrad(x, dy)
    again = hip_ops.conv3x3_wgrad(x, dy)
    assert torch.equal(got, again)
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wgrad_rejects_unsupported():
    from hcmoco_amd import hip_ops
    dev = torch.device('cuda:0')
    with pytest.raises(ValueError):
        hip_ops.conv3x3_wgrad(torch.randn(2, 4, 6, 6, device=dev), torch.randn(2, 4, 6, 6, device=dev))   # W % 4 != 0
    with pytest.raises(RuntimeError):
        hip_ops.conv3x3_wgrad(torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8))                            # CPU tensors


SHAPES_1X1 = [(32, 36, 18, 32, 32), (8, 72, 18, 16, 16), (4, 144, 36, 8, 8), (8, 64, 256, 16, 16), (3, 5, 7, 12, 20),
              (2, 18, 144, 8, 8),
              (4, 256, 64, 64, 64),      # layer1's 256 -> 64 on the 64-wide map: used to exceed LDS and fall back to MIOpen
              # the shared MLPs of PointNet++ on ball tensors [B, C, npoint, nsample]: constant-folded (W, rows) = (32, 2), (16, 2), (32, 1)
              (2, 32, 64, 300, 32), (2, 99, 64, 130, 16), (2, 64, 128, 67, 16), (2, 99, 64, 50, 32), (2, 64, 128, 33, 32)]


@pytest.mark.parametrize('shape', SHAPES_1X1)
def test_wgrad_1x1_matches_aten(shape):
    from hcmoco_amd import hip_ops
    N, C, K, H, W = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    dy = torch.randn(N, K, H, W, generator=g).to(dev)
    w = torch.zeros(K, C, 1, 1, device=dev)
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [0, 0], [1, 1], False,
                                              [0, 0], 1, [False, True, False])[1]
    got = hip_ops.conv3x3_wgrad(x, dy, ksize=1)
    assert torch.equal(got, hip_ops.conv3x3_wgrad(x, dy, ksize=1))
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


SHAPES_S2 = [(32, 18, 18, 64, 64), (8, 18, 36, 64, 64), (4, 36, 72, 32, 32), (4, 72, 144, 16, 16), (2, 5, 7, 12, 24),
             (8, 36, 36, 32, 32)]


@pytest.mark.parametrize('shape', SHAPES_S2)
def test_wgrad_3x3_stride2_matches_aten(shape):
    from hcmoco_amd import hip_ops
    N, C, K, H, W = shape                    # input map H x W, output map H/2 x W/2
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(shape) + 2)
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    dy = torch.randn(N, K, H // 2, W // 2, generator=g).to(dev)
    w = torch.zeros(K, C, 3, 3, device=dev)
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [2, 2], [1, 1], [1, 1], False,
                                              [0, 0], 1, [False, True, False])[1]
    got = hip_ops.conv3x3_wgrad(x, dy, ksize=3, stride=2)
    assert torch.equal(got, hip_ops.conv3x3_wgrad(x, dy, ksize=3, stride=2))
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_partial_plus_batched_reduction_equals_the_single_calls():
    """hcm_conv_wgrad_partial for several layers of different kinds, then ONE hcm_wgrad_reduce_batch: every dW is
    bit-identical to its hcm_conv{3x3,1x1,3x3s2}_wgrad call (same partial sums, same fixed-order reduction)."""
    import ctypes as C
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())

    class Desc(C.Structure):
        _fields_ = [('partial', C.c_void_p), ('dw', C.c_void_p), ('total', C.c_int), ('chunks', C.c_int)]

    layers = [(3, 8, 18, 18, 64, 64), (1, 8, 36, 18, 32, 32), (2, 8, 18, 36, 32, 32), (3, 8, 36, 36, 32, 32),
              (3, 4, 18, 18, 16, 16), (1, 4, 144, 72, 8, 8)] * 12                 # 72 layers: two launches of the batch kernel
    keep, descs, refs, outs = [], [], [], []
    for kind, N, Cc, K, H, W in layers:
        s = 2 if kind == 2 else 1
        x = torch.randn(N, Cc, s * H, s * W, generator=g).to(dev)
        dy = torch.randn(N, K, H, W, generator=g).to(dev)
        ks = 1 if kind == 1 else 3
        single = {3: L.hcm_conv3x3_wgrad, 1: L.hcm_conv1x1_wgrad, 2: L.hcm_conv3x3s2_wgrad}[kind]
        nbytes = {3: L.hcm_conv3x3_wgrad_workspace_bytes, 1: L.hcm_conv1x1_wgrad_workspace_bytes,
                  2: L.hcm_conv3x3s2_wgrad_workspace_bytes}[kind](N, Cc, K, H, W)
        assert nbytes > 0
        ws1, ws2 = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ref = torch.empty(K, Cc, ks, ks, device=dev)
        check(single(p(x), p(dy), N, Cc, K, H, W, p(ref), p(ws1), nbytes, st), 'single')
        out = torch.full((K, Cc, ks, ks), float('nan'), device=dev)
        chunks = C.c_int(0)
        check(L.hcm_conv_wgrad_partial(kind, p(x), p(dy), N, Cc, K, H, W, p(ws2), nbytes, C.byref(chunks), st), 'partial')
        assert chunks.value > 0
        descs.append(Desc(ws2.data_ptr(), out.data_ptr(), out.numel(), chunks.value))
        keep += [x, dy, ws1, ws2]
        refs.append(ref)
        outs.append(out)
    arr = (Desc * len(descs))(*descs)
    check(L.hcm_wgrad_reduce_batch(C.cast(arr, C.c_void_p), len(descs), st), 'reduce_batch')
    torch.cuda.synchronize()
    for ref, out in zip(refs, outs):
        assert torch.equal(ref, out)
