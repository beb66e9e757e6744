"""Isolated timing of hcm_conv3x3_forward / _backward_data against the library convolution torch dispatches
(MIOpen) on the HRNet-w18 branch shapes at B=32.  Usage (GPU box): python tools/bench_conv.py"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcmoco_amd import _lib                                  # noqa: E402


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def main():
    if len(sys.argv) > 1:                      # a diagnostic build of the library
        _lib.LIB_PATH = sys.argv[1]
    dev = torch.device('cuda:0')
    L = _lib.lib()
    for N, Cc, H, W in [(32, 18, 64, 64), (32, 36, 32, 32)]:
        x = torch.randn(N, Cc, H, W, device=dev)
        w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.1
        y = torch.empty_like(x)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        own_f = timeit(lambda: L.hcm_conv3x3_forward(p(x), p(w), p(y), N, Cc, Cc, H, W, st))
        own_b = timeit(lambda: L.hcm_conv3x3_backward_data(p(x), p(w), p(y), N, Cc, Cc, H, W, st))
        lib_f = lib_b = float('nan')
        if len(sys.argv) == 1:
            lib_f = timeit(lambda: F.conv2d(x, w, padding=1))
            lib_b = timeit(lambda: torch.ops.aten.convolution_backward(x, y, w, None, [1, 1], [1, 1], [1, 1], False,
                                                                       [0, 0], 1, [True, False, False]))
        flops = 2.0 * N * H * W * Cc * Cc * 9
        print('%2dch@%dx%d  forward own %.1f us (%.1f TFLOP/s) library %.1f us | data gradient own %.1f us library %.1f us'
              % (Cc, H, W, own_f, flops / own_f / 1e6, lib_f, own_b, lib_b))


if __name__ == '__main__':
    main()
