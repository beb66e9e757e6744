"""hcmoco_amd -- MI355X-native (gfx950) engine for the HCMoCo contrastive pre-training hot path.

Layout
  csrc/        hand-written HIP kernels + the C ABI (include/hcmoco_hip.h) -> libhcmoco_hip.so
  _lib.py      ctypes binding of the C ABI (raw device pointers, no torch types cross it)
  hip_ops.py   torch.autograd.Function wrappers over the C ABI
  pycontrast/  host-side mirror of the reference surface: main_contrast.py, options/,
               networks/, memory/, learning/

The product path never falls back to a CPU or eager implementation: every op in ``hip_ops``
raises if the shared library is missing or the tensors are not on a ROCm device.
"""
__version__ = '0.1.0'
