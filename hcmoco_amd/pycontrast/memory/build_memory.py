"""``build_mem(opt, n_data)`` (reference: /root/reference/pycontrast/memory/build_memory.py:5-17)."""
from .mem_bank import CMCMem3
from .mem_moco import RGBMoCo, CMCMoCo


def build_mem(opt, n_data):
    if opt.mem.startswith('bank'):
        import torch
        dtype = torch.bfloat16 if getattr(opt, 'bank_dtype', 'fp32') == 'bf16' else torch.float32
        return CMCMem3(opt.feat_dim, n_data, opt.nce_k, opt.nce_t, opt.nce_m, bank_dtype=dtype)
    if opt.mem == 'moco':
        mem_func = RGBMoCo if opt.modal == 'RGB' else CMCMoCo
        return mem_func(opt.feat_dim, opt.nce_k, opt.nce_t)
    raise NotImplementedError('mem not suported: {}'.format(opt.mem))
