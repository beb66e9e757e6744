"""Extract one encoder from a pre-training checkpoint for downstream use.

Same job as the reference's ``transfer_ckpt.py`` / ``transfer_ckpt_depth.py``
(/root/reference/pycontrast/transfer_ckpt.py): keep the tensors under ``module.encoder1.`` (RGB) or
``module.encoder2.`` (depth) and drop the prefix, so the file loads straight into an HRNet backbone.

    python -m hcmoco_amd.pycontrast.transfer_ckpt current.pth rgb_backbone.pth --encoder 1
"""
import argparse

import torch


def extract_encoder(state, which=1):
    prefix = 'module.encoder{}.'.format(which)
    return {k[len(prefix):]: v for k, v in state['model'].items() if k.startswith(prefix)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('src')
    ap.add_argument('dst')
    ap.add_argument('--encoder', type=int, default=1, choices=[1, 2])
    a = ap.parse_args(argv)
    out = extract_encoder(torch.load(a.src, map_location='cpu'), a.encoder)
    if not out:
        raise SystemExit('no tensors under module.encoder{}.'.format(a.encoder))
    torch.save(out, a.dst)
    print('wrote {} tensors to {}'.format(len(out), a.dst))


if __name__ == '__main__':
    main()
