"""Is the encoders' FORWARD bit-reproducible?  The stage-2 model runs four times on the same batch in train mode
(batch-norm buffers restored in between); the eight branch maps and the SemGCN output are compared bit for bit.
Environment knobs to bisect: HCM_TWO_STREAMS, HCM_CONV_KERNEL, HCM_CONV_STATS, PLAIN=1 (module path instead of the
encoder program).  Usage (GPU box): python tools/probes/determinism_forward.py [size] [batch]"""
import copy
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                             # noqa: E402


def main():
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.networks import hrnet
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if os.environ.get('PLAIN', '0') != '0':
        hrnet.ENCODER_PROGRAM = False
    dev = torch.device('cuda:0')
    args = bench.make_args(batch, 1024, 4096, size, 'coco17', 'nccl', tempfile.mkdtemp(), 8)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    tr = ContrastTrainer(args)
    tr.device = dev
    model, contrast, opt, data = bench.build(args, tr, dev)
    net = tr.unwrap(model)
    b = next(iter(data))
    state = copy.deepcopy(net.state_dict())
    outs = []
    with torch.no_grad():
        for rep in range(4):
            net.load_state_dict(state)
            f1, f2, f3, f, aux = net(b[0], b[2], return_fm=True)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in list(f1) + list(f2) + [f3]])
    names = ['enc1.b%d' % i for i in range(4)] + ['enc2.b%d' % i for i in range(4)] + ['semgcn']
    for rep in range(1, 4):
        diffs = []
        for n, a, c in zip(names, outs[0], outs[rep]):
            if not torch.equal(a, c):
                diffs.append('%s %.1e' % (n, float((a - c).abs().max() / a.abs().max())))
        print('run 0 vs run %d: %s' % (rep, 'bit-identical' if not diffs else ', '.join(diffs)), flush=True)


if __name__ == '__main__':
    main()
