"""Where does run-to-run non-determinism enter one stage-2 step?  Two fresh, identically seeded models in one
process; compares bit-wise: encoder outputs, features, bank-loss gradient of the features, branch-map gradients."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hcmoco_amd import _lib
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
from hcmoco_amd.pycontrast.learning.engine import HipLossEngine

dev = torch.device('cuda:0')
plain = len(sys.argv) > 1 and sys.argv[1] == 'plain'
if plain:
    from hcmoco_amd.pycontrast.networks import hrnet
    hrnet.ENCODER_PROGRAM = False
    os.environ['HCM_ASYNC_WGRAD'] = '0'


def run():
    args = bench.make_args(8, 1024, 4096, 128, 'coco17', 'nccl', tempfile.mkdtemp(), 2)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    tr = ContrastTrainer(args)
    tr.device = dev
    model, contrast, opt, data = bench.build(args, tr, dev)
    if plain:
        model.two_streams = 0
    torch.cuda.manual_seed(99)
    batch = next(iter(data))
    x, skel = batch[0].float(), batch[2]
    rec = {}
    f1s, f2s, feat3, f, aux = model(x, skel, return_fm=True)
    for i, m in enumerate(f1s):
        rec['feat1_%d' % i] = m.detach().clone()
        m.retain_grad()
    for i, m in enumerate(f2s):
        rec['feat2_%d' % i] = m.detach().clone()
        m.retain_grad()
    rec['feat3'] = feat3.detach().clone()
    rec['f'] = f.detach().clone()
    f.retain_grad(); feat3.retain_grad()
    a, b, c = torch.chunk(f, 3, dim=1)
    eng = HipLossEngine()
    total, losses, accs = eng.bank(contrast, a, b, c, batch[1], a.detach(), b.detach(), c.detach(), batch[1], use_depth=batch[6])
    rec['bank_losses'] = losses.clone()
    fm_total, meters = eng.fmap_sampled(f1s, f2s, model.encoder1_linear, model.encoder2_linear, feat3, batch[7], batch[4],
                                        batch[5], batch[6], None, 400, 0.07)
    rec['meters'] = meters.clone()
    (total + fm_total).backward()
    _lib.torch_glue().wgrad_join()
    torch.cuda.synchronize()
    rec['g_f'] = f.grad.clone()
    rec['g_feat3'] = feat3.grad.clone()
    for i, m in enumerate(f1s):
        rec['g_feat1_%d' % i] = m.grad.clone()
    for n in ('head1.0.weight', 'head3.0.weight', 'encoder1_linear.weight', 'encoder3.gconv_output.W',
              'encoder1.stage4.2.fuse_layers.3.2.0.1.bias', 'encoder1.stage4.2.branches.3.3.conv2.weight', 'encoder1.conv1.weight'):
        p = dict(model.named_parameters())[n]
        rec['grad ' + n] = p.grad.clone()
    _lib.torch_glue().set_async_wgrad(False)
    return rec


a, b = run(), run()
for k in a:
    same = torch.equal(a[k], b[k])
    d = float((a[k].double() - b[k].double()).abs().max()) / (float(b[k].double().abs().max()) + 1e-30)
    print('%-52s %s  max rel diff %.2e' % (k, 'same' if same else 'DIFF', d))
