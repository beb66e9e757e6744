"""Process-group bring-up and learning-rate schedule.

Reference: /root/reference/pycontrast/learning/base_trainer.py:13-103.  The reference only runs
under SLURM (``SLURM_PROCID``...) and divides by ``device_count()`` (zero on a CPU box); this
version reads, in order, the torchrun variables (``RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*``), the
SLURM variables, or falls back to a single process -- one process per GPU in every case, RCCL
(``--dist-backend nccl``) over xGMI on the GPU node and gloo on CPU.
"""
import math
import os
import subprocess

import numpy as np
import torch
import torch.distributed as dist


class BaseTrainer(object):
    def __init__(self, args):
        self.args = args
        self.local_group = None
        self.logger = None
        self.device = torch.device('cpu')

    def init_ddp_environment(self, gpu, ngpus_per_node):
        args = self.args
        env = os.environ
        if 'RANK' in env and 'WORLD_SIZE' in env:                       # torchrun
            rank, world = int(env['RANK']), int(env['WORLD_SIZE'])
            local = int(env.get('LOCAL_RANK', rank))
        elif 'SLURM_PROCID' in env:                                     # reference path (:38-47)
            rank, world = int(env['SLURM_PROCID']), int(env['SLURM_NTASKS'])
            local = rank % max(1, torch.cuda.device_count())
            env.setdefault('MASTER_ADDR', subprocess.getoutput(
                'scontrol show hostname {} | head -n1'.format(env['SLURM_NODELIST'])))
            env['WORLD_SIZE'], env['RANK'] = str(world), str(rank)
        else:
            rank, world, local = 0, 1, 0
        env.setdefault('MASTER_ADDR', '127.0.0.1')
        env.setdefault('MASTER_PORT', '23456')
        if world > 1:
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC between the ranks' processes (RCCL)

        backend = args.dist_backend
        use_gpu = torch.cuda.is_available() and backend != 'gloo'
        if use_gpu:
            torch.cuda.set_device(local % torch.cuda.device_count())
            self.device = torch.device('cuda', torch.cuda.current_device())
            from .affinity import pin_to_gpu_node                       # the reference relies on SLURM's --cpu-bind
            pin_to_gpu_node(self.device.index)
            torch.backends.cudnn.benchmark = True                       # MIOpen find mode (:28)
        else:
            backend = 'gloo'
        if not dist.is_initialized():
            env['WORLD_SIZE'], env['RANK'] = str(world), str(rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

        if use_gpu and world > 1:
            # per-replica device generator (pixel sampling); weights stay identical via the CPU seed
            torch.cuda.manual_seed((torch.initial_seed() + 7919 * rank) & 0x7fffffffffffffff)
        # ranks per node: torchrun says it (LOCAL_WORLD_SIZE); the reference's SLURM launch has one task per GPU
        # (device_count).  The per-node groups below must tile the world exactly (a trailing partial node was
        # left without a group and gathered over WORLD while its peers used their node group).
        if 'LOCAL_WORLD_SIZE' in env:
            ngpus_per_node = int(env['LOCAL_WORLD_SIZE'])
        ngpus_per_node = max(1, min(ngpus_per_node, world))
        if world % ngpus_per_node != 0:
            raise ValueError('world size %d is not a multiple of the %d ranks per node' % (world, ngpus_per_node))
        args.distributed = True
        args.world_size = world
        args.rank = rank
        args.gpu = self.device.index if use_gpu else None
        args.ngpus_per_node = ngpus_per_node
        args.node_rank = rank // ngpus_per_node
        args.local_rank = local
        args.local_center = args.node_rank * ngpus_per_node
        # per-node groups exist for ShuffleBN only (MoCo path, :60-73): every rank creates every node's group (new_group
        # is collective) and keeps its own; the bank path never uses them
        self.local_group = None
        if getattr(args, 'mem', '') == 'moco' and world > 1:
            for node in range(0, max(1, world // ngpus_per_node)):
                ranks = list(range(node * ngpus_per_node, min(world, (node + 1) * ngpus_per_node)))
                group = dist.new_group(ranks)
                if rank in ranks:
                    self.local_group = group
        if rank == 0:
            print('world size {}, backend {}, device {}'.format(world, backend, self.device))

    def init_tensorboard_logger(self):
        """tensorboard_logger is optional here (absent in this image): scalars fall back to stdout."""
        if self.args.rank != 0:
            return
        try:
            import tensorboard_logger as tb_logger
            self.logger = tb_logger.Logger(logdir=self.args.tb_folder, flush_secs=2)
        except ImportError:
            self.logger = None

    def adjust_learning_rate(self, optimizer, epoch):
        """cosine or step decay (:80-92)."""
        args = self.args
        lr = args.learning_rate
        if args.cosine:
            eta_min = lr * (args.lr_decay_rate ** 3)
            lr = eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / args.epochs)) / 2
        else:
            steps = int(np.sum(epoch > np.asarray(args.lr_decay_epochs)))
            if steps > 0:
                lr = lr * (args.lr_decay_rate ** steps)
        for group in optimizer.param_groups:
            group['lr'] = lr
        return lr

    def warmup_learning_rate(self, epoch, batch_id, total_batches, optimizer):
        """linear warm-up over the first warm_epochs (:94-103)."""
        args = self.args
        if getattr(args, 'warm', False) and epoch <= args.warm_epochs:
            p = (batch_id + (epoch - 1) * total_batches) / (args.warm_epochs * total_batches)
            lr = args.warmup_from + p * (args.warmup_to - args.warmup_from)
            for group in optimizer.param_groups:
                group['lr'] = lr
