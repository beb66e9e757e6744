#!/usr/bin/env python3
"""Benchmark of the HCMoCo contrastive pre-training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
            --master-port P bench.py --gpus N --steps K --warmup W)

A "step" = one full second-stage training step (BASELINE.json configs[1]) on a synthetic batch
already resident in HBM: 2x HRNet-w18 + SemGCN forward, packed feature/index all-gather, fused
memory-bank NCE (K=16384 negatives per sample out of three 131072x128 banks), dense
intra-sample soft-InfoNCE (S=400), joint<->graph InfoNCE and cross-subject SCL (J=17), backward,
SGD step, momentum bank update.  fp32 throughout (the reference's arithmetic).  Rank 0 prints ONE
JSON line; `value` is whole-job samples/s = B*N*K / max-over-ranks(time of K steps).

roofline    : the dominant hand-written kernel (the bank gather pass, HBM-bound).  `achieved` =
              algorithmic bytes per launch / mean launch duration measured with hipEvents placed
              around that kernel, on its stream, INSIDE the timed steps (hcm_prof_enable/read).
roofline_secondary: the MFMA-bound loss kernels (dense soft-InfoNCE, cross-subject SCL: algorithmic flops of
              SURVEY 8d / launch duration against the fp32 -- or bf16 -- MFMA peak) and the SemGCN layer
              kernels (latency class; bytes / duration reported for scale), timed the same way.
cpu_baseline: the same training step on the host CPU (model in torch-CPU, losses by the oracle =
              a port of the reference math), rank 0 / N=1 only, in CPU-only subprocesses with hard
              timeouts: (a) all host threads at the bench's OWN batch (32): 1 warm-up + as many timed steps
              as fit ~25 s (SURVEY 8d asks 3 + 10 steps; at ~5 s per step that is minutes, so the sample
              is bounded and its size reported); (b) ONE thread at batch 4 for a per-core figure.  A
              reported baseline, never the thing measured.
              Row 8's three kernels (sampled merge + projection, its weight gradient, the branch-map gradients) carry both
              their algorithmic flops and bytes.  NOTE: the hipEvent pairs behind `roofline` / `roofline_secondary`
              (hcm_prof_enable) are recorded INSIDE the timed loop, on the kernels' own streams -- ~20 event records per
              step; `ms_per_step` and `ms_per_step_hipevent` agree to microseconds with and without them.
ms_per_step : wall clock (perf_counter around K steps, barrier + synchronize on both sides); the
              hipEvent-timed duration of the same K steps on the main stream is `ms_per_step_hipevent`.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFS = 157.3      # fp32-input MFMA = the fp32 vector rate (same guide)
MFMA_BF16_PEAK_TFS = 2500.0    # dense bf16 MFMA peak


def make_args(batch_global, nce_k, n_data, size, skeleton, backend, tmp, steps, sampled=1, arch='HRNet', width=18,
              bank_dtype='fp32', fmap_dtype='fp32', encoder_dtype='fp32', wgrad_stream=8):
    from hcmoco_amd.pycontrast.options.train_options import TrainOptions
    argv = ['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', arch, '--width', str(width),
            '--bank_dtype', bank_dtype, '--fmap_dtype', fmap_dtype, '--encoder_dtype', encoder_dtype,
            '--in_channel_list', '3,3', '--linear_feat_map', '1', '--modality_missing', '1',
            '--pri3d_num_samples_per_image', '400', '--temperature', '0.07', '--nce_k', str(nce_k),
            '--nce_m', '0.5', '--batch_size', str(batch_global), '--skeleton_meta_name', skeleton,
            '--learning_rate', '0.03', '--dist-backend', backend, '--synthetic',
            '--synthetic_n_data', str(n_data), '--synthetic_size', str(size), '--synthetic_steps', str(steps),
            '--model_path', tmp, '--tb_path', tmp, '--seed', '0', '--print_freq', '1000000',
            '--sampled_projection', str(sampled), '--wgrad_stream', str(wgrad_stream)]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return TrainOptions().parse(argv)


def build(args, trainer, engine_device):
    from hcmoco_amd.pycontrast.networks.build_backbone import build_model
    from hcmoco_amd.pycontrast.memory.build_memory import build_mem
    from hcmoco_amd.pycontrast.datasets.synthetic import build_synthetic_contrast_loader
    torch.manual_seed(0)
    model, _ = build_model(args)
    data, loader, _ = build_synthetic_contrast_loader(args, engine_device, args.rank, args.world_size)
    contrast = build_mem(args, len(data))
    contrast.to(engine_device)
    model.to(engine_device)
    # same update rule as the reference's torch.optim.SGD; `fused` applies it to all ~1000 parameter
    # tensors in a handful of multi-tensor launches (6.9 -> ~1 ms of host time per step)
    opt = torch.optim.SGD(model.parameters(), lr=args.learning_rate, momentum=args.momentum,
                          weight_decay=args.weight_decay, fused=torch.device(engine_device).type == 'cuda')
    model, _, opt = trainer.wrap_up(model, None, opt)
    trainer.broadcast_memory(contrast)
    model.train()
    return model, contrast, opt, data


def cpu_baseline_worker(nce_k, n_data, size, skeleton, batch, budget_s, max_steps, warmup_steps=3):
    """Runs in a fresh CPU-only process (see cpu_baseline): same step, same config, oracle losses."""
    import tempfile
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from oracle.oracle_engine import OracleLossEngine       # oracle = checker/baseline only
    threads = torch.get_num_threads()
    args = make_args(batch, nce_k, n_data, size, skeleton, 'gloo', tempfile.mkdtemp(), 64)
    args.rank, args.world_size, args.local_rank, args.channels_last = 0, 1, 0, False
    trainer = ContrastTrainer(args, engine=OracleLossEngine())
    trainer.device = torch.device('cpu')
    model, contrast, opt, data = build(args, trainer, 'cpu')
    it = iter(data)
    # SURVEY 8d asks for 3 warm-up + 10 timed steps; both are attempted inside the time budget (a step is ~5 s at batch
    # 32), and the line says how many were actually taken
    t0 = time.perf_counter()
    warmups = 0
    while warmups < 1 or (warmups < warmup_steps and (time.perf_counter() - t0) * (1 + 1.0 / warmups) < 0.25 * budget_s):
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        warmups += 1
    warm = time.perf_counter() - t0
    steps, t0 = 0, time.perf_counter()
    while steps < 1 or (time.perf_counter() - t0 + warm + (time.perf_counter() - t0) / steps < budget_s
                        and steps < max_steps):
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': round(batch * steps / dt, 3), 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'ms_per_step': round(1e3 * dt / steps, 1), 'batch': batch, 'timed_steps': steps, 'warmup_steps': warmups,
            'sample': '%d timed step(s) after %d warm-up of the same stage-2 step at batch %d, K=%d, %dx%d, '
                      'torch-CPU model + oracle losses, %d thread(s)' % (steps, warmups, batch, nce_k, size, size, threads)}


def check_step(records_path, timeout_s=600):
    """The recorded step against the oracle, in a CPU-only subprocess (``python -m oracle.check_step``): this
    process never imports the oracle.  -> the checker's report ({'checked': True, max errors ...})."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='16', MKL_NUM_THREADS='16', HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', PYTHONPATH=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'HCM_FORCE_COLLECTIVES'):
        env.pop(k, None)
    try:
        res = subprocess.run([sys.executable, '-m', 'oracle.check_step', records_path], capture_output=True, text=True,
                             env=env, timeout=timeout_s, cwd=ROOT)
        for line in reversed(res.stdout.splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {'checked': False, 'error': (res.stderr.strip().splitlines() or ['checker printed nothing'])[-1][:300]}
    except subprocess.TimeoutExpired:
        return {'checked': False, 'error': 'checker exceeded %d s' % timeout_s}
    finally:
        try:
            os.remove(records_path)
        except OSError:
            pass


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_leg(nce_k, n_data, size, skeleton, batch, threads, budget_s, max_steps, timeout_s):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'HCM_FORCE_COLLECTIVES'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu_baseline_worker', '--nce_k', str(nce_k), '--n_data',
           str(n_data), '--size', str(size), '--skeleton', skeleton, '--batch_per_gpu', str(batch),
           '--cpu_budget_s', str(budget_s), '--cpu_max_steps', str(max_steps)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout_s)
        for line in reversed(res.stdout.splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return {'value': None, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
                'sample': 'cpu leg failed: ' + (res.stderr.strip().splitlines() or ['?'])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
                'sample': 'cpu leg exceeded %d s and was stopped' % timeout_s}


def cpu_baseline(nce_k, n_data, size, skeleton, batch):
    """Bounded CPU legs in subprocesses (clean thread pools, hard timeouts) -> dict."""
    # 16 threads: the fastest setting measured on the GPU box's host (2 x EPYC 9575F, 256 hardware threads) for this
    # step at batch 32 -- 8 threads 4.03, 16 threads 4.75, 32 threads 4.37, 64 threads 2.1 samples/s: HRNet's
    # small convolutions do not scale across sockets in torch-CPU
    threads = max(1, min(os.cpu_count() or 1, 16))
    out = _cpu_leg(nce_k, n_data, size, skeleton, batch, threads, budget_s=75.0, max_steps=10, timeout_s=400)
    out['cpu_model'] = cpu_model()
    out['host_threads_available'] = os.cpu_count()
    one = _cpu_leg(nce_k, n_data, size, skeleton, 4, 1, budget_s=20.0, max_steps=3, timeout_s=240)
    out['one_thread'] = {k: one.get(k) for k in ('value', 'unit', 'cores', 'ms_per_step', 'batch', 'timed_steps', 'sample')}
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def rccl_report(max_lines=12):
    """What RCCL itself said about this job (NCCL_DEBUG=INFO, subsystems INIT + GRAPH, written to NCCL_DEBUG_FILE by this
    process): the version banner and the first ring / tree / transport lines, so that the line of the first real N > 1 run
    says which topology its numbers were measured on.  {'version': ..., 'lines': [...]} or a reason."""
    out = {'version': None, 'lines': [], 'nccl_debug': os.environ.get('NCCL_DEBUG')}
    try:
        out['version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                    # noqa: BLE001
        out['version'] = 'unavailable: %s' % e
    pat = os.environ.get('NCCL_DEBUG_FILE')
    if not pat:
        return out
    import socket
    path = pat.replace('%h', socket.gethostname()).replace('%p', str(os.getpid()))
    try:
        keep = ('NCCL version', 'RCCL version', 'Ring 00', 'Channel 00', 'Trees', 'nranks', 'via', 'XGMI', 'xgmi', 'P2P',
                'Connected all', 'channels')
        with open(path, errors='replace') as f:
            for ln in f:
                ln = ln.strip()
                if any(k in ln for k in keep) and len(out['lines']) < max_lines:
                    out['lines'].append(ln[-220:])
    except OSError as e:
        out['lines'] = ['no RCCL log at %s: %s' % (path, e)]
    return out


class FailSafe(object):
    """The N > 1 line must not fail silently: whatever goes wrong on ANY rank, rank 0 still prints ONE
    JSON line -- the usual keys, ``value`` null, ``error`` and ``phase`` saying what happened where, and the ``comm`` block
    with what was known by then.  Three channels, all watched by a daemon thread of rank 0 (the main thread may be blocked
    inside a collective; torch releases the GIL there):
      * a rank that catches an exception writes it under ``hcm_bench_error/<rank>`` into the rendez-vous TCPStore;
      * a rank that dies without a word makes the launcher tear the job down with SIGTERM: ``signal.set_wakeup_fd``
        hands the signal number to the thread at C level, whatever the main thread is doing;
      * a phase that makes no progress for ``limit`` seconds (a mis-wired rank: the peers sit in a collective) -- chosen
        BELOW the process group's own timeout (120 s), whose RCCL watchdog aborts the process without unwinding Python."""
    PG_TIMEOUT_S = 120
    STALL_S = {'default': 100.0, 'build': 400.0, 'first steps': 400.0, 'warmup': 400.0}   # --no_check: the cold start is in warmup

    def __init__(self, rank, world, header):
        import threading
        self.rank, self.world, self.header = rank, world, dict(header)
        self.phase, self.t_phase = 'start', time.monotonic()
        self.comm = {}
        self.lock = threading.Lock()
        self.store_lock = threading.Lock()      # one TCPStore client, two threads
        self.printed = False
        self.store = None
        self.armed = world > 1
        self._stop = False
        if rank == 0 and self.armed:
            import signal
            import socket
            self._rd, self._wr = socket.socketpair()
            self._rd.setblocking(False)
            self._wr.setblocking(False)
            for sig in (signal.SIGTERM, signal.SIGINT):
                signal.signal(sig, lambda *_: None)          # a Python-level handler makes the C handler feed the fd
            signal.set_wakeup_fd(self._wr.fileno(), warn_on_full_buffer=False)
            threading.Thread(target=self._watch, name='bench-failsafe', daemon=True).start()

    def enter(self, phase):
        self.phase, self.t_phase = phase, time.monotonic()

    def connect_store(self):
        """A client connection of our own to the rendez-vous store (the process group's is not shared across threads)."""
        try:
            from datetime import timedelta
            self.store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), is_master=False,
                                       timeout=timedelta(seconds=10))
        except Exception as e:                # noqa: BLE001 -- the other two channels still work
            self.comm['failsafe_store'] = 'unavailable: %s' % e

    def done(self):
        self._stop = True

    def error_line(self, msg):
        out = dict(self.header)
        out.update(value=None, error=str(msg)[:2000], phase=self.phase, comm=self.comm or None, checked=False,
                   roofline=None, cpu_baseline=None)
        return out

    def emit(self, msg):
        """Print the error line once (rank 0)."""
        with self.lock:
            if self.printed:
                return
            self.printed = True
            import ctypes
            sys.stdout.flush()
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:                 # noqa: BLE001
                pass
            print(json.dumps(self.error_line(msg)), flush=True)

    def report(self, msg):
        """Called by the rank that caught an exception."""
        if self.rank == 0:
            # a peer's failure usually reaches rank 0 twice -- as the peer's note in the store and as a transport error
            # of rank 0's own next collective -- in either order: give the note a moment, the cause goes first in the line
            if self.store is not None and self.world > 1:
                t_end = time.monotonic() + 3.0
                while time.monotonic() < t_end and not self.printed:
                    try:
                        with self.store_lock:
                            hit = [k for k in ('hcm_bench_error/%d' % r_ for r_ in range(1, self.world)) if self.store.check([k])]
                            note = self.store.get(hit[0]).decode(errors='replace') if hit else None
                    except Exception:         # noqa: BLE001
                        break
                    if hit:
                        msg = '%s: %s || rank 0 then saw: %s' % (hit[0], note, msg)
                        break
                    time.sleep(0.1)
            self.emit(msg)
            return
        if self.store is not None:
            try:
                self.store.set('hcm_bench_error/%d' % self.rank, str(msg)[:2000])
            except Exception:                 # noqa: BLE001
                pass

    def _watch(self):
        import select
        while not self._stop:
            r, _, _ = select.select([self._rd], [], [], 0.5)
            if self._stop:
                return
            if r:
                try:
                    sigs = list(self._rd.recv(64))
                except OSError:
                    sigs = []
                if sigs:
                    self.emit('rank 0 received signal %s in phase %r: the launcher tore the job down (a peer rank died?)'
                              % (sigs, self.phase))
                    os._exit(3)
            if self.store is not None:
                try:
                    keys = ['hcm_bench_error/%d' % r_ for r_ in range(1, self.world)]
                    with self.store_lock:
                        hit = [k for k in keys if self.store.check([k])]
                        note = self.store.get(hit[0]).decode(errors='replace') if hit else None
                    if hit:
                        self.emit('%s: %s' % (hit[0], note))
                        os._exit(3)
                except Exception:             # noqa: BLE001 -- store gone: the launcher's SIGTERM follows
                    pass
            limit = self.STALL_S.get(self.phase, self.STALL_S['default'])
            if time.monotonic() - self.t_phase > limit:
                self.emit('no progress for %.0f s in phase %r (a peer rank is not answering a collective?)'
                          % (limit, self.phase))
                os._exit(4)


def self_launch(n):
    """Re-run this command line under torch.distributed.run with n ranks on this node (rendez-vous on
    127.0.0.1, a free port).  The ranks' stdout is relayed as it comes, except that rank 0's JSON line is held
    back and printed LAST, so the caller reads the same single line a torchrun launch would give it."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '8')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs between the ranks' processes here
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is not None:
        sys.stdout.write(line_json)
        sys.stdout.flush()
    return rc if rc else (0 if line_json is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch_per_gpu', type=int, default=32)
    ap.add_argument('--nce_k', type=int, default=16384)
    ap.add_argument('--n_data', type=int, default=131072)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--skeleton', type=str, default='coco17')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_check', action='store_true',
                    help='skip the whole-step oracle check (one extra UNTIMED step whose loss-kernel inputs/outputs are '
                         're-evaluated by oracle/check_step.py in a CPU-only subprocess)')
    ap.add_argument('--cpu_baseline_worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu_budget_s', type=float, default=20.0)
    ap.add_argument('--cpu_max_steps', type=int, default=10, help=argparse.SUPPRESS)
    ap.add_argument('--channels_last', type=int, default=0, help=argparse.SUPPRESS)     # r01 experiment; refused with the encoder runtime
    ap.add_argument('--miopen_find', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--arch', type=str, default='HRNet', choices=['HRNet', 'HRNetPN'],
                    help='HRNetPN = BASELINE config 4 (PointNet++ depth encoder); not the headline config')
    ap.add_argument('--width', type=int, default=18, choices=[18, 32, 48])
    ap.add_argument('--bank_dtype', type=str, default='fp32', choices=['fp32', 'bf16'],
                    help='bf16 = BASELINE config 5 bank storage; not the headline config')
    ap.add_argument('--fmap_dtype', type=str, default='fp32', choices=['fp32', 'bf16', 'fp32_exact'],
                    help='bf16 = BASELINE config 5 feature-map GEMMs; not the headline config')
    ap.add_argument('--encoder_dtype', type=str, default='fp32', choices=['fp32', 'bf16'],
                    help='bf16 = bf16 encoder convolutions under autocast (fp32 batch-norm statistics / master weights / loss '
                         'section).  CORRECT, NOT ACCELERATED: it takes the stock module path instead of the encoder '
                         'programs and is SLOWER than fp32 (465.6 vs 739.4 samples/s, profiles/r05_secondary_configs.log); '
                         'BASELINE config 5 as SURVEY 8d words it is --bank_dtype bf16 --fmap_dtype bf16.  Not the headline '
                         'config')
    ap.add_argument('--sampled_projection', type=int, default=1,
                    help='project the feature maps only at the sampled pixels (SURVEY 8f-1)')
    ap.add_argument('--wgrad_stream', type=int, default=8,
                    help='layers per hand-over of the encoders\' weight gradients to their side stream (0: in line)')
    ap.add_argument('--row8_channels_last', type=int, default=0, help=argparse.SUPPRESS)   # A/B of hcm_project_rows_cl (r06)
    ap.add_argument('--fault', type=str, default=None, help=argparse.SUPPRESS)      # tests: "rank:step:exit|raise|hang"
    ap.add_argument('--backend', type=str, default='nccl',
                    help='process-group backend; nccl (= RCCL over xGMI) is the product, gloo only lets the '
                         'multi-rank control flow be exercised on a single-GPU box')
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(a.nce_k, a.n_data, a.size, a.skeleton, a.batch_per_gpu, a.cpu_budget_s,
                                             a.cpu_max_steps)))
        return

    if a.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU, like the reference's
        # scripts start one task per GPU (scripts/SecondStage/train_ntumpiirgbd2s_hrnet_w18.sh:8-14,
        # learning/base_trainer.py:38-47)
        raise SystemExit(self_launch(a.gpus))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    hsa_ipc_inherited = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')
    if world > 1:
        # dmabuf IPC between the ranks' processes (the image exports it already; hipIpcGetMemHandle fails without it on this
        # host driver).  Must be in place before the HIP runtime comes up, i.e. it cannot be retried in-process: the value
        # used and where it came from go into the line's `comm` block.
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if a.backend == 'nccl':
            # RCCL's own account of the job (version, rings / trees, transports) into a per-rank file, INIT + GRAPH only:
            # nothing is logged per collective.  Rank 0 quotes the topology lines in `comm`.
            os.environ.setdefault('NCCL_DEBUG', 'INFO')
            os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH')
            os.environ.setdefault('NCCL_DEBUG_FILE', os.path.join(__import__('tempfile').gettempdir(), 'hcm_rccl_%h_%p.log'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    header = {'metric': 'pretrain samples/sec (RGB+depth+kpt triples) HRNet-w18', 'value': None, 'unit': 'samples/s',
              'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'higher_is_better': True, 'scaling': 'weak',
              'vs_baseline': None, 'data': 'synthetic'}
    fs = FailSafe(rank, world, header)
    fs.comm.update(world_size=world, rank=rank, backend=a.backend, failsafe_armed=fs.armed,
                   hsa_enable_ipc_mode_legacy={'value': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                                               'source': 'inherited' if hsa_ipc_inherited is not None else 'set by bench.py'})
    try:
        run(a, rank, world, local, fs)
    except BaseException as e:            # noqa: BLE001 -- SystemExit with a message included: the line must still come out
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        msg = '%s: %s' % (type(e).__name__, e)
        tb = traceback.format_exc().strip().splitlines()
        fs.report('rank %d, phase %r: %s | %s' % (rank, fs.phase, msg, ' <- '.join(tb[-6:])))
        fs.done()
        if world > 1:
            os._exit(1)                   # no interpreter teardown: a half-dead communicator can hang in its destructor
        raise


def run(a, rank, world, local, fs):
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    if a.backend == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit('--gpus %d on a box with %d GPU(s): RCCL wants one device per rank (--backend gloo lets '
                         'several ranks share a device to exercise the control flow)' % (world, torch.cuda.device_count()))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in os.environ:
        os.environ['MASTER_PORT'] = str(free_port())       # only reached with world == 1 (forced collectives)
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    from hcmoco_amd.pycontrast.learning.affinity import pin_to_gpu_node
    pinned = pin_to_gpu_node(dev.index)           # the cores of the GPU's own socket (HCM_PIN_NUMA=0: leave it to the OS)
    forced = world == 1 and os.environ.get('HCM_FORCE_COLLECTIVES', '0') != '0'
    if world > 1 or forced:      # forced: a 1-rank group that still runs every collective (cost of the N>1 path)
        from datetime import timedelta
        fs.enter('init_process_group')
        # 120 s, not the default 10 minutes: a mis-wired rank must not burn the lease without a JSON line
        dist.init_process_group(a.backend, rank=rank, world_size=world, device_id=dev,
                                timeout=timedelta(seconds=FailSafe.PG_TIMEOUT_S))      # nccl = RCCL over xGMI
        fs.connect_store()
    fault = None
    if a.fault:
        fr, fstep, fkind = a.fault.split(':')
        fault = (int(fstep), fkind) if int(fr) == rank else None
    fs.enter('build')

    import tempfile
    from hcmoco_amd import hip_ops
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    B = a.batch_per_gpu
    args = make_args(B * world, a.nce_k, a.n_data, a.size, a.skeleton, a.backend, tempfile.mkdtemp(),
                     a.steps + a.warmup + 2, sampled=a.sampled_projection, arch=a.arch, width=a.width,
                     bank_dtype=a.bank_dtype, fmap_dtype=a.fmap_dtype, encoder_dtype=a.encoder_dtype,
                     wgrad_stream=a.wgrad_stream)
    args.rank, args.world_size, args.local_rank, args.gpu = rank, world, local, dev.index
    args.channels_last = bool(a.channels_last)
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    recorder = None
    if not a.no_check and rank == 0:
        # product engine + a tape of what the loss kernels saw and returned during ONE untimed step (see check_step)
        from hcmoco_amd.pycontrast.learning.engine import RecordingEngine
        recorder = RecordingEngine(a.fmap_dtype)
        recorder.armed = False
    hip_ops.ROW8_CHANNELS_LAST = bool(a.row8_channels_last)
    trainer = ContrastTrainer(args, engine=recorder)                         # HIP loss engine
    trainer.device = dev
    model, contrast, opt, data = build(args, trainer, dev)
    torch.cuda.manual_seed(1234 + rank)          # per-replica pixel sampling; weights were built from seed 0

    it = iter(data)
    records_path = None
    fs.enter('first steps')
    if not a.no_check:
        # two extra untimed steps on every rank (the collectives must match): the quiet-Find step, then a step of
        # the default runtime that rank 0 records for the checker
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        if recorder is not None:
            recorder.armed = True
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        if recorder is not None:
            recorder.armed = False
            torch.cuda.synchronize()
            records_path = os.path.join(tempfile.mkdtemp(), 'step_records.pt')
            torch.save(recorder.records, records_path)
            recorder.records = []
    fs.enter('warmup')
    for _ in range(a.warmup):
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        fs.t_phase = time.monotonic()         # progress
    ranks_seen = None
    if dist.is_initialized():
        # the first multi-GPU run must be able to say WHY it scales as it does: exposed communication per step from HIP
        # events around the two places where the trainer's stream waits for a collective, and the number of ranks that
        # actually answered a sum
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        fs.comm['ranks_seen'] = ranks_seen
        if trainer.grad_sync is not None:
            trainer.grad_sync.wait_events = []
            trainer.grad_sync.ready_events = []
        hip_ops.GATHER_WAIT_EVENTS = []
    hip_ops.prof_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    last = None
    fs.enter('timed steps')
    for step_ in range(a.steps):
        if fault is not None and step_ == fault[0]:
            if fault[1] == 'exit':
                os._exit(17)                  # dies without a word: the launcher's SIGTERM is what rank 0 hears
            if fault[1] == 'hang':
                time.sleep(10 ** 6)
            raise RuntimeError('injected fault (bench.py --fault)')
        last = trainer.train_step(next(it), model, contrast, opt, stage2=True)
        fs.t_phase = time.monotonic()         # progress
    ev1.record()                 # main stream: it is ordered behind the side streams / RCCL at the end of a step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    comm = None
    if dist.is_initialized():
        ar = trainer.grad_sync.wait_events if trainer.grad_sync is not None else []
        ag = hip_ops.GATHER_WAIT_EVENTS or []
        mean_ms = lambda evs: round(sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs), 4) if evs else None
        comm = dict(fs.comm)
        comm.update(rccl=rccl_report() if a.backend == 'nccl' else None)
        comm.update({'allreduce_exposed_ms': mean_ms(ar), 'allgather_wait_ms': mean_ms(ag),
                'launches': trainer.grad_sync.launched if trainer.grad_sync is not None else 0,
                'ranks_seen': ranks_seen, 'world_size': world, 'backend': dist.get_backend(),
                'steps_measured': len(ar), 'rank': rank,
                'note': 'HIP events on the trainer stream around work.wait() of the gradient all-reduces (after backward '
                        'returned) and around the wait for the packed feature/index all-gather in front of the bank '
                        'update: the time the stream stands still for communication that compute did not cover'})
        # when is the LAST gradient chunk of a step ready on each rank, counted from the end of the previous step's collectives
        # (a point every rank passes together)?  max - min over the ranks = what the early ranks spend waiting inside the
        # all-reduce for the late one: an under-scaling curve can be read from the line without another run.
        rd = trainer.grad_sync.ready_events if trainer.grad_sync is not None else None
        mine = torch.tensor([mean_ms(rd) if rd else -1.0], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(t.item()), 4) for t in every]
        have = [v for v in per_rank if v >= 0]
        comm.update({'chunk_ready_ms_per_rank': per_rank if have else None,
                     'chunk_ready_skew_ms': round(max(have) - min(have), 4) if len(have) == world else None})
        if trainer.grad_sync is not None:
            trainer.grad_sync.wait_events = None
            trainer.grad_sync.ready_events = None
        hip_ops.GATHER_WAIT_EVENTS = None
    kern_ms, kern_n = hip_ops.prof_read()
    secondary_raw = {tag: hip_ops.prof_read(tag) for tag in ('dense_stats', 'dense_grad', 'scl_stats', 'scl_grad',
                                                               'sgc_fwd', 'sgc_bwd', 'row8_fwd', 'row8_dw', 'row8_bwd', 'joint',
                                                               'row8_nhwc')}
    row8_nhwc_bytes = hip_ops.prof_read_work('row8_nhwc') / max(secondary_raw['row8_nhwc'][1], 1)
    hip_ops.prof_enable(False)
    # The SemGCN runs on a side stream next to two encoder streams: a hipEvent pair around its launches inside the step
    # spans queueing behind them, not kernel time (r02: 0.0696 ms reported against 14.8 us of kernels in rocprofv3).
    # Its roofline entry is therefore timed on the IDLE GPU, after the timed region: the same layer kernels at the same
    # shape (one _GraphConv of the model: [B, J, 128] -> [B, J, 128] with BatchNorm1d + ReLU), 20 forward + backward
    # passes; the in-step spans are kept next to it.
    sgc_idle = {}
    if rank == 0 and secondary_raw['sgc_fwd'][1]:
        layer = trainer.unwrap(model).encoder3.gconv_layers[0].gconv1
        xin = torch.randn(B, layer.gconv.adj.shape[0], layer.gconv.in_features, device=dev, requires_grad=True)
        torch.cuda.synchronize()
        for it_ in range(23):
            if it_ == 3:
                torch.cuda.synchronize()
                hip_ops.prof_enable(True)
            y = layer(xin)
            y.backward(torch.ones_like(y))
            torch.cuda.synchronize()            # one layer in flight at a time: nothing else on the device
        sgc_idle = {tag: hip_ops.prof_read(tag) for tag in ('sgc_fwd', 'sgc_bwd')}
        hip_ops.prof_enable(False)
        for p_ in layer.parameters():
            p_.grad = None
    # BASELINE config 4: the PointNet++ kernels' rooflines.  Inside the step the cloud branch shares the
    # device with two more busy streams (event spans there include queueing, like the SemGCN's above), so the encoder is run
    # ALONE after the timed region: the model's own Pointnet2MSG at the step's shapes ([B, 4096, 3] clouds back-projected from a
    # batch of the synthetic source), forward + backward, 1 + 3 passes; every launcher adds its algorithmic work to its tag.
    pn_idle = {}
    if rank == 0 and a.arch == 'HRNetPN':
        net_pn = trainer.unwrap(model)
        batch_pn = data.pool[0]
        with torch.no_grad():
            sample_pn, _, _ = net_pn.depth2pts(batch_pn[0][:, 3:].float(), batch_pn[7], batch_pn[12], int(batch_pn[13][0]),
                                               int(batch_pn[14][0]), batch_pn[15])
        cloud_pn = sample_pn.transpose(1, 2).contiguous()
        torch.cuda.synchronize()
        for it_ in range(4):
            if it_ == 1:
                torch.cuda.synchronize()
                hip_ops.prof_enable(True)
            net_pn.encoder2(cloud_pn).sum().backward()
            torch.cuda.synchronize()
        for tag in ('conv1x1_fwd', 'conv1x1_dx', 'conv1x1_dw', 'ball_fwd', 'ball_bwd', 'ballmax_fwd', 'ballmax_bwd', 'fps',
                    'three_nn', 'ball_query'):
            pn_idle[tag] = hip_ops.prof_read(tag) + (hip_ops.prof_read_work(tag),)
        hip_ops.prof_enable(False)
        for p_ in net_pn.encoder2.parameters():
            p_.grad = None
    kept_per_step = float(sum(int(b[6].sum()) for b in data.pool)) / len(data.pool)    # images with depth: B' of the dense loss
    tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
    dt = float(tdt.item())
    loss = float(last['loss'].item())
    if dist.is_initialized():
        # every rank leaves the process group HERE, together: what follows (stand-alone kernel timings, the oracle check
        # of the recorded step, the CPU baseline) is rank 0's own work and takes a minute; the other ranks must not sit in
        # a communicator teardown -- or trip its watchdog -- while it runs
        fs.enter('finish')
        dist.barrier()
        dist.destroy_process_group()
    fs.done()
    if not (loss == loss):
        raise SystemExit('non-finite loss in the timed region')

    if rank == 0:
        K1, D = a.nce_k + 1, 128
        # algorithmic bytes of one gather pass: 3 gathered rows + the int64 row index
        # per (sample, negative), plus the 3 query rows and 3 gradient rows per sample
        row_bytes = 2 if a.bank_dtype == 'bf16' else 4
        bytes_per_sample = 3 * K1 * D * row_bytes + K1 * 8 + 12 * D * 4
        bytes_per_launch = B * bytes_per_sample
        avg_ms = kern_ms / max(kern_n, 1)
        # the three banks of the headline configuration (201 MB) fit the 256 MiB Infinity Cache (MALL): between two
        # steps the encoders' activations evict them, but WITHIN a launch a row fetched for one sample can be served
        # from the MALL to the next -- the fabric-side rate can then exceed what HBM alone delivers, so the bound is
        # labelled for what it is; profiles/r03_bank_pass_sweep.json holds the HBM-resident cases
        bank_bytes = 3 * a.n_data * D * row_bytes
        bound = 'hbm+mall' if bank_bytes < 256 * 2 ** 20 else 'hbm'
        bound_note = ('banks (%.0f MB) fit the 256 MiB Infinity Cache: fabric-side bytes / peak HBM rate'
                      % (bank_bytes / 1e6) if bound == 'hbm+mall' else 'banks exceed the Infinity Cache: HBM-resident')
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if kern_n else None
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (it cannot be read live);
        # reported only when the committed measurement was taken at this exact configuration
        traffic, traffic_src = None, None
        for name in ('r06_bank_pass_pmc.json', 'r06_bank_pass_pmc_bf16_K131072.json'):      # this round's counters first
            if traffic is not None:
                break
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
                c = pmc['config']
                if ((c['B'], c['K'], c['n_data'], c['D']) == (B, a.nce_k, a.n_data, D)
                        and c['dtype'] == ('bf16' if a.bank_dtype == 'bf16' else 'f32')
                        and pmc['algorithmic_bytes_per_launch'] == bytes_per_launch):
                    traffic, traffic_src = pmc['traffic_bytes_per_launch'], 'profiles/' + name
            except (OSError, KeyError, ValueError):
                pass
        for sweep_name in ('r04_bank_pass_sweep.json', 'r03_bank_pass_sweep.json'):
            if traffic is not None:
                break
            try:                                        # one PMC cell per (n_data, K, dtype), B = 32, D = 128
                sweep = json.load(open(os.path.join(ROOT, 'profiles', sweep_name)))
                for c in sweep['pmc']:
                    if ((c['n_data'], c['K'], c['dtype']) == (a.n_data, a.nce_k, 'bf16' if a.bank_dtype == 'bf16' else 'fp32')
                            and (B, D) == (32, 128) and c['algorithmic_bytes'] == bytes_per_launch):
                        traffic, traffic_src = c['traffic_bytes'], 'profiles/' + sweep_name
            except (OSError, KeyError, ValueError):
                pass
        for name in ('r02_bank_pass_pmc.json', 'r02_bank_pass_pmc_bf16_K131072.json'):
            if traffic is not None:
                break
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
                c = pmc['config']
                if ((c['B'], c['K'], c['n_data'], c['D']) == (B, a.nce_k, a.n_data, D)
                        and c['dtype'] == ('bf16' if a.bank_dtype == 'bf16' else 'f32')):
                    traffic, traffic_src = pmc['traffic_bytes_per_launch'], 'profiles/' + name
            except (OSError, KeyError, ValueError):
                pass
        # secondary rooflines: algorithmic flops of one launch / mean launch duration
        from hcmoco_amd.pycontrast.networks.sgcn import num_joints
        S, J = 400, num_joints(a.skeleton)
        mfma_peak = MFMA_BF16_PEAK_TFS if a.fmap_dtype == 'bf16' else MFMA_F32_PEAK_TFS
        dense_gemm = 2.0 * (2.0 * kept_per_step * S * S * D)        # both orientations of Q K^T
        scl_gemm = 2.0 * (2 * B * J) ** 2 * D
        sgc_fwd_bytes = B * J * (2 * D + 2 * D) * 4                  # read H [B*J, 2C]; write out + xhat [B*J, C]
        sgc_bwd_bytes = B * J * (2 * D + 3 * D + 2 * D) * 4          # read H, dOut, out, xhat; write dH [B*J, 2C]
        # row 8 at the sampled pixels (csrc/rowproj.hip; networks/build_backbone.py:243-254, :290-300), both modalities (one for
        # HRNetPN): R = S + J rows per image; a row reads 1 tap of the finest branch and 4 of each coarse one
        net_ = trainer.unwrap(model)
        widths = [a.width * 2 ** i for i in range(4)]
        Ctot, ld = sum(widths), (sum(widths) + 1 + 3) // 4 * 4
        R, nmod = S + J, (2 if a.arch == 'HRNet' else 1)
        hw0 = (a.size // 4) ** 2
        row8_fwd_bytes = nmod * B * R * ((widths[0] + 4 * (Ctot - widths[0])) * 4 + (D + ld + D) * 4)
        row8_fwd_flops = nmod * B * R * 2.0 * (Ctot + 1) * D
        row8_dw_flops = nmod * 2.0 * (B * R) * D * ld
        row8_dw_bytes = nmod * B * R * (D + ld) * 4
        pix_c = sum(widths[i] * (hw0 >> (2 * i)) for i in range(4))            # sum_i C_i H_i W_i
        # (r06: the finest branch runs rows-first, S_0^T (grows W_0): R x C_0 x 128 per image instead of H_0 W_0 x C_0 x 128, and
        # its projected rows dxs0 [R, C_0 padded to 4] are written once and gathered once)
        c0p = (widths[0] + 3) // 4 * 4
        rows_first = (hw0 % 4 == 0) and c0p <= 64
        row8_bwd_flops = nmod * B * 2.0 * D * ((R * widths[0] + pix_c - widths[0] * hw0) if rows_first else pix_c)
        row8_bwd_bytes = nmod * B * (pix_c * 4 + R * D * 4 + (2 * R * c0p * 4 if rows_first else 0))
        joint_bytes = 3 * B * J * D * 4 * 2                                    # rows in, row gradients out
        spec = [('dense_stats', 'strip_kernel<Dense, stats> (S x S similarity + online softmax / soft targets)', 'mfma', dense_gemm),
                ('dense_grad', 'strip_kernel<Dense, grad> (similarity re-formed + G K contraction)', 'mfma', 2 * dense_gemm),
                ('scl_stats', 'strip_kernel<Scl, stats> + chunk merge (N x N, N = 2BJ)', 'mfma', scl_gemm),
                ('scl_grad', 'strip_kernel<Scl, grad> + chunk merge', 'mfma', 2 * scl_gemm),
                ('row8_fwd', 'project_rows_kernel (merge_all_res + 1x1 projection at the sampled pixels, fp32 MFMA)', 'mfma+bytes',
                 (row8_fwd_flops, row8_fwd_bytes)),
                ('row8_nhwc', 'nchw_to_nhwc_kernel (channels-last copies of the branches project_rows gathers from global memory)',
                 'hbm', row8_nhwc_bytes),
                ('row8_dw', 'proj_dw_partial + proj_dw_reduce (d[W | b] = grows^T xs, fp32 MFMA)', 'mfma+bytes',
                 (row8_dw_flops, row8_dw_bytes)),
                ('row8_bwd', 'finest_rows_kernel + branch_grad_t_kernel (branch-map gradients: finest branch S_0^T (grows W_0), the others '
                             'W_i^T (S_i^T grows), + pooling gradient, fp32 MFMA)',
                 'mfma+bytes', (row8_bwd_flops, row8_bwd_bytes)),
                ('joint', 'joint_nce_kernel + joint_finish_kernel (joint <-> graph-node InfoNCE)', 'latency', joint_bytes),
                ('sgc_fwd', 'SemGCN layer forward (sgc_mix + sgc_norm), per layer', 'latency', sgc_fwd_bytes),
                ('sgc_bwd', 'SemGCN layer backward (stats + bwd + finish), per layer', 'latency', sgc_bwd_bytes)]
        secondary = []
        for tag, name, kind, work in spec:
            ms, n = secondary_raw[tag]
            if not n:
                continue
            avg = ms / n
            in_step = None
            if tag in sgc_idle and sgc_idle[tag][1]:
                in_step = round(avg, 5)
                avg = sgc_idle[tag][0] / sgc_idle[tag][1]
                n = sgc_idle[tag][1]
            if kind == 'mfma+bytes':
                fl, by = work
                ach = fl / (avg * 1e-3) / 1e12
                gbs = by / (avg * 1e-3) / 1e9
                secondary.append({'kernel': name, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TFS,
                                  'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F32_PEAK_TFS, 4), 'flops_per_launch': int(fl),
                                  'bytes_per_launch': int(by), 'achieved_gbs': round(gbs, 1),
                                  'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4), 'avg_launch_ms': round(avg, 5),
                                  'launches_timed': n})
            elif kind == 'mfma':
                ach = work / (avg * 1e-3) / 1e12
                entry = {'kernel': name, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': mfma_peak,
                         'unit': 'TFLOP/s', 'frac': round(ach / mfma_peak, 4), 'flops_per_launch': int(work),
                         'avg_launch_ms': round(avg, 5), 'launches_timed': n}
                if a.fmap_dtype in ('fp32', 'fp32_exact'):
                    # what the fp32 instantiation really issues: every fp32 operand is split into two bf16 pieces and
                    # a product is 3 (dense) / 4 (SCL) v_mfma_f32_16x16x32_bf16 terms with fp32 accumulation; `frac` prices the
                    # ALGORITHMIC flops against the fp32-input MFMA rate an exact fp32 contraction would be bound by
                    terms = 9 if a.fmap_dtype == 'fp32_exact' else (4 if tag.startswith('scl') else 3)
                    entry.update({'arith': ('exact-product fp32: three bf16 pieces per operand, all 9 MFMA terms, fp32 accumulate'
                                            if terms == 9 else
                                            'fp32-accurate split-bf16: %d bf16 MFMA terms per product, fp32 accumulate '
                                            '(error vs float64 4e-6..6e-6, profiles/r05_split_bf16_error_study.txt)' % terms),
                                  'issued_tflops': round(ach * terms, 2),
                                  'frac_of_bf16_mfma_peak_issued': round(ach * terms / MFMA_BF16_PEAK_TFS, 4),
                                  'note': 'VALU / latency bound, not MFMA bound: matrix pipes busy 7 % of wave cycles '
                                          '(profiles/r05_strip_mfma_pmc_fp32.json)'})
                secondary.append(entry)
            elif kind == 'hbm':
                ach = work / (avg * 1e-3) / 1e9
                secondary.append({'kernel': name, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                                  'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'bytes_per_launch': int(work),
                                  'avg_launch_ms': round(avg, 5), 'launches_timed': n})
            else:
                ach = work / (avg * 1e-3) / 1e9
                secondary.append({'kernel': name, 'bound': 'latency', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                                  'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'bytes_per_launch': int(work),
                                  'avg_launch_ms': round(avg, 5), 'launches_timed': n,
                                  'timed': 'idle GPU after the timed region (kernel time; agrees with rocprofv3)',
                                  'span_ms_inside_the_step_incl_queueing': in_step})
        # config 4: sum(work) / sum(time) per PointNet++ kernel over every shape it was launched at (3 passes of the encoder)
        # r06: the 1x1 layers run split-bf16 from 64 channels up (3 bf16 MFMA terms, fp32 accumulate) and exact fp32 below: with
        # the matrix time a fifth of the fp32 form every layer is bound by moving its operands once -- priced against HBM
        c1_arith = ('exact fp32 MFMA (--fmap_dtype fp32_exact)' if a.fmap_dtype == 'fp32_exact' else
                    'split-bf16 from 64 channels up (3 bf16 MFMA terms per product, fp32 accumulate, 4.4e-6 of float64), exact fp32 below')
        PN_SPEC = [('conv1x1_fwd', 'conv1x1_split_kernel / conv1x1_rows_kernel forward (SharedMLP 1x1 convolutions + the source-point '
                                   'projection); arith: ' + c1_arith, 'hbm'),
                   ('conv1x1_dx', 'conv1x1_split_kernel / conv1x1_rows_kernel data gradient', 'hbm'),
                   ('conv1x1_dw', 'wgrad1x1_ball_kernel + wgrad1x1_reduce_kernel (weight gradient)', 'hbm'),
                   ('ball_fwd', 'ball_stats_kernel + ball_apply_kernel (first SharedMLP layer on the implicit grouped tensor)', 'hbm'),
                   ('ball_bwd', 'ball_bwd_reduce_kernel + ball_bwd_apply_kernel (+ dW_xyz merge)', 'hbm'),
                   ('ballmax_fwd', 'bn_stats_kernel + bn_relu_ballmax_kernel (last layer: BatchNorm + ReLU + max over the ball)', 'hbm'),
                   ('ballmax_bwd', 'ballmax_bwd_reduce_kernel + ballmax_bwd_apply_kernel', 'hbm'),
                   ('fps', 'fps_kernel (furthest point sampling, 4 levels)', 'pairs'),
                   ('three_nn', 'three_nn_split_kernel (FP levels)', 'pairs'),
                   ('ball_query', 'ball_query_wave_kernel (8 scales; early exit: b m n is an upper bound on its work)', 'pairs')]
        for tag, name, kind in PN_SPEC:
            ms, n, work = pn_idle.get(tag, (0.0, 0, 0.0))
            if not n or not ms:
                continue
            rate = work / (ms * 1e-3)
            common = {'kernel': name, 'avg_launch_ms': round(ms / n, 5), 'launches_timed': n,
                      'timed': 'cloud encoder alone on the idle GPU after the timed region, all shapes of one step summed'}
            if kind == 'mfma':
                secondary.append(dict(common, bound='mfma', achieved=round(rate / 1e12, 2), peak=MFMA_F32_PEAK_TFS, unit='TFLOP/s',
                                      frac=round(rate / 1e12 / MFMA_F32_PEAK_TFS, 4), flops_timed=int(work)))
            elif kind == 'hbm':
                secondary.append(dict(common, bound='hbm', achieved=round(rate / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                      frac=round(rate / 1e9 / HBM_PEAK_GBS, 4), bytes_timed=int(work)))
            else:       # rows 12 / 13 / 16 as distance evaluations per second against the fp32 vector peak (9 flop each)
                peak = MFMA_F32_PEAK_TFS * 1e12 / 9.0
                secondary.append(dict(common, bound='valu', achieved=round(rate / 1e9, 2), peak=round(peak / 1e9, 1),
                                      unit='G distance evaluations/s', frac=round(rate / peak, 4), pairs_timed=int(work)))
        out = {
            'metric': 'pretrain samples/sec (RGB+depth+kpt triples) HRNet-w18',
            'value': round(B * world * a.steps / dt, 3), 'unit': 'samples/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(1e3 * dt / a.steps, 3), 'ms_per_step_hipevent': round(ev_ms / a.steps, 3),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if a.encoder_dtype == 'fp32' else 'bf16 (encoders) / f32', 'data': 'synthetic',
            'config': {'workload': 'second-stage HCMoCo step (bank NCE + dense + joint + SCL losses, fwd+bwd+SGD+bank '
                                   'update), %s + SemGCN, %dx%d RGB+depth+%s keypoints'
                                   % ('HRNet-w%d x2' % a.width if a.arch == 'HRNet' else
                                      'HRNet-w%d (RGB) + PointNet++ MSG (depth cloud, 4096 pts)' % a.width,
                                      a.size, a.size, a.skeleton),
                       'bank_dtype': a.bank_dtype, 'fmap_dtype': a.fmap_dtype, 'encoder_dtype': a.encoder_dtype,
                       'batch_per_gpu': B, 'global_batch': B * world, 'nce_k': a.nce_k, 'n_data': a.n_data,
                       'samples_per_image': 400, 'feat_dim': D, 'parallelism': 'dp%d' % world,
                       'channels_last': bool(a.channels_last), 'sampled_projection': bool(a.sampled_projection),
                       'wgrad_stream': int(a.wgrad_stream),
                       'grad_collectives_per_step': (trainer.grad_sync.launched if trainer.grad_sync is not None
                                                     else (None if world > 1 else 0)),
                       'backend': a.backend if (world > 1 or forced) else None,
                       'cpu_affinity': None if pinned is None else {'numa_node': pinned[0], 'cpus': len(pinned[1])},
                       'final_loss': round(loss, 4)},
            'roofline': {'kernel': ('bank_pass_lean_kernel<bf16,ring 4>' if a.bank_dtype == 'bf16' else
                                    'bank_pass_lean_kernel<f32,ring 2>') + ' (gather + 6 logit sets + softmax + d/dx)',
                         'bound': bound, 'bank_bytes': bank_bytes, 'bound_note': bound_note,
                         'achieved': None if achieved is None else round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'traffic_source': traffic_src, 'bytes_per_launch': bytes_per_launch,
                         'avg_launch_ms': round(avg_ms, 5), 'launches_timed': kern_n},
        }
        if a.encoder_dtype == 'bf16':
            out['config']['note'] = ('--encoder_dtype bf16 is correct but not accelerated: stock module path instead of the '
                                     'encoder programs, slower than fp32 (465.6 vs 739.4 samples/s in r05); the encoders are '
                                     'outside SURVEY 8 (2.1 #9)')
        out['roofline_secondary'] = secondary
        out['comm'] = comm
        if records_path is not None:
            out['check'] = check_step(records_path)
            out['checked'] = bool(out['check'].get('checked'))
        else:
            out['checked'] = False
        if world == 1 and not a.no_cpu_baseline and a.arch == 'HRNet':
            out['cpu_baseline'] = cpu_baseline(a.nce_k, a.n_data, a.size, a.skeleton, B)
        else:
            out['cpu_baseline'] = None
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio, which is block-
        # buffered on a pipe and would otherwise come out at process exit, after python's own (flushed) print
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
