import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # Built artefacts are git-ignored: on a fresh checkout compile the product library (hipcc
    # cross-compiles gfx950 without a GPU) and the C oracle before any test imports them.
    from hcmoco_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import pointnet2_oracle
    if not os.path.exists(pointnet2_oracle._SO):
        pointnet2_oracle.build()
    # The GPU box's host has 128 cores / 256 threads: torch then runs every small CPU op of the ORACLE (a gather of
    # 131 K rows, a 128-long dot product per row) across 256 OpenMP threads, and the fork-join cost is 10-20 x the work
    # (r06: the K = 65536 oracle check took 50 s there, 3.7 s on the 8-core build container).  16 threads, like the
    # checker subprocess of bench.py.  Host-side arithmetic is unaffected (same ops, fewer workers).
    import torch
    if (os.cpu_count() or 1) > 32 and 'OMP_NUM_THREADS' not in os.environ:
        torch.set_num_threads(16)


@pytest.fixture(scope='session', autouse=True)
def _scratch_dir():
    """Every ``tempfile.mkdtemp()`` of the suite (model / tensorboard folders: a stage-2 checkpoint is 150-450 MB) lands
    under ONE directory that is removed when the session ends; the tests used to leave them in /tmp, which filled the
    build container's disk over three rounds."""
    import shutil
    import tempfile
    base = tempfile.mkdtemp(prefix='hcmoco_tests_')
    old = tempfile.tempdir
    tempfile.tempdir = base
    try:
        yield base
    finally:
        tempfile.tempdir = old
        shutil.rmtree(base, ignore_errors=True)


def load_golden(name):
    import torch
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in 'US':
            out[k] = v
        elif v.ndim == 0:
            out[k] = v.item()
        else:
            out[k] = torch.from_numpy(v)
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden


def free_port():
    """A TCP port nobody is listening on right now (bind to 0, read it back): the process-group tests used to derive
    their port from the pid, which collides with a lingering listener or a TIME_WAIT socket once in a while."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]
