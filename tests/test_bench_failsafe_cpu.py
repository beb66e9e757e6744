"""bench.py's FailSafe without a GPU: rank 0 of a 2-rank job prints ONE JSON error line when (a) a peer
leaves a note in the rendez-vous store, (b) the launcher sends SIGTERM while the main thread is blocked inside a store wait
(a C call, like a collective), (c) a phase makes no progress for its limit.  The GPU form of the same check, through
``python bench.py --gpus 2 --backend gloo --fault ...``, is tests/test_rccl_gpu.py::test_bench_line_survives_a_dead_rank."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest
from conftest import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time
from datetime import timedelta
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import bench
mode = sys.argv[1]
port = int(sys.argv[2])
os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
server = dist.TCPStore('127.0.0.1', port, is_master=True, timeout=timedelta(seconds=30), wait_for_workers=False)
if mode == 'stall':
    bench.FailSafe.STALL_S = dict(bench.FailSafe.STALL_S, default=2.0)
fs = bench.FailSafe(0, 2, {'metric': 'pretrain samples/sec', 'value': None, 'n_gpus': 2})
fs.connect_store()
fs.comm['ranks_seen'] = 2
fs.enter('timed steps')
print('READY', flush=True)
try:
    server.wait(['never_set'], timedelta(seconds=60))      # blocked in C++ with the GIL released, like a collective
except Exception as e:
    print('wait ended:', type(e).__name__, flush=True)
time.sleep(60)
'''


def _spawn(mode, port):
    return subprocess.Popen([sys.executable, '-c', CHILD % {'root': ROOT}, mode, str(port)], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


def _finish(proc, limit=40):
    try:
        out, err = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, err = proc.communicate()
        pytest.fail('rank 0 did not exit: ' + out[-500:] + err[-1500:])
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, (out[-800:], err[-1500:])
    return json.loads(lines[0]), proc.returncode


def _wait_ready(proc):
    assert proc.stdout.readline().strip() == 'READY'


def test_peer_note_in_the_store_becomes_the_error_line():
    import torch.distributed as dist
    from datetime import timedelta
    port = free_port()
    proc = _spawn('note', port)
    _wait_ready(proc)
    client = dist.TCPStore('127.0.0.1', port, is_master=False, timeout=timedelta(seconds=10))
    client.set('hcm_bench_error/1', 'rank 1, phase timed steps: RuntimeError: injected')
    line, rc = _finish(proc)
    assert rc == 3 and 'injected' in line['error'] and line['phase'] == 'timed steps'
    assert line['value'] is None and line['comm']['ranks_seen'] == 2


def test_sigterm_while_blocked_in_a_c_call():
    port = free_port()
    proc = _spawn('term', port)
    _wait_ready(proc)
    time.sleep(0.5)
    proc.send_signal(signal.SIGTERM)
    line, rc = _finish(proc)
    assert rc == 3 and 'signal' in line['error'] and line['phase'] == 'timed steps'


def test_stalled_phase_hits_its_limit():
    port = free_port()
    proc = _spawn('stall', port)
    _wait_ready(proc)
    line, rc = _finish(proc)
    assert rc == 4 and 'no progress' in line['error']
