"""CPU affinity of a rank: the cores of the NUMA node its GPU hangs off.

An MI355X node is two sockets with four GPUs each; a rank that the scheduler parks on the other socket rings its GPU's
doorbells and runs MIOpen's host code across the socket link.  The reference leaves placement to SLURM's ``--cpu-bind``
(scripts/SecondStage/*.sh launch one task per GPU); under ``torch.distributed.run`` nothing binds, so the trainer does it:
``pin_to_gpu_node(device_index)`` reads the GPU's PCI address from the device properties, its ``numa_node`` /
``local_cpulist`` from sysfs and restricts the process (and every thread it starts later: the encoder runtime's helper
threads) to that list.  ``HCM_PIN_NUMA=0`` turns it off, ``HCM_PIN_NUMA=share`` additionally deals the node's cores out
among the ranks of that node.  Anything unreadable -> no change, ``None`` returned."""
import os

import torch


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_node(device_index):
    """(numa node, cpu list) of a visible GPU, or None."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        addr = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        base = os.path.join('/sys/bus/pci/devices', addr)
        node = int(open(os.path.join(base, 'numa_node')).read())
        cpus = _parse_cpulist(open(os.path.join(base, 'local_cpulist')).read())
        if node < 0 or not cpus:
            return None
        return node, cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def pin_to_gpu_node(device_index, mode=None):
    mode = os.environ.get('HCM_PIN_NUMA', 'node') if mode is None else mode
    if mode in ('0', 'off', 'none') or not hasattr(os, 'sched_setaffinity'):
        return None
    info = gpu_node(device_index)
    if info is None:
        return None
    node, cpus = info
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if mode == 'share':
        peers = [i for i in range(torch.cuda.device_count()) if (gpu_node(i) or (None,))[0] == node]
        if device_index in peers and len(peers) > 1:
            # physical cores first, their SMT siblings second (cpulist "64-127,192-255"): deal whole cores
            half = len(allowed) // 2
            cores = list(zip(allowed[:half], allowed[half:])) if half and len(allowed) % 2 == 0 else [(c,) for c in allowed]
            k, n = peers.index(device_index), len(peers)
            mine = cores[k * len(cores) // n:(k + 1) * len(cores) // n]
            allowed = sorted(c for core in mine for c in core)
    if not allowed:
        return None
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return None
    return node, allowed
