"""Process-group helpers (mirror of mega_core/utils/comm.py:10-51, dist_env.py:9-23).  One process per
GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def init_dist(backend=None, force=False, timeout_s=None):
    """env:// rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK.  A single process normally needs no
    group; force=True builds a one-rank group anyway (the RCCL path exercised on one GPU: tests, bench.py --force-dist).
    timeout_s bounds every collective: a rank that died (or left a side measurement through an exception) makes its peers' pending
    collective raise after that long instead of waiting for ever."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world <= 1 and not force) or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    kw = {}
    if timeout_s:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    dist.init_process_group(backend=backend, init_method="env://", **kw)


def cpu_quota():
    """CPUs' worth of run time the container may use per period (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cfs_period_us`), or None
    without a limit.  A host that shows 256 CPUs may schedule a container on 16 of them at a time: thread pools and decode pools sized by
    os.cpu_count() then run slower than pools sized by the quota."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def rank_cpu_share(local_rank, local_world, gpu_numa_nodes=None, node_cpus=None, allowed=None):
    """CPUs for rank `local_rank` of `local_world` ranks on this host: the ranks whose GPU hangs off NUMA node k share node k's
    CPUs in contiguous slices (the host threads of a rank -- its Python loop, pinned-memory copies, decode workers -- then run
    next to their GPU's PCIe root); without topology information the allowed CPUs are cut into `local_world` contiguous
    slices.  Pure function of its arguments (tests inject the topology): gpu_numa_nodes[i] = NUMA node of GPU i (-1 unknown),
    node_cpus = {node: [cpu, ...]}, allowed = CPUs this process may run on."""
    allowed = sorted(allowed if allowed is not None else range(os.cpu_count() or 1))
    if local_world <= 1:
        return allowed
    known = gpu_numa_nodes is not None and node_cpus and len(gpu_numa_nodes) >= local_world and \
        all(n in node_cpus for n in gpu_numa_nodes[:local_world])
    if known:
        node = gpu_numa_nodes[local_rank]
        peers = [r for r in range(local_world) if gpu_numa_nodes[r] == node]
        cpus = [c for c in node_cpus[node] if c in set(allowed)]
        per = len(cpus) // len(peers)
        if per >= 1:
            k = peers.index(local_rank)
            return cpus[k * per:(k + 1) * per]
    per = max(1, len(allowed) // local_world)
    lo = min(local_rank * per, len(allowed) - 1)
    return allowed[lo:lo + per]


def gpu_numa_topology(n_gpus):
    """([NUMA node of GPU i], {node: cpus}) from sysfs, or (None, None) when the platform does not say"""
    try:
        nodes = []
        for i in range(n_gpus):
            p = torch.cuda.get_device_properties(i)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
                nodes.append(int(f.read().strip()))
        node_cpus = {}
        for n in set(nodes):
            if n < 0:
                return None, None
            with open("/sys/devices/system/node/node%d/cpulist" % n) as f:
                node_cpus[n] = _parse_cpulist(f.read())
        return nodes, node_cpus
    except (OSError, AttributeError, ValueError, RuntimeError):
        return None, None


def rank_thread_cap(n_cpus, local_world, quota=None, ceiling=32):
    """Host threads (torch intra-op pool, decode pool) one rank should run: its CPU share, at most `ceiling`, and -- when the container
    has a CPU-time quota (cpu_quota(): the GPU boxes of this pool show 256 CPUs and grant 16) -- at most its share of the quota: eight
    ranks with 32 threads each on 16 CPUs' worth of run time spend their time in the scheduler (profiles/r05m_feed_groups.txt).
    Pure function (tests inject the numbers)."""
    cap = max(1, min(int(n_cpus), int(ceiling)))
    if quota is not None and quota > 0:
        cap = max(1, min(cap, int(quota // max(1, local_world))))
    return cap


def bind_rank_to_cpus(local_rank, local_world):
    """Pin this process (and the threads it starts later) to its share of the host's CPUs and cap its thread pool by its share of
    the CPU quota (rank_thread_cap); returns the CPU list (for the bench line) or None when affinity cannot be set here.  No-op for
    a single rank."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        nodes, node_cpus = gpu_numa_topology(local_world) if torch.cuda.is_available() else (None, None)
        cpus = rank_cpu_share(local_rank, local_world, nodes, node_cpus, allowed)
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(rank_thread_cap(len(cpus), local_world, cpu_quota()))
        return cpus
    except (OSError, ValueError):
        return None
