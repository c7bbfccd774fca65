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


def init_dist(backend=None, force=False):
    """env:// rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK.  A single process normally needs no
    group; force=True builds a one-rank group anyway (the RCCL path exercised on one GPU: tests, bench.py --force-dist)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world <= 1 and not force) or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")
