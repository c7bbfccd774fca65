"""world_size-2 gloo test of the one collective on the path: gather of fixed-layout predictions to rank 0
(diffusionvid_amd/engine/inference.py), plus video-sharded ownership (samplers)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from diffusionvid_amd.data.samplers import VIDTestDistributedSampler
    from diffusionvid_amd.engine import inference as eng
    from diffusionvid_amd.structures.bounding_box import BoxList
    from diffusionvid_amd.utils import comm
    comm.init_dist("gloo")
    assert comm.get_world_size() == world and comm.get_rank() == rank
    ds = type("DS", (), {"start_index": [0, 13, 21, 40], "__len__": lambda self: 50})()
    mine = list(VIDTestDistributedSampler(ds, world, rank))
    g = torch.Generator().manual_seed(100 + rank)
    res = {}
    for i in mine:                                    # ragged detections per frame, rank-specific content
        k = (i * 7) % 11
        bl = BoxList(torch.full((k, 4), float(i)), (1000, 600))
        bl.add_field("scores", torch.rand(k, generator=g))
        bl.add_field("labels", torch.full((k,), i % 30 + 1))
        res[i] = bl
    comm.synchronize()
    merged = eng.gather_predictions(res, max_det=300)
    if rank == 0:
        ok = sorted(merged) == list(range(50))
        ok = ok and all(len(merged[i]) == (i * 7) % 11 for i in range(50))
        ok = ok and all(float(merged[i].bbox.sum()) == 4.0 * i * ((i * 7) % 11) for i in range(50))
        ok = ok and all(int(merged[i].get_field("labels")[0]) == i % 30 + 1 for i in range(50) if len(merged[i]))
        q.put(("root", ok, len(mine)))
    else:
        q.put(("leaf", merged is None, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_predictions_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] for o in out), out
    assert sum(o[2] for o in out) == 50           # every frame owned exactly once
