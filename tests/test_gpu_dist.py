"""The multi-GPU path on ONE GPU: everything an N-rank job runs through RCCL -- process-group set-up, the all_gather of shard
sizes, the gather of the predictions to rank 0, the broadcast of a video's memory, the all-reduces of bench.py -- executed in
a one-rank `nccl` group in a subprocess, and the VID-val-shaped workload (BASELINE.json configs[4]) end to end on one rank.
The N > 1 logic itself (ranks owning disjoint videos, ragged shards) is covered on the CPU by tests/test_dist_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


_COLLECTIVES = r"""
import os, torch, torch.distributed as dist
from diffusionvid_amd.engine import inference as eng
from diffusionvid_amd.structures.bounding_box import BoxList
from diffusionvid_amd.utils import comm
comm.init_dist("nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
res = {}
for i in range(40):
    k = (i * 7) % 11 if i != 17 else 777            # one frame beyond any fixed cap (x4 ensembles keep up to 900)
    bl = BoxList(torch.rand(k, 4, generator=g) * 500, (1000, 600))
    bl.add_field("scores", torch.rand(k, generator=g))
    bl.add_field("labels", torch.randint(1, 31, (k,), generator=g))
    res[i] = bl
merged = eng.gather_predictions(res, device=dev, always=True)          # all_gather of sizes + 4 gathers of device tensors through RCCL
assert sorted(merged) == list(range(40))
for i in range(40):
    assert torch.equal(merged[i].bbox, res[i].bbox) and torch.equal(merged[i].get_field("scores"), res[i].get_field("scores"))
    assert torch.equal(merged[i].get_field("labels"), res[i].get_field("labels")) and merged[i].size == (1000, 600)
mem = [torch.randn(900, 256, device=dev), torch.randn(150, 256, device=dev)]    # a video's global memory (engine.compute_on_video_sharded)
ref = [m.clone() for m in mem]
for m in mem:
    dist.broadcast(m, src=0)
assert all(torch.equal(a, b) for a, b in zip(mem, ref))
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_rccl_one_rank_group_gather_broadcast_allreduce():
    env = _env()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", _COLLECTIVES], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_under_torchrun_one_rank_runs_the_rccl_gather():
    """bench.py exactly as the driver launches it for N ranks (python -m torch.distributed.run ... bench.py --gpus N), with N = 1
    and --force-dist: the timed region ends with the gather of the predictions and the max / sum all-reduces through RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--frames", "32", "--lookahead", "2",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-fed", "--no-side-configs", "--no-feed-rate"]
    out = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "RCCL group of 1 rank(s)" in line["config"]["parallelism"]
    assert line["host_view"]["gather_ms_max_over_ranks"] > 0


def test_vidval_workload_twelve_videos_end_to_end():
    """bench.py --workload vidval --videos 12: ragged video lengths (87 .. 820 frames), per-video memory builds, engine-built
    look-ahead groups with ragged tails, one rank, predictions of every frame accounted for (the line's own assertion)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "vidval", "--videos", "12", "--force-dist"]
    out = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["value"] > 0 and "12 synthetic videos / 3486 frames" in line["config"]["workload"]
    assert line["config"]["process_group"] == "nccl, 1 rank(s)"
