"""Inference loop and result gathering (mirror of mega_core/engine/inference.py:22-181).

`compute_on_dataset` keeps the reference's conventions: one dataset item per call, wall time
around `model(images)` with a device sync (inference.py:30, :70-73), outputs moved to the CPU and
zipped with the item's image ids (:75, :91-93).  Multi-GPU: ranks own disjoint video ranges and
never talk during inference; at the end ONE collective gathers fixed-layout tensors
[frames, max_det, 6] (+ per-frame counts and ids) to rank 0 -- over RCCL/xGMI each non-root rank
has its own link to root -- replacing the reference's pickled byte all_gather to every rank
(mega_core/utils/comm.py:54-94).
"""
import time

import torch
import torch.distributed as dist

from ..structures.bounding_box import BoxList
from ..utils import comm


def lookahead_items(dataset, indices, infer_batch, lookahead):
    """The look-ahead hand-over built by the ENGINE from an unchanged dataset: yields `dataset[idx]` for idx in `indices`, in
    order, each item loaded exactly once; on the first call of every look-ahead group (frame_id a multiple of infer_batch *
    lookahead) the items of the group's later calls are read ahead and their `ref_l` frames -- exactly what those calls
    would deliver -- are attached as `ref_ahead = {batch start frame: [frames]}` (INPUT.LOOKAHEAD_BATCHES of the MI355X
    schedule).  The reference's own VIDMEGADataset (vid_mega.py:164-250) therefore needs no change: the detector still
    receives every call with its usual keys and returns the later batches' results from the group's first call's work.
    Items that already carry `ref_ahead` (datasets that emit it themselves) pass through untouched."""
    indices = list(indices)
    unit = infer_batch * lookahead
    cache = {}
    pos_of = {idx: k for k, idx in enumerate(indices)}

    def get(idx):
        if idx not in cache:
            cache[idx] = dataset[idx]
        return cache[idx]

    for idx in indices:
        item = get(idx)
        images = item[0]
        if lookahead > 1 and "ref_ahead" not in images and images["frame_id"] % unit == 0:
            f0, end = images["frame_id"], images["end_id"]
            ahead = {}
            k = pos_of[idx]
            # batches of this group after the first: fb = f0 + infer_batch, ... ; batch fb is fed by the calls fb - infer_batch + 1 .. fb
            for fb in range(f0 + infer_batch, min(f0 + unit, end + 1), infer_batch):
                frames = []
                for f in range(fb - infer_batch + 1, fb + 1):
                    j = k + (f - f0)
                    if j >= len(indices):
                        frames = None
                        break
                    nxt = get(indices[j])[0]
                    if nxt["frame_id"] != f or nxt["frame_category"] == 0:
                        frames = None             # not the same video / not consecutive: leave this batch to its own call
                        break
                    frames += list(nxt["ref_l"])
                if frames is None or len(frames) != infer_batch:
                    break
                ahead[fb] = frames
            images = dict(images)
            images["ref_ahead"] = ahead
            item = (images,) + tuple(item[1:])
        cache.pop(idx, None)
        yield idx, item


def compute_on_dataset(model, dataset, indices, device, timer=None):
    """mega_core/engine/inference.py:22-94.  With INPUT.LOOKAHEAD_BATCHES > 1 on the model the hand-over is built here
    (`lookahead_items`), so any dataset that follows the reference's item protocol gets the grouped schedule."""
    model.eval()
    results = {}
    cpu = torch.device("cpu")
    la = int(getattr(model, "lookahead", 1) or 1)
    for idx, item in lookahead_items(dataset, indices, getattr(model, "infer_batch", 1), la):
        images, _, image_ids = item
        with torch.no_grad():
            t0 = time.perf_counter()
            output = model(images)
            if device.type != "cpu":
                torch.cuda.synchronize()
            if timer is not None:
                timer.append(time.perf_counter() - t0)
            output = [o.to(cpu) for o in output]
        results.update({img_id: r for img_id, r in zip(image_ids, output)})
    return results


def max_detections(results):
    """largest per-frame detection count of a shard (the x4 ensemble keeps up to 3 x NUM_PROPOSALS candidates through
    NMS, diffusion_det.py:607-627, so the count is a property of the data, not of the config)"""
    return max((len(bl) for bl in results.values()), default=0)


def video_shard_plan(num_frames, infer_batch, lookahead, world):
    """One video over `world` ranks (single-video case).  The unit is a look-ahead group (`infer_batch *
    lookahead` frames): given the video's global memory every group is independent -- the memory is final after the
    first call (GLOBAL.STOP_UPDATE_AFTER_INIT_TEST), local attention is off, the local queue is exactly one batch, and the
    random draws are keyed by (video, call frame, step), not by RNG stream position.  Group g runs on rank g % world; group 0
    (whose first call also carries the global frames) therefore on rank 0.  Returns, per rank, the list of inclusive
    dataset-offset ranges (first call, last call) to feed: a group's first batch needs the `infer_batch - 1` calls before
    it, which only queue their local frame (diffusion_det.py:410-412)."""
    unit = infer_batch * lookahead
    plan = [[] for _ in range(world)]
    for g, f0 in enumerate(range(0, num_frames, unit)):
        last_batch = min(f0 + unit, -(-num_frames // infer_batch) * infer_batch) - infer_batch
        last_batch = min(last_batch, (num_frames - 1) // infer_batch * infer_batch)
        plan[g % world].append((max(0, f0 - (infer_batch - 1)), last_batch))
    return plan


def compute_on_video_sharded(model, dataset, start, num_frames, device, rank=None, world=None, broadcast=None):
    """Run ONE video (dataset indices [start, start + num_frames)) across the ranks of the process group following
    `video_shard_plan`: rank 0 runs the first call (24 global + the first local frames) and the memory it builds --
    [900, d] + [150, d] fp32, 1.07 MB -- is broadcast (RCCL over xGMI; the only data-path exchange of this mode);
    every rank then runs its own groups.  Returns this rank's {image id: BoxList}; merge with gather_predictions.
    `broadcast(tensors, src)`: injected for tests; default torch.distributed.broadcast of each tensor."""
    rank = comm.get_rank() if rank is None else rank
    world = comm.get_world_size() if world is None else world
    plan = video_shard_plan(num_frames, model.infer_batch, model.lookahead, world)[rank]
    results = {}
    cpu = torch.device("cpu")

    def feed(a, b):
        for off in range(a, b + 1):
            images, _, ids = dataset[start + off]
            with torch.no_grad():
                out = model(images)
            results.update({i: o.to(cpu) for i, o in zip(ids, out)})

    todo = list(plan)
    if rank == 0:
        feed(0, 0)
        mem = [m.contiguous() for m in model.global_memory()]
    else:
        mem = [torch.empty(s, dtype=torch.float32, device=device) for s in model.global_memory_shapes()]
    if world > 1:
        if broadcast is not None:
            mem = broadcast(mem, 0)
        else:
            for m in mem:
                dist.broadcast(m, src=0)
    if rank != 0:
        model.adopt_video_memory(mem)
    for a, b in todo:
        feed(max(a, 1) if (rank == 0 and a == 0) else a, b)
    return results


def pack_predictions(results, max_det=None):
    """dict{id: BoxList} -> (ids [n] int64, counts [n] int32, dets [n, max_det, 6] fp32 = box4, score, label,
    sizes [n,2] int32).  max_det None = the shard's own maximum; a smaller explicit value is an error (nothing is
    truncated silently)."""
    ids = sorted(results.keys())
    n = len(ids)
    need = max_detections(results)
    if max_det is None:
        max_det = need
    elif need > max_det:
        raise ValueError("a frame holds %d detections but max_det is %d" % (need, max_det))
    dets = torch.zeros((n, max_det, 6), dtype=torch.float32)
    counts = torch.zeros((n,), dtype=torch.int32)
    sizes = torch.zeros((n, 2), dtype=torch.int32)
    for j, i in enumerate(ids):
        bl = results[i]
        k = len(bl)
        counts[j] = k
        sizes[j, 0], sizes[j, 1] = bl.size
        if k:
            dets[j, :k, :4] = bl.bbox
            dets[j, :k, 4] = bl.get_field("scores")
            dets[j, :k, 5] = bl.get_field("labels").to(torch.float32)
    return torch.tensor(ids, dtype=torch.int64), counts, dets, sizes


def unpack_predictions(ids, counts, dets, sizes):
    out = {}
    for j, i in enumerate(ids.tolist()):
        k = int(counts[j])
        bl = BoxList(dets[j, :k, :4].clone(), (int(sizes[j, 0]), int(sizes[j, 1])), mode="xyxy")
        bl.add_field("scores", dets[j, :k, 4].clone())
        bl.add_field("labels", dets[j, :k, 5].to(torch.int64))
        out[i] = bl
    return out


def gather_predictions(results, max_det=None, device=None, always=False):
    """Gather every rank's {image_id: BoxList} on rank 0 (returns None elsewhere).  The padded per-frame capacity is
    the maximum detection count over all ranks (exchanged with the shard sizes) unless `max_det` forces a larger one.
    always=True runs the collectives even in a one-rank group (the RCCL path on a single GPU)."""
    world = comm.get_world_size()
    if world == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return results
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    n_local = torch.tensor([len(results), max_detections(results)], dtype=torch.int64, device=dev)
    all_n = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(all_n, n_local)                       # 16 bytes per rank: shard size, largest detection count
    all_n = [x.cpu() for x in all_n]
    n_max = int(max(int(x[0]) for x in all_n))
    det_max = max([int(x[1]) for x in all_n] + [int(max_det or 0), 1])
    ids, counts, dets, sizes = pack_predictions(results, det_max)
    all_n = [x[:1] for x in all_n]

    def pad(t):
        p = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype)
        p[: t.shape[0]] = t
        return p.to(dev)

    payload = [pad(ids), pad(counts), pad(dets), pad(sizes)]
    rank = comm.get_rank()
    gathered = []
    for t in payload:                                     # the data collective: gather to rank 0
        bufs = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gather_list=bufs, dst=0)
        gathered.append(bufs)
    if rank != 0:
        return None
    merged = {}
    for r in range(world):
        n = int(all_n[r].item())
        merged.update(unpack_predictions(gathered[0][r][:n].cpu(), gathered[1][r][:n].cpu(), gathered[2][r][:n].cpu(),
                                         gathered[3][r][:n].cpu()))
    return merged


def predictions_list(merged):
    """dict -> list indexed by dataset image id, as torch.save'd to predictions.pth (inference.py:101-115)."""
    ids = sorted(merged.keys())
    return [merged[i] for i in ids]


def inference(model, dataset, indices, device, output_folder=None, gt_boxlists=None, max_det=None, motion_specific=False,
              motion_ious=None, logger=None, class_module=None):
    """Reference `inference` (mega_core/engine/inference.py:118-181) for the in-scope path: run this rank's share, gather on
    rank 0, write `predictions.pth` (`class_module`: see vid_eval.save_predictions) and evaluate as the reference's
    `evaluate` -> `do_vid_evaluation` does (vid_eval.py:14-78): when the dataset serves `get_img_info` / `get_groundtruth`
    the predictions are mapped from the resized frame to the annotation's original size first and `result.txt` is written;
    `gt_boxlists` (a list of BoxList, one per image id) stands in for a dataset without annotations -- each prediction is
    resized to its ground truth's size the same way.  Returns (predictions list | None off rank 0, eval | None):
    eval = the AP dict, or the list of one dict per motion range with `motion_specific`."""
    import os

    from ..data.evaluation import vid_eval
    from ..utils import comm
    results = compute_on_dataset(model, dataset, indices, device)
    merged = gather_predictions(results, max_det, device=device)
    if not comm.is_main_process():
        return None, None
    preds = predictions_list(merged)
    if output_folder:
        os.makedirs(output_folder, exist_ok=True)
        vid_eval.save_predictions(preds, os.path.join(output_folder, "predictions.pth"), class_module)
    source = None
    if gt_boxlists is not None:
        source = vid_eval.GroundTruthList(gt_boxlists, getattr(dataset, "map_class_id_to_class_name", None))
    elif hasattr(dataset, "get_groundtruth") and getattr(dataset, "annotations", True) is not None:
        source = dataset
    if source is None:
        return preds, None
    ev = vid_eval.do_vid_evaluation(source, preds, output_folder, motion_specific=motion_specific, logger=logger,
                                    motion_ious=motion_ious)
    return preds, (ev if motion_specific else ev[0])
