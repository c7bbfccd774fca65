"""Oracle: DiffusionDet.inference post-processing and batched NMS (CPU, numpy fp32).

Follows mega_core/modeling/detector/diffusion_det.py:754-839 (inference), :607-627
(x4 ensemble NMS) and mega_core/structures/bounding_box.py:214-224 (clip_to_image).
`batched_nms` is third-party (detectron2.layers.batched_nms -> torchvision
`_batched_nms_coordinate_trick` + `nms`, un-pinned): restated from the published
algorithm (IoU without +1, class separation by adding idx * (max_coord + 1), greedy
sweep in descending-score order).  Tie order is implementation-defined upstream; this
restatement fixes it to (score descending, then position ascending) = a stable sort.
"""
import numpy as np
import torch


def nms_fp32(boxes, scores, iou_threshold, legacy=False):
    """torchvision.ops.nms CPU kernel (nms_kernel_impl), fp32 arithmetic in the same order.
    boxes [n,4] float32, scores [n] float32 -> kept indices in descending score order.

    legacy=True switches the two places where the reference's own `_C.nms`
    (mega_core/csrc/cpu/nms_cpu.cpp:24, :57-62; cuda/nms.cu:13-21) differs from torchvision's: pixel-inclusive
    extents (`x2 - x1 + 1`) and suppression at `ovr >= threshold` instead of `>`.  DiffusionVID does not call the
    legacy op (it uses detectron2's batched_nms, diffusion_det.py:617,:793); the switch exists so that the greedy
    sweep shared by both variants is pinned by the known-answer vectors of the reference's tests/test_nms.py:11-58
    (golden g12), which that source file passes."""
    one = np.float32(1 if legacy else 0)
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    zero = np.float32(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(zero, xx2 - xx1 + one)
            h = np.maximum(zero, yy2 - yy1 + one)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[(ovr >= thr) if legacy else (ovr > thr)]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision _batched_nms_coordinate_trick (used for < 4000 box coordinates)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    max_coordinate = boxes.max()
    offsets = np.asarray(idxs).astype(np.float32) * (max_coordinate + np.float32(1))
    boxes_for_nms = boxes + offsets[:, None]
    return nms_fp32(boxes_for_nms, scores, iou_threshold)


def clip_to_image(boxes, size_wh):
    """bounding_box.py:214-224 with remove_empty=False."""
    w, h = size_wh
    b = np.array(boxes, dtype=np.float32, copy=True)
    b[:, 0] = np.clip(b[:, 0], 0, w - 1)
    b[:, 1] = np.clip(b[:, 1], 0, h - 1)
    b[:, 2] = np.clip(b[:, 2], 0, w - 1)
    b[:, 3] = np.clip(b[:, 3], 0, h - 1)
    return b


def topk_candidates(box_cls, box_pred, num_classes):
    """diffusion_det.py:772-784 for one image: sigmoid, top-`num_proposals` of M*C scores.

    Returns (boxes [M,4], scores [M], labels [M] int64, flat indices) ordered by
    (score desc, flat index asc) -- upstream `topk(sorted=False)` leaves the order open."""
    scores = torch.sigmoid(box_cls).flatten(0, 1)
    M = box_cls.shape[0]
    order = torch.sort(scores, descending=True, stable=True).indices[:M]
    s = scores[order]
    labels = order % num_classes + 1
    boxes = box_pred[order // num_classes]
    return boxes.numpy().astype(np.float32), s.numpy().astype(np.float32), labels.numpy().astype(np.int64), order.numpy()


def inference_x1(box_cls, box_pred, image_size_wh, num_classes, use_nms=True, iou=0.5):
    """diffusion_det.py:777-812, per batch.  Returns list of dict(boxes, scores, labels)."""
    results = []
    for b in range(box_cls.shape[0]):
        boxes, scores, labels, _ = topk_candidates(box_cls[b], box_pred[b], num_classes)
        if use_nms:
            keep = batched_nms(boxes, scores, labels, iou)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        if (labels == 0).sum():
            raise NotImplementedError("Not supported model")
        results.append({"boxes": clip_to_image(boxes, image_size_wh), "scores": scores, "labels": labels})
    return results


def inference_ensemble(cands, image_size_wh, use_nms=True, iou=0.5):
    """diffusion_det.py:607-627: cands = list over steps of list over images of
    (boxes, scores, labels); concatenated along the candidate axis, then one NMS."""
    nimg = len(cands[0])
    results = []
    for b in range(nimg):
        boxes = np.concatenate([c[b][0] for c in cands], axis=0)
        scores = np.concatenate([c[b][1] for c in cands], axis=0)
        labels = np.concatenate([c[b][2] for c in cands], axis=0)
        if use_nms:
            keep = batched_nms(boxes, scores, labels, iou)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        if (labels == 0).sum():
            raise NotImplementedError("Not supported model")
        results.append({"boxes": clip_to_image(boxes, image_size_wh), "scores": scores, "labels": labels})
    return results
