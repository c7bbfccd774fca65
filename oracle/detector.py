"""Oracle: DiffusionDet test-time video state machine (CPU fp32).

Follows mega_core/modeling/detector/diffusion_det.py:306-336 (forward), :377-646
(_forward_test), :655-677 (model_predictions), :754-839 (inference).  The reference
cannot run this function on CPU as written (CUDA hard-codes at :590-591, :892-893 and
fps.h:15-36), so it is restated here; the pieces it calls are pinned individually.

Noise.  The reference draws `torch.randn` on the device (:449, :542, :587, :595).  For
parity the draws are injected: `noise_fn(kind, frame_id, step, image, shape)` with
kind in {"box_init", "img", "ddim", "renew"}; ragged draws take the leading rows of a
full [num_proposals, 4] draw.
"""
from collections import deque
from dataclasses import dataclass, field

import numpy as np
import torch

from . import backbone_r101, head, memory, postproc, schedule
from .head import HeadCfg


@dataclass
class DetCfg:
    num_proposals: int = 300
    num_classes: int = 30
    hidden_dim: int = 256
    sample_step: int = 1
    snr_scale: float = 2.0
    infer_batch: int = 8
    all_frame_interval: int = 8
    key_frame_location: int = 0
    mem_size_test: int = 900          # MODEL.VID.MEGA.MEMORY_MANAGEMENT_SIZE_TEST
    mem_size_dis: int = 150           # hard-coded at diffusion_det.py:487
    use_nms: bool = True
    pixel_mean: tuple = (123.675, 116.280, 103.530)
    pixel_std: tuple = (58.395, 57.120, 57.375)
    in_features: tuple = ("p3", "p4", "p5")
    blocks: tuple = backbone_r101.R101_BLOCKS
    head: HeadCfg = field(default_factory=HeadCfg)


def renew_and_ddim_step(buf, logits, pred_noise, x_start, time, time_next, noise, fresh, keep_thr=0.5):
    """diffusion_det.py:559-596 for time_next >= 0: box renewal (keep boxes whose best sigmoid score exceeds 0.5, in
    index order) + DDIM update with eta = 1 + replenishment with fresh N(0,1) boxes.
    logits [B, M, C]; pred_noise, x_start [B, M, 4]; noise[i], fresh[i]: full [M, 4] draws of image i, of which the leading
    `num_remain` / `M - num_remain` rows are consumed (:587, :595).  -> img [B, M, 4]."""
    batch, M = logits.shape[:2]
    score = torch.sigmoid(logits)                               # :560
    value, _ = torch.max(score, dim=-1)                         # :562
    keep_idx = value > keep_thr                                 # :563
    num_remain = torch.sum(keep_idx, dim=-1)                    # :565
    pred_noise_l = [pred_noise[i, keep_idx[i]] for i in range(batch)]       # :567-572
    x_start_l = [x_start[i, keep_idx[i]] for i in range(batch)]
    sqrt_an, cc, sigma = schedule.ddim_coefficients(buf["alphas_cumprod"], time, time_next)     # :577-584
    img_l = []
    for i in range(batch):                                      # :586-595
        nr = int(num_remain[i])
        x = x_start_l[i] * sqrt_an + cc * pred_noise_l[i] + sigma * noise[i][:nr]
        img_l.append(torch.cat((x, fresh[i][:M - nr]), dim=0))
    return torch.stack(img_l, dim=0)


class OracleDiffusionDet:
    def __init__(self, sd, cfg: DetCfg, noise_fn, backbone_fn=None):
        self.sd = sd
        self.cfg = cfg
        self.noise_fn = noise_fn
        self.buf = schedule.schedule_buffers(1000)
        self.num_timesteps = 1000
        self.backbone_fn = backbone_fn or (lambda x: backbone_r101.backbone_r101_fpn(x, sd, "backbone.", cfg.blocks))
        self.taps = {}

    # diffusion_det.py:655-677
    def model_predictions(self, feats, images_whwh, x, t, cached=None, mem=None, box_extract=0):
        c = self.cfg
        x_boxes = schedule.noise_to_boxes(x, images_whwh, c.snr_scale)
        if box_extract:
            return head.head_extract(self.sd, "head.", feats, x_boxes, t, c.head)
        outputs_class, outputs_coord = head.head_final(self.sd, "head.", feats, x_boxes, t, c.head, cached=cached, memory=mem)
        x_start = schedule.boxes_to_x_start(outputs_coord[-1], images_whwh, c.snr_scale)
        pred_noise = schedule.predict_noise_from_start(self.buf, x, t, x_start)
        return (pred_noise, x_start), outputs_class, outputs_coord

    def forward(self, images):
        """images: dict with cur [1,3,H,W] (padded), image_size (h,w) of the unpadded frame,
        ref_l / ref_g: lists of [1,3,H,W], and the int fields of vid_mega.py:236-248."""
        c = self.cfg
        if images["frame_category"] == 0:                       # :389-401
            self.local_img_queue = []
            self.mem = [None, None]
            self.feats = deque(maxlen=c.all_frame_interval)
            self.classes_300 = deque(maxlen=c.all_frame_interval)
            self.proposals_300 = deque(maxlen=c.all_frame_interval)
            self.proposals_feat_300 = deque(maxlen=c.all_frame_interval)
        frame_id, start_id, end_id = images["frame_id"], images["start_id"], images["end_id"]
        if frame_id % c.infer_batch != 0:                       # :410-412
            self.local_img_queue += images["ref_l"]
            return []
        ref_l = self.local_img_queue + images["ref_l"]
        self.local_img_queue = []
        ref_g = images["ref_g"]
        h, w = images["image_size"]
        whwh = torch.tensor([w, h, w, h], dtype=torch.float32)

        if ref_l or ref_g:                                      # :418-476
            total = torch.cat(list(ref_l) + list(ref_g))
            total = backbone_r101.normalizer(total, c.pixel_mean, c.pixel_std)
            splits = total.split(c.infer_batch)
            feats_split = [self.backbone_fn(s) for s in splits]
            len_l = len(ref_l)
            cls_all, box_all, prop_all, k1_all, k2_all = [], [], [], [], []
            for bi, fs in enumerate(feats_split):
                f = [fs[p] for p in c.in_features]
                B = len(f[0])
                box_init = self.noise_fn("box_init", frame_id, bi, 0, (B, c.num_proposals, 4))
                t = torch.full((B,), 999, dtype=torch.long)
                (cl, bx, pf), k1, k2 = self.model_predictions(f, whwh.unsqueeze(0).expand(B, -1), box_init, t, box_extract=bi + 1)
                cls_all.append(cl)
                box_all.append(bx)
                prop_all.append(pf)
                k1_all.append(k1)
                k2_all.append(k2)
            total_feats = {p: torch.cat([fs[p] for fs in feats_split], dim=0) for p in c.in_features}
            feats_l = {p: total_feats[p][:len_l] for p in c.in_features}
            self.taps["feats"] = total_feats          # p3 / p4 / p5 [n, 256, h, w] of every frame of the call (parity tests)
            classes_t = torch.cat(cls_all, dim=0).view(-1, c.num_proposals, c.num_classes)
            boxes_t = torch.cat(box_all, dim=0).view(-1, c.num_proposals, 4)
            proposals_t = torch.cat(prop_all, dim=1).view(-1, c.num_proposals, c.hidden_dim)
            proposals_t1 = torch.cat(k1_all, dim=0).view(-1, c.head.top_k[0], c.hidden_dim)
            proposals_t2 = torch.cat(k2_all, dim=0).view(-1, c.head.top_k[1], c.hidden_dim)
            classes_all, boxes_all, proposals_all = classes_t[:len_l], boxes_t[:len_l], proposals_t[:len_l]
            proposals_g1, proposals_g2 = proposals_t1[len_l:], proposals_t2[len_l:]
            self.taps["extract"] = (classes_t, boxes_t, proposals_t)

        if ref_g:                                               # :479-488
            m0, _ = memory.update_erase_memory(proposals_g1.reshape(-1, c.hidden_dim), self.mem[0], c.mem_size_test)
            m1, _ = memory.update_erase_memory(proposals_g2.reshape(-1, c.hidden_dim), self.mem[1], c.mem_size_dis)
            self.mem = [m0, m1]

        n_local = len(ref_l)                                    # :491-506
        if images["frame_category"] == 0:
            frame_diff = frame_id - start_id
            lead = c.key_frame_location - frame_diff
            fill_idx = [0] * lead + list(range(n_local)) + [n_local - 1] * (c.all_frame_interval - (lead + n_local))
        else:
            fill_idx = range(n_local)
        for i in fill_idx:
            self.feats.append([feats_l[p][i].unsqueeze(0) for p in c.in_features])
            self.classes_300.append(classes_all[i].unsqueeze(0))
            self.proposals_300.append(boxes_all[i].unsqueeze(0))
            self.proposals_feat_300.append(proposals_all[i])

        batch = min(c.infer_batch, end_id - frame_id + 1)       # :515-523
        r0, r1 = c.key_frame_location, c.key_frame_location + batch
        feats_cur = [torch.cat([self.feats[i][j] for i in range(r0, r1)]) for j in range(len(c.in_features))]
        cached = (torch.cat([self.classes_300[i] for i in range(r0, r1)], dim=0),
                  torch.cat([self.proposals_300[i] for i in range(r0, r1)], dim=0),
                  torch.cat([self.proposals_feat_300[i] for i in range(r0, r1)], dim=0).unsqueeze(0))
        images_whwh = whwh.unsqueeze(0).repeat(batch, 1)

        pairs = schedule.time_pairs(self.num_timesteps, c.sample_step)   # :536-539
        img = self.noise_fn("img", frame_id, 0, 0, (batch, c.num_proposals, 4))
        ensemble = []
        for step, (time, time_next) in enumerate(pairs):
            t = torch.full((batch,), time, dtype=torch.long)
            self.taps[f"img_{step}"] = img
            (pred_noise, x_start), outputs_class, outputs_coord = self.model_predictions(
                feats_cur, images_whwh, img, t, cached=cached, mem=self.mem)
            self.taps[f"final_{step}"] = (outputs_class[-1], outputs_coord[-1])
            if time_next < 0:                                   # :573-575 (the renewed lists are never read again)
                continue
            noise = [self.noise_fn("ddim", frame_id, step, i, (c.num_proposals, 4)) for i in range(batch)]
            fresh = [self.noise_fn("renew", frame_id, step, i, (c.num_proposals, 4)) for i in range(batch)]
            img = renew_and_ddim_step(self.buf, outputs_class[-1], pred_noise, x_start, time, time_next, noise, fresh)
            if c.sample_step > 1:                               # :598-604
                cands = [postproc.topk_candidates(outputs_class[-1][b], outputs_coord[-1][b], c.num_classes)[:3]
                         for b in range(batch)]
                ensemble.append(cands)

        if c.sample_step > 1:                                   # :607-627
            return postproc.inference_ensemble(ensemble, (w, h), c.use_nms)
        return postproc.inference_x1(outputs_class[-1], outputs_coord[-1], (w, h), c.num_classes, c.use_nms)
