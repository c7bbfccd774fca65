"""DiffusionDet meta-architecture -- host-side mirror of
mega_core/modeling/detector/diffusion_det.py:188-896 (inference only) driving libdvid_hip.

Surface kept for `tools/test_net.py`-style callers: `DiffusionDet(cfg)`, `nn.Module` with the
reference's state_dict names (backbone.bottom_up.*, backbone.fpn_*, head.*, diffusion buffers),
`.to(device)`, `.eval()`, `forward(images: dict, targets=None) -> list[BoxList]` with the input
keys of vid_mega.py:236-248, `[]` on calls whose frame_id is not a multiple of INPUT.INFER_BATCH,
ValueError for targets at test time, AssertionError for degenerate predicted boxes.

The video state machine (per-video reset, local frame queue, once-per-video global memory,
DDIM loop, ensemble) is sequenced here; every numerical step is a HIP kernel launched through
`ops` on the current stream.  Random draws go through `self.noise_fn(kind, frame_id, step, image,
shape)` when set (parity/bench), else torch.randn on the device like the reference.
"""
import math
import os
import time
from collections import deque

import torch
from torch import nn

from ... import ops
from ...structures.bounding_box import BoxList
from ...structures.image_list import ImageList, to_image_list
from ...utils import synthetic
from ..roi_heads.box_head.box_head import DynamicHead

_DEPTH_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
# size2config of mega_core/modeling/backbone/swintransformer.py:655-712 (window-7 variants; head dim 32 in all of them)
_SWIN_SIZES = {
    "T": dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=7),
    "S": dict(embed_dim=96, depths=(2, 2, 18, 2), heads=(3, 6, 12, 24), window=7),
    "B": dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), window=7),
    "B-22k": dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), window=7),
    "L-22k": dict(embed_dim=192, depths=(2, 2, 18, 2), heads=(6, 12, 24, 48), window=7),
}


def cosine_beta_schedule(timesteps, s=0.008):
    """cosine schedule, fp64 (diffusion_det.py:50-61)."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def _register_nested(root, name, tensor, buffer=False):
    """Registers `tensor` under the dotted `name` so state_dict keys match the reference's."""
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, nn.Module())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


_FOREIGN_BOXLIST = {}


def _result_class(image_list):
    """BoxList class that belongs to the caller's ImageList: `<pkg>.structures.image_list.ImageList` ->
    `<pkg>.structures.bounding_box.BoxList` (the reference keeps both under mega_core/structures); this repo's class for
    tensors, this repo's ImageList, or when no such module exists."""
    mod = type(image_list).__module__ or ""
    if not mod.endswith(".image_list") or mod.startswith("diffusionvid_amd."):
        return BoxList
    if mod not in _FOREIGN_BOXLIST:
        cls = BoxList
        try:
            import importlib
            cls = getattr(importlib.import_module(mod[: -len("image_list")] + "bounding_box"), "BoxList", BoxList)
        except ImportError:
            pass
        _FOREIGN_BOXLIST[mod] = cls
    return _FOREIGN_BOXLIST[mod]


class DiffusionDet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        d = cfg.MODEL.DiffusionDet
        mega = cfg.MODEL.VID.MEGA
        self.global_enable = mega.GLOBAL.ENABLE
        self.all_frame_interval = mega.ALL_FRAME_INTERVAL
        self.key_frame_location = mega.KEY_FRAME_LOCATION
        self.local_box_enable = cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE
        self.mem_management_size_test = mega.MEMORY_MANAGEMENT_SIZE_TEST
        self.in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        self.num_classes = d.NUM_CLASSES
        self.num_proposals = d.NUM_PROPOSALS
        self.hidden_dim = d.HIDDEN_DIM
        self.num_heads = d.NUM_HEADS
        self.infer_batch = cfg.INPUT.INFER_BATCH
        self.lookahead = max(1, int(getattr(cfg.INPUT, "LOOKAHEAD_BATCHES", 1)))
        self.size_divisibility = 32
        # The reference's global precision switch (mega_core/config/defaults.py:582: "float32" unless the command line says
        # `DTYPE float16`, README.md:86-96; tools/test_net.py:97-98 turns apex amp on for float16 only).  float16 = fp16 storage /
        # fp16 MFMA (the headline path), float32 = fp32 storage / fp32 MFMA (csrc/f32.hip).  Anything else is refused: no value of
        # this key silently selects another precision.
        self.dtype = str(cfg.DTYPE)
        if self.dtype not in ops.PRECISIONS:
            raise NotImplementedError("DTYPE %r: the MI355X path builds float16 and float32" % (self.dtype,))
        if self.lookahead > 1:
            # The look-ahead schedule finishes later batches inside the group's call: that equals the reference only when
            # a batch is exactly one full local queue starting at its key frame, and the global memory is final after the
            # video's first call (later `ref_g` deliveries would otherwise be missed by the batches finished early).
            ok = (self.key_frame_location == 0 and self.all_frame_interval == self.infer_batch == mega.MAX_OFFSET + 1
                  and (not mega.GLOBAL.ENABLE or mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST))
            if not ok:
                raise NotImplementedError(
                    "INPUT.LOOKAHEAD_BATCHES > 1 needs KEY_FRAME_LOCATION 0, ALL_FRAME_INTERVAL == INFER_BATCH == MAX_OFFSET + 1 "
                    "and GLOBAL.STOP_UPDATE_AFTER_INIT_TEST True (got %d, %d, %d, %d, %s)"
                    % (self.key_frame_location, self.all_frame_interval, self.infer_batch, mega.MAX_OFFSET,
                       mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST))
        if list(self.in_features) != ["p3", "p4", "p5"]:
            raise NotImplementedError("ROI_HEADS.IN_FEATURES must be [p3, p4, p5] (configs/vid_*_DiffusionVID.yaml)")
        self.swin = None
        if cfg.MODEL.BACKBONE.NAME == "build_swintransformer_fpn_backbone":
            if cfg.MODEL.SWIN.SIZE not in _SWIN_SIZES:
                raise NotImplementedError("Swin size %r is not built (window-7 sizes only)" % cfg.MODEL.SWIN.SIZE)
            if tuple(cfg.MODEL.SWIN.OUT_FEATURES) != (1, 2, 3):
                raise NotImplementedError("MODEL.SWIN.OUT_FEATURES must be (1, 2, 3)")
            self.swin = dict(getattr(cfg.MODEL.SWIN, "CONFIG_OVERRIDE", None) or _SWIN_SIZES[cfg.MODEL.SWIN.SIZE])
            self.res_blocks = (0, 0, 0, 0)
        elif cfg.MODEL.BACKBONE.NAME == "build_resnet_fpn_backbone":
            if cfg.MODEL.RESNETS.DEPTH not in _DEPTH_BLOCKS or cfg.MODEL.RESNETS.STRIDE_IN_1X1:
                raise NotImplementedError("ResNet depth %s / STRIDE_IN_1X1 unsupported" % cfg.MODEL.RESNETS.DEPTH)
            self.res_blocks = getattr(cfg.MODEL.RESNETS, "BLOCKS_OVERRIDE", None) or _DEPTH_BLOCKS[cfg.MODEL.RESNETS.DEPTH]
        else:
            raise NotImplementedError("backbone '%s' is not built" % cfg.MODEL.BACKBONE.NAME)

        # diffusion constants (diffusion_det.py:222-267); registered so checkpoints load unchanged
        timesteps = 1000
        betas = cosine_beta_schedule(timesteps)
        alphas = 1.0 - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0).to(torch.float32)
        alphas_cumprod_prev = torch.nn.functional.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
        self.num_timesteps = timesteps
        self.sampling_timesteps = d.SAMPLE_STEP
        self.skip_unobservable = bool(getattr(d, "SKIP_UNOBSERVABLE", False)) and d.SAMPLE_STEP > 1
        assert self.sampling_timesteps <= timesteps
        self.ddim_sampling_eta = 1.0
        self.scale = d.SNR_SCALE
        self.box_renewal = True
        self.use_ensemble = True
        self.use_nms = d.USE_NMS
        self.use_focal = d.USE_FOCAL
        if not self.use_focal:
            raise NotImplementedError("only the focal-loss (sigmoid) inference branch is built (USE_FOCAL: True)")
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        for name, val in (
                ("betas", betas), ("alphas_cumprod", alphas_cumprod), ("alphas_cumprod_prev", alphas_cumprod_prev),
                ("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod)),
                ("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod)),
                ("log_one_minus_alphas_cumprod", torch.log(1.0 - alphas_cumprod)),
                ("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod)),
                ("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod - 1)),
                ("posterior_variance", posterior_variance),
                ("posterior_log_variance_clipped", torch.log(posterior_variance.clamp(min=1e-20))),
                ("posterior_mean_coef1", betas * torch.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod)),
                ("posterior_mean_coef2", (1.0 - alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - alphas_cumprod))):
            self.register_buffer(name, val)

        # parameters under the reference's names; seeded random init (the reference also starts from
        # random init before DetectronCheckpointer.load)
        sd = synthetic.make_state_dict(0, blocks=self.res_blocks, swin=self.swin, hidden=d.HIDDEN_DIM, nheads=d.NHEADS,
                                       dim_ff=d.DIM_FEEDFORWARD, dim_dynamic=d.DIM_DYNAMIC, num_classes=d.NUM_CLASSES,
                                       num_cls=d.NUM_CLS, num_reg=d.NUM_REG, num_heads=d.NUM_HEADS,
                                       num_heads_cond=d.NUM_HEADS_LOCAL, pooler=cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION,
                                       prior_prob=d.PRIOR_PROB)
        self.head = DynamicHead(cfg, None, engine_provider=self._get_engine)
        self.num_heads_local = self.head.num_heads_local
        self.top_k = self.head.top_k
        for name, t in sd.items():
            _register_nested(self, name, t, buffer=name.endswith(("running_mean", "running_var")))
        self._engine = None
        self._ac_host = alphas_cumprod.clone()      # host copies: schedule lookups must not sync the device
        self._sr_host = torch.sqrt(1.0 / alphas_cumprod)
        self._srm1_host = torch.sqrt(1.0 / alphas_cumprod - 1)
        self.noise_fn = None
        self.after_first_launch = None      # optional callable, run once the first backbone launch of a call is queued
        self.memory_on_side_stream = True          # (attribute, not an environment switch: tests compare both settings)
        self._mem_stream = None
        self._mem_stream2 = None
        self._mem_side_pending = False
        self._mem_side_inputs = None
        self.debug_taps = None      # dict -> receives intermediates (parity tests)
        # True: a batch's detections come back with ONE device->host copy and the BoxLists hold CPU tensors
        # (what engine/inference.py does next anyway, there with ~3 copies per frame); False: GPU tensors
        self.results_on_host = False
        # class of the returned detections.  None: this repo's BoxList -- unless the frames arrive in another package's
        # ImageList (the reference's collator, collate_batch.py:24-35), in which case the detections are that package's
        # own `structures.bounding_box.BoxList`, so its evaluator / torch.save see their own type (`_result_class`).
        self.boxlist_cls = None
        # hipGraph replay of the steady-state call of the reference's own protocol (one INFER_BATCH batch per call, no look-ahead):
        # see _graphed_call.  DVID_CALL_GRAPH=0 / `use_call_graph = False` launches every call kernel by kernel (A/B, profiling).
        self.use_call_graph = os.environ.get("DVID_CALL_GRAPH", "1") != "0"
        self._graphs, self._graph_seen, self._graph_generation = {}, {}, -1
        # the streaming mode's steady call as a graph (_stream_graph_applies): built, bit-identical, and NOT faster -- 247 against 252 frames/s
        # (profiles/r05n_streaming_graph_ab.txt): that call is 4 ms of kernels (two-frame launches, two dependent farthest-point sweeps), the
        # host's launching hides behind them.  Off unless a caller sets the attribute (tests do).
        self.use_stream_graph = False
        self.graph_replays = 0
        self.host_wait_s = 0.0      # seconds this process spent blocked in the per-batch device->host result copy
        self.video_index = 0
        self.demo = False

    # ---- engine (repacked weights on the GPU) -------------------------------------------------
    def _get_engine(self):
        if self._engine is None:
            d = self.cfg.MODEL.DiffusionDet
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = ops.Model(
                sd, hidden_dim=d.HIDDEN_DIM, nheads=d.NHEADS, dim_feedforward=d.DIM_FEEDFORWARD, dim_dynamic=d.DIM_DYNAMIC,
                num_classes=d.NUM_CLASSES, num_cls=d.NUM_CLS, num_reg=d.NUM_REG, num_heads=d.NUM_HEADS,
                num_heads_cond=d.NUM_HEADS_LOCAL, pooler_resolution=self.cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION,
                sampling_ratio=self.cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, res_blocks=tuple(self.res_blocks),
                pixel_mean=tuple(self.cfg.MODEL.PIXEL_MEAN), pixel_std=tuple(self.cfg.MODEL.PIXEL_STD), precision=self.dtype,
                **({} if self.swin is None else dict(backbone="swin", swin_embed_dim=self.swin["embed_dim"],
                                                      swin_depths=tuple(self.swin["depths"]), swin_heads=tuple(self.swin["heads"]),
                                                      swin_window=self.swin["window"])))
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        """`.to()` / `.cuda()` move parameters through here: keep `self.device` (noise, time steps, the engine's
        device) in step with them, and drop an engine built for another device."""
        out = super()._apply(fn, *args, **kwargs)
        try:
            dev = next(self.parameters()).device
        except StopIteration:
            return out
        if dev != self.device:
            self.device = dev
            if self._engine is not None:
                self._engine.close()
                self._engine = None
        return out

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self._engine is not None:           # weights changed: repack on next use
            self._engine.close()
            self._engine = None
        self._graphs, self._graph_seen = {}, {}          # captured launches point at the old engine's weights
        return out

    def _noise(self, kind, frame_id, step, image, shape):
        if getattr(self.noise_fn, "on_device", False):          # counter-based draws generated by a kernel (synthetic.DeviceNoise)
            return self.noise_fn.draw(kind, frame_id, step, image, 1, shape, self.device)[0]
        if self.noise_fn is not None:
            return self.noise_fn(kind, frame_id, step, image, shape).to(self.device, torch.float32)
        return torch.randn(shape, device=self.device)

    def _noise_images(self, kind, frame_id, step, batch, shape):
        """[batch, *shape]: one draw per image of the batch, uploaded as ONE tensor (an injected noise_fn draws on the host)"""
        if getattr(self.noise_fn, "on_device", False):          # consecutive images have consecutive keys: one launch
            return self.noise_fn.draw(kind, frame_id, step, 0, batch, shape, self.device)
        if self.noise_fn is not None:
            host = torch.stack([self.noise_fn(kind, frame_id, step, i, shape) for i in range(batch)])
            return host.to(self.device, torch.float32)
        return torch.randn((batch,) + tuple(shape), device=self.device)

    def _ddim_draws(self, batch, frame_id, pairs):
        """every random draw of the DDIM loop of one call (diffusion_det.py:542,:587,:595), made before any kernel of the
        call is queued -- see the note on uploads in _forward_test"""
        M = self.num_proposals
        draws = {"img": self._noise("img", frame_id, 0, 0, (batch, M, 4))}
        for step, (time, time_next) in enumerate(pairs):
            if time_next >= 0:
                draws[step] = (self._noise_images("ddim", frame_id, step, batch, (M, 4)),
                               self._noise_images("renew", frame_id, step, batch, (M, 4)))
        return draws

    # ---- forward (diffusion_det.py:306-336) ---------------------------------------------------
    def forward(self, images, targets=None):
        if self.training:
            raise NotImplementedError("training is out of scope of the MI355X inference path")
        images = dict(images)
        self._result_cls = self.boxlist_cls or _result_class(images["cur"])
        images["cur"] = to_image_list(images["cur"])
        images["ref_l"] = [to_image_list(image) for image in images["ref_l"]]
        images["ref_g"] = [to_image_list(image) for image in images["ref_g"]]
        infos = dict(images)
        infos.pop("cur")
        if self.device.type == "cuda" and self.device.index is not None:
            # the library allocates and launches on the current device: scoped to this call, the caller's current device
            # is restored on return
            with torch.cuda.device(self.device):
                return self._forward_test(images["cur"], infos, targets)
        return self._forward_test(images["cur"], infos, targets)

    def _reset_video(self):
        self._ahead = {}
        self._results_ahead = {}
        n = self.all_frame_interval
        self.local_img_queue = []
        self._set_global_memory([None, None])
        self.head.proposal_feats_local = [None, None]
        self.queue = deque(maxlen=n)      # entries: (split_outputs, frame index inside the split)
        self.video_index += 1

    def _set_global_memory(self, memory):
        """every assignment of the video's global memory goes through here: the engine's cached K/V projections of the old
        memory are dropped explicitly"""
        self.head.proposal_feats_global = list(memory)
        if self._engine is not None:
            self._engine.invalidate_memory()

    # ---- one video over several ranks (engine/inference.py: compute_on_video_sharded) -------------------------
    def global_memory(self):
        """[memory 900 x d, memory 150 x d] of the current video (diffusion_det.py:479-488)"""
        return list(self.head.proposal_feats_global)

    def global_memory_shapes(self):
        g = self.cfg.MODEL.VID.MEGA.GLOBAL.SIZE
        return [(min(self.mem_management_size_test, g * self.top_k[0]), self.hidden_dim), (min(150, g * self.top_k[1]), self.hidden_dim)]

    def adopt_video_memory(self, memory):
        """Start a video whose global memory was built by another rank: per-video reset, then the memory as if this
        process had seen the global frames."""
        self._reset_video()
        self._set_global_memory([m.to(self.device, torch.float32) for m in memory])

    def model_predictions(self, backbone_feats, images_whwh, x, t, box_extract=0):
        """diffusion_det.py:655-677.  images_whwh: (w, h) of the un-padded frame (same for all frames)."""
        w, h = images_whwh
        x_boxes = ops.noise_to_boxes(x, self.scale, w, h)
        if box_extract:
            return self.head(backbone_feats, x_boxes, t, None, box_extract)
        outputs_class, outputs_coord = self.head(backbone_feats, x_boxes, t, None)
        return outputs_class, outputs_coord

    def _forward_test(self, imgs, infos, targets=None):
        if targets is not None and not self.demo:
            raise ValueError("In testing mode, targets should be None")
        if infos["frame_category"] == 0:
            self._reset_video()
        frame_id, start_id, end_id = infos["frame_id"], infos["start_id"], infos["end_id"]
        if frame_id % self.infer_batch != 0:
            self.local_img_queue += infos["ref_l"]
            return []
        if frame_id in self._results_ahead:          # this batch was finished inside its look-ahead group's call
            self.local_img_queue = []
            return self._results_ahead.pop(frame_id)
        ref_l = self.local_img_queue + infos["ref_l"]
        self.local_img_queue = []
        ref_g = infos["ref_g"]
        h, w = imgs.image_sizes[0]
        whwh = (float(w), float(h))
        batch = min(self.infer_batch, end_id - frame_id + 1)
        pairs = self._time_pairs()
        ahead = infos.get("ref_ahead") or {}
        ahead_keys = sorted(k for k in ahead if k not in self._ahead)
        nb_of = {frame_id: batch}
        nb_of.update({fb: min(self.infer_batch, end_id - fb + 1) for fb in ahead_keys})
        # DDIM draws of every batch finished in this call, keyed and shaped exactly as that batch's own call would draw them
        ddim_draws = {fb: self._ddim_draws(nb, fb, pairs) for fb, nb in nb_of.items()} if self.sampling_timesteps > 1 else {}

        # 0. the steady-state call of the reference's protocol as ONE hipGraph launch (see _graphed_call)
        if self._call_graph_applies(infos, ref_l, ref_g, ahead, batch, frame_id):
            out = self._graphed_call(frame_id, ref_l, whwh, w, h, pairs, batch, ddim_draws.get(frame_id))
            if out is not None:
                return out
        elif self._stream_graph_applies(infos, ref_l, ref_g, ahead, batch, frame_id):
            out = self._graphed_call(frame_id, ref_l, whwh, w, h, pairs, batch, ddim_draws.get(frame_id), ref_g=ref_g)
            if out is not None:
                return out

        # 1. features + extraction pass over [local frames | global frames] (+ the look-ahead batches)
        local_split = self._ahead.pop(frame_id, None)
        if ref_g:
            local_split = None          # a call that delivers global frames re-extracts its local frames with them
        ref_l_run = [] if local_split is not None else ref_l
        if len(ref_l_run) > self.infer_batch:
            raise NotImplementedError("more local frames than INFER_BATCH in one call")
        gsplit = None
        if ref_l_run or ref_g or ahead_keys:
            fresh_local, gsplit, fresh_ahead = self._extract(frame_id, ref_l_run, ref_g, {fb: ahead[fb] for fb in ahead_keys}, whwh,
                                                             on_global=self._build_memory_aside if ref_g else None)
            if fresh_local is not None:
                local_split = fresh_local
            self._ahead.update(fresh_ahead)
        splits = [local_split]

        # 2. global memory, once per video with the shipped config (diffusion_det.py:479-488).  When more launch sequences followed
        # the one that held the global frames, _extract has already queued it on the side stream (1800 x 1800 distances and two
        # greedy farthest-point passes = 899 + 149 dependent arg-max steps on ONE workgroup, ~3 ms beside which the rest of the
        # chip would idle); everything after this point reads the memory, so the launch stream waits for it here.
        if ref_g:
            if self._mem_side_pending:
                torch.cuda.current_stream().wait_stream(self._mem_stream)
                self._mem_side_pending = False
                self._mem_side_inputs = None          # the side stream's reads are ordered before everything queued from here on
            else:
                self._build_memory(gsplit)

        # 3. local queue (diffusion_det.py:491-506)
        n_local = len(ref_l)
        if infos["frame_category"] == 0:
            lead = self.key_frame_location - (frame_id - start_id)
            fill_idx = [0] * lead + list(range(n_local)) + [n_local - 1] * (self.all_frame_interval - (lead + n_local))
        else:
            fill_idx = list(range(n_local))
        for i in fill_idx:
            self.queue.append((splits[0], i))

        # 4. final stage.  Per frame it needs the frame's own extraction results and the video's global memory, nothing
        # of its neighbours, so the batches extracted ahead in this call are finished here as well -- in one pass per
        # run of adjacent frames -- and their detections wait on the host side for their own calls.
        joinable = n_local == self.infer_batch and batch == self.infer_batch and splits[0].get("_src") is not None
        group = [(fb, nb_of[fb], self._ahead.pop(fb)) for fb in ahead_keys if fb in self._ahead] if joinable else []
        if group:
            group = [(frame_id, batch, splits[0])] + group
            runs = []
            for fb, nb, sp in group:
                if runs and runs[-1]["src"] is sp.get("_src") and runs[-1]["i1"] == sp.get("_i0"):
                    runs[-1]["i1"] = sp["_i1"]
                    runs[-1]["items"].append((fb, nb))
                else:
                    runs.append({"src": sp.get("_src"), "i0": sp.get("_i0"), "i1": sp.get("_i1"), "split": sp, "items": [(fb, nb)]})
            mine = None
            for run in runs:
                if run["src"] is not None:
                    src, i0, i1 = run["src"], run["i0"], run["i1"]
                    feats_run = [f[i0:i1] for f in src["feats"]]
                    cached = (src["logits"][i0:i1], src["boxes"][i0:i1], src["obj"][i0:i1])
                else:
                    sp = run["split"]
                    feats_run, cached = sp["feats"], (sp["logits"], sp["boxes"], sp["obj"])
                per_slot = self._final_stage(feats_run, cached, whwh, w, h, pairs, run["items"], ddim_draws)
                for k, (fb, nb) in enumerate(run["items"]):
                    out = per_slot[k * self.infer_batch: k * self.infer_batch + nb]
                    if fb == frame_id:
                        mine = out
                    else:
                        self._results_ahead[fb] = out
            return mine

        # current batch only (diffusion_det.py:515-523)
        entries = [self.queue[i] for i in range(self.key_frame_location, self.key_frame_location + batch)]
        feats_cur, cached = self._gather_entries(entries)
        return self._final_stage(feats_cur, cached, whwh, w, h, pairs, [(frame_id, batch)], ddim_draws, slots=batch)

    @staticmethod
    def _fire_on_global(on_global, take, len_l, n_own, done, total):
        """Once the launch sequences so far cover every global frame and at least one more sequence follows: hand the global
        frames' split over (-> True when the hook has run, so it runs once)."""
        if done < n_own or done >= total:
            return done >= n_own          # nothing follows: the caller builds the memory in line
        on_global(take(len_l, n_own, keys=("k1", "k2"), with_feats=False))          # the memory build reads k1 / k2 only
        return True

    def _build_memory(self, gsplit):
        g1 = gsplit["k1"].reshape(-1, self.hidden_dim)
        g2 = gsplit["k2"].reshape(-1, self.hidden_dim)
        # The two memories (900 / 150 rows) are pruned independently, each by a sweep of dependent arg-max steps on ONE workgroup:
        # the short one runs on a second stream beside the long one (same kernels, same inputs -> same memories).
        if self.memory_on_side_stream:
            if self._mem_stream2 is None:
                self._mem_stream2 = torch.cuda.Stream(device=self.device)
            cur = torch.cuda.current_stream()
            self._mem_stream2.wait_stream(cur)
            with torch.cuda.stream(self._mem_stream2):
                m1, _ = ops.update_erase_memory(g2, self.head.proposal_feats_global[1], 150)
            m0, _ = ops.update_erase_memory(g1, self.head.proposal_feats_global[0], self.mem_management_size_test)
            cur.wait_stream(self._mem_stream2)
            g2.record_stream(self._mem_stream2)          # read on the second stream: its block is not recycled under that read
            m1.record_stream(cur)                        # allocated on the second stream, read on this one from here on
        else:
            m0, _ = ops.update_erase_memory(g1, self.head.proposal_feats_global[0], self.mem_management_size_test)
            m1, _ = ops.update_erase_memory(g2, self.head.proposal_feats_global[1], 150)
        self._set_global_memory([m0, m1])
        if self.debug_taps is not None:
            self.debug_taps["memory"] = [m0, m1]

    def _build_memory_aside(self, gsplit):
        """The memory build on a second stream, behind everything queued so far (the global frames' extraction) and beside what
        the launch stream queues next; same kernels on the same inputs, so the same memory."""
        if not self.memory_on_side_stream:
            return False
        if self._mem_stream is None:
            self._mem_stream = torch.cuda.Stream(device=self.device)
        self._mem_stream.wait_stream(torch.cuda.current_stream())
        # The split may be fresh concatenations allocated on the launch stream (global frames straddling two launch
        # sequences): they stay referenced until the launch stream has waited for the side stream, so the caching allocator
        # cannot hand their blocks to the next sequence's allocations while the side stream still reads them.
        self._mem_side_inputs = gsplit
        with torch.cuda.stream(self._mem_stream):
            self._build_memory(gsplit)
        self._mem_side_pending = True
        return True

    # ---- the steady-state call as a hipGraph ------------------------------------------------------------------------------
    def _call_graph_applies(self, infos, ref_l, ref_g, ahead, batch, frame_id):
        """A call that runs one full batch through the whole per-call pipeline and nothing else: the shipped protocol
        (KEY_FRAME_LOCATION 0, ALL_FRAME_INTERVAL == INFER_BATCH, memory final after the video's first call), not the video's
        first call, no global frames, no look-ahead hand-over, every frame a resident fp32 tensor of one size, memory present."""
        if not self.use_call_graph or self.lookahead != 1 or self.debug_taps is not None or self.demo:
            return False
        mega = self.cfg.MODEL.VID.MEGA
        if not (self.key_frame_location == 0 and self.all_frame_interval == self.infer_batch == mega.MAX_OFFSET + 1
                and self.global_enable and mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST):
            return False
        if infos["frame_category"] == 0 or ref_g or ahead or frame_id in self._ahead or self._mem_side_pending:
            return False
        if len(ref_l) != self.infer_batch or batch != self.infer_batch or self.head.proposal_feats_global[0] is None:
            return False
        f0 = ref_l[0].tensors
        return all(f.tensors.is_cuda and f.tensors.device == self.device and f.tensors.dtype == torch.float32
                   and f.tensors.shape == f0.shape and f.tensors.shape[0] == 1 for f in ref_l)

    def _stream_graph_applies(self, infos, ref_l, ref_g, ahead, batch, frame_id):
        """The steady-state call of the STREAMING mode (demo/demo.py:60-68: INFER_BATCH 1, one new global frame per call, the memory
        updated and pruned back on every call, GLOBAL.STOP_UPDATE_AFTER_INIT_TEST False) once both memories have reached their full
        size: backbone + extraction on [the frame | the new global frame], both memory updates (merge, distances, farthest-point sweep,
        gather), K / V projection, final stage, post-processing -- ~200 launches whose shapes no longer change.  Such a call is 3-4 ms
        of kernels beside ~1 ms of host-side launching with a result hand-over per call (bench.py: host_blocked_on_gpu_frac 0.76).
        Measured: the replay is bit-identical and no faster (see `use_stream_graph`): the kernels, not the launching, are the 4 ms."""
        if not self.use_call_graph or not self.use_stream_graph or self.lookahead != 1 or self.debug_taps is not None or self.demo:
            return False
        mega = self.cfg.MODEL.VID.MEGA
        if not (self.key_frame_location == 0 and self.all_frame_interval == self.infer_batch == 1 and self.global_enable
                and not mega.GLOBAL.STOP_UPDATE_AFTER_INIT_TEST):
            return False
        if infos["frame_category"] == 0 or ahead or frame_id in self._ahead or self._mem_side_pending or len(ref_l) != 1 or len(ref_g) != 1 or batch != 1:
            return False
        mem = self.head.proposal_feats_global
        if mem[0] is None or mem[1] is None or mem[0].shape[0] != self.mem_management_size_test or mem[1].shape[0] != 150:
            return False          # the memories still grow: shapes change from call to call
        f0 = ref_l[0].tensors
        return all(f.tensors.is_cuda and f.tensors.device == self.device and f.tensors.dtype == torch.float32
                   and f.tensors.shape == f0.shape and f.tensors.shape[0] == 1 for f in list(ref_l) + list(ref_g))

    def _graphed_call(self, frame_id, ref_l, whwh, w, h, pairs, batch, draws, ref_g=None):
        """One batch of the reference's call protocol (mega_core/engine/inference.py:22-94: 8 frames per working call) is ~180
        kernel launches of 10-90 us -- backbone, 3 + 1 heads, attention, post-processing -- that differ from call to call only in
        their INPUT VALUES: the frames and the random draws.  The first steady call of a shape runs kernel by kernel (tuner,
        workspace growth, scale/shift rows); the second is captured into a hipGraph (torch.cuda.CUDAGraph on the launch stream:
        the library launches on torch's current stream, so its kernels are captured like torch's own) reading static input
        buffers; every later call copies its frames and draws into those buffers and replays the graph: one launch instead of
        ~180, same kernels, same arguments, same order -- bit-identical detections (tests/test_gpu_e2e.py::
        test_call_graph_replay_is_bit_identical).  Keyed by (frame shape, batch, DDIM steps, memory rows, engine); a graph is
        dropped with its engine (load_state_dict).  Returns None when this call ran (or must run) the ordinary way."""
        frames = [im.tensors for im in ref_l]
        mem = self.head.proposal_feats_global
        # A captured launch holds raw addresses inside the engine's workspace, and the workspace moves when it grows (a video of a larger
        # frame size, a memory with more rows): every graph captured before such a move is dropped, never replayed into freed memory.
        eng = self._get_engine()          # (may not exist yet: a rank that adopted its video's memory reaches its first full batch without
        gen = eng.workspace_generation()  #  having run anything, and a device move drops the engine; _get_engine builds it)
        if gen != self._graph_generation:
            self._graphs, self._graph_seen, self._graph_generation = {}, {}, gen
        # ... and it bakes in every switch the eager path reads per call
        gframes = [im.tensors for im in ref_g] if ref_g else []
        key = (tuple(frames[0].shape), len(frames), self.sampling_timesteps, int(mem[0].shape[0]), int(mem[1].shape[0]) if mem[1] is not None else 0,
               id(eng), (float(w), float(h)), bool(self.skip_unobservable), bool(self.use_nms), eng.chains, eng.precision, len(gframes))
        M = self.num_proposals
        g = self._graphs.get(key)
        if g is None:
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if self._graph_seen[key] < 2:
                return None                      # the first steady call of this shape: eager (it also warms everything a capture may not do)
            g = self._capture_call(key, frames, whwh, w, h, pairs, batch, draws, gframes)
            if g is None:
                return None
        # this call's inputs into the graph's static buffers (device-to-device, on the launch stream), then ONE launch
        torch._foreach_copy_(g["frames"], frames)
        g["box_init"].copy_(self._noise("box_init", frame_id, 0, 0, (batch, M, 4)))
        if gframes:
            # streaming: the new global frame and its draw (split 1 of this call); the memories live in the graph's static buffers from
            # one replay to the next -- after an eager call (a new video's first calls) they are brought up to date first
            torch._foreach_copy_(g["gframes"], gframes)
            g["box_init_g"].copy_(self._noise("box_init", frame_id, 1, 0, (len(gframes), M, 4)))
            for i in range(2):
                if mem[i] is not g["mem"][i]:
                    g["mem"][i].copy_(mem[i])
            self.head.proposal_feats_global = list(g["mem"])
        if draws is not None:
            g["draws"]["img"].copy_(draws["img"])
            for step, pair in g["draws"].items():
                if step != "img":
                    pair[0].copy_(draws[step][0])
                    pair[1].copy_(draws[step][1])
        if self.after_first_launch is not None:
            self.after_first_launch()
        # the captured launches read the memory's K / V projections from the engine's buffers; projecting is not part of the capture
        # (a memory adopted from another rank, or replaced in place, has not been through global_xattn yet) -- except in the streaming
        # graph, whose memory changes inside the capture and is projected there
        if not gframes:
            self._get_engine().ensure_memory_projected(mem[0])
        g["graph"].replay()
        if gframes:
            self._get_engine().invalidate_memory()          # the projection buffers hold this replay's memory; no tensor version says so
        self.graph_replays += 1
        self.local_img_queue = []
        # the local queue as the eager path leaves it (diffusion_det.py:491-506): this call's frames, whose extraction results are the
        # graph's static buffers -- an eager call that follows never reads an older call's entries
        for i in range(batch):
            self.queue.append((g["local"], i))
        out = g["out"]
        if not self.results_on_host:
            # the BoxLists would be views of the graph's static output buffer, which the next replay overwrites: hand out copies
            # (results_on_host copies the buffer to the host anyway)
            out = tuple(t.clone() for t in out)
        return self._to_boxlists(*out, (int(w), int(h)))

    def _capture_call(self, key, frames, whwh, w, h, pairs, batch, draws, gframes=()):
        M = self.num_proposals
        static = {"frames": [torch.empty_like(f) for f in frames], "box_init": torch.empty((batch, M, 4), device=self.device),
                  "draws": None}
        if gframes:
            static["gframes"] = [torch.empty_like(f) for f in gframes]
            static["box_init_g"] = torch.empty((len(gframes), M, 4), device=self.device)
            static["mem"] = [m.clone() for m in self.head.proposal_feats_global]
        if draws is not None:
            static["draws"] = {k: (torch.empty_like(v) if k == "img" else (torch.empty_like(v[0]), torch.empty_like(v[1]))) for k, v in draws.items()}
        static_l = [ImageList(f, im.image_sizes) for f, im in zip(static["frames"], [to_image_list(f) for f in frames])]
        hook, self.after_first_launch = self.after_first_launch, None

        static_g = [ImageList(f, im.image_sizes) for f, im in zip(static.get("gframes", []), [to_image_list(f) for f in gframes])]
        side_mem, self.memory_on_side_stream = self.memory_on_side_stream, (self.memory_on_side_stream and not gframes)          # one stream inside the streaming capture
        mem_orig = list(self.head.proposal_feats_global)

        def body():
            if gframes:
                # [frame | new global frame] through backbone + extraction, then both memory updates on the STATIC memory buffers: the
                # pruned memories are written back into them, so that the next replay starts from this one's result
                self.head.proposal_feats_global = list(static["mem"])
                local, gsplit, _ = self._extract(0, static_l, static_g, {}, whwh, box_init=[static["box_init"], static["box_init_g"]])
                self._build_memory(gsplit)
                for i in range(2):
                    static["mem"][i].copy_(self.head.proposal_feats_global[i])
                self._set_global_memory(static["mem"])
            else:
                local, _, _ = self._extract(0, static_l, [], {}, whwh, box_init=[static["box_init"]])
            static["local"] = local
            feats_cur, cached = local["feats"], (local["logits"], local["boxes"], local["obj"])
            return self._final_stage_launch(feats_cur, cached, whwh, w, h, pairs, [(0, batch)], {0: static["draws"]} if static["draws"] is not None else {}, slots=batch)

        try:
            torch._foreach_copy_(static["frames"], frames)
            static["box_init"].normal_()
            if gframes:
                torch._foreach_copy_(static["gframes"], gframes)
                static["box_init_g"].normal_()
                mem_before = [m.clone() for m in static["mem"]]          # the warm-up run and the capture itself must not advance the memory
            if static["draws"] is not None:
                for v in static["draws"].values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        t.normal_()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                body()                               # on the capture's kind of stream once: nothing left to allocate or tune
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):          # other host threads (a data loader's copies) stay legal
                out = body()
            if gframes:
                for i in range(2):
                    static["mem"][i].copy_(mem_before[i])
        except Exception as e:                       # a platform that cannot capture falls back to kernel-by-kernel launches, loudly
            import warnings
            warnings.warn("DiffusionDet: hipGraph capture of the steady-state call failed (%r); calls are launched kernel by kernel" % (e,))
            self.use_call_graph = False
            if gframes:
                self._set_global_memory(mem_orig)          # the streaming body advances the memory: the eager call that follows starts from the old one
            return None
        finally:
            self.after_first_launch = hook
            self.memory_on_side_stream = side_mem
        static["graph"], static["out"] = graph, out
        if self._get_engine().workspace_generation() != self._graph_generation:          # the capture's own warm-up run may have grown the workspace:
            self._graphs, self._graph_seen = {}, {}                        # older graphs go; this one was captured after the move
            self._graph_generation = self._get_engine().workspace_generation()
        self._graphs[key] = static
        return static

    def _extract(self, frame_id, ref_l, ref_g, ahead, whwh, on_global=None, box_init=None):
        """Backbone + the 3 extraction RCNNHeads + top-k feature selection over [local | global | look-ahead] frames.
        Every stage here is per-frame independent, so the reference's splits of INFER_BATCH -- and with
        INPUT.LOOKAHEAD_BATCHES > 1 the frames of the next batches -- run as launches of up to INFER_BATCH *
        LOOKAHEAD_BATCHES frames.  -> (local split | None, global split | None, {batch start: split}); a split is a dict
        of per-frame tensors (feats, logits, boxes, obj, k1, k2), views into the launch's outputs where possible."""
        eng = self._get_engine()
        M = self.num_proposals
        frames = [im.tensors for im in ref_l] + [im.tensors for im in ref_g]
        len_l, n_own = len(ref_l), len(frames)
        # random draws: one (B, M, 4) tensor per reference split `bi` of this call, then one per look-ahead batch (drawn
        # as that batch's own call would: split 0 of call `fb`).  All uploads happen BEFORE any kernel is queued: a
        # host->device copy from pageable memory blocks the host until the stream has drained.
        sizes = [min(self.infer_batch, n_own - a) for a in range(0, n_own, self.infer_batch)]
        noise = list(box_init) if box_init is not None else [self._noise("box_init", frame_id, bi, 0, (b, M, 4)) for bi, b in enumerate(sizes)]
        for fb in sorted(ahead):
            group = [to_image_list(im).tensors for im in ahead[fb]]
            frames += group
            noise.append(self._noise("box_init", fb, 0, 0, (len(group), M, 4)))
        # Frames stay where they are: the backbone reads each through a pointer table (ops.Model.backbone_frames).  Only frames
        # that are not fp32 device tensors of one size (host tensors of a plain DataLoader, other dtypes) are concatenated and
        # moved as the reference does (diffusion_det.py:418-421).
        as_list = all(f.is_cuda and f.device == self.device and f.dtype == torch.float32 and f.shape[0] == 1 and f.shape == frames[0].shape
                      for f in frames)
        total = None if as_list else torch.cat(frames).to(self.device, torch.float32)
        n_total = len(frames) if as_list else total.shape[0]
        fh, fw = frames[0].shape[-2:]
        box_init_all = torch.cat(noise)
        # frames per launch sequence: one look-ahead group (taking the first call's 24 global frames into the same sequence --
        # 128 frames instead of 104 + 24 -- measured 1 % slower, A/B on one box)
        cap = self.infer_batch * self.lookahead
        # A call whose OWN frames exceed a look-ahead group -- the first call of a video without look-ahead (8 local + 24 global frames
        # against groups of 8), every call of the streaming mode (1 local + 1 global frame against groups of 1) -- runs them as ONE
        # launch sequence: the reference's splits of INFER_BATCH (diffusion_det.py:424-476) bound its memory, not its arithmetic; every
        # stage is per-frame independent and the draws stay keyed by the reference's split index.
        starts = list(range(0, n_total, cap)) if n_own <= cap else [0] + list(range(n_own, n_total, cap))
        bounds = list(zip(starts, starts[1:] + [n_total]))
        eng.reserve(max(b - a for a, b in bounds), fh, fw, M)
        per_frame = []          # (launch result dict, index inside the launch) for every frame slot of `total`

        def take(a, b, keys=("logits", "boxes", "obj", "k1", "k2"), with_feats=True):
            """result slots [a, b) as one split: views when they sit in one launch, else a concatenation"""
            runs = []
            for j in range(a, b):
                src, i = per_frame[j]
                if runs and runs[-1][0] is src:
                    runs[-1][2] = i + 1
                else:
                    runs.append([src, i, i + 1])
            if len(runs) == 1:
                src, i0, i1 = runs[0]
                out = {k: src[k][i0:i1] for k in keys}
                if with_feats:
                    out["feats"] = [f[i0:i1] for f in src["feats"]]
                out["_src"], out["_i0"], out["_i1"] = src, i0, i1          # lets neighbouring splits be re-joined as views
                return out
            out = {k: torch.cat([src[k][i0:i1] for src, i0, i1 in runs]) for k in keys}
            if with_feats:
                out["feats"] = [torch.cat([src["feats"][l][i0:i1] for src, i0, i1 in runs]) for l in range(3)]
            return out

        fired = on_global is None or not ref_g
        for ci, (a, a_end) in enumerate(bounds):
            feats = eng.backbone_frames(frames[a:a_end]) if as_list else eng.backbone(total[a:a_end].contiguous())
            if ci == 0 and self.after_first_launch is not None:
                self.after_first_launch()          # e.g. the data layer's prefetch of the next group: behind this call's own uploads
            B = feats[0].shape[0]
            if self.skip_unobservable:
                # x4: of the extraction pass only the backbone features of every frame and the top-k
                # object features of the GLOBAL frames are ever read -- the 3 heads run on the global frames of this chunk only
                g0, g1 = max(len_l, a) - a, min(n_own, a + B) - a
                dev, d = self.device, self.hidden_dim
                res = {"feats": feats, "logits": torch.empty((B, M, self.num_classes), device=dev), "boxes": torch.empty((B, M, 4), device=dev),
                       "obj": torch.empty((B, M, d), device=dev), "k1": torch.empty((B, self.top_k[0], d), device=dev),
                       "k2": torch.empty((B, self.top_k[1], d), device=dev)}
                if g1 > g0:
                    t = torch.full((g1 - g0,), 999, dtype=torch.long)
                    (cl, bx, pf), k1, k2 = self.model_predictions([f[g0:g1] for f in feats], whwh, box_init_all[a + g0:a + g1], t, box_extract=ci + 1)
                    res["k1"][g0:g1] = k1.view(g1 - g0, self.top_k[0], d)
                    res["k2"][g0:g1] = k2.view(g1 - g0, self.top_k[1], d)
                per_frame += [(res, i) for i in range(B)]
                fired = fired or self._fire_on_global(on_global, take, len_l, n_own, a + B, n_total)
                continue
            t = torch.full((B,), 999, dtype=torch.long)
            (cl, bx, pf), k1, k2 = self.model_predictions(feats, whwh, box_init_all[a:a + B], t, box_extract=ci + 1)
            res = {"feats": feats, "logits": cl, "boxes": bx, "obj": pf[0].view(B, M, self.hidden_dim),
                   "k1": k1.view(B, self.top_k[0], self.hidden_dim), "k2": k2.view(B, self.top_k[1], self.hidden_dim)}
            per_frame += [(res, i) for i in range(B)]
            if self.debug_taps is not None:
                self.debug_taps.setdefault("extract", []).append((cl, bx, pf[0], feats))
            fired = fired or self._fire_on_global(on_global, take, len_l, n_own, a + B, n_total)

        local = take(0, len_l) if len_l else None
        glob = take(len_l, n_own) if ref_g else None
        out_ahead, pos = {}, n_own
        for fb in sorted(ahead):
            out_ahead[fb] = take(pos, pos + len(ahead[fb]))
            pos += len(ahead[fb])
        return local, glob, out_ahead

    def _final_stage(self, feats, cached, whwh, w, h, pairs, items, ddim_draws, slots=None):
        """Global attention + conditioned head (+ DDIM loop) + top-k/NMS over `R` frame slots holding the batches `items`
        = [(call frame id, real frames)], INFER_BATCH slots each unless `slots` says otherwise; -> one BoxList per slot."""
        return self._to_boxlists(*self._final_stage_launch(feats, cached, whwh, w, h, pairs, items, ddim_draws, slots), (int(w), int(h)))

    def _final_stage_launch(self, feats, cached, whwh, w, h, pairs, items, ddim_draws, slots=None):
        """the device part of _final_stage: queues every kernel, no host synchronisation -> the post-processing outputs"""
        M = self.num_proposals
        R = cached[0].shape[0]
        per = self.infer_batch if slots is None else slots
        assert R == per * len(items)
        if self.sampling_timesteps == 1:
            # x1: the randn `img` of diffusion_det.py:542 never reaches the output (the head pops the cached
            # stages, box_head.py:300-302) and the DDIM update after the single step is dead code (:573-575)
            self.head.proposals_feat_cur = [[cached[0], cached[1], cached[2].reshape(1, R * M, self.hidden_dim)]]
            t = torch.full((R,), pairs[0][0], dtype=torch.long)
            img = torch.zeros((R, M, 4), device=self.device)
            outputs_class, outputs_coord = self.model_predictions(feats, whwh, img, t)
            ob, osc, ol, oc = ops.postproc_topk_nms(outputs_class[-1], outputs_coord[-1], w, h, 0.5, self.use_nms)
            if self.debug_taps is not None:
                self.debug_taps["final_0"] = (outputs_class[-1], outputs_coord[-1])
        else:
            def padded(x, nb):          # a batch's draws cover its real frames; the duplicate tail slots get zeros
                return x if nb == per else torch.cat([x, torch.zeros((per - nb,) + tuple(x.shape[1:]), device=x.device)])
            draws = {"img": torch.cat([padded(ddim_draws[fb]["img"], nb) for fb, nb in items])}
            for step, (_, time_next) in enumerate(pairs):
                if time_next >= 0:
                    draws[step] = tuple(torch.cat([padded(ddim_draws[fb][step][i], nb) for fb, nb in items]) for i in (0, 1))
            ob, osc, ol, oc = self._ddim_ensemble(feats, whwh, R, items[0][0], pairs, w, h, draws)
        return ob, osc, ol, oc

    # ---- helpers --------------------------------------------------------------------------------
    def _time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _gather_entries(self, entries):
        """Frames of the current batch: contiguous views when they are consecutive frames of one split
        (the steady state), otherwise an index_select copy (video head/tail duplicates)."""
        src = entries[0][0]
        idx = [e[1] for e in entries]
        same = all(e[0] is src for e in entries)
        if same and idx == list(range(idx[0], idx[0] + len(idx))):
            a, b = idx[0], idx[0] + len(idx)
            return [f[a:b] for f in src["feats"]], (src["logits"][a:b], src["boxes"][a:b], src["obj"][a:b])
        if not same:
            raise NotImplementedError("current batch spans several backbone splits (INFER_BATCH != ALL_FRAME_INTERVAL)")
        sel = torch.tensor(idx, device=self.device)
        return ([f.index_select(0, sel) for f in src["feats"]],
                tuple(src[k].index_select(0, sel) for k in ("logits", "boxes", "obj")))

    def _ddim_ensemble(self, feats_cur, whwh, batch, frame_id, pairs, w, h, draws=None):
        """SAMPLE_STEP > 1 (diffusion_det.py:551-627): every step re-runs the 3 heads + global attention +
        cond head on the current noisy boxes, renews low-score boxes, and all but the last step feed the
        NMS ensemble."""
        if draws is None:
            draws = self._ddim_draws(batch, frame_id, pairs)
        img = draws["img"]
        coef = {t: (float(self._sr_host[t]), float(self._srm1_host[t])) for t, _ in pairs}
        ens_logits, ens_boxes = [], []
        for step, (time, time_next) in enumerate(pairs):
            if time_next < 0 and self.skip_unobservable:
                continue            # nothing reads the last step's outputs (diffusion_det.py:573-575, :607-627)
            t = torch.full((batch,), time, dtype=torch.long)
            outputs_class, outputs_coord = self.model_predictions(feats_cur, whwh, img, t)
            if self.debug_taps is not None:
                self.debug_taps[f"final_{step}"] = (outputs_class[-1], outputs_coord[-1])
            if time_next < 0:
                continue            # the last step never reaches the ensemble (diffusion_det.py:573-575)
            a = self._ac_host[time].double()
            an = self._ac_host[time_next].double()
            sigma = self.ddim_sampling_eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            noise, fresh = draws[step]
            img = ops.ddim_renew_step(outputs_class[-1], outputs_coord[-1], img, noise, fresh, whwh, self.scale,
                                      coef[time][0], coef[time][1], float(self._ac_host[time_next].sqrt()), float(c), float(sigma), 0.5)
            ens_logits.append(outputs_class[-1])
            ens_boxes.append(outputs_coord[-1])
        return ops.postproc_topk_nms(torch.stack(ens_logits), torch.stack(ens_boxes), w, h, 0.5, self.use_nms)

    def _to_boxlists(self, ob, osc, ol, oc, size_wh):
        flat = getattr(ob, "_dvid_flat", None)
        t0 = time.perf_counter()
        if self.results_on_host and flat is not None:
            n, cap = osc.shape
            ob, osc, ol, oc = ops.split_detection_buffer(flat.cpu(), n, cap)      # single D2H copy + sync
        counts = oc.tolist()                 # the one host sync of a batch (the caller moves results to CPU anyway)
        self.host_wait_s += time.perf_counter() - t0          # the host blocked on the GPU: its slack (bench.py reports it per rank)
        self.head.check_boxes_valid()
        if self.dtype == "float32" and self._engine is not None and self._engine.take_range_flag():
            raise ops._lib.DvidError("DTYPE float32: an activation exceeded the fp16 range (65504) of the split-operand products; the detections of this batch are "
                                     "not valid.  ops.set_option('f32_split', 0) runs the same layers on the fp32 MFMA, which has no range limit")
        results = []
        for b, k in enumerate(counts):
            bl = (getattr(self, "_result_cls", None) or BoxList)(ob[b, :k], size_wh, mode="xyxy")
            bl.add_field("scores", osc[b, :k])
            labels = ol[b, :k].to(torch.int64)
            bl.add_field("labels", labels)
            results.append(bl)
        return results
