"""DynamicHead -- host-side mirror of the reference decoder's plugin surface
(mega_core/modeling/roi_heads/box_head/box_head.py:155-435), computing on libdvid_hip.

Same constructor signature, attributes the detector mutates (`proposal_feats_global`,
`proposal_feats_local`, `proposals_feat_cur`, `top_k`, `num_heads_local`, `use_topk`) and
`forward(features, init_bboxes, t, init_features, box_extract=0)` return conventions:

  box_extract > 0 : ([class_logits [B,M,C], bboxes [B,M,4], proposal_features [1,B*M,d]],
                     top-75 features [B*75,d], top-25 features [B*25,d])      (:286-317)
  box_extract == 0: (class_logits[None], pred_bboxes[None]) of the conditioned head (:319-432)

Every RCNNHead / RCNNHead_cond pass, the global cross-attention and the top-k feature selection
run as HIP kernels (ops.Model); this class only sequences them.  Training is out of scope.
"""
import torch
from torch import nn

from .... import ops


class DynamicHead(nn.Module):
    def __init__(self, cfg, roi_input_shape=None, engine_provider=None):
        super().__init__()
        d = cfg.MODEL.DiffusionDet
        self.num_classes = d.NUM_CLASSES
        self.d_model = d.HIDDEN_DIM
        self.num_heads = d.NUM_HEADS
        self.num_heads_local = d.NUM_HEADS_LOCAL
        self.adaptive_norm = True
        self.return_intermediate = d.DEEP_SUPERVISION
        self.local_enable = cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE
        self.global_enable = cfg.MODEL.VID.MEGA.GLOBAL.ENABLE
        self.global_stage = cfg.MODEL.VID.MEGA.GLOBAL.RES_STAGE
        if self.local_enable:
            raise NotImplementedError("local box-level attention (MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE) is not on the "
                                      "DiffusionVID inference path of the shipped configs and is not built")
        if self.global_enable and self.global_stage != 1:
            raise NotImplementedError("GLOBAL.RES_STAGE != 1 is not supported")
        self.infer_batch = cfg.INPUT.INFER_BATCH
        self.top_k = [min(x, d.NUM_PROPOSALS) for x in (75, 25)]          # box_head.py:235-236
        self.sampling_timesteps = d.SAMPLE_STEP
        self.use_topk = False
        self.proposal_feats_global = [None, None]
        self.proposal_feats_local = [None, None]
        self.proposals_feat_cur = []
        self._engine_provider = engine_provider

    # ------------------------------------------------------------------------------------
    def _engine(self):
        if self._engine_provider is None:
            raise ops._lib.DvidError("DynamicHead has no compute engine attached (build it through DiffusionDet)")
        return self._engine_provider()

    @staticmethod
    def _as_nhwc(features, eng):
        """The engine's layout: NHWC in its feature dtype (fp16, or fp32 with DTYPE float32).  The detector hands its own backbone
        outputs over in that layout; a caller with the reference's NCHW maps (box_head.py:273: list[Tensor NCHW]) gets them converted."""
        feats = []
        for f in features:
            c = eng.hidden_dim
            if f.dtype == eng.feat_dtype and f.dim() == 4 and f.shape[-1] == c and (f.dtype == torch.float16 or f.shape[1] != c):
                feats.append(f)                       # already the engine's layout
            elif eng.feat_dtype == torch.float32:
                if f.dim() != 4 or f.shape[1] != c:
                    raise ops._lib.DvidError("DynamicHead: feature maps must be NCHW with %d channels (or the engine's NHWC tensors)" % c)
                feats.append(f.float().permute(0, 2, 3, 1).contiguous())
            else:
                feats.append(ops.nhwc_from_nchw(f.float()))   # reference layout: NCHW
        return feats

    def forward(self, features, init_bboxes, t, init_features, box_extract=0):
        if self.training:
            raise NotImplementedError("training is out of scope of the MI355X inference path")
        if init_features is not None:
            raise NotImplementedError("init_features is unused by DiffusionVID inference (always None, diffusion_det.py:662-664)")
        eng = self._engine()
        feats = self._as_nhwc(features, eng)
        bs, num_boxes = init_bboxes.shape[:2]
        height, width = feats[0].shape[1] * 8, feats[0].shape[2] * 8
        eng.reserve(max(bs, self.infer_batch), height, width, num_boxes)
        t_host = torch.as_tensor(t).to("cpu", torch.int64)
        flag = self._bad_flag(init_bboxes.device)
        bboxes = init_bboxes

        if box_extract > 0 or self.sampling_timesteps > 1:
            proposal_features = None
            for i in range(self.num_heads):
                class_logits, bboxes, proposal_features = eng.rcnn_head(i, feats, height, width, bboxes, proposal_features, t_host,
                                                                       bad_flag=flag)
        else:
            class_logits, bboxes, proposal_features = self.proposals_feat_cur.pop()      # box_head.py:300-302
            proposal_features = proposal_features.reshape(-1, self.d_model)

        if box_extract > 0:
            k1, k2 = ops.select_topk_features(class_logits, proposal_features, self.top_k[0], self.top_k[1])
            return [class_logits, bboxes, proposal_features.unsqueeze(0)], k1, k2

        if not self.global_enable:
            return class_logits[None], bboxes[None]

        memory = self.proposal_feats_global[0]
        if memory is None:
            raise RuntimeError("proposal_feats_global is empty: the detector fills it from the global frames of a video")
        attn_ = eng.global_xattn(proposal_features, memory)                               # box_head.py:366-394
        class_logits2, bboxes2 = class_logits, bboxes
        for i in range(self.num_heads_local):
            class_logits2, bboxes2, proposal_features = eng.rcnn_head(i, feats, height, width, bboxes2, proposal_features, t_host,
                                                                      cond=attn_, bad_flag=flag)
        return class_logits2[None], bboxes2[None]

    # device flag standing in for `assert (pred_boxes[:, 2:] >= pred_boxes[:, :2]).all()` (box_head.py:588)
    def _bad_flag(self, device):
        f = getattr(self, "_flag", None)
        if f is None or f.device != device:
            self._flag = torch.zeros(1, dtype=torch.int32, device=device)
        return self._flag

    def check_boxes_valid(self):
        """Raises the reference's AssertionError if any head produced x2<x1 / y2<y1 (host sync)."""
        f = getattr(self, "_flag", None)
        if f is not None and int(f.item()) != 0:
            f.zero_()
            raise AssertionError("pred_boxes[:, 2:] >= pred_boxes[:, :2] violated (box_head.py:588)")
