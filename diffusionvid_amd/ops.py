"""Tensor-level wrappers over the libdvid_hip C ABI (torch supplies device memory and streams only).

Every function launches HIP kernels on the current torch stream; none falls back to torch math.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def _cuda(t, dtype=None):
    if not t.is_cuda:
        raise _lib.DvidError("libdvid_hip ops need tensors on the GPU (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.contiguous()


def nhwc_from_nchw(x):
    """fp32 NCHW -> fp16 NHWC."""
    x = _cuda(x, torch.float32)
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=x.device)
    call("dvid_nhwc_from_nchw", ptr(x), ptr(out), n, h, w, c, stream_ptr())
    return out


def nchw_from_nhwc(x):
    """fp16 NHWC -> fp32 NCHW (an fp32 NHWC map, DTYPE float32, is a pure permutation: a strided copy)."""
    if x.dtype == torch.float32:
        return _cuda(x).permute(0, 3, 1, 2).contiguous()
    x = _cuda(x, torch.float16)
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    call("dvid_nchw_from_nhwc", ptr(x), ptr(out), n, h, w, c, stream_ptr())
    return out


def pack_conv_weight(w, cin_pad=0):
    """OIHW fp32 (or [out,in]) -> fp16 [cout, kpad] with k = (ky*kw+kx)*cin + c (host side repack)."""
    w = w.detach().float().cpu()
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    cp = cin_pad or cin
    kreal = kh * kw * cp
    kpad = (kreal + 63) // 64 * 64
    packed = torch.zeros((cout, kh * kw, cp), dtype=torch.float32)
    packed[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    out = torch.zeros((cout, kpad), dtype=torch.float16)
    out[:, :kreal] = packed.reshape(cout, kreal).to(torch.float16)
    return out, kpad


def conv2d_nhwc(x, w_packed, kpad, bias, cout, kh, kw, stride, pad, relu=False, residual=None, residual_mode=0,
                out_f32=False):
    """x fp16 NHWC; returns NHWC fp16 (or fp32)."""
    x = _cuda(x, torch.float16)
    n, h, wd, cin = x.shape
    if pad < 0:          # same-size window: |pad| before, the rest after (stride 1)
        ho, wo = h, wd
    else:
        ho = (h + 2 * pad - kh) // stride + 1
        wo = (wd + 2 * pad - kw) // stride + 1
    out = torch.empty((n, ho, wo, cout), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    call("dvid_conv2d_nhwc_f16", ptr(x), ptr(w_packed), ptr(bias), ptr(residual), ptr(out), n, h, wd, cin, cout, kh, kw,
         stride, pad, kpad, int(relu), int(out_f32), residual_mode, stream_ptr())
    return out


# ---- DTYPE float32 forms (csrc/f32.hip): fp32 NHWC activations, un-rounded weights, fp32 MFMA --------------------------------------
def pack_conv_weight_f32(w, scale_rows=False):
    """OIHW fp32 (or [out, in]) -> fp32 [cout, kpad], k = (ky*kw+kx)*cin4 + c with cin4 = cin rounded up to 4, kpad to 16 (zeros).
    scale_rows: additionally multiply each row by the power of two that puts its largest magnitude in [0.5, 1) and return
    (packed, kpad, row_scale) with row_scale = 2^-e for conv2d_nhwc_f32 (what csrc/model.hip: make_conv does for DTYPE float32)."""
    w = w.detach().float().cpu()
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    c4 = (cin + 3) // 4 * 4
    kpad = (kh * kw * c4 + 15) // 16 * 16
    packed = torch.zeros((cout, kh * kw, c4), dtype=torch.float32)
    packed[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    out = torch.zeros((cout, kpad), dtype=torch.float32)
    out[:, :kh * kw * c4] = packed.reshape(cout, -1)
    if scale_rows:
        mx = out.abs().amax(dim=1)
        e = torch.where(mx > 0, torch.frexp(mx)[1], torch.zeros_like(mx, dtype=torch.int32)).to(torch.float32)
        return out * torch.exp2(-e)[:, None], kpad, torch.exp2(e)
    return out, kpad


def split_f16(w_packed):
    """fp32 -> (hi, lo) fp16 planes, hi = fp16(w), lo = fp16(w - hi): the weight operand of the split-operand kernel (conv2d_nhwc_f32)"""
    hi = w_packed.to(torch.float16)
    return hi, (w_packed - hi.to(torch.float32)).to(torch.float16)


def conv2d_nhwc_f32(x, w_packed, kpad, bias, cout, kh, kw, stride, pad, relu=0, residual=None, residual_mode=0, row_scale=None, w_split=None):
    """x fp32 NHWC (channels a multiple of 4); returns fp32 NHWC.  relu: 0 none, 1 ReLU, 2 exact GELU; row_scale: see pack_conv_weight_f32;
    w_split = split_f16(w_packed): run the products on split fp16 operands (library option f32_split, default on); None: the fp32 MFMA."""
    x = _cuda(x, torch.float32)
    n, h, wd, cin = x.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    out = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    wh, wl = w_split if w_split is not None else (None, None)
    call("dvid_conv2d_nhwc_f32", ptr(x), ptr(w_packed), ptr(wh), ptr(wl), ptr(bias), ptr(row_scale), ptr(residual), ptr(out), n, h, wd, cin, cout, kh, kw, stride, pad, kpad,
         int(relu), residual_mode, stream_ptr())
    return out


def linear_f32(x, w_packed, kpad, bias, relu=0, row_scale=None, w_split=None):
    rows, k = x.shape
    return conv2d_nhwc_f32(x.view(rows, 1, 1, k), w_packed, kpad, bias, w_packed.shape[0], 1, 1, 1, 0, relu=relu, row_scale=row_scale, w_split=w_split).view(rows, -1)


def roialign_f32(feats_nhwc, boxes, height, width, want_mean=False):
    """feats_nhwc: [p3, p4, p5] fp32 NHWC; boxes fp32 [n, M, 4] -> roi fp32 [n*M, 49, C] (+ mean fp32 [n*M, C])."""
    p3, p4, p5 = (_cuda(f, torch.float32) for f in feats_nhwc)
    boxes = _cuda(boxes, torch.float32)
    n, M = boxes.shape[:2]
    c = p3.shape[-1]
    roi = torch.empty((n * M, 49, c), dtype=torch.float32, device=boxes.device)
    mean = torch.empty((n * M, c), dtype=torch.float32, device=boxes.device) if want_mean else None
    call("dvid_roialign_v2_multilevel_f32", ptr(p3), ptr(p4), ptr(p5), n, height, width, c, ptr(boxes), M, ptr(roi), ptr(mean), stream_ptr())
    return (roi, mean) if want_mean else roi


def mha_f32(q, k, v, nheads):
    """fp32 attention: q [B, Lq, d], k / v [B, Lk, d] -> fp32 [B, Lq, d] (head dim 32)."""
    q, k, v = _cuda(q, torch.float32), _cuda(k, torch.float32), _cuda(v, torch.float32)
    B, lq, d = q.shape
    lk = k.shape[1]
    out = torch.empty_like(q)
    call("dvid_mha_f32", ptr(q), ptr(k), ptr(v), ptr(out), B, lq, lk, nheads, d, d, d, lq * d, lk * d, lq * d, stream_ptr())
    return out


def dynconv_f32(roi, params, g1, b1, g2, b2):
    """roi fp32 [R, 49, 256], params fp32 [R, 32768] as P1T[64][256] | P2T[256][64] -> fp32 [R, 49, 256]"""
    roi, params = _cuda(roi, torch.float32), _cuda(params, torch.float32)
    out = torch.empty_like(roi)
    call("dvid_dynconv_f32", ptr(roi), ptr(params), ptr(g1), ptr(b1), ptr(g2), ptr(b2), ptr(out), roi.shape[0], stream_ptr())
    return out


def bottleneck64_tail(t1, w2, b2, w3, b3, residual, w_sc=None, b_sc=None, w1n=None, b1n=None):
    """Everything behind conv1 of a res2 bottleneck block as one launch (csrc/bneck.hip): t1 fp16 [n,h,w,64]; packed weights w2
    [64,576], w3 [256,64], optional shortcut [256,64] (then `residual` is the 64-channel block input) and next conv1 [64 or 128, 256].
    Returns (out [n,h,w,256], t1_next [n,h,w,64 or 128] or None)."""
    t1 = _cuda(t1, torch.float16)
    n, h, wd, _ = t1.shape
    nn = int(w1n.shape[0]) if w1n is not None else 0
    out = torch.empty((n, h, wd, 256), dtype=torch.float16, device=t1.device)
    t1n = torch.empty((n, h, wd, nn), dtype=torch.float16, device=t1.device) if w1n is not None else None
    call("dvid_bottleneck64_tail_f16", ptr(t1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(residual), ptr(w_sc), ptr(b_sc), ptr(w1n),
         ptr(b1n), nn, ptr(out), ptr(t1n), n, h, wd, stream_ptr())
    return out, t1n


def bottleneck128_tail(t1, w2, b2, w3, b3, residual, w1n=None, b1n=None):
    """The 128-wide form (res3; csrc/bneck.hip): t1 fp16 [n,h,w,128] (w2 None: already the conv2 output); packed w2 [128,1152],
    w3 [512,128], optional next conv1 [128,512]; residual [n,h,w,512].  Returns (out [n,h,w,512], t1_next [n,h,w,128] or None)."""
    t1 = _cuda(t1, torch.float16)
    n, h, wd, _ = t1.shape
    out = torch.empty((n, h, wd, 512), dtype=torch.float16, device=t1.device)
    t1n = torch.empty((n, h, wd, 128), dtype=torch.float16, device=t1.device) if w1n is not None else None
    call("dvid_bottleneck128_tail_f16", ptr(t1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(residual), ptr(w1n), ptr(b1n), ptr(out),
         ptr(t1n), n, h, wd, stream_ptr())
    return out, t1n


def linear(x16, w_packed, kpad, bias, relu=False, out_f32=True):
    rows, k = x16.shape
    y = conv2d_nhwc(x16.view(rows, 1, 1, k), w_packed, kpad, bias, w_packed.shape[0], 1, 1, 1, 0, relu=relu, out_f32=out_f32)
    return y.view(rows, -1)


def roialign(feats_nhwc, boxes, height, width, want_mean=False):
    """feats_nhwc: [p3,p4,p5] fp16 NHWC; boxes fp32 [n, M, 4] -> roi fp16 [n*M, 49, C] (+ mean fp32 [n*M, C])."""
    p3, p4, p5 = (_cuda(f, torch.float16) for f in feats_nhwc)
    boxes = _cuda(boxes, torch.float32)
    n, M = boxes.shape[:2]
    c = p3.shape[-1]
    roi = torch.empty((n * M, 49, c), dtype=torch.float16, device=boxes.device)
    mean = torch.empty((n * M, c), dtype=torch.float32, device=boxes.device) if want_mean else None
    call("dvid_roialign_v2_multilevel", ptr(p3), ptr(p4), ptr(p5), n, height, width, c, ptr(boxes), M, ptr(roi), ptr(mean),
         stream_ptr())
    return (roi, mean) if want_mean else roi


def mha_f16(q, k, v, nheads):
    """MFMA attention: q [B, Lq, d], k/v [B, Lk, d] (fp32 or fp16, rounded to fp16) -> fp16 [B, Lq, d]."""
    def h(t):
        t = _cuda(t)
        if t.dtype == torch.float16:
            return t
        o = torch.empty(t.shape, dtype=torch.float16, device=t.device)
        call("dvid_f32_to_f16", ptr(t), ptr(o), t.numel(), stream_ptr())
        return o
    q, k, v = h(q), h(k), h(v)
    B, lq, d = q.shape
    lk = k.shape[1]
    out = torch.empty_like(q)
    vt = torch.empty((B * nheads * 32 * ((lk + 31) // 32 * 32 + 32),), dtype=torch.float16, device=q.device)
    call("dvid_mha_f16", ptr(q), ptr(k), ptr(v), ptr(out), ptr(vt), B, lq, lk, nheads, d, d, d, lq * d, lk * d, lq * d,
         stream_ptr())
    return out


def dynconv(roi16, params16, g1, b1, g2, b2):
    roi16, params16 = _cuda(roi16, torch.float16), _cuda(params16, torch.float16)
    out = torch.empty_like(roi16)
    call("dvid_dynconv", ptr(roi16), ptr(params16), ptr(g1), ptr(b1), ptr(g2), ptr(b2), ptr(out), roi16.shape[0], stream_ptr())
    return out


def add_layernorm(x, r, g, b, relu=False):
    x = _cuda(x, torch.float32)
    y = torch.empty_like(x)
    call("dvid_add_layernorm", ptr(x), ptr(r), ptr(g), ptr(b), ptr(y), x.shape[0], x.shape[1], int(relu), stream_ptr())
    return y


def resize_u8_to_f32(src_hwc, oh, ow, ph, pw, xtab, ytab):
    """uint8 [H, W, 3] (device) -> fp32 [1, 3, ph, pw] in [0, 1]: Pillow-identical bilinear resize to (oh, ow), zero padding to
    (ph, pw).  xtab / ytab: (bounds, weights) int32 device tensors of data/transforms.resample_tables, None for an axis
    that keeps its size."""
    src = _cuda(src_hwc, torch.uint8)
    h, w = src.shape[:2]
    out = torch.empty((1, 3, ph, pw), dtype=torch.float32, device=src.device)
    tmp = torch.empty((h, ow, 3), dtype=torch.uint8, device=src.device) if xtab is not None else None
    xb, xk = xtab if xtab is not None else (None, None)
    yb, yk = ytab if ytab is not None else (None, None)
    call("dvid_resize_u8_to_f32", ptr(src), h, w, ptr(tmp), ptr(out), oh, ow, ph, pw, ptr(xb), ptr(xk), 0 if xk is None else xk.shape[1],
         ptr(yb), ptr(yk), 0 if yk is None else yk.shape[1], stream_ptr())
    return out


def noise_to_boxes(x, snr_scale, img_w, img_h):
    x = _cuda(x, torch.float32)
    out = torch.empty_like(x)
    call("dvid_noise_to_boxes", ptr(x), ptr(out), x.numel() // 4, float(snr_scale), float(img_w), float(img_h), stream_ptr())
    return out


def counter_normal(key0, n_images, shape, device=None):
    """[n_images, *shape] N(0, 1) draws generated on the device: image i is the stream keyed `key0 + i` (dvid_counter_normal)."""
    if not torch.cuda.is_available():
        raise _lib.DvidError("no HIP device visible: counter_normal draws on the GPU (no CPU fallback; the CPU restatement is oracle/noise.py)")
    out = torch.empty((n_images,) + tuple(shape), dtype=torch.float32, device=device or torch.device("cuda", torch.cuda.current_device()))
    per = 1
    for v in shape:
        per *= int(v)
    call("dvid_counter_normal", ptr(out), per, int(n_images), int(key0) & 0xFFFFFFFFFFFFFFFF, stream_ptr())
    return out


def ddim_renew_step(logits, boxes, x_t, noise, fresh, whwh, snr_scale, sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_next, coef_c,
                    sigma, keep_thr=0.5):
    """Box renewal + DDIM update (diffusion_det.py:559-596); all [n, M, 4] fp32, logits [n, M, C]."""
    logits, boxes, x_t = _cuda(logits, torch.float32), _cuda(boxes, torch.float32), _cuda(x_t, torch.float32)
    noise, fresh = _cuda(noise, torch.float32), _cuda(fresh, torch.float32)
    n, M, c = logits.shape
    out = torch.empty_like(x_t)
    call("dvid_ddim_renew_step", ptr(logits), ptr(boxes), ptr(x_t), ptr(noise), ptr(fresh), ptr(out), n, M, c, float(whwh[0]),
         float(whwh[1]), float(snr_scale), float(sqrt_recip_ac), float(sqrt_recipm1_ac), float(sqrt_ac_next), float(coef_c),
         float(sigma), float(keep_thr), stream_ptr())
    return out


def select_topk_features(logits, feats, k1, k2):
    """logits [n, M, C], feats [n*M, d] -> ([n*k1, d], [n*k2, d]) in box-index (mask) order."""
    logits, feats = _cuda(logits, torch.float32), _cuda(feats, torch.float32)
    n, M, c = logits.shape
    d = feats.shape[-1]
    o1 = torch.empty((n * k1, d), dtype=torch.float32, device=feats.device)
    o2 = torch.empty((n * k2, d), dtype=torch.float32, device=feats.device)
    call("dvid_select_topk_features", ptr(logits), n, M, c, k1, k2, ptr(feats), d, ptr(o1), ptr(o2), stream_ptr())
    return o1, o2


def postproc_topk_nms(logits, boxes, img_w, img_h, iou=0.5, use_nms=True):
    """logits [S, n, M, C] (or [n, M, C]), boxes [S, n, M, 4] -> (boxes [n,S*M,4], scores, labels int32, counts int32)."""
    if logits.dim() == 3:
        logits, boxes = logits[None], boxes[None]
    logits, boxes = _cuda(logits, torch.float32), _cuda(boxes, torch.float32)
    S, n, M, c = logits.shape
    dev = logits.device
    ob, osc, ol, oc = split_detection_buffer(torch.empty((n * S * M * 6 + n,), dtype=torch.float32, device=dev), n, S * M)
    scratch = torch.empty((n * S * M * 6,), dtype=torch.float32, device=dev)
    call("dvid_postproc_topk_nms", ptr(logits), ptr(boxes), S, n, M, c, float(img_w), float(img_h), float(iou), int(use_nms),
         ptr(ob), ptr(osc), ptr(ol), ptr(oc), ptr(scratch), stream_ptr())
    return ob, osc, ol, oc


def split_detection_buffer(buf, n, cap):
    """The four post-processing outputs are views of ONE flat fp32 buffer [boxes | scores | labels(int32) |
    counts(int32)], so a caller can bring a whole batch to the host with a single D2H copy
    (`split_detection_buffer(ob.untyped... ` see DiffusionDet._to_boxlists)."""
    a, b = n * cap * 4, n * cap * 5
    ob = buf[:a].view(n, cap, 4)
    osc = buf[a:b].view(n, cap)
    ol = buf[b:b + n * cap].view(torch.int32).view(n, cap)
    oc = buf[b + n * cap:b + n * cap + n].view(torch.int32)
    ob._dvid_flat = buf            # keep the flat buffer reachable from the first view
    return ob, osc, ol, oc


def cdist(x):
    x = _cuda(x, torch.float32)
    n, d = x.shape
    out = torch.empty((n, n), dtype=torch.float32, device=x.device)
    call("dvid_cdist", ptr(x), n, d, ptr(out), stream_ptr())
    return out


def fps_greedy(dist, m, bs_emul=0):
    dist = _cuda(dist, torch.float32)
    n = dist.shape[0]
    idx = torch.empty((m,), dtype=torch.int32, device=dist.device)
    call("dvid_fps_greedy", ptr(dist), n, m, bs_emul, ptr(idx), stream_ptr())
    return idx


def gather_rows(x, idx):
    x = _cuda(x, torch.float32)
    idx = _cuda(idx, torch.int32)
    out = torch.empty((idx.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    call("dvid_gather_rows", ptr(x), ptr(idx), ptr(out), idx.shape[0], x.shape[1], stream_ptr())
    return out


def update_erase_memory(feats_new, feats_mem, target_size):
    """diffusion_det.py:841-867 on device: cat -> (cdist -> greedy FPS -> gather) if over target."""
    merged = feats_new if feats_mem is None else torch.cat([feats_mem, feats_new], dim=0)
    merged = merged.contiguous()
    if merged.shape[0] <= target_size:
        return merged, None
    idx = fps_greedy(cdist(merged), target_size)
    return gather_rows(merged, idx), idx


def set_option(name, value):
    """one entry of the library's option table (include/dvid_hip.h: dvid_set_option; csrc/options.h lists names and defaults)"""
    call("dvid_set_option", name.encode(), int(value))


def get_option(name):
    v = C.c_int()
    call("dvid_get_option", name.encode(), C.byref(v))
    return v.value


def reset_options():
    call("dvid_reset_options")


def effective_config():
    """"name=value ..." of every library option + the environment variables the library still reads (dvid_effective_config)"""
    buf = C.create_string_buffer(1024)
    call("dvid_effective_config", buf, 1024)
    return buf.value.decode()




PRECISIONS = {"float16": 0, "float32": 1}          # the reference's DTYPE values (mega_core/config/defaults.py:582) -> dvid_model_set_precision


class Model:
    """Owns a dvid_model handle: repacked weights + activation workspace on the current device."""

    def __init__(self, state_dict, *, hidden_dim=256, nheads=8, dim_feedforward=2048, dim_dynamic=64, num_classes=30,
                 num_cls=1, num_reg=3, num_heads=3, num_heads_cond=1, pooler_resolution=7, sampling_ratio=2,
                 res_blocks=(3, 4, 23, 3), pixel_mean=(123.675, 116.280, 103.530), pixel_std=(58.395, 57.120, 57.375),
                 backbone="resnet", swin_embed_dim=128, swin_depths=(2, 2, 18, 2), swin_heads=(4, 8, 16, 32), swin_window=7,
                 precision="float16"):
        """precision: the reference's DTYPE key -- "float16" (fp16 storage, fp16 MFMA, fp32 accumulation) or "float32" (fp32 storage,
        fp32 MFMA; csrc/f32.hip).  Feature maps are fp16 / fp32 NHWC tensors accordingly."""
        if precision not in PRECISIONS:
            raise _lib.DvidError(f"precision must be one of {sorted(PRECISIONS)} (the reference's DTYPE), got {precision!r}")
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.DvidError("no HIP device visible: the DiffusionVID hot path runs only on the GPU (no CPU fallback)")
        cfg = _lib.DvidConfig(hidden_dim, nheads, dim_feedforward, dim_dynamic, num_classes, num_cls, num_reg, num_heads,
                              num_heads_cond, pooler_resolution, sampling_ratio, (C.c_int * 4)(*res_blocks),
                              (C.c_float * 3)(*pixel_mean), (C.c_float * 3)(*pixel_std),
                              1 if backbone == "swin" else 0, swin_embed_dim, (C.c_int * 4)(*swin_depths),
                              (C.c_int * 4)(*swin_heads), swin_window)
        self.backbone_kind = backbone
        self.cfg = cfg
        self.precision = precision
        self.feat_dtype = torch.float32 if precision == "float32" else torch.float16
        h = C.c_void_p()
        _lib.check(lib.dvid_model_create(C.byref(cfg), C.byref(h)), "dvid_model_create")
        self.handle = h
        _lib.check(lib.dvid_model_set_precision(h, PRECISIONS[precision]), "dvid_model_set_precision")
        self.hidden_dim, self.num_classes, self.nheads = hidden_dim, num_classes, nheads
        self.has_backbone = (swin_depths[0] > 0) if backbone == "swin" else (res_blocks[0] > 0)
        wanted = ("head.", "backbone.") if self.has_backbone else ("head.",)
        for name, t in state_dict.items():
            if not name.startswith(wanted) or not torch.is_floating_point(t):
                continue
            t = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(lib.dvid_model_set_tensor(h, name.encode(), t.data_ptr(), shape, t.dim()), f"set_tensor({name})")
        _lib.check(lib.dvid_model_finalize(h), "dvid_model_finalize")
        self._ws = None
        self.chains = -1                 # the library's default until set_chains
        self._kv_src = None

    def take_range_flag(self):
        """DTYPE float32 with split operands: True if a launch since the last call met an activation beyond the fp16 range (65504); synchronises
        the current stream and clears the flag (include/dvid_hip.h: dvid_model_take_range_flag).  Always False for float16."""
        if self.precision != "float32":
            return False
        v = C.c_int(0)
        call("dvid_model_take_range_flag", self.handle, C.byref(v), stream_ptr())
        return bool(v.value)

    def workspace_generation(self):
        """moves of this model's workspace buffers so far (include/dvid_hip.h: dvid_workspace_generation); a captured launch sequence is
        valid only while this stays what it was at capture time"""
        return int(_lib.load().dvid_workspace_generation(self.handle))

    def set_chains(self, n):
        """concurrent sub-batch chains inside the library (1 = sequential kernels, for per-kernel profiling)"""
        call("dvid_set_chains", self.handle, int(n))
        self.chains = int(n)

    def set_stem_layout(self, space_to_depth=True):
        """ResNet stem over the 2x2 space-to-depth image (default) or the NHWC8 image; see dvid_set_stem_layout"""
        call("dvid_set_stem_layout", self.handle, int(bool(space_to_depth)))

    def reserve(self, max_frames, height, width, boxes_per_frame):
        """Workspace for up to max_frames frames of height x width with boxes_per_frame boxes; only ever grows."""
        key = (max_frames, height, width, boxes_per_frame)
        if self._ws is not None:
            key = tuple(max(a, b) for a, b in zip(key, self._ws))
        if self._ws != key:
            call("dvid_workspace_reserve", self.handle, *key)
            self._ws = key

    def backbone(self, images, frames_per_launch=None):
        """images fp32 NCHW [n,3,H,W] in [0,1] -> (p3, p4, p5) fp16 (precision float32: fp32) NHWC.  frames_per_launch: run the n frames as
        consecutive launch sequences of at most that many frames (same results; smaller activation working set)."""
        images = _cuda(images, torch.float32)
        n, _, h, w = images.shape
        dev = images.device
        outs = [torch.empty((n, h >> s, w >> s, self.hidden_dim), dtype=self.feat_dtype, device=dev) for s in (3, 4, 5)]
        fn = "dvid_backbone_swin_fpn" if self.backbone_kind == "swin" else "dvid_backbone_resnet_fpn"
        step = n if not frames_per_launch else max(1, min(n, int(frames_per_launch)))
        for a in range(0, n, step):
            b = min(n, a + step)
            call(fn, self.handle, ptr(images[a:b]), b - a, h, w, ptr(outs[0][a:b]), ptr(outs[1][a:b]), ptr(outs[2][a:b]), stream_ptr())
        return outs

    def backbone_frames(self, frames):
        """The same over a LIST of per-frame tensors, each fp32 [1, 3, H, W] (or [3, H, W]) on the device, all of one size: the
        kernels read every frame where it lies (a table of pointers), no concatenated copy is made
        (diffusion_det.py:418-421 concatenates)."""
        first = frames[0]
        h, w = first.shape[-2:]
        keep = []
        for f in frames:
            if not f.is_cuda or f.dtype != torch.float32 or f.shape[-2:] != (h, w) or f.shape[-3] != 3 or f.numel() != 3 * h * w:
                raise _lib.DvidError("backbone_frames: frames must be fp32 [1, 3, H, W] device tensors of one size")
            keep.append(f if f.is_contiguous() else f.contiguous())
        n = len(keep)
        dev = first.device
        outs = [torch.empty((n, h >> s, w >> s, self.hidden_dim), dtype=self.feat_dtype, device=dev) for s in (3, 4, 5)]
        table = (C.c_void_p * n)(*[f.data_ptr() for f in keep])
        fn = "dvid_backbone_swin_fpn_frames" if self.backbone_kind == "swin" else "dvid_backbone_resnet_fpn_frames"
        call(fn, self.handle, table, n, h, w, ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), stream_ptr())
        return outs          # `keep` may die here: the stream-ordered caching allocator re-uses a frame's block only behind this stream's reads

    def rcnn_head(self, head_index, feats_nhwc, height, width, boxes, pro_features, t, cond=None, bad_flag=None):
        """One RCNNHead / RCNNHead_cond pass.  boxes [n, M, 4]; pro_features [n*M, d] or None; t: int64 [n] (CPU)."""
        boxes = _cuda(boxes, torch.float32)
        n, M = boxes.shape[:2]
        dev = boxes.device
        d = self.hidden_dim
        for f in feats_nhwc:          # the library reads them as the model's precision says: a mismatch would be silent garbage
            if f.dtype != self.feat_dtype or not f.is_contiguous():
                raise _lib.DvidError(f"rcnn_head: feature maps must be contiguous {self.feat_dtype} NHWC tensors for precision {self.precision}")
        logits = torch.empty((n, M, self.num_classes), dtype=torch.float32, device=dev)
        boxes_out = torch.empty((n, M, 4), dtype=torch.float32, device=dev)
        obj = torch.empty((n * M, d), dtype=torch.float32, device=dev)
        t = torch.as_tensor(t, dtype=torch.int64, device="cpu").contiguous()
        assert t.numel() == n
        tp = C.cast(t.data_ptr(), C.POINTER(C.c_int64))
        if pro_features is not None:
            pro_features = _cuda(pro_features, torch.float32)
        if cond is not None:
            cond = _cuda(cond, torch.float32)
        call("dvid_rcnn_head", self.handle, head_index, int(cond is not None), ptr(feats_nhwc[0]), ptr(feats_nhwc[1]),
             ptr(feats_nhwc[2]), n, height, width, M, ptr(boxes), ptr(pro_features), tp, ptr(cond), ptr(logits), ptr(boxes_out),
             ptr(obj), ptr(bad_flag), stream_ptr())
        return logits, boxes_out, obj

    def invalidate_memory(self):
        """The global memory changed (new video, memory update, adopted memory, or an in-place write through a raw pointer,
        which no tensor version counter sees): the next global_xattn projects K/V again."""
        self._kv_src = None

    def ensure_memory_projected(self, memory):
        """K / V projections of `memory` into the engine's buffers unless they are current (same tensor object, same version).
        global_xattn calls it; a captured launch sequence that READS those buffers (DiffusionDet._graphed_call) calls it before
        every replay, because the projection itself is not part of the capture."""
        if self._kv_src is None or self._kv_src is not memory or self._kv_ver != memory._version:
            mem = _cuda(memory, torch.float32)
            call("dvid_global_memory_project", self.handle, ptr(mem), mem.shape[0], stream_ptr())
            self._kv_src = memory            # holds the tensor: its storage cannot be recycled under the cache
            self._kv_ver = memory._version

    def global_xattn(self, query, memory):
        """cond = MHA(query, memory, memory).  The K/V projections of `memory` are kept until `invalidate_memory()` (the
        detector calls it wherever it assigns the memory) or until another tensor object is passed, i.e. they are computed
        once per memory update of a video.  The cache key is (tensor object, its version counter): an in-place torch write to
        the memory (copy_, index_put_, ...) bumps the counter and projects again; a write through a raw pointer, which no
        counter sees, needs `invalidate_memory()`."""
        query = _cuda(query, torch.float32)
        self.ensure_memory_projected(memory)
        out = torch.empty_like(query)
        call("dvid_global_xattn", self.handle, ptr(query), query.shape[0], None, memory.shape[0], ptr(out), stream_ptr())
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().dvid_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
