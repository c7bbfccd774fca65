/* libdvid_hip -- MI355X (gfx950) native DiffusionVID inference hot path, C ABI.
 *
 * Drop-in boundary for the reference's per-batch compute (sdroh1027/DiffusionVID).  Every entry
 * point takes plain device/host pointers and sizes, enqueues work on the HIP stream given
 * (`stream` = hipStream_t, NULL = default stream), performs no hidden synchronisation and returns
 * 0 on success (see DVID_* codes; dvid_last_error() has the text).  The Python host side
 * (diffusionvid_amd/) binds these with ctypes on torch `data_ptr()`s; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   dvid_backbone_resnet_fpn   detectron2 build_resnet_fpn_backbone called at
 *                              mega_core/modeling/detector/diffusion_det.py:219,:427 (+ normalizer :301-303,:422)
 *   dvid_backbone_swin_fpn     build_swintransformer_fpn_backbone, mega_core/modeling/backbone/swintransformer.py:464-751
 *   dvid_rcnn_head             RCNNHead.forward / RCNNHead_cond.forward, DynamicConv.forward, apply_deltas
 *                              mega_core/modeling/roi_heads/box_head/box_head.py:495-548, :605-664, :687-711, :550-590
 *   dvid_roialign_v2_multilevel detectron2 ROIPooler(ROIAlignV2) built at box_head.py:250-271, called :507,:617
 *   dvid_global_xattn          nn.MultiheadAttention global stage, box_head.py:366-380
 *   dvid_select_topk_features  box_head.py:304-317
 *   dvid_noise_to_boxes        diffusion_det.py:657-660
 *   dvid_counter_normal        torch.randn(shape, device=self.device) at diffusion_det.py:449, :542, :587, :595
 *   dvid_ddim_renew_step       box renewal + DDIM update, diffusion_det.py:559-596
 *   dvid_postproc_topk_nms     DiffusionDet.inference + detectron2 batched_nms + BoxList.clip_to_image
 *                              diffusion_det.py:754-839, :607-627; structures/bounding_box.py:214-224
 *   dvid_cdist / dvid_fps_greedy / dvid_gather_rows
 *                              update_erase_memory / select_farthest_k_greedy_cuda, diffusion_det.py:841-896;
 *                              mega_core/csrc/fps.h:15-36 + csrc/cuda/fps.cu:25-185 (`_C.furthest_point_sampling`)
 *   dvid_model_set_tensor      DetectronCheckpointer.load name contract, mega_core/utils/model_serialization.py:12-73
 *
 * Layouts: images fp32 NCHW in [0,1] (what the reference model receives); feature maps fp16 NHWC
 * (channels fastest; fp32 NHWC with DTYPE float32, dvid_model_set_precision); boxes fp32 xyxy absolute pixels; object features
 * fp32 [rows, hidden].
 */
#ifndef DVID_HIP_H
#define DVID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVID_OK 0
#define DVID_ERR_ARG 1
#define DVID_ERR_HIP 2
#define DVID_ERR_UNSUPPORTED 3
#define DVID_ERR_STATE 4

typedef struct dvid_model dvid_model;

typedef struct dvid_config {
    int hidden_dim;        /* MODEL.DiffusionDet.HIDDEN_DIM      (256) */
    int nheads;            /* MODEL.DiffusionDet.NHEADS          (8)   */
    int dim_feedforward;   /* MODEL.DiffusionDet.DIM_FEEDFORWARD (2048) */
    int dim_dynamic;       /* MODEL.DiffusionDet.DIM_DYNAMIC     (64)  */
    int num_classes;       /* MODEL.DiffusionDet.NUM_CLASSES     (30)  */
    int num_cls;           /* MODEL.DiffusionDet.NUM_CLS         (1)   */
    int num_reg;           /* MODEL.DiffusionDet.NUM_REG         (3)   */
    int num_heads;         /* MODEL.DiffusionDet.NUM_HEADS       (3)   head_series */
    int num_heads_cond;    /* MODEL.DiffusionDet.NUM_HEADS_LOCAL (1)   head_series_cond */
    int pooler_resolution; /* MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION (7) */
    int sampling_ratio;    /* MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO (2) */
    int res_blocks[4];     /* bottleneck blocks per stage, {3,4,23,3} for R-101; all 0 = no backbone */
    float pixel_mean[3];   /* MODEL.PIXEL_MEAN (0..255 scale) */
    float pixel_std[3];
    int backbone_type;     /* 0: ResNet-FPN (res_blocks), 1: Swin-FPN (swin_*), MODEL.BACKBONE.NAME */
    int swin_embed_dim;    /* size2config[MODEL.SWIN.SIZE]: 128 for B */
    int swin_depths[4];    /* {2,2,18,2} */
    int swin_heads[4];     /* {4,8,16,32}; head dim must be 32 */
    int swin_window;       /* 7 */
} dvid_config;

const char* dvid_last_error(void);
int dvid_version(void);

/* ---- model: weights (host fp32 in, repacked fp16 on device) + workspace ---------------------- */
int dvid_model_create(const dvid_config* cfg, dvid_model** out);
int dvid_model_destroy(dvid_model* m);
/* state_dict entry, reference parameter names ("backbone.bottom_up.res2.0.conv1.weight",
 * "head.head_series.0.inst_interact.dynamic_layer.weight", ...); data is copied. */
int dvid_model_set_tensor(dvid_model* m, const char* name, const float* data, const int64_t* shape, int ndim);
/* The reference's global DTYPE switch (mega_core/config/defaults.py:582, default "float32"; tools/test_net.py:97-98 turns apex amp on for
 * "float16" only): precision 0 = DTYPE float16 -- fp16 weights / stored activations, fp16 MFMA, fp32 accumulation (the apex O1 policy);
 * precision 1 = DTYPE float32 -- every weight and activation fp32, products on the fp32 MFMA (csrc/f32.hip; both backbones).  Feature maps handed to / taken from the stage functions below are fp16 NHWC in mode 0 and fp32 NHWC in
 * mode 1.  Must be called before dvid_model_finalize (the weights are packed for one precision); the default is 0. */
int dvid_model_set_precision(dvid_model* m, int precision);
/* precision 1 with library option f32_split = 1 (default) multiplies (hi, lo) fp16 pairs: an ACTIVATION whose magnitude exceeds the fp16 range
 * (65504) cannot be split, and its products would be inf / NaN where fp32 arithmetic is finite.  Such a launch sets a device flag instead of
 * failing silently; this call copies it to *exceeded (synchronising `stream`) and clears it.  The detector reads it at each batch's host
 * synchronisation and raises, naming f32_split = 0 (the fp32 MFMA, no range limit) as the remedy.  Always 0 for precision 0. */
int dvid_model_take_range_flag(dvid_model* m, int* exceeded, void* stream);
/* fold FrozenBN, repack to MFMA operand layouts, upload.  Fails (DVID_ERR_STATE) naming the first
 * missing tensor. */
int dvid_model_finalize(dvid_model* m);
/* (re)allocate the activation workspace for batches of up to max_frames frames of height x width
 * (multiples of 32) and boxes_per_frame boxes. */
int dvid_workspace_reserve(dvid_model* m, int max_frames, int height, int width, int boxes_per_frame);
/* Counts the MOVES of this model's workspace buffers (dvid_workspace_reserve growing the arena, dvid_global_memory_project /
 * dvid_global_xattn growing the memory's projection buffers; a buffer's first allocation does not count: nothing can hold its address yet).
 * A caller that captured launches into a hipGraph holds addresses inside the workspace; when the counter differs from its value at
 * capture time the graph must be dropped.  Per model: another model's growth leaves this one's graphs alone.
 * (The reference has no counterpart: its workspace is torch's caching allocator, mega_core/modeling/detector/diffusion_det.py:418-476.) */
unsigned long long dvid_workspace_generation(const dvid_model* m);

/* Number of concurrent sub-batch chains (separate HIP streams inside the library, joined back to the caller's
 * stream before returning) used by the backbone and the heads; 1 = strictly sequential kernels (profiling). */
int dvid_set_chains(dvid_model* m, int nchain);

/* ResNet stem: 1 (default) = the 7x7 / stride-2 convolution runs as a 4x4 / stride-1 convolution over the 2x2 space-to-depth image
 * (16 channels per block, K = 256 packed columns); 0 = over the NHWC8 image (K = 448).  Same products, different summation order. */
int dvid_set_stem_layout(dvid_model* m, int space_to_depth);

/* ---- stages -------------------------------------------------------------------------------- */
/* images: fp32 NCHW [n,3,height,width] in [0,1] (zero padded, un-normalised).  Outputs fp16 NHWC
 * p3 [n,h/8,w/8,256], p4 [n,h/16,w/16,256], p5 [n,h/32,w/32,256]. */
int dvid_backbone_resnet_fpn(dvid_model* m, const float* images, int n, int height, int width, void* p3, void* p4, void* p5,
                             void* stream);
/* The same with the frames as a host array of n device pointers, each an fp32 CHW [3, height, width] frame: the reference hands
 * the detector a list of per-frame tensors and concatenates them (diffusion_det.py:418-421); here nothing is copied. */
int dvid_backbone_resnet_fpn_frames(dvid_model* m, const float* const* frames, int n, int height, int width, void* p3, void* p4, void* p5,
                                    void* stream);

/* Swin-Transformer + FPN (mega_core/modeling/backbone/swintransformer.py:464-751, out_indices (1,2,3)); same
 * inputs/outputs as dvid_backbone_resnet_fpn. */
int dvid_backbone_swin_fpn(dvid_model* m, const float* images, int n, int height, int width, void* p3, void* p4, void* p5,
                           void* stream);
int dvid_backbone_swin_fpn_frames(dvid_model* m, const float* const* frames, int n, int height, int width, void* p3, void* p4, void* p5,
                                  void* stream);

/* One RCNNHead (cond == NULL) or RCNNHead_cond pass.  head_index indexes head_series, or
 * head_series_cond when is_cond.  t: host int64 [n_frames] diffusion timesteps.
 * pro_features may be NULL (-> mean of the RoI features).  Outputs: logits [R,num_classes],
 * boxes [R,4], obj_features [R,hidden] (fp32, R = n_frames*boxes_per_frame).
 * bad_box_flag (device int, may be NULL) is OR-ed with 1 if any predicted box has x2<x1 or y2<y1
 * (the reference's AssertionError at box_head.py:588, reported without a host sync). */
int dvid_rcnn_head(dvid_model* m, int head_index, int is_cond, const void* p3, const void* p4, const void* p5, int n_frames,
                   int height, int width, int boxes_per_frame, const float* boxes, const float* pro_features,
                   const int64_t* t, const float* cond, float* logits, float* boxes_out, float* obj_features,
                   int* bad_box_flag, void* stream);

/* cond[R,hidden] = MHA(query = obj_features[R,hidden], key = value = memory[lk,hidden]).  memory == NULL: use the K/V
 * projections kept by the last dvid_global_memory_project (lk 0 or the same row count). */
int dvid_global_xattn(dvid_model* m, const float* query, int rows, const float* memory, int lk, float* out, void* stream);
/* Project the video's global memory [lk, hidden] to K/V once (box_head.py:366-380 re-projects the same rows on every
 * call); valid until the next call of this function. */
int dvid_global_memory_project(dvid_model* m, const float* memory, int lk, void* stream);

/* ---- stand-alone ops (also used by the parity tests) -------------------------------------- */
int dvid_roialign_v2_multilevel(const void* p3, const void* p4, const void* p5, int n_frames, int height, int width,
                                int channels, const float* boxes, int boxes_per_frame, void* roi_out /* fp16 [R,49,C] */,
                                float* mean_out /* [R,C] or NULL */, void* stream);
/* The DTYPE float32 forms of the stand-alone ops (csrc/f32.hip): fp32 NHWC pyramids -> fp32 [R,49,C] tiles; fp32 conv / linear with
 * w [cout][kpad] fp32, k = (ky*kw+kx)*cin + c, cin % 4 == 0, kpad = round_up(kh*kw*cin, 16) zero-padded, and row_scale [cout] (may be
 * NULL): out channel n = act(acc_n * row_scale[n] + bias[n] + residual) -- the model packs each row times a power of two that puts its
 * largest magnitude in [0.5, 1) and hands 2^-e here, which keeps the split-operand products at fp32 grade for small weights; w_hi / w_lo
 * (fp16 [cout][kpad], may be NULL): w_hi = fp16(w), w_lo = fp16(w - w_hi) -- given, and with option f32_split = 1 (default), the products run
 * as three fp16-MFMA passes over (hi, lo) operand pairs (the activations are split in the kernel); NULL, or f32_split = 0: fp32 MFMA; fp32 attention (head dim 32, head h at columns [32h, 32h+32)); fp32 DynamicConv (params [R][32768] = P1T[64][256] |
 * P2T[256][64]). */
int dvid_roialign_v2_multilevel_f32(const float* p3, const float* p4, const float* p5, int n_frames, int height, int width, int channels,
                                    const float* boxes, int boxes_per_frame, float* roi_out, float* mean_out, void* stream);
int dvid_conv2d_nhwc_f32(const float* in, const float* w, const void* w_hi, const void* w_lo, const float* bias, const float* row_scale, const float* residual,
                         float* out, int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int pad, int kpad, int relu, int residual_mode,
                         void* stream);
int dvid_mha_f32(const float* q, const float* k, const float* v, float* out, int batch, int lq, int lk, int nheads, int q_ld, int kv_ld, int out_ld,
                 int64_t q_bs, int64_t kv_bs, int64_t out_bs, void* stream);
int dvid_dynconv_f32(const float* roi, const float* params, const float* g1, const float* b1, const float* g2, const float* b2, float* out,
                     int rows, void* stream);
int dvid_select_topk_features(const float* logits, int n_frames, int m, int num_classes, int k1, int k2, const float* feats,
                              int hidden, float* out_k1, float* out_k2, void* stream);
/* N(0, 1) draws as a pure function of (key, element index): out[i][e], i < n_images, e < per_image, = element e of the stream
 * keyed key0 + i.  Philox4x32-10 + fp64 Box-Muller rounded to fp32; oracle/noise.py restates it on the CPU (same values). */
int dvid_counter_normal(float* out, int64_t per_image, int n_images, uint64_t key0, void* stream);
int dvid_noise_to_boxes(const float* x, float* boxes, int n, float snr_scale, float img_w, float img_h, void* stream);
/* Box renewal + DDIM update (eta = 1) of diffusion_det.py:559-596 (+ :649-653, :666-672): per frame, boxes whose
 * sigmoid(max logit) > keep_thr are kept (index order), updated as x0*sqrt_ac_next + coef_c*eps + sigma*noise[j],
 * and the tail is refilled from `fresh`.  All tensors [n_frames, m, 4] (logits [n_frames, m, c]). */
int dvid_ddim_renew_step(const float* logits, const float* boxes, const float* x_t, const float* noise, const float* fresh,
                         float* x_next, int n_frames, int m, int c, float img_w, float img_h, float snr_scale,
                         float sqrt_recip_ac, float sqrt_recipm1_ac, float sqrt_ac_next, float coef_c, float sigma, float keep_thr,
                         void* stream);
/* logits [nsets, n_frames, m, c], boxes [nsets, n_frames, m, 4]; outputs [n_frames, nsets*m, ...] sorted by
 * descending score, first counts[f] entries valid.  scratch: >= n_frames*nsets*m*24 bytes. */
int dvid_postproc_topk_nms(const float* logits, const float* boxes, int nsets, int n_frames, int m, int c, float img_w,
                           float img_h, float iou_threshold, int use_nms, float* out_boxes, float* out_scores,
                           int* out_labels, int* out_counts, void* scratch, void* stream);
int dvid_cdist(const float* x, int n, int d, float* dist, void* stream);
/* bs_emul: block size of the reference CUDA launch to emulate for tie-breaking (0 = fps.cu's own rule) */
int dvid_fps_greedy(const float* dist, int n, int m, int bs_emul, int* idx, void* stream);
int dvid_gather_rows(const float* x, const int* idx, float* y, int m, int d, void* stream);
/* NHWC fp16 conv / linear (implicit GEMM on MFMA): w is [cout][kpad] fp16 with k = (ky*kw+kx)*cin + c,
 * kpad = round_up(kh*kw*cin, 64).  residual_mode: 0 none, 1 same shape, 2 nearest-x2 upsample.  pad < 0 (stride 1 only): |pad|
 * rows / columns before the image and as many after as keep the output at the input's size (the space-to-depth stem's 4x4 window). */
int dvid_conv2d_nhwc_f16(const void* in, const void* w, const float* bias, const void* residual, void* out, int n, int h,
                         int wd, int cin, int cout, int kh, int kw, int stride, int pad, int kpad, int relu, int out_f32,
                         int residual_mode, void* stream);
/* Everything behind conv1 of a res2 bottleneck block (detectron2 BottleneckBlock, widths 64 -> 64 -> 256, stride 1) as one launch:
 * conv2 3x3 + ReLU, conv3 1x1 + residual + ReLU and -- w1_next != NULL -- the next block's conv1 256 -> next_channels + ReLU
 * (next_channels 64: the next res2 block; 128, without a shortcut in the same launch: res3's first block).  t1 [n,h,wd,64] is the
 * block's conv1 output; w2 [64][576], w3 [256][64], w1_next [next_channels][256] in dvid_conv2d_nhwc_f16's packing (FrozenBN folded,
 * biases fp32).  w_shortcut == NULL: `residual` is the block input [n,h,wd,256]; else `residual` is the 64-channel block input
 * [n,h,wd,64] and the residual is its shortcut convolution w_shortcut [256][64].  out [n,h,wd,256], t1_next [n,h,wd,next_channels]
 * (must not alias t1).  Bit-identical to the same layers run through dvid_conv2d_nhwc_f16 (csrc/bneck.hip). */
int dvid_bottleneck64_tail_f16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* residual,
                               const void* w_shortcut, const float* b_shortcut, const void* w1_next, const float* b1_next, int next_channels,
                               void* out, void* t1_next, int n, int h, int wd, void* stream);
/* The same for the 128-wide blocks of res3 (128 -> 128 -> 512; the weights stream through an LDS ring): t1 [n,h,wd,128], w2 [128][1152],
 * w3 [512][128], residual [n,h,wd,512] (the block input, or the first block's shortcut output), w1_next [128][512], out [n,h,wd,512],
 * t1_next [n,h,wd,128].  w2 == NULL: `t1` is already the block's conv2 output (res3's first block: its 3x3 / stride-2 conv2 runs through
 * dvid_conv2d_nhwc_f16).  Bit-identical to the layer-by-layer launches on the igemm2 kernel (dvid_igemm_set_conv3x3(0)); the chunked 3x3
 * patch kernel sums conv2 in another order. */
int dvid_bottleneck128_tail_f16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* residual,
                                const void* w1_next, const float* b1_next, void* out, void* t1_next, int n, int h, int wd, void* stream);
/* MFMA attention, head_dim 32: fp16 q/k/v with head h at columns [32h, 32h+32) of each row, fp16 out;
 * vt_scratch: >= batch*nheads*32*(round_up(lk,32)+32) halves (receives V transposed per head). */
int dvid_mha_f16(const void* q, const void* k, const void* v, void* out, void* vt_scratch, int batch, int lq, int lk, int nheads,
                 int q_ld, int kv_ld, int out_ld, int64_t q_bs, int64_t kv_bs, int64_t out_bs, void* stream);
int dvid_dynconv(const void* roi, const void* params, const float* g1, const float* b1, const float* g2, const float* b2,
                 void* out, int rows, void* stream);
int dvid_add_layernorm(const float* x, const float* r, const float* g, const float* b, float* y, int rows, int d, int relu,
                       void* stream);
int dvid_nhwc_from_nchw(const float* in, void* out_f16, int n, int h, int w, int c, void* stream);
int dvid_nchw_from_nhwc(const void* in_f16, float* out, int n, int h, int w, int c, void* stream);
int dvid_f32_to_f16(const float* x, void* y, int64_t n, void* stream);

/* Test-time Resize + ToTensor + padding on the device (mega_core/data/transforms/build.py:89-97, transforms.py:31-70:
 * PIL.Image.resize(BILINEAR) through torchvision; structures/image_list.py:54-61): src uint8 [h][w][3] RGB ->
 * out fp32 [3][ph][pw] in [0,1], rows >= oh / columns >= ow zero.  Bit-identical to Pillow: two integer passes with the
 * 22-bit weight tables of diffusionvid_amd/data/transforms.py:resample_tables (bounds int32 [n][2] = first source sample,
 * count; weights int32 [n][ksize]); a table pointer is NULL exactly when that axis keeps its size.  tmp: >= h*ow*3 bytes
 * (unused when ow == w). */
int dvid_resize_u8_to_f32(const void* src_hwc, int h, int w, void* tmp, float* out_chw, int oh, int ow, int ph, int pw,
                          const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize, void* stream);

/* ---- options ------------------------------------------------------------------------------
 * The library's behaviour switches are ONE table set through this ABI, never through the environment (csrc/options.h; the defaults
 * are the benchmarked configuration).  Names: conv3x3, wstat, bneck_fuse (0 off / 1 by the layer's shape rule / 2 wherever the layer
 * type fits), stem_pool, head_tail, ln_rows (0 / 1), igemm_cfg (-1 = tuner, >= 0 forced tile configuration), igemm_tune (-1 / 0 / 1, see
 * dvid_igemm_set_tuning), igemm_generic (0 / 1), f32_split (DTYPE float32 products: 1 = split fp16 operands on the fp16 MFMA, 0 = the fp32
 * MFMA), bneck_lds (diagnostics).  Every choice gives the same values up to fp32 summation
 * order (most are bit-identical; the tests say which).  dvid_effective_config writes "name=value ..." of all options followed by the
 * environment variables the library still reads (DVID_IGEMM_TUNE, DVID_IGEMM_TUNE_CACHE, DVID_CHAINS, DVID_POISON_WORKSPACE) as they
 * are set; bench.py echoes it.  The dvid_igemm_set_* / dvid_set_stem_pool entry points below write the same table (-1 = the default). */
int dvid_set_option(const char* name, int value);
int dvid_get_option(const char* name, int* value);
int dvid_reset_options(void);
int dvid_effective_config(char* buf, int cap);

/* ---- measurement -------------------------------------------------------------------------- */
/* Tile configurations of the implicit-GEMM kernel (all bit-identical in their results): the per-shape tuner picks one;
 * dvid_igemm_set_config(k) forces table entry k wherever it is valid (k = -1: back to the tuner). */
int dvid_igemm_num_configs(void);
int dvid_igemm_set_config(int cfg);
/* Per-shape tile tuning: 1 = the first launch of a new (row bucket, N, K, ...) key times every valid configuration on the
 * launch's own stream (stream sync + ~80 launches once per key; keys bucket the row count 8 steps per octave, so ragged
 * video tails do not create new ones); 0 = never time on the calling path: cached winners (DVID_IGEMM_TUNE_CACHE, earlier
 * launches) or the hand rule; -1 = follow the environment: DVID_IGEMM_TUNE if set, else 0 when DVID_IGEMM_TUNE_CACHE names a
 * non-empty winners file (a deployment that ships one is in serving mode by default), else 1. */
int dvid_igemm_set_tuning(int mode);
/* Shape buckets timed on the calling path in this process so far (each = one stream synchronisation + 4 .. ~40 launches per valid
 * configuration): a serving process wants this to stop growing after warm-up -- ragged video tails whose launches are under two rounds of
 * the CUs inherit a tuned bucket only within a quarter octave and may add passes mid-stream (bench.py reports the count per run). */
long long dvid_igemm_tuning_passes(void);
/* 3x3 / stride-1 / pad-1 convolutions with Cin % 32 == 0 and Cout % 128 == 0 (the bottleneck conv2 layers of res3-res5, the FPN
 * output convolutions) on the halo-staged kernel (csrc/conv3x3.hip: the 8 x 32 output patch's input pixels are staged once per
 * 32-channel chunk and serve all nine taps): 1 = on where the shape rule prefers it (W within 1/8 of a multiple of 32, maps of at least
 * 512 pixels -- a function of the layer and the image size, not of the number of frames in the launch), 2 = on wherever the layer type fits (tests), 0 = off (the igemm2 kernel), -1 = the default
 * (1).  The choice depends on the layer's shape only; the two kernels differ in fp32 summation order. */
int dvid_igemm_set_conv3x3(int mode);
/* 1x1 convolutions / linear layers with K in {128, 256} (512 without a residual) and N a multiple of 256 (bottleneck conv3 + residual, the decoder's
 * dynamic_layer and linear1) on the weight-stationary kernel (csrc/wstat.hip: a workgroup keeps the weights of 256 output channels in
 * registers and streams its rows through a DMA ring; epilogue straight from the accumulator layout): 1 = on for launches large
 * enough for 256 persistent workgroups, 2 = wherever the layer type fits (tests), 0 = off (igemm2), -1 = the default (1).
 * Bit-identical to igemm2. */
int dvid_igemm_set_wstat(int mode);

/* res2 / res3 bottleneck blocks behind their conv1 as one launch each (csrc/bneck.hip: dvid_bottleneck64_tail_f16 /
 * dvid_bottleneck128_tail_f16 inside the ResNet backbone): 1 = where the shape rule of the 3x3 patch kernels holds for the stage's map
 * (a function of the image size only), 2 = whenever the stage is made of such bottlenecks (tests), 0 = off (layer-by-layer launches),
 * -1 = the default (1).  res2: bit-identical to the layer-by-layer launches; res3: to those on igemm2 (see above). */
int dvid_igemm_set_bottleneck_fusion(int mode);

/* The ResNet stem (7x7 / stride 2 as a 4x4 convolution over the space-to-depth image, + FrozenBN + ReLU) and the 3x3 / stride-2 max pool
 * behind it (detectron2 BasicStem, reached from mega_core/modeling/detector/diffusion_det.py:427) as ONE launch: 1 = on, 0 = two launches,
 * -1 = the default (1).  Bit-identical either way (csrc/conv3x3.hip: stem_pool_kernel). */
int dvid_set_stem_pool(int mode);

/* When enabled, every igemm launch is bracketed by HIP events on its stream; dvid_profile_read
 * synchronises those events and returns totals since the last reset. */
int dvid_profile_enable(int on);
int dvid_profile_reset(void);
int dvid_profile_read(double* igemm_ms, double* igemm_flop, int64_t* igemm_launches);
/* sum over the recorded launches of the algorithmic HBM bytes (input + weights + output + residual, each touched once) */
int dvid_profile_read_bytes(double* igemm_alg_bytes);
/* CSV (kernel,family,M,N,K,taps,stride,res_mode,ms,tflops,alg_mbytes,alg_gbs), one line per recorded launch: the implicit-GEMM family
 * (family = 1: what dvid_profile_read sums) and the heads' / backbone's other kernels (RoIAlign, DynamicConv, attention, head tail, max pool) */
int dvid_profile_dump(const char* path);

#ifdef __cplusplus
}
#endif
#endif
