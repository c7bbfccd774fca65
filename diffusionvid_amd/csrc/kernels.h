// Internal launcher declarations (C++), one per HIP kernel family.  The public C ABI is
// include/dvid_hip.h; these are what the runtime (model.hip) and the C-ABI op wrappers call.
#pragma once
#include "common.h"

struct IgemmParams {
    const half_t* in;    // NHWC fp16 [N,H,W,Cin]  (Linear: [M,K] as N=M, H=W=1, Cin=K)
    const half_t* w;     // [Cout][Kpad], k = (ky*KW + kx)*Cin + c
    const float* bias;   // [Cout] or nullptr
    const void* res;     // residual (fp16, or fp32 if res_f32), see res_mode
    void* out;           // [M][ldc] fp16 (or fp32 if out_f32)
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, Kpad, ntaps, ldc, alg_k;
    int relu, out_f32, res_mode, res_f32;   // relu: 0 none, 1 ReLU, 2 exact GELU; res_mode: 0 none, 1 same shape, 2 nearest-x2 upsample
    int splitk;                             // > 1: K split over `splitk` workgroups per tile, fp32 partials (no bias/relu/residual)
    long split_stride;                      //       written to out + split * split_stride (elements)
    int tiles_m, tiles_n;                   // filled by the launcher
};
int dvid_igemm_launch(const IgemmParams& p, hipStream_t s);   // igemm2.hip: picks the kernel family (wstat / conv3x3 / igemm2) by the layer's shape
int dvid_igemm2_launch(const IgemmParams& p, hipStream_t s);  // igemm2.hip: the igemm2 kernel with its per-shape tuned tile configuration

// conv3x3.hip: 3x3 / stride-1 / pad-1 layers with the input patch + halo staged once per channel chunk (all nine taps read it)
bool dvid_conv3x3_halo_supported(const IgemmParams& p);   // the layer type fits the kernel
bool dvid_conv3x3_halo_preferred(const IgemmParams& p);   // ... and the shape rule (patch grid waste, patches per CU) picks it
int dvid_conv3x3_halo_launch(const IgemmParams& p, hipStream_t s);
// ... the space-to-depth stem + ReLU + 3x3 / stride-2 max pool as one launch (p.out = the pooled map); bit-identical to the two launches
bool dvid_stem_pool_supported(const IgemmParams& p);
int dvid_stem_pool_launch(const IgemmParams& p, hipStream_t s);

// wstat.hip: short-K / wide-N 1x1 layers with the weights stationary in registers (bit-identical to igemm2)
bool dvid_wstat_supported(const IgemmParams& p);
bool dvid_wstat_preferred(const IgemmParams& p);
int dvid_wstat_launch(const IgemmParams& p, hipStream_t s);

// bneck.hip: the tail of a res2 bottleneck block (conv2 3x3 64 -> 64, conv3 64 -> 256 + residual or shortcut convolution, ReLU and the
// next block's conv1 256 -> 64) as one launch; bit-identical to the layer-by-layer launches.  ws == null: the residual is `res`
// [rows][W][256]; else `res` is the 64-channel block input and the residual is its shortcut convolution.  w1n == null: no next conv1.
bool dvid_bneck64_tail_preferred(int H, int W);          // the shape rule (a function of the map size only), for either width
// 128-wide blocks (res3: 128 -> 128 -> 512; weights streamed through an LDS ring).  w2 == null: `t1` is the conv2 output (res3's first
// block, whose 3x3 / stride-2 conv2 runs as its own launch); w1n == null: no next conv1.  Bit-identical to the layer-by-layer launches
// on igemm2 (the chunked 3x3 patch kernel sums in another order).
int dvid_bneck128_tail_launch(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res,
                              const half_t* w1n, const float* b1n, half_t* out, half_t* t1n, int n, int H, int W, hipStream_t s);
int dvid_bneck64_tail_launch(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res,
                             const half_t* ws, const float* bs, const half_t* w1n, const float* b1n, int n_next, half_t* out, half_t* t1n,
                             int n, int H, int W, hipStream_t s);      // n_next: rows of w1n (64; 128 without a shortcut)

// elementwise.hip
// per-frame source pointers of one image-prep launch (passed by value as a kernel argument)
struct FrameTable {
    static constexpr int kMax = 32;
    const float* p[kMax];
};
// frames[i]: fp32 CHW [3, h, w] in [0, 1] of frame i (a host array of n device pointers; the frames need not be contiguous)
int dvid_prep_images_launch(const float* const* frames, half_t* nhwc8, int n, int h, int w, const float* mean, const float* inv_std,
                            hipStream_t s);
int dvid_prep_images_s2d_launch(const float* const* frames, half_t* s2d16, int n, int h, int w, const float* mean, const float* inv_std,
                                hipStream_t s);
int dvid_maxpool3x3s2_launch(const half_t* in, half_t* out, int n, int h, int w, int c, hipStream_t s);
int dvid_nchw_from_nhwc_launch(const half_t* in, float* out, int n, int h, int w, int c, hipStream_t s);
int dvid_nhwc_from_nchw_launch(const float* in, half_t* out, int n, int h, int w, int c, hipStream_t s);
// y = LN(x + r) * g + b, rows of D (= 256 or 64-multiple <= 1024); writes fp32 and/or fp16 copies
// x may be `nsplit` split-K partial slabs (x + s * split_stride) plus a per-column bias `xbias` (or nsplit 1, null)
int dvid_add_layernorm_launch(const float* x, const float* r, const float* g, const float* b, float* y32, half_t* y16,
                              int rows, int d, int relu, hipStream_t s, int nsplit = 1, long split_stride = 0,
                              const float* xbias = nullptr);
int dvid_f32_to_f16_launch(const float* x, half_t* y, long n, hipStream_t s);
int dvid_resize_u8_launch(const unsigned char* src, int h, int w, unsigned char* tmp, float* out, int oh, int ow, int ph, int pw,
                          const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize, hipStream_t s);
// fc = x * (scale[frame] + 1) + shift  (shift per frame [B,D] or per row [R,D])
int dvid_modulate_launch(const float* x, const float* scale, int scale_ld, const float* shift, int shift_per_row, int shift_ld,
                         half_t* y16, int rows, int rows_per_frame, int d, hipStream_t s);
int dvid_silu_f16_launch(const float* x, half_t* y, long n, hipStream_t s);

// roialign.hip
struct RoiLevels {
    const half_t* feat[3];
    int h[3], w[3];
    float scale[3];
};
int dvid_roialign_launch(const RoiLevels& lv, int channels, const float* boxes, int n_img, int boxes_per_img, half_t* roi_out,
                         float* mean_out, hipStream_t s);

// attention.hip: out[b][q][:] = softmax(Q K^T * scale) V per head on MFMA: fp16 q/k/v (head h at columns h*32..), fp16 out; vt_scratch >= batch*nheads*32*(round_up(lk,32)+32) halves
int dvid_mha_mfma_launch(const half_t* q, const half_t* k, const half_t* v, half_t* out, half_t* vt_scratch, int batch, int lq,
                         int lk, int nheads, int q_ld, int kv_ld, int out_ld, long q_bs, long kv_bs, long out_bs, hipStream_t s);

constexpr int SWIN_RELBIAS_PITCH = 64;      // floats per query row of the relative-position bias table [heads][49][64] (keys 49..63 = 0)
int dvid_swin_window_attn_launch(const half_t* qkv, const half_t* qkv_bias16, const float* relbias, half_t* out, int batch, int H,
                                 int W, int C, int nheads, int shift, hipStream_t s);
int dvid_patch_merge_ln_launch(const float* x, const float* g, const float* b, half_t* y16, int B, int H, int W, int C, hipStream_t s,
                               float* y32 = nullptr);          // y32: an fp32 copy of the result (DTYPE float32: y16 null)

// dynconv.hip
int dvid_dynconv_launch(const half_t* roi, const half_t* params, const float* g1, const float* b1, const float* g2,
                        const float* b2, half_t* out, int rows, hipStream_t s);
// RoIAlign gathered straight into DynamicConv's LDS tile (csrc/dynconv.hip, FUSED_ROI): bit-identical to dvid_roialign_launch + dvid_dynconv_launch
int dvid_dynconv_roi_launch(const RoiLevels& lv, int channels, const float* boxes, int n_img, int boxes_per_img, const half_t* params,
                            const float* g1, const float* b1, const float* g2, const float* b2, half_t* out, hipStream_t s);

// headtail.hip: FFN + norm3 + modulation + cls / reg towers + class_logits + bboxes_delta + apply_deltas of one RCNNHead pass as one
// row-tile kernel.  Every `*f` weight is in MFMA fragment order (model.hip: make_frags): [n-tile of 32 rows][K step of 16][lane][8].
struct HeadTailParams {
    const half_t* x16;          // [R, 256] fp16   norm2 output (operand of linear1)
    const float* obj32;         // [R, 256] fp32   the same rows, the FFN's residual
    const half_t* w1f;          // linear1 [dff][256]
    const float* b1;
    const half_t* w2f;          // linear2 [256][dff]
    const float* b2;
    const float* n3g;           // norm3
    const float* n3b;
    const float* scale;         // block_time_mlp rows: scale (and, plain head, shift at + 256) of frame f at scale + f * ss_stride
    int ss_stride, rows_per_frame;
    const float* cond32;        // [R, 256] fp32 or null (RCNNHead_cond: shift = c_mlp(SiLU(cond)))
    const half_t* wcf;          // c_mlp.1 [256][256]
    const float* bc;
    int num_cls, num_reg, num_classes, dff;
    const half_t* clsf[4];
    const float* clsg[4];
    const float* clsb[4];
    const half_t* regf[4];
    const float* regg[4];
    const float* regb[4];
    const half_t* wlogf;        // class_logits, rows zero-padded to a multiple of 32
    const float* blog;
    const half_t* wdelf;        // bboxes_delta, 4 rows zero-padded to 32
    const float* bdel;
    const float* boxes;         // [R, 4] input boxes
    float* obj_out;             // [R, 256] fp32 obj_features
    float* logits;              // [R, num_classes]
    float* boxes_out;           // [R, 4]
    int* bad_flag;
    long R;
    float wx, wy, ww, wh, clamp;
};
bool dvid_head_tail_supported(int hidden, int dff, int num_cls, int num_reg, int num_classes);
int dvid_head_tail_launch(const HeadTailParams& p, hipStream_t s);

// boxes.hip
int dvid_apply_deltas_launch(const float* deltas, int delta_ld, const float* boxes, float* out, int n, float wx, float wy, float ww,
                             float wh, float clamp, int* bad_flag, hipStream_t s);
int dvid_counter_normal_launch(float* out, long per_image, int n_images, uint64_t key0, hipStream_t s);
int dvid_noise_to_boxes_launch(const float* x, float* boxes, int n, float scale, float w, float h, hipStream_t s);
int dvid_topk_mask_launch(const float* logits, int n_img, int m, int c, int k1, int k2, const float* feats, int d, float* out1,
                          float* out2, hipStream_t s);

int dvid_ddim_renew_launch(const float* logits, const float* boxes, const float* xt, const float* noise, const float* fresh,
                           float* out, int n_img, int m, int c, float w, float h, float scale, float sra, float srm1, float sqrt_an,
                           float cc, float sigma, float thr, hipStream_t s);

// postproc.hip
int dvid_topk_candidates_launch(const float* logits, const float* boxes, int n_img, int nsets, int m, int c, float* cand_boxes,
                                float* cand_scores, int* cand_labels, hipStream_t s);
int dvid_nms_frames_launch(const float* cand_boxes, const float* cand_scores, const int* cand_labels, int n_img, int n, float img_w,
                           float img_h, float iou, int use_nms, int out_cap, float* out_boxes, float* out_scores, int* out_labels,
                           int* out_counts, hipStream_t s);

// fps.hip
int dvid_cdist_launch(const float* x, int n, int d, float* dist, hipStream_t s);
int dvid_fps_launch(const float* dist, int n, int m, int bs_emul, int* idx, hipStream_t s);
int dvid_gather_rows_launch(const float* x, const int* idx, float* y, int m, int d, hipStream_t s);

// f32.hip: the DTYPE float32 path (fp32 storage, v_mfma_f32_32x32x2_f32 products)
struct F32GemmParams {
    const float* in;     // NHWC fp32 [N,H,W,Cin], Cin % 4 == 0  (Linear: [M,K] as N = M, H = W = 1, Cin = K)
    const float* w;      // [Cout][Kpad], k = (ky*KW + kx)*Cin + c, zero beyond K; Kpad % 16 == 0
    const half_t* w_hi;  // the same rows split for the split-operand kernel: w_hi = fp16(w), w_lo = fp16(w - w_hi) (null: the fp32-MFMA kernel runs)
    const half_t* w_lo;
    const float* bias;   // [Cout] or nullptr
    int* range_flag;     // device int or nullptr: the split-operand kernel ORs 1 into it when an activation's magnitude exceeds the fp16 range (65504)
    const float* wscale; // [Cout] or nullptr: the accumulator of channel n is multiplied by wscale[n] (undoes a power-of-two scaling of the packed row)
    const float* res;    // fp32 residual, see res_mode
    float* out;          // [M][ldc]
    int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int M, K, Kpad, ldc;
    int relu, res_mode;  // relu: 0 none, 1 ReLU, 2 exact GELU; res_mode: 0 none, 1 same shape, 2 nearest-x2 upsample
    int tiles_m, tiles_n;   // filled by the launcher
};
int dvid_f32_igemm_launch(const F32GemmParams& p, hipStream_t s);
// csrc/f32_wstat.hip: the weight-stationary form of the split-operand kernel (bit-identical to it)
bool dvid_f32_wstat_supported(const F32GemmParams& p);
bool dvid_f32_wstat_preferred(const F32GemmParams& p);
int dvid_f32_wstat_tile_rows(const F32GemmParams& p);
int dvid_f32_wstat_launch_tiles(const F32GemmParams& p, hipStream_t s);
// csrc/f32_conv3x3.hip: 3x3 / stride-1 layers with the halo staged and split once per channel chunk (another summation order than the tiled kernel)
bool dvid_f32_conv3x3_supported(const F32GemmParams& p);
int dvid_f32_conv3x3_launch(const F32GemmParams& p, hipStream_t s);
struct RoiLevels32 {
    const float* feat[3];
    int h[3], w[3];
    float scale[3];
};
int dvid_f32_prep_images_launch(const float* const* frames, float* nhwc4, int n, int h, int w, const float* mean, const float* std_, hipStream_t s);
int dvid_f32_maxpool3x3s2_launch(const float* in, float* out, int n, int h, int w, int c, hipStream_t s);
int dvid_f32_silu_launch(const float* x, float* y, long n, hipStream_t s);
int dvid_f32_modulate_launch(const float* x, const float* scale, int scale_ld, const float* shift, int shift_per_row, int shift_ld, float* y,
                             int rows, int rows_per_frame, int d, hipStream_t s);
int dvid_f32_roialign_launch(const RoiLevels32& lv, int channels, const float* boxes, int n_img, int boxes_per_img, float* roi_out,
                             float* mean_out, hipStream_t s);
// q / k / v fp32 with head h at columns [32 h, 32 h + 32) of a row (head dim 32)
int dvid_f32_mha_launch(const float* q, const float* k, const float* v, float* out, int batch, int lq, int lk, int nheads, int q_ld, int kv_ld,
                        int out_ld, long q_bs, long kv_bs, long out_bs, hipStream_t s);
// Swin window attention (shift + 7 x 7 windows + relative-position bias + region mask), fp32: qkv [B*H*W][3C], qkv_bias [3C] (q / k / v of a
// padded window position), relbias [nheads][49][SWIN_RELBIAS_PITCH], out [B*H*W][C]
int dvid_f32_swin_window_attn_launch(const float* qkv, const float* qkv_bias, const float* relbias, float* out, int batch, int H, int W, int C,
                                     int nheads, int shift, hipStream_t s);
// roi [R][49][256], params [R][32768] as P1T[64][256] | P2T[256][64], out [R][49][256]; range_flag (device int or null): the split-operand
// form ORs 1 into it when a RoI / parameter magnitude exceeds the fp16 range
int dvid_f32_dynconv_launch(const float* roi, const float* params, const float* g1, const float* b1, const float* g2, const float* b2,
                            float* out, int rows, int* range_flag, hipStream_t s);
