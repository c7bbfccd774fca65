// The tail of an RCNNHead / RCNNHead_cond pass as ONE row-tile kernel (box_head.py:524-548 / :636-664): FFN (linear1 + ReLU +
// linear2) + residual + norm3, the time (and cond) modulation, the cls tower (Linear + LayerNorm + ReLU) + class_logits, the reg
// tower (3 x Linear + LayerNorm + ReLU) + bboxes_delta and apply_deltas -- 17 launches of the layer-by-layer form (two GEMM
// launches + one LayerNorm launch per tower layer, every intermediate a round trip through HBM in fp32 AND fp16).
//
// A workgroup (8 waves) owns 32 box rows.  The 256-wide activation row never leaves the CU: it sits in LDS as the fp16
// A-operand tile of the next product ([rows][256] halves, 16 bytes of padding per row: conflict-free ds_read_b128 fragments), the
// fp32 accumulators go through one LDS tile for the row statistics of the LayerNorms (4 or 8 lanes per row, xor shuffles), and
// the weights stream from L2 in MFMA fragment order (1 KiB per 32 x 16 fragment, one 16-byte global load per lane) -- 2.7 MB per tile,
// re-read by every workgroup through the L2.  The kernel is paced by the latency of those reads, not by the matrix pipe: wave (kh, wq)
// owns output columns [64 wq, 64 wq + 64) and the K steps [8 kh, 8 kh + 8) of every 256-deep product and requests all 16 of its
// fragments before its first MFMA (128 KiB of reads in flight per CU; the first version -- 4 waves, whole K per wave, 4 steps of
// prefetch -- kept 32 KiB in flight and ran the 2400-row launch of a one-batch call no faster than the 17 launches it replaces:
// 1222 vs 1237 frames/s).  Products are computed transposed (weight fragment first), as in igemm2, so a lane holds 4 consecutive
// channels of one row.  The FFN runs in 256-column chunks of its hidden layer: chunk c of
// linear1 (+ bias, ReLU, fp16) is written to LDS and immediately consumed as K columns [256 c, 256 c + 256) of linear2.
//
// Arithmetic follows the layer-by-layer path (fp16 operands, fp32 accumulate, fp32 LayerNorm statistics two-pass on registers,
// fp16 rounding of every activation that feeds a product, fp32 residual stream); summation orders differ, so results agree to
// rounding, not bit for bit (tests/test_gpu_kernels.py::test_head_tail_fused_matches_layerwise).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int D = 256;                 // HIDDEN_DIM (dvid_model_create refuses anything else)
constexpr int KSN = D / 16;            // K steps of one 256-deep product
constexpr int APITCH = D * 2 + 16;     // bytes per row of an fp16 operand tile
constexpr int CPITCH = D + 4;          // floats per row of the fp32 accumulator tile
constexpr int FRAG = 64 * 8;           // halves per 32 x 16 weight fragment

// acc[i][j] += W[n-tile nt0 + j] . A[m-tile i]^T over the 8 K steps [ks0, ks0 + 8) of a weight matrix with ks_total K steps; the A
// fragments are K steps [ka0, ka0 + 8) of the LDS tile.  All 8 NJ weight fragments are requested before the first MFMA: a wave
// keeps 8 NJ KiB of L2 reads in flight per product (the kernel is paced by that latency, not by the matrix pipe).
template <int NJ>
struct Frags {
    half8 b[8][NJ];
};
// request the 8 NJ weight fragments of n-tiles [nt0, nt0 + NJ), K steps [ks0, ks0 + 8) (nothing waits here)
template <int NJ>
__device__ __forceinline__ void load_frags(Frags<NJ>& f, const half_t* __restrict__ wf, int ks0, int ks_total, int nt0, int lane) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const half_t* bp = wf + (((long)(nt0 + j) * ks_total + ks0) * 64 + lane) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) f.b[u][j] = *reinterpret_cast<const half8*>(bp + u * FRAG);
    }
}
// acc[i][j] += W-fragments . A[m-tile i]^T with the A fragments taken from K steps [ka0, ka0 + 8) of the LDS tile
template <int MT, int NJ>
__device__ __forceinline__ void mma_frags(const char* a_tile, int ka0, const Frags<NJ>& f, int lane, float16v (&acc)[MT][NJ]) {
    const char* ap = a_tile + (lane & 31) * APITCH + (lane >> 5) * 16 + ka0 * 32;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        half8 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const half8*>(ap + i * 32 * APITCH + u * 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[u][j], a[i], acc[i][j], 0, 0, 0);
    }
}
template <int MT, int NJ>
__device__ __forceinline__ void wave_gemm8(const char* a_tile, int ka0, const half_t* __restrict__ wf, int ks0, int ks_total, int nt0, int lane,
                                           float16v (&acc)[MT][NJ]) {
    Frags<NJ> f;
    load_frags<NJ>(f, wf, ks0, ks_total, nt0, lane);
    mma_frags<MT, NJ>(a_tile, ka0, f, lane, acc);
}

template <int MT, int NJ>
__device__ __forceinline__ void zero_acc(float16v (&acc)[MT][NJ]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Eight waves: wave = (kh, wq).  wq owns output columns [64 wq, 64 wq + 64) of every 256-wide product, kh the K steps [8 kh, 8 kh + 8) of
// its 16: the two halves of a sum meet in the fp32 LDS tile (kh = 0 writes, barrier, kh = 1 adds, barrier).
// accumulator layout: lane (row = lane & 31) holds columns 8 r4 + 4 (lane >> 5) + {0..3} of n-tile j
template <int MT>
__device__ __forceinline__ void reduce_to_ctile(float* C, int kh, int wq, int lane, const float16v (&acc)[MT][2]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (kh == half) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = i * 32 + (lane & 31), col = (2 * wq + j) * 32 + 8 * r4 + 4 * (lane >> 5);
                        float4v v = (float4v){acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                        if (half) v += *reinterpret_cast<const float4v*>(C + row * CPITCH + col);
                        *reinterpret_cast<float4v*>(C + row * CPITCH + col) = v;
                    }
        }
        __syncthreads();
    }
}

template <int MT>
__global__ __launch_bounds__(512) void head_tail_kernel(HeadTailParams p) {
    constexpr int BM = 32 * MT;
    constexpr int TPR = 512 / BM;            // lanes per row in the row passes (16 or 8, adjacent lanes)
    constexpr int VPT = D / 4 / TPR;         // float4 per lane and row (4 or 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bufX = smem;                                   // fp16 tile: x, later fc
    char* bufH = smem + BM * APITCH;                     // fp16 tile: hidden chunk, SiLU(cond), tower activations
    float* C = reinterpret_cast<float*>(smem + 2 * BM * APITCH);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wq = wave & 3;
    const long row0 = (long)blockIdx.x * BM;
    const int prow = tid / TPR, pseg = tid % TPR;        // row-pass coordinates
    const long grow = row0 + prow;
    const bool prow_ok = grow < p.R;
    const long grow_c = prow_ok ? grow : (long)p.R - 1;
    const long frame = grow_c / p.rows_per_frame;

    // ---- x tile -> LDS
    for (int v = tid; v < BM * (D / 8); v += 512) {
        const int r = v / (D / 8), c8 = v % (D / 8);
        const long gr = row0 + r < p.R ? row0 + r : (long)p.R - 1;
        *reinterpret_cast<half8*>(bufX + r * APITCH + c8 * 16) = *reinterpret_cast<const half8*>(p.x16 + gr * D + c8 * 8);
    }
    __syncthreads();

    // ---- FFN: hidden chunk c of linear1 (+ bias, ReLU, fp16) -> LDS -> K columns of linear2.  linear2's sum stays split over the two
    // K halves of every chunk until the end.
    float16v accY[MT][2];
    zero_acc<MT, 2>(accY);
    const int nchunk = p.dff / D, ks2_total = p.dff / 16;
    // The weight fragments of the NEXT product are requested before the barriers / LDS passes that separate it from the current one:
    // their L2 latency (the pacing term of this kernel) runs under the reduction of the current product.
    Frags<2> f1, f2;
    load_frags<2>(f1, p.w1f, 8 * kh, KSN, 2 * wq, lane);
    for (int c = 0; c < nchunk; ++c) {
        float16v accH[MT][2];
        zero_acc<MT, 2>(accH);
        mma_frags<MT, 2>(bufX, 8 * kh, f1, lane, accH);
        load_frags<2>(f2, p.w2f, c * KSN + 8 * kh, ks2_total, 2 * wq, lane);
        // halves of the hidden chunk meet in the fp32 tile (free during the FFN); the barrier also ends the previous chunk's reads of bufH
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = i * 32 + (lane & 31), col = (2 * wq + j) * 32 + 8 * r4 + 4 * (lane >> 5);
                        *reinterpret_cast<float4v*>(C + row * CPITCH + col) =
                            (float4v){accH[i][j][4 * r4], accH[i][j][4 * r4 + 1], accH[i][j][4 * r4 + 2], accH[i][j][4 * r4 + 3]};
                    }
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = i * 32 + (lane & 31), col = (2 * wq + j) * 32 + 8 * r4 + 4 * (lane >> 5);
                        const float4v bb = *reinterpret_cast<const float4v*>(p.b1 + c * D + col);
                        const float4v other = *reinterpret_cast<const float4v*>(C + row * CPITCH + col);
                        half4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (half_t)fmaxf(accH[i][j][4 * r4 + e] + other[e] + bb[e], 0.f);
                        *reinterpret_cast<half4*>(bufH + row * APITCH + col * 2) = h;
                    }
        }
        if (c + 1 < nchunk) load_frags<2>(f1, p.w1f, 8 * kh, KSN, (c + 1) * 8 + 2 * wq, lane);
        __syncthreads();
        mma_frags<MT, 2>(bufH, 8 * kh, f2, lane, accY);
    }
    __syncthreads();                                      // the last chunk's reads of the fp32 tile's halves (kh = 0) are done
    reduce_to_ctile<MT>(C, kh, wq, lane, accY);

    // ---- row pass 1: obj_features = LayerNorm(obj + linear2 + b2) (norm3) -> global fp32; then the modulation
    //   plain head:  fc = obj_features * (scale + 1) + shift           (box_head.py:533-536) -> bufX
    //   cond head:   SiLU(cond) -> bufH for the c_mlp product; obj_features stay in the fp32 tile for its epilogue
    {
        float4v v[VPT];
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int col = (q * TPR + pseg) * 4;
            v[q] = *reinterpret_cast<const float4v*>(C + prow * CPITCH + col) + *reinterpret_cast<const float4v*>(p.b2 + col) +
                   *reinterpret_cast<const float4v*>(p.obj32 + grow_c * D + col);
            sum += v[q][0] + v[q][1] + v[q][2] + v[q][3];
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / D;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[q][e] - mean;
                sq += t * t;
            }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq / D + 1e-5f);
        const float* ss = p.scale + frame * p.ss_stride;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int col = (q * TPR + pseg) * 4;
            const float4v gg = *reinterpret_cast<const float4v*>(p.n3g + col), bb = *reinterpret_cast<const float4v*>(p.n3b + col);
            float4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[q][e] - mean) * rstd * gg[e] + bb[e];
            if (prow_ok) *reinterpret_cast<float4v*>(p.obj_out + grow * D + col) = o;
            if (p.cond32) {
                *reinterpret_cast<float4v*>(C + prow * CPITCH + col) = o;
                const float4v cv = *reinterpret_cast<const float4v*>(p.cond32 + grow_c * D + col);
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)(cv[e] / (1.f + __expf(-cv[e])));
                *reinterpret_cast<half4*>(bufH + prow * APITCH + col * 2) = h;
            } else {
                const float4v sc = *reinterpret_cast<const float4v*>(ss + col), sh = *reinterpret_cast<const float4v*>(ss + D + col);
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[e] * (sc[e] + 1.f) + sh[e]);
                *reinterpret_cast<half4*>(bufX + prow * APITCH + col * 2) = h;
            }
        }
    }
    __syncthreads();
    if (p.cond32) {
        // shift = c_mlp(SiLU(cond)) per row (box_head.py:643-647); fc = obj_features * (scale + 1) + shift from the accumulator layout.
        // The fp32 tile holds obj_features here, so this one product is not split over K: the kh = 0 waves run all 16 K steps.
        if (kh == 0) {
            float16v acc[MT][2];
            zero_acc<MT, 2>(acc);
            wave_gemm8<MT, 2>(bufH, 0, p.wcf, 0, KSN, 2 * wq, lane, acc);
            wave_gemm8<MT, 2>(bufH, 8, p.wcf, 8, KSN, 2 * wq, lane, acc);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = i * 32 + (lane & 31), col = (2 * wq + j) * 32 + 8 * r4 + 4 * (lane >> 5);
                        const long gr = row0 + row < p.R ? row0 + row : (long)p.R - 1;
                        const float* ssr = p.scale + (gr / p.rows_per_frame) * p.ss_stride;
                        const float4v sc = *reinterpret_cast<const float4v*>(ssr + col), bc = *reinterpret_cast<const float4v*>(p.bc + col);
                        const float4v of = *reinterpret_cast<const float4v*>(C + row * CPITCH + col);
                        half4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (half_t)(of[e] * (sc[e] + 1.f) + (acc[i][j][4 * r4 + e] + bc[e]));
                        *reinterpret_cast<half4*>(bufX + row * APITCH + col * 2) = h;
                    }
        }
        __syncthreads();
    }

    // ---- towers: Linear (no bias) + LayerNorm + ReLU, activations in bufH; the cls tower first, then the reg tower, both from fc
    // `ft` holds the fragments of the layer about to run (requested during the previous layer's reduction); `next` = the weights of the
    // layer after it (null: none), requested before this layer's barriers
    Frags<2> ft;
    auto tower_layer = [&](const char* in, const half_t* next, const float* g, const float* b) {
        float16v acc[MT][2];
        zero_acc<MT, 2>(acc);
        mma_frags<MT, 2>(in, 8 * kh, ft, lane, acc);
        if (next) load_frags<2>(ft, next, 8 * kh, KSN, 2 * wq, lane);
        __syncthreads();                                  // every wave is done reading `in` (bufH is rewritten below) and the fp32 tile
        reduce_to_ctile<MT>(C, kh, wq, lane, acc);
        float4v v[VPT];
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            v[q] = *reinterpret_cast<const float4v*>(C + prow * CPITCH + (q * TPR + pseg) * 4);
            sum += v[q][0] + v[q][1] + v[q][2] + v[q][3];
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / D;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[q][e] - mean;
                sq += t * t;
            }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq / D + 1e-5f);
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int col = (q * TPR + pseg) * 4;
            const float4v gg = *reinterpret_cast<const float4v*>(g + col), bb = *reinterpret_cast<const float4v*>(b + col);
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)fmaxf((v[q][e] - mean) * rstd * gg[e] + bb[e], 0.f);
            *reinterpret_cast<half4*>(bufH + prow * APITCH + col * 2) = h;
        }
        __syncthreads();
    };

    // layer sequence: cls tower, then reg tower (the class_logits / bboxes_delta products are single-wave and load their own fragments)
    const half_t* seq[9];
    int nseq = 0;
    for (int i = 0; i < p.num_cls; ++i) seq[nseq++] = p.clsf[i];
    for (int i = 0; i < p.num_reg; ++i) seq[nseq++] = p.regf[i];
    seq[nseq] = nullptr;
    if (nseq) load_frags<2>(ft, seq[0], 8 * kh, KSN, 2 * wq, lane);
    for (int i = 0; i < p.num_cls; ++i) tower_layer(i ? bufH : bufX, seq[i + 1], p.clsg[i], p.clsb[i]);
    {
        // class_logits (num_classes <= 64 rows, zero-padded to whole 32-row tiles): one n-tile per wave, all 16 K steps
        const int ntl = (p.num_classes + 31) / 32;
        if (wave < ntl) {
            const char* in = p.num_cls ? bufH : bufX;
            float16v acc[MT][1];
            zero_acc<MT, 1>(acc);
            wave_gemm8<MT, 1>(in, 0, p.wlogf, 0, KSN, wave, lane, acc);
            wave_gemm8<MT, 1>(in, 8, p.wlogf, 8, KSN, wave, lane, acc);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const long gr = row0 + i * 32 + (lane & 31);
                if (gr < p.R) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int col = wave * 32 + 8 * r4 + 4 * (lane >> 5) + e;
                            if (col < p.num_classes) p.logits[gr * p.num_classes + col] = acc[i][0][4 * r4 + e] + p.blog[col];
                        }
                }
            }
        }
    }
    for (int i = 0; i < p.num_reg; ++i) tower_layer(i ? bufH : bufX, seq[p.num_cls + i + 1], p.regg[i], p.regb[i]);
    if (wave == 0) {
        // bboxes_delta (4 rows, zero-padded to one tile): lanes 0..31 hold (dx, dy, dw, dh) of their row; apply_deltas (box_head.py:550-590)
        const char* in = p.num_reg ? bufH : bufX;
        float16v acc[MT][1];
        zero_acc<MT, 1>(acc);
        wave_gemm8<MT, 1>(in, 0, p.wdelf, 0, KSN, 0, lane, acc);
        wave_gemm8<MT, 1>(in, 8, p.wdelf, 8, KSN, 0, lane, acc);
        if (lane < 32) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const long gr = row0 + i * 32 + lane;
                if (gr < p.R) {
                    const float4v b = *reinterpret_cast<const float4v*>(p.boxes + gr * 4);
                    const float d0 = acc[i][0][0] + p.bdel[0], d1 = acc[i][0][1] + p.bdel[1], d2 = acc[i][0][2] + p.bdel[2],
                                d3 = acc[i][0][3] + p.bdel[3];
                    const float widths = b[2] - b[0], heights = b[3] - b[1];
                    const float ctr_x = b[0] + 0.5f * widths, ctr_y = b[1] + 0.5f * heights;
                    const float dx = d0 / p.wx, dy = d1 / p.wy;
                    const float dw = fminf(d2 / p.ww, p.clamp), dh = fminf(d3 / p.wh, p.clamp);
                    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
                    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
                    float4v o;
                    o[0] = pcx - 0.5f * pw;
                    o[1] = pcy - 0.5f * ph;
                    o[2] = pcx + 0.5f * pw;
                    o[3] = pcy + 0.5f * ph;
                    *reinterpret_cast<float4v*>(p.boxes_out + gr * 4) = o;
                    if (p.bad_flag && !(o[2] >= o[0] && o[3] >= o[1])) atomicOr(p.bad_flag, 1);
                }
            }
        }
    }
}

template <int MT>
int launch(const HeadTailParams& p, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int smem = 2 * BM * APITCH + BM * CPITCH * 4;
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set) && smem > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&head_tail_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        mark_on_device(attr_set);
    }
    hipLaunchKernelGGL(head_tail_kernel<MT>, dim3((unsigned)((p.R + BM - 1) / BM)), dim3(512), smem, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

bool dvid_head_tail_supported(int hidden, int dff, int num_cls, int num_reg, int num_classes) {
    return hidden == D && dff % D == 0 && dff >= D && num_cls >= 0 && num_cls <= 4 && num_reg >= 0 && num_reg <= 4 && num_classes >= 1 &&
           num_classes <= 64;
}

int dvid_head_tail_launch(const HeadTailParams& p, hipStream_t s) {
    if (p.R <= 0) return DVID_OK;
    if (!dvid_head_tail_supported(D, p.dff, p.num_cls, p.num_reg, p.num_classes)) return DVID_ERR_UNSUPPORTED;
    // 32-row tiles at every size -- which kernel (and tile) a row runs on must not depend on how many rows share its launch
    // (look-ahead invariance).  Measured (profiles/r03_lookahead1_ab.txt): one-batch calls 1393 frames/s against 1353 layer by layer;
    // 304-frame groups 2176 against 2192 (64-row tiles: 2180 / 1338).
    return launch<1>(p, s);
}
