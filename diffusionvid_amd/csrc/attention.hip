// Multi-head attention on MFMA (fp16 operands, fp32 softmax / accumulate), head_dim 32.
//
// Used for the 300x300 per-frame self-attention of RCNNHead (box_head.py:514-518), the 2400x900 global-memory
// cross-attention of DynamicHead (box_head.py:366-380) and Swin's 7x7 window attention (swintransformer.py:98-176); the
// in/out projections are igemm launches.  (The round-1 fp32 VALU kernel is in this file's history.)
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"

// =============================================================================================
// MFMA attention (fp16 operands, fp32 softmax/accumulate), head_dim 32.
//
// One wave = 16 queries, one workgroup = 4 waves = 64 queries of one (batch, head).  Per 32 keys:
//   S^T[key][q] = K[16 keys x 32 dims] . Q^T            two v_mfma_f32_16x16x32_f16 (swapped product, so a
//                                                        lane holds scores of ONE query: (lane & 15))
//   online softmax in registers: lane-local max/sum over its 8 keys, 2 xor-shuffles across the four
//   16-lane groups for the running max; the row sum is reduced across groups once at the end
//   O^T[d][q] += V^T[16 dims x 32 keys] . P^T           two MFMAs; P^T is built from the S^T accumulators
//                                                        in registers: MFMA k-slot (group g, j) <-> key
//                                                        (j < 4 ? 4g + j : 16 + 4g + j - 4), the same slot
//                                                        map is used for the V^T operand, so no LDS / no
//                                                        cross-lane traffic is needed for P.
// K rows and the pre-transposed V^T ([batch][head][32][lk_pad], attn_vt_kernel) are read straight from
// global memory (L1/L2 resident: 19 KB per (frame, head)).
// =============================================================================================
namespace {

constexpr int DH = 32;      // head dimension

// v16 [batch][lk][v_ld] (head h at column h*32) -> vt [batch][nheads][32][lk_pad] (zero padded keys)
__global__ void attn_vt_kernel(const half_t* __restrict__ v, half_t* __restrict__ vt, int lk, int lk_pad, int v_ld, long v_bs,
                               int nheads) {
    __shared__ half_t tile[32][33];
    const int k0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads
    for (int r = ty; r < 32; r += 8) {
        const int key = k0 + r;
        tile[r][tx] = key < lk ? v[b * v_bs + (long)key * v_ld + h * 32 + tx] : (half_t)0.f;
    }
    __syncthreads();
    half_t* dst = vt + ((long)(b * nheads + h) * 32) * lk_pad;
    for (int r = ty; r < 32; r += 8) {
        const int key = k0 + tx;
        if (key < lk_pad) dst[(long)r * lk_pad + key] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void mha_mfma_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                        const half_t* __restrict__ vt, half_t* __restrict__ out, int lq, int lk,
                                                        int lk_pad, int q_ld, int k_ld, int out_ld, long q_bs, long k_bs,
                                                        long out_bs, int nheads, float scaling) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 64 + wave * 16;
    if (q0 >= lq) return;                                          // whole wave out of range (no barriers below)
    const int qrow = min(q0 + qi, lq - 1);
    const half8 qf = *reinterpret_cast<const half8*>(q + b * q_bs + (long)qrow * q_ld + h * 32 + g * 8);   // B operand of S^T
    const half_t* kb = k + b * k_bs + h * 32 + g * 8;
    const half_t* vb = vt + ((long)(b * nheads + h) * 32) * lk_pad;

    float4v o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};   // O^T rows d = 4g + r (o0) and 16 + 4g + r (o1), column = query
    float m = -1e30f, l = 0.f;
    for (int k0 = 0; k0 < lk; k0 += 32) {
        // ---- scores for keys k0 .. k0+31 (two 16-key tiles) ----
        const int ka = min(k0 + qi, lk - 1), kbi = min(k0 + 16 + qi, lk - 1);       // A operand rows (clamped; masked below)
        const half8 kf0 = *reinterpret_cast<const half8*>(kb + (long)ka * k_ld);
        const half8 kf1 = *reinterpret_cast<const half8*>(kb + (long)kbi * k_ld);
        float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf, s0, 0, 0, 0);          // rows = keys k0 + 4g + r
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf, s1, 0, 0, 0);          // rows = keys k0 + 16 + 4g + r
        float sc[8];
        float cmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (k0 + 4 * g + r < lk) ? s0[r] * scaling : -1e30f;
            sc[4 + r] = (k0 + 16 + 4 * g + r < lk) ? s1[r] * scaling : -1e30f;
            cmax = fmaxf(cmax, fmaxf(sc[r], sc[4 + r]));
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float mn = fmaxf(m, cmax);
        const float f = __expf(m - mn);
        float psum = 0.f;
        half8 pf;                                                   // B operand of O^T: k-slot j <-> sc[j]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pj = (sc[j] > -1e29f) ? __expf(sc[j] - mn) : 0.f;
            psum += pj;
            pf[j] = (half_t)pj;
        }
        l = l * f + psum;
        m = mn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] *= f;
            o1[r] *= f;
        }
        // ---- V^T operand: row d = (lane & 15) (+16), k-slots = keys k0+4g..+3 and k0+16+4g..+3 ----
        const half_t* v0 = vb + (long)qi * lk_pad + k0 + 4 * g;
        const half_t* v1 = v0 + 16L * lk_pad;
        const half4 a00 = *reinterpret_cast<const half4*>(v0), a01 = *reinterpret_cast<const half4*>(v0 + 16);
        const half4 a10 = *reinterpret_cast<const half4*>(v1), a11 = *reinterpret_cast<const half4*>(v1 + 16);
        const half8 vf0 = {a00[0], a00[1], a00[2], a00[3], a01[0], a01[1], a01[2], a01[3]};
        const half8 vf1 = {a10[0], a10[1], a10[2], a10[3], a11[0], a11[1], a11[2], a11[3]};
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0, pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1, pf, o1, 0, 0, 0);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q0 + qi >= lq) return;
    const float inv = 1.f / l;
    half_t* op = out + b * out_bs + (long)(q0 + qi) * out_ld + h * 32 + 4 * g;
    const half4 w0 = {(half_t)(o0[0] * inv), (half_t)(o0[1] * inv), (half_t)(o0[2] * inv), (half_t)(o0[3] * inv)};
    const half4 w1 = {(half_t)(o1[0] * inv), (half_t)(o1[1] * inv), (half_t)(o1[2] * inv), (half_t)(o1[3] * inv)};
    *reinterpret_cast<half4*>(op) = w0;
    *reinterpret_cast<half4*>(op + 16) = w1;
}

}  // namespace

// q16/k16/v16: fp16, head h at columns [h*32, h*32+32) of each row; vt_scratch: >= batch*nheads*32*lk_pad halves,
// lk_pad = round_up(lk, 32) + 32.
int dvid_mha_mfma_launch(const half_t* q, const half_t* k, const half_t* v, half_t* out, half_t* vt_scratch, int batch, int lq,
                         int lk, int nheads, int q_ld, int kv_ld, int out_ld, long q_bs, long kv_bs, long out_bs, hipStream_t s) {
    if (lq == 0 || batch == 0) return DVID_OK;
    if (lk <= 0 || (q_ld & 7) || (kv_ld & 7) || (out_ld & 3)) return DVID_ERR_ARG;
    const int lk_pad = (lk + 31) / 32 * 32 + 32;
    hipLaunchKernelGGL(attn_vt_kernel, dim3(lk_pad / 32, nheads, batch), dim3(256), 0, s, v, vt_scratch, lk, lk_pad, kv_ld, kv_bs, nheads);
    LAUNCH_CHECK();
    const float scaling = 1.0f / sqrtf((float)DH);
    hipLaunchKernelGGL(mha_mfma_kernel, dim3(ceil_div(lq, 64), nheads, batch), dim3(256), 0, s, q, k, vt_scratch, out, lq, lk, lk_pad,
                       q_ld, kv_ld, out_ld, q_bs, kv_bs, out_bs, nheads, scaling);
    LAUNCH_CHECK();
    return DVID_OK;
}

// =============================================================================================
// Swin (shifted-)window attention, window 7x7 (49 tokens), head_dim 32, MFMA.
// Replaces WindowAttention.forward + the pad / roll / window_partition / window_reverse plumbing of
// SwinTransformerBlock.forward (mega_core/modeling/backbone/swintransformer.py:135-176, :236-270):
// the window <-> token mapping (padding to multiples of 7, cyclic shift by 3) is resolved in the gather
// addresses, padded positions behave as tokens whose q/k/v equal the qkv bias (the reference pads the
// normalised activations with zeros BEFORE the qkv Linear), the relative-position bias comes from a
// per-layer [heads][49][49] fp32 table, and the SW-MSA mask (-100 between different shift regions,
// swintransformer.py:387-406) is recomputed from the positions' region ids.
// One workgroup = one (window, head): K and V^T of the 49 keys (padded to 64) are staged in LDS,
// each of the 4 waves owns 16 queries; same swapped-product / register-resident P scheme as mha_mfma.
// =============================================================================================
namespace {

template <int WPB>
__global__ __launch_bounds__(256) void swin_window_attn_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ qkv_bias16,
                                                                const float* __restrict__ relbias, half_t* __restrict__ out, int H,
                                                                int W, int C, int nheads, int shift, float scaling, int nwin,
                                                                int head_major) {
    constexpr int WS = 7, NT = 49, RB_PITCH = SWIN_RELBIAS_PITCH;
    __shared__ __attribute__((aligned(16))) half_t Ks[64 * 32];      // [key][32 dims]
    __shared__ __attribute__((aligned(16))) half_t Vt[32 * 72];      // [dim][key], pitch 72 halves
    __shared__ int tok[64];                                          // token row or -1 (padded position)
    __shared__ int region[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Launch order.  The heads of a window read neighbouring 64-byte pieces of the same qkv rows (two heads per 128-byte line) and
    // write neighbouring pieces of the same output rows: with the head innermost AND the workgroups of an XCD taking one contiguous
    // run of (window, head) pairs, those pieces meet in that XCD's L2 while the line is there; window-major order per head (the
    // former 2-D grid) fetched every line once per head that touches it.  head_major = 0 keeps the old order for A/B runs.
    // A workgroup takes WPB consecutive windows of ONE head: a lane's relative-position bias depends on its query position and the
    // head only, so it is fetched once per workgroup instead of once per window (16 of the 36 load / store instructions a
    // (window, head) pair costs the CU's L1 path otherwise).
    const int ngrp = (nwin + WPB - 1) / WPB;
    int h, grp;
    if (head_major) {
        const int lid = igemm_xcd_remap((int)blockIdx.x, ngrp * nheads);
        grp = lid / nheads;
        h = lid - grp * nheads;
    } else {
        h = (int)blockIdx.x / ngrp;
        grp = (int)blockIdx.x - h * ngrp;
    }
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    const int nwx = Wp / WS, nwy = Hp / WS;
    const int qi = lane & 15, g = lane >> 4;
    const int qpos = wave * 16 + qi;                                   // query position inside the window (>= 49: idle)
    const int qp = min(qpos, NT - 1);
    const float* brow = relbias + ((long)h * NT + qp) * RB_PITCH + 4 * g;
    float4v bias4[4];                                                  // keys 4g.., 16 + 4g.., 32 + 4g.., 48 + 4g.. of the query's row
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = *reinterpret_cast<const float4v*>(brow + 16 * i);

    for (int w = 0; w < WPB; ++w) {
    int wid = grp * WPB + w;
    if (wid >= nwin) break;                                            // (uniform over the workgroup)
    if (w) __syncthreads();                                            // the previous window's LDS reads are done
    const int wx = wid % nwx;
    wid /= nwx;
    const int wy = wid % nwy;
    const int b = wid / nwy;

    if (tid < 64) {
        int t = -1, reg = 0;
        if (tid < NT) {
            const int py = tid / WS, px = tid - py * WS;
            const int ys = wy * WS + py, xs = wx * WS + px;                  // coordinates in the shifted, padded map
            int y = ys + shift, x = xs + shift;                              // source coordinates before the roll
            if (y >= Hp) y -= Hp;
            if (x >= Wp) x -= Wp;
            if (y < H && x < W) t = (b * H + y) * W + x;
            if (shift > 0) {
                const int hr = ys < Hp - WS ? 0 : (ys < Hp - shift ? 1 : 2);
                const int wr = xs < Wp - WS ? 0 : (xs < Wp - shift ? 1 : 2);
                reg = hr * 3 + wr;
            }
        }
        tok[tid] = t;
        region[tid] = reg;
    }
    __syncthreads();
    // ---- every global read of the window goes out here, ahead of the LDS staging and its barrier: K / V of the thread's
    // (key, 16-byte chunk) and the lane's query fragment (its bias, 4 x 16 bytes of the query's 64-float row, is loaded above)
    const int tq = tok[qp];
    const int qreg = region[qp];
    const int key_s = tid >> 2, ch = tid & 3;
    half8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (key_s < NT) {
        const int t = tok[key_s];
        const half_t* src = t >= 0 ? qkv + (long)t * 3 * C : qkv_bias16;
        kv = *reinterpret_cast<const half8*>(src + C + h * 32 + ch * 8);
        vv = *reinterpret_cast<const half8*>(src + 2 * C + h * 32 + ch * 8);
    }
    const half_t* qsrc = tq >= 0 ? qkv + (long)tq * 3 * C : qkv_bias16;
    const half8 qf = *reinterpret_cast<const half8*>(qsrc + h * 32 + g * 8);
    // ---- stage K rows and V^T ----
    *reinterpret_cast<half8*>(Ks + key_s * 32 + ch * 8) = kv;
#pragma unroll
    for (int e = 0; e < 8; ++e) Vt[(ch * 8 + e) * 72 + key_s] = vv[e];
    __syncthreads();

    float4v o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < 64; k0 += 32) {
        const half8 kf0 = *reinterpret_cast<const half8*>(Ks + (k0 + qi) * 32 + g * 8);
        const half8 kf1 = *reinterpret_cast<const half8*>(Ks + (k0 + 16 + qi) * 32 + g * 8);
        float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf, s1, 0, 0, 0);
        float sc[8];
        float cmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int key = k0 + (r < 4 ? 4 * g + r : 16 + 4 * g + (r - 4));
            float v = -1e30f;
            if (key < NT) {
                v = (r < 4 ? s0[r] * scaling + bias4[k0 / 16][r] : s1[r - 4] * scaling + bias4[k0 / 16 + 1][r - 4]);
                if (shift > 0 && region[key] != qreg) v += -100.0f;
            }
            sc[r] = v;
            cmax = fmaxf(cmax, v);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float mn = fmaxf(m, cmax);
        const float f = __expf(m - mn);
        float psum = 0.f;
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pj = (sc[j] > -1e29f) ? __expf(sc[j] - mn) : 0.f;
            psum += pj;
            pf[j] = (half_t)pj;
        }
        l = l * f + psum;
        m = mn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] *= f;
            o1[r] *= f;
        }
        const half_t* v0 = Vt + qi * 72 + k0 + 4 * g;
        const half_t* v1 = v0 + 16 * 72;
        const half4 a00 = *reinterpret_cast<const half4*>(v0), a01 = *reinterpret_cast<const half4*>(v0 + 16);
        const half4 a10 = *reinterpret_cast<const half4*>(v1), a11 = *reinterpret_cast<const half4*>(v1 + 16);
        const half8 vf0 = {a00[0], a00[1], a00[2], a00[3], a01[0], a01[1], a01[2], a01[3]};
        const half8 vf1 = {a10[0], a10[1], a10[2], a10[3], a11[0], a11[1], a11[2], a11[3]};
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0, pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1, pf, o1, 0, 0, 0);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qpos < NT && tq >= 0) {                                        // padded positions produce no output
        const float inv = 1.f / l;
        half_t* op = out + (long)tq * C + h * 32 + 4 * g;
        const half4 w0 = {(half_t)(o0[0] * inv), (half_t)(o0[1] * inv), (half_t)(o0[2] * inv), (half_t)(o0[3] * inv)};
        const half4 w1 = {(half_t)(o1[0] * inv), (half_t)(o1[1] * inv), (half_t)(o1[2] * inv), (half_t)(o1[3] * inv)};
        *reinterpret_cast<half4*>(op) = w0;
        *reinterpret_cast<half4*>(op + 16) = w1;
    }
    }      // windows of the workgroup
}

}  // namespace

// qkv fp16 [B*H*W, 3C] (q | k | v, head h at columns 32h..), relbias fp32 [nheads][49][SWIN_RELBIAS_PITCH] (rows padded to 64 keys), out fp16 [B*H*W, C]
int dvid_swin_window_attn_launch(const half_t* qkv, const half_t* qkv_bias16, const float* relbias, half_t* out, int batch, int H,
                                 int W, int C, int nheads, int shift, hipStream_t s) {
    if (C != nheads * 32) return DVID_ERR_UNSUPPORTED;
    const int nwy = (H + 6) / 7, nwx = (W + 6) / 7;
    const long nwin = (long)batch * nwy * nwx;
    if (nwin * nheads > 0x7fffffffL) return DVID_ERR_UNSUPPORTED;
    // four windows per workgroup, head-major order (one window per workgroup and window-major order were measured slower: profiles/r02e_*)
    hipLaunchKernelGGL(swin_window_attn_kernel<4>, dim3((unsigned)((nwin + 3) / 4 * nheads)), dim3(256), 0, s, qkv, qkv_bias16, relbias, out,
                       H, W, C, nheads, shift, 1.0f / sqrtf(32.f), (int)nwin, 1);
    LAUNCH_CHECK();
    return DVID_OK;
}
