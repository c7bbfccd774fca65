// The DTYPE float32 path: fp32 storage end to end and exact fp32 products.
//
// The reference runs fp32 unless `DTYPE float16` is given (mega_core/config/defaults.py:582; tools/test_net.py:97-98 switches apex amp
// on for float16 only).  The fp16 path of this library (igemm2 / conv3x3 / wstat / bneck / headtail / dynconv / attention) rounds every
// stored activation and every weight to fp16 -- the apex O1 policy -- which is where its distance from an fp32 evaluation comes from
// (profiles/r05_logit_error_stages.txt).  The kernels here keep every tensor in fp32 and multiply on v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, bit-for-bit a k-ordered fmaf chain (MI355X_MICROARCH.md), 157 TFLOP/s dense peak = 1/16 of the fp16 MFMA rate.
// They are deliberately plain -- one implicit-GEMM kernel for every convolution and linear layer, VALU kernels for the reductions --
// because the mode exists for conformance (the fp32 CPU oracle's results to ~1e-5), not for the headline rate.
//
//   f32_igemm_kernel      conv / linear: NHWC fp32 in, [Cout][Kpad] fp32 weights, + bias, + residual (same shape or FPN nearest-x2
//                         top-down), ReLU / exact GELU, fp32 out; 128 x BN x 16 tiles through LDS, 2 x 2 waves
//   f32_roialign_kernel   detectron2 ROIPooler(ROIAlignV2) as called at box_head.py:507/:617 on fp32 pyramids (csrc/roialign.hip's walk)
//   f32_mha_kernel        nn.MultiheadAttention's softmax(q k^T / sqrt(32)) v per head, one query per lane, keys through LDS
//   f32_dynconv_kernel    DynamicConv.forward (box_head.py:687-711): two per-box products + LayerNorm + ReLU, one workgroup per box
//   elementwise           image normaliser -> NHWC4, 3x3/2 max pool, SiLU, scale / shift modulation
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "options.h"

namespace {

__device__ __forceinline__ float16v mfma_f32(float a, float b, float16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------------------------
// implicit GEMM.  D[m][n] = sum_k A[m][k] W[n][k]; A is the im2col view of the NHWC input (k = (ky * KW + kx) * Cin + c, Cin % 4 == 0),
// W rows are zero-padded to Kpad (a multiple of 16).  Workgroup = 4 waves as 2 x 2, wave tile 64 x (BN / 2), K step 16:
// global -> registers -> LDS (rows of 16 floats at a pitch of 20: the ds_read_b128 fragment reads are conflict-free), one barrier
// per step, the next step's global loads in flight under this step's 32 (BN 128) MFMAs of 64 cycles each.
// MFMA operand maps (cdna_hip_programming.md): A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31]; a lane
// reads 4 consecutive k of its row as one float4, so the MFMA of element e multiplies k = 8 j + e (lanes 0-31) and 8 j + 4 + e.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int F32_BM = 128, F32_BK = 16, F32_LDT = 20;

// Epilogue shared by the two implicit-GEMM kernels below.  Register r of an accumulator block = row (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3),
// column lane & 31.  out = act((acc * wscale[n]) + bias[n] + residual): wscale undoes the power-of-two row scaling of the packed weights
// (exact; null = none).
// Row-coalesced form (N and ldc multiples of 4): the tile crosses LDS in two halves of 64 rows (the operand buffers are free by now),
// and every lane finishes 4 consecutive channels of a row -- 16-byte residual reads and stores, 512 contiguous bytes per 32 lanes --
// instead of 4-byte accesses 2 rows x 128 bytes per instruction (the short-K layers, res3 / res4 conv3 + residual, ran at 0.39 of
// the MFMA rate on their stores: profiles/r06c_bench.json).  General form (N tails: class_logits, bboxes_delta): straight from the
// accumulator layout.  The caller has passed a barrier behind its last LDS read.
template <int BN>
__device__ __forceinline__ void f32_epilogue(const F32GemmParams& p, float16v (&acc)[2][BN / 64], float* Cs, int m0, int n0, int wm, int wn, int lane,
                                             int tid) {
    constexpr int NB = BN / 64;
    const int fr = lane & 31;
    if (((p.Cout | p.ldc) & 3) == 0) {
        constexpr int CP = BN + 4;          // pitch of an LDS row (floats)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            Cs[(mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * CP + wn * (BN / 2) + nb * 32 + fr] = acc[mb][nb][r];
            }
            __syncthreads();
            constexpr int VPR = BN / 4;          // float4 per row
            for (int idx = tid; idx < 64 * VPR; idx += 256) {
                const int row = idx / VPR, c4 = (idx - row * VPR) * 4;
                const int m = m0 + half * 64 + row, n = n0 + c4;
                if (m >= p.M || n >= p.Cout) continue;
                float4v v = *reinterpret_cast<const float4v*>(&Cs[row * CP + c4]);
                if (p.wscale) v *= *reinterpret_cast<const float4v*>(p.wscale + n);
                if (p.bias) v += *reinterpret_cast<const float4v*>(p.bias + n);
                if (p.res_mode == 1) {
                    v += *reinterpret_cast<const float4v*>(p.res + (long)m * p.Cout + n);
                } else if (p.res_mode == 2) {
                    const int ox = m % p.Wo;
                    const int t2 = m / p.Wo;
                    const int oy = t2 % p.Ho;
                    const int img = t2 / p.Ho;
                    v += *reinterpret_cast<const float4v*>(p.res + ((long)(img * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + n);
                }
                if (p.relu == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.relu == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                *reinterpret_cast<float4v*>(p.out + (long)m * p.ldc + n) = v;
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + wn * (BN / 2) + nb * 32 + fr;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float ws = p.wscale ? p.wscale[n] : 1.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                if (m >= p.M) continue;
                float v = acc[mb][nb][r] * ws + bias;
                if (p.res_mode == 1) {
                    v += p.res[(long)m * p.Cout + n];
                } else if (p.res_mode == 2) {
                    const int ox = m % p.Wo;
                    const int t2 = m / p.Wo;
                    const int oy = t2 % p.Ho;
                    const int img = t2 / p.Ho;
                    v += p.res[((long)(img * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + n];
                }
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = gelu_erf(v);
                p.out[(long)m * p.ldc + n] = v;
            }
    }
}

// (four waves per SIMD: with 128 registers per wave the accumulators stay in VGPRs and four 40-KB workgroups fill a CU's LDS exactly; 1-10 %
// faster than three per SIMD on every layer shape, profiles/r06k_f32_gemm_w4.txt)
template <int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void f32_igemm_kernel(F32GemmParams p) {
    constexpr int NB = BN / 64;          // 32-column blocks per wave
    __shared__ float Smem[2 * F32_BM * F32_LDT + 2 * BN * F32_LDT];          // A stages | B stages; the epilogue's half tile afterwards
    float (*As)[F32_BM * F32_LDT] = reinterpret_cast<float (*)[F32_BM * F32_LDT]>(Smem);
    float (*Bs)[BN * F32_LDT] = reinterpret_cast<float (*)[BN * F32_LDT]>(Smem + 2 * F32_BM * F32_LDT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int t = igemm_xcd_remap((int)blockIdx.x, p.tiles_m * p.tiles_n);          // an XCD owns a contiguous run of row tiles
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * F32_BM, n0 = tn * BN;
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    const bool pointwise = p.KH == 1 && p.KW == 1 && p.pad == 0;

    // this thread's two A rows
    long abase[2];
    int ay[2], ax[2];
    bool aok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + lr + 64 * i;
        aok[i] = m < p.M;
        const int mm = aok[i] ? m : 0;
        const int ox = mm % p.Wo;
        const int t2 = mm / p.Wo;
        const int oy = t2 % p.Ho;
        const int img = t2 / p.Ho;
        ay[i] = oy * p.stride - p.pad;
        ax[i] = ox * p.stride - p.pad;
        abase[i] = (long)img * p.H * p.W;
    }
    const float* wrow[NB];
    bool wok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + lr + 64 * i;
        wok[i] = n < p.Cout;
        wrow[i] = p.w + (long)(wok[i] ? n : 0) * p.Kpad + kq;
    }

    float4v ra[2], rb[NB];
    auto fetch = [&](int kt) {
        const int k = kt * F32_BK + kq;
        int c = k, ky = 0, kx = 0;
        if (!pointwise) {
            const int tap = k / p.Cin;
            c = k - tap * p.Cin;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy = ay[i] + ky, ix = ax[i] + kx;
            const bool ok = aok[i] && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            ra[i] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (ok) ra[i] = *reinterpret_cast<const float4v*>(p.in + (abase[i] + (long)iy * p.W + ix) * p.Cin + c);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (wok[i]) rb[i] = *reinterpret_cast<const float4v*>(wrow[i] + kt * F32_BK);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4v*>(&As[buf][(lr + 64 * i) * F32_LDT + kq]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4v*>(&Bs[buf][(lr + 64 * i) * F32_LDT + kq]) = rb[i];
    };

    float16v acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = p.Kpad / F32_BK;
    fetch(0);
    stage(0);
    __syncthreads();
    const int fr = lane & 31, fk = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch(kt + 1);
        float4v a[2][2], b[NB][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < 2; ++j) a[mb][j] = *reinterpret_cast<const float4v*>(&As[buf][(wm * 64 + mb * 32 + fr) * F32_LDT + j * 8 + fk]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[nb][j] = *reinterpret_cast<const float4v*>(&Bs[buf][(wn * (BN / 2) + nb * 32 + fr) * F32_LDT + j * 8 + fk]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma_f32(a[mb][j][e], b[nb][j][e], acc[mb][nb]);
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }

    f32_epilogue<BN>(p, acc, Smem, m0, n0, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same implicit GEMM with SPLIT operands: every fp32 operand value v is staged to LDS as two fp16 numbers hi = fp16(v),
// lo = fp16(v - hi), and a product row is accumulated as  lo_a * hi_b + hi_a * lo_b + hi_a * hi_b  on v_mfma_f32_32x32x16_f16 (exact
// fp16 x fp16 products, fp32 accumulation): three passes of the fp16 MFMA, 16 / 3 = 5.3 x the fp32 MFMA's rate.  What is dropped is
// lo_a * lo_b (2^-22 of the product) and the rounding of lo (2^-22 relative while lo is a normal fp16 number, i.e. |v| >= 2^-3; an absolute
// 3e-8 below that).  The packed weight rows are scaled by a power of two so that each row's largest magnitude lies in [0.5, 1) (csrc/model.hip:
// make_conv; undone exactly by `wscale` in the epilogue) -- a weight of 0.02 would otherwise carry its lo part as an fp16 subnormal.
// Activations need |v| < 65504 (true of every tensor on this path by orders of magnitude).  Library option f32_split (default 1) selects
// it; 0 = the exact-fp32 kernel above.  Both are held to the same bounds by tests/test_gpu_f32.py and the end-to-end float32 tests.
// Tile 128 x BN x 32, four waves as 2 x 2; LDS holds A_hi | A_lo | B_hi | B_lo as [row][32 halves] at a pitch of 40 halves (80 bytes: the
// ds_read_b128 fragment reads are conflict-free), single-buffered: the next step's global loads are in flight under this step's 24 MFMAs,
// converted and written between two barriers.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int X3_BM = 128, X3_BK = 32, X3_PITCH = 40;
// (BN 128 at three waves per SIMD -- 162 registers, no spills -- is 2-30 % faster than the compiler's two: profiles/r06o_x3_w3.txt)
template <int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 3 : 4, BN == 128 ? 3 : 4))) void f32x3_igemm_kernel(F32GemmParams p) {
    constexpr int NB = BN / 64;
    constexpr int AP = X3_BM / 32, BP = BN / 32;          // loader passes (32 tile rows each: 8 lanes x float4 per row)
    constexpr int ROWS = 2 * X3_BM + 2 * BN;
    static_assert(64 * (BN + 4) * 4 <= ROWS * X3_PITCH * 2, "the epilogue's half tile fits the operand buffers");
    __shared__ __attribute__((aligned(16))) half_t Sm[ROWS * X3_PITCH];
    half_t* const Ahi = Sm;
    half_t* const Alo = Sm + X3_BM * X3_PITCH;
    half_t* const Bhi = Sm + 2 * X3_BM * X3_PITCH;
    half_t* const Blo = Bhi + BN * X3_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int t = igemm_xcd_remap((int)blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * X3_BM, n0 = tn * BN;
    const int lr = tid >> 3, kq = (tid & 7) * 4;
    const bool pointwise = p.KH == 1 && p.KW == 1 && p.pad == 0;

    long abase[AP];
    int ay[AP], ax[AP];
    bool aok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lr + 32 * i;
        aok[i] = m < p.M;
        const int mm = aok[i] ? m : 0;
        const int ox = mm % p.Wo;
        const int t2 = mm / p.Wo;
        const int oy = t2 % p.Ho;
        const int img = t2 / p.Ho;
        ay[i] = oy * p.stride - p.pad;
        ax[i] = ox * p.stride - p.pad;
        abase[i] = (long)img * p.H * p.W;
    }
    // the weights arrive already split (p.w_hi / p.w_lo: the packed, row-scaled fp32 rows as fp16 (hi, lo) planes, made once at load):
    // their half of the tile needs no conversion
    long wrow[BP];
    bool wok[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + lr + 32 * i;
        wok[i] = n < p.Cout;
        wrow[i] = (long)(wok[i] ? n : 0) * p.Kpad + kq;
    }

    // unconditional loads (an out-of-range lane reads a valid dummy address; its value becomes zero when staged)
    float4v ra[AP];
    half4 rbh[BP], rbl[BP];
    bool rok[AP], rwk[BP];
    auto fetch = [&](int kt) {
        const int k = kt * X3_BK + kq;
        int c = k, ky = 0, kx = 0;
        if (!pointwise) {
            const int tap = k / p.Cin;
            c = k - tap * p.Cin;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int iy = ay[i] + ky, ix = ax[i] + kx;
            rok[i] = aok[i] && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            ra[i] = *reinterpret_cast<const float4v*>(rok[i] ? p.in + (abase[i] + (long)iy * p.W + ix) * p.Cin + c : p.in);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            rwk[i] = wok[i] && k < p.Kpad;
            const long off = rwk[i] ? wrow[i] + kt * X3_BK : 0;
            rbh[i] = *reinterpret_cast<const half4*>(p.w_hi + off);
            rbl[i] = *reinterpret_cast<const half4*>(p.w_lo + off);
        }
    };
    auto split_store = [&](float4v v, bool ok, half_t* hi_row, half_t* lo_row) {
        if (!ok) v = (float4v){0.f, 0.f, 0.f, 0.f};
        // an activation beyond the fp16 range would become inf here where fp32 arithmetic would not: reported, never silent (the model
        // checks the flag at the batch's host synchronisation and raises; f32_split = 0 has no such limit)
        if (p.range_flag && fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))) > 65504.f)
            atomicOr(p.range_flag, 1);
        const half4 h = __builtin_convertvector(v, half4);                     // round to nearest even
        const float4v back = __builtin_convertvector(h, float4v);
        const half4 l = __builtin_convertvector(v - back, half4);              // v - back is exact in fp32
        *reinterpret_cast<half4*>(hi_row) = h;
        *reinterpret_cast<half4*>(lo_row) = l;
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i) split_store(ra[i], rok[i], Ahi + (lr + 32 * i) * X3_PITCH + kq, Alo + (lr + 32 * i) * X3_PITCH + kq);
        const half4 hz = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            *reinterpret_cast<half4*>(Bhi + (lr + 32 * i) * X3_PITCH + kq) = rwk[i] ? rbh[i] : hz;
            *reinterpret_cast<half4*>(Blo + (lr + 32 * i) * X3_PITCH + kq) = rwk[i] ? rbl[i] : hz;
        }
    };

    float16v acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.Kpad + X3_BK - 1) / X3_BK;
    const int fr = lane & 31, fk = (lane >> 5) * 8;          // MFMA operand maps: lane l = row l & 31, k = 8 (l >> 5) .. + 8 of a 16-deep K step
    fetch(0);
    stage();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < X3_BK / 16; ++ks) {
            half8 ah[2], al[2], bh[NB], bl[NB];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int off = (wm * 64 + mb * 32 + fr) * X3_PITCH + ks * 16 + fk;
                ah[mb] = *reinterpret_cast<const half8*>(Ahi + off);
                al[mb] = *reinterpret_cast<const half8*>(Alo + off);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int off = (wn * (BN / 2) + nb * 32 + fr) * X3_PITCH + ks * 16 + fk;
                bh[nb] = *reinterpret_cast<const half8*>(Bhi + off);
                bl[nb] = *reinterpret_cast<const half8*>(Blo + off);
            }
            // the two small terms first, then the leading one (the accumulator is fp32 either way; the order is fixed)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();                       // every wave has read this step's fragments
        if (kt + 1 < nk) {
            stage();
            __syncthreads();
        }
    }
    f32_epilogue<BN>(p, acc, reinterpret_cast<float*>(Sm), m0, n0, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------------------------------------------
// fp32 NCHW frames in [0, 1] (a table of per-frame pointers, as csrc/elementwise.hip) -> normalised fp32 NHWC4 (channel 3 zero).
// (x - mean) / std as the reference's normalizer divides (diffusion_det.py:301-303).
__global__ void f32_prep_images_kernel(FrameTable in, float* __restrict__ out, long npix, long hw, float m0, float m1, float m2, float s0,
                                       float s1, float s2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const long img = i / hw, pix = i - img * hw;
    const float* q = in.p[img] + pix;
    *reinterpret_cast<float4v*>(out + i * 4) = (float4v){(q[0] - m0) / s0, (q[hw] - m1) / s1, (q[2 * hw] - m2) / s2, 0.f};
}

__global__ void f32_maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w, int c, int ho, int wo) {
    const int cv = c >> 2;
    const long total = (long)n * ho * wo * cv;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int v = i % cv;
    long t = i / cv;
    const int ox = t % wo;
    t /= wo;
    const int oy = t % ho;
    const int img = t / ho;
    float4v best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if ((unsigned)ix >= (unsigned)w) continue;
            const float4v x = *reinterpret_cast<const float4v*>(in + (((long)img * h + iy) * w + ix) * c + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], x[e]);
        }
    }
    *reinterpret_cast<float4v*>(out + i * 4) = best;
}

__global__ void f32_silu_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    float4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] / (1.f + expf(-v[e]));
    *reinterpret_cast<float4v*>(y + i * 4) = o;
}

// box_head.py:533-536 / :643-647: fc = x * (scale + 1) + shift
__global__ void f32_modulate_kernel(const float* __restrict__ x, const float* __restrict__ scale, int scale_ld, const float* __restrict__ shift,
                                    int shift_per_row, int shift_ld, float* __restrict__ y, long n4, int rows_per_frame, int d) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int dv = d >> 2;
    const long row = i / dv;
    const int col = (int)(i - row * dv) * 4;
    const long frame = row / rows_per_frame;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    const float4v sc = *reinterpret_cast<const float4v*>(scale + frame * scale_ld + col);
    const float4v sh = *reinterpret_cast<const float4v*>(shift + (shift_per_row ? row : frame) * shift_ld + col);
    float4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * (sc[e] + 1.f) + sh[e];
    *reinterpret_cast<float4v*>(y + i * 4) = o;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// RoIAlignV2 (aligned, 7 x 7 bins, 2 x 2 samples), fp32 pyramids: csrc/roialign.hip's kernel with fp32 taps -- one workgroup per box,
// 32 lanes x 8 channels cover a tap's 256 channels, the 8 lane groups walk the 49 bins
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int RP = 7, RG = 2;

struct Tap32 {
    int lo, hi;
    float wl, wh;
    bool ok;
};
__device__ __forceinline__ Tap32 axis_tap32(float y, int limit) {          // torchvision bilinear_interpolate, one axis
    Tap32 t;
    t.ok = !(y < -1.0f || y > (float)limit);
    if (y <= 0.f) y = 0.f;
    int lo = (int)y;
    int hi;
    if (lo >= limit - 1) {
        hi = lo = limit - 1;
        y = (float)lo;
    } else {
        hi = lo + 1;
    }
    const float l = y - (float)lo;
    t.lo = lo;
    t.hi = hi;
    t.wl = 1.f - l;
    t.wh = l;
    return t;
}

__global__ __launch_bounds__(256) void f32_roialign_kernel(RoiLevels32 lv, const float* __restrict__ boxes, int boxes_per_img,
                                                            float* __restrict__ roi_out, float* __restrict__ mean_out, int nbox) {
#pragma clang fp contract(off)
    __shared__ float red[8][256];
    const int box = igemm_xcd_remap((int)blockIdx.x, nbox);
    const int img = box / boxes_per_img;
    const int tid = threadIdx.x;
    const int grp = tid >> 5, ln = tid & 31;
    const float bx1 = boxes[box * 4 + 0], by1 = boxes[box * 4 + 1], bx2 = boxes[box * 4 + 2], by2 = boxes[box * 4 + 3];
    const float area = (bx2 - bx1) * (by2 - by1);          // detectron2 assign_boxes_to_levels (canonical 224 / level 4, levels 3..5)
    const bool valid_box = area >= 0.f;
    float lvf = floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 3.f), 5.f);
    const int level = valid_box ? (int)lvf - 3 : 0;
    const int H = lv.h[level], W = lv.w[level];
    const float sc = lv.scale[level];
    const float* feat = lv.feat[level] + (long)img * H * W * 256;
    const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f;
    const float x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
    const float bin_w = (x2 - x1) / RP, bin_h = (y2 - y1) / RP;
    float macc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) macc[e] = 0.f;
    for (int pb = grp; pb < RP * RP; pb += 8) {
        const int ph = pb / RP, pw = pb - ph * RP;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (valid_box) {
#pragma unroll
            for (int iy = 0; iy < RG; ++iy) {
                const float y = y1 + ph * bin_h + (iy + 0.5f) * bin_h / RG;
                const Tap32 ty = axis_tap32(y, H);
#pragma unroll
                for (int ix = 0; ix < RG; ++ix) {
                    const float x = x1 + pw * bin_w + (ix + 0.5f) * bin_w / RG;
                    const Tap32 tx = axis_tap32(x, W);
                    if (!(ty.ok && tx.ok)) continue;
                    const float w4[4] = {ty.wl * tx.wl, ty.wl * tx.wh, ty.wh * tx.wl, ty.wh * tx.wh};
                    const long o4[4] = {(long)ty.lo * W + tx.lo, (long)ty.lo * W + tx.hi, (long)ty.hi * W + tx.lo, (long)ty.hi * W + tx.hi};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4v a = *reinterpret_cast<const float4v*>(feat + o4[q] * 256 + ln * 8);
                        const float4v b = *reinterpret_cast<const float4v*>(feat + o4[q] * 256 + ln * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[e] += w4[q] * a[e];
                            acc[4 + e] += w4[q] * b[e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] *= 1.f / (RG * RG);
            macc[e] += acc[e];
        }
        float* o = roi_out + ((long)box * (RP * RP) + pb) * 256 + ln * 8;
        *reinterpret_cast<float4v*>(o) = (float4v){acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<float4v*>(o + 4) = (float4v){acc[4], acc[5], acc[6], acc[7]};
    }
    if (mean_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[grp][ln * 8 + e] = macc[e];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][tid];
        mean_out[(long)box * 256 + tid] = s / (RP * RP);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// attention, head dim 32: out[b][q][h*32 ..] = softmax_k(q . k / sqrt(32)) v.  One query per lane (its 32 q values and 32 output
// accumulators in registers), keys and values in chunks of 64 through LDS (every lane reads the same key: broadcast reads), the
// running maximum / sum of the streaming softmax updated once per 16 keys.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void f32_mha_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                     float* __restrict__ out, int lq, int lk, int q_ld, int kv_ld, int out_ld, long q_bs,
                                                     long kv_bs, long out_bs, float scale) {
    __shared__ float Ks[64 * 32];
    __shared__ float Vs[64 * 32];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 64 + tid;
    const bool live = qi < lq;
    float qv[32], acc[32];
    {
        const float* qp = q + b * q_bs + (long)(live ? qi : 0) * q_ld + head * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4v t = *reinterpret_cast<const float4v*>(qp + j * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[j * 4 + e] = t[e] * scale;
        }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    float mx = -INFINITY, den = 0.f;
    const float* kb = k + b * kv_bs + head * 32;
    const float* vb = v + b * kv_bs + head * 32;
    for (int k0 = 0; k0 < lk; k0 += 64) {
        const int nk = lk - k0 < 64 ? lk - k0 : 64;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // 64 keys x 8 float4: lane -> (key = (i * 64 + tid) / 8, quarter = .. % 8)
            const int idx = i * 64 + tid, kr = idx >> 3, c4 = (idx & 7) * 4;
            float4v kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kr < nk) {
                kk = *reinterpret_cast<const float4v*>(kb + (long)(k0 + kr) * kv_ld + c4);
                vv = *reinterpret_cast<const float4v*>(vb + (long)(k0 + kr) * kv_ld + c4);
            }
            *reinterpret_cast<float4v*>(&Ks[kr * 32 + c4]) = kk;
            *reinterpret_cast<float4v*>(&Vs[kr * 32 + c4]) = vv;
        }
        __syncthreads();
        // sub-chunks of 16 keys: scores into registers, one maximum / rescale per sub-chunk, then the weighted values
#pragma unroll 1
        for (int c0 = 0; c0 < nk; c0 += 16) {
            float sc16[16];
            float cmx = -INFINITY;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4v t = *reinterpret_cast<const float4v*>(&Ks[(c0 + kk) * 32 + j * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d = __builtin_fmaf(qv[j * 4 + e], t[e], d);
                }
                sc16[kk] = c0 + kk < nk ? d : -INFINITY;
                cmx = fmaxf(cmx, sc16[kk]);
            }
            const float nmx = fmaxf(mx, cmx);
            const float rescale = expf(mx - nmx);          // first sub-chunk: exp(-inf) = 0 on zero accumulators
            den *= rescale;
#pragma unroll
            for (int e = 0; e < 32; ++e) acc[e] *= rescale;
            mx = nmx;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float pr = expf(sc16[kk] - mx);          // masked keys: exp(-inf) = 0
                den += pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4v t = *reinterpret_cast<const float4v*>(&Vs[(c0 + kk) * 32 + j * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j * 4 + e] = __builtin_fmaf(pr, t[e], acc[j * 4 + e]);
                }
            }
        }
    }
    if (live) {
        const float inv = 1.f / den;
        float* op = out + b * out_bs + (long)qi * out_ld + head * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<float4v*>(op + j * 4) = (float4v){acc[j * 4] * inv, acc[j * 4 + 1] * inv, acc[j * 4 + 2] * inv, acc[j * 4 + 3] * inv};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Swin window attention (swintransformer.py:68-176, :226-254), fp32: one workgroup (one wave) per (window, head), lane = query position of
// the 7 x 7 window, keys and values through LDS.  A padded window position holds the qkv BIAS (the reference pads the normalised tokens
// with zeros before the qkv Linear); the shifted map's region mask adds -100 between positions of different regions.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void f32_swin_window_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                  const float* __restrict__ relbias, float* __restrict__ out, int H, int W, int C,
                                                                  int nheads, int shift, float scaling, int nwin) {
    constexpr int WS = 7, NT = 49;
    __shared__ float Ks[NT * 32];
    __shared__ float Vs[NT * 32];
    __shared__ int tok[64];
    __shared__ int region[64];
    const int tid = threadIdx.x;
    const int lid = igemm_xcd_remap((int)blockIdx.x, nwin * nheads);
    int wid = lid / nheads;
    const int h = lid - wid * nheads;
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    const int nwx = Wp / WS, nwy = Hp / WS;
    const int wx = wid % nwx;
    wid /= nwx;
    const int wy = wid % nwy;
    const int b = wid / nwy;
    {
        int t = -1, reg = 0;
        if (tid < NT) {
            const int py = tid / WS, px = tid - py * WS;
            const int ys = wy * WS + py, xs = wx * WS + px;          // coordinates in the shifted, padded map
            int y = ys + shift, x = xs + shift;                      // source coordinates before the roll
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
    for (int idx = tid; idx < NT * 8; idx += 64) {
        const int key = idx >> 3, c4 = (idx & 7) * 4;
        const int t = tok[key];
        const float* src = t >= 0 ? qkv + (long)t * 3 * C : qkv_bias;
        *reinterpret_cast<float4v*>(&Ks[key * 32 + c4]) = *reinterpret_cast<const float4v*>(src + C + h * 32 + c4);
        *reinterpret_cast<float4v*>(&Vs[key * 32 + c4]) = *reinterpret_cast<const float4v*>(src + 2 * C + h * 32 + c4);
    }
    const int qp = tid < NT ? tid : NT - 1;
    const int tq = tok[qp], qreg = region[qp];
    float qv[32];
    {
        const float* qsrc = (tq >= 0 ? qkv + (long)tq * 3 * C : qkv_bias) + h * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4v t = *reinterpret_cast<const float4v*>(qsrc + j * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[j * 4 + e] = t[e] * scaling;
        }
    }
    __syncthreads();
    const float* brow = relbias + ((long)h * NT + qp) * SWIN_RELBIAS_PITCH;
    float sc[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int key = 0; key < NT; ++key) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4v t = *reinterpret_cast<const float4v*>(&Ks[key * 32 + j * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) d = __builtin_fmaf(qv[j * 4 + e], t[e], d);
        }
        d += brow[key];
        if (shift > 0 && region[key] != qreg) d += -100.0f;
        sc[key] = d;
        mx = fmaxf(mx, d);
    }
    float acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    float den = 0.f;
#pragma unroll
    for (int key = 0; key < NT; ++key) {
        const float pr = expf(sc[key] - mx);
        den += pr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4v t = *reinterpret_cast<const float4v*>(&Vs[key * 32 + j * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j * 4 + e] = __builtin_fmaf(pr, t[e], acc[j * 4 + e]);
        }
    }
    if (tid < NT && tq >= 0) {          // padded positions produce no output
        const float inv = 1.f / den;
        float* op = out + (long)tq * C + h * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<float4v*>(op + j * 4) = (float4v){acc[j * 4] * inv, acc[j * 4 + 1] * inv, acc[j * 4 + 2] * inv, acc[j * 4 + 3] * inv};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// DynamicConv (box_head.py:687-711), one workgroup per box: F1 = roi[49 x 256] . param1[256 x 64] -> LayerNorm(64) + ReLU ->
// F2 = F1 . param2[64 x 256] -> LayerNorm(256) + ReLU -> out[49 x 256].  The per-box parameters arrive as P1T[64][256] | P2T[256][64]
// ([N][K] rows, the row order csrc/model.hip gives dynamic_layer), so both MFMA operands are K-contiguous rows read straight from
// global as float4 fragments; F1 / F2 cross LDS for the row statistics.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DC_P1 = 68, DC_P2 = 260;
// (two workgroups per CU: left to itself the compiler takes 200 VGPRs + 64 AGPRs, over the 256 a wave may have at two waves per SIMD, and
// a box's loads, products and LayerNorms then run strictly one after the other on the CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void f32_dynconv_kernel(const float* __restrict__ roi, const float* __restrict__ params,
                                                           const float* __restrict__ g1, const float* __restrict__ b1,
                                                           const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ out,
                                                           int nbox) {
#pragma clang fp contract(off)
    __shared__ float F1[64 * DC_P1];
    __shared__ float F2[49 * DC_P2];
    const int box = igemm_xcd_remap((int)blockIdx.x, nbox);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = (lane >> 5) * 4;
    const float* x = roi + (long)box * 49 * 256;
    const float* p1 = params + (long)box * 32768;
    const float* p2 = p1 + 64 * 256;

    // the second product's parameter fragments are requested first: they land under the first product
    float4v w2[2][8];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 8; ++j) w2[nb][j] = *reinterpret_cast<const float4v*>(p2 + (long)((wave * 2 + nb) * 32 + fr) * 64 + j * 8 + fk);

    // ---- product 1: wave = (row block mb, column block nb) of the 64 x 64 result, K = 256 in four chunks of 64
    {
        const int mb = wave >> 1, nb = wave & 1;
        const int prow = mb * 32 + fr;
        const bool pok = prow < 49;
        const float* ap = x + (long)(pok ? prow : 0) * 256 + fk;
        const float* bp = p1 + (long)(nb * 32 + fr) * 256 + fk;
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float4v a[2][8], b[2][8];
        auto fetch = [&](int c, int buf) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[buf][j] = pok ? *reinterpret_cast<const float4v*>(ap + c * 64 + j * 8) : (float4v){0.f, 0.f, 0.f, 0.f};
                b[buf][j] = *reinterpret_cast<const float4v*>(bp + c * 64 + j * 8);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) fetch(c + 1, (c + 1) & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma_f32(a[c & 1][j][e], b[c & 1][j][e], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) F1[(mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * DC_P1 + nb * 32 + fr] = acc[r];
    }
    __syncthreads();
    // ---- LayerNorm(64) + ReLU on rows 0..48: four lanes per row, 16 values each
    if (tid < 49 * 4) {
        const int row = tid >> 2, part = tid & 3;
        float* rp = &F1[row * DC_P1 + part * 16];
        float vals[16], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            vals[e] = rp[e];
            sum += vals[e];
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float mean = sum / 64.f;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = vals[e] - mean;
            sq += t * t;
        }
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        const float rstd = rsqrtf(sq / 64.f + 1e-5f);
#pragma unroll
        for (int e = 0; e < 16; ++e) rp[e] = fmaxf((vals[e] - mean) * rstd * g1[part * 16 + e] + b1[part * 16 + e], 0.f);
    }
    __syncthreads();
    // ---- product 2: wave w owns columns [64 w, 64 w + 64) of the 64 x 256 result, K = 64
    {
        float16v acc[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4v a[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const float4v*>(&F1[(mb * 32 + fr) * DC_P1 + j * 8 + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma_f32(a[mb][e], w2[nb][j][e], acc[mb][nb]);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                    if (row < 49) F2[row * DC_P2 + (wave * 2 + nb) * 32 + fr] = acc[mb][nb][r];
                }
    }
    __syncthreads();
    // ---- LayerNorm(256) + ReLU, one wave per row, 4 values per lane; rows go straight to global
    const float4v gg = *reinterpret_cast<const float4v*>(g2 + lane * 4);
    const float4v bb = *reinterpret_cast<const float4v*>(b2 + lane * 4);
    for (int row = wave; row < 49; row += 4) {
        const float4v t = *reinterpret_cast<const float4v*>(&F2[row * DC_P2 + lane * 4]);
        const float mean = wave_sum(t[0] + t[1] + t[2] + t[3]) / 256.f;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = t[e] - mean;
            sq += u * u;
        }
        const float rstd = rsqrtf(wave_sum(sq) / 256.f + 1e-5f);
        float4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf((t[e] - mean) * rstd * gg[e] + bb[e], 0.f);
        *reinterpret_cast<float4v*>(out + ((long)box * 49 + row) * 256 + lane * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same per-box pipeline with SPLIT operands (library option f32_split = 1, the default): both products run as three passes of the
// fp16 MFMA on (hi, lo) halves of the fp32 values -- the arithmetic of f32x3_igemm_kernel: lo_a hi_b + hi_a lo_b + hi_a hi_b, fp32
// accumulation -- instead of 256 fp32-MFMA instructions of 64 cycles per wave and box.  Each wave splits the fragments it multiplies, in
// registers, straight from the global loads; product 1 is split over K across the four waves (see the kernel), its partial sums and
// F1 cross LDS as fp32 for the LayerNorm statistics, and F1 comes back as two fp16 planes (rows of 64 halves at a pitch of 72:
// conflict-free ds_read_b128 fragments) over the same bytes.
// LayerNorm / ReLU arithmetic is the fp32 kernel's.  |value| > 65504 in the RoI tile or the parameters sets `range_flag`.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DX_PH = 72;          // halves per F1 plane row
__device__ __forceinline__ void split8(const float4v v0, const float4v v1, half8& h, half8& l, float& mx) {
#pragma unroll
    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(__builtin_fabsf(v0[e]), __builtin_fabsf(v1[e])));
    const half4 h0 = __builtin_convertvector(v0, half4), h1 = __builtin_convertvector(v1, half4);          // round to nearest even
    const half4 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, float4v), half4);            // v - hi is exact in fp32
    const half4 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, float4v), half4);
    h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float16v mfma_x3(const half8 ah, const half8 al, const half8 bh, const half8 bl, float16v acc) {
    // the two small terms first, then the leading one: f32x3_igemm_kernel's order
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void f32x3_dynconv_kernel(const float* __restrict__ roi, const float* __restrict__ params,
                                                             const float* __restrict__ g1, const float* __restrict__ b1,
                                                             const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ out,
                                                             int nbox, int* __restrict__ range_flag) {
#pragma clang fp contract(off)
    // LDS: product 1's four K-quarter partial sums [4][64][68] fp32 (69632 bytes); afterwards the same bytes hold F1 as two fp16 planes
    // (18432) and, behind them, F2 [49][260] fp32
    constexpr int PART = 64 * DC_P1;                      // floats per partial
    constexpr int F1_BYTES = 2 * 64 * DX_PH * 2;
    constexpr int LDS_BYTES = 4 * PART * 4 > F1_BYTES + 49 * DC_P2 * 4 ? 4 * PART * 4 : F1_BYTES + 49 * DC_P2 * 4;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    float* const P = reinterpret_cast<float*>(lds);
    half_t* const F1h = reinterpret_cast<half_t*>(lds);
    half_t* const F1l = F1h + 64 * DX_PH;
    float* const F2 = reinterpret_cast<float*>(lds + F1_BYTES);
    const int box = igemm_xcd_remap((int)blockIdx.x, nbox);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = (lane >> 5) * 8;          // fp16 MFMA operand map: lane l = row l & 31, k = 8 (l >> 5) .. + 8 of a 16-deep step
    const float* x = roi + (long)box * 49 * 256;
    const float* p1 = params + (long)box * 32768;
    const float* p2 = p1 + 64 * 256;
    float mx = 0.f;

    // ---- product 1, split over K: wave w multiplies k in [64 w, 64 w + 64) for the whole 64 x 64 result.  Every RoI / parameter value is
    // loaded by exactly one lane, and ALL of a box's 114 KB of first-product operands are requested before the first one is used (the
    // (row block, column block) split of the fp32 kernel double-buffers 64-deep chunks: ~32 KB in flight per workgroup, which is what
    // paced it -- 2.4-2.6 TB/s with either MFMA).
    float16v acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    {
        float4v a[2][4][2], b[2][4][2];
        const bool pok = 32 + fr < 49;          // row block 1 holds rows 32..48
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 64 * wave + 16 * ks + fk + 4 * h;
                a[0][ks][h] = *reinterpret_cast<const float4v*>(x + (long)fr * 256 + k);
                a[1][ks][h] = pok ? *reinterpret_cast<const float4v*>(x + (long)(32 + fr) * 256 + k) : (float4v){0.f, 0.f, 0.f, 0.f};
                b[0][ks][h] = *reinterpret_cast<const float4v*>(p1 + (long)fr * 256 + k);
                b[1][ks][h] = *reinterpret_cast<const float4v*>(p1 + (long)(32 + fr) * 256 + k);
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                split8(a[i][ks][0], a[i][ks][1], ah[i], al[i], mx);
                split8(b[i][ks][0], b[i][ks][1], bh[i], bl[i], mx);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma_x3(ah[mb], al[mb], bh[nb], bl[nb], acc[mb][nb]);
        }
    }
    // the second product's parameter fragments (raw fp32; wave w owns columns [64 w, 64 w + 64)): requested now, they land under the
    // reduction and the LayerNorm below
    float4v w2[2][4][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* q = p2 + (long)((wave * 2 + nb) * 32 + fr) * 64 + ks * 16 + fk;
            w2[nb][ks][0] = *reinterpret_cast<const float4v*>(q);
            w2[nb][ks][1] = *reinterpret_cast<const float4v*>(q + 4);
        }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                P[wave * PART + (mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * DC_P1 + nb * 32 + fr] = acc[mb][nb][r];
    __syncthreads();
    // ---- the four partial sums in a fixed order, LayerNorm(64) + ReLU on rows 0..48: four lanes per row, 16 values each; the result goes
    // back as (hi, lo) planes, rows 49..63 zero
    {
        const int row = tid >> 2, part = tid & 3;
        float vals[16];
        if (row < 49) {
            const float* rp = &P[row * DC_P1 + part * 16];
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                vals[e] = (rp[e] + rp[PART + e]) + (rp[2 * PART + e] + rp[3 * PART + e]);
                sum += vals[e];
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            const float mean = sum / 64.f;
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float t = vals[e] - mean;
                sq += t * t;
            }
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            const float rstd = rsqrtf(sq / 64.f + 1e-5f);
#pragma unroll
            for (int e = 0; e < 16; ++e) vals[e] = fmaxf((vals[e] - mean) * rstd * g1[part * 16 + e] + b1[part * 16 + e], 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) vals[e] = 0.f;
        }
        __syncthreads();          // every partial sum has been read
        float unused = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            half8 vh, vl;
            split8((float4v){vals[8 * h], vals[8 * h + 1], vals[8 * h + 2], vals[8 * h + 3]},
                   (float4v){vals[8 * h + 4], vals[8 * h + 5], vals[8 * h + 6], vals[8 * h + 7]}, vh, vl, unused);
            *reinterpret_cast<half8*>(F1h + row * DX_PH + part * 16 + 8 * h) = vh;
            *reinterpret_cast<half8*>(F1l + row * DX_PH + part * 16 + 8 * h) = vl;
        }
    }
    __syncthreads();
    // ---- product 2: wave w owns columns [64 w, 64 w + 64) of the 64 x 256 result, K = 64
    {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                ah[mb] = *reinterpret_cast<const half8*>(F1h + (mb * 32 + fr) * DX_PH + ks * 16 + fk);
                al[mb] = *reinterpret_cast<const half8*>(F1l + (mb * 32 + fr) * DX_PH + ks * 16 + fk);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) split8(w2[nb][ks][0], w2[nb][ks][1], bh[nb], bl[nb], mx);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma_x3(ah[mb], al[mb], bh[nb], bl[nb], acc[mb][nb]);
        }
        // (F2 lies behind the planes: no wave's fragment reads are disturbed by another wave's results)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mb * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                    if (row < 49) F2[row * DC_P2 + (wave * 2 + nb) * 32 + fr] = acc[mb][nb][r];
                }
    }
    if (range_flag && mx > 65504.f) atomicOr(range_flag, 1);          // reported, never a silent inf (f32_split = 0 has no such limit)
    __syncthreads();
    // ---- LayerNorm(256) + ReLU, one wave per row, 4 values per lane; rows go straight to global
    const float4v gg = *reinterpret_cast<const float4v*>(g2 + lane * 4);
    const float4v bb = *reinterpret_cast<const float4v*>(b2 + lane * 4);
    for (int row = wave; row < 49; row += 4) {
        const float4v t = *reinterpret_cast<const float4v*>(&F2[row * DC_P2 + lane * 4]);
        const float mean = wave_sum(t[0] + t[1] + t[2] + t[3]) / 256.f;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = t[e] - mean;
            sq += u * u;
        }
        const float rstd = rsqrtf(wave_sum(sq) / 256.f + 1e-5f);
        float4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf((t[e] - mean) * rstd * gg[e] + bb[e], 0.f);
        *reinterpret_cast<float4v*>(out + ((long)box * 49 + row) * 256 + lane * 4) = o;
    }
}

}  // namespace

// =================================================================================================================================
int dvid_f32_igemm_launch(const F32GemmParams& p0, hipStream_t s) {
    F32GemmParams p = p0;
    if (p.M <= 0 || p.Cout <= 0) return DVID_OK;
    if (p.Cin % 4 || p.Kpad % F32_BK || p.K > p.Kpad || p.ldc < p.Cout) return DVID_ERR_ARG;
    if (p.res_mode == 2 && ((p.Ho | p.Wo) & 1)) return DVID_ERR_ARG;
    const bool split = g_opt.f32_split != 0 && p.w_hi && p.w_lo;          // split (hi, lo) fp16 operands on the fp16 MFMA, or exact fp32 products on the fp32 MFMA
    // 3x3 / stride-1 layers: the halo staged and split once per channel chunk (csrc/f32_conv3x3.hip)
    if (split && g_opt.f32_conv3x3 && dvid_f32_conv3x3_supported(p)) return dvid_f32_conv3x3_launch(p, s);
    // short-K / wide-N 1x1 layers: the weight-stationary form of the same arithmetic (whole 32-row blocks; a ragged tail falls through
    // to the tiled kernel below on the remaining rows -- same values either way)
    if (split && g_opt.f32_wstat && (g_opt.f32_wstat == 2 ? dvid_f32_wstat_supported(p) : dvid_f32_wstat_preferred(p)) && p.M >= dvid_f32_wstat_tile_rows(p)) {
        const int m0 = p.M - p.M % dvid_f32_wstat_tile_rows(p);
        F32GemmParams q = p;
        q.M = m0;
        q.H = m0;          // (a 1x1 layer over contiguous rows: the row count is all the kernel reads of the geometry)
        q.W = 1;
        q.Ho = m0;
        q.Wo = 1;
        const int rc = dvid_f32_wstat_launch_tiles(q, s);
        if (rc != DVID_OK) return rc;
        if (p.M == m0) return DVID_OK;
        p.in += (long)m0 * p.Cin;
        p.out += (long)m0 * p.ldc;
        if (p.res) p.res += (long)m0 * p.Cout;
        p.M -= m0;
        p.H = p.M;
        p.W = 1;
        p.Ho = p.M;
        p.Wo = 1;
    }
    p.tiles_m = ceil_div(p.M, F32_BM);
    if (p.Cout <= 64) {
        p.tiles_n = ceil_div(p.Cout, 64);
        if (split) hipLaunchKernelGGL(f32x3_igemm_kernel<64>, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(f32_igemm_kernel<64>, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    } else {
        p.tiles_n = ceil_div(p.Cout, 128);
        if (split) hipLaunchKernelGGL(f32x3_igemm_kernel<128>, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(f32_igemm_kernel<128>, dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    }
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_prep_images_launch(const float* const* frames, float* nhwc4, int n, int h, int w, const float* mean, const float* std_,
                                hipStream_t s) {
    const long hw = (long)h * w;
    for (int f0 = 0; f0 < n; f0 += FrameTable::kMax) {
        const int nf = n - f0 < FrameTable::kMax ? n - f0 : FrameTable::kMax;
        FrameTable tab;
        for (int i = 0; i < FrameTable::kMax; ++i) tab.p[i] = frames[f0 + (i < nf ? i : 0)];
        const long npix = hw * nf;
        hipLaunchKernelGGL(f32_prep_images_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, tab, nhwc4 + (long)f0 * hw * 4, npix, hw,
                           mean[0], mean[1], mean[2], std_[0], std_[1], std_[2]);
        LAUNCH_CHECK();
    }
    return DVID_OK;
}

int dvid_f32_maxpool3x3s2_launch(const float* in, float* out, int n, int h, int w, int c, hipStream_t s) {
    if (c % 4) return DVID_ERR_ARG;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(f32_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, n, h, w, c, ho, wo);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_silu_launch(const float* x, float* y, long n, hipStream_t s) {
    if (n % 4) return DVID_ERR_ARG;
    hipLaunchKernelGGL(f32_silu_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, y, n / 4);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_modulate_launch(const float* x, const float* scale, int scale_ld, const float* shift, int shift_per_row, int shift_ld, float* y,
                             int rows, int rows_per_frame, int d, hipStream_t s) {
    if (d % 4) return DVID_ERR_ARG;
    const long n4 = (long)rows * d / 4;
    hipLaunchKernelGGL(f32_modulate_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, scale, scale_ld, shift, shift_per_row, shift_ld,
                       y, n4, rows_per_frame, d);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_roialign_launch(const RoiLevels32& lv, int channels, const float* boxes, int n_img, int boxes_per_img, float* roi_out,
                             float* mean_out, hipStream_t s) {
    if (channels != 256) return DVID_ERR_UNSUPPORTED;
    const int nbox = n_img * boxes_per_img;
    if (nbox == 0) return DVID_OK;
    hipLaunchKernelGGL(f32_roialign_kernel, dim3(nbox), dim3(256), 0, s, lv, boxes, boxes_per_img, roi_out, mean_out, nbox);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_mha_launch(const float* q, const float* k, const float* v, float* out, int batch, int lq, int lk, int nheads, int q_ld, int kv_ld,
                        int out_ld, long q_bs, long kv_bs, long out_bs, hipStream_t s) {
    if (batch <= 0 || lq <= 0) return DVID_OK;
    if (lk <= 0 || nheads <= 0 || (q_ld | kv_ld | out_ld) % 4) return DVID_ERR_ARG;
    hipLaunchKernelGGL(f32_mha_kernel, dim3(ceil_div(lq, 64), nheads, batch), dim3(64), 0, s, q, k, v, out, lq, lk, q_ld, kv_ld, out_ld, q_bs, kv_bs,
                       out_bs, 0.17677669529663688110f);          // 1 / sqrt(32)
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_swin_window_attn_launch(const float* qkv, const float* qkv_bias, const float* relbias, float* out, int batch, int H, int W, int C,
                                     int nheads, int shift, hipStream_t s) {
    if (C != nheads * 32) return DVID_ERR_UNSUPPORTED;
    const long nwin = (long)batch * ((H + 6) / 7) * ((W + 6) / 7);
    if (nwin * nheads > 0x7fffffffL) return DVID_ERR_UNSUPPORTED;
    if (nwin == 0) return DVID_OK;
    hipLaunchKernelGGL(f32_swin_window_attn_kernel, dim3((unsigned)(nwin * nheads)), dim3(64), 0, s, qkv, qkv_bias, relbias, out, H, W, C, nheads, shift,
                       0.17677669529663688110f, (int)nwin);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_dynconv_launch(const float* roi, const float* params, const float* g1, const float* b1, const float* g2, const float* b2,
                            float* out, int rows, int* range_flag, hipStream_t s) {
    if (rows <= 0) return DVID_OK;
    if (g_opt.f32_split != 0) hipLaunchKernelGGL(f32x3_dynconv_kernel, dim3(rows), dim3(256), 0, s, roi, params, g1, b1, g2, b2, out, rows, range_flag);
    else hipLaunchKernelGGL(f32_dynconv_kernel, dim3(rows), dim3(256), 0, s, roi, params, g1, b1, g2, b2, out, rows);
    LAUNCH_CHECK();
    return DVID_OK;
}
