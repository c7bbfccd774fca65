// The tail of a res2 bottleneck block as ONE launch: conv2 (3x3, 64 -> 64) + ReLU -> conv3 (1x1, 64 -> 256) + residual (the block
// input, or the block's 1x1 shortcut convolution of it) + ReLU -> the NEXT block's conv1 (1x1, 256 -> 64) + ReLU.
//
// Why: layer by layer these launches already run at the HBM rate of their bytes (conv3 + residual 13.6 GB per 304-frame pass at
// 5.3 TB/s, profiles/r03a_layers.csv), so only removing bytes makes them faster.  An identity block moves 2048 B per pixel layer by
// layer (x 512 + t1 128 w/r + t2 128 w/r + residual 512 + out 512); this kernel reads the conv1 output t1 with its halo (128 x 1.33),
// the residual (512) and writes the block output (512) and the next block's t1 (128): 1322 B per pixel, and block 0 (whose shortcut
// convolution is computed here from the 64-channel block input) 938 instead of 2176 + the next conv1's 640.  The intermediate t2 never
// leaves the registers, and neither conv1 is re-computed on a halo: the next block's t1 goes through HBM (its halo comes from the
// neighbouring patches' writes), which costs 298 B per pixel against the 170 B of re-reading the 256-channel halo -- and needs no
// second copy of the block input in LDS or registers.
//
// Structure: a workgroup of 8 waves owns an 8 x 32 patch of output pixels (patch rows = rows of the whole batch laid end to end, as
// conv3x3.hip), wave w owns patch row w from start to finish.  All three products are computed TRANSPOSED (D[n][m] = W A^T: weights
// are the MFMA's first operand, read from LDS where the workgroup stages them once), so a lane ends up with 16 channels of ONE pixel,
// and after one v_permlane32_swap per register pair with 2 x 8 consecutive channels -- which is exactly the second-operand fragment
// of the next product's K step.  So conv2's result feeds conv3 and conv3's result feeds the next conv1 from registers: no LDS round
// trip, no barrier after conv2, residual and outputs move as 16-byte pieces straight from / to the accumulator layout.  The residual
// of the wave's row (16 KB) is requested once every DMA piece of the prologue has landed (behind taps 0-2) and lands while taps 3-8 run.
//
// LDS (156 KB used, the whole 160 KB requested; one workgroup per CU): t1 halo 10 x 34 pixels x 128 B (48 KB with padding; after conv2 the next conv1's weights) |
// conv2 weights, 9 taps x [64][64] (72 KB; after conv2 the shortcut weights, or a 128-channel next conv1's 64 KB) | conv3 weights
// [256][64] (32 KB) | biases (4 KB).
//
// Same MFMA, same K order (tap, then channel, 16 per instruction; transposing a product does not change its sums) and the same
// epilogue arithmetic (fp32 + bias [+ fp16 residual], round to fp16, ReLU) as the layer-by-layer kernels: bit-identical to them
// (tests/test_gpu_kernels.py::test_bottleneck_tail_matches_layers).
#include <stdlib.h>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "options.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16_bn[4] = {0u, 0u, 0u, 0u};

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void bn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bn_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// x + float(h): v_fma_mix_f32 (h * 1.0 + x: fp16 source promoted exactly, one rounding -- the value of convert + add)
__device__ __forceinline__ float bn_mix_add_lo(unsigned int h2, float x) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(x));
    return d;
}
__device__ __forceinline__ float bn_mix_add_hi(unsigned int h2, float x) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(x));
    return d;
}

constexpr int TH = 8, TW = 32;                 // output patch: 8 rows of 32 pixels, one row per wave
constexpr int HH = TH + 2, HW = 34;            // halo rows / pitch in pixels
constexpr int PXB = 128;                       // bytes per 64-channel pixel / weight row
constexpr int A_PIECES = 48;                   // 1-KiB pieces of 8 pixels: 44 cover the 340 halo pixels; 6 per wave
constexpr int kHalo = 0;                       // ... later the next conv1's weights: 4 K chunks x [64][64]
constexpr int kW2 = A_PIECES * 1024;           // 9 taps x [64 out][64 in]; later the shortcut's [256][64]
constexpr int kW3 = kW2 + 9 * 8192;            // [256][64]
constexpr int kBias = kW3 + 32768;             // four 1-KiB slots of floats: b2 [64] | b3 [256] | b1n [64] | bs [256]
// The launch asks for ALL of the CU's LDS although kBias + 4096 bytes are used: nothing else with LDS then shares the CU.  History: the
// first build (155 KB) could sit beside a small workgroup of another stream's kernel -- the farthest-point sweep of the global-memory build
// -- and then, about once in ten 24-frame launch sequences, single patch rows came out wrong.  Cause (found with tools/lab/spin_kernel.hip
// as the neighbour: barriers alone are harmless, LDS traffic is not): ORDINARY LOADS AND LDS-DMA PIECES DO NOT RETIRE IN ISSUE ORDER
// RELATIVE TO EACH OTHER.  Each kind does among itself, but when the LDS pipe is busy the pieces lag, younger ordinary loads retire first,
// and a counted vmcnt that has ordinary loads among the pieces it counts lets a step start on weights that have not landed.  Both kernels
// here now keep ordinary loads out of every counted wait (below); the full allocation stays as a margin (library option bneck_lds = <bytes>: diagnostics).
constexpr int kBytes = 160 * 1024;

struct BneckParams {
    const half_t* t1;      // [rows][W][64]   conv1 output (ReLU applied)
    const half_t* w2;      // [64][576]       k = tap * 64 + c
    const float* b2;
    const half_t* w3;      // [256][64]
    const float* b3;
    const half_t* res;     // identity block: the block input [rows][W][256]; shortcut block: the block input [rows][W][64]
    const half_t* ws;      // [256][64] shortcut weights (SC)
    const float* bs;
    const half_t* w1n;     // [TN][256] the next block's conv1
    const float* b1n;
    half_t* out;           // [rows][W][256]
    half_t* t1n;           // [rows][W][TN]
    int H, W, nrows, tiles_x, ntiles;
};

// acc (transposed 32x32 product: register 4 r4 + r = channel 8 r4 + 4 hi + r of pixel lane % 32) -> after one half-wave exchange per
// register pair, u[8 g + e] = channel 16 g + 8 hi + e
__device__ __forceinline__ void bn_swap(const float16v& acc, unsigned int (&u)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float f = acc[r];
        u[r] = __float_as_uint(f);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane32_swap(u[8 * g + r], u[8 * g + 4 + r], false, false);
            u[8 * g + r] = sw[0];
            u[8 * g + 4 + r] = sw[1];
        }
}

// SC: the residual is the block's shortcut convolution (1x1, 64 -> 256, no ReLU) of the 64-channel block input, computed here and
// rounded to fp16 as the layer-by-layer path stores it; TN > 0: the next block's conv1 (TN = 64 output channels: the next res2 block;
// 128: res3's first block, whose 1x1 / stride-1 conv1 reads res2's output) is computed from the block output.
template <bool SC, int TN>
__global__ __launch_bounds__(512) void bneck64_tail_kernel(BneckParams p) {
    constexpr bool TAIL = TN > 0;
    constexpr int NT1 = TN > 0 ? TN / 32 : 1;  // accumulator tiles of the next conv1
    constexpr int kW1 = TN == 128 ? kW2 : kHalo;          // its weights: 4 K chunks of [TN][64]; 64 KB go where the conv2 taps were
    static_assert(TN == 0 || TN == 64 || (TN == 128 && !SC), "next conv1: 64 channels, or 128 without a shortcut in the same launch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;
    const int lid = igemm_xcd_remap((int)blockIdx.x, p.ntiles);
    const int tx = lid % p.tiles_x, ty = lid / p.tiles_x;
    const int r0 = ty * TH, x0 = tx * TW;
    const char* zero = reinterpret_cast<const char*>(g_zero16_bn);

    // ---- biases -> LDS, by DMA like everything else (a ds_write between the DMA pieces makes the compiler wait for vmcnt(0), residual
    // loads included): waves 0-3 fetch b2 / b3 / b1n / bs into 1-KiB slots (lanes past an array's end read the zero page)
    if (wave < 4) {
        const float* src = wave == 0 ? p.b2 : wave == 1 ? p.b3 : wave == 2 ? (TAIL ? p.b1n : nullptr) : (SC ? p.bs : nullptr);
        const int n16 = (wave & 1) ? 64 : (wave == 2 ? TN / 4 : 16);          // float4s in the array
        bn_glds16(src && lane < n16 ? reinterpret_cast<const char*>(src + lane * 4) : zero, smem + kBias + wave * 1024);
    }
    // ---- DMA group A: the t1 halo (6 pieces per wave) and the conv2 weights of taps 0-2; group B: taps 3-8 and the conv3 weights
    // piece of 8 rows x 128 B: lane -> (row 8 q + lane / 8, 16-byte slot lane % 8 holding logical chunk slot ^ ((row >> 1) & 7))
    const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int i = 0; i < A_PIECES / 8; ++i) {
        const int q = wave + 8 * i;
        const int pidx = 8 * q + prow;
        const int hy = pidx / HW, hx = pidx - hy * HW;
        const int gr = r0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = pidx < HH * HW && (unsigned)gr < (unsigned)p.nrows && (unsigned)gx < (unsigned)p.W;
        const int lch = pslot ^ ((hx >> 1) & 7);
        bn_glds16(ok ? reinterpret_cast<const char*>(p.t1 + ((long)gr * p.W + gx) * 64 + lch * 8) : zero, smem + kHalo + q * 1024);
    }
    // rows [8 wave, 8 wave + 8) of a [64][64] tile: the same lane mapping serves every weight tile
    const int wrow = 8 * wave + prow;
    const int wlch = pslot ^ ((wrow >> 1) & 7);
    const char* const w2src = reinterpret_cast<const char*>(p.w2 + (long)wrow * 576 + wlch * 8);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) bn_glds16(w2src + tap * 128, smem + kW2 + tap * 8192 + wave * 1024);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tap = 3; tap < 9; ++tap) bn_glds16(w2src + tap * 128, smem + kW2 + tap * 8192 + wave * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 64 * i + wrow;           // (row >> 1) & 7 == (wrow >> 1) & 7
        bn_glds16(reinterpret_cast<const char*>(p.w3 + (long)row * 64 + wlch * 8), smem + kW3 + i * 8192 + wave * 1024);
    }
    asm volatile("" ::: "memory");

    // ---- this lane's pixel: batch row r0 + wave, column x0 + lane % 32 (clamped for the loads, stores are predicated)
    const int gr = r0 + wave, gx = x0 + frow;
    const bool valid = gr < p.nrows && gx < p.W;
    const long pix = (long)(gr < p.nrows ? gr : p.nrows - 1) * p.W + (gx < p.W ? gx : p.W - 1);
    constexpr int NRES = SC ? 4 : 16;
    half8 resv[NRES];
    // ---- fragment addressing
    int a_off[3][4];                           // [dx][ks]: halo column dx + frow inside a halo row
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int hx = dx + frow;
            a_off[dx][ks] = hx * PXB + (((2 * ks + hi) ^ ((hx >> 1) & 7)) << 4);
        }
    int b_off[4];                              // weight row frow (+ 32 j) of a [..][64] tile, K step ks
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_off[ks] = frow * PXB + (((2 * ks + hi) ^ ((frow >> 1) & 7)) << 4);
    const int yimg = (r0 + wave) % p.H;
    const bool up_ok = yimg > 0, down_ok = yimg < p.H - 1;      // wave-uniform: the rows above / below belong to the same image

    float16v acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

    auto conv2_tap = [&](int tap) {
        const int dy = tap / 3, dx = tap % 3;
        if ((dy == 0 && !up_ok) || (dy == 2 && !down_ok)) return;
        const char* bst = smem + kW2 + tap * 8192;
        const char* hrow = smem + kHalo + (wave + dy) * (HW * PXB);
        half8 fw[2][4], fp[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fp[ks] = *reinterpret_cast<const half8*>(hrow + a_off[dx][ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j][ks] = *reinterpret_cast<const half8*>(bst + j * 4096 + b_off[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][ks], fp[ks], acc2[j], 0, 0, 0);
    };

    // Only DMA pieces are in flight up to here, and they retire in issue order among themselves: the counted wait is exact.  (An
    // ordinary load does NOT retire in order with them -- see bneck128_tail_kernel -- so the residual is requested only behind the
    // last counted wait.)
    bn_wait_vmcnt<10>();                       // group A landed (issued after it: group B's 10 pieces)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) conv2_tap(tap);
    bn_wait_vmcnt<0>();                        // group B landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // residual of the wave's row, requested now, used after conv2 (six taps and an epilogue away): [tile j][g] = channels 32 j + 16 g + 8 hi + [0, 8)
    if (SC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) resv[ks] = *reinterpret_cast<const half8*>(p.res + pix * 64 + 16 * ks + 8 * hi);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) resv[k] = *reinterpret_cast<const half8*>(p.res + pix * 256 + 16 * k + 8 * hi);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tap = 3; tap < 9; ++tap) conv2_tap(tap);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // nobody reads the halo or the conv2 weights any more
    asm volatile("" ::: "memory");

    // ---- the regions conv2 is done with take the next conv1's weights (4 K chunks of [64][64]) and the shortcut's
    if (TAIL) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < TN / 64; ++i)
                bn_glds16(reinterpret_cast<const char*>(p.w1n + (long)(64 * i + wrow) * 256 + c * 64 + wlch * 8),
                          smem + kW1 + c * (TN * 128) + (8 * i + wave) * 1024);
    }
    if (SC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 64 * i + wrow;
            bn_glds16(reinterpret_cast<const char*>(p.ws + (long)row * 64 + wlch * 8), smem + kW2 + i * 8192 + wave * 1024);
        }
    }
    asm volatile("" ::: "memory");

    const float* const bl = reinterpret_cast<const float*>(smem + kBias);
    // ---- conv2 epilogue: + bias, round, ReLU -> the four K-step fragments of conv3's second operand
    half8 t2f[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned int u[16];
        bn_swap(acc2[j], u);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4v blo = *reinterpret_cast<const float4v*>(bl + 32 * j + 16 * g + 8 * hi);
            const float4v bhi = *reinterpret_cast<const float4v*>(bl + 32 * j + 16 * g + 8 * hi + 4);
            float4v lo, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
            }
            const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
            half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
            t2f[2 * j + g] = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
        }
    }
    if (TAIL || SC) {
        bn_wait_vmcnt<0>();                    // (L2-warm pieces: a few hundred cycles, most of them behind the epilogue above)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- conv3 (+ shortcut) + residual + ReLU, 32 output channels at a time; the results stay as the next conv1's K-step fragments
    half8 outf[16];
    half_t* const orow = p.out + ((long)gr * p.W + gx) * 256 + 8 * hi;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        half8 fw3[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw3[ks] = *reinterpret_cast<const half8*>(smem + kW3 + jj * 4096 + b_off[ks]);
        float16v acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw3[ks], t2f[ks], acc3, 0, 0, 0);
        unsigned int us[16];
        if (SC) {
            half8 fws[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fws[ks] = *reinterpret_cast<const half8*>(smem + kW2 + jj * 4096 + b_off[ks]);
            float16v accs;
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(fws[ks], resv[ks], accs, 0, 0, 0);
            bn_swap(accs, us);
        }
        unsigned int u[16];
        bn_swap(acc3, u);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c0 = 32 * jj + 16 * g + 8 * hi;
            const float4v blo = *reinterpret_cast<const float4v*>(bl + 256 + c0);
            const float4v bhi = *reinterpret_cast<const float4v*>(bl + 256 + c0 + 4);
            float4v lo, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
            }
            uint4v rr;
            if (SC) {
                const float4v slo = *reinterpret_cast<const float4v*>(bl + 768 + c0);
                const float4v shi = *reinterpret_cast<const float4v*>(bl + 768 + c0 + 4);
                float4v a, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = __uint_as_float(us[8 * g + e]) + slo[e];
                    b[e] = __uint_as_float(us[8 * g + 4 + e]) + shi[e];
                }
                const half4 ha = __builtin_convertvector(a, half4), hb = __builtin_convertvector(b, half4);
                rr = __builtin_bit_cast(uint4v, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                rr = __builtin_bit_cast(uint4v, resv[2 * jj + g]);
            }
            lo[0] = bn_mix_add_lo(rr[0], lo[0]);
            lo[1] = bn_mix_add_hi(rr[0], lo[1]);
            lo[2] = bn_mix_add_lo(rr[1], lo[2]);
            lo[3] = bn_mix_add_hi(rr[1], lo[3]);
            hv[0] = bn_mix_add_lo(rr[2], hv[0]);
            hv[1] = bn_mix_add_hi(rr[2], hv[1]);
            hv[2] = bn_mix_add_lo(rr[3], hv[2]);
            hv[3] = bn_mix_add_hi(rr[3], hv[3]);
            const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
            half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
            o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
            outf[2 * jj + g] = o;
            if (valid) *reinterpret_cast<half8*>(orow + 32 * jj + 16 * g) = o;
        }
    }

    // ---- the next block's conv1: K = 256 over the block output held in registers
    if (TAIL) {
        float16v acc1[NT1];
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const char* wt = smem + kW1 + (k >> 2) * (TN * 128) + b_off[k & 3];
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
                const half8 fw = *reinterpret_cast<const half8*>(wt + j * 4096);
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, outf[k], acc1[j], 0, 0, 0);
            }
        }
        half_t* const trow = p.t1n + ((long)gr * p.W + gx) * TN + 8 * hi;
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
            unsigned int u[16];
            bn_swap(acc1[j], u);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4v blo = *reinterpret_cast<const float4v*>(bl + 512 + 32 * j + 16 * g + 8 * hi);
                const float4v bhi = *reinterpret_cast<const float4v*>(bl + 512 + 32 * j + 16 * g + 8 * hi + 4);
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                if (valid) *reinterpret_cast<half8*>(trow + 32 * j + 16 * g) = o;
            }
        }
    }
}

template <bool SC, int TN>
int bn_launch(const BneckParams& p, hipStream_t s) {
    static_assert(kBytes <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck64_tail_kernel<SC, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes));
        mark_on_device(attr_set);
    }
    const int lds_opt = g_opt.bneck_lds;          // diagnostics: less than the whole LDS (>= kBias + 4096)
    hipLaunchKernelGGL((bneck64_tail_kernel<SC, TN>), dim3(p.ntiles), dim3(512), lds_opt >= kBias + 4096 && lds_opt <= kBytes ? lds_opt : kBytes, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}


// ---- 128-wide bottlenecks (res3: 128 -> 128 -> 512) ----------------------------------------------------------------------------------
// The same chain -- conv2 3x3 -> conv3 + residual + ReLU -> the next block's conv1, a wave owning a patch row of 32 pixels from start
// to finish, products transposed and chained through registers -- but the weights (288 + 128 + 128 KB) no longer fit the LDS: they
// stream from L2 through a 4-stage ring of 16-KB steps, one barrier and one counted vmcnt per step:
//   conv2: 18 steps = 9 taps x 2 halves of the input channels, [128 out][64 in] each (16 MFMAs per wave);
//   conv3 + next conv1: 16 steps = 32 output channels each: the conv3 rows [32][128] and -- TAIL -- the next conv1's columns
//   [128][32] that multiply exactly those channels (8 + 8 MFMAs per wave), so the block output is consumed as it is produced
//   and only the next conv1's accumulators (64 registers) live across steps.
// The residual of a step's 32 channels is requested four steps ahead (by DMA into a per-wave ring in the halo region, which is free
// after conv2; the first four tiles into registers, requested BEFORE any DMA piece).  DMA pieces retire in issue order AMONG THEMSELVES,
// so the step's own counted wait for its weight pieces is also the wait for that residual's pieces.  CONTRACT of every counted wait in
// this file (wait_n / dma_cnt): it counts LDS-DMA pieces only.  An ordinary load does NOT retire in order with OLDER pieces (a piece
// decrements vmcnt only after its LDS write: when the LDS pipe is busy a younger ordinary load retires first and vmcnt(N) is reached
// with an older piece still in flight -- the fault described in DESIGN.md section 5).  So no ordinary global load may sit among the
// youngest N operations of a vmcnt(N) that covers a piece; loads issued BEFORE the pieces are fine (data returns in issue order), and
// stores are never counted (a wait that counts too few operations only waits longer).
// tools/check_dma_waits.py compiles this file to assembly and checks that rule on every counted wait (tests/test_host_logic.py).  LDS: t1 halo 10 x 34 pixels x 256 B (88 KB) + ring 64 KB
// + biases 3 KB.  CONV2 = false: the block's conv2 ran as its own launch (res3's first block: 3x3 / stride 2) and `t1` is its output.
// K order of conv2: tap, then channel (igemm2's; the chunked patch kernel of conv3x3.hip sums channel-chunk-major), so this
// kernel is bit-identical to the layer-by-layer launches on igemm2 and differs from the patch kernel in fp32 summation order only;
// which of the two a backbone runs is a function of the image size alone.
constexpr int H8_PIECES = 88;                  // 1-KiB pieces of 4 pixels: 85 cover the 340 halo pixels; 11 per wave
constexpr int k8Ring = H8_PIECES * 1024;
constexpr int k8Stage = 16384;
constexpr int k8Bias = k8Ring + 4 * k8Stage;   // floats: b3 [512] | b2 [128] | b1n [128]
constexpr int k8Bytes = 160 * 1024;           // all of the CU's LDS (k8Bias + 3072 are used): see kBytes

struct Bneck128Params {
    const half_t* t1;      // CONV2: [rows][W][128] conv1 output; else the conv2 output t2
    const half_t* w2;      // [128][1152]
    const float* b2;
    const half_t* w3;      // [512][128]
    const float* b3;
    const half_t* res;     // [rows][W][512]
    const half_t* w1n;     // [128][512] (TAIL)
    const float* b1n;
    half_t* out;           // [rows][W][512]
    half_t* t1n;           // [rows][W][128]
    int H, W, nrows, tiles_x, ntiles;
};

__device__ __forceinline__ void bn_wait_vmcnt_n(int n) {          // n is a constant after unrolling
    switch (n) {
        case 0: bn_wait_vmcnt<0>(); break;
        case 1: bn_wait_vmcnt<1>(); break;
        case 2: bn_wait_vmcnt<2>(); break;
        case 3: bn_wait_vmcnt<3>(); break;
        case 4: bn_wait_vmcnt<4>(); break;
        case 5: bn_wait_vmcnt<5>(); break;
        case 6: bn_wait_vmcnt<6>(); break;
        case 7: bn_wait_vmcnt<7>(); break;
        case 8: bn_wait_vmcnt<8>(); break;
        case 9: bn_wait_vmcnt<9>(); break;
        default: bn_wait_vmcnt<10>(); break;
    }
}

template <bool CONV2, bool TAIL>
__global__ __launch_bounds__(512) void bneck128_tail_kernel(Bneck128Params p) {
    constexpr int NC2 = CONV2 ? 18 : 0, NST = NC2 + 16, D3 = TAIL ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, frow = lane & 31;
    const int lid = igemm_xcd_remap((int)blockIdx.x, p.ntiles);
    const int tx = lid % p.tiles_x, ty = lid / p.tiles_x;
    const int r0 = ty * TH, x0 = tx * TW;
    const char* zero = reinterpret_cast<const char*>(g_zero16_bn);
    char* const ring = smem + k8Ring;

    // DMA pieces a wave issues for step t (its weights), and the residual loads a wave issues at the end of step u -- both
    // unconditional, so they can be counted; stores are predicated (a wave may skip them) and are not: the counts are lower bounds
    // of what was issued after a step's pieces, i.e. the waits err on the early side of the ring only.
    auto dma_cnt = [](int t) { return (t < 0 || t >= NST) ? 0 : (t < NC2 ? 2 : D3); };
    // residual: tiles 0-3 are ordinary loads into registers (in the prologue, ahead of every DMA piece),
    // tiles 4-15 ride a per-wave DMA ring of 5 x 2 KB in the halo region, which is free once conv2 is done: tile jj + 4 is requested
    // at the start of conv3 step jj.  (Ordinary loads for all of them make the compiler wait vmcnt(0) at every first use while DMA
    // pieces are in flight -- its scoreboard treats a pending LDS-DMA as out of order.)
    auto r_cnt = [&](int u) { return (u >= NC2 && u < NC2 + 12) ? 2 : 0; };
    auto wait_n = [&](int s) { return dma_cnt(s + 1) + dma_cnt(s + 2) + r_cnt(s - 3) + r_cnt(s - 2) + r_cnt(s - 1); };

    // ---- biases by DMA: b3 (two pieces), b2 | b1n (one piece: lanes 0-31 / 32-63)
    if (wave < 3) {
        const float* src = wave < 2 ? p.b3 + wave * 256 + lane * 4 : (lane < 32 ? p.b2 + lane * 4 : (TAIL ? p.b1n + (lane - 32) * 4 : nullptr));
        bn_glds16(src ? reinterpret_cast<const char*>(src) : zero, smem + k8Bias + wave * 1024);
    }
    // ---- this lane's pixel
    const int gr = r0 + wave, gx = x0 + frow;
    const bool valid = gr < p.nrows && gx < p.W;
    const long pix = (long)(gr < p.nrows ? gr : p.nrows - 1) * p.W + (gx < p.W ? gx : p.W - 1);
    const half_t* const rrow = p.res + pix * 512 + 8 * hi;
    half8 resr[4][2];
    half8 t2f[8];
    char* const rring = smem + wave * (5 * 2048);
    auto issue_res = [&](int jn) {             // tile jn's two pieces: every lane fetches the 16 bytes it will add, and reads them back lane-linearly
#pragma unroll
        for (int g = 0; g < 2; ++g) bn_glds16(reinterpret_cast<const char*>(rrow + 32 * jn + 16 * g), rring + (jn % 5) * 2048 + g * 1024);
    };
    // The first four residual tiles: ordinary loads, issued BEFORE any DMA piece and never counted.  An ordinary load and an LDS-DMA
    // piece do not retire in issue order relative to each other (each kind does among itself): with such loads between the pieces of
    // the ring a counted vmcnt let a step start on weights that had not landed -- once the LDS pipe was busy with another workgroup's
    // traffic (tools/diag_chain_contention.py, SIDE=spin:...:3).  The waits below count DMA pieces only.
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int g = 0; g < 2; ++g) resr[jj][g] = *reinterpret_cast<const half8*>(rrow + 32 * jj + 16 * g);
    asm volatile("" ::: "memory");
    if (CONV2) {
        // t1 halo: piece q = wave + 8 i covers halo pixels [4 q, 4 q + 4); lane -> (pixel, 16-byte slot holding chunk slot ^ (hx & 15))
#pragma unroll
        for (int i = 0; i < H8_PIECES / 8; ++i) {
            const int q = wave + 8 * i;
            const int pidx = 4 * q + (lane >> 4);
            const int hy = pidx / HW, hx = pidx - hy * HW;
            const int hgr = r0 - 1 + hy, hgx = x0 - 1 + hx;
            const bool ok = pidx < HH * HW && (unsigned)hgr < (unsigned)p.nrows && (unsigned)hgx < (unsigned)p.W;
            const int lch = (lane & 15) ^ (hx & 15);
            bn_glds16(ok ? reinterpret_cast<const char*>(p.t1 + ((long)hgr * p.W + hgx) * 128 + lch * 8) : zero, smem + q * 1024);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) t2f[k] = *reinterpret_cast<const half8*>(p.t1 + pix * 128 + 16 * k + 8 * hi);
    }
    asm volatile("" ::: "memory");

    // ---- weight pieces of step t into stage t & 3
    const int prow = lane >> 3, pslot = lane & 7;
    const int wrow = 8 * wave + prow;                                   // conv2: rows wrow and wrow + 64 of the [128][64] tile
    const char* const w2src = CONV2 ? reinterpret_cast<const char*>(p.w2 + (long)wrow * 1152 + (pslot ^ ((wrow >> 1) & 7)) * 8) : zero;
    const int q3 = wave & 3, kh3 = wave >> 2, row3 = 8 * q3 + prow;       // conv3: rows row3 of the [32][128] tile, K half kh3
    const char* const w3src = reinterpret_cast<const char*>(p.w3 + (long)row3 * 128 + kh3 * 64 + (pslot ^ ((row3 >> 1) & 7)) * 8);
    const int rown = 16 * wave + (lane >> 2);                            // next conv1: rows rown of the [128][32] column slice
    const char* const w1src = TAIL ? reinterpret_cast<const char*>(p.w1n + (long)rown * 512 + ((lane & 3) ^ ((rown >> 2) & 3)) * 8) : zero;
    auto issue_step = [&](int t) {
        if (t >= NST) return;
        char* const stg = ring + (t & 3) * k8Stage;
        if (t < NC2) {
            const int koff = ((t >> 1) * 128 + (t & 1) * 64) * 2;
            bn_glds16(w2src + koff, stg + wave * 1024);
            bn_glds16(w2src + koff + 64 * 1152 * 2, stg + (wave + 8) * 1024);
        } else {
            const int jj = t - NC2;
            bn_glds16(w3src + (long)jj * (32 * 128 * 2), stg + kh3 * 4096 + q3 * 1024);
            if (TAIL) bn_glds16(w1src + jj * 64, stg + 8192 + wave * 1024);
        }
    };
    issue_step(0);
    issue_step(1);
    issue_step(2);
    asm volatile("" ::: "memory");

    int b_off[4];                              // row frow (+ 32 j) of a 128-byte-pitch weight tile, K step ks
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_off[ks] = frow * PXB + (((2 * ks + hi) ^ ((frow >> 1) & 7)) << 4);
    const float* const bl = reinterpret_cast<const float*>(smem + k8Bias);

    if (CONV2) {
        const int yimg = (r0 + wave) % p.H;
        const bool up_ok = yimg > 0, down_ok = yimg < p.H - 1;
        int a_base[3], a_key[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            a_base[dx] = (dx + frow) * 256;
            a_key[dx] = (dx + frow) & 15;
        }
        float16v acc2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NC2; ++s) {
            bn_wait_vmcnt_n(wait_n(s));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_step(s + 3);
            const int tap = s >> 1, kh = s & 1, dy = tap / 3, dx = tap % 3;
            if (!((dy == 0 && !up_ok) || (dy == 2 && !down_ok))) {
                const char* stg = ring + (s & 3) * k8Stage;
                const char* hrow = smem + (wave + dy) * (HW * 256) + a_base[dx];
                half8 fp[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fp[ks] = *reinterpret_cast<const half8*>(hrow + (((2 * (4 * kh + ks) + hi) ^ a_key[dx]) << 4));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    half8 fw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const half8*>(stg + j * 4096 + b_off[ks]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fp[ks], acc2[j], 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
        }
        // conv2 epilogue: + bias, round, ReLU -> the eight K-step fragments of conv3's second operand
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int u[16];
            bn_swap(acc2[j], u);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4v blo = *reinterpret_cast<const float4v*>(bl + 512 + 32 * j + 16 * g + 8 * hi);
                const float4v bhi = *reinterpret_cast<const float4v*>(bl + 512 + 32 * j + 16 * g + 8 * hi + 4);
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                t2f[2 * j + g] = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
            }
        }
    }

    // ---- conv3 + residual + ReLU, 32 channels per step, each step's output feeding the next conv1's accumulators
    float16v acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    half_t* const orow = p.out + ((long)gr * p.W + gx) * 512 + 8 * hi;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int s = NC2 + jj;
        bn_wait_vmcnt_n(wait_n(s));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_step(s + 3);
        if (jj < 12) issue_res(jj + 4);
        asm volatile("" ::: "memory");
        const char* stg = ring + (s & 3) * k8Stage;
        float16v acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const half8 fw = *reinterpret_cast<const half8*>(stg + (kk >> 2) * 4096 + b_off[kk & 3]);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, t2f[kk], acc3, 0, 0, 0);
        }
        unsigned int u[16];
        bn_swap(acc3, u);
        half8 outf[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c0 = 32 * jj + 16 * g + 8 * hi;
            const float4v blo = *reinterpret_cast<const float4v*>(bl + c0);
            const float4v bhi = *reinterpret_cast<const float4v*>(bl + c0 + 4);
            float4v lo, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
            }
            const uint4v rr = jj < 4 ? __builtin_bit_cast(uint4v, resr[jj & 3][g])
                                     : *reinterpret_cast<const uint4v*>(rring + (jj % 5) * 2048 + g * 1024 + lane * 16);
            lo[0] = bn_mix_add_lo(rr[0], lo[0]);
            lo[1] = bn_mix_add_hi(rr[0], lo[1]);
            lo[2] = bn_mix_add_lo(rr[1], lo[2]);
            lo[3] = bn_mix_add_hi(rr[1], lo[3]);
            hv[0] = bn_mix_add_lo(rr[2], hv[0]);
            hv[1] = bn_mix_add_hi(rr[2], hv[1]);
            hv[2] = bn_mix_add_lo(rr[3], hv[2]);
            hv[3] = bn_mix_add_hi(rr[3], hv[3]);
            const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
            half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
            o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
            outf[g] = o;
            if (valid) *reinterpret_cast<half8*>(orow + 32 * jj + 16 * g) = o;
        }
        if (TAIL) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half8 fw = *reinterpret_cast<const half8*>(stg + 8192 + j * 2048 + frow * 64 + (((2 * g + hi) ^ ((frow >> 2) & 3)) << 4));
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, outf[g], acc1[j], 0, 0, 0);
                }
        }
        asm volatile("" ::: "memory");
    }

    if (TAIL) {
        half_t* const trow = p.t1n + ((long)gr * p.W + gx) * 128 + 8 * hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int u[16];
            bn_swap(acc1[j], u);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4v blo = *reinterpret_cast<const float4v*>(bl + 640 + 32 * j + 16 * g + 8 * hi);
                const float4v bhi = *reinterpret_cast<const float4v*>(bl + 640 + 32 * j + 16 * g + 8 * hi + 4);
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + blo[e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bhi[e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                if (valid) *reinterpret_cast<half8*>(trow + 32 * j + 16 * g) = o;
            }
        }
    }
}

template <bool CONV2, bool TAIL>
int bn128_launch(const Bneck128Params& p, hipStream_t s) {
    static_assert(k8Bytes <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck128_tail_kernel<CONV2, TAIL>), hipFuncAttributeMaxDynamicSharedMemorySize, k8Bytes));
        mark_on_device(attr_set);
    }
    const int lds_opt = g_opt.bneck_lds;          // diagnostics: less than the whole LDS (>= k8Bias + 3072)
    hipLaunchKernelGGL((bneck128_tail_kernel<CONV2, TAIL>), dim3(p.ntiles), dim3(512), lds_opt >= k8Bias + 3072 && lds_opt <= k8Bytes ? lds_opt : k8Bytes, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

// The shape rule: the map is at least as large as conv3x3.hip asks for its patch kernels (W within 1/8 of a multiple of 32, 512
// pixels) -- a function of the image size only.  Values do not depend on it (bit-identical to the layer-by-layer launches).
int dvid_igemm_set_bottleneck_fusion(int mode) {          // g_opt.bneck_fuse: 0 off; 1 by the shape rule; 2 wherever the stage fits; -1 the default (1)
    if (mode < -1 || mode > 2) return DVID_ERR_ARG;
    g_opt.bneck_fuse = mode < 0 ? DvidOptions().bneck_fuse : mode;
    return DVID_OK;
}
bool dvid_bneck64_tail_preferred(int H, int W) {
    const int mode = g_opt.bneck_fuse;
    if (!mode) return false;
    if (mode >= 2) return true;
    return ceil_div(W, TW) * TW * 7 <= W * 8 && H * W >= 512;
}

int dvid_bneck64_tail_launch(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res,
                             const half_t* ws, const float* bs, const half_t* w1n, const float* b1n, int n_next, half_t* out, half_t* t1n,
                             int n, int H, int W, hipStream_t s) {
    if (!t1 || !w2 || !b2 || !w3 || !b3 || !res || !out || n <= 0 || H <= 0 || W <= 0) return DVID_ERR_ARG;
    if ((ws && !bs) || (w1n && (!b1n || !t1n))) return DVID_ERR_ARG;
    if (w1n && n_next != 64 && !(n_next == 128 && !ws)) return DVID_ERR_UNSUPPORTED;
    BneckParams p;
    p.t1 = t1;
    p.w2 = w2;
    p.b2 = b2;
    p.w3 = w3;
    p.b3 = b3;
    p.res = res;
    p.ws = ws;
    p.bs = bs;
    p.w1n = w1n;
    p.b1n = b1n;
    p.out = out;
    p.t1n = t1n;
    p.H = H;
    p.W = W;
    p.nrows = n * H;
    p.tiles_x = ceil_div(W, TW);
    p.ntiles = p.tiles_x * ceil_div(p.nrows, TH);
    if (ws) return w1n ? bn_launch<true, 64>(p, s) : bn_launch<true, 0>(p, s);
    if (!w1n) return bn_launch<false, 0>(p, s);
    return n_next == 128 ? bn_launch<false, 128>(p, s) : bn_launch<false, 64>(p, s);
}

// 128-wide blocks (res3).  w2 == null: `t1` is the conv2 output (the block's conv2 ran as its own launch); w1n == null: no next conv1.
int dvid_bneck128_tail_launch(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res,
                              const half_t* w1n, const float* b1n, half_t* out, half_t* t1n, int n, int H, int W, hipStream_t s) {
    if (!t1 || !w3 || !b3 || !res || !out || n <= 0 || H <= 0 || W <= 0) return DVID_ERR_ARG;
    if ((w2 && !b2) || (w1n && (!b1n || !t1n))) return DVID_ERR_ARG;
    Bneck128Params p;
    p.t1 = t1;
    p.w2 = w2;
    p.b2 = b2 ? b2 : b3;          // (the bias piece is fetched either way)
    p.w3 = w3;
    p.b3 = b3;
    p.res = res;
    p.w1n = w1n;
    p.b1n = b1n;
    p.out = out;
    p.t1n = t1n;
    p.H = H;
    p.W = W;
    p.nrows = n * H;
    p.tiles_x = ceil_div(W, TW);
    p.ntiles = p.tiles_x * ceil_div(p.nrows, TH);
    if (w2) return w1n ? bn128_launch<true, true>(p, s) : bn128_launch<true, false>(p, s);
    return w1n ? bn128_launch<false, true>(p, s) : bn128_launch<false, false>(p, s);
}
