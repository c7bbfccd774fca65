// 3x3 / stride-1 / pad-1 convolution (every bottleneck conv2 but the three strided ones, the FPN output convolutions) with the
// A operand staged ONCE per input-channel chunk instead of once per filter tap.
//
// Why: in igemm2 the 256x256x32 step moves 32 KB through the global -> LDS DMA (16 KB of im2col rows + 16 KB of weights), and
// that fill path, not the MFMA pipe, sets the pace of the long-K layers -- tools/lab/gemm_pingpong with half of the DMA pieces
// removed runs at 1370 TFLOP/s against 866 with all of them (profiles/r02_lab_dma_ablation.txt).  A 3x3 convolution reads every
// input pixel nine times, once per tap; here a workgroup owns an 8 x 32 patch of output pixels, stages the 10 x 34 halo of the
// patch for 32 input channels (24 KB, pitch 36 pixels) and takes the A fragments of all nine taps from it by shifting the LDS
// read address.  DMA per 256x256x32 step: 16 KB of weights + 24 / 9 KB of pixels = 18.7 KB instead of 32.
//
// Patch rows are rows of the batch laid end to end (row R = image * H + y), so a patch may straddle two images and no row of a
// tile is wasted whatever H is (H = 38: 988 tiles per 104 frames instead of 1040 -- four rounds of the 256 CUs, not five).  One
// 32-row MFMA block is one patch row, so "the row above / below belongs to another image (or lies outside)" is a per-fragment,
// wave-uniform fact: such a fragment is read from a row of zeros in LDS.  Left / right borders are zero-filled by the DMA.
//
// K order: channel chunk outermost, then tap, then channel inside the chunk (igemm2: tap, then channel) -- the same fp16
// products, another fp32 summation order; which of the two kernels a layer runs on is a function of its shape only (never of a
// timing), so results stay reproducible.
//
// Schedule: the anti-phase schedule of igemm2 (NSTAGE 5): 8 waves, wave rows 0-3 / 4-7 alternate between a read slot R(s) (all
// fragments of step s into registers) and an MFMA slot M(s) (the MFMAs of step s with the DMA pieces of step s + 3 between them),
// one row a slot behind the other, one barrier per slot.  Weights: 4-stage ring of BN x 32 tiles.  Pixels: two halo buffers; the 3
// pieces per wave of chunk c + 1 go out in the MFMA slots of taps 0, 1, 2 of chunk c.  In-order retirement makes one counted
// s_waitcnt per read slot enough: B(s + 1) has landed when at most the pieces issued in M(s - 1) are outstanding.
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int TH = 8, TW = 32;                 // output patch: 8 rows of 32 pixels = 256 GEMM rows, one 32-row MFMA block per patch row
constexpr int HH = TH + 2, HW = 36;            // halo rows / halo pitch in pixels (34 used; a multiple of 4 keeps the swizzle key a function of the column)
constexpr int PX_BYTES = 64;                   // 32 channels of one pixel
constexpr int A_PIECES = 24;                   // 1-KiB DMA pieces (16 pixels each) per halo chunk: 384 >= 360 pixels, 3 per wave
constexpr int A_BUF = A_PIECES * 1024;
constexpr int NB = 4;                          // weight ring depth

template <int BN, int WN>
struct Halo {
    static constexpr int WM = 8 / WN;                       // waves along M
    static constexpr int TM = 8 / WM, TN = BN / (32 * WN);  // 32x32 MFMA tiles per wave
    static constexpr int B_STAGE = BN * PX_BYTES;
    static constexpr int B_IT = BN / 16 / 8;                // weight pieces per wave per step
    static constexpr int CP = BN + 4;                       // fp32 epilogue pitch
    static constexpr int PASSES = BN > 128 ? 2 : 1;         // epilogue passes over the 256 rows
    static constexpr int GR = 256 / PASSES;
    static constexpr int kZRow = 2 * A_BUF + NB * B_STAGE;  // a halo row of zeros (fragments whose input row is another image's)
    static constexpr int kRing = kZRow + HW * PX_BYTES;
    static constexpr int kC = GR * CP * 4;
    static constexpr int kBytes = kRing > kC ? kRing : kC;
    static_assert(B_IT >= 1, "every wave stages at least one weight piece per step");
};

template <int BN, int WN>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(IgemmParams p, int tiles_x, int tiles_y) {
    using C = Halo<BN, WN>;
    constexpr int TM = C::TM, TN = C::TN, B_IT = C::B_IT, B_STAGE = C::B_STAGE, CP = C::CP;
    constexpr int NM = TM * TN * 2;            // MFMAs per wave per step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // schedule phase: waves 0-3 lead, 4-7 run one slot behind (one wave of each per SIMD)
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int lid = igemm_xcd_remap((int)blockIdx.x, ntiles);
    const int tile_n = lid % p.tiles_n;
    const int tile_m = lid / p.tiles_n;
    const int tx = tile_m % tiles_x, ty = tile_m / tiles_x;
    const int r0 = ty * TH, x0 = tx * TW, n0 = tile_n * BN;          // r0: first batch row (image * H + y) of the patch
    const int nrows = (p.M / (p.H * p.W)) * p.H;
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    if (tid < HW * PX_BYTES / 16) *reinterpret_cast<float4v*>(smem + C::kZRow + tid * 16) = float4v{0.f, 0.f, 0.f, 0.f};   // published by the prologue barrier

    // ---- halo DMA: piece q = wave + 8 i covers halo pixels [16 q, 16 q + 16); lane -> (pixel, 16-byte slot)
    const char* a_ptr[3];
    int a_step[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pidx = 16 * (wave + 8 * i) + (lane >> 2);
        const int hy = pidx / HW, hx = pidx - hy * HW;
        const int gr = r0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = pidx < HH * HW && (unsigned)gr < (unsigned)nrows && (unsigned)gx < (unsigned)p.W;
        const int lch = (lane & 3) ^ ((hx >> 2) & 3);          // logical channel group stored in this slot
        a_ptr[i] = ok ? reinterpret_cast<const char*>(p.in + ((long)gr * p.W + gx) * p.Cin + lch * 8) : zero;
        a_step[i] = ok ? PX_BYTES : 0;
    }
    const char* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = 16 * (wave + 8 * i) + (lane >> 2);
        b_ptr[i] = reinterpret_cast<const char*>(p.w + (long)(n0 + row) * p.Kpad + ((lane & 3) ^ ((row >> 2) & 3)) * 8);
    }
    const int nch = p.Cin >> 5;                // channel chunks
    char* const a_lds = smem;
    char* const b_lds = smem + 2 * A_BUF;
    // halo piece `i` of chunk `c` (the zero page past the last chunk: the wait counts stay the same for every step)
    auto issue_a = [&](int c, int i) {
        glds16(c < nch ? a_ptr[i] : zero, a_lds + (c & 1) * A_BUF + (wave + 8 * i) * 1024);
        a_ptr[i] += a_step[i];
    };
    // weight pieces of step (c, tap) into ring slot (c + tap) & 3   [9 c + tap = c + tap mod 4]
    auto issue_b = [&](int c, int tap, int i) {
        const long koff = (long)(tap * p.Cin + c * 32) * 2;
        glds16(c < nch ? b_ptr[i] + koff : zero, b_lds + ((c + tap) & 3) * B_STAGE + (wave + 8 * i) * 1024);
    };

    // ---- fragment addressing
    const int frow = lane & 31;
    int a_off[3][2];                           // [dx][ks]: halo column dx + frow inside a halo row
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int hx = dx + frow;
            a_off[dx][ks] = hx * PX_BYTES + (((2 * ks + (lane >> 5)) ^ ((hx >> 2) & 3)) << 4);
        }
    // patch row wm * TM + i of this wave: is the input row above (dy = 0) / below (dy = 2) a row of the same image?
    unsigned row_ok = 0;                       // bit 2 i: above, bit 2 i + 1: below (wave-uniform)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int yimg = (r0 + wm * TM + i) % p.H;
        row_ok |= (yimg > 0 ? 1u : 0u) << (2 * i) | (yimg < p.H - 1 ? 2u : 0u) << (2 * i);
    }
    const char* const zrow = smem + C::kZRow;
    int b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_off[ks] = (wn * (BN / WN) + frow) * PX_BYTES + (((2 * ks + (lane >> 5)) ^ ((frow >> 2) & 3)) << 4);

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: halo of chunk 0, weights of steps 0, 1, 2
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_a(0, i);
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < B_IT; ++i) issue_b(0, s, i);
    wait_vmcnt<2 * B_IT>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the zero row
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    for (int c = 0; c < nch; ++c) {
        const char* abuf = a_lds + (c & 1) * A_BUF;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            // ---- R(s): fragments of step s = 9 c + tap
            const char* bst = b_lds + ((c + tap) & 3) * B_STAGE;
            half8 fa[TM][2], fb[TN][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bool ok = dy == 1 || ((row_ok >> (2 * i + (dy >> 1))) & 1u);
                    const char* hrow = ok ? abuf + (wm * TM + i + dy) * (HW * PX_BYTES) : zrow;
                    fa[i][ks] = *reinterpret_cast<const half8*>(hrow + a_off[dx][ks]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j][ks] = *reinterpret_cast<const half8*>(bst + b_off[ks] + j * 32 * PX_BYTES);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // issued so far: everything up to M(s - 1); the weights of step s + 1 (and with them every older piece, the halo of
            // the next chunk included) have landed once only the pieces of M(s - 1) are outstanding
            if (tap >= 1 && tap <= 3) wait_vmcnt<B_IT + 1>(); else wait_vmcnt<B_IT>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(s): MFMAs of step s; weights of step s + 3 and, in taps 0-2, one halo piece of chunk c + 1 between them
            const int c3 = tap < 6 ? c : c + 1, tap3 = tap < 6 ? tap + 3 : tap - 6;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int ks = q / (TM * TN), i = (q / TN) % TM, j = q % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);          // transposed (D[n][m]): same sums, see the epilogue
                if ((q & 3) == 1) {
                    const int k = q >> 2;
                    if (k < B_IT) issue_b(c3, tap3, k);
                    else if (k == B_IT && tap < 3) issue_a(c + 1, tap);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();                           // the look-ahead pieces past the last step target LDS the epilogue re-uses
    __syncthreads();

    // ---- epilogue: fp32 tile through LDS, + bias, ReLU, fp16 rows of 8 channels per thread
    constexpr int VPR = BN / 8, ERPP = 512 / VPR, GR = C::GR, EROWS = GR / ERPP;
    float* Cs = reinterpret_cast<float*>(smem);
    const int c8 = (tid % VPR) * 8;
    const int n = n0 + c8;
    float4v b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        b_lo = *reinterpret_cast<const float4v*>(p.bias + n);
        b_hi = *reinterpret_cast<const float4v*>(p.bias + n + 4);
    }
    half_t* const outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int pass = 0; pass < C::PASSES; ++pass) {
        if (pass) __syncthreads();
        if ((wm * TM * 32) / GR == pass) {
            const int rbase = wm * TM * 32 - pass * GR;
            // the products are computed transposed (the weight fragment is the MFMA's first operand): a lane holds 4 consecutive channels of
            // ONE pixel per register quad -- four 16-byte LDS writes per accumulator tile instead of sixteen scalar ones
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = rbase + i * 32 + (lane & 31);
                        const int col = wn * (BN / WN) + j * 32 + 8 * r4 + 4 * (lane >> 5);
                        *reinterpret_cast<float4v*>(Cs + row * CP + col) =
                            (float4v){acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                    }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EROWS; ++e) {
            const int rl = tid / VPR + e * ERPP;
            const int r = pass * GR + rl;
            const int gr = r0 + (r >> 5), gx = x0 + (r & 31);
            if (gr < nrows && gx < p.W) {
                const float* csp = Cs + rl * CP + c8;
                float4v lo = *reinterpret_cast<const float4v*>(csp) + b_lo;
                float4v hi = *reinterpret_cast<const float4v*>(csp + 4) + b_hi;
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hi, half4);
                half8 hv = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (p.relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
                *reinterpret_cast<half8*>(outp + ((long)gr * p.W + gx) * p.ldc + n) = hv;
            }
        }
    }
}


// ---- 64 -> 64 channels (the res2 conv2 layers): all 64 input channels of the halo at once ----------------------------------------
// K = 576 is too short for a channel-chunk loop, and 128 bytes per pixel make the whole 10 x 34 halo 44 KB: it is staged once, in the
// prologue; the K loop is the nine taps (one 64-row x 64-channel weight tile each, 4-stage ring, counted vmcnt, one barrier per tap).
// 4 waves, each 2 patch rows x 64 output channels; 76 KB of LDS, so two workgroups share a CU and cover each other's prologue and
// epilogue.  DMA per 256-row tile: 44 + 72 KB against ~430 KB in igemm2 (im2col rows 9 x 32 KB + weights per 128 rows).  K order is
// igemm2's (tap, then channel): results are bit-identical to it.
constexpr int C64_HW = 34;                                  // halo pitch in pixels
constexpr int C64_PX = 128;                                 // bytes per pixel (64 channels)
constexpr int C64_A_PIECES = 44;                            // 8 pixels per 1-KiB piece: 352 >= 340 pixels, 11 per wave
constexpr int C64_A = C64_A_PIECES * 1024;
constexpr int C64_B_STAGE = 64 * C64_PX;                    // 64 output channels x 64 input channels of one tap
constexpr int C64_CP = 68;
constexpr int C64_BYTES = C64_A + NB * C64_B_STAGE;         // 77824 >= 256 * 68 * 4 (epilogue tile)

__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(IgemmParams p, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = igemm_xcd_remap((int)blockIdx.x, p.tiles_m);
    const int tx = lid % tiles_x, ty = lid / tiles_x;
    const int r0 = ty * TH, x0 = tx * TW;
    const int nrows = (p.M / (p.H * p.W)) * p.H;
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    char* const a_lds = smem;
    char* const b_lds = smem + C64_A;

    // ---- prologue: the whole halo (11 pieces per wave), weights of taps 0, 1, 2 (2 pieces per wave each)
#pragma unroll
    for (int i = 0; i < C64_A_PIECES / 4; ++i) {
        const int q = wave + 4 * i;
        const int pidx = 8 * q + (lane >> 3);
        const int hy = pidx / C64_HW, hx = pidx - hy * C64_HW;
        const int gr = r0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = pidx < HH * C64_HW && (unsigned)gr < (unsigned)nrows && (unsigned)gx < (unsigned)p.W;
        const int lch = (lane & 7) ^ ((hx >> 1) & 7);
        glds16(ok ? reinterpret_cast<const char*>(p.in + ((long)gr * p.W + gx) * 64 + lch * 8) : zero, a_lds + q * 1024);
    }
    const char* b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + 4 * i) + (lane >> 3);
        b_ptr[i] = reinterpret_cast<const char*>(p.w + (long)row * p.Kpad + ((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
    auto issue_b = [&](int tap) {
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(b_ptr[i] + tap * 128, b_lds + (tap & 3) * C64_B_STAGE + (wave + 4 * i) * 1024);
    };
    issue_b(0);
    issue_b(1);
    issue_b(2);

    // ---- fragment addressing: wave w owns patch rows 2 w, 2 w + 1 and all 64 output channels
    const int frow = lane & 31;
    int a_off[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int hx = dx + frow;
            a_off[dx][ks] = hx * C64_PX + (((2 * ks + (lane >> 5)) ^ ((hx >> 1) & 7)) << 4);
        }
    int b_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_off[ks] = frow * C64_PX + (((2 * ks + (lane >> 5)) ^ ((frow >> 1) & 7)) << 4);
    unsigned row_ok = 0;                       // bit 2 i: the row above patch row 2 w + i is a row of the same image; bit 2 i + 1: the row below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yimg = (r0 + 2 * wave + i) % p.H;
        row_ok |= (yimg > 0 ? 1u : 0u) << (2 * i) | (yimg < p.H - 1 ? 2u : 0u) << (2 * i);
    }

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        // weights of this tap (and, at tap 0, the halo) landed; taps tap + 1, tap + 2 may stay in flight
        if (tap <= 6) wait_vmcnt<4>(); else if (tap == 7) wait_vmcnt<2>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // ... for every wave; the stage of tap - 1 is no longer read
        if (tap + 3 < 9) issue_b(tap + 3);
        const char* bst = b_lds + (tap & 3) * C64_B_STAGE;
        half8 fb[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j][ks] = *reinterpret_cast<const half8*>(bst + b_off[ks] + j * 32 * C64_PX);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // a row above / below that belongs to another image (or lies outside the batch) contributes nothing: wave-uniform skip
            if (dy != 1 && !((row_ok >> (2 * i + (dy >> 1))) & 1u)) continue;
            const char* hrow = a_lds + (2 * wave + i + dy) * (C64_HW * C64_PX);
            half8 fa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const half8*>(hrow + a_off[dx][ks]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[j][ks], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue: the 256 x 64 fp32 tile through LDS; 8 threads finish a row of 64 channels
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * C64_CP + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();
    const int c8 = (tid & 7) * 8;
    float4v b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        b_lo = *reinterpret_cast<const float4v*>(p.bias + c8);
        b_hi = *reinterpret_cast<const float4v*>(p.bias + c8 + 4);
    }
    half_t* const outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = (tid >> 3) + e * 32;
        const int gr = r0 + (r >> 5), gx = x0 + (r & 31);
        if (gr < nrows && gx < p.W) {
            const float* csp = Cs + r * C64_CP + c8;
            float4v lo = *reinterpret_cast<const float4v*>(csp) + b_lo;
            float4v hi = *reinterpret_cast<const float4v*>(csp + 4) + b_hi;
            const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hi, half4);
            half8 hv = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
            if (p.relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
            *reinterpret_cast<half8*>(outp + ((long)gr * p.W + gx) * p.ldc + c8) = hv;
        }
    }
}

int launch_c64(IgemmParams p, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_done)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C64_BYTES));
        mark_on_device(attr_done);
    }
    const int tiles_x = ceil_div(p.W, TW);
    p.tiles_m = tiles_x * ceil_div((p.M / (p.H * p.W)) * p.H, TH);
    p.tiles_n = 1;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(p.tiles_m), dim3(256), C64_BYTES, s, p, tiles_x);
    LAUNCH_CHECK();
    return DVID_OK;
}

// ---- the space-to-depth stem: 4x4 taps / stride 1 over 16 channels (32 bytes per pixel), 64 output channels -------------------------
// (model.hip make_stem_s2d: the 7x7 / stride-2 stem over the 2x2 space-to-depth image; window rows y - 2 .. y + 1, columns x - 2 .. x + 1.)
// One MFMA K step (16 halves) is one tap.  Per 8 x 32 patch: the 11 x 35 halo (12 KB) and all 16 taps' weights (32 KB, tap-major) are
// staged in the prologue -- 44 KB of DMA per 256 rows against 192 KB in igemm2 -- and 48 KB of LDS put three workgroups on a CU, which
// cover each other's prologue and epilogue.  32-byte rows: two 16-byte slots per pixel / weight row, slot = half ^ bit 3 of the
// column (rows p and p + 8 would meet in the same banks).  K order is igemm2's (tap by tap): bit-identical.
constexpr int S2D_HW = 35, S2D_HH = 11, S2D_PX = 32;
constexpr int S2D_A = 16 * 1024;                            // 13 pieces of 32 pixels cover the 385 halo pixels; 16 issued (4 per wave)
constexpr int S2D_B = 16 * 64 * S2D_PX;                     // [tap][n][32 B]
constexpr int S2D_CP = 68;
constexpr int S2D_BYTES = S2D_A + S2D_B;                    // 49152 >= 128 * 68 * 4 (epilogue half tile)

__global__ __launch_bounds__(256) void conv4x4_s2d_kernel(IgemmParams p, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = igemm_xcd_remap((int)blockIdx.x, p.tiles_m);
    const int tx0 = lid % tiles_x, ty0 = lid / tiles_x;
    const int r0 = ty0 * TH, x0 = tx0 * TW;
    const int nrows = (p.M / (p.H * p.W)) * p.H;
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    char* const a_lds = smem;
    char* const b_lds = smem + S2D_A;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave + 4 * i;
        const int pidx = 32 * q + (lane >> 1);
        const int hy = pidx / S2D_HW, hx = pidx - hy * S2D_HW;
        const int gr = r0 - 2 + hy, gx = x0 - 2 + hx;
        const bool ok = pidx < S2D_HH * S2D_HW && (unsigned)gr < (unsigned)nrows && (unsigned)gx < (unsigned)p.W;
        const int half = (lane & 1) ^ ((hx >> 3) & 1);
        glds16(ok ? reinterpret_cast<const char*>(p.in + ((long)gr * p.W + gx) * 16 + half * 8) : zero, a_lds + q * 1024);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = wave + 4 * i;                         // piece: tap q >> 1, output channels 32 (q & 1) ..
        const int n = 32 * (q & 1) + (lane >> 1);
        const int half = (lane & 1) ^ ((n >> 3) & 1);
        glds16(reinterpret_cast<const char*>(p.w + (long)n * p.Kpad + (q >> 1) * 16 + half * 8), b_lds + q * 1024);
    }

    const int frow = lane & 31, hsel = lane >> 5;
    int a_off[4];
#pragma unroll
    for (int tx = 0; tx < 4; ++tx) a_off[tx] = (frow + tx) * S2D_PX + ((hsel ^ (((frow + tx) >> 3) & 1)) << 4);
    const int b_off = frow * S2D_PX + ((hsel ^ ((frow >> 3) & 1)) << 4);
    // taps ty = 0, 1 reach 2 / 1 rows up, ty = 3 one row down: rows of another image (or outside the batch) contribute nothing
    unsigned row_ok = 0;                       // bits 4 i + ty
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yimg = (r0 + 2 * wave + i) % p.H;
        row_ok |= ((yimg >= 2 ? 1u : 0u) | (yimg >= 1 ? 2u : 0u) | 4u | (yimg < p.H - 1 ? 8u : 0u)) << (4 * i);
    }

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
        const int ty = tap >> 2, tx = tap & 3;
        half8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const half8*>(b_lds + tap * (64 * S2D_PX) + j * 32 * S2D_PX + b_off);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!((row_ok >> (4 * i + ty)) & 1u)) continue;
            const half8 fa = *reinterpret_cast<const half8*>(a_lds + (2 * wave + i + ty) * (S2D_HW * S2D_PX) + a_off[tx]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue: two half tiles of 128 rows through LDS
    float* Cs = reinterpret_cast<float*>(smem);
    const int c8 = (tid & 7) * 8;
    float4v b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        b_lo = *reinterpret_cast<const float4v*>(p.bias + c8);
        b_hi = *reinterpret_cast<const float4v*>(p.bias + c8 + 4);
    }
    half_t* const outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if ((wave >> 1) == pass) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wave & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[row * S2D_CP + j * 32 + (lane & 31)] = acc[i][j][r];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rl = (tid >> 3) + e * 32;
            const int r = pass * 128 + rl;
            const int gr = r0 + (r >> 5), gx = x0 + (r & 31);
            if (gr < nrows && gx < p.W) {
                const float* csp = Cs + rl * S2D_CP + c8;
                float4v lo = *reinterpret_cast<const float4v*>(csp) + b_lo;
                float4v hi = *reinterpret_cast<const float4v*>(csp + 4) + b_hi;
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hi, half4);
                half8 hv = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (p.relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
                *reinterpret_cast<half8*>(outp + ((long)gr * p.W + gx) * p.ldc + c8) = hv;
            }
        }
    }
}

int launch_s2d(IgemmParams p, hipStream_t s) {
    const int tiles_x = ceil_div(p.W, TW);
    p.tiles_m = tiles_x * ceil_div((p.M / (p.H * p.W)) * p.H, TH);
    p.tiles_n = 1;
    hipLaunchKernelGGL(conv4x4_s2d_kernel, dim3(p.tiles_m), dim3(256), S2D_BYTES, s, p, tiles_x);
    LAUNCH_CHECK();
    return DVID_OK;
}

// ---- the stem and its max pool as one launch ------------------------------------------------------------------------------------------
// The stem writes 64 channels at HALF resolution (304 x 512 pixels per 608 x 1024 frame: 19.9 MB, the largest activation of the net) only
// for the 3x3 / stride-2 max pool to read it back and keep a quarter: 6.5 GB out + 6.5 GB in per 304-frame video, against 1.6 GB for the
// space-to-depth image going in and 1.6 GB for the pooled map going out.  Here a workgroup owns an 8 x 16 patch of POOLED pixels of one
// image: it computes the 17 x 33 stem pixels under it (conv rows 2 py0 - 1 .. 2 py0 + 15, columns 2 px0 - 1 .. 2 px0 + 31: the pool's
// one-pixel halo is recomputed, +12.5 % MFMA work on a layer that is far from the MFMA roof), rounds them to fp16 exactly as the stem's
// epilogue does (fp32 sum + bias, round, ReLU), parks them in LDS -- in the bytes the input halo and the weights occupied -- and pools
// from there.  Same products in the same order (tap by tap, 16 channels per MFMA; a tap outside the image multiplies zeros, which the
// layer-by-layer kernel skips: the same fp32 value), the same rounding, and max is exact: bit-identical to stem + max pool
// (tests/test_gpu_kernels.py::test_stem_pool_fusion_bit_identical).  Stem pixels outside the map enter the pool as 0, which is below or
// equal to every value behind a ReLU.
// M blocks of 32 stem pixels: block r < 17 = stem row r, columns 0 .. 31 of the patch; block 17 = column 32 of the 17 rows.  6 waves x 3
// blocks x 64 channels; products transposed (D[n][m]: a lane holds 4 consecutive channels of one pixel -> 8-byte LDS writes).
constexpr int SP_PW = 16, SP_CC = 2 * SP_PW + 1, SP_HW = 36;          // pooled patch width; stem columns under it (33); input halo pitch
// PH pooled rows per patch -> 2 PH + 1 stem rows (+ 1 block for column 32) over NW waves of BPW blocks:
//   PH 8: 18 blocks = 6 waves x 3, 72 KB of LDS (2 workgroups per CU), 12.5 % of the stem pixels computed twice
//   PH 4: 10 blocks = 5 waves x 2, 46 KB (3 per CU), 25 %
template <int PH>
struct StemPool {
    static constexpr int CR = 2 * PH + 1;                    // stem rows
    static constexpr int NBLK = CR + 1;
    static constexpr int BPW = PH == 8 ? 3 : 2;
    static constexpr int NW = NBLK / BPW;
    static constexpr int HH = CR + 3;                        // input halo rows
    static constexpr int A_PIECES = (HH * SP_HW + 31) / 32;  // 32 pixels of 32 bytes per 1-KiB piece
    static constexpr int A_IT = (A_PIECES + NW - 1) / NW, B_IT = (32 + NW - 1) / NW;
    static constexpr int A = A_IT * NW * 1024;
    static constexpr int C = CR * SP_CC * 128;               // fp16 stem pixels, 64 channels each
    static constexpr int kBytes = C > A + S2D_B ? C : A + S2D_B;
    static_assert(NW * BPW == NBLK, "blocks divide over the waves");
};

template <int PH>
__global__ __launch_bounds__(StemPool<PH>::NW * 64) void stem_pool_kernel(IgemmParams p, int tiles_x, int tiles_y, int hp, int wp) {
    using C = StemPool<PH>;
    constexpr int CR = C::CR, BPW = C::BPW, NW = C::NW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_img = tiles_x * tiles_y;
    const int lid = igemm_xcd_remap((int)blockIdx.x, p.tiles_m);
    const int img = lid / per_img, t = lid - img * per_img;
    const int py0 = (t / tiles_x) * PH, px0 = (t % tiles_x) * SP_PW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;          // stem pixel of patch position (0, 0)
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    char* const a_lds = smem;
    char* const b_lds = smem + C::A;
    const half_t* const in_img = p.in + (long)img * p.H * p.W * 16;

    // ---- prologue DMA: input halo (origin = stem pixel (0, 0) - 2), all 16 taps' weights
#pragma unroll
    for (int i = 0; i < C::A_IT; ++i) {
        const int q = wave + NW * i;
        const int pidx = 32 * q + (lane >> 1);
        const int hy = pidx / SP_HW, hx = pidx - hy * SP_HW;
        const int gy = cy0 - 2 + hy, gx = cx0 - 2 + hx;
        const bool ok = pidx < C::HH * SP_HW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int half = (lane & 1) ^ ((hx >> 3) & 1);
        glds16(ok ? reinterpret_cast<const char*>(in_img + ((long)gy * p.W + gx) * 16 + half * 8) : zero, a_lds + q * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::B_IT; ++i) {
        const int q = wave + NW * i;                         // piece: tap q >> 1, output channels 32 (q & 1) ..
        if (q < 32) {                                       // wave-uniform
            const int n = 32 * (q & 1) + (lane >> 1);
            const int half = (lane & 1) ^ ((n >> 3) & 1);
            glds16(reinterpret_cast<const char*>(p.w + (long)n * p.Kpad + (q >> 1) * 16 + half * 8), b_lds + q * 1024);
        }
    }

    // ---- this lane's stem pixel in each of the wave's blocks
    const int frow = lane & 31, hsel = lane >> 5;
    int pr[BPW], pc[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int blk = BPW * wave + i;
        pr[i] = blk < CR ? blk : (frow < CR ? frow : CR - 1);
        pc[i] = blk < CR ? frow : SP_CC - 1;
    }
    const int b_off = frow * S2D_PX + ((hsel ^ ((frow >> 3) & 1)) << 4);

    float16v acc[BPW][2];
#pragma unroll
    for (int i = 0; i < BPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
        const int ty = tap >> 2, tx = tap & 3;
        half8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const half8*>(b_lds + tap * (64 * S2D_PX) + j * 32 * S2D_PX + b_off);
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int hx = pc[i] + tx;
            const half8 fa = *reinterpret_cast<const half8*>(a_lds + ((pr[i] + ty) * SP_HW + hx) * S2D_PX + ((hsel ^ ((hx >> 3) & 1)) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa, acc[i][j], 0, 0, 0);      // D[channel][pixel]
        }
    }
    __syncthreads();                           // every wave is done with the halo and the weights: their bytes become the stem-pixel image

    // ---- stem epilogue into LDS: + bias, round to fp16, ReLU; pixel (r, c) at (r * 33 + c) * 128, 16-byte slot = channel group ^ key(pixel)
    char* const c_lds = smem;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int blk = BPW * wave + i;
        const bool lane_ok = blk < CR || frow < CR;                                // the last block has CR pixels
        const int gy = cy0 + pr[i], gx = cx0 + pc[i];
        const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;      // a stem pixel of the map (else it pools as 0)
        const int pix = pr[i] * SP_CC + pc[i];
        const int key = (pix >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ch = 32 * j + 8 * r4 + 4 * hsel;
                float4v v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                if (p.bias) v += *reinterpret_cast<const float4v*>(p.bias + ch);
                half4 hv = __builtin_convertvector(v, half4);
                hv = __builtin_elementwise_max(hv, half4{0, 0, 0, 0});
                if (!inside) hv = half4{0, 0, 0, 0};
                if (lane_ok) *reinterpret_cast<half4*>(c_lds + pix * 128 + (((ch >> 3) ^ key) << 4) + (ch & 4) * 2) = hv;
            }
    }
    __syncthreads();

    // ---- 3x3 / stride-2 max pool out of LDS: item = (pooled pixel, 8-channel group); 8 lanes write one pixel's 128 bytes
    half_t* const out_img = reinterpret_cast<half_t*>(p.out) + (long)img * hp * wp * 64;
    for (int it = tid; it < PH * SP_PW * 8; it += NT) {
        const int g = it & 7, q = it >> 3;
        const int qy = q / SP_PW, qx = q - qy * SP_PW;
        half8 best = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int pix = (2 * qy + dy) * SP_CC + 2 * qx + dx;
                const half8 x = *reinterpret_cast<const half8*>(c_lds + pix * 128 + ((g ^ ((pix >> 1) & 7)) << 4));
                best = __builtin_elementwise_max(best, x);
            }
        const int py = py0 + qy, px = px0 + qx;
        if (py < hp && px < wp) *reinterpret_cast<half8*>(out_img + ((long)py * wp + px) * 64 + g * 8) = best;
    }
}

template <int PH>
int launch_stem_pool_k(IgemmParams p, hipStream_t s) {
    using C = StemPool<PH>;
    static std::atomic<unsigned long long> attr_done{0};
    if (C::kBytes > 64 * 1024 && first_on_device(attr_done)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<PH>), hipFuncAttributeMaxDynamicSharedMemorySize, C::kBytes));
        mark_on_device(attr_done);
    }
    const int n = p.M / (p.H * p.W);
    const int hp = (p.H + 2 - 3) / 2 + 1, wp = (p.W + 2 - 3) / 2 + 1;
    const int tiles_x = (int)ceil_div(wp, SP_PW), tiles_y = (int)ceil_div(hp, PH);
    p.tiles_m = n * tiles_x * tiles_y;
    p.tiles_n = 1;
    hipLaunchKernelGGL(stem_pool_kernel<PH>, dim3(p.tiles_m), dim3(C::NW * 64), C::kBytes, s, p, tiles_x, tiles_y, hp, wp);
    LAUNCH_CHECK();
    return DVID_OK;
}
int launch_stem_pool(const IgemmParams& p, hipStream_t s) {
    return launch_stem_pool_k<8>(p, s);          // (a patch height of 4 gives the same values and measured no faster)
}

template <int BN, int WN>
int launch(IgemmParams p, hipStream_t s) {
    using C = Halo<BN, WN>;
    static std::atomic<unsigned long long> attr_done{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_done)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BN, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, C::kBytes));
        mark_on_device(attr_done);
    }
    const int tiles_x = ceil_div(p.W, TW), tiles_y = ceil_div((p.M / (p.H * p.W)) * p.H, TH);
    p.tiles_m = tiles_x * tiles_y;
    p.tiles_n = p.Cout / BN;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, WN>), dim3(p.tiles_m * p.tiles_n), dim3(512), C::kBytes, s, p, tiles_x, tiles_y);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

// Which launches take these kernels is decided by the layer and the image size alone -- never by a timing, and not by the number of
// images in the launch either: the chunked kernel differs from igemm2 in fp32 summation order, and a frame's features must not depend
// on how many frames share its launch (look-ahead groups, ragged video tails, streaming).  Rule: the layer type fits; the patch grid
// wastes at most 1/8 of its columns (W against the next multiple of 32); the map has at least 512 pixels.  (The 256- / 128-wide
// weight tile of the chunked kernel IS chosen by launch size: both sum in the same order.)  Measured, frames of 608 x 1024
// (ms, these kernels / igemm2; profiles/r02_conv3x3_halo.txt):
//   res2 conv2 (152 x 256, 64 ch)   8 frames 0.033 / 0.049                         104: 0.381 / 0.615
//   res3 conv2 (76 x 128, 128 ch)   8 frames 0.037 / 0.043    24: 0.084 / 0.126    104: 0.336 / 0.419
//   res4 conv2 (38 x 64, 256 ch)    8 frames 0.031 / 0.044    24: 0.069 / 0.081    104: 0.268 / 0.309     (22 layers)
//   res5 conv2 (19 x 32, 512 ch)    8 frames 0.048 / 0.048    24: 0.066 / 0.087    104: 0.260 / 0.292
//   FPN out p5 (19 x 32, 256 ch)    8 frames 0.026 / 0.020    24: 0.029 / 0.034    104: 0.071 / 0.093
//   FPN out p4 / p3                 as res4 / 8 frames 0.113 / 0.124, 104: 1.053 / 1.268
static bool s2d_stem_shape(const IgemmParams& p) {
    return p.KH == 4 && p.KW == 4 && p.stride == 1 && p.pad == 2 && p.Ho == p.H && p.Wo == p.W && p.Cin == 16 && p.Cout == 64 && p.Kpad == 256 &&
           p.res_mode == 0 && !p.out_f32 && p.splitk <= 1 && p.relu <= 1 && (p.ldc & 7) == 0 && p.H > 0 && p.W > 0 &&
           p.M == (p.M / (p.H * p.W)) * p.H * p.W;
}

bool dvid_conv3x3_halo_supported(const IgemmParams& p) {
    if (s2d_stem_shape(p)) return true;
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Ho == p.H && p.Wo == p.W && (p.Cin & 31) == 0 && p.Cin >= 64 &&
           p.Kpad == 9 * p.Cin && ((p.Cout & 127) == 0 || (p.Cout == 64 && p.Cin == 64)) && p.res_mode == 0 && !p.out_f32 && p.splitk <= 1 && p.relu <= 1 && (p.ldc & 7) == 0 &&
           p.H > 0 && p.W > 0 && p.M == (p.M / (p.H * p.W)) * p.H * p.W;
}

bool dvid_conv3x3_halo_preferred(const IgemmParams& p) {
    if (!dvid_conv3x3_halo_supported(p)) return false;
    return ceil_div(p.W, TW) * TW * 7 <= p.W * 8 && p.H * p.W >= 512;
}

int dvid_conv3x3_halo_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_conv3x3_halo_supported(p)) return DVID_ERR_UNSUPPORTED;
    if (s2d_stem_shape(p)) return launch_s2d(p, s);
    if (p.Cout == 64) return launch_c64(p, s);
    // 256-wide or 128-wide weight tiles: the same K order, bit-identical results -- so this choice may look at the launch size.  With
    // fewer 256-wide workgroups than half the CUs (8 frames of 608 x 1024 at res4: 76) the narrower tile doubles the workgroups.
    const long patches = (long)ceil_div(p.W, TW) * ceil_div(p.M / p.W, TH);
    return ((p.Cout & 255) == 0 && patches * (p.Cout >> 8) > 128) ? launch<256, 4>(p, s) : launch<128, 2>(p, s);
}

// The space-to-depth stem (bias + ReLU) and the 3x3 / stride-2 / pad-1 max pool behind it as one launch: `p` = the stem's parameters with
// `out` = the POOLED map [n][(H + 1) / 2][(W + 1) / 2][64].  Bit-identical to dvid_conv3x3_halo_launch + dvid_maxpool3x3s2_launch.
bool dvid_stem_pool_supported(const IgemmParams& p) { return s2d_stem_shape(p) && p.relu == 1 && p.ldc == 64; }          // the pooled map is written densely
int dvid_stem_pool_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_stem_pool_supported(p)) return DVID_ERR_UNSUPPORTED;
    return launch_stem_pool(p, s);
}
