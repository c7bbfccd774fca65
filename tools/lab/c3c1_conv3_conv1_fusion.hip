// Back-to-back fusion of a bottleneck's last 1x1 convolution with the NEXT bottleneck's first one:
//
//   Y = relu(A . W3^T + b3 + R)        conv3 (+ FrozenBN folded) + residual + ReLU   [M, N1]   stored (the next block's residual)
//   Z = relu(Y . W1^T + b1)            next block's conv1 (+ FrozenBN folded) + ReLU  [M, N2]   stored
//
// (detectron2 BottleneckBlock; the two launches this replaces are csrc/model.hip's conv_run(blk.c3)
// and conv_run(next.c1).)  Both layers are short-K 1x1 convolutions whose time is HBM traffic: conv3 writes Y and conv1
// reads it straight back (for res4: 120 of the pair's 420 MB per 24 frames).  Here a workgroup owns 64 rows (pixels): the
// Y chunk it has just produced is rounded to fp16 -- exactly what the unfused path stores -- kept in LDS, and used at once
// as the A operand of the second product, so Y is written once and never re-read.
//
// Structure (1024 threads = 16 waves = two independent 64-row halves of 8 waves, one workgroup per CU):
//   * A[128, K1] is loaded into LDS once (pitch K1 + 8 halves: conflict-free ds_read_b128 fragments).
//   * N1 is walked in chunks of up to 256 columns.  Per chunk every wave computes 64 rows x 32 columns of the first
//     product, finishes them in registers (bias + residual + ReLU; the MFMA accumulator layout gives each lane one column
//     and 16 rows per 32x32 tile, i.e. 64-byte runs per row across a half-wave for the residual loads and the Y stores),
//     writes the fp16 values to global memory and into the LDS chunk buffer; after a barrier the chunk is one K slice of
//     the second product, accumulated across chunks in K-ascending order.
//   * The weights are never staged through LDS: they are re-packed at load time into MFMA B-fragment order (one
//     contiguous 1-KiB block per 32 x 16 tile), so a wave fetches a fragment with one fully coalesced 16-byte-per-lane
//     load from L2; the two halves fetch the same fragments at about the same time, so the second fetch hits L1.
//     No barrier inside either K loop.
// Measured (profiles/r02_c3c1_fusion.txt): no faster than the two tuned igemm2 launches it replaces -- off by default.
// Sums run in the same order as igemm2's (K ascending, 16 per MFMA, fp32 accumulate, one fp16 rounding after bias +
// residual), so Y and Z are bit-identical to the two separate launches (tests/test_gpu_kernels.py).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int MH = 2;          // 64-row halves per workgroup: both halves fetch the same weight fragments at about the same time, so the
                               // second fetch hits the CU's L1 instead of crossing the L2 fabric again (the kernel is bound by that traffic)
constexpr int BM = 64 * MH;
constexpr int NW = 8;          // waves per half; 16 waves per workgroup, one workgroup per CU = 4 waves per SIMD

template <int K1, int N1, int N2>
struct C3C1Cfg {
    static constexpr int NC = N1 < 256 ? N1 : 256;        // chunk of the first product's columns = K slice of the second
    static constexpr int NCH = N1 / NC;
    static constexpr int TN1 = NC / (32 * NW);            // 32-column tiles per wave in the first product (NW waves side by side)
    static constexpr int TM2 = N2 >= 256 ? 2 : 1;         // second product: 2 x (N2 / 32) tiles over the NW waves
    static constexpr int TN2 = N2 >= 256 ? N2 / 256 : 1;
    static constexpr int A_PITCH = K1 + 8, Y_PITCH = NC + 8;
    static constexpr int kSmem = (BM * A_PITCH + BM * Y_PITCH) * 2;
};

template <int K1, int N1, int N2, int ABL = 0>
__global__ __launch_bounds__(64 * NW * MH, 4) void c3c1_kernel(C3C1Params p) {
    using Cfg = C3C1Cfg<K1, N1, N2>;
    constexpr int NC = Cfg::NC, NCH = Cfg::NCH, TN1 = Cfg::TN1, TM2 = Cfg::TM2, TN2 = Cfg::TN2;
    constexpr int AP = Cfg::A_PITCH, YP = Cfg::Y_PITCH;
    constexpr int KS1 = K1 / 16, KS2 = NC / 16;
    constexpr int PF = 4;              // prefetch depth of the weight fragments (register budget: 256 / lane)
    static_assert(NC % (32 * NW) == 0 && N1 % NC == 0 && K1 % 64 == 0 && N2 % 64 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);
    half_t* Ys = As + BM * AP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % NW;              // column role inside the half
    const int rb = (wave_all / NW) * 64;         // first row of this wave's half inside the tile
    const int l31 = lane & 31, lhi = lane >> 5;
    const long m0 = (long)blockIdx.x * BM;

    // ---- A tile -> LDS (rows beyond M read row M - 1; their results are never stored) ----
    {
        constexpr int VPR = K1 / 8;                       // 16-byte vectors per row
        for (int i = tid; i < BM * VPR; i += 64 * NW * MH) {
            const int r = i / VPR, v = i % VPR;
            long m = m0 + r;
            if (m >= p.M) m = p.M - 1;
            *reinterpret_cast<half8*>(As + r * AP + v * 8) = *reinterpret_cast<const half8*>(p.a + m * K1 + v * 8);
        }
    }
    // second product's tile coordinates of this wave
    // N2 >= 256: every wave both row tiles of its own column tile(s); N2 = 128: one tile per wave; N2 = 64: waves 0-3 only
    const bool has2 = N2 >= 128 || wave < 4;
    const int mt2_0 = (N2 >= 256) ? 0 : (N2 == 128 ? (wave >> 2) : ((wave >> 1) & 1));
    const int nt2_0 = (N2 >= 256) ? wave * TN2 : (N2 == 128 ? (wave & 3) : (wave & 1));
    float16v acc2[TM2][TN2];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    __syncthreads();

    const half8* __restrict__ w3f = reinterpret_cast<const half8*>(p.w3f);
    const half8* __restrict__ w1f = reinterpret_cast<const half8*>(p.w1f);
    half_t* __restrict__ zg = p.z;
    const float* __restrict__ b3g = p.b3;
    const int rows_here = (p.M - m0) < BM ? (int)(p.M - m0) : BM;       // valid rows of this tile (>= 1)

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        // ================= first product: rows 0..63 x columns [c*NC + wave*TN1*32, +TN1*32) =================
        float16v acc1[2][TN1];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN1; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
        const int nt1 = c * (NC / 32) + wave * TN1;       // first 32-column tile (global index) of this wave
        {
            constexpr int D = KS1 < PF ? KS1 : PF;        // weight fragments in flight
            half8 bq[D][TN1];
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int j = 0; j < TN1; ++j) bq[d][j] = w3f[((long)(nt1 + j) * KS1 + d) * 64 + lane];
#pragma unroll 1
            for (int kg = 0; kg < ((ABL & 8) ? 0 : KS1); kg += D) {         // rolled: bounds the scheduler's look-ahead (registers) and the code size
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int ks = kg + d;
                    half8 fa[2], fb[TN1];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const half8*>(As + (rb + i * 32 + l31) * AP + ks * 16 + lhi * 8);
#pragma unroll
                    for (int j = 0; j < TN1; ++j) fb[j] = bq[d][j];
                    if (ks + D < KS1) {
#pragma unroll
                        for (int j = 0; j < TN1; ++j) bq[d][j] = w3f[((long)(nt1 + j) * KS1 + ks + D) * 64 + lane];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN1; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc1[i][j], 0, 0, 0);
                }
            }
        }
        if (c > 0) __syncthreads();                       // every wave is done reading the previous chunk from Ys
        // ---- epilogue 1: + bias + residual, one fp16 rounding, ReLU; -> Y (global) and Ys (LDS) ----
        // The 16 residual loads of a 32 x 32 tile are issued before its first store (the loads are independent; behind a
        // store the compiler would have to assume aliasing and serialise them, one HBM latency each).  Addresses are a
        // wave-uniform base + a 32-bit lane offset, so they cost one VGPR each instead of a 64-bit pair.
        const half_t* rbase = p.r + m0 * N1 + c * NC;
        half_t* ybase = p.y + m0 * N1 + c * NC;
#pragma unroll
        for (int j = 0; j < TN1; ++j) {
            const int coln = wave * (TN1 * 32) + j * 32 + l31;       // column inside the chunk
            const float bias = b3g[c * NC + coln];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // lane offsets are re-derived per tile from a value the optimiser cannot see through: otherwise it hoists all
                // 128 element offsets of the epilogue out of the chunk loop and spills the accumulators to make room
                unsigned lane_off = (unsigned)((rb + i * 32 + 4 * lhi) * N1 + coln);
                int row0 = rb + i * 32 + 4 * lhi;
                asm volatile("" : "+v"(lane_off), "+v"(row0));
                half_t rv[16];
                bool ok[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int dr = (r & 3) + 8 * (r >> 2);
                    ok[r] = row0 + dr < rows_here;
                    if (ABL & 1) rv[r] = (half_t)0.f; else
                    // unconditional load from a clamped offset (row 0 of the tile is always valid): a conditional load would
                    // become a branch with its own s_waitcnt, i.e. one exposed HBM latency per element
                    rv[r] = rbase[ok[r] ? lane_off + (unsigned)(dr * N1) : (unsigned)coln];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    const float v = (acc1[i][j][r] + bias) + (float)rv[r];
                    half_t y = (half_t)v;
                    y = y > (half_t)0.f ? y : (half_t)0.f;
                    if (!(ABL & 2) && ok[r]) ybase[lane_off + (unsigned)(dr * N1)] = y;
                    Ys[(rb + i * 32 + 4 * lhi + dr) * YP + coln] = y;
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the next tile's loads from being hoisted over this one (registers)
            }
        }
        // first weight fragments of the second product: in flight across the barrier
        constexpr int D2 = KS2 < PF ? KS2 : PF;
        half8 bq2[D2][TN2];
#pragma unroll
        for (int d = 0; d < D2; ++d)
#pragma unroll
            for (int j = 0; j < TN2; ++j) bq2[d][j] = w1f[((long)(nt2_0 + j) * (N1 / 16) + c * KS2 + d) * 64 + lane];
        __syncthreads();
        // ================= second product: K slice = this chunk =================
#pragma unroll 1
        for (int kg = 0; kg < (((ABL & 4) || !has2) ? 0 : KS2); kg += D2) {
#pragma unroll
            for (int d = 0; d < D2; ++d) {
                const int ks = kg + d;
                half8 fa[TM2], fb[TN2];
#pragma unroll
                for (int i = 0; i < TM2; ++i) fa[i] = *reinterpret_cast<const half8*>(Ys + (rb + (mt2_0 + i) * 32 + l31) * YP + ks * 16 + lhi * 8);
#pragma unroll
                for (int j = 0; j < TN2; ++j) fb[j] = bq2[d][j];
                if (ks + D2 < KS2) {
#pragma unroll
                    for (int j = 0; j < TN2; ++j) bq2[d][j] = w1f[((long)(nt2_0 + j) * (N1 / 16) + c * KS2 + ks + D2) * 64 + lane];
                }
#pragma unroll
                for (int i = 0; i < TM2; ++i)
#pragma unroll
                    for (int j = 0; j < TN2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc2[i][j], 0, 0, 0);
            }
        }
    }
    // ---- epilogue 2: + bias, fp16, ReLU -> Z ----
    half_t* __restrict__ zbase = zg + m0 * N2;
    if (has2)
#pragma unroll
    for (int j = 0; j < TN2; ++j) {
        const int n = (nt2_0 + j) * 32 + l31;
        const float bias = p.b1[n];
#pragma unroll
        for (int i = 0; i < TM2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (mt2_0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < rows_here) {
                    half_t z = (half_t)(acc2[i][j][r] + bias);
                    zbase[(unsigned)(row * N2 + n)] = z > (half_t)0.f ? z : (half_t)0.f;
                }
            }
    }
}

template <int K1, int N1, int N2, int ABL = 0>
int launch(const C3C1Params& p, hipStream_t s) {
    constexpr int smem = C3C1Cfg<K1, N1, N2>::kSmem;
    static bool attr_set = false;
    if (smem > 64 * 1024 && !attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&c3c1_kernel<K1, N1, N2, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((c3c1_kernel<K1, N1, N2, ABL>), dim3((unsigned)((p.M + BM - 1) / BM)), dim3(64 * NW * MH), smem, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

bool dvid_c3c1_supported(int k1, int n1, int n2) {
    return (k1 == 64 && n1 == 256 && (n2 == 64 || n2 == 128)) || (k1 == 128 && n1 == 512 && (n2 == 128 || n2 == 256)) ||
           (k1 == 256 && n1 == 1024 && n2 == 256);
}

int dvid_c3c1_launch(const C3C1Params& p, int k1, int n1, int n2, hipStream_t s) {
    if (p.M <= 0) return DVID_OK;
    if (k1 == 64 && n1 == 256 && n2 == 64) return launch<64, 256, 64>(p, s);
    if (k1 == 64 && n1 == 256 && n2 == 128) return launch<64, 256, 128>(p, s);
    if (k1 == 128 && n1 == 512 && n2 == 128) return launch<128, 512, 128>(p, s);
    if (k1 == 128 && n1 == 512 && n2 == 256) return launch<128, 512, 256>(p, s);
    if (k1 == 256 && n1 == 1024 && n2 == 256) {
        static const int abl = getenv("DVID_C3C1_ABLATE") ? atoi(getenv("DVID_C3C1_ABLATE")) : 0;       // measurement only
        switch (abl) {
            case 1: return launch<256, 1024, 256, 1>(p, s);
            case 2: return launch<256, 1024, 256, 2>(p, s);
            case 3: return launch<256, 1024, 256, 3>(p, s);
            case 4: return launch<256, 1024, 256, 4>(p, s);
            case 8: return launch<256, 1024, 256, 8>(p, s);
            case 12: return launch<256, 1024, 256, 12>(p, s);
            case 15: return launch<256, 1024, 256, 15>(p, s);
            default: break;
        }
        return launch<256, 1024, 256>(p, s);
    }
    return DVID_ERR_UNSUPPORTED;
}
