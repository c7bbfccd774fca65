// Implicit-GEMM convolution / linear layer: direct-to-LDS staging.
//
// out = act(A_im2col . Wt^T + bias + residual), fp16 in, fp32 accumulate on v_mfma_f32_32x32x16_f16 (IgemmParams,
// kernels.h).  The first, register-staged version (global -> VGPR -> LDS, 214 TFLOP/s on this path) is in the history
// of this file's predecessor igemm.hip; this one is built around
// `global_load_lds_dwordx4`: every lane hands the DMA its own 16-byte source address -- the im2col
// gather, filter-tap bounds and the M/N tails are all resolved in that address (out-of-range lanes
// point at a 16-byte zero page) -- and the data lands in LDS without touching VGPRs.  The LDS image
// of a tile is lane-linear (64-byte rows for BK = 32), so bank conflicts are removed by an XOR
// swizzle applied to the SOURCE chunk index and again on the fragment read:
// physical 16-byte chunk = logical chunk ^ ((row >> 2) & 3), conflict-free for ds_read_b128.
// Two LDS stages of (BM+BN)*64 B and a half-tile fp32 epilogue buffer keep a workgroup at <= 33 KB,
// so 4 workgroups (16 waves) share a CU and hide the DMA latency of each other's K steps.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "options.h"

namespace {

// K tile = BKT halves.  BKT = 64 stages full 128-byte lines per row (8 rows per 1-KiB DMA piece, swizzle
// key (row >> 1) & 7); BKT = 32 stages 64-byte half lines (16 rows per piece, key (row >> 2) & 3) at half the
// LDS footprint.  Both keys make the four 16-lane groups of ds_read_b128 conflict-free.

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

template <int BM, int BN, int BKT, int NSTAGE>
struct Smem2 {
    static constexpr int kStage = (BM + BN) * BKT * 2;
    static constexpr int kCPitch = BN + 4;
    static constexpr int kCHalf = (BM / 2) * kCPitch * 4;
    static constexpr int kRing = NSTAGE == 5 ? 4 : NSTAGE;        // NSTAGE 5 = the anti-phase schedule over a 4-stage ring
    static constexpr int kBytes = (kRing * kStage > kCHalf) ? kRing * kStage : kCHalf;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// NSTAGE 2: one DMA tile in flight, `__syncthreads()` per K step (the compiler drains vmcnt there).
// NSTAGE 3, 4: ring of stages, NSTAGE - 1 DMA tiles in flight across a raw `s_barrier`; each wave waits with a
//           COUNTED `s_waitcnt vmcnt((NSTAGE - 2) * P)` (P = its DMA pieces per tile) so only the tile about to be
//           read has landed, and re-fills the stage freed by the previous step right after the barrier.
// NSTAGE 5: anti-phase schedule (8 waves, 4-stage ring of BK 32 tiles).  The two wave rows (wm = 0 / 1) alternate between a
//           read slot R(t) -- the 12 fragment reads of tile t, wait for them and for the own DMA pieces of tile t + 1 --
//           and an MFMA slot M(t) -- 16 MFMAs with the 4 DMA pieces of tile t + 3 issued between them (a DMA piece
//           costs its wave ~100 issue cycles; in front of the MFMAs or inside the read slot it costs 15 % throughput,
//           tools/lab/gemm_pingpong.hip) -- one barrier per slot, row 1 one slot behind row 0, so each SIMD has one
//           wave in its MFMA slot while the other reads.  Tile t + 3 re-uses the stage of tile t - 1, whose last reads
//           were retired (lgkmcnt(0)) in front of the barrier that ended the previous slot.
// WN: waves along N (2 -> 4 waves / 256 threads, 4 -> 8 waves / 512 threads); always 2 waves along M.
// AK: how the A operand is addressed -- 0: filter-tap walk (KHxKW convolutions), 1: the Cin = 8 stem (one tap per
//     16-byte chunk), 2: FLAT = 1x1 / linear with pad 0: a row of A is K contiguous halves, no taps, no masks (stride 1
//     with Ho == H, Wo == W additionally needs no division: row m is pixel m).
// EPI: epilogue specialisation -- 1: fp16 out, residual none / same-shape fp16, ReLU or none (the backbone); 2: fp32 out,
//     residual none / same-shape fp32 (possibly in place), ReLU or none (Swin's residual stream, decoder linears);
//     3: fp16 out, exact GELU (Swin fc1); all three need a vector-aligned N.  0: the general one (igemm_store_row8:
//     upsampled residual, ragged N, mixed types), which inlines to ~7000 instructions per kernel.  Short-K layers are instruction-issue bound (23 scalar/vector instructions per MFMA,
//     profiles/r01_pmc_conv3.txt), so both specialisations exist to cut executed instructions, not bytes.
template <int BM, int BN, int BKT, int NSTAGE, int WN, int AK, int EPI>
__global__ __launch_bounds__(128 * WN) void igemm2_kernel(IgemmParams p) {
    constexpr bool SMALLC = AK == 1, FLAT = AK == 2;
    constexpr int NW = 2 * WN;                      // waves per workgroup
    constexpr int NT = 64 * NW;                     // threads
    constexpr int ROW_BYTES = BKT * 2;
    constexpr int CHUNKS = BKT / 8;                 // 16-byte chunks per row
    constexpr int RPP = 1024 / ROW_BYTES;           // tile rows per 1-KiB DMA piece
    constexpr int KEY_SHIFT = (BKT == 64) ? 1 : 2;  // swizzle key = (row >> KEY_SHIFT) & (CHUNKS - 1)
    constexpr int TM = BM / 64, TN = BN / (32 * WN);   // 32x32 MFMA tiles per wave (waves 2 x WN)
    constexpr int A_IT = BM / RPP / NW, B_IT = BN / RPP / NW;   // DMA pieces per wave per K tile
    static_assert(NSTAGE >= 2 && NSTAGE <= 5, "ring depth / schedule code");
    static_assert(NSTAGE != 5 || (WN == 4 && AK != 1), "the anti-phase schedule is written for 8 waves");
    static_assert(A_IT >= 1 && B_IT >= 1 && A_IT * RPP * NW == BM && B_IT * RPP * NW == BN, "tile / wave-count mismatch");
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int STAGE = Smem2<BM, BN, BKT, NSTAGE>::kStage;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int split = p.splitk > 1 ? (int)blockIdx.x / ntiles : 0;       // wave-uniform
    const int lid = igemm_xcd_remap((int)blockIdx.x - split * ntiles, ntiles);
    const int tile_n = lid % p.tiles_n, tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- DMA descriptors: piece j = wave + 4*i covers tile rows [16j, 16j+16), lane -> (row, chunk)
    const int lrow = lane / CHUNKS;
    const int pch = lane % CHUNKS;                              // physical chunk this lane fills
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    // Per DMA piece the lane keeps ONE precomputed source pointer (tap (0,0), its own chunk) and a bitmask of the
    // filter taps that fall inside the image for its output pixel; a K step then costs a bit test, a 64-bit add of
    // a wave-uniform byte offset and a select against the zero page (the per-step bounds/address arithmetic of the
    // first version was ~10 VALU per MFMA and made the waves issue-bound, profiles/r01_pmc_res4.txt).
    const char* a_ptr[A_IT];
    unsigned a_mask[A_IT];
    int a_iy[A_IT], a_ix[A_IT], a_lch[A_IT];       // only used by the SMALLC (stem) path
    int a_ty[A_IT], a_tx[A_IT];                    // SMALLC: filter tap of this lane's chunk in the current K tile
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = RPP * (wave + NW * i) + lrow;
        a_lch[i] = pch ^ ((row >> KEY_SHIFT) & (CHUNKS - 1));   // logical chunk stored at this position
        const int m = m0 + row;
        a_mask[i] = 0u;
        a_ptr[i] = zero;
        a_iy[i] = -(1 << 28);
        a_ix[i] = 0;
        // SMALLC: Cin = 8 or 16 (one or two 16-byte chunks per filter tap); this lane's chunk belongs to tap lch / cpt
        const int cpt = SMALLC ? (p.Cin >> 3) : 1;
        a_ty[i] = SMALLC ? (a_lch[i] / cpt) / p.KW : 0;
        a_tx[i] = SMALLC ? (a_lch[i] / cpt) - a_ty[i] * p.KW : 0;
        if (FLAT) {
            if (m < p.M) {
                long pix = m;
                if (p.stride != 1) {
                    const int ox = m % p.Wo;
                    const int t = m / p.Wo;
                    const int oy = t % p.Ho;
                    const int img = t / p.Ho;
                    pix = ((long)img * p.H + oy * p.stride) * p.W + ox * p.stride;
                }
                a_ptr[i] = reinterpret_cast<const char*>(p.in + pix * p.Cin + a_lch[i] * 8);
                a_mask[i] = BKT * 2;              // FLAT: per-step pointer increment (0 keeps padded rows on the zero page)
            }
        } else if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int img = t / p.Ho;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_iy[i] = iy0;
            a_ix[i] = ix0;
            a_ptr[i] = reinterpret_cast<const char*>(p.in + ((long)(img * p.H + iy0) * p.W + ix0) * p.Cin + (SMALLC ? (a_lch[i] % cpt) * 8 : a_lch[i] * 8));
            if (!SMALLC) {
                for (int t2 = 0; t2 < p.ntaps; ++t2) {
                    const int ty = t2 / p.KW, tx = t2 - ty * p.KW;
                    if ((unsigned)(iy0 + ty) < (unsigned)p.H && (unsigned)(ix0 + tx) < (unsigned)p.W) a_mask[i] |= 1u << t2;
                }
            }
        }
    }
    const char* b_ptr[B_IT];
    int b_step[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = RPP * (wave + NW * i) + lrow;
        const int n = n0 + row;
        const bool ok = n < p.Cout;
        b_ptr[i] = ok ? reinterpret_cast<const char*>(p.w + (long)n * p.Kpad + (pch ^ ((row >> KEY_SHIFT) & (CHUNKS - 1))) * 8) : zero;
        b_step[i] = ok ? BKT * 2 : 0;
    }

    // K range of this workgroup (split-K) and the filter-tap walk of the K loop (wave-uniform): k0 = tap*Cin + c0
    const int nk_all = p.Kpad / BKT;
    const int nk = p.splitk > 1 ? nk_all / p.splitk : nk_all;
    const int kt0 = split * nk;
    int ky = 0, kx = 0, c0 = 0, tap = 0;
    if (AK == 0 && kt0) {
        tap = (kt0 * BKT) / p.Cin;
        c0 = kt0 * BKT - tap * p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
    }
    if (FLAT) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a_ptr[i] += (long)kt0 * a_mask[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_ptr[i] += (long)kt0 * b_step[i];

    // source address of piece i of the NEXT K tile to be fetched (pieces [0, A_IT) are A rows, [A_IT, A_IT + B_IT) are B rows);
    // advances that piece's own walk state
    auto piece_src = [&](int i) -> const char* {
        if (i < A_IT) {
            if (SMALLC) {                                        // Cin == 8: one tap per 16-byte chunk
                // this lane's tap (a_ty, a_tx) walks the filter CHUNKS taps per K step -- no division in the loop
                const int tky = a_ty[i], tkx = a_tx[i];
                const int iy = a_iy[i] + tky, ix = a_ix[i] + tkx;
                const bool ok = tky < p.KH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const char* src = ok ? a_ptr[i] + (long)((tky * p.W + tkx) * p.Cin) * 2 : zero;
                a_tx[i] += CHUNKS / (p.Cin >> 3);
                while (a_tx[i] >= p.KW) {
                    a_tx[i] -= p.KW;
                    ++a_ty[i];
                }
                return src;
            } else if (FLAT) {
                const char* src = a_ptr[i];
                a_ptr[i] += a_mask[i];
                return src;
            } else {
                const long koff = (long)((ky * p.W + kx) * p.Cin + c0) * 2;      // wave-uniform byte offset of this K tile
                return ((a_mask[i] >> tap) & 1u) ? a_ptr[i] + koff : zero;
            }
        } else {
            const int j = i - A_IT;
            const char* src = b_ptr[j];
            b_ptr[j] += b_step[j];
            return src;
        }
    };
    // LDS destination (wave base; the DMA and the register path both place lane l's 16 bytes at base + 16 l)
    auto piece_dst = [&](int stage, int i) -> char* {
        char* sa = smem + stage * STAGE;
        return i < A_IT ? sa + (wave + NW * i) * 1024 : sa + A_BYTES + (wave + NW * (i - A_IT)) * 1024;
    };
    // one DMA piece of K tile `kt` into `stage`
    auto issue_piece = [&](int kt, int stage, int i) { glds16(piece_src(i), piece_dst(stage, i)); };
    // advance the filter-tap walk to the next K tile (after the last piece of a tile has been issued)
    auto advance = [&]() {
        if (AK == 0) {
            c0 += BKT;
            if (c0 >= p.Cin) {
                c0 = 0;
                ++tap;
                if (++kx == p.KW) {
                    kx = 0;
                    ++ky;
                }
            }
        }
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int i = 0; i < A_IT + B_IT; ++i) issue_piece(kt, stage, i);
        advance();
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- residual prefetch: issued before the K loop so its HBM latency hides under it -------------
    constexpr int VPR = BN / 8;
    constexpr int ERPP = NT / VPR;
    constexpr int EROWS = (BM / 2) / ERPP;
    const int c8 = (tid % VPR) * 8;
    const int n = n0 + c8;
    constexpr bool PRE = BM * BN <= 128 * 128;      // larger tiles need the registers for accumulators (and lose anyway: 128x256 with the prefetch 453 vs 317 us on res4 conv3)
    const bool pre_ok = PRE && p.res_mode == 1 && !p.res_f32 && (p.Cout & 7) == 0;
    half8 rpre[2][PRE ? EROWS : 1];
    if (pre_ok) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int e = 0; e < EROWS; ++e) {
                const int m = m0 + half * (BM / 2) + tid / VPR + e * ERPP;
                const bool ok = m < p.M && n < p.Cout;
                rpre[half][PRE ? e : 0] = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.res) + (ok ? (long)m * p.Cout + n : 0));
            }
    }

    // fragment addressing: row = base32 + (lane & 31); logical chunk = 2*ks + (lane >> 5)
    constexpr int KS = BKT / 16;
    const int frow = lane & 31;
    const int sw = (frow >> KEY_SHIFT) & (CHUNKS - 1);
    const int fa_off = (wm * (BM / 2) + frow) * ROW_BYTES;
    const int fb_off = A_BYTES + (wn * (BN / WN) + frow) * ROW_BYTES;
    int choff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) choff[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;

    // Fragment reads run one K sub-step ahead of the MFMAs that consume them (two register sets), so that the LDS
    // latency of sub-step ks+1 is covered by the MFMAs of sub-step ks (one wave per SIMD has nobody else to hide it).
    auto load_frags = [&](const char* st, int ks, half8 (&fa)[TM], half8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const half8*>(st + fa_off + i * 32 * ROW_BYTES + choff[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const half8*>(st + fb_off + j * 32 * ROW_BYTES + choff[ks]);
    };
    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE;
        half8 fa[2][TM], fb[2][TN];
        load_frags(st, 0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) load_frags(st, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[ks & 1][j], fa[ks & 1][i], acc[i][j], 0, 0, 0);      // transposed: see the epilogue
        }
        // pin the issue order the source spells out (hipcc otherwise folds the two register sets into one and waits
        // for every read right before its MFMAs): reads(0), then per sub-step reads(ks+1) ahead of MFMAs(ks)
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
    };

    if constexpr (NSTAGE == 5) {
        constexpr int P = A_IT + B_IT, NM = TM * TN * KS;
        static_assert(NSTAGE != 5 || NM % P == 0, "DMA pieces are spread evenly over the MFMAs of a tile");
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < nk) issue(kt0 + d, d);
        if (nk > 2) wait_vmcnt<2 * P>(); else if (nk > 1) wait_vmcnt<P>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();                 // wave row 1 runs one slot behind row 0
        for (int t = 0; t < nk; ++t) {
            // ---- R(t)
            const char* st = smem + (t & 3) * STAGE;
            half8 fa[TM][KS], fb[TN][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][ks] = *reinterpret_cast<const half8*>(st + fa_off + i * 32 * ROW_BYTES + choff[ks]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j][ks] = *reinterpret_cast<const half8*>(st + fb_off + j * 32 * ROW_BYTES + choff[ks]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // issued so far: tiles <= t + 2; the own pieces of tile t + 1 must have landed
            if (t + 2 < nk) wait_vmcnt<P>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- M(t)
            const bool dma = t + 3 < nk;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int ks = q / (TM * TN), i = (q / TN) % TM, j = q % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
                if (dma && q % (NM / P) == 1) issue_piece(kt0 + t + 3, (t + 3) & 3, q / (NM / P));
            }
            if (dma) advance();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();                 // balances the extra barrier of row 1
        __syncthreads();
    } else if constexpr (NSTAGE == 2) {
        issue(kt0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(kt0 + kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();   // drains this wave's DMA (vmcnt) and publishes the next stage
        }
    } else {
        constexpr int P = A_IT + B_IT;          // DMA instructions per wave per K tile
        constexpr int AHEAD = NSTAGE - 1;       // K tiles issued before the first one is consumed
#pragma unroll
        for (int d = 0; d < AHEAD; ++d)
            if (d < nk) issue(kt0 + d, d);
        int cs = 0, is = AHEAD;                 // stage being computed / stage to refill
        for (int kt = 0; kt < nk; ++kt) {
            // tiles still allowed in flight while tile kt is read: the AHEAD - 1 issued after it (fewer at the tail)
            const int later = nk - 1 - kt;
            if (later >= AHEAD - 1) wait_vmcnt<(AHEAD - 1) * P>();
            else if (AHEAD > 2 && later == 1) wait_vmcnt<P>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();       // tile kt visible to all waves; stage `is` no longer read
            if (kt + AHEAD < nk) issue(kt0 + kt + AHEAD, is);
            compute(cs);
            cs = (cs == NSTAGE - 1) ? 0 : cs + 1;
            is = (is == NSTAGE - 1) ? 0 : is + 1;
        }
        __syncthreads();                        // all waves done with the ring before it becomes the C buffer
    }

    // ---- epilogue: two half tiles (rows of wm = 0, then wm = 1) through an fp32 LDS buffer ---------
    constexpr int CP = Smem2<BM, BN, BKT, NSTAGE>::kCPitch;
    float* Cs = reinterpret_cast<float*>(smem);
    if (p.splitk > 1) p.out = reinterpret_cast<float*>(p.out) + split * p.split_stride;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (p.bias && n + e < p.Cout) ? p.bias[n + e] : 0.f;
    // lean epilogue state: element offsets of this thread's first row of a half tile, advanced by constant strides
    const bool has_res = p.res_mode == 1;
    half_t* const outp = reinterpret_cast<half_t*>(p.out);
    const half_t* const resp = reinterpret_cast<const half_t*>(p.res);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if (wm == half) {
            // The products are computed transposed (the weight fragment is the MFMA's first operand: D[n][m], same sums bit for bit),
            // so a lane holds 4 consecutive channels of ONE tile row per register quad: 16-byte LDS writes, 4 per accumulator tile
            // instead of 16 scalar ones (the short-K layers are bound by their instruction count).
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int row = i * 32 + (lane & 31);
                        const int col = wn * (BN / WN) + j * 32 + 8 * r4 + 4 * (lane >> 5);
                        *reinterpret_cast<float4v*>(Cs + row * CP + col) =
                            (float4v){acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                    }
        }
        __syncthreads();
        if (EPI != 0) {
            const int mrow = m0 + half * (BM / 2) + tid / VPR;
            long oidx = (long)mrow * p.ldc + n;
            long ridx = (long)mrow * p.Cout + n;
            const float* csp = Cs + (tid / VPR) * CP + c8;
            const float4v b_lo = {bias8[0], bias8[1], bias8[2], bias8[3]}, b_hi = {bias8[4], bias8[5], bias8[6], bias8[7]};
            if (n < p.Cout) {
#pragma unroll
                for (int e = 0; e < EROWS; ++e) {
                    if (mrow + e * ERPP < p.M) {
                        // vector forms: the compiler emits packed fp32 adds and packed fp16 max for these
                        float4v lo = *reinterpret_cast<const float4v*>(csp + e * ERPP * CP) + b_lo;
                        float4v hi = *reinterpret_cast<const float4v*>(csp + e * ERPP * CP + 4) + b_hi;
                        if (EPI == 2) {
                            if (has_res) {
                                lo += *reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(p.res) + ridx);
                                hi += *reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(p.res) + ridx + 4);
                            }
                            if (p.relu) {
                                lo = __builtin_elementwise_max(lo, float4v{0.f, 0.f, 0.f, 0.f});
                                hi = __builtin_elementwise_max(hi, float4v{0.f, 0.f, 0.f, 0.f});
                            }
                            *reinterpret_cast<float4v*>(reinterpret_cast<float*>(p.out) + oidx) = lo;
                            *reinterpret_cast<float4v*>(reinterpret_cast<float*>(p.out) + oidx + 4) = hi;
                        } else {
                            if (EPI == 1 && has_res) {
                                const half8 rv = pre_ok ? rpre[half][PRE ? e : 0] : *reinterpret_cast<const half8*>(resp + ridx);
                                lo += __builtin_convertvector(__builtin_shufflevector(rv, rv, 0, 1, 2, 3), float4v);
                                hi += __builtin_convertvector(__builtin_shufflevector(rv, rv, 4, 5, 6, 7), float4v);
                            }
                            if (EPI == 3) {          // exact GELU (nn.GELU default)
                                // pairs: packed fp32 multiplies / fmas (same bits as the scalar gelu_erf of the general epilogue)
                                const float2v g0 = gelu_erf2((float2v){lo[0], lo[1]}), g1 = gelu_erf2((float2v){lo[2], lo[3]});
                                const float2v g2 = gelu_erf2((float2v){hi[0], hi[1]}), g3 = gelu_erf2((float2v){hi[2], hi[3]});
                                lo = (float4v){g0[0], g0[1], g1[0], g1[1]};
                                hi = (float4v){g2[0], g2[1], g3[0], g3[1]};
                            }
                            const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hi, half4);
                            half8 hv = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                            // ReLU after the rounding equals ReLU before it (rounding is monotonic and keeps 0)
                            if (EPI == 1 && p.relu) hv = __builtin_elementwise_max(hv, half8{0, 0, 0, 0, 0, 0, 0, 0});
                            *reinterpret_cast<half8*>(outp + oidx) = hv;
                        }
                    }
                    oidx += (long)ERPP * p.ldc;
                    ridx += (long)ERPP * p.Cout;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < EROWS; ++e) {
                const int r = tid / VPR + e * ERPP;
                const int m = m0 + half * (BM / 2) + r;
                if (m < p.M && n < p.Cout) igemm_store_row8(p, Cs + r * CP + c8, m, n, bias8, pre_ok, rpre[half][PRE ? e : 0]);
            }
        }
    }
}

template <int BM, int BN, int BKT, int NSTAGE, int WN, int AK, int EPI>
int launch2k(const IgemmParams& p, hipStream_t s) {
    constexpr int smem = Smem2<BM, BN, BKT, NSTAGE>::kBytes;
    if (smem > 64 * 1024) {
        static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
        if (first_on_device(attr_set)) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, BKT, NSTAGE, WN, AK, EPI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            mark_on_device(attr_set);
        }
    }
    const int nsplit = p.splitk > 1 ? p.splitk : 1;
    hipLaunchKernelGGL((igemm2_kernel<BM, BN, BKT, NSTAGE, WN, AK, EPI>), dim3(p.tiles_m * p.tiles_n * nsplit), dim3(128 * WN), smem, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

// which epilogue specialisation covers the launch (0: none, the general one)
int epilogue_kind(const IgemmParams& p) {
    if ((p.Cout & 7) || p.splitk > 1 || p.res_mode > 1) return 0;
    if (!p.out_f32 && (p.ldc & 7) == 0 && !(p.res_mode && p.res_f32)) {
        if (p.relu <= 1) return 1;
        if (p.relu == 2 && !p.res_mode) return 3;
    }
    if (p.out_f32 && (p.ldc & 3) == 0 && p.relu <= 1 && (!p.res_mode || p.res_f32)) return 2;
    return 0;
}
// FLAT addressing: 1x1 / linear, pad 0, K = Cin (stride 1 must cover the whole input: row m = pixel m)
bool flat_ok(const IgemmParams& p) {
    return p.ntaps == 1 && p.pad == 0 && p.Cin == p.Kpad && (p.stride != 1 || (p.Ho == p.H && p.Wo == p.W));
}

template <int BM, int BN, int BKT, int NSTAGE, bool SMALLC, int WN = 2>
int launch2(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.tiles_m = ceil_div(p.M, BM);
    p.tiles_n = ceil_div(p.Cout, BN);
    // option igemm_generic = 1: always the general addressing + general epilogue -- the parity tests compare the specialised
    // paths against it bit for bit
    const bool generic = g_opt.igemm_generic != 0;
    const int epi = generic ? 0 : epilogue_kind(p);
    if constexpr (SMALLC) {
        return epi == 1 ? launch2k<BM, BN, BKT, NSTAGE, WN, 1, 1>(p, s) : launch2k<BM, BN, BKT, NSTAGE, WN, 1, 0>(p, s);
    } else {
        if (!generic && flat_ok(p)) {
            switch (epi) {
                case 1: return launch2k<BM, BN, BKT, NSTAGE, WN, 2, 1>(p, s);
                case 2: return launch2k<BM, BN, BKT, NSTAGE, WN, 2, 2>(p, s);
                case 3: return launch2k<BM, BN, BKT, NSTAGE, WN, 2, 3>(p, s);
                default: return launch2k<BM, BN, BKT, NSTAGE, WN, 2, 0>(p, s);
            }
        }
        return epi == 1 ? launch2k<BM, BN, BKT, NSTAGE, WN, 0, 1>(p, s) : launch2k<BM, BN, BKT, NSTAGE, WN, 0, 0>(p, s);
    }
}

// ---- tile configurations ----------------------------------------------------------------------------------------
// Every configuration computes the same sums in the same order (K ascending, 16 at a time inside the MFMA), so the
// choice changes the time of a layer and never its result.
struct TileCfg {
    int bm, bn, bkt, nstage;
    int (*launch)(const IgemmParams&, hipStream_t);
};
const TileCfg kCfgs[] = {
    {64, 64, 32, 2, &launch2<64, 64, 32, 2, false>},        {128, 64, 32, 2, &launch2<128, 64, 32, 2, false>},
    {128, 128, 32, 2, &launch2<128, 128, 32, 2, false>},    {64, 64, 64, 2, &launch2<64, 64, 64, 2, false>},
    {128, 64, 64, 2, &launch2<128, 64, 64, 2, false>},      {128, 128, 64, 2, &launch2<128, 128, 64, 2, false>},
    {128, 256, 64, 2, &launch2<128, 256, 64, 2, false, 4>}, {256, 256, 64, 2, &launch2<256, 256, 64, 2, false, 4>},
    {128, 128, 64, 3, &launch2<128, 128, 64, 3, false>},    {128, 64, 64, 3, &launch2<128, 64, 64, 3, false>},
    {128, 128, 32, 3, &launch2<128, 128, 32, 3, false>},    {128, 256, 32, 3, &launch2<128, 256, 32, 3, false, 4>},
    {256, 256, 32, 3, &launch2<256, 256, 32, 3, false, 4>}, {256, 256, 32, 2, &launch2<256, 256, 32, 2, false, 4>},
    {128, 64, 32, 3, &launch2<128, 64, 32, 3, false>},      {128, 128, 32, 4, &launch2<128, 128, 32, 4, false>},
    {128, 256, 32, 4, &launch2<128, 256, 32, 4, false, 4>}, {128, 64, 32, 4, &launch2<128, 64, 32, 4, false>},
    // 4 waves with 64x128 per wave (128 accumulator registers): half the per-MFMA address / loop overhead of the 8-wave form
    {128, 256, 32, 3, &launch2<128, 256, 32, 3, false, 2>}, {128, 256, 64, 2, &launch2<128, 256, 64, 2, false, 2>},
    // nstage 5: the anti-phase schedule over a 4-stage ring (see igemm2_kernel)
    {256, 256, 32, 5, &launch2<256, 256, 32, 5, false, 4>},
    // small tiles with deep rings for the launches of a one-batch call (8 frames: res4 has 19456 rows): many workgroups per CU, three
    // K tiles in flight each, instead of one or two workgroups per CU waiting on a 2-stage ring
    {64, 64, 32, 4, &launch2<64, 64, 32, 4, false>},        {64, 128, 32, 4, &launch2<64, 128, 32, 4, false>},
    {64, 64, 64, 3, &launch2<64, 64, 64, 3, false>},
    // (whole-line BKT 64 tiles in 3- / 4-stage rings -- 128x256x64/3, 128x128x64/4 -- were timed on the HBM-paced long-K N = 256 layers:
    // 195 / 247 us against 168 for 256x256x32/5 on res4 conv1; not in the table)
};
constexpr int kNumCfg = sizeof(kCfgs) / sizeof(kCfgs[0]);
const TileCfg kStemCfgs[] = {   // Cin == 8 stem (one filter tap per 16-byte chunk)
    {128, 64, 32, 2, &launch2<128, 64, 32, 2, true>},
    {128, 64, 64, 2, &launch2<128, 64, 64, 2, true>},
};

bool cfg_valid(const TileCfg& c, const IgemmParams& p) {
    if (p.splitk > 1 && (p.Kpad / c.bkt) % p.splitk) return false;
    if (c.bm > 128 && p.M < 2 * c.bm) return false;
    if (c.bn > 64 && p.Cout <= c.bn / 2) return false;          // more than half of the tile would be padding
    return true;
}

// the hand rule used without the tuner (DVID_IGEMM_TUNE=0) and as the tuner's starting point
int heuristic_cfg(const IgemmParams& p) {
    // Long-K layers are bound by the global->LDS staging rate: full 128-byte lines (BKT 64) win.  Short-K layers
    // (<= 4 K steps) are bound by HBM traffic and the epilogue: the smaller LDS footprint of BKT 32 (4 resident
    // workgroups per CU instead of 2) wins (measured per layer, tools/bench_igemm.py).
    const int base = p.Kpad >= 512 ? 3 : 0;
    const int ns = p.splitk > 1 ? p.splitk : 1;
    const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.Cout, 128) * ns;
    if (t128 >= 512 && p.Cout >= 128) return base + 2;
    const long t12864 = (long)ceil_div(p.M, 128) * ceil_div(p.Cout, 64) * ns;
    return t12864 >= 512 ? base + 1 : base;
}

// ---- per-shape autotuning ---------------------------------------------------------------------------------------
// The best tile depends on how M x N lands on 256 CUs (a 128x128 grid of 304 tiles runs two rounds, the second 19 %
// full) and on whether the layer is staging-, MFMA- or epilogue-bound; rules of thumb lose 10-40 % on individual
// layers (profiles/r01_igemm_tile_sweep.txt).  The first launch of every distinct problem shape therefore times each
// valid configuration on the real operands (output redirected to a scratch buffer) and the winner is cached.
struct ShapeKey {
    int M, Cout, Kpad, Cin, ntaps, stride, res_mode, flags;
    bool operator==(const ShapeKey& o) const {
        return M == o.M && Cout == o.Cout && Kpad == o.Kpad && Cin == o.Cin && ntaps == o.ntaps && stride == o.stride &&
               res_mode == o.res_mode && flags == o.flags;
    }
};
struct ShapeHash {
    size_t operator()(const ShapeKey& k) const {
        size_t h = 1469598103934665603ull;
        for (int v : {k.M, k.Cout, k.Kpad, k.Cin, k.ntaps, k.stride, k.res_mode, k.flags}) h = (h ^ (size_t)(unsigned)v) * 1099511628211ull;
        return h;
    }
};
std::mutex g_tune_mu;
std::unordered_map<ShapeKey, int, ShapeHash> g_tuned;

// DVID_IGEMM_TUNE_CACHE=<file>: winners are appended as they are found and read back at the first launch of the next
// process, so a deployment (or a profiling run) starts without the timing launches.  One line per shape:
// M Cout Kpad Cin ntaps stride res_mode flags cfg table_size
bool g_cache_serving = false;    // a non-empty DVID_IGEMM_TUNE_CACHE was loaded and DVID_IGEMM_TUNE is unset: no timing launches
void tune_cache_load() {
    const char* path = getenv("DVID_IGEMM_TUNE_CACHE");
    if (!path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    ShapeKey k;
    int cfg, n;
    size_t loaded = 0;
    while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d", &k.M, &k.Cout, &k.Kpad, &k.Cin, &k.ntaps, &k.stride, &k.res_mode, &k.flags, &cfg, &n) == 10)
        if (cfg >= 0 && ((k.flags & 4) ? (n == 2 && cfg < 2) : (n == kNumCfg && cfg < kNumCfg))) {
            g_tuned[k] = cfg;
            ++loaded;
        }
    fclose(f);
    // A deployment that ships a winners file gets the serving behaviour by default: cached / nearest-bucket winners, the hand
    // rule for anything unseen, never a stream synchronisation or an allocation on the launch path.  DVID_IGEMM_TUNE=1 (or
    // dvid_igemm_set_tuning(1)) keeps timing new shape buckets and appending them to the file.
    g_cache_serving = loaded > 0 && getenv("DVID_IGEMM_TUNE") == nullptr;
}
void tune_cache_append(const ShapeKey& k, int cfg, int n) {
    const char* path = getenv("DVID_IGEMM_TUNE_CACHE");
    if (!path) return;
    if (FILE* f = fopen(path, "a")) {
        fprintf(f, "%d %d %d %d %d %d %d %d %d %d\n", k.M, k.Cout, k.Kpad, k.Cin, k.ntaps, k.stride, k.res_mode, k.flags, cfg, n);
        fclose(f);
    }
}

// Tuning scratch: the timing launches write to a buffer that only ever grows (no hipMalloc / hipFree per shape); it is
// needed because a layer may accumulate in place (Swin's residual stream: out == res), where repeating the launch on the
// real output would add twice.
void* g_tune_scratch = nullptr;
size_t g_tune_scratch_bytes = 0;

std::atomic<long long> g_tune_passes{0};          // shape buckets timed on the calling path so far (dvid_igemm_tuning_passes; bench.py reports it)

int tune_shape(const IgemmParams& p, hipStream_t s, const TileCfg* cfgs, int ncfg, int fallback, int* best_out) {
    g_tune_passes.fetch_add(1, std::memory_order_relaxed);
    const size_t out_bytes = (size_t)p.M * p.ldc * (p.out_f32 ? 4 : 2) * (p.splitk > 1 ? p.splitk : 1);
    HIP_TRY(hipStreamSynchronize(s));           // a quiet stream for the timings; other streams keep running
    if (out_bytes > g_tune_scratch_bytes) {
        if (g_tune_scratch) HIP_TRY(hipFree(g_tune_scratch));
        g_tune_scratch = nullptr;
        g_tune_scratch_bytes = 0;
        const size_t want = out_bytes + out_bytes / 4;
        HIP_TRY(hipMalloc(&g_tune_scratch, want));
        g_tune_scratch_bytes = want;
    }
    static hipEvent_t a = nullptr, b = nullptr;
    if (!a) {
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
    }
    IgemmParams q = p;
    q.out = g_tune_scratch;
    constexpr bool log = false;          // (set to true to print every timing of the tuner)
    int best = fallback;
    float best_ms = 1e30f;
    for (int c = 0; c < ncfg; ++c) {
        if (!cfg_valid(cfgs[c], p)) continue;
        int rc = cfgs[c].launch(q, s);           // warm-up: code object load, L2 / MALL state
        if (rc != DVID_OK) continue;
        // three launches; a short launch (the 20-40 us layers of a one-batch call) is timed again over enough launches to fill ~0.3 ms (unless it is already 30 % behind the best so far), and
        // the lower mean counts: three 25-us launches are inside the noise of the clock ramp and of the event pair itself, and a wrong winner
        // there costs 10-15 us on each of ~900 launches per video (round 5: res4 conv1 at 8 frames ran 37 us in one process, 23 in another)
        float ms = 1e30f;
        int reps = 3;
        for (int round = 0; round < 2; ++round) {
            HIP_TRY(hipEventRecord(a, s));
            for (int r = 0; r < reps; ++r) (void)cfgs[c].launch(q, s);
            HIP_TRY(hipEventRecord(b, s));
            HIP_TRY(hipEventSynchronize(b));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, a, b));
            t /= reps;
            if (t < ms) ms = t;
            // long enough already, or clearly not the winner: no second measurement (a stream of ragged video tails pays for every timing launch)
            if (t * reps >= 0.3f || t > 1.3f * best_ms) break;
            reps = (int)fminf(32.f, ceilf(0.3f / fmaxf(t, 1e-3f)));
        }
        if (log)
            fprintf(stderr, "[igemm tune] M %d N %d K %d taps %d res %d split %d : %dx%dx%d/%d  %.2f us\n", p.M, p.Cout, p.Kpad, p.ntaps,
                    p.res_mode, p.splitk, cfgs[c].bm, cfgs[c].bn, cfgs[c].bkt, cfgs[c].nstage, ms * 1e3f);
        if (ms < best_ms) {
            best_ms = ms;
            best = c;
        }
    }
    if (log)
        fprintf(stderr, "[igemm tune] M %d N %d K %d -> %dx%dx%d/%d (%.2f us, %.0f TFLOP/s)\n", p.M, p.Cout, p.Kpad, cfgs[best].bm,
                cfgs[best].bn, cfgs[best].bkt, cfgs[best].nstage, best_ms * 1e3f, 2.0 * p.M * p.Cout * (double)p.alg_k / best_ms / 1e9);
    *best_out = best;
    return DVID_OK;
}

// Row counts are bucketed for the tuner's key (8 steps per octave above 1024 rows, multiples of 64 below): a stream of
// videos ends in ragged groups, so exact M values keep appearing for ever, while the best tile only depends on how many
// tiles the launch has to the nearest ~10 %.
int bucket_rows(int M) {
    if (M <= 1024) return (M + 63) / 64 * 64;
    const int sh = 31 - __builtin_clz((unsigned)M) - 3;
    return ((M + (1 << sh) - 1) >> sh) << sh;
}

// A layer's best tile changes slowly with the row count: a bucket that has not been timed takes the winner of the nearest
// timed bucket of the same layer within one octave, so a stream of ragged video tails costs one timing pass per layer and
// octave, not one per bucket.
// ... as long as the launch keeps the chip busy for several rounds either way.  Where the winner's tile count at THIS row count is
// under two rounds of the 256 CUs the best tile is a question of how the tiles land on the CUs and changes within an octave (round 5:
// res4 conv1 at 8 frames, M = 19456, inherited 256x256x32/5 from the 16-frame launch of the video's first call -- 76 workgroups, 37 us
// against 23 us for its own winner 128x64x64/2, on 814 launches per video): such a bucket inherits only from within a quarter octave.
int nearest_bucket_cfg(const ShapeKey& k, const TileCfg* cfgs) {
    int best = -1;
    double best_d = 1.0;            // |log2(M / M')| <= 1
    for (const auto& kv : g_tuned) {
        const ShapeKey& o = kv.first;
        if (o.Cout != k.Cout || o.Kpad != k.Kpad || o.Cin != k.Cin || o.ntaps != k.ntaps || o.stride != k.stride || o.res_mode != k.res_mode ||
            o.flags != k.flags || o.M <= 0)
            continue;
        const double d = fabs(log2((double)k.M / (double)o.M));
        const TileCfg& c = cfgs[kv.second];
        const long tiles = (long)ceil_div(k.M, c.bm) * ceil_div(k.Cout, c.bn) * (((k.flags >> 4) & 0xff) > 1 ? ((k.flags >> 4) & 0xff) : 1);
        if (tiles < 512 && d > 0.25) continue;
        if (d <= best_d) {
            best_d = d;
            best = kv.second;
        }
    }
    return best;
}

}  // namespace

int dvid_igemm_num_configs(void) { return kNumCfg; }
long long dvid_igemm_tuning_passes(void) { return g_tune_passes.load(std::memory_order_relaxed); }

int dvid_igemm_set_config(int cfg) {
    if (cfg < -1 || cfg >= kNumCfg) return DVID_ERR_ARG;
    g_opt.igemm_cfg = cfg;
    return DVID_OK;
}

int dvid_igemm_set_tuning(int mode) {
    if (mode < -1 || mode > 1) return DVID_ERR_ARG;
    g_opt.igemm_tune = mode;
    return DVID_OK;
}

// 3x3 / stride-1 layers on the halo-staged kernel (conv3x3.hip): 0 = off, 1 = on where the shape rule prefers it, 2 = on wherever the
// layer type fits (tests), -1 = the default (1)
int dvid_igemm_set_conv3x3(int mode) {
    if (mode < -1 || mode > 2) return DVID_ERR_ARG;
    g_opt.conv3x3 = mode < 0 ? DvidOptions().conv3x3 : mode;
    return DVID_OK;
}

// short-K / wide-N 1x1 layers on the weight-stationary kernel (wstat.hip): 0 = off, 1 = on where the shape rule prefers it, 2 = on
// wherever the layer type fits (tests), -1 = the default (1).  Bit-identical to igemm2, so the rule may look at the row count.
int dvid_igemm_set_wstat(int mode) {
    if (mode < -1 || mode > 2) return DVID_ERR_ARG;
    g_opt.wstat = mode < 0 ? DvidOptions().wstat : mode;
    return DVID_OK;
}

int dvid_igemm_launch(const IgemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.Cout <= 0) return DVID_OK;
    if (p.Kpad % 64 != 0 || p.Kpad < 64) return DVID_ERR_ARG;
    {
        const bool forced = g_opt.igemm_cfg >= 0;
        const int ws = g_opt.wstat;
        if (ws && !forced && (ws >= 2 ? dvid_wstat_supported(p) : dvid_wstat_preferred(p))) return dvid_wstat_launch(p, s);
    }
    return dvid_igemm2_launch(p, s);
}

int dvid_igemm2_launch(const IgemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.Cout <= 0) return DVID_OK;
    if (p.Kpad % 64 != 0 || p.Kpad < 64) return DVID_ERR_ARG;
    {
        // a function of the shape only -- never of a timing: the two kernels sum the same products in different orders.  A forced
        // tile configuration (the bit-identity tests, experiments) means the igemm2 kernel.
        const bool forced = g_opt.igemm_cfg >= 0;
        const int halo = g_opt.conv3x3;
        if (halo && !forced && (halo >= 2 ? dvid_conv3x3_halo_supported(p) : dvid_conv3x3_halo_preferred(p))) return dvid_conv3x3_halo_launch(p, s);
    }
    const bool smallc = ((p.Cin == 8 || p.Cin == 16) && p.KH * p.KW > 1);
    if (!smallc && (p.Cin % 64 != 0)) return DVID_ERR_UNSUPPORTED;
    if (p.res_mode == 2 && ((p.Ho | p.Wo) & 1)) return DVID_ERR_ARG;
    if (p.splitk > 1 && (!p.out_f32 || p.bias || p.relu || p.res_mode || smallc || (p.Kpad / 64) % p.splitk)) return DVID_ERR_ARG;
    const TileCfg* cfgs = smallc ? kStemCfgs : kCfgs;
    const int ncfg = smallc ? 2 : kNumCfg;
    // dvid_igemm_set_config() forces one configuration for every layer it is valid for (parity tests compare configurations bit
    // for bit; experiments)
    const int cfg_env = g_opt.igemm_cfg;
    const int g_tune_mode = g_opt.igemm_tune;
    static const bool tune_env = !(getenv("DVID_IGEMM_TUNE") && atoi(getenv("DVID_IGEMM_TUNE")) == 0);
    const bool tune = g_tune_mode < 0 ? tune_env : true;         // mode 0 still uses cached / preloaded winners
    int fallback = smallc ? (p.Kpad >= 512 ? 1 : 0) : heuristic_cfg(p);
    if (!cfg_valid(cfgs[fallback], p)) fallback = smallc ? 0 : (p.Kpad >= 512 ? 3 : 0);
    if (cfg_env >= 0 && cfg_env < ncfg && cfg_valid(cfgs[cfg_env], p)) return cfgs[cfg_env].launch(p, s);
    if (!tune) return cfgs[fallback].launch(p, s);
    const ShapeKey key{bucket_rows(p.M), p.Cout, p.Kpad, p.Cin, p.ntaps, p.stride, p.res_mode,
                       (p.out_f32 ? 1 : 0) | (p.res_f32 ? 2 : 0) | (smallc ? 4 : 0) | (p.splitk << 4) | (p.relu << 12)};
    int cfg = -1;
    {
        std::lock_guard<std::mutex> lock(g_tune_mu);
        static bool loaded = false;
        if (!loaded) {
            loaded = true;
            tune_cache_load();
        }
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            cfg = it->second;
        } else if (int near = nearest_bucket_cfg(key, cfgs); near >= 0 && cfg_valid(cfgs[near], p)) {
            cfg = near;                          // same layer, row count within a factor of two of a tuned bucket: inherit
            g_tuned.emplace(key, cfg);
        } else if (g_tune_mode == 0 || (g_tune_mode < 0 && g_cache_serving)) {
            cfg = fallback;                     // serving mode: no timing launches; the hand rule for unseen buckets
        } else {
            const int rc = tune_shape(p, s, cfgs, ncfg, fallback, &cfg);
            if (rc != DVID_OK) return rc;
            g_tuned.emplace(key, cfg);
            tune_cache_append(key, cfg, ncfg);
        }
    }
    if (!cfg_valid(cfgs[cfg], p)) cfg = fallback;      // a bucket's winner may not fit its smallest member
    return cfgs[cfg].launch(p, s);
}
