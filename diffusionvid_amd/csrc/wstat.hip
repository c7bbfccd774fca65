// Weight-stationary kernel for the short-K, wide-N 1x1 convolutions / linear layers (bottleneck conv3 + residual + ReLU:
// K = 128 / 256 / 512 -> N = 512 / 1024 / 2048; the decoder's dynamic_layer: K = 256 -> N = 32768; linear1: 256 -> 2048).
//
// These layers move ~4.6 KB of HBM per output row and need ~12 % of the MFMA peak at the HBM rate; tiled as independent
// 128 x 128 workgroups (igemm2) a tile lives ~19 us for 8 K steps -- every tile re-stages its A rows AND the same weight
// rows through the global -> LDS path (4 bytes of DMA per output byte at K = 256), pays a first-load latency and an fp32
// LDS round trip for its epilogue, and the launch ends at 2.9 TB/s (profiles/r02_igemm_vs_vendor_b104.txt).  Here the
// WEIGHTS do not move at all: a workgroup of 8 waves owns a slab of 256 output channels, wave w keeps the [32 x K] weight rows of
// its 32 channels in VGPRs as MFMA operands (K / 4 registers) for the whole launch, and the workgroup streams its range of
// output rows through a ring of [32 x K] A tiles (DMA, 0.5 byte per output byte at K = 256).  The product is computed transposed
// (D[n][m] = W A^T: the weights are the MFMA's first operand), so a lane ends up with channels of ONE output row; after a
// v_permlane32_swap per register pair it holds 2 x 8 consecutive channels -> residual and output move as 16-byte pieces
// straight from / to the accumulator layout, no fp32 LDS pass, no second barrier.  The residual rides the same DMA ring (each wave
// fetches the 2 x 1 KiB it will read back lane-linearly), so the only waits are one counted `vmcnt` (DMA pieces only) + one barrier per 32 rows.
// One persistent workgroup per CU (256); XCD x owns rows [x M / 8, (x + 1) M / 8) so the slabs that read the same A rows share an L2.
//
// Same MFMA, same K order (ascending, 16 per instruction) and the same epilogue arithmetic as igemm2 (fp32: + bias, + residual; round
// to fp16; ReLU): results are bit-identical to it (tests/test_gpu_kernels.py::test_wstat_matches_igemm2), so which of the two a
// launch runs on never changes a value.
#include <stdlib.h>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_ws[4] = {0u, 0u, 0u, 0u};

// x + float(h) in one instruction: v_fma_mix_f32 (h * 1.0 + x, fp16 source promoted exactly, one rounding -- the value of cvt + add)
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ws_mix_add_lo(unsigned int h2, float x) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(x));
    return d;
}
__device__ __forceinline__ float ws_mix_add_hi(unsigned int h2, float x) {
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(x));
    return d;
}
// lo / hv += the eight halves of rv
__device__ __forceinline__ void ws_add_res8(float4v& lo, float4v& hv, half8 rv) {
    const uint4v r = __builtin_bit_cast(uint4v, rv);
    lo[0] = ws_mix_add_lo(r[0], lo[0]);
    lo[1] = ws_mix_add_hi(r[0], lo[1]);
    lo[2] = ws_mix_add_lo(r[1], lo[2]);
    lo[3] = ws_mix_add_hi(r[1], lo[3]);
    hv[0] = ws_mix_add_lo(r[2], hv[0]);
    hv[1] = ws_mix_add_hi(r[2], hv[1]);
    hv[2] = ws_mix_add_lo(r[3], hv[2]);
    hv[3] = ws_mix_add_hi(r[3], hv[3]);
}

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ws_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int K, int D, int TN, bool HAS_RES>
struct WsSmem {
    static constexpr int kATile = 32 * K * 2;
    static constexpr int kRTile = HAS_RES ? 8 * TN * 2048 : 0;       // per wave and accumulator tile: 2 pieces of 1 KiB
    static constexpr int kStage = kATile + kRTile;
    static constexpr int kBytes = D * kStage;
};

// K: reduction length (= Cin = Kpad); D: ring depth (tiles t + 1 .. t + D - 1 in flight while tile t is computed); TN: 32-channel
// accumulator tiles per wave (a workgroup owns 256 TN channels; the TN chains of a wave share every A fragment read and the A tile's
// DMA serves TN times the outputs); HAS_RES: same-shape fp16 residual; ACT: 0 none, 1 ReLU, 2 exact GELU (Swin's fc1, no residual: the
// pair form gelu_erf2 of the igemm2 epilogue on the same fp32 sums, so the same bits).  Grid: 256 workgroups x 512 threads.  M % 32 == 0.
template <int K, int D, int TN, bool HAS_RES, int ACT>
__global__ __launch_bounds__(512) void wstat_kernel(IgemmParams p, int nslab) {
    constexpr bool RELU = ACT == 1;
    constexpr int NW = 8, WPX = 32;
    constexpr int SLAB = 256 * TN;                // output channels per workgroup
    constexpr int CH = K / 8;                     // 16-byte chunks per A row
    constexpr int RPP = 512 / K;                  // A rows per 1-KiB DMA piece
    constexpr int PIECES = 32 / RPP;              // A pieces per tile
    constexpr int APW = PIECES / NW;              // ... per wave
    constexpr int RP = HAS_RES ? 2 * TN : 0;      // residual pieces per wave per tile
    constexpr int LG = APW + RP;                  // DMA instructions per wave per step
    constexpr int SG = 2 * TN;                    // store instructions per wave per step
    constexpr int KS = K / 16;
    constexpr int A_TILE = WsSmem<K, D, TN, HAS_RES>::kATile;
    constexpr int STAGE = WsSmem<K, D, TN, HAS_RES>::kStage;
    static_assert(PIECES % NW == 0 && APW >= 1, "every wave issues the same number of A pieces");
    static_assert((D - 2) * LG + (D - 1) * SG < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lrow = lane & 31;

    // ---- rows of this workgroup: XCD x (= blockIdx % 8 under round-robin dispatch) owns an eighth of the 32-row blocks; inside it
    // the 32 workgroups are (sub-range, slab) pairs, or -- more than 32 slabs -- each walks slabs q, q + 32, ... over the whole eighth
    const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    const int MB = p.M >> 5;
    const int xb0 = (int)((long)MB * xcd / 8), xb1 = (int)((long)MB * (xcd + 1) / 8);
    int blk0 = xb0, blk1 = xb1, slab0 = q, slab_step = WPX;
    if (nslab <= WPX) {
        const int nsub = WPX / nslab, sub = q / nslab;
        slab0 = q - sub * nslab;
        slab_step = nslab;                       // one slab per workgroup
        blk0 = xb0 + (int)((long)(xb1 - xb0) * sub / nsub);
        blk1 = xb0 + (int)((long)(xb1 - xb0) * (sub + 1) / nsub);
        if (sub >= nsub) blk1 = blk0;            // slab counts that do not divide 32 (3, 6: Swin's qkv) leave 32 % nslab workgroups of an XCD idle
    }
    const int T = blk1 - blk0;                   // tiles
    if (T <= 0) return;                          // workgroup-uniform

    // ---- A pieces: piece j = wave + NW * i covers tile rows [RPP j, RPP j + RPP); lane -> (row, physical chunk).  The LDS image is
    // lane-linear; the XOR swizzle (key = row & 15) is applied to the SOURCE chunk and again on the fragment read.
    const char* a_src[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int piece = wave + NW * i;
        const int row = piece * RPP + lane / CH;
        const int lch = (lane % CH) ^ (row & 15);
        a_src[i] = reinterpret_cast<const char*>(p.in) + ((long)(blk0 * 32 + row) * K + lch * 8) * 2;
    }
    const int frag_key = lrow & 15;
    const int frag_row_off = lrow * (K * 2);

    for (int slab = slab0; slab < nslab; slab += slab_step) {
        const int n0 = slab * SLAB + 32 * TN * wave;          // first channel of this wave
        // ---- weights of this wave's 32 TN channels as MFMA first operands: lane -> (channel n0 + 32 j + lane % 32, k = 16 ks + 8 (lane / 32))
        half8 bf[TN][KS];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const half_t* wrow = p.w + (long)(n0 + 32 * j + lrow) * p.Kpad + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[j][ks] = *reinterpret_cast<const half8*>(wrow + 16 * ks);
        }
        // bias of the channels this lane finishes: tile j, group g = channels n0 + 32 j + 16 g + 8 hi + [0, 8)
        float bs[TN][2][8];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) bs[j][g][e] = p.bias ? p.bias[n0 + 32 * j + 16 * g + 8 * hi + e] : 0.f;
        // residual / output addresses of this lane's row in tile 0, accumulator tile 0, group 0
        const long col = n0 + 8 * hi;
        const char* r_src = HAS_RES ? reinterpret_cast<const char*>(p.res) + ((long)(blk0 * 32 + lrow) * p.Cout + col) * 2 : nullptr;
        half_t* o_dst = reinterpret_cast<half_t*>(p.out) + (long)(blk0 * 32 + lrow) * p.ldc + col;
        const char* a_cur[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i) a_cur[i] = a_src[i];

        // the ordinary loads above are complete before the first DMA is issued (the compiler would otherwise wait vmcnt(0) at their
        // first use, with the prologue's DMA in flight), and no wave still reads the previous slab's last tiles
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bf[j][ks]));
        __builtin_amdgcn_s_barrier();

        // DMA of tile `ti` (A pieces, then this wave's residual pieces) into stage ti % D.  M is a multiple of 32 here (the launcher hands
        // a ragged tail to igemm2), so no lane needs a mask; the waits count instructions, so steps past the range issue theirs too
        // (re-fetching the last tile)
        auto issue = [&](int ti) {
            char* const stg = smem + (ti % D) * STAGE;
            const bool more = ti + 1 < T;        // wave-uniform
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                ws_glds16(a_cur[i], stg + (wave + NW * i) * 1024);
                a_cur[i] += more ? 32 * K * 2 : 0;
            }
            if (HAS_RES) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 2; ++g) ws_glds16(r_src + 64 * j + 32 * g, stg + A_TILE + ((wave * TN + j) * 2 + g) * 1024);
                r_src += more ? (long)32 * p.Cout * 2 : 0L;
            }
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(d);

        for (int t = 0; t < T; ++t) {
            // own pieces of tile t have landed.  Issued after them: the pieces of tiles t + 1 .. t + D - 2 -- and the stores of the last
            // D - 1 steps, which are NOT counted: DMA pieces retire in issue order among themselves, stores (and ordinary loads) do not
            // retire in order with them (csrc/bneck.hip, note at kBytes), so a count that includes stores could be reached while tile t
            // is still in flight.  Counting the pieces only waits for at most the same operations (measured: no slower).
            ws_wait_vmcnt<(D - 2) * LG>();
            __builtin_amdgcn_s_barrier();        // tile t visible to every wave; nobody reads tile t - 1 any more
            asm volatile("" ::: "memory");
            issue(t + D - 1);                    // into the stage of tile t - 1

            const char* const stg = smem + (t % D) * STAGE;
            float16v acc[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // TN chains, K ascending, sharing the A fragment of each K step.  (Pinning all fragment reads in front of the first MFMA, two
            // row tiles per step, a skewed epilogue, anti-phase wave groups and 4-wave workgroups two to a CU were all measured: none is
            // faster -- the step is bound by its instruction count, profiles/r02_wstat.txt.)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 fa = *reinterpret_cast<const half8*>(stg + frag_row_off + (((2 * ks + hi) ^ frag_key) * 16));
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][ks], fa, acc[j], 0, 0, 0);
            }
            // ---- epilogue from the accumulator layout.  acc[4 r4 + r] = channel 8 r4 + 4 hi + r of row lane % 32; one half-wave
            // exchange per register pair (r4, r4 + 1) leaves lane < 32 with channels 16 g + [0, 8) and lane >= 32 with 16 g + 8 + [0, 8).
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned int u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float f = acc[j][r];   // (a bit_cast of the vector element itself reads element 0 whatever r is)
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
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float4v lo, hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = __uint_as_float(u[8 * g + e]) + bs[j][g][e];
                        hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bs[j][g][4 + e];
                    }
                    if (HAS_RES) {
                        const half8 rv = *reinterpret_cast<const half8*>(stg + A_TILE + ((wave * TN + j) * 2 + g) * 1024 + lane * 16);
                        ws_add_res8(lo, hv, rv);
                    }
                    if (ACT == 2) {
                        const float2v g0 = gelu_erf2((float2v){lo[0], lo[1]}), g1 = gelu_erf2((float2v){lo[2], lo[3]});
                        const float2v g2 = gelu_erf2((float2v){hv[0], hv[1]}), g3 = gelu_erf2((float2v){hv[2], hv[3]});
                        lo = (float4v){g0[0], g0[1], g1[0], g1[1]};
                        hv = (float4v){g2[0], g2[1], g3[0], g3[1]};
                    }
                    const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                    half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    if (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                    *reinterpret_cast<half8*>(o_dst + 32 * j + 16 * g) = o;
                }
            }
            o_dst += (long)32 * p.ldc;
        }
        // the tail's extra DMAs and this slab's residual reads retire before the next slab's prologue re-uses the stages
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
}

// ---- row-coalesced variant ---------------------------------------------------------------------------------------------------
// Same products and epilogue arithmetic; what changes is how the residual and the output meet global memory.  Above, a wave's
// residual piece / store covers 32 rows x 32 bytes (one accumulator tile's rows): 32 requests of a quarter line each, rows one row
// pitch apart.  Here the workgroup's [32 x 256] residual tile rides the DMA ring row-major (2 rows x 512 contiguous bytes per piece,
// XOR-swizzled like the A tile, read back in the accumulator layout), and the finished fp16 tile goes through a double-buffered
// [32 x 256] LDS image: written in the accumulator layout at the end of step t, read back lane-linearly after the barrier of step
// t + 1 and stored 2 rows x 512 contiguous bytes per instruction -- every global access of the kernel is whole 512-byte row segments.
template <int K, int D>
struct Ws2Smem {
    static constexpr int kATile = 32 * K * 2;
    static constexpr int kRTile = 32 * 256 * 2;
    static constexpr int kStage = kATile + kRTile;
    static constexpr int kCOff = D * kStage;
    static constexpr int kBytes = kCOff + 2 * kRTile;
};

template <int K, int D, bool HAS_RES, bool RELU>
__global__ __launch_bounds__(512) void wstat2_kernel(IgemmParams p, int nslab) {
    constexpr int NW = 8, WPX = 32, SLAB = 256;
    constexpr int CH = K / 8;
    constexpr int RPP = 512 / K;
    constexpr int PIECES = 32 / RPP;
    constexpr int APW = PIECES / NW;
    constexpr int RP = HAS_RES ? 2 : 0;
    constexpr int LG = APW + RP;
    constexpr int KS = K / 16;
    constexpr int A_TILE = Ws2Smem<K, D>::kATile;
    constexpr int R_TILE = Ws2Smem<K, D>::kRTile;
    constexpr int STAGE = Ws2Smem<K, D>::kStage;
    static_assert(PIECES % NW == 0 && APW >= 1, "every wave issues the same number of A pieces");
    static_assert((D - 2) * LG + (D - 1) * 2 < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const c_img = smem + Ws2Smem<K, D>::kCOff;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lrow = lane & 31;
    const char* const zero = reinterpret_cast<const char*>(g_zero_page_ws);

    const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    const int MB = (p.M + 31) >> 5;
    const int xb0 = (int)((long)MB * xcd / 8), xb1 = (int)((long)MB * (xcd + 1) / 8);
    int blk0 = xb0, blk1 = xb1, slab0 = q, slab_step = WPX;
    if (nslab <= WPX) {
        const int nsub = WPX / nslab, sub = q / nslab;
        slab0 = q - sub * nslab;
        slab_step = nslab;
        blk0 = xb0 + (int)((long)(xb1 - xb0) * sub / nsub);
        blk1 = xb0 + (int)((long)(xb1 - xb0) * (sub + 1) / nsub);
        if (sub >= nsub) blk1 = blk0;
    }
    const int T = blk1 - blk0;
    if (T <= 0) return;

    const char* a_src[APW];
    int a_rowm[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int piece = wave + NW * i;
        const int row = piece * RPP + lane / CH;
        const int lch = (lane % CH) ^ (row & 15);
        a_rowm[i] = blk0 * 32 + row;
        a_src[i] = reinterpret_cast<const char*>(p.in) + ((long)a_rowm[i] * K + lch * 8) * 2;
    }
    const int frag_key = lrow & 15;
    const int frag_row_off = lrow * (K * 2);
    // row-major [32 x 256] images (residual stage, output image): piece / store j = wave + 8 i covers rows 2 j, 2 j + 1; lane -> (row 2 j
    // + lane / 32, physical chunk lane % 32); logical chunk = physical ^ (row & 15)
    int rc_row[2], rc_lch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rc_row[i] = 2 * (wave + NW * i) + hi;
        rc_lch[i] = lrow ^ (rc_row[i] & 15);
    }
    // this lane's two accumulator-layout chunks in such an image: row lane % 32, logical chunk 4 wave + 2 g + hi
    int acc_off[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) acc_off[g] = lrow * 512 + (((4 * wave + 2 * g + hi) ^ frag_key) * 16);

    for (int slab = slab0; slab < nslab; slab += slab_step) {
        const int n0 = slab * SLAB + 32 * wave;
        half8 bf[KS];
        {
            const half_t* wrow = p.w + (long)(n0 + lrow) * p.Kpad + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[ks] = *reinterpret_cast<const half8*>(wrow + 16 * ks);
        }
        float bs[2][8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[g][e] = p.bias ? p.bias[n0 + 16 * g + 8 * hi + e] : 0.f;
        const char* r_src[2];
        half_t* o_dst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long row = (long)blk0 * 32 + rc_row[i];
            const long col = (long)slab * SLAB + rc_lch[i] * 8;
            r_src[i] = HAS_RES ? reinterpret_cast<const char*>(p.res) + (row * p.Cout + col) * 2 : zero;
            o_dst[i] = reinterpret_cast<half_t*>(p.out) + row * p.ldc + col;
        }
        const char* a_cur[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i) a_cur[i] = a_src[i];

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bf[ks]));
        __builtin_amdgcn_s_barrier();

        auto issue = [&](int ti) {
            char* const stg = smem + (ti % D) * STAGE;
            const bool more = ti + 1 < T;            // past the range the last tile is fetched again (same instruction count, no masks)
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                ws_glds16(a_cur[i], stg + (wave + NW * i) * 1024);
                a_cur[i] += more ? 32 * K * 2 : 0;
            }
            if (HAS_RES) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ws_glds16(r_src[i], stg + A_TILE + (wave + NW * i) * 1024);
                    r_src[i] += more ? (long)32 * p.Cout * 2 : 0L;
                }
            }
        };
        // the finished tile `ti` from its LDS image to global memory: 2 stores of 2 rows x 512 bytes
        auto store_tile = [&](int ti) {
            const char* img = c_img + (ti & 1) * R_TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half8 o = *reinterpret_cast<const half8*>(img + (wave + NW * i) * 1024 + lane * 16);
                *reinterpret_cast<half8*>(o_dst[i]) = o;
                o_dst[i] += (long)32 * p.ldc;
            }
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(d);

        for (int t = 0; t < T; ++t) {
            ws_wait_vmcnt<(D - 2) * LG>();          // pieces only (see wstat_kernel): the stores of the last steps are not counted
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's part of the output image is written
            __builtin_amdgcn_s_barrier();        // tile t (A and residual) visible; tile t - 1's output image complete; stage t - 1 free
            asm volatile("" ::: "memory");
            issue(t + D - 1);
            if (t) store_tile(t - 1);

            const char* const stg = smem + (t % D) * STAGE;
            float16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 fa = *reinterpret_cast<const half8*>(stg + frag_row_off + (((2 * ks + hi) ^ frag_key) * 16));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], fa, acc, 0, 0, 0);
            }
            unsigned int u[16];
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
            char* const img = c_img + (t & 1) * R_TILE;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + bs[g][e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bs[g][4 + e];
                }
                if (HAS_RES) {
                    const half8 rv = *reinterpret_cast<const half8*>(stg + A_TILE + acc_off[g]);
                    ws_add_res8(lo, hv, rv);
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                *reinterpret_cast<half8*>(img + acc_off[g]) = o;
            }
        }
        // last tile: its image is complete once every wave is here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        store_tile(T - 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
}

template <int K, int D, bool HAS_RES, bool RELU>
int ws2_launch_k(const IgemmParams& p, hipStream_t s) {
    constexpr int smem = Ws2Smem<K, D>::kBytes;
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstat2_kernel<K, D, HAS_RES, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        mark_on_device(attr_set);
    }
    hipLaunchKernelGGL((wstat2_kernel<K, D, HAS_RES, RELU>), dim3(256), dim3(512), smem, s, p, p.Cout / 256);
    LAUNCH_CHECK();
    return DVID_OK;
}

template <int K, int D>
int ws2_launch_v(const IgemmParams& p, hipStream_t s) {
    const bool res = p.res_mode == 1, relu = p.relu == 1;
    if (res) return relu ? ws2_launch_k<K, D, true, true>(p, s) : ws2_launch_k<K, D, true, false>(p, s);
    return relu ? ws2_launch_k<K, D, false, true>(p, s) : ws2_launch_k<K, D, false, false>(p, s);
}

template <int K, int D, int TN, bool HAS_RES, int RELU>
int ws_launch_k(const IgemmParams& p, hipStream_t s) {
    constexpr int smem = WsSmem<K, D, TN, HAS_RES>::kBytes;
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstat_kernel<K, D, TN, HAS_RES, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        mark_on_device(attr_set);
    }
    hipLaunchKernelGGL((wstat_kernel<K, D, TN, HAS_RES, RELU>), dim3(256), dim3(512), smem, s, p, p.Cout / (256 * TN));
    LAUNCH_CHECK();
    return DVID_OK;
}

template <int K, int D, int TN>
int ws_launch_v(const IgemmParams& p, hipStream_t s) {
    const bool res = p.res_mode == 1, relu = p.relu == 1;
    if constexpr (TN == 1 && K <= 256) {          // (64 channels per wave or K = 512 + a residual ring does not fit the LDS; those layers take the row-coalesced kernel / igemm2)
        if (res) return relu ? ws_launch_k<K, D, TN, true, 1>(p, s) : ws_launch_k<K, D, TN, true, 0>(p, s);
    }
    if (res) return DVID_ERR_UNSUPPORTED;
    if (p.relu == 2) return ws_launch_k<K, D, TN, false, 2>(p, s);
    return relu ? ws_launch_k<K, D, TN, false, 1>(p, s) : ws_launch_k<K, D, TN, false, 0>(p, s);
}

}  // namespace

// the layer type fits: 1x1 / linear over contiguous rows, K in {128, 256}, N a multiple of 256 with the slab count dividing (or a
// multiple of) the workgroups of an XCD, fp16 out, bias / ReLU / same-shape fp16 residual
bool dvid_wstat_supported(const IgemmParams& p) {
    if (p.ntaps != 1 || p.pad != 0 || p.stride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    // K = 512: only without a residual (the row-coalesced variant's residual ring does not fit beside 32-KB A tiles) -- Swin's stage-3 fc1
    if (p.Cin != p.Kpad || (p.Kpad != 128 && p.Kpad != 256 && !(p.Kpad == 512 && p.res_mode == 0))) return false;
    if (p.Cout % 256) return false;
    const int ns = p.Cout / 256;
    if (!(ns <= 32 ? (32 % ns == 0 || ns == 3 || ns == 6) : ns % 32 == 0)) return false;          // 3 / 6 slabs (Swin qkv, N = 3C): 30 of an XCD's 32 workgroups work
    if (p.out_f32 || p.splitk > 1 || p.relu > 2 || (p.ldc & 7)) return false;
    if (p.res_mode > 1 || (p.res_mode == 1 && p.res_f32)) return false;
    if (p.relu == 2 && p.res_mode) return false;          // exact GELU: Swin's fc1, no residual
    return true;
}

// ... and the launch is large enough for the persistent workgroups: each streams at least kMinBlocks row blocks per weight load.
// (24 until round 3; the 8-frame launches of a one-batch call -- res4 conv3: 9 blocks per workgroup, dynamic_layer: 9 -- run faster
// here than on igemm2's small tiles: 1311 vs 1264 frames/s with every supported launch forced onto this kernel,
// profiles/r03_lookahead1_ab.txt.)
bool dvid_wstat_preferred(const IgemmParams& p) {
    if (!dvid_wstat_supported(p)) return false;
    constexpr int kMinBlocks = 8;
    const int ns = p.Cout / 256;
    const long blocks_per_xcd = ((long)p.M + 31) / 32 / 8;
    const long per_wg = ns <= 32 ? blocks_per_xcd / (32 / ns) : blocks_per_xcd;
    return per_wg >= kMinBlocks;
}

static int wstat_launch_rows32(const IgemmParams& p, hipStream_t s);

// The kernels take whole 32-row blocks (no lane masks in their loops); a ragged tail of < 32 rows goes to igemm2 -- same values.
int dvid_wstat_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_wstat_supported(p)) return DVID_ERR_UNSUPPORTED;
    const int m0 = p.M & ~31, rem = p.M - m0;
    if (m0) {
        IgemmParams q = p;
        q.M = m0;
        const int rc = wstat_launch_rows32(q, s);
        if (rc != DVID_OK) return rc;
    }
    if (rem) {
        IgemmParams q = p;
        q.M = rem;
        q.in = p.in + (long)m0 * p.Kpad;
        q.out = reinterpret_cast<half_t*>(p.out) + (long)m0 * p.ldc;
        if (p.res) q.res = reinterpret_cast<const half_t*>(p.res) + (long)m0 * p.Cout;
        return dvid_igemm2_launch(q, s);
    }
    return DVID_OK;
}

static int wstat_launch_rows32(const IgemmParams& p, hipStream_t s) {
    // layers with a residual: the row-coalesced variant (res3 conv3 0.460 vs 0.497 ms, res4 conv3 0.291 vs 0.300 at 104 frames); without
    // one (dynamic_layer, linear1) the accumulator-layout stores are as fast or faster (0.780 vs 0.791), and 64 channels per wave
    // (512-channel slabs: every A fragment read and every DMA piece serves two MFMA chains) where the slab count allows it.
    if (p.Kpad != 512 && p.res_mode == 1) return p.Kpad == 128 ? ws2_launch_v<128, 4>(p, s) : ws2_launch_v<256, 3>(p, s);
    const int ns2 = p.Cout / 512;
    const bool wide = p.res_mode == 0 && p.Cout % 512 == 0 && ns2 >= 32 && ns2 % 32 == 0;      // dynamic_layer: 0.749 vs 0.775 ms; linear1 (4 slabs) is slower that way
    if (p.Kpad == 512) return ws_launch_v<512, 4, 1>(p, s);          // 128 registers of weights per wave, four 32-KB A tiles
    if (p.Kpad == 128) return wide ? ws_launch_v<128, 6, 2>(p, s) : ws_launch_v<128, 6, 1>(p, s);
    return wide ? ws_launch_v<256, 4, 2>(p, s) : ws_launch_v<256, 4, 1>(p, s);
}
