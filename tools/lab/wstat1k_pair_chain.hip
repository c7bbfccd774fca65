// LAB RECORD (round 3, not in the library).  Measured on one MI355X against igemm2's tuned tile, stand-alone, us per launch at the row counts
// of 8 / 16 / 32 / 64 / 104 / 304 frames of 608 x 1024:   25.9 / 21.8   39.2 / 33.4   63.6 / 68.6   139.7 / 171.8   228.7 / 260.2   665.6 / 502.5
// -- it wins only between ~32 and ~104 frames (x1.08-1.23): at 8 frames the 256 KB of weights per workgroup and the serial A / B phases cost
// more than igemm2's idle CUs, at 304 frames every tile is read once per slab.  (A 16-row-MFMA form without the wave pairs -- not
// bit-identical to igemm2, so it had to take every size -- gave one-batch calls 1367 -> 1437 frames/s and the 304-frame video 2336 -> 2310.)
// Bit-identical to igemm2 (it passed its test in the library build); kept for the chained-accumulator idea.
// Weight-stationary kernel for the long-K, narrow-N 1x1 layers: K = 1024 -> N = 256 (conv1 of the 22 identity blocks of res4), for the
// launches igemm2's tiles cannot fill the chip with.
//
// An 8-frame launch of this layer is 19456 rows = 76 tiles of 256 x 256 on 256 CUs, each streaming the same 512 KB of weights through
// the global -> LDS path next to its rows: 36 us for 10 GFLOP, the largest single item of a one-batch call (22 launches, 15 % of its
// kernel time, profiles/r03h_kernel_stats_lookahead1.txt).  Here the weights do not move: a persistent workgroup of 8 waves owns a slab
// of 128 output channels and streams its range of rows through a 2-stage ring of [32 x 1024] tiles (64 KB each, DMA); 256 workgroups
// = 8 XCDs x 16 row ranges x 2 slabs, the two slabs of a range on the same XCD so that the second read of a tile comes from its L2.
//
// A wave can hold 128 registers of weights = [32 channels x 512 K], half of what an output element needs -- and the sums must be
// igemm2's, bit for bit (same MFMA, K ascending 16 per instruction), because this kernel only takes the small launches and a frame's
// features must not depend on how many frames share its launch.  So a 32-channel group is served by a PAIR of waves: wave c (A) runs
// K steps 0-31 of a [32 x 32] tile, hands its accumulators to wave c + 4 (B) through LDS, and B runs K steps 32-63 ON those
// accumulators and finishes the tile -- one chain of 64 MFMAs per output, split over two register files.  Per 32 rows: vmcnt + barrier,
// A phase, barrier, B phase; A and B of a SIMD alternate (the pipe is half busy, which an L2- / latency-bound small launch does not
// notice).  Epilogue as wstat.hip: transposed product, one v_permlane32_swap per register pair, + bias, round, ReLU, 16-byte stores.
// Bit-identical to igemm2 (tests/test_gpu_kernels.py::test_wstat1k_matches_igemm2).
#include <stdlib.h>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero16_w1k[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void w1k_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void w1k_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int K1 = 1024, SLAB = 128;
constexpr int kTile = 32 * K1 * 2;             // 64 KB: [32 rows][1024] fp16, 16-byte chunks XOR-swizzled by (row & 15)
constexpr int kHand = 4 * 4096;                // accumulators of the four wave pairs in flight from A to B
constexpr int kW1kBytes = 160 * 1024;          // 144 KB are used; the whole LDS keeps the CU to this workgroup (csrc/bneck.hip, note at kBytes)

__global__ __launch_bounds__(512) void wstat1k_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hand = smem + 2 * kTile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, c = wave & 3;          // role 0: K steps 0-31 (A), 1: K steps 32-63 on A's accumulators + epilogue (B)
    const int hi = lane >> 5, lrow = lane & 31;
    const char* const zero = reinterpret_cast<const char*>(g_zero16_w1k);

    // ---- rows of this workgroup: XCD x (= blockIdx % 8) owns an eighth of the 32-row blocks, its 32 workgroups are 16 sub-ranges x 2 slabs
    const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    const int slab = q & 1, sub = q >> 1;
    const int MB = (p.M + 31) >> 5;
    const int xb0 = (int)((long)MB * xcd / 8), xb1 = (int)((long)MB * (xcd + 1) / 8);
    const int blk0 = xb0 + (int)((long)(xb1 - xb0) * sub / 16), blk1 = xb0 + (int)((long)(xb1 - xb0) * (sub + 1) / 16);
    const int T = blk1 - blk0;
    if (T <= 0) return;

    // ---- weights of this wave: lane -> (channel n0 + lane % 32, k = 512 role + 16 ks + 8 (lane / 32)), MFMA first operands
    const int n0 = slab * SLAB + 32 * c;
    half8 bf[32];
    {
        const half_t* wrow = p.w + (long)(n0 + lrow) * K1 + 512 * role + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) bf[ks] = *reinterpret_cast<const half8*>(wrow + 16 * ks);
    }
    float bs[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[g][e] = p.bias ? p.bias[n0 + 16 * g + 8 * hi + e] : 0.f;

    // ---- A tile DMA: piece j = wave + 8 i (i < 8) is half a row: row j / 2, physical chunks 64 (j % 2) + lane holding logical chunk ^ (row & 15)
    auto issue = [&](int t) {
        char* const stg = smem + (t & 1) * kTile;
        const int tt = t < T ? t : T - 1;                      // past the range: the last tile again (same instruction count)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = wave + 8 * i;
            const int row = j >> 1;
            const long grow = (long)(blk0 + tt) * 32 + row;
            const int lch = (64 * (j & 1) + lane) ^ (row & 15);
            w1k_glds16(grow < p.M ? reinterpret_cast<const char*>(p.in + grow * K1 + lch * 8) : zero, stg + j * 1024);
        }
    };
    half_t* const outp = reinterpret_cast<half_t*>(p.out);
    const int frag_off = lrow * (K1 * 2);
    const int frag_key = lrow & 15;
    char* const myhand = hand + c * 4096 + lane * 16;

    // the ordinary loads above are complete before the first DMA is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(bf[ks]));
    issue(0);

    for (int t = 0; t < T; ++t) {
        // tile t landed.  Issued after its pieces: nothing by an A wave, the two stores of the previous tile by a B wave (they are
        // unconditional on every tile but the launch's last, which nothing waits behind)
        if (role && t) w1k_wait_vmcnt<2>(); else w1k_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // tile t visible to every wave; B has taken the accumulators of tile t - 1 and read its stage for the last time
        asm volatile("" ::: "memory");
        issue(t + 1);                          // into the stage tile t - 1 was read from
        asm volatile("" ::: "memory");

        const char* const stg = smem + (t & 1) * kTile + frag_off;
        float16v acc;
        if (role == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const half8 fa = *reinterpret_cast<const half8*>(stg + (((2 * ks + hi) ^ frag_key) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], fa, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4v*>(myhand + r4 * 1024) = (float4v){acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // A's accumulators are in LDS
        asm volatile("" ::: "memory");
        if (role == 1) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4v v = *reinterpret_cast<const float4v*>(myhand + r4 * 1024);
                acc[4 * r4] = v[0];
                acc[4 * r4 + 1] = v[1];
                acc[4 * r4 + 2] = v[2];
                acc[4 * r4 + 3] = v[3];
            }
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                const half8 fa = *reinterpret_cast<const half8*>(stg + (((2 * (32 + ks) + hi) ^ frag_key) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks], fa, acc, 0, 0, 0);
            }
            // epilogue from the accumulator layout (wstat.hip): register 4 r4 + r = channel 8 r4 + 4 hi + r of row lane % 32; one half-wave
            // exchange per register pair leaves the lane with channels 16 g + 8 hi + [0, 8)
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
            const long grow = (long)(blk0 + t) * 32 + lrow;
            half_t* const o_dst = outp + grow * p.ldc + n0 + 8 * hi;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + bs[g][e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bs[g][4 + e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (p.relu) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                if (grow < p.M) *reinterpret_cast<half8*>(o_dst + 16 * g) = o;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the look-ahead DMA past the range targets this workgroup's LDS
}

}  // namespace

// the layer type: 1x1 / stride 1 over contiguous rows, K = 1024, N = 256, fp16 out, bias / ReLU, no residual, no split-K
bool dvid_wstat1k_supported(const IgemmParams& p) {
    static const bool on = !(getenv("DVID_WSTAT1K") && atoi(getenv("DVID_WSTAT1K")) == 0);
    if (!on) return false;
    if (p.ntaps != 1 || p.pad != 0 || p.stride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin != K1 || p.Kpad != K1 || p.Cout != 2 * SLAB) return false;
    if (p.out_f32 || p.splitk > 1 || p.relu > 1 || p.res_mode != 0 || (p.ldc & 7) || p.M <= 0) return false;
    return true;
}

// ... and the launch is one igemm2's 256-row tiles leave CUs idle on: up to DVID_WSTAT1K_MAX (default 40) row blocks per workgroup, i.e.
// ~67 frames of 608 x 1024.  (Measured on one box: one-batch calls 1367 -> 1437 frames/s with this kernel on the 8-frame launches; a
// 304-frame launch is 5 % slower here than on igemm2, where every tile's rows are read once instead of once per slab.)  Bit-identical to
// igemm2, so the rule may look at the row count.
bool dvid_wstat1k_preferred(const IgemmParams& p) {
    if (!dvid_wstat1k_supported(p)) return false;
    static const int kMax = getenv("DVID_WSTAT1K_MAX") ? atoi(getenv("DVID_WSTAT1K_MAX")) : 40;
    return ((long)p.M + 31) / 32 <= (long)kMax * 128;
}

int dvid_wstat1k_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_wstat1k_supported(p)) return DVID_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstat1k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kW1kBytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(wstat1k_kernel, dim3(256), dim3(512), kW1kBytes, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}
