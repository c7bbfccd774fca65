// 1x1 convolution / linear layer with N = 256 outputs and a long K (res4 conv1: K = 1024, 22 layers; res5 / FPN / decoder
// linears of the same shape class): the A operand never passes through LDS.
//
// In igemm2 both operands are staged by DMA (global_load_lds): a 256 x 256 x 32 step costs every wave four 1-KiB pieces, and the
// lab loop (tools/lab/gemm_pingpong.hip, profiles/r02_lab_dma_ablation.txt) showed what that issue cost does to the K loop: all
// four pieces 866 TFLOP/s, the two weight pieces alone 1369, none 1380.  With N = 256 = one tile, a row of A is consumed by ONE
// workgroup, so the only reason it goes through LDS is to be shared by the waves that split the tile's columns.  Here they do not:
// wave w owns rows [32 w, 32 w + 32) of the 256-row tile and ALL 256 columns (8 accumulator tiles, 128 registers), so its A
// fragments are private -- lane l reads the 16 bytes of row l & 31 at k = 16 ks + 8 (l >> 5) straight from global into the
// MFMA operand registers, twelve K steps ahead of their use (no DMA piece, no LDS write, no barrier on its path).  Only the weight
// tile [256 n x 32 k] rides the DMA ring (4 stages of 16 KB, 2 pieces per wave and step, XOR-swizzled exactly like igemm2's B
// tile) -- the same bytes for every workgroup, from L2.  One counted vmcnt + one barrier per 32-K step.
//
// Same MFMA, same K order (ascending, 16 per instruction), same epilogue arithmetic (fp32 + bias, round to fp16, ReLU) as igemm2:
// bit-identical to it (tests/test_gpu_kernels.py::test_adirect_matches_igemm2), so the launch-size rule may look at the row count.
// The epilogue leaves the accumulator layout by one v_permlane32_swap per register pair (as wstat.hip): a lane stores 8 consecutive
// channels (16 bytes) of its row.
#include <stdlib.h>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int BN = 256, BM = 256, BKT = 32, NST = 4, NW = 8;
constexpr int STAGE = BN * BKT * 2;          // 16 KB
constexpr int PA = 16;                       // ring of A fragments per wave (K steps of 16)

// ... and every LDS read of the previous tile has returned (lgkmcnt): the stage it was read from is refilled right behind the barrier
template <int N>
__device__ __forceinline__ void ad_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ad_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// K is a template parameter and the K loop is fully unrolled: with a rolled loop the register ring of A fragments becomes loop-carried
// values, and the compiler's own wait-count insertion then drains every outstanding load (vmcnt(0)) once per trip -- right behind the
// loads it has just issued.
template <int K, bool RELU>
__global__ __launch_bounds__(512) void adirect_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lrow = lane & 31;
    const int ntiles = p.tiles_m;
    const int tile = igemm_xcd_remap((int)blockIdx.x, ntiles);
    const long m0 = (long)tile * BM;
    constexpr int nk = K / BKT;
    static_assert(nk % 8 == 0 && nk >= 8, "ring of 16 A fragments: 8 steps per rotation");
    // bias -> LDS once (read back in the epilogue without a vector-memory wait between the stores); published by the first step's barrier
    float* bias_lds = reinterpret_cast<float*>(smem + NST * STAGE);
    if (tid < BN) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing ordinary in flight when the counted DMA / asm-load stream starts

    // ---- weight DMA: piece j = wave + 8 i (i = 0, 1) covers tile rows [16 j, 16 j + 16); lane -> (row, 16-byte chunk); the XOR
    // swizzle (key = (row >> 2) & 3) is applied to the SOURCE chunk and again on the fragment read (igemm2's B tile, BKT = 32)
    const char* b_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + NW * i) + (lane >> 2);
        b_ptr[i] = reinterpret_cast<const char*>(p.w + (long)row * K + (((lane & 3) ^ ((row >> 2) & 3)) * 8));
    }
    auto issue_b = [&](int kt) {
        // tiles past the end re-fetch the last tile into a stage nobody reads any more: every step issues the same instruction count
        const int src = kt < nk ? kt : nk - 1;
        char* stg = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) ad_glds16(b_ptr[i] + (long)src * (BKT * 2), stg + (wave + NW * i) * 1024);
    };
    // ---- A: this lane's row, its 8-half slice of every K step
    const long arow = m0 + 32 * wave + lrow;
    const half_t* a_base = p.in + (arow < p.M ? arow : (long)p.M - 1) * K + 8 * hi;
    // The A loads are inline asm on purpose.  With LDS-DMA instructions and ordinary loads in flight together the compiler's wait-count
    // insertion treats the vector-memory counter as out of order and drains it (vmcnt(0)) at the first use of every ordinary load -- a
    // full memory latency per K step (wstat.hip routes its residual through the DMA ring for the same reason).  An asm load is invisible
    // to that pass; its result register is only read behind the kernel's own counted wait, which covers it (see the step loop).
    half8 areg[PA];
#define AD_LOAD_A(dst, ks) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(a_base), "i"((ks) * 32) : "memory")

    float16v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // fragment addressing of the weight tile: row = 32 j + lrow, logical chunk = 2 ks + hi
    const int sw = (lrow >> 2) & 3;
    const int fb_off = lrow * (BKT * 2);
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((2 * ks + hi) ^ sw) * 16;

    // ---- prologue, in the order the steady state issues (virtual steps -4 .. -1: [weights of tile s + 3], [A of K steps 2 s + 8, 2 s + 9])
    // (the empty asm statements pin the issue order of ordinary loads against the DMA instructions: the counted wait below relies on it)
#define AD_FENCE() asm volatile("" ::: "memory")
    // virtual steps -6 .. -1.  A row's 128-byte line holds four K steps: its four loads are issued back to back (the first misses, the
    // other three hit the line in L1); issued one per step they come ~500 cycles apart with 32 KB of other waves' lines between them
    // -- the whole L1 -- and every 32-byte piece is a fresh L2 request (measured: 0.185 ms against igemm2's 0.169 on res4 conv1).
    AD_LOAD_A(areg[0], 0);
    AD_LOAD_A(areg[1], 1);
    AD_LOAD_A(areg[2], 2);
    AD_LOAD_A(areg[3], 3);
    AD_LOAD_A(areg[4], 4);
    AD_LOAD_A(areg[5], 5);
    AD_LOAD_A(areg[6], 6);
    AD_LOAD_A(areg[7], 7);
    AD_FENCE();
    issue_b(0);
    issue_b(1);
    AD_FENCE();
    AD_LOAD_A(areg[8], 8);
    AD_LOAD_A(areg[9], 9);
    AD_LOAD_A(areg[10], 10);
    AD_LOAD_A(areg[11], 11);
    AD_FENCE();
    issue_b(2);
    AD_FENCE();

#pragma unroll
    for (int kt0 = 0; kt0 < nk; kt0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kt = kt0 + u;
            // weights of tile kt and A of K steps 2 kt, 2 kt + 1 have landed.  Issued after the weights of tile kt: the A loads of their
            // own step (four, on even steps) and the 2 weight pieces + A loads of each of the two steps since (vmcnt retires in issue
            // order).  The last six steps issue no A loads: a load whose result nothing reads would get a destination register the
            // compiler re-uses at once, and land in it later.
            auto a4 = [&](int s2) { return ((s2 & 1) == 0 && s2 < nk - 6) ? 4 : 0; };
            const int after = a4(kt - 3) + 2 + a4(kt - 2) + 2 + a4(kt - 1);
            if (after == 12) ad_wait_vmcnt<12>();
            else if (after == 8) ad_wait_vmcnt<8>();
            else ad_wait_vmcnt<4>();
            __builtin_amdgcn_s_barrier();          // tile kt visible to every wave; nobody reads tile kt - 1 any more
            asm volatile("" ::: "memory");
            issue_b(kt + 3);                       // into the stage of tile kt - 1
            AD_FENCE();
            const half8 a0 = areg[2 * u], a1 = areg[2 * u + 1];
            if ((kt & 1) == 0 && kt < nk - 6) {          // slots of K steps 2 kt - 4 .. 2 kt - 1: consumed in the two steps before this one
                AD_LOAD_A(areg[(2 * u + 12) & 15], 2 * kt + 12);
                AD_LOAD_A(areg[(2 * u + 13) & 15], 2 * kt + 13);
                AD_LOAD_A(areg[(2 * u + 14) & 15], 2 * kt + 14);
                AD_LOAD_A(areg[(2 * u + 15) & 15], 2 * kt + 15);
            }
            AD_FENCE();
            const char* st = smem + (kt & (NST - 1)) * STAGE + fb_off;
            // the 8 weight fragments of K sub-step 1 are read while the 8 MFMAs of sub-step 0 run (two register sets; the issue order is
            // pinned below -- hipcc otherwise reads two fragments, waits, issues two MFMAs, ...)
            half8 b0[8], b1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b0[j] = *reinterpret_cast<const half8*>(st + j * 32 * (BKT * 2) + choff[0]);
#pragma unroll
            for (int j = 0; j < 8; ++j) b1[j] = *reinterpret_cast<const half8*>(st + j * 32 * (BKT * 2) + choff[1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0[j], a0, acc[j], 0, 0, 0);          // transposed: D[n][m]
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1[j], a1, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);          // reads of sub-step 0
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one read of sub-step 1 beside each MFMA of sub-step 0
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail's extra fetches

    // ---- epilogue from the accumulator layout: acc[j][4 r4 + r] = channel 32 j + 8 r4 + 4 hi + r of row lrow; one half-wave exchange per
    // register pair leaves lane < 32 with channels 16 g + [0, 8) and lane >= 32 with 16 g + 8 + [0, 8) of n-tile j
    if (arow < p.M) {
        half_t* o_dst = reinterpret_cast<half_t*>(p.out) + arow * p.ldc + 8 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned int u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = acc[j][r];
                u[r] = __float_as_uint(f);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(u[8 * g + r], u[8 * g + 4 + r], false, false);
                    u[8 * g + r] = sx[0];
                    u[8 * g + 4 + r] = sx[1];
                }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = 32 * j + 16 * g + 8 * hi;
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + bias_lds[ch + e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bias_lds[ch + 4 + e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                *reinterpret_cast<half8*>(o_dst + 32 * j + 16 * g) = o;
            }
        }
    }
}

template <int K, bool RELU>
int launch(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = 1;
    constexpr int smem = NST * STAGE + BN * 4;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&adirect_kernel<K, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((adirect_kernel<K, RELU>), dim3(p.tiles_m), dim3(512), smem, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

// the layer type fits: 1x1 / linear over contiguous rows (row m = pixel m), N = 256, K = 512 / 1024 / 2048, fp16 out, bias / ReLU, no
// residual, no split-K
bool dvid_adirect_supported(const IgemmParams& p) {
    if (p.ntaps != 1 || p.pad != 0 || p.stride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin != p.Kpad || (p.Kpad != 512 && p.Kpad != 1024 && p.Kpad != 2048)) return false;          // the instantiated K values
    if (p.Cout != BN || (p.ldc & 7)) return false;
    if (p.out_f32 || p.splitk > 1 || p.relu > 1 || p.res_mode) return false;
    return true;
}

// ... and the launch fills the chip: at least 2 tiles of 256 rows per CU (bit-identical to igemm2, so the rule may look at the row count)
bool dvid_adirect_preferred(const IgemmParams& p) {
    if (!dvid_adirect_supported(p)) return false;
    static const int min_tiles = getenv("DVID_ADIRECT_MIN_TILES") ? atoi(getenv("DVID_ADIRECT_MIN_TILES")) : 512;
    return (p.M + BM - 1) / BM >= min_tiles;
}

int dvid_adirect_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_adirect_supported(p)) return DVID_ERR_UNSUPPORTED;
    switch (p.Kpad) {
        case 512: return p.relu ? launch<512, true>(p, s) : launch<512, false>(p, s);
        case 1024: return p.relu ? launch<1024, true>(p, s) : launch<1024, false>(p, s);
        default: return p.relu ? launch<2048, true>(p, s) : launch<2048, false>(p, s);
    }
}

// ---- round-3 record (why this is in tools/lab/ and not in the library) ----------------------------------------------------------
// Bit-identical to igemm2 (5 shapes, ragged rows) and SLOWER on the layer it was written for: res4 conv1 (252928 x 256 x 1024, 104 frames)
// 0.185 ms against igemm2's 0.169-0.175 (716 vs 760-790 TFLOP/s), with the A loads issued one per K step or four per 128-byte line.
// The premise was wrong: this shape is not paced by DMA issue slots but by HBM -- 648 MB per launch; the vendor GEMM's 0.129 ms IS
// 5.0 TB/s (AI = 205 FLOP/B -> 1030 TFLOP/s at that rate), igemm2 sits at 77 % of it.  Register-direct A loads fetch 16 bytes per lane
// from 32 different rows per instruction; the DMA pieces of igemm2 at least fetch 64 contiguous bytes per row.  What would close the
// gap is whole-line (128 B per row) requests with >= 40 KB per CU in flight, i.e. BKT = 64 tiles in a deeper ring than LDS allows next
// to a 256 x 256 accumulator tile -- not fewer LDS passes.
// Lessons kept for the product code: (1) with LDS-DMA and ordinary loads in flight together hipcc drains vmcnt(0) at every use of an
// ordinary load, (2) an inline-asm load whose result is dead gets a destination register the compiler re-uses immediately -- the load
// lands in it later.
