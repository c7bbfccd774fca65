// 1x1 convolution / linear layer with N = 256 outputs and a long K (res4 conv1: 1024 -> 256, 22 layers of the ResNet-101 path; the
// same shape class in res5 / the FPN laterals): the WEIGHTS never pass through LDS.
//
// What paces this layer in igemm2: it is HBM-bound (AI 205 FLOP/B) and sits at 3.9 TB/s; both operands of a
// 256 x 256 x 32 step are staged by LDS-DMA -- four 1-KiB pieces per wave and step, 32 KB of LDS per stage, so at most five stages
// (64 KB of A rows in flight per CU) fit beside nothing else -- and the lab loop puts "all four pieces" at 866 TFLOP/s against 1369 for
// two.  Half of those pieces carry the weight tile: the same 512 KB for every workgroup, L2-resident, re-staged 2888 times per launch.
// Here the roles of the operands are split by where they come from:
//   * A rows (HBM, read once): LDS-DMA ring of EIGHT 16-KB stages [256 rows x 32 k] -- two pieces per wave and step, up to seven tiles
//     (112 KB per CU) in flight: twice the memory-level parallelism of igemm2's ring at half its DMA issue cost;
//   * weights (L2): wave w owns output channels [32 w, 32 w + 32) for all 256 rows of the tile (8 accumulator tiles, 128 registers)
//     and reads ITS OWN weight fragments straight from global into the MFMA operand registers, in the fragment order the model packs
//     at load (make_frags: one contiguous 1-KiB wave load per [32 n x 16 k] fragment, from L2), three K steps ahead.
// The product is computed transposed (D[n][m] = W A^T, the weights are the MFMA's first operand): same MFMA, same ascending K order, same
// epilogue arithmetic (fp32 + bias, round to fp16, ReLU) as igemm2 -> bit-identical to it (tests/test_gpu_kernels.py::
// test_wdirect_matches_igemm2), so the launch-size rule may look at the row count.
//
// Counted waits (the contract of csrc/bneck.hip; tools/check_dma_waits.py): the weight loads are ordinary loads issued between DMA
// pieces.  A wait that must cover a PIECE counts pieces only (ordinary loads younger than the piece may retire ahead of it, so they
// must not be relied on to still be outstanding); a wait that must cover an ORDINARY load may count everything younger than it
// (younger loads retire behind it, younger pieces decrement later still).  One `s_waitcnt vmcnt(N)` per step serves both with
// N = min(pieces issued after this step's tile, everything issued after this step's weight loads): exact or longer, never early.
// The weight loads are inline asm: with LDS-DMA and ordinary loads in flight together hipcc drains vmcnt(0) at the first use of every
// ordinary load it knows about (a full latency per step); an asm load is invisible to that pass and its register is read only
// behind the kernel's own wait.
#include <stdlib.h>

#include <map>
#include <mutex>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"

namespace {

constexpr int BN = 256, BM = 256, BKT = 32, NST = 8, NW = 8;
constexpr int STAGE = BM * BKT * 2;          // 16 KB: [256 rows][32 k] fp16, 64 bytes per row, 16-byte chunks XOR-swizzled by (row >> 2) & 3
constexpr int PW = 3;                        // weight fragments are requested PW steps ahead; ring of PW + 1 steps x 2 fragments
constexpr int WSLOTS = PW + 1;

template <int N>
__device__ __forceinline__ void wd_wait() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");          // + every LDS read of the previous step has returned
}
__device__ __forceinline__ void wd_glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// issue pattern of the K loop, as compile-time arithmetic: virtual step v (from -(NST - 1)) issues [2 weight loads of K step v + PW, if it
// exists] then [2 pieces of tile min(v + NST - 1, nk - 1)].  -> the vmcnt of the wait at the top of step kt.
constexpr int wd_has_w(int v, int nk) { return (v + PW >= 0 && v + PW < nk) ? 2 : 0; }
constexpr int wd_wait_count(int kt, int nk) {
    // pieces issued after the pieces of tile kt (issued in step kt - NST + 1): steps kt - NST + 2 .. kt - 1
    int pieces = 2 * (NST - 2);
    // everything issued after the weight loads of K step kt (issued first in step kt - PW): that step's pieces + whole steps kt - PW + 1 .. kt - 1
    int after_w = 2;
    for (int v = kt - PW + 1; v <= kt - 1; ++v) after_w += 2 + wd_has_w(v, nk);
    return pieces < after_w ? pieces : after_w;
}

// K is a template parameter and the K loop is fully unrolled (ring slots and wait counts are compile-time; a rolled loop would make the
// register ring loop-carried and let the compiler's own wait insertion drain every outstanding load once per trip).
template <int K, bool RELU>
__global__ __launch_bounds__(512) void wdirect_kernel(IgemmParams p, const half_t* __restrict__ wfrag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lrow = lane & 31;
    const int tile = igemm_xcd_remap((int)blockIdx.x, p.tiles_m);
    const long m0 = (long)tile * BM;
    constexpr int nk = K / BKT, ksn = K / 16;
    float* bias_lds = reinterpret_cast<float*>(smem + NST * STAGE);
    if (tid < BN) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing ordinary in flight when the counted stream starts

    // ---- A by DMA: piece j = wave + 8 i (i = 0, 1) covers tile rows [16 j, 16 j + 16); lane -> (row, 16-byte chunk); the XOR swizzle
    // (key = (row >> 2) & 3) is applied to the SOURCE chunk and again on the fragment read
    const char* a_ptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + NW * i) + (lane >> 2);
        const long grow = m0 + row < p.M ? m0 + row : (long)p.M - 1;          // rows past the end re-read the last row (never stored)
        a_ptr[i] = reinterpret_cast<const char*>(p.in + grow * K + (((lane & 3) ^ ((row >> 2) & 3)) * 8));
    }
    auto issue_a = [&](int kt) {
        const int src = kt < nk ? kt : nk - 1;          // tiles past the end re-fetch the last one into a stage nobody reads: uniform counts
        char* stg = smem + (kt & (NST - 1)) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) wd_glds16(a_ptr[i] + (long)src * (BKT * 2), stg + (wave + NW * i) * 1024);
    };
    // ---- weights: fragment (n-tile = wave, K step g of 16) = 1 KiB at wfrag + ((wave * ksn + g) * 64 + lane) * 16 bytes
    const char* w_lane = reinterpret_cast<const char*>(wfrag) + ((long)wave * ksn * 64 + lane) * 16;
    half8 wreg[2 * WSLOTS];
#define WD_LOAD_W(dst, g) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(w_lane + (long)((g) >> 2) * 4096), "i"(((g) & 3) * 1024) : "memory")
#define WD_FENCE() asm volatile("" ::: "memory")

    float16v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // fragment addressing of the A tile: row = 32 jb + lrow, logical chunk = 2 ks + hi
    const int sw = (lrow >> 2) & 3;
    const int fa_off = lrow * (BKT * 2);
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((2 * ks + hi) ^ sw) * 16;

    // ---- prologue: virtual steps -(NST - 1) .. -1 in the order the steady state issues (the fences pin ordinary loads against pieces)
#pragma unroll
    for (int v = -(NST - 1); v < 0; ++v) {
        if (wd_has_w(v, nk)) {
            WD_LOAD_W(wreg[2 * ((v + PW) % WSLOTS)], 2 * (v + PW));
            WD_LOAD_W(wreg[2 * ((v + PW) % WSLOTS) + 1], 2 * (v + PW) + 1);
        }
        WD_FENCE();
        issue_a(v + NST - 1);
        WD_FENCE();
    }

#pragma unroll
    for (int kt = 0; kt < nk; ++kt) {
        constexpr int dummy = 0;
        (void)dummy;
        // tile kt's pieces and the weight fragments of K step kt have landed
        switch (wd_wait_count(kt, nk)) {
            case 12: wd_wait<12>(); break;
            case 10: wd_wait<10>(); break;
            case 8: wd_wait<8>(); break;
            case 6: wd_wait<6>(); break;
            default: wd_wait<4>(); break;
        }
        __builtin_amdgcn_s_barrier();          // tile kt visible to every wave; nobody reads tile kt - 1 any more
        WD_FENCE();
        if (wd_has_w(kt, nk)) {                // weights of K step kt + PW into the slot of step kt - 1
            WD_LOAD_W(wreg[2 * ((kt + PW) % WSLOTS)], 2 * (kt + PW));
            WD_LOAD_W(wreg[2 * ((kt + PW) % WSLOTS) + 1], 2 * (kt + PW) + 1);
        }
        WD_FENCE();
        issue_a(kt + NST - 1);                 // into the stage of tile kt - 1
        WD_FENCE();
        const half8 w0 = wreg[2 * (kt % WSLOTS)], w1 = wreg[2 * (kt % WSLOTS) + 1];
        const char* st = smem + (kt & (NST - 1)) * STAGE + fa_off;
        // the 8 A fragments of K sub-step 1 are read while the 8 MFMAs of sub-step 0 run (two register sets, issue order pinned below)
        half8 a0[8], a1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a0[j] = *reinterpret_cast<const half8*>(st + j * 32 * (BKT * 2) + choff[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) a1[j] = *reinterpret_cast<const half8*>(st + j * 32 * (BKT * 2) + choff[1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0[j], acc[j], 0, 0, 0);          // transposed: D[n][m]
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);          // reads of sub-step 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one read of sub-step 1 beside each MFMA of sub-step 0
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail's extra fetches

    // ---- epilogue from the accumulator layout: acc[jb][4 r4 + r] = channel 32 wave + 8 r4 + 4 hi + r of row 32 jb + lrow; one half-wave
    // exchange per register pair leaves lane < 32 with channels 16 g + [0, 8) and lane >= 32 with 16 g + 8 + [0, 8) of the wave's 32
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        const long row = m0 + 32 * jb + lrow;
        unsigned int u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = __float_as_uint(acc[jb][r]);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto sx = __builtin_amdgcn_permlane32_swap(u[8 * g + r], u[8 * g + 4 + r], false, false);
                u[8 * g + r] = sx[0];
                u[8 * g + 4 + r] = sx[1];
            }
        if (row < p.M) {
            half_t* o_dst = reinterpret_cast<half_t*>(p.out) + row * p.ldc + 32 * wave + 8 * hi;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = 32 * wave + 16 * g + 8 * hi;
                float4v lo, hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = __uint_as_float(u[8 * g + e]) + bias_lds[ch + e];
                    hv[e] = __uint_as_float(u[8 * g + 4 + e]) + bias_lds[ch + 4 + e];
                }
                const half4 hlo = __builtin_convertvector(lo, half4), hhi = __builtin_convertvector(hv, half4);
                half8 o = __builtin_shufflevector(hlo, hhi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (RELU) o = __builtin_elementwise_max(o, half8{0, 0, 0, 0, 0, 0, 0, 0});
                *reinterpret_cast<half8*>(o_dst + 16 * g) = o;
            }
        }
    }
}

// [cout][K] (K contiguous) -> fragment order (model.hip: make_frags), on the device: for callers that hand over the plain layout
__global__ void wd_make_frags_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int cout, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece per thread
    const int ksn = K / 16;
    const long total = (long)(cout / 32) * ksn * 64;
    if (i >= total) return;
    const int l = (int)(i & 63);
    const long f = i >> 6;
    const int ks = (int)(f % ksn), nt = (int)(f / ksn);
    *reinterpret_cast<half8*>(out + i * 8) = *reinterpret_cast<const half8*>(w + (long)(nt * 32 + (l & 31)) * K + ks * 16 + (l >> 5) * 8);
}

std::mutex g_wd_mu;
std::map<std::pair<const void*, int>, half_t*> g_wd_frags;          // (plain weights, K) -> fragment-order copy, made on first use

template <int K, bool RELU>
int launch(const IgemmParams& p0, const half_t* wfrag, hipStream_t s) {
    IgemmParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = 1;
    constexpr int smem = NST * STAGE + BN * 4;
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdirect_kernel<K, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    hipLaunchKernelGGL((wdirect_kernel<K, RELU>), dim3(p.tiles_m), dim3(512), smem, s, p, wfrag);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

// the layer type fits: 1x1 / linear over contiguous rows (row m = pixel m), N = 256, K = 512 / 1024 / 2048, fp16 out, bias / ReLU, no
// residual, no split-K
bool dvid_wdirect_supported(const IgemmParams& p) {
    if (p.ntaps != 1 || p.pad != 0 || p.stride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin != p.Kpad || (p.Kpad != 512 && p.Kpad != 1024 && p.Kpad != 2048)) return false;          // the instantiated K values
    if (p.Cout != BN || (p.ldc & 7)) return false;
    if (p.out_f32 || p.splitk > 1 || p.relu > 1 || p.res_mode) return false;
    return true;
}

// ... and the launch fills the chip: at least DVID_WDIRECT_MIN_TILES tiles of 256 rows (bit-identical to igemm2, so the rule may look at
// the row count)
bool dvid_wdirect_preferred(const IgemmParams& p) {
    if (!dvid_wdirect_supported(p)) return false;
    static const int min_tiles = getenv("DVID_WDIRECT_MIN_TILES") ? atoi(getenv("DVID_WDIRECT_MIN_TILES")) : 512;
    return (p.M + BM - 1) / BM >= min_tiles;
}

int dvid_wdirect_launch(const IgemmParams& p, hipStream_t s) {
    if (!dvid_wdirect_supported(p)) return DVID_ERR_UNSUPPORTED;
    const half_t* wfrag = p.wfrag;
    if (!wfrag) {
        // the caller handed over the plain [cout][K] layout only (stand-alone entry points, tests): fragment-order copy made once per
        // weight tensor, on the launch stream, and kept for the life of the process
        std::lock_guard<std::mutex> lock(g_wd_mu);
        auto key = std::make_pair(static_cast<const void*>(p.w), p.Kpad);
        auto it = g_wd_frags.find(key);
        if (it == g_wd_frags.end()) {
            half_t* buf = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), (size_t)p.Cout * p.Kpad * sizeof(half_t)));
            it = g_wd_frags.emplace(key, buf).first;
        }
        const long total = (long)(p.Cout / 32) * (p.Kpad / 16) * 64;
        hipLaunchKernelGGL(wd_make_frags_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p.w, it->second, p.Cout, p.Kpad);
        LAUNCH_CHECK();
        wfrag = it->second;
    }
    switch (p.Kpad) {
        case 512: return p.relu ? launch<512, true>(p, wfrag, s) : launch<512, false>(p, wfrag, s);
        case 1024: return p.relu ? launch<1024, true>(p, wfrag, s) : launch<1024, false>(p, wfrag, s);
        default: return p.relu ? launch<2048, true>(p, wfrag, s) : launch<2048, false>(p, wfrag, s);
    }
}

// ---- round-4 record (why this is in tools/lab/ and not in the library) ----------------------------------------------------------
// Bit-identical to igemm2 (tests: 5 shapes, ragged rows, K 512 / 1024 / 2048; 220 VGPRs, no spills) and SLOWER on the layer it was
// written for (profiles/r04d_wdirect_bench.txt, tools/bench_igemm.py --only res4.1.conv1): res4 conv1 at 304 frames 0.601 ms against
// igemm2's 0.497 (645 vs 780 TFLOP/s; the vendor GEMM 0.509), at 104 frames 0.212 vs 0.176, at 8 frames 0.033 vs 0.022.  Halving the
// DMA pieces and doubling the ring depth did not help because all eight waves move through a step together: after the barrier every wave
// issues its loads, then nine fragment reads, then waits for them (lgkmcnt) with the matrix pipe empty -- the two waves of a SIMD
// are in the same phase, so nobody covers the LDS latency.  igemm2's 256 x 256 configuration wins with twice the pieces because its
// two wave rows run in anti-phase (one reads while the other multiplies).  The idea that survived is the operand split itself --
// A rows by DMA, weight fragments from L2 into registers -- built into igemm2's anti-phase schedule as tile configuration NSTAGE 6
// (csrc/igemm2.hip), where the tuner times it against the others per shape.
