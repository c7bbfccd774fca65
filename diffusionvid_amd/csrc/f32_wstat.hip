// DTYPE float32, split operands: weight-stationary kernel for the short-K / wide-N 1x1 layers (bottleneck conv3 + residual + ReLU:
// K = 128 / 256 -> N = 512 / 1024; the decoder's dynamic_layer: 256 -> 32768; linear1: 256 -> 2048; Swin's fc1: C -> 4 C with GELU).
//
// On 128 x 128 tiles (csrc/f32.hip: f32x3_igemm_kernel) these layers are paced by operand traffic through the L2 -> CU path: at K = 256
// every tile re-stages 128 KB of fp32 activations AND 128 KB of (hi, lo) weights for 64 KB of output -- res4 conv3 moves 17.7 GB per
// 304-frame launch through that path for 6.8 GB of HBM traffic (2.4 TB/s), dynamic_layer 60 GB for 12 (1.5 TB/s;
// profiles/r06_layers_r101_x1_float32.csv).  Here, as in csrc/wstat.hip, the WEIGHTS do not move: a workgroup of 8 waves owns a slab
// of 256 output channels, wave w keeps the (hi, lo) rows of its 32 channels in registers as MFMA first operands (K / 2 registers) for
// the whole launch, and the workgroup streams its range of output rows in tiles of 32: global fp32 -> registers -> split once,
// cooperatively -> LDS as two fp16 planes (XOR-swizzled rows, double-buffered, one barrier per tile).  The product is computed
// transposed (D[n][m] = W A^T), so after a v_permlane32_swap per register pair a lane holds 8 consecutive channels of one output row:
// residual and output move as 32-byte runs (a whole 128-byte line per row and wave) straight from / to the accumulator layout.
// One persistent workgroup per CU; XCD x owns rows [x M / 8, (x + 1) M / 8) so the slabs that read the same rows share an L2.
//
// Same split (round to nearest even, lo = fp16(v - hi)), same pre-split weight planes, same three products per 16-deep K step in
// the same order (w_hi a_lo, w_lo a_hi, w_hi a_hi), K ascending, same epilogue arithmetic as f32x3_igemm_kernel: results are
// bit-identical to it (tests/test_gpu_f32.py::test_f32_wstat_matches_tiled), so which of the two a launch runs on never changes a value.
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "options.h"

namespace {

template <int K, int RB>
struct F32WsSmem {
    static constexpr int kPlane = 32 * RB * K * 2;     // one fp16 plane of a tile of 32 RB rows
    static constexpr int kStage = 2 * kPlane;          // hi | lo
    static constexpr int kVecOff = 2 * kStage;         // two stages, then the slab's per-channel scale | bias (256 floats each)
    static constexpr int kBytes = kVecOff + 2 * 256 * 4;
};

// RB: 32-row blocks per tile (a wave runs RB accumulator chains against the same weight registers; K = 64 needs 2 for every thread to have
// a chunk to split).  ACT: 0 none, 1 ReLU, 2 exact GELU.  Grid: 256 workgroups x 512 threads.  M % (32 RB) == 0, Cout % 256 == 0, Kpad == Cin == K.
template <int K, int RB, bool HAS_RES, int ACT>
__global__ __launch_bounds__(512) void f32x3_wstat_kernel(F32GemmParams p, int nslab) {
    constexpr int WPX = 32;
    constexpr int KS = K / 16;
    constexpr int TR = 32 * RB;                   // tile rows
    constexpr int CPR = K / 8;                    // 8-value chunks per row
    constexpr int CPT = TR * CPR / 512;           // chunks per thread and tile
    constexpr int RSTEP = 512 / CPR;              // tile rows between a thread's chunks
    using SM = F32WsSmem<K, RB>;
    constexpr int PLANE = SM::kPlane;
    constexpr int STAGE = SM::kStage;
    static_assert(CPT >= 1, "every thread splits at least one chunk per tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const vec_scale = reinterpret_cast<float*>(smem + SM::kVecOff);
    float* const vec_bias = vec_scale + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lrow = lane & 31;

    // rows of this workgroup: the decomposition of csrc/wstat.hip (XCD = blockIdx % 8 owns an eighth of the 32-row blocks; inside it the 32
    // workgroups are (sub-range, slab) pairs, or -- more than 32 slabs -- each walks slabs q, q + 32, ... over the whole eighth)
    const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    const int MB = p.M / TR;
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
    if (T <= 0) return;                           // workgroup-uniform

    // XOR key of a row's 16-byte chunks: the rows of one ds_read_b128 lane group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) must land on
    // 16 different (bank quarter, chunk) pairs.  Rows of 256 / 512 bytes all start on bank 0: key = row % 16; rows of 128 bytes (K = 64, 8
    // chunks) alternate between two bank halves: key = (row / 2) % 8.
    auto swz_key = [](int row) { return K == 64 ? (row >> 1) & 7 : row & 15; };
    // this thread's chunks of a tile: chunk c = tid + 512 i -> (row, chunk in row); 64 lanes x 32 bytes = whole rows of the source
    const float* a_src[CPT];
    int lds_off[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int row = tid / CPR + RSTEP * i, ch = tid % CPR;
        a_src[i] = p.in + ((long)(blk0 * TR + row) * K + ch * 8);
        lds_off[i] = row * (K * 2) + ((ch ^ swz_key(row)) * 16);
    }
    const int frag_key = swz_key(lrow);          // (a tile's second row block: 32 more rows leave the key unchanged)
    const int frag_row_off = lrow * (K * 2);

    float range_max = 0.f;
    float4v ra[CPT][2];
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float* src = a_src[i] + (long)t * TR * K;
            ra[i][0] = *reinterpret_cast<const float4v*>(src);
            ra[i][1] = *reinterpret_cast<const float4v*>(src + 4);
        }
    };
    auto split_to = [&](int stage) {
        char* const stg = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float4v v0 = ra[i][0], v1 = ra[i][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) range_max = fmaxf(range_max, fmaxf(__builtin_fabsf(v0[e]), __builtin_fabsf(v1[e])));
            const half4 h0 = __builtin_convertvector(v0, half4), h1 = __builtin_convertvector(v1, half4);          // round to nearest even
            const half4 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, float4v), half4);            // v - hi is exact in fp32
            const half4 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, float4v), half4);
            *reinterpret_cast<half8*>(stg + lds_off[i]) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<half8*>(stg + PLANE + lds_off[i]) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };

    for (int slab = slab0; slab < nslab; slab += slab_step) {
        const int n0 = slab * 256 + 32 * wave;          // first channel of this wave
        // weights of this wave's 32 channels as MFMA first operands: lane -> (channel n0 + lane % 32, k = 16 ks + 8 (lane / 32) .. + 8)
        half8 wh[KS], wl[KS];
        {
            const long wrow = (long)(n0 + lrow) * p.Kpad + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wh[ks] = *reinterpret_cast<const half8*>(p.w_hi + wrow + 16 * ks);
                wl[ks] = *reinterpret_cast<const half8*>(p.w_lo + wrow + 16 * ks);
            }
        }
        // (the previous slab's last barrier is behind every wave's last read of these)
        if (tid < 256) vec_scale[tid] = p.wscale ? p.wscale[slab * 256 + tid] : 1.f;
        else vec_bias[tid - 256] = p.bias ? p.bias[slab * 256 + tid - 256] : 0.f;

        // residual / output of this lane's row in tile 0: channels n0 + 16 g + 8 hi + [0, 8)
        const long col = n0 + 8 * hi;
        const float* r_src = HAS_RES ? p.res + ((long)(blk0 * TR + lrow) * p.Cout + col) : nullptr;
        float* o_dst = p.out + (long)(blk0 * TR + lrow) * p.ldc + col;
        const int vec_off = 32 * wave + 8 * hi;

        fetch(0);
        split_to(0);
        __syncthreads();

        float16v acc[RB];
        float4v rr[RB][2][2];
        auto load_res = [&](int t) {
            if (HAS_RES) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float* r = r_src + ((long)t * TR + 32 * rb) * p.Cout;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        rr[rb][g][0] = *reinterpret_cast<const float4v*>(r + 16 * g);
                        rr[rb][g][1] = *reinterpret_cast<const float4v*>(r + 16 * g + 4);
                    }
                }
            }
        };
        auto products = [&](int t) {
            const char* const stg = smem + (t & 1) * STAGE;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
            // the fragments of K step ks + 1 are requested before the products of step ks (left to itself the compiler reads each
            // fragment right in front of its first use -- `ds_read; s_waitcnt lgkmcnt(0); v_mfma` once per K step)
            half8 ah[2][RB], al[2][RB];
            auto frag = [&](int ks) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int off = rb * 32 * (K * 2) + frag_row_off + (((2 * ks + hi) ^ frag_key) * 16);
                    ah[ks & 1][rb] = *reinterpret_cast<const half8*>(stg + off);
                    al[ks & 1][rb] = *reinterpret_cast<const half8*>(stg + PLANE + off);
                }
            };
            frag(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) frag(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                // the two small terms first, then the leading one: f32x3_igemm_kernel's order
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], al[ks & 1][rb], acc[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], ah[ks & 1][rb], acc[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], ah[ks & 1][rb], acc[rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // epilogue from the accumulator layout.  acc[4 r4 + r] = channel 8 r4 + 4 hi + r of row lane % 32; one half-wave exchange per
        // register pair (r4, r4 + 1) leaves lane < 32 with channels 16 g + [0, 8) and lane >= 32 with 16 g + 8 + [0, 8)
        auto finish = [&](int t) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float* const o = o_dst + ((long)t * TR + 32 * rb) * p.ldc;
                unsigned int u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float f = acc[rb][r];
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
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4v sc = *reinterpret_cast<const float4v*>(vec_scale + vec_off + 16 * g + 4 * h);
                        const float4v bi = *reinterpret_cast<const float4v*>(vec_bias + vec_off + 16 * g + 4 * h);
                        float4v v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(u[8 * g + 4 * h + e]);
                        v *= sc;                      // a power of two: exact
                        v += bi;
                        if (HAS_RES) v += rr[rb][g][h];
                        if (ACT == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        } else if (ACT == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        *reinterpret_cast<float4v*>(o + 16 * g + 4 * h) = v;
                    }
                }
            }
        };

        // (Running the two waves of a SIMD in opposite phase -- waves 4-7 finish tile t - 1 while waves 0-3 multiply tile t, and vice
        // versa -- was built and measured: 0-4 % on these shapes, and the half-step loop it needs costs the straight-line loop below
        // 20 % on res4 conv3: profiles/r06v_f32_wstat_antiphase.txt.  Not kept.)
        for (int t = 0; t < T; ++t) {
            const bool more = t + 1 < T;          // workgroup-uniform
            if (more) fetch(t + 1);               // in flight under this tile's MFMAs
            load_res(t);
            products(t);
            finish(t);
            if (more) split_to((t + 1) & 1);
            __syncthreads();                      // tile t + 1 visible; nobody reads tile t any more
        }
    }
    // an activation beyond the fp16 range became inf in its hi part where fp32 arithmetic would not: reported, never silent (as
    // f32x3_igemm_kernel; one test per thread and launch instead of one per chunk)
    if (p.range_flag && range_max > 65504.f) atomicOr(p.range_flag, 1);
}

template <int K, int RB, bool HAS_RES, int ACT>
int f32ws_launch_k(const F32GemmParams& p, hipStream_t s) {
    constexpr int smem = F32WsSmem<K, RB>::kBytes;
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&f32x3_wstat_kernel<K, RB, HAS_RES, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        mark_on_device(attr_set);
    }
    hipLaunchKernelGGL((f32x3_wstat_kernel<K, RB, HAS_RES, ACT>), dim3(256), dim3(512), smem, s, p, p.Cout / 256);
    LAUNCH_CHECK();
    return DVID_OK;
}

template <int K, int RB>
int f32ws_launch_v(const F32GemmParams& p, hipStream_t s) {
    if (p.res_mode == 1) {
        if (p.relu == 1) return f32ws_launch_k<K, RB, true, 1>(p, s);
        if (p.relu == 0) return f32ws_launch_k<K, RB, true, 0>(p, s);
        return DVID_ERR_UNSUPPORTED;
    }
    if (p.relu == 2) return f32ws_launch_k<K, RB, false, 2>(p, s);
    return p.relu == 1 ? f32ws_launch_k<K, RB, false, 1>(p, s) : f32ws_launch_k<K, RB, false, 0>(p, s);
}

}  // namespace

// the layer type fits: split operands present, 1x1 / linear over contiguous rows, K in {128, 256}, N a multiple of 256 with the slab
// count dividing (or a multiple of) the workgroups of an XCD, bias / ReLU / GELU / same-shape residual, 16-byte aligned rows
bool dvid_f32_wstat_supported(const F32GemmParams& p) {
    if (!p.w_hi || !p.w_lo) return false;
    if (p.KH != 1 || p.KW != 1 || p.pad != 0 || p.stride != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin != p.Kpad || p.K != p.Kpad || (p.Kpad != 64 && p.Kpad != 128 && p.Kpad != 256)) return false;
    if (p.Cout % 256 || (p.ldc & 3)) return false;
    const int ns = p.Cout / 256;
    if (!(ns <= 32 ? (32 % ns == 0 || ns == 3 || ns == 6) : ns % 32 == 0)) return false;
    if (p.res_mode > 1 || p.relu > 2 || (p.relu == 2 && p.res_mode)) return false;
    return true;
}

// ... and the launch is large enough for the persistent workgroups (the rule of csrc/wstat.hip: at least 8 row blocks per weight load)
bool dvid_f32_wstat_preferred(const F32GemmParams& p) {
    if (!dvid_f32_wstat_supported(p)) return false;
    constexpr int kMinBlocks = 8;
    const int ns = p.Cout / 256;
    const long blocks_per_xcd = ((long)p.M + 31) / 32 / 8;
    const long per_wg = ns <= 32 ? blocks_per_xcd / (32 / ns) : blocks_per_xcd;
    return per_wg >= kMinBlocks;
}

// rows per tile of the variant a layer runs on (K = 64: two 32-row blocks, so that every thread has a chunk of the tile to split)
int dvid_f32_wstat_tile_rows(const F32GemmParams& p) { return p.Kpad == 64 ? 64 : 32; }

// Whole tiles only; the caller (dvid_f32_igemm_launch) hands a ragged tail to the tiled kernel -- same values.
int dvid_f32_wstat_launch_tiles(const F32GemmParams& p, hipStream_t s) {
    if (!dvid_f32_wstat_supported(p) || p.M % dvid_f32_wstat_tile_rows(p)) return DVID_ERR_UNSUPPORTED;
    if (p.Kpad == 64) return f32ws_launch_v<64, 2>(p, s);
    return p.Kpad == 128 ? f32ws_launch_v<128, 1>(p, s) : f32ws_launch_v<256, 1>(p, s);
}
