// Fused DynamicConv instance interaction (box_head.py:692-704), one workgroup per box:
//
//   F1 = relu(LN64 (roi[49x256] . P1[256x64]))      bmm #1 + norm1 + ReLU
//   F2 = relu(LN256(F1 [49x64]  . P2[64x256]))      bmm #2 + norm2 + ReLU     -> out[49][256] fp16
//
// `params` is the dynamic_layer output for this box in the REPACKED order produced by the
// runtime's weight repack: P1T[j][c] (64x256) then P2T[c][j] (256x64), i.e. both are
// "[N][K], K contiguous" MFMA B operands that each wave pulls straight from global memory
// (every parameter is read exactly once per box).  The 49x256 RoI tile (zero-padded to 64 rows)
// is staged once in LDS (528-byte pitch: conflict-free ds_read_b128 fragments); the N
// dimension of both products is split over the 4 waves, so LayerNorm row statistics are
// combined across waves through a tiny LDS buffer (mean first, then centred variance, fp32).
//
// FUSED_ROI (round 6): the RoI tile is not read from memory but GATHERED here -- the per-box walk of multi-level RoIAlignV2
// (csrc/roi_taps.h, the arithmetic of csrc/roialign.hip bit for bit) writes its 49 bins straight into the LDS image the first product
// reads.  The separate launches move the fp16 tile through HBM twice (25 KB out, 25 KB in per box) and run an L1-bound gather and an
// HBM-bound stream one after the other; here the first product's 32 KB of parameters are requested BEFORE the gather, and with three
// workgroups per CU one box's taps (L1 path) overlap another's parameter stream (HBM).  The unfused pair stays for the passes that need
// the tile's mean over the bins before the self-attention (box_head.py:509-510: a head without incoming proposal features).
#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "roi_taps.h"

namespace {

constexpr int NP = 49;          // 7x7 bins
constexpr int D = 256;          // hidden dim
constexpr int DD = 64;          // dynamic dim
constexpr int A_PITCH = D + 8;  // halves
constexpr int H_PITCH = DD + 8;

// sum over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48)
__device__ __forceinline__ float groups4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Both products are computed TRANSPOSED (the per-box parameters are the MFMA's first operand): a lane then holds 4 consecutive
// channels of ONE tile row instead of 4 rows of one channel, so a LayerNorm statistic is 3-15 in-lane adds + 2 shuffles per row tile
// (it was 4 shuffles per accumulator register: 64 per pass), and the fp16 results leave as 8-byte LDS writes (they were 2-byte ones).
// Same products, same K order; only the order of the fp32 sums inside the LayerNorm statistics differs from the row-major form.
template <bool FUSED_ROI>
__global__ __launch_bounds__(256, 3) void dynconv_kernel(const half_t* __restrict__ roi, const half_t* __restrict__ params,
                                                       const float* __restrict__ g1, const float* __restrict__ b1,
                                                       const float* __restrict__ g2, const float* __restrict__ b2,
                                                       half_t* __restrict__ out, RoiLevels lv, const float* __restrict__ boxes,
                                                       int boxes_per_img, int nbox) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);                     // [64][A_PITCH]; later the output tile
    half_t* Hs = As + 64 * A_PITCH;                                   // [64][H_PITCH]
    float* red = reinterpret_cast<float*>(Hs + 64 * H_PITCH);         // [64 rows][4 waves]

    // (fused: an XCD takes one contiguous run of boxes, i.e. whole images, as csrc/roialign.hip -- the boxes that gather from one image's
    // pyramid meet in one L2)
    const int box = FUSED_ROI ? igemm_xcd_remap((int)blockIdx.x, nbox) : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const half_t* p1t = params + (long)box * (2 * D * DD);
    const half_t* p2t = p1t + D * DD;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 bf[8];          // parameter fragments of bmm #1: this wave owns output channels [wave*16, +16)
    half8 b2f[4][2];      // ... of bmm #2: channels [wave*64, +64)

    if (FUSED_ROI) {
        // both products' parameters (64 KB per box) are requested first: they stream in from HBM under the gather, which runs on the L1 path
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bf[ks] = *reinterpret_cast<const half8*>(p1t + (wave * 16 + l15) * D + ks * 32 + l4 * 8);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b2f[nt][ks] = *reinterpret_cast<const half8*>(p2t + (wave * 64 + nt * 16 + l15) * DD + ks * 32 + l4 * 8);
        // ---- gather the RoI tile into its LDS image (rows 49..63 zero) ------------------
        float macc[8];
        roi_taps::gather_box(lv, boxes, boxes_per_img, box, tid >> 5, tid & 31, macc,
                             [&](int p, half8 o) { *reinterpret_cast<half8*>(As + p * A_PITCH + (tid & 31) * 8) = o; });
        for (int i = tid; i < (64 - NP) * 32; i += 256) *reinterpret_cast<half8*>(As + (NP + (i >> 5)) * A_PITCH + (i & 31) * 8) = zero8;
    } else {
        // ---- stage the RoI tile (rows 49..63 zero) ----------------------------------------
        const half_t* roi_b = roi + (long)box * NP * D;
        for (int i = tid; i < 64 * 32; i += 256) {
            const int r = i >> 5, cv = i & 31;
            const half8 v = *reinterpret_cast<const half8*>(roi_b + (long)(r < NP ? r : 0) * D + cv * 8);
            *reinterpret_cast<half8*>(As + r * A_PITCH + cv * 8) = (r < NP) ? v : zero8;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bf[ks] = *reinterpret_cast<const half8*>(p1t + (wave * 16 + l15) * D + ks * 32 + l4 * 8);
    }
    __syncthreads();

    // ---- bmm #1 (transposed): acc1[mt][r] = F1[row mt*16 + l15][channel wave*16 + 4*l4 + r] -----------------------------
    float4v acc1[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc1[mt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half8 af = *reinterpret_cast<const half8*>(As + (mt * 16 + l15) * A_PITCH + ks * 32 + l4 * 8);
            acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks], af, acc1[mt], 0, 0, 0);
        }
    }
    // prefetch this wave's bmm #2 parameter fragments (channels [wave*64, +64)); fused: they were requested ahead of the gather
    if (!FUSED_ROI) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b2f[nt][ks] = *reinterpret_cast<const half8*>(p2t + (wave * 64 + nt * 16 + l15) * DD + ks * 32 + l4 * 8);
    }

    // ---- LayerNorm(64) + ReLU over rows ------------------------------------------------------------------------------------
    float mean[4], rstd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float s = groups4_sum((acc1[mt][0] + acc1[mt][1]) + (acc1[mt][2] + acc1[mt][3]));
        if (l4 == 0) red[(mt * 16 + l15) * 4 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l15) * 4);
        mean[mt] = (t[0] + t[1] + t[2] + t[3]) * (1.f / DD);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = acc1[mt][r] - mean[mt];
            s += c * c;
        }
        s = groups4_sum(s);
        if (l4 == 0) red[(mt * 16 + l15) * 4 + wave] = s;
    }
    __syncthreads();
    {
        const float4v gg = *reinterpret_cast<const float4v*>(g1 + wave * 16 + l4 * 4), bb = *reinterpret_cast<const float4v*>(b1 + wave * 16 + l4 * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l15) * 4);
            rstd[mt] = rsqrtf((t[0] + t[1] + t[2] + t[3]) * (1.f / DD) + 1e-5f);
            half4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (half_t)fmaxf((acc1[mt][r] - mean[mt]) * rstd[mt] * gg[r] + bb[r], 0.f);
            *reinterpret_cast<half4*>(Hs + (mt * 16 + l15) * H_PITCH + wave * 16 + l4 * 4) = y;
        }
    }
    __syncthreads();   // Hs complete; As (RoI) is dead from here on; red reusable

    // ---- bmm #2 (transposed): acc2[mt][nt][r] = F2[row mt*16 + l15][channel wave*64 + nt*16 + 4*l4 + r] -------------------
    float4v acc2[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc2[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half8 af = *reinterpret_cast<const half8*>(Hs + (mt * 16 + l15) * H_PITCH + ks * 32 + l4 * 8);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2f[nt][ks], af, acc2[mt][nt], 0, 0, 0);
        }
    }
    // ---- LayerNorm(256) + ReLU ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s += (acc2[mt][nt][0] + acc2[mt][nt][1]) + (acc2[mt][nt][2] + acc2[mt][nt][3]);
        s = groups4_sum(s);
        if (l4 == 0) red[(mt * 16 + l15) * 4 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l15) * 4);
        mean[mt] = (t[0] + t[1] + t[2] + t[3]) * (1.f / D);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c = acc2[mt][nt][r] - mean[mt];
                s += c * c;
            }
        s = groups4_sum(s);
        if (l4 == 0) red[(mt * 16 + l15) * 4 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l15) * 4);
        rstd[mt] = rsqrtf((t[0] + t[1] + t[2] + t[3]) * (1.f / D) + 1e-5f);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = wave * 64 + nt * 16 + l4 * 4;
        const float4v gg = *reinterpret_cast<const float4v*>(g2 + col), bb = *reinterpret_cast<const float4v*>(b2 + col);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            half4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (half_t)fmaxf((acc2[mt][nt][r] - mean[mt]) * rstd[mt] * gg[r] + bb[r], 0.f);
            *reinterpret_cast<half4*>(As + (mt * 16 + l15) * A_PITCH + col) = y;
        }
    }
    __syncthreads();
    // ---- coalesced 16-byte stores of the 49 valid rows ---------------------------------
    half_t* out_b = out + (long)box * NP * D;
    for (int i = tid; i < NP * 32; i += 256) {
        const int r = i >> 5, cv = i & 31;
        *reinterpret_cast<half8*>(out_b + (long)r * D + cv * 8) = *reinterpret_cast<const half8*>(As + r * A_PITCH + cv * 8);
    }
}

constexpr int kSmem = 64 * A_PITCH * 2 + 64 * H_PITCH * 2 + 64 * 4 * 4;

}  // namespace

int dvid_dynconv_launch(const half_t* roi, const half_t* params, const float* g1, const float* b1, const float* g2,
                        const float* b2, half_t* out, int rows, hipStream_t s) {
    if (rows == 0) return DVID_OK;
    hipLaunchKernelGGL(dynconv_kernel<false>, dim3(rows), dim3(256), kSmem, s, roi, params, g1, b1, g2, b2, out, RoiLevels{}, nullptr, 1, rows);
    LAUNCH_CHECK();
    return DVID_OK;
}

// RoIAlign + DynamicConv as one launch: the tile of box b of image b / boxes_per_img is gathered from the pyramid `lv` (see the header)
int dvid_dynconv_roi_launch(const RoiLevels& lv, int channels, const float* boxes, int n_img, int boxes_per_img, const half_t* params,
                            const float* g1, const float* b1, const float* g2, const float* b2, half_t* out, hipStream_t s) {
    if (channels != 256) return DVID_ERR_UNSUPPORTED;
    const int rows = n_img * boxes_per_img;
    if (rows == 0) return DVID_OK;
    hipLaunchKernelGGL(dynconv_kernel<true>, dim3(rows), dim3(256), kSmem, s, nullptr, params, g1, b1, g2, b2, out, lv, boxes, boxes_per_img, rows);
    LAUNCH_CHECK();
    return DVID_OK;
}
