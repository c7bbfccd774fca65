// Fused DynamicConv instance interaction (box_head.py:692-704), one workgroup per box:
//
//   F1 = relu(LN64 (roi[49x256] . P1[256x64]))      bmm #1 + norm1 + ReLU
//   F2 = relu(LN256(F1 [49x64]  . P2[64x256]))      bmm #2 + norm2 + ReLU     -> out[49][256] fp16
//
// `params` is the dynamic_layer output for this box in the REPACKED order produced by the
// runtime's weight repack: P1T[j][c] (64x256) then P2T[c][j] (256x64), i.e. both are
// "[N][K], K contiguous" MFMA B operands that each wave pulls straight from global memory
// (every parameter is read exactly once per box).  The 49x256 RoI tile (zero-padded to 64 rows)
// is staged once in LDS (528-byte pitch: conflict-free ds_read_b128 A fragments); the N
// dimension of both products is split over the 4 waves, so LayerNorm row statistics are
// combined across waves through a tiny LDS buffer (mean first, then centred variance, fp32).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NP = 49;          // 7x7 bins
constexpr int D = 256;          // hidden dim
constexpr int DD = 64;          // dynamic dim
constexpr int A_PITCH = D + 8;  // halves
constexpr int H_PITCH = DD + 8;

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

__global__ __launch_bounds__(256) void dynconv_kernel(const half_t* __restrict__ roi, const half_t* __restrict__ params,
                                                       const float* __restrict__ g1, const float* __restrict__ b1,
                                                       const float* __restrict__ g2, const float* __restrict__ b2,
                                                       half_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);                     // [64][A_PITCH]; later the output tile
    half_t* Hs = As + 64 * A_PITCH;                                   // [64][H_PITCH]
    float* red = reinterpret_cast<float*>(Hs + 64 * H_PITCH);         // [64 rows][4 waves]

    const int box = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const half_t* roi_b = roi + (long)box * NP * D;
    const half_t* p1t = params + (long)box * (2 * D * DD);
    const half_t* p2t = p1t + D * DD;

    // ---- stage the RoI tile (rows 49..63 zero) ----------------------------------------
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 64 * 32; i += 256) {
        const int r = i >> 5, cv = i & 31;
        const half8 v = *reinterpret_cast<const half8*>(roi_b + (long)(r < NP ? r : 0) * D + cv * 8);
        *reinterpret_cast<half8*>(As + r * A_PITCH + cv * 8) = (r < NP) ? v : zero8;
    }
    // B fragments of bmm #1: this wave owns output columns [wave*16, +16)
    half8 bf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        bf[ks] = *reinterpret_cast<const half8*>(p1t + (wave * 16 + l15) * D + ks * 32 + l4 * 8);
    __syncthreads();

    // ---- bmm #1: [64x256] x [256x16] per wave ----------------------------------------
    float4v acc1[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc1[mt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half8 af = *reinterpret_cast<const half8*>(As + (mt * 16 + l15) * A_PITCH + ks * 32 + l4 * 8);
            acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[ks], acc1[mt], 0, 0, 0);
        }
    }
    // prefetch this wave's bmm #2 B fragments (columns [wave*64, +64)), first K half
    half8 b2f[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            b2f[nt][ks] = *reinterpret_cast<const half8*>(p2t + (wave * 64 + nt * 16 + l15) * DD + ks * 32 + l4 * 8);

    // ---- LayerNorm(64) + ReLU over rows; element (row = mt*16 + l4*4 + r, col = wave*16 + l15)
    float mean[4][4], rstd[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = group16_sum(acc1[mt][r]);
            if (l15 == 0) red[(mt * 16 + l4 * 4 + r) * 4 + wave] = s;
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l4 * 4 + r) * 4);
            mean[mt][r] = (t[0] + t[1] + t[2] + t[3]) * (1.f / DD);
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = acc1[mt][r] - mean[mt][r];
            const float s = group16_sum(c * c);
            if (l15 == 0) red[(mt * 16 + l4 * 4 + r) * 4 + wave] = s;
        }
    __syncthreads();
    {
        const float gg = g1[wave * 16 + l15], bb = b1[wave * 16 + l15];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l4 * 4 + r) * 4);
                rstd[mt][r] = rsqrtf((t[0] + t[1] + t[2] + t[3]) * (1.f / DD) + 1e-5f);
                const float y = fmaxf((acc1[mt][r] - mean[mt][r]) * rstd[mt][r] * gg + bb, 0.f);
                Hs[(mt * 16 + l4 * 4 + r) * H_PITCH + wave * 16 + l15] = (half_t)y;
            }
    }
    __syncthreads();   // Hs complete; As (RoI) is dead from here on; red reusable

    // ---- bmm #2: [64x64] x [64x64] per wave --------------------------------------------
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
                acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b2f[nt][ks], acc2[mt][nt], 0, 0, 0);
        }
    }
    // ---- LayerNorm(256) + ReLU; element (row = mt*16 + l4*4 + r, col = wave*64 + nt*16 + l15)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = group16_sum(acc2[mt][0][r] + acc2[mt][1][r] + acc2[mt][2][r] + acc2[mt][3][r]);
            if (l15 == 0) red[(mt * 16 + l4 * 4 + r) * 4 + wave] = s;
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l4 * 4 + r) * 4);
            mean[mt][r] = (t[0] + t[1] + t[2] + t[3]) * (1.f / D);
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float c = acc2[mt][nt][r] - mean[mt][r];
                s += c * c;
            }
            s = group16_sum(s);
            if (l15 == 0) red[(mt * 16 + l4 * 4 + r) * 4 + wave] = s;
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4v t = *reinterpret_cast<const float4v*>(red + (mt * 16 + l4 * 4 + r) * 4);
            rstd[mt][r] = rsqrtf((t[0] + t[1] + t[2] + t[3]) * (1.f / D) + 1e-5f);
        }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = wave * 64 + nt * 16 + l15;
        const float gg = g2[col], bb = b2[col];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = fmaxf((acc2[mt][nt][r] - mean[mt][r]) * rstd[mt][r] * gg + bb, 0.f);
                As[(mt * 16 + l4 * 4 + r) * A_PITCH + col] = (half_t)y;
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
    hipLaunchKernelGGL(dynconv_kernel, dim3(rows), dim3(256), kSmem, s, roi, params, g1, b1, g2, b2, out);
    LAUNCH_CHECK();
    return DVID_OK;
}
