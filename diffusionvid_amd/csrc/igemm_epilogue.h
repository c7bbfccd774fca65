// Shared igemm epilogue: one thread finishes 8 consecutive output channels of one output row from the
// fp32 LDS tile: + bias, + residual (same-shape or nearest-x2-upsampled FPN top-down), ReLU, store
// fp16/fp32 with 16-byte accesses where the layout allows.
#pragma once
#include "common.h"
#include "kernels.h"

// `pre`: residual vector already fetched by the caller (res_mode 1, fp16, Cout % 8 == 0) when use_pre.
__device__ __forceinline__ void igemm_store_row8(const IgemmParams& p, const float* __restrict__ cs, int m, int n,
                                                 const float (&bias8)[8], bool use_pre = false,
                                                 half8 pre = half8{0, 0, 0, 0, 0, 0, 0, 0}) {
    float v[8];
    const float4v lo = *reinterpret_cast<const float4v*>(cs);
    const float4v hi = *reinterpret_cast<const float4v*>(cs + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = lo[e] + bias8[e];
        v[e + 4] = hi[e] + bias8[e + 4];
    }
    const bool vec_ok = (p.Cout & 7) == 0;
    if (use_pre) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)pre[e];
    } else if (p.res_mode) {
        long ridx;
        if (p.res_mode == 1) {
            ridx = (long)m * p.Cout + n;
        } else {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int img = t / p.Ho;
            ridx = ((long)(img * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + n;
        }
        if (p.res_f32) {
            const float* rp = reinterpret_cast<const float*>(p.res) + ridx;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.Cout) v[e] += rp[e];
        } else if (vec_ok) {
            const half8 rv = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.res) + ridx);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        } else {
            const half_t* rp = reinterpret_cast<const half_t*>(p.res) + ridx;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.Cout) v[e] += (float)rp[e];
        }
    }
    if (p.relu == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.relu == 2) {      // exact GELU (nn.GELU default), Swin MLP
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    }
    const long oidx = (long)m * p.ldc + n;
    if (p.out_f32) {
        float* op = reinterpret_cast<float*>(p.out) + oidx;
        if (vec_ok && (p.ldc & 3) == 0) {
            *reinterpret_cast<float4v*>(op) = (float4v){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4v*>(op + 4) = (float4v){v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.Cout) op[e] = v[e];
        }
    } else {
        half_t* op = reinterpret_cast<half_t*>(p.out) + oidx;
        if (vec_ok && (p.ldc & 7) == 0) {
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
            *reinterpret_cast<half8*>(op) = hv;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.Cout) op[e] = (half_t)v[e];
        }
    }
}

__device__ __forceinline__ int igemm_xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
