// Global-memory pruning: pairwise L2 distances + greedy farthest-point sampling + row gather
// (diffusion_det.py:841-896, replacing mega_core/csrc/cuda/fps.cu:25-142).
//
// FPS is latency-bound (m-1 dependent arg-max steps); one workgroup, `temp` lives in LDS, the
// distance row of the last pick streams from L2.  Ties are resolved exactly as the reference
// CUDA kernel does for its block size `bs` (largest power of two <= n, capped at 1024): the
// candidate with the smallest (bitreverse(k % bs), k / bs) wins -- see oracle/memory.py.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned long long u64;

// D[i][j] = sqrt(max(|xi|^2 + |xj|^2 - 2 xi.xj, 0)) -- torch.cdist's matmul formulation, fp32.
constexpr int CT = 64, CK = 16;
__global__ __launch_bounds__(256) void cdist_kernel(const float* __restrict__ x, int n, int d, float* __restrict__ dist) {
    __shared__ float As[CK][CT + 4];
    __shared__ float Bs[CK][CT + 4];
    __shared__ float na[CT], nb[CT];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * CT, j0 = blockIdx.x * CT;
    const int ty = tid >> 4, tx = tid & 15;  // 16x16 threads, 4x4 outputs each
    float acc[4][4] = {};
    float nrm = 0.f;  // threads 0..63: |x_{i0+tid}|^2 ; 64..127: |x_{j0+tid-64}|^2
    for (int k0 = 0; k0 < d; k0 += CK) {
        for (int t = tid; t < CT * CK; t += 256) {
            const int r = t / CK, c = t - r * CK;
            As[c][r] = (i0 + r < n && k0 + c < d) ? x[(long)(i0 + r) * d + k0 + c] : 0.f;
            Bs[c][r] = (j0 + r < n && k0 + c < d) ? x[(long)(j0 + r) * d + k0 + c] : 0.f;
        }
        __syncthreads();
        if (tid < 64) {
#pragma unroll
            for (int c = 0; c < CK; ++c) nrm += As[c][tid] * As[c][tid];
        } else if (tid < 128) {
#pragma unroll
            for (int c = 0; c < CK; ++c) nrm += Bs[c][tid - 64] * Bs[c][tid - 64];
        }
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            float a[4], b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = As[c][ty * 4 + e];
                b[e] = Bs[c][tx * 4 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[e][g] += a[e] * b[g];
        }
        __syncthreads();
    }
    if (tid < 64) na[tid] = nrm;
    else if (tid < 128) nb[tid - 64] = nrm;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = i0 + ty * 4 + e, j = j0 + tx * 4 + g;
            if (i < n && j < n) {
                const float d2 = na[ty * 4 + e] + nb[tx * 4 + g] - 2.f * acc[e][g];
                dist[(long)i * n + j] = sqrtf(fmaxf(d2, 0.f));
            }
        }
}

__device__ __forceinline__ unsigned bitrev_n(unsigned v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0u; }

__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ dist, int n, int m, int bs, int bs_bits,
                                                    int* __restrict__ idx) {
    extern __shared__ float temp[];          // [n]
    __shared__ u64 wbest[16];
    __shared__ int s_old;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned pmul = (unsigned)(n / bs + 2);
    for (int k = tid; k < n; k += blockDim.x) temp[k] = 1e10f;
    if (tid == 0) idx[0] = 0;
    int old = 0;
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const float* row = dist + (long)old * n;
        u64 best = 0;
        for (int k = tid; k < n; k += blockDim.x) {
            const float d2 = fminf(row[k], temp[k]);
            temp[k] = d2;
            const unsigned prio = bitrev_n((unsigned)k & (bs - 1), bs_bits) * pmul + (unsigned)k / bs;
            const u64 key = ((u64)__float_as_uint(d2) << 32) | (0xFFFFFFFFu - prio);   // d2 >= 0
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) wbest[wave] = best;
        __syncthreads();
        u64 b = wbest[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) b = wbest[w] > b ? wbest[w] : b;
        // the unique owner of the winning (value, priority) publishes its index
        for (int k = tid; k < n; k += blockDim.x) {
            const unsigned prio = bitrev_n((unsigned)k & (bs - 1), bs_bits) * pmul + (unsigned)k / bs;
            if ((0xFFFFFFFFu - prio) == (unsigned)b && __float_as_uint(temp[k]) == (unsigned)(b >> 32)) s_old = k;
        }
        __syncthreads();
        old = s_old;
        if (tid == 0) idx[j] = old;
    }
}

// The same greedy sweep with the running distances in registers (EPT points per thread, 256 threads) and the winner's index decoded
// from the winning key itself -- prio = bitrev(k % bs) * pmul + k / bs is invertible -- so a step is one row read, one wave
// reduction, ONE barrier (winner slots double-buffered) instead of two barriers and an owner search through LDS.  Same keys, same
// maximum: the same picks as fps_kernel (tests/test_gpu_kernels.py compares both with the thread-level emulation of fps.cu).
template <int EPT>
__global__ __launch_bounds__(256) void fps_reg_kernel(const float* __restrict__ dist, int n, int m, int bs, int bs_bits,
                                                       int* __restrict__ idx) {
    __shared__ u64 wbest[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned pmul = (unsigned)(n / bs + 2);
    float temp[EPT];
    unsigned pinv[EPT];
    int kc[EPT];                             // column of this lane's e-th point; past the end: the last column, result unused
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const unsigned k = (unsigned)(tid + 256 * e);
        temp[e] = 1e10f;
        pinv[e] = 0xFFFFFFFFu - (bitrev_n(k & (bs - 1), bs_bits) * pmul + k / bs);
        kc[e] = (int)k < n ? (int)k : n - 1;
    }
    if (tid == 0) idx[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float* row = dist + (long)old * n;
        u64 best = 0;
        float v[EPT];
        // every load of the step is in flight before the first is used (loads under per-point exec masks are issued and waited for one
        // after the other: 8 dependent memory latencies per step, 6.7 ms for 1800 -> 900 against 3.6 with the LDS kernel)
#pragma unroll
        for (int e = 0; e < EPT; ++e) v[e] = row[kc[e]];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float d2 = fminf(v[e], temp[e]);
            temp[e] = d2;
            const u64 key = tid + 256 * e < n ? (((u64)__float_as_uint(d2) << 32) | pinv[e]) : 0ull;          // d2 >= 0
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) wbest[j & 1][wave] = best;
        __syncthreads();
        u64 b = wbest[j & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = wbest[j & 1][w] > b ? wbest[j & 1][w] : b;
        const unsigned prio = 0xFFFFFFFFu - (unsigned)b;
        const unsigned br = prio / pmul, q = prio - br * pmul;
        old = (int)(q * (unsigned)bs + bitrev_n(br, bs_bits));
        if (tid == 0) idx[j] = old;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx, float* __restrict__ y, int m, int d) {
    const int dv = d >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)m * dv) return;
    const int r = i / dv, v = i - (long)r * dv;
    *reinterpret_cast<float4v*>(y + (long)r * d + v * 4) = *reinterpret_cast<const float4v*>(x + (long)idx[r] * d + v * 4);
}

}  // namespace

int dvid_cdist_launch(const float* x, int n, int d, float* dist, hipStream_t s) {
    if (n == 0) return DVID_OK;
    hipLaunchKernelGGL(cdist_kernel, dim3(ceil_div(n, CT), ceil_div(n, CT)), dim3(256), 0, s, x, n, d, dist);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_fps_launch(const float* dist, int n, int m, int bs_emul, int* idx, hipStream_t s) {
    if (m <= 0) return DVID_OK;
    if (n <= 0 || m > n || n * 4 > 96 * 1024) return DVID_ERR_ARG;
    int bs = bs_emul;
    if (bs <= 0) {  // fps.cu:11-15 opt_n_threads
        bs = 1;
        while (bs * 2 <= n && bs < 1024) bs *= 2;
    }
    int bits = 0;
    while ((1 << bits) < bs) ++bits;
    if ((1 << bits) != bs) return DVID_ERR_ARG;
    if (n <= 256 * 16) {          // the register kernel; larger inputs take the LDS form below
        const int ept = (n + 255) / 256;
        if (ept <= 4) hipLaunchKernelGGL(fps_reg_kernel<4>, dim3(1), dim3(256), 0, s, dist, n, m, bs, bits, idx);
        else if (ept <= 8) hipLaunchKernelGGL(fps_reg_kernel<8>, dim3(1), dim3(256), 0, s, dist, n, m, bs, bits, idx);
        else hipLaunchKernelGGL(fps_reg_kernel<16>, dim3(1), dim3(256), 0, s, dist, n, m, bs, bits, idx);
        LAUNCH_CHECK();
        return DVID_OK;
    }
    const size_t smem = (size_t)n * 4;
    static std::atomic<unsigned long long> attr{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        mark_on_device(attr);
    }
    hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), smem, s, dist, n, m, bs, bits, idx);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_gather_rows_launch(const float* x, const int* idx, float* y, int m, int d, hipStream_t s) {
    if (m == 0) return DVID_OK;
    if (d % 4) return DVID_ERR_ARG;
    const long total = (long)m * (d / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, idx, y, m, d);
    LAUNCH_CHECK();
    return DVID_OK;
}
