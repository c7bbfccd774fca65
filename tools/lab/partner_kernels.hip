// Synthetic co-residents for the res4 kernels (round 5, tools/lab/coresidency_probe.py).
//
// The review's co-residency proposal: run an HBM-paced layer (res4 conv1, conv3 + residual) of one sub-batch chain beside the MFMA-paced
// conv2 of the other chain on the same CUs.  Before any kernel is re-cut for that, the question is what a PERFECT partner would gain:
// these two kernels are as small as a partner can be (no LDS, <= 64 VGPRs, 4 waves per workgroup), so they fit beside the product kernels
// as they are today (conv3x3_halo 216 VGPRs x 2 waves per SIMD leave 80 registers per SIMD lane; 133 KB of LDS leave 27 KB):
//   stream_copy   -- a pure HBM stream (16-byte loads, 8 in flight per lane, grid-stride), the ideal "HBM-paced" neighbour
//   mfma_spin     -- back-to-back v_mfma_f32_32x32x16_f16 on two accumulators, the ideal "MFMA-paced" neighbour
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/partner_kernels.hip -o tools/lab/libpartner.so
#include <hip/hip_runtime.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 8) void stream_copy_kernel(const float4v* __restrict__ src, float4v* __restrict__ dst, long n16, int write) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 7 * stride < n16; i += 8 * stride) {
        float4v v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
        if (write) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    for (; i < n16; i += stride) {
        const float4v v = src[i];
        if (write) dst[i] = v; else acc += v;
    }
    if (!write && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) dst[0] = acc;
}

__global__ __launch_bounds__(256, 8) void mfma_spin_kernel(long iters, float* sink) {
    float16v a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    half8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x[e] = (_Float16)(0.001f * (threadIdx.x + e));
        y[e] = (_Float16)(0.002f * (threadIdx.x ^ e));
    }
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 1.2345e-30f) sink[0] = s;
}

extern "C" int partner_copy(void* stream, const void* src, void* dst, long bytes, int write, int grid) {
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4v*>(src),
                       reinterpret_cast<float4v*>(dst), bytes / 16, write);
    return (int)hipGetLastError();
}
// `iters` x 16 MFMAs of 32x32x16 per wave, 4 waves per workgroup: 2 * 32 * 32 * 16 * 16 * 4 = 2.1 MFLOP per iteration and workgroup
extern "C" int partner_mfma(void* stream, long iters, int grid, float* sink) {
    hipLaunchKernelGGL(mfma_spin_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), iters, sink);
    return (int)hipGetLastError();
}
