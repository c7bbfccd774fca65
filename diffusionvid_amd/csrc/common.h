// Shared device/host helpers for libdvid_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define DVID_OK 0
#define DVID_ERR_ARG 1
#define DVID_ERR_HIP 2
#define DVID_ERR_UNSUPPORTED 3
#define DVID_ERR_STATE 4

#define WAVE 64

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            fprintf(stderr, "[dvid_hip] %s:%d: %s -> %s\n", __FILE__, __LINE__, #expr,     \
                    hipGetErrorString(_e));                                                \
            return DVID_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a function ON a device: a process that runs models on
// several devices (the detector scopes the device per call) has to set it once per device, not once per process.
// `mask` holds one bit per device ordinal.  first_on_device: the attribute has not been applied on the current device yet (a test
// only); mark_on_device: it has -- called AFTER hipFuncSetAttribute succeeded, so a second host thread never launches ahead of the
// attribute (it applies it again, which is harmless) and a failed call is retried by the next launch.
static inline unsigned long long device_bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return 1ull << (dev & 63);
}
static inline bool first_on_device(const std::atomic<unsigned long long>& mask) { return (mask.load(std::memory_order_acquire) & device_bit()) == 0; }
static inline void mark_on_device(std::atomic<unsigned long long>& mask) { mask.fetch_or(device_bit(), std::memory_order_release); }

static inline long ceil_div(long a, long b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Exact (erf) GELU, nn.GELU's default, as x * Phi(x) with Phi through the complementary error function:
//   erfc(z) = t exp(-z^2 + P(t)),  t = 1 / (1 + z / 2),  z = |x| / sqrt(2) >= 0     (W. H. Press et al., Numerical Recipes, erfcc:
//   fractional error < 1.2e-7 for every z),   Phi(-|x|) = erfc(z) / 2,   Phi(|x|) = 1 - erfc(z) / 2.
// 20 branch-free instructions (one v_rcp_f32, one v_exp_f32) against ~45 with two divergent branches for 0.5 x (1 + erff(x / sqrt 2)):
// Swin's fc1 layers apply it to 4C values per token and their launches were bound by exactly those instructions
// (profiles/r02e_swinb_kernel_stats.txt).  It is also the more accurate form where x < 0: 1 + erf(..) cancels there (relative error
// up to 0.6 for x < -5), erfc does not (2.7e-6 over [-14, 14]; 0.011 % of the results differ from the exactly rounded fp16 value,
// 2.8 % with the erf form).  Explicit fma / no contraction: every call site rounds alike.
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.5f, z, 1.f));
    float p = 0.17087277f;
    p = __builtin_fmaf(p, t, -0.82215223f);
    p = __builtin_fmaf(p, t, 1.48851587f);
    p = __builtin_fmaf(p, t, -1.13520398f);
    p = __builtin_fmaf(p, t, 0.27886807f);
    p = __builtin_fmaf(p, t, -0.18628806f);
    p = __builtin_fmaf(p, t, 0.09678418f);
    p = __builtin_fmaf(p, t, 0.37409196f);
    p = __builtin_fmaf(p, t, 1.00002368f);
    p = __builtin_fmaf(p, t, -1.26551223f);
    const float e = t * __builtin_amdgcn_exp2f(__builtin_fmaf(-z, z, p) * 1.4426950408889634f);      // erfc(z); exp(a) = 2^(a log2 e)
    const float tail = 0.5f * e;                                // Phi(-|x|)
    const float phi = x >= 0.f ? __builtin_fmaf(-0.5f, e, 1.f) : tail;
    return x * phi;
}
// Two values at a time: the same operations in the same order as gelu_erf (so the same bits), with the multiplies and fused
// multiply-adds as packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot) -- the epilogue that
// applies it is bound by exactly these instructions.
__device__ __forceinline__ float2v gelu_erf2(float2v x) {
#pragma clang fp contract(off)
    const float2v z = __builtin_elementwise_abs(x) * (float2v){0.70710678118654752440f, 0.70710678118654752440f};
    const float2v d = __builtin_elementwise_fma((float2v){0.5f, 0.5f}, z, (float2v){1.f, 1.f});
    const float2v t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    float2v p = {0.17087277f, 0.17087277f};
    p = __builtin_elementwise_fma(p, t, (float2v){-0.82215223f, -0.82215223f});
    p = __builtin_elementwise_fma(p, t, (float2v){1.48851587f, 1.48851587f});
    p = __builtin_elementwise_fma(p, t, (float2v){-1.13520398f, -1.13520398f});
    p = __builtin_elementwise_fma(p, t, (float2v){0.27886807f, 0.27886807f});
    p = __builtin_elementwise_fma(p, t, (float2v){-0.18628806f, -0.18628806f});
    p = __builtin_elementwise_fma(p, t, (float2v){0.09678418f, 0.09678418f});
    p = __builtin_elementwise_fma(p, t, (float2v){0.37409196f, 0.37409196f});
    p = __builtin_elementwise_fma(p, t, (float2v){1.00002368f, 1.00002368f});
    p = __builtin_elementwise_fma(p, t, (float2v){-1.26551223f, -1.26551223f});
    const float2v a = __builtin_elementwise_fma(-z, z, p);
    const float2v b = a * (float2v){1.4426950408889634f, 1.4426950408889634f};
    const float2v e = t * (float2v){__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
    const float2v tail = (float2v){0.5f, 0.5f} * e;
    const float2v head = __builtin_elementwise_fma((float2v){-0.5f, -0.5f}, e, (float2v){1.f, 1.f});
    const float2v phi = {x[0] >= 0.f ? head[0] : tail[0], x[1] >= 0.f ? head[1] : tail[1]};
    return x * phi;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
