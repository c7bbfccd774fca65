// Shared device/host helpers for libdvid_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
