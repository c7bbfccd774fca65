// What do a workgroup barrier and a bare MFMA stream cost on gfx950?  (1 workgroup per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int NW, bool BAR, int NMFMA, int KIND>
__global__ __launch_bounds__(64 * NW) void k(int iters, float* out) {
    float16v acc[4];
    float4v acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) acc4[q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[q & 7], 0, 0, 0);
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    if (s == 12345.f) out[0] = s;
}

template <int NW, bool BAR, int NMFMA, int KIND>
void run(const char* tag, float* out) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int t = 0; t < 3; ++t) {
        hipEventRecord(a);
        k<NW, BAR, NMFMA, KIND><<<256, 64 * NW>>>(iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / iters;
    const double flop = (double)NMFMA * 32768.0 * NW * 256 * iters;
    printf("%-44s waves %d barrier %d mfma/iter %2d : %7.1f ns per iter  (%.0f TFLOP/s)\n", tag, NW, (int)BAR, NMFMA, ns, flop / (best * 1e-3) / 1e12);
}

int main() {
    float* out; (void)hipMalloc(&out, 4);
    run<4, true, 0, 0>("barrier only", out);
    run<8, true, 0, 0>("barrier only", out);
    run<4, false, 16, 0>("32x32x16 x16, no barrier", out);
    run<4, true, 16, 0>("32x32x16 x16 + barrier", out);
    run<8, false, 16, 0>("32x32x16 x16, no barrier", out);
    run<8, true, 16, 0>("32x32x16 x16 + barrier", out);
    run<4, false, 32, 1>("16x16x32 x32, no barrier", out);
    run<4, true, 32, 1>("16x16x32 x32 + barrier", out);
    run<8, true, 32, 1>("16x16x32 x32 + barrier", out);
    run<4, true, 4, 0>("32x32x16 x4 + barrier", out);
    run<8, true, 4, 0>("32x32x16 x4 + barrier", out);
    return 0;
}
