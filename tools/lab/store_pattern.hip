// What does the WRITE pattern of the dynamic_layer GEMM cost by itself?   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
// csrc/wstat.hip writes the 91200 x 32768 fp16 parameter tensor (6 GB) at 3.7-3.8 TB/s where a plain fill reaches 6.9.  The kernels below
// issue ONLY the stores, in the kernel's decomposition: 256 persistent workgroups of 8 waves; XCD x (= blockIdx & 7) owns an eighth of the
// 32-row blocks, workgroup q of the XCD walks slabs q, q + 32 (512 channels = 1 KB per row) over the whole eighth; per 32-row step wave w
// stores rows x its 64 channels (128 bytes per row) as 16-byte pieces from the accumulator layout (lane -> row lane & 31, half lane >> 5).
//   mode 0: linear fill (every workgroup a contiguous share)            mode 1: the kernel's pattern
//   mode 2: the pattern with whole 1-KB row segments per wave instruction (row-coalesced: 64 lanes x 16 B = one row of the slab)
//   mode 3: slabs of 1024 channels (2 KB per row), 16 workgroups per slab pair ... (wider contiguous runs)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void store_kernel(char* out, long M, long N2 /* bytes per row */, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4v v = {1.f, 2.f, 3.f, (float)tid};
    const long MB = M / 32;
    if (mode == 0) {
        const long total = M * N2 / 16, per = total / 256;
        for (long i = (long)blockIdx.x * per + tid; i < (long)(blockIdx.x + 1) * per; i += 512) *reinterpret_cast<float4v*>(out + i * 16) = v;
        return;
    }
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const long xb0 = MB * xcd / 8, xb1 = MB * (xcd + 1) / 8;
    const long slab_bytes = mode == 3 ? 2048 : 1024;
    const int nslab = (int)(N2 / slab_bytes);
    for (int slab = q; slab < nslab; slab += 32) {
        for (long blk = xb0; blk < xb1; ++blk) {
            if (mode == 1) {          // wave w: 32 rows x 128 B, four 16-byte pieces per lane: piece i covers bytes [32 i + 16 (lane >> 5), +16) of the wave's 128
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4v*>(out + (blk * 32 + (lane & 31)) * N2 + slab * 1024 + wave * 128 + i * 32 + (lane >> 5) * 16) = v;
            } else if (mode == 2) {   // wave w: rows 4 w .. 4 w + 3 of the block, each a whole 1-KB slab row per instruction
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<float4v*>(out + (blk * 32 + wave * 4 + i) * N2 + slab * 1024 + lane * 16) = v;
            } else {                  // mode 3: 2-KB slab rows, wave w rows 4 w .. 4 w + 3, two instructions per row
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<float4v*>(out + (blk * 32 + wave * 4 + (i >> 1)) * N2 + slab * 2048 + (i & 1) * 1024 + lane * 16) = v;
            }
        }
    }
}

int main() {
    const long M = 91200, N2 = 32768 * 2;
    char* out;
    CK(hipMalloc(&out, M * N2));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; ++mode) {
        const int grid = mode == 3 ? 128 : 256;          // mode 3: 16 workgroups per XCD (32 slabs of 2 KB, two per workgroup)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel, dim3(mode == 3 ? 256 : grid), dim3(512), 0, 0, out, M, N2, mode);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d: %.3f ms per 6-GB pass, %.2f TB/s\n", mode, ms / 5, M * N2 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
