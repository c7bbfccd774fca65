// Per-CU fill-rate probe: how fast can one CU pull an L2-resident buffer (a) into VGPRs with global_load_dwordx4,
// (b) into LDS with global_load_lds_dwordx4.  Every workgroup streams the same `bytes` region `reps` times.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/l2_rate.hip -o /tmp/l2_rate && /tmp/l2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void vgpr_kernel(const u32x4* __restrict__ src, long n16, int reps, unsigned* out, int stagger) {
    const int tid = threadIdx.x, nt = blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    const long off0 = stagger ? ((long)blockIdx.x * 4099) % n16 : 0;
    for (int r = 0; r < reps; ++r) {
        for (long i = tid; i < n16; i += (long)nt * UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                long j = i + (long)u * nt + off0;
                if (j >= n16) j -= n16;
                v[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
    }
    if (acc.x == 0x12345678u) out[0] = acc.y;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void lds_kernel(const u32x4* __restrict__ src, long n16, int reps, unsigned* out, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
    const long off0 = stagger ? ((long)blockIdx.x * 4099) % n16 : 0;
    for (int r = 0; r < reps; ++r) {
        for (long i = tid; i < n16; i += (long)nt * UNROLL) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                long j = i + (long)u * nt + off0;
                if (j >= n16) j -= n16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j),
                                                 (__attribute__((address_space(3))) void*)(smem + (u * (nt / 64) + wave) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) out[0] = 1;
}

int main() {
    const int reps = 40;
    unsigned* out;
    hipMalloc(&out, 4);
    for (long bytes : {512l << 10, 2l << 20, 8l << 20}) {
        u32x4* src;
        hipMalloc(&src, bytes);
        hipMemset(src, 1, bytes);
        for (int nwg : {256, 512, 1024}) {
            for (int mode = 0; mode < 4; ++mode) {
                hipEvent_t a, b;
                hipEventCreate(&a);
                hipEventCreate(&b);
                float best = 1e9;
                for (int it = 0; it < 3; ++it) {
                    hipEventRecord(a);
                    const int stagger = mode & 1;
                    if (mode < 2) vgpr_kernel<8><<<nwg, 256>>>(src, bytes / 16, reps, out, stagger);
                    else lds_kernel<8><<<nwg, 256, 8 * 4 * 1024>>>(src, bytes / 16, reps, out, stagger);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                const double total = (double)bytes * reps * nwg;
                printf("%-4s stagger=%d buf %5ld KB, %4d wgs x 256 thr: %8.3f ms  %7.2f TB/s aggregate  %6.1f GB/s per CU\n", mode < 2 ? "VGPR" : "LDS",
                       mode & 1, bytes >> 10, nwg, best, total / best / 1e9, total / best / 1e6 / 256);
            }
        }
        hipFree(src);
    }
    return 0;
}
