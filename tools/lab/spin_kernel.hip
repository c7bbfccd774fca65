// A long-lived single workgroup with a chosen amount of static-size dynamic LDS, to sit beside other kernels' workgroups on one CU
// (tools/diag_chain_contention.py SIDE=spin): which property of a co-resident workgroup disturbs the fused block kernels when they
// do not own the whole LDS?   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/spin_kernel.hip -o tools/lab/libspin.so
#include <hip/hip_runtime.h>

__global__ void spin_kernel(long cycles, int* sink, int mode) {
    extern __shared__ int lds[];
    const long t0 = clock64();
    int acc = 0;
    if (mode == 0) {
        while (clock64() - t0 < cycles) acc += 1;
    } else {          // the same duration with some of the farthest-point sweep's habits per turn: bit 0 a barrier, bit 1 LDS writes / reads, bit 2 a cross-lane shuffle
        while (clock64() - t0 < cycles) {
            int v = acc + (int)threadIdx.x;
            if (mode & 4) v = __shfl_xor(v, 32, 64);
            if (mode & 2) lds[threadIdx.x] = v;
            if (mode & 1) __syncthreads();
            if (mode & 2) acc += lds[threadIdx.x ^ 1];
            else acc += v;
        }
    }
    if (acc == -1) sink[0] = lds[threadIdx.x & 1];
}

extern "C" int spin_launch(void* stream, int lds_bytes, long cycles, int threads, int* sink, int mode) {
    static int done = 0;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        done = 1;
    }
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(threads), lds_bytes, reinterpret_cast<hipStream_t>(stream), cycles, sink, mode);
    return (int)hipGetLastError();
}
