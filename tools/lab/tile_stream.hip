// DMA-only model of the igemm main loop: each workgroup streams its own A panel [BM rows x K halves] (far memory)
// and a shared B panel [BN x K] (L2 resident) into LDS in K steps of 64 halves, `DEPTH` steps in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/tile_stream.hip -o tools/lab/tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// NW waves; BM = BN = 128; stage = 32 KB; pieces per wave per step = 32 / NW
template <int NW, int DEPTH, bool BARRIER, bool DO_A, bool DO_B>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const char* A, const char* B, int K, int mtiles, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPW = 16 / NW;                    // A pieces per wave per step (same for B)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_m = blockIdx.x % mtiles;
    const long rowb = (long)K * 2;
    const char* ap[PPW];
    const char* bp[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = 8 * (wave + NW * i) + lane / 8;
        ap[i] = A + ((long)tile_m * 128 + row) * rowb + (lane % 8) * 16;
        bp[i] = B + (long)row * rowb + (lane % 8) * 16;
    }
    const int nk = K / 64;
    constexpr int P = (DO_A ? PPW : 0) + (DO_B ? PPW : 0);
    auto issue = [&](int kt, int stage) {
        char* s = smem + stage * 32768;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (DO_A) glds16(ap[i] + kt * 128, s + (wave + NW * i) * 1024);
            if (DO_B) glds16(bp[i] + kt * 128, s + 16384 + (wave + NW * i) * 1024);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    for (int kt = 0; kt < nk; ++kt) {
        if (DEPTH == 1) wait_vmcnt<0>();
        else if (DEPTH == 2) wait_vmcnt<P>();
        else if (DEPTH == 3) wait_vmcnt<2 * P>();
        else wait_vmcnt<3 * P>();
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (kt + DEPTH < nk) issue(kt + DEPTH, (kt + DEPTH) % (DEPTH));
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (reinterpret_cast<unsigned*>(smem)[threadIdx.x] == 0x12345678u) out[0] = 1;
}

template <int NW, int DEPTH, bool BARRIER, bool DO_A, bool DO_B>
void run(const char* A, const char* B, int M, int K, int ntile_n, unsigned* out) {
    const int mtiles = M / 128;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        stream_kernel<NW, DEPTH, BARRIER, DO_A, DO_B><<<mtiles * ntile_n, 64 * NW, DEPTH * 32768>>>(A, B, K, mtiles, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double bytes = (double)mtiles * ntile_n * (K / 64) * ((DO_A ? 16384 : 0) + (DO_B ? 16384 : 0));
    printf("M %6d K %5d wgs %4d waves %d depth %d barrier %d A %d B %d : %7.2f us  %6.2f TB/s  %6.1f GB/s per CU  (%.3f us per K step)\n", M, K,
           mtiles * ntile_n, NW, DEPTH, (int)BARRIER, (int)DO_A, (int)DO_B, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256,
           best * 1e3 / (K / 64));
}

int main() {
    unsigned* out;
    (void)hipMalloc(&out, 4);
    const int M = 32768, K = 2304;
    char *A, *B;
    (void)hipMalloc(&A, (long)M * K * 2);
    (void)hipMalloc(&B, (long)256 * K * 2);
    (void)hipMemset(A, 1, (long)M * K * 2);
    (void)hipMemset(B, 1, (long)256 * K * 2);
    for (int m : {16384, 32768}) {
        run<4, 1, true, true, true>(A, B, m, K, 2, out);
        run<4, 2, true, true, true>(A, B, m, K, 2, out);
        run<4, 3, true, true, true>(A, B, m, K, 2, out);
        run<4, 4, true, true, true>(A, B, m, K, 2, out);
        run<4, 1, false, true, true>(A, B, m, K, 2, out);
        run<4, 3, false, true, true>(A, B, m, K, 2, out);
        run<8, 1, true, true, true>(A, B, m, K, 2, out);
        run<8, 2, true, true, true>(A, B, m, K, 2, out);
        run<8, 3, true, true, true>(A, B, m, K, 2, out);
        run<4, 1, true, true, false>(A, B, m, K, 2, out);
        run<4, 3, true, true, false>(A, B, m, K, 2, out);
        run<4, 1, true, false, true>(A, B, m, K, 2, out);
        run<4, 3, true, false, true>(A, B, m, K, 2, out);
    }
    return 0;
}
