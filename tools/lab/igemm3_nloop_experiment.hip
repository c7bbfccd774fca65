// EXPERIMENT (not built into libdvid_hip): A-stationary N loop for short-K 1x1 convolutions / linear layers.
// Result on MI355X: correct but SLOWER than igemm2 tiles (58368x1024x256 + residual: 118 us vs 83 us; 7200x32768x256:
// 368 us vs 262 us) -- with one workgroup per CU every latency of the N-tile boundary (store drain, bias / residual
// loads) is exposed; it needs a producer/consumer wave split to pay off.  Kept for the next round.
//
// The short-K, wide-N layers of the path (bottleneck conv3 + residual: K = 64..512 -> N = 256..2048; the decoder's
// dynamic_layer: K = 256 -> N = 32768) move ~4 KB of HBM per output row and do little arithmetic; tiled as independent
// 128x128 workgroups (igemm2) every tile re-stages the same A rows, pays its own address prologue and first-load
// latency, and lives ~15 us for 8 short K steps (profiles/r01_memory_probes.txt).  Here one workgroup owns 128 output
// rows: it stages their full-K A panel in LDS ONCE, then walks the N tiles, streaming only weight tiles (L2 resident)
// through a 3-slot DMA ring that runs across N-tile boundaries, while the epilogue of each N tile (bias, residual,
// activation, 16-byte stores; residual prefetched into registers at the start of the tile) uses its own LDS buffer.
// Same MFMA (v_mfma_f32_32x32x16_f16), same K order and the same epilogue function as igemm2: results are
// bit-identical, so the per-shape tuner may pick either kernel.
//
// Ordering of the DMA ring against the other memory traffic of a wave: `vmcnt` retires loads in order, but stores may
// retire out of order with loads, so a counted wait is only used where no store is outstanding (inside an N tile);
// the N-tile boundary drains with vmcnt(0).
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page3[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt3() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16_3(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM3 = 128, BN3 = 128, BK3 = 64, NW3 = 4, NRING3 = 3;
constexpr int TILE_BYTES3 = 128 * BK3 * 2;                 // one 128-row x 64-half operand tile: 16 KiB
constexpr int CP3 = BN3 + 4;
constexpr int CS_BYTES3 = (BM3 / 2) * CP3 * 4;

template <int KT>
constexpr int smem3() { return KT * TILE_BYTES3 + NRING3 * TILE_BYTES3 + CS_BYTES3; }

// KT = Kpad / 64 (1, 2 or 4).  Grid: tiles_m x nparts; workgroup (tile_m, part) computes rows [128 tile_m, +128) of the
// N tiles [part * nj_per, ...).
template <int KT>
__global__ __launch_bounds__(256) void igemm3_kernel(IgemmParams p, int nparts, int nj_per) {
    constexpr int ROW_BYTES = BK3 * 2, CHUNKS = 8, RPP = 8;      // 128-byte rows, 8 rows per 1-KiB DMA piece
    constexpr int IT = 128 / RPP / NW3;                          // DMA pieces per wave per operand tile (4)
    constexpr int KS = BK3 / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_panel = smem;
    char* b_ring = smem + KT * TILE_BYTES3;
    float* Cs = reinterpret_cast<float*>(smem + (KT + NRING3) * TILE_BYTES3);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int part = (int)blockIdx.x / p.tiles_m;
    const int tile_m = (int)blockIdx.x - part * p.tiles_m;
    const int m0 = tile_m * BM3;
    const int j0 = part * nj_per;
    const int NJ = min(nj_per, p.tiles_n - j0);
    if (NJ <= 0) return;
    const char* zero = reinterpret_cast<const char*>(g_zero_page3);

    // ---- DMA descriptors (lane -> row of the piece, physical 16-byte chunk; logical chunk = physical ^ key(row)) ----
    const int lrow = lane / CHUNKS, pch = lane % CHUNKS;
    const char* a_src[IT];
    const char* b_src[IT];          // row pointer of N tile j0, K tile 0; other tiles by wave-uniform offsets
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int row = RPP * (wave + NW3 * i) + lrow;
        const int lch = pch ^ ((row >> 1) & (CHUNKS - 1));
        const int m = m0 + row;
        a_src[i] = zero;
        if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int img = t / p.Ho;
            a_src[i] = reinterpret_cast<const char*>(p.in + ((long)(img * p.H + oy * p.stride) * p.W + ox * p.stride) * p.Cin + lch * 8);
        }
        b_src[i] = reinterpret_cast<const char*>(p.w + (long)(j0 * BN3 + row) * p.Kpad + lch * 8);
    }

    // B tile `t` of this workgroup: N tile j0 + t / KT, K tile t % KT -> ring slot t % NRING3
    auto issue_b = [&](int t) {
        const int j = t / KT, kt = t - j * KT;
        char* dst = b_ring + (t % NRING3) * TILE_BYTES3;
        const long off = ((long)j * BN3 * p.Kpad + kt * BK3) * 2;
        const int nbase = (j0 + j) * BN3;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int row = RPP * (wave + NW3 * i) + lrow;
            const bool ok = j < NJ && nbase + row < p.Cout;      // past the last tile / past Cout: zeros (keeps the counts uniform)
            glds16_3(ok ? b_src[i] + off : zero, dst + (wave + NW3 * i) * 1024);
        }
    };

    // ---- prologue: the whole A panel, then the first two B tiles ---------------------------------------------------
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < IT; ++i)
            glds16_3(a_src[i] == zero ? zero : a_src[i] + kt * BK3 * 2, a_panel + kt * TILE_BYTES3 + (wave + NW3 * i) * 1024);
    issue_b(0);
    issue_b(1);

    // fragment addressing (as igemm2, BKT 64): row = base32 + (lane & 31); logical chunk = 2*ks + (lane >> 5)
    const int frow = lane & 31;
    const int sw = (frow >> 1) & (CHUNKS - 1);
    const int fa_off = (wm * 64 + frow) * ROW_BYTES;
    const int fb_off = (wn * 64 + frow) * ROW_BYTES;
    int choff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) choff[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;

    float16v acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto compute = [&](int kt, int slot) {
        const char* sa = a_panel + kt * TILE_BYTES3;
        const char* sb = b_ring + slot * TILE_BYTES3;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const half8*>(sa + fa_off + i * 32 * ROW_BYTES + choff[ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const half8*>(sb + fb_off + j * 32 * ROW_BYTES + choff[ks]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    // epilogue geometry (as igemm2 for a 128x128 tile and 256 threads)
    constexpr int VPR = BN3 / 8, ERPP = 256 / VPR, EROWS = (BM3 / 2) / ERPP;
    const int c8 = (tid % VPR) * 8;
    const bool pre_ok = p.res_mode == 1 && !p.res_f32 && (p.Cout & 7) == 0;

    for (int j = 0; j < NJ; ++j) {
        const int t0 = j * KT;
        const int n0 = (j0 + j) * BN3;
        const int n = n0 + c8;
        // N-tile boundary: everything issued so far has retired (the first two B tiles of this N tile among it; stores of
        // the previous epilogue may retire out of order with loads, hence no counted wait here)
        wait_vmcnt3<0>();
        __syncthreads();
        zero_acc();
        // residual of this N tile into registers, ahead of the tile's arithmetic
        half8 rpre[2][EROWS];
        if (pre_ok) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int e = 0; e < EROWS; ++e) {
                    const int m = m0 + half * (BM3 / 2) + tid / VPR + e * ERPP;
                    const bool ok = m < p.M && n < p.Cout;
                    rpre[half][e] = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.res) + (ok ? (long)m * p.Cout + n : 0));
                }
        }
        issue_b(t0 + 2);
        compute(0, t0 % NRING3);
#pragma unroll
        for (int kt = 1; kt < KT; ++kt) {
            // B(t0 + kt): for kt == 1 it retired at the boundary; later ones were issued inside this N tile, one tile
            // (IT pieces) behind them is still allowed in flight
            if (kt >= 2) wait_vmcnt3<IT>();
            __builtin_amdgcn_s_barrier();          // slot of B(t0 + kt - 1) is free, B(t0 + kt) visible to all waves
            issue_b(t0 + kt + 2);
            compute(kt, (t0 + kt) % NRING3);
        }

        // ---- epilogue of N tile j: two half tiles through the fp32 buffer ----------------------------------------
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = (p.bias && n + e < p.Cout) ? p.bias[n + e] : 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            const int col = wn * 64 + jj * 32 + (lane & 31);
                            Cs[row * CP3 + col] = acc[i][jj][r];
                        }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EROWS; ++e) {
                const int r = tid / VPR + e * ERPP;
                const int m = m0 + half * (BM3 / 2) + r;
                if (m < p.M && n < p.Cout) igemm_store_row8(p, Cs + r * CP3 + c8, m, n, bias8, pre_ok, rpre[half][e]);
            }
        }
    }
    wait_vmcnt3<0>();       // trailing dummy DMA pieces must not outlive the workgroup's LDS allocation
}

template <int KT>
int launch3(const IgemmParams& p0, hipStream_t s, int min_wgs) {
    IgemmParams p = p0;
    p.tiles_m = ceil_div(p.M, BM3);
    p.tiles_n = ceil_div(p.Cout, BN3);
    int nparts = 1;
    while (p.tiles_m * nparts < min_wgs && nparts * 2 <= p.tiles_n) nparts *= 2;
    const int nj_per = ceil_div(p.tiles_n, nparts);
    nparts = ceil_div(p.tiles_n, nj_per);
    constexpr int smem = smem3<KT>();
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm3_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm3_kernel<KT>), dim3(p.tiles_m * nparts), dim3(256), smem, s, p, nparts, nj_per);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

bool dvid_igemm3_supported(const IgemmParams& p) {
    return p.ntaps == 1 && p.pad == 0 && p.splitk <= 1 && p.Cin == p.Kpad && (p.Kpad == 64 || p.Kpad == 128 || p.Kpad == 256) &&
           p.Cout >= 256 && p.res_mode != 2;
}

int dvid_igemm3_launch(const IgemmParams& p, hipStream_t s, int min_wgs) {
    if (!dvid_igemm3_supported(p)) return DVID_ERR_UNSUPPORTED;
    switch (p.Kpad / 64) {
        case 1: return launch3<1>(p, s, min_wgs);
        case 2: return launch3<2>(p, s, min_wgs);
        default: return launch3<4>(p, s, min_wgs);
    }
}
