// LAB (not part of libdvid_hip): 256x256 GEMM tile with two wave groups in anti-phase.
//
// C[M,N] (fp16) = A[M,K] . B[N,K]^T, fp16 operands, fp32 accumulate on v_mfma_f32_32x32x16_f16 -- the same MFMA, K order
// and LDS tile format (BK 32: 64-byte rows, XOR key (row >> 2) & 3) as csrc/igemm2.hip, so a production version stays
// bit-identical to the other tile configurations.  8 waves = 2 groups x 4; group g owns rows [128 g, 128 g + 128) of the
// tile, wave (g, wn) the 128 x 64 block at columns 64 wn.  Every K tile (32 wide) is handled in two slots per group,
//   R(t): issue this wave's 4 DMA pieces of tile t + 3, read the 12 fragments of tile t, wait for them, wait for the own
//         pieces of tile t + 1, barrier
//   M(t): 16 MFMAs, barrier
// and group 1 runs one slot behind group 0, so that on every SIMD one wave is in its MFMA slot while the other is in its
// LDS / DMA slot.  Four 32-KiB stages: tile t + 3 re-uses the stage of tile t - 1, whose last fragment reads (group 1,
// previous slot) were retired by the lgkmcnt(0) in front of that slot's barrier.
//
// Measured on MI355X (round 1; profiles/r01_lab_gemm_pingpong.txt), TFLOP/s at 4096^3 | 58368x256x1024 | 58368x256x2304:
//   R slot also issues the DMA (first version)            933 | 680 | 758
//   all 4 DMA pieces between the MFMAs of the M slot     1130 | 791 | 885      <- adopted as igemm2 configuration 256x256x32/5
//   same, both wave rows in lock-step                     975 | 687 | 783
//   "pipe": no read slot, fragments read one tile ahead  1038 | 736 | 840
//   "bdirect": B fragments global -> VGPR, only A by DMA   789 | 589 | 569     (fragment-shaped loads are dear)
//   all DMA pieces in the R slot after its reads         1023 | 740 | 820
//   same loop with the DMA removed (garbage results)     1557 | 973 | 1272
//   round 2: pieces removed -- 3 of 4: 1080, weights only: 1369, A only: 1125 (profiles/r02_lab_dma_ablation.txt); buffer-addressed
//   DMA (raw_ptr_buffer_load_lds, K step in the scalar offset) instead of global_load_lds: 892 vs 895 -- no difference
// i.e. the global -> LDS DMA of the two operands costs ~0.25 us per 256x256x32 step wherever its instructions are placed
// (between MFMAs, in the read slot, after the reads), and that is what separates this structure from ~1.5 PFLOP/s.  (igemm2's 256x256x64/2: 806 | 719 | 811-873.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_pingpong.hip -o tools/lab/gemm_pingpong && tools/lab/gemm_pingpong
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void glds16(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// buffer-addressed variant of the DMA: resource descriptor + 32-bit per-lane offset + scalar offset (the K step) -- no 64-bit
// address arithmetic per piece, hardware range check
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

constexpr int BM = 256, BN = 256, BK = 32, NSTAGE = 4;
constexpr int STAGE = (BM + BN) * BK * 2;        // 32 KiB
constexpr int A_BYTES = BM * BK * 2;

template <bool STAGGER, bool PRIO, int RP, int ABL = 0>      // ABL (diagnostics): 1 no DMA in the loop, 2 no fragment reads, 4 no A pieces (B only), 8 no B pieces (A only), 16 one A piece of two; RP: DMA pieces (of 4 per tile) issued in the R slot; the rest go between the MFMAs
__global__ __launch_bounds__(512) void gemm_pingpong(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C,
                                                     int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;

    // DMA: wave w stages pieces {w, w + 8} of A and of B (16 rows x 64 B each); lane -> row 16 j + lane / 4, chunk lane % 4
    const char* a_src[2];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + 8 * i) + (lane >> 2);
        const int lch = (lane & 3) ^ ((row >> 2) & 3);
        a_src[i] = reinterpret_cast<const char*>(A + (long)(m0 + row) * K + lch * 8);
        b_src[i] = reinterpret_cast<const char*>(B + (long)(n0 + row) * K + lch * 8);
    }
    // which of the 4 pieces (0, 1: A; 2, 3: B) exist under the ablation
    constexpr bool BUF = (ABL & 32) != 0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(A), 0, (int)((long)M * K * 2 > 0x7fffffffL ? 0x7fffffff : (long)M * K * 2), 0x00027000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(B), 0, (int)((long)N * K * 2), 0x00027000);
    int a_vo[2], b_vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_vo[i] = (int)(a_src[i] - reinterpret_cast<const char*>(A));
        b_vo[i] = (int)(b_src[i] - reinterpret_cast<const char*>(B));
    }
    auto live = [](int pc) { return !((ABL & 4) && pc < 2) && !((ABL & 8) && pc >= 2) && !((ABL & 16) && pc == 1); };
    auto issue = [&](int t) {
        char* st = smem + (t % NSTAGE) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (BUF) {
                blds16(ra, a_vo[i], t * BK * 2, st + (wave + 8 * i) * 1024);
                blds16(rb, b_vo[i], t * BK * 2, st + A_BYTES + (wave + 8 * i) * 1024);
                continue;
            }
            if (live(i)) glds16(a_src[i] + (long)t * BK * 2, st + (wave + 8 * i) * 1024);
            if (live(2 + i)) glds16(b_src[i] + (long)t * BK * 2, st + A_BYTES + (wave + 8 * i) * 1024);
        }
    };
    constexpr int P = 4 - ((ABL & 4) ? 2 : 0) - ((ABL & 8) ? 2 : 0) - ((ABL & 16) ? 1 : 0);      // DMA instructions per wave per tile

    // fragment addressing
    const int frow = lane & 31;
    const int sw = (frow >> 2) & 3;
    const int fa_off = (grp * 128 + frow) * 64;
    const int fb_off = A_BYTES + (wn * 64 + frow) * 64;
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;

    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: tiles 0..2 in flight, tile 0 landed
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    if (nk > 2) wait_vmcnt<2 * P>(); else if (nk > 1) wait_vmcnt<P>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one slot behind group 0

    for (int t = 0; t < nk; ++t) {
        // ---- R(t) ---------------------------------------------------------------------------------------------
        const char* st = smem + (t % NSTAGE) * STAGE;
        half8 fa[4][2], fb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][ks] = *reinterpret_cast<const half8*>(st + ((ABL & 2) ? 0 : fa_off + i * 32 * 64 + choff[ks]));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j][ks] = *reinterpret_cast<const half8*>(st + fb_off + j * 32 * 64 + choff[ks]);
        }
        wait_lgkm0();
        if (RP > 0 && t + 3 < nk && !(ABL & 1)) {
            char* dst = smem + ((t + 3) % NSTAGE) * STAGE;
#pragma unroll
            for (int pc = 0; pc < RP; ++pc) {
                if (pc < 2) glds16(a_src[pc] + (long)(t + 3) * BK * 2, dst + (wave + 8 * pc) * 1024);
                else glds16(b_src[pc - 2] + (long)(t + 3) * BK * 2, dst + A_BYTES + (wave + 8 * (pc - 2)) * 1024);
            }
        }
        // own pieces of tile t + 1 landed (tiles t + 2, t + 3 may stay in flight; fewer exist at the tail)
        {
            // issued so far: tiles <= t + 2 completely, RP pieces of tile t + 3; tile t + 1 must have landed
            const bool t2 = t + 2 < nk, t3 = t + 3 < nk;
            if (ABL & 1) wait_vmcnt<0>(); else if (t3) wait_vmcnt<P + RP>(); else if (t2) wait_vmcnt<P>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M(t) ---------------------------------------------------------------------------------------------
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        {
            const bool dma = RP < 4 && t + 3 < nk && !(ABL & 1);
            char* dst = smem + ((t + 3) % NSTAGE) * STAGE;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ks = q >> 3, i = (q >> 1) & 3, j = q & 1;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][ks], fb[j][ks], acc[i][j], 0, 0, 0);
                if (dma && (q & 3) == 1 && (q >> 2) >= RP) {          // remaining DMA pieces after MFMAs 1, 5, 9, 13
                    const int pc = q >> 2;          // 0, 1: A pieces; 2, 3: B pieces
                    if (!live(pc)) continue;
                    if (BUF) {
                        if (pc < 2) blds16(ra, a_vo[pc], (t + 3) * BK * 2, dst + (wave + 8 * pc) * 1024);
                        else blds16(rb, b_vo[pc - 2], (t + 3) * BK * 2, dst + A_BYTES + (wave + 8 * (pc - 2)) * 1024);
                        continue;
                    }
                    if (pc < 2) glds16(a_src[pc] + (long)(t + 3) * BK * 2, dst + (wave + 8 * pc) * 1024);
                    else glds16(b_src[pc - 2] + (long)(t + 3) * BK * 2, dst + A_BYTES + (wave + 8 * (pc - 2)) * 1024);
                }
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();          // balance the extra barrier of group 1
    __syncthreads();

    // ---- epilogue: each wave stages 64 rows x 64 cols fp32 at a time in its own 16-KiB LDS slice ----------------------
    float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 64);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    cs[row * 64 + col] = acc[hh * 2 + i][j][r];
                }
        wait_lgkm0();
        // a wave reads back only what it wrote itself: 8 lanes cover one 64-col row (8 halves each), 8 rows per pass
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)cs[row * 64 + c8 + e];
            const long m = m0 + grp * 128 + hh * 64 + row;
            *reinterpret_cast<half8*>(C + m * N + n0 + wn * 64 + c8) = hv;
        }
        wait_lgkm0();
    }
}

// Variant "pipe": no separate read slot.  The fragments of tile t + 1 are read (into a second register set) at the head
// of tile t's MFMA block, the DMA pieces of tile t + 3 go between the MFMAs, one barrier per K tile.
__global__ __launch_bounds__(512) void gemm_pipe(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C,
                                                 int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;
    const char* a_src[2];
    const char* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + 8 * i) + (lane >> 2);
        const int lch = (lane & 3) ^ ((row >> 2) & 3);
        a_src[i] = reinterpret_cast<const char*>(A + (long)(m0 + row) * K + lch * 8);
        b_src[i] = reinterpret_cast<const char*>(B + (long)(n0 + row) * K + lch * 8);
    }
    auto piece = [&](int t, int pc) {
        char* dst = smem + (t % NSTAGE) * STAGE;
        if (pc < 2) glds16(a_src[pc] + (long)t * BK * 2, dst + (wave + 8 * pc) * 1024);
        else glds16(b_src[pc - 2] + (long)t * BK * 2, dst + A_BYTES + (wave + 8 * (pc - 2)) * 1024);
    };
    constexpr int P = 4;
    const int frow = lane & 31;
    const int sw = (frow >> 2) & 3;
    const int fa_off = (grp * 128 + frow) * 64;
    const int fb_off = A_BYTES + (wn * 64 + frow) * 64;
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;
    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    half8 fa[2][4][2], fb[2][2][2];
    auto read_frags = [&](int t, int buf) {
        const char* st = smem + (t % NSTAGE) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[buf][i][ks] = *reinterpret_cast<const half8*>(st + fa_off + i * 32 * 64 + choff[ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[buf][j][ks] = *reinterpret_cast<const half8*>(st + fb_off + j * 32 * 64 + choff[ks]);
        }
    };
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d < nk)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) piece(d, pc);
    if (nk > 2) wait_vmcnt<2 * P>(); else if (nk > 1) wait_vmcnt<P>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0);
    if (nk > 2) wait_vmcnt<P>(); else wait_vmcnt<0>();          // tile 1
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    auto step = [&](int t, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (t + 1 < nk) read_frags(t + 1, buf ^ 1);
        const bool dma = t + 3 < nk;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ks = q >> 3, i = (q >> 1) & 3, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[buf][i][ks], fb[buf][j][ks], acc[i][j], 0, 0, 0);
            if (dma && (q & 3) == 1) piece(t + 3, q >> 2);
        }
        __builtin_amdgcn_s_setprio(0);
        wait_lgkm0();
        // own pieces of tile t + 2 landed; tile t + 3 (if issued) may stay in flight
        if (t + 3 < nk) wait_vmcnt<P>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
    }
    if (t < nk) step(t, std::integral_constant<int, 0>{});
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 64);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    cs[row * 64 + col] = acc[hh * 2 + i][j][r];
                }
        wait_lgkm0();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)cs[row * 64 + c8 + e];
            const long m = m0 + grp * 128 + hh * 64 + row;
            *reinterpret_cast<half8*>(C + m * N + n0 + wn * 64 + c8) = hv;
        }
        wait_lgkm0();
    }
}

// Variant "bdirect": the B operand never touches LDS.  Every wave loads its own B fragments (64 columns x 32 k) straight
// from global / L2 into registers in MFMA layout, two tiles ahead; only A goes through the DMA ring.  Halves the LDS writes
// and removes a third of the fragment reads.  Anti-phase schedule as above; loads beyond the last tile are clamped
// re-loads so that every slot issues the same number of VMEM operations (uniform vmcnt bookkeeping).
__global__ __launch_bounds__(512) void gemm_bdirect(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C,
                                                    int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ASTAGE = BM * BK * 2;          // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;
    const char* a_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * (wave + 8 * i) + (lane >> 2);
        const int lch = (lane & 3) ^ ((row >> 2) & 3);
        a_src[i] = reinterpret_cast<const char*>(A + (long)(m0 + row) * K + lch * 8);
    }
    // B fragment (j, ks): row n0 + 64 wn + 32 j + (lane & 31), halves [32 t + 16 ks + 8 (lane >> 5), +8)
    const char* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_src[j] = reinterpret_cast<const char*>(B + (long)(n0 + wn * 64 + j * 32 + (lane & 31)) * K + (lane >> 5) * 8);
    auto a_piece = [&](int t, int pc) {
        const int tc = min(t, nk - 1);
        glds16(a_src[pc] + (long)tc * BK * 2, smem + (t & 3) * ASTAGE + (wave + 8 * pc) * 1024);
    };
    const int frow = lane & 31;
    const int sw = (frow >> 2) & 3;
    const int fa_off = (grp * 128 + frow) * 64;
    int choff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) choff[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;
    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    half8 fbr[3][2][2];
    auto b_load = [&](int t, auto setc) {
        constexpr int set = decltype(setc)::value;
        const int tc = min(t, nk - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fbr[set][j][ks] = *reinterpret_cast<const half8*>(b_src[j] + ((long)tc * BK + ks * 16) * 2);
    };
    // prologue: A tiles 0..2 (6 ops), B sets for tiles 0, 1 (8 ops)
    a_piece(0, 0); a_piece(0, 1); a_piece(1, 0); a_piece(1, 1);
    b_load(0, std::integral_constant<int, 0>{});
    a_piece(2, 0); a_piece(2, 1);
    b_load(1, std::integral_constant<int, 1>{});
    // A tile 0 landed everywhere: everything but the last 6 ops (A(2) x2, B(1) x4) of this wave
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    auto step = [&](int t, auto setc) {
        constexpr int set = decltype(setc)::value;
        constexpr int set2 = (set + 2) % 3;
        // ---- R(t): A fragments from LDS
        const char* st = smem + (t & 3) * ASTAGE;
        half8 fa[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][ks] = *reinterpret_cast<const half8*>(st + fa_off + i * 32 * 64 + choff[ks]);
        wait_lgkm0();
        // VMEM order so far: ... [M(t-2): B(t) x4, A(t+1) x2] [M(t-1): B(t+1) x4, A(t+2) x2]; B(t) and A(t+1) must be complete
        wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M(t): 16 MFMAs; B(t+2) loads and A(t+3) DMA pieces between them
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ks = q >> 3, i = (q >> 1) & 3, j = q & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][ks], fbr[set][j][ks], acc[i][j], 0, 0, 0);
            if (q == 1) b_load(t + 2, std::integral_constant<int, set2>{});
            if (q == 9) a_piece(t + 3, 0);
            if (q == 13) a_piece(t + 3, 1);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; t + 2 < nk; t += 3) {
        step(t, std::integral_constant<int, 0>{});
        step(t + 1, std::integral_constant<int, 1>{});
        step(t + 2, std::integral_constant<int, 2>{});
    }
    if (t < nk) step(t, std::integral_constant<int, 0>{});
    if (t + 1 < nk) step(t + 1, std::integral_constant<int, 1>{});
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
    __syncthreads();
    float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 64);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    cs[row * 64 + col] = acc[hh * 2 + i][j][r];
                }
        wait_lgkm0();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)cs[row * 64 + c8 + e];
            const long m = m0 + grp * 128 + hh * 64 + row;
            *reinterpret_cast<half8*>(C + m * N + n0 + wn * 64 + c8) = hv;
        }
        wait_lgkm0();
    }
}

__global__ void gemm_naive(const half_t* A, const half_t* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)B[(long)n * K + k];
    C[(long)m * N + n] = s;
}

int main() {
    const int smem = NSTAGE * STAGE;
    typedef void (*kern_t)(const half_t*, const half_t*, half_t*, int, int, int);
    constexpr int NV = 7;
    const kern_t kerns[NV] = {gemm_pingpong<true, true, 0>, gemm_pingpong<true, true, 0, 32>, gemm_pingpong<true, true, 0, 16>, gemm_pingpong<true, true, 0, 4>, gemm_pingpong<true, true, 0, 8>,
                              gemm_pingpong<true, true, 0, 1>, gemm_pingpong<true, true, 0, 3>};
    const char* names[NV] = {"R0/M4", "R0/M4 buffer-addr", "3 of 4 pieces", "B pieces only", "A pieces only", "no DMA", "no DMA no reads"};
    for (int v = 0; v < NV; ++v) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    for (int v = 0; v < NV; ++v) {
    const kern_t gemm_kernel = kerns[v];
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{512, 512, 96}, {4096, 4096, 4096}, {58368, 256, 1024}, {58368, 256, 2304}, {126464, 256, 2304}};
    for (const Shape& sh : shapes) {
        const long na = (long)sh.M * sh.K, nb = (long)sh.N * sh.K, nc = (long)sh.M * sh.N;
        std::vector<half_t> ha(na), hb(nb);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.f - 0.5f; };
        for (auto& v : ha) v = (half_t)rnd();
        for (auto& v : hb) v = (half_t)(rnd() * 0.25f);
        half_t *dA, *dB, *dC;
        CHECK(hipMalloc(&dA, na * 2)); CHECK(hipMalloc(&dB, nb * 2)); CHECK(hipMalloc(&dC, nc * 2));
        CHECK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
        const int grid = (sh.M / BM) * (sh.N / BN);
        gemm_kernel<<<grid, 512, smem>>>(dA, dB, dC, sh.M, sh.N, sh.K);
        CHECK(hipDeviceSynchronize());
        double worst = 0;
        if ((double)sh.M * sh.N * sh.K < 3e10) {
            float* dR;
            CHECK(hipMalloc(&dR, nc * 4));
            gemm_naive<<<dim3((sh.N + 255) / 256, sh.M), 256>>>(dA, dB, dR, sh.M, sh.N, sh.K);
            std::vector<float> hr(nc);
            std::vector<half_t> hc(nc);
            CHECK(hipMemcpy(hr.data(), dR, nc * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost));
            for (long i = 0; i < nc; ++i) worst = fmax(worst, fabs((double)hc[i] - hr[i]) / (1.0 + fabs(hr[i])));
            CHECK(hipFree(dR));
        } else {
            worst = -1;
        }
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(a));
            for (int it = 0; it < 5; ++it) gemm_kernel<<<grid, 512, smem>>>(dA, dB, dC, sh.M, sh.N, sh.K);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            best = fminf(best, ms / 5);
        }
        printf("%-14s %6d x %5d x %5d : %8.2f us  %7.1f TFLOP/s   max rel err %s%.3e\n", names[v], sh.M, sh.N, sh.K, best * 1e3,
               2.0 * sh.M * sh.N * sh.K / best / 1e9, worst < 0 ? "(unchecked) " : "", worst < 0 ? 0.0 : worst);
        CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC));
    }
    }
    return 0;
}
