// Implicit-GEMM convolution / linear layer on MFMA (gfx950, fp16 in, fp32 accumulate).
//
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] (+ residual[m][n]) )
//
// A is the implicit im2col view of an NHWC fp16 tensor: m -> (image, oy, ox),
// k -> (ky, kx, c) with c fastest, so a K-slice of one tap is a contiguous run of channels.
// Wt is [Cout][Kpad] (the PyTorch Linear / OHWI conv layout), i.e. both MFMA operands are
// K-contiguous and are staged with 16-byte loads.  A Linear layer is the 1x1 case (H=W=1).
//
// Tiling: BMxBNx64 block tile, 4 waves (2x2), each wave (BM/2)x(BN/2) built from
// v_mfma_f32_32x32x16_f16; register-staged, double-buffered LDS with 144-byte row pitch
// (conflict-free for ds_read_b128 of 16 consecutive rows); epilogue goes through an fp32 LDS
// tile so bias/residual/ReLU are applied in fp32 and stores are 16 bytes per lane.
// Blocks are remapped so that each XCD (private L2) owns a contiguous range of tiles.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BK = 64;
constexpr int LDS_PITCH = BK + 8;  // halves; 144 B

template <int BM, int BN>
struct Smem {
    static constexpr int kStage = (BM + BN) * LDS_PITCH * 2;         // bytes per buffer
    static constexpr int kCPitch = BN + 4;                            // floats
    static constexpr int kCTile = BM * kCPitch * 4;
    static constexpr int kBytes = (2 * kStage > kCTile) ? 2 * kStage : kCTile;
};

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// SMALLC: Cin == 8 (stem, NHWC8 input): every 16-byte vector is one filter tap.
template <int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;      // 32x32 MFMA tiles per wave
    constexpr int A_IT = BM / 32, B_IT = BN / 32;   // 16-byte vectors per thread per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);
    half_t* Bs = As + BM * LDS_PITCH;
    constexpr int STAGE_HALVES = (BM + BN) * LDS_PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_n = lid % p.tiles_n, tile_m = lid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread load descriptors -------------------------------------------------
    const int vj = tid & 7;          // which 16-byte vector of the 64-wide K slice
    const int vr = tid >> 3;         // 0..31
    int a_base[A_IT], a_iy[A_IT], a_ix[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + vr + 32 * i;
        if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int img = t / p.Ho;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_base[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.Cin;
        } else {
            a_iy[i] = -(1 << 28);  // never valid
            a_ix[i] = 0;
            a_base[i] = 0;
        }
    }
    int b_off[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + vr + 32 * i;
        b_ok[i] = n < p.Cout;
        b_off[i] = (b_ok[i] ? n : 0) * p.Kpad + vj * 8;
    }

    half8 ra[A_IT], rb[B_IT];
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        int ky, kx, coff;
        bool tap_ok = true;
        if (SMALLC) {
            const int tap = kt * 8 + vj;
            tap_ok = tap < p.ntaps;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
            coff = 0;
        } else {
            const int k0 = kt * BK;
            const int tap = k0 / p.Cin;          // wave-uniform
            const int c0 = k0 - tap * p.Cin;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
            coff = c0 + vj * 8;
        }
        const int tap_off = (ky * p.W + kx) * p.Cin + coff;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // unconditional load from a clamped address + select: keeps the loads branch-free
            const half8 v = *reinterpret_cast<const half8*>(p.in + (ok ? a_base[i] + tap_off : 0));
            ra[i] = ok ? v : zero8;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const half8 v = *reinterpret_cast<const half8*>(p.w + b_off[i] + kt * BK);
            rb[i] = b_ok[i] ? v : zero8;
        }
    };
    auto store_tile = [&](int buf) {
        half_t* a = As + buf * STAGE_HALVES;
        half_t* b = Bs + buf * STAGE_HALVES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<half8*>(a + (vr + 32 * i) * LDS_PITCH + vj * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<half8*>(b + (vr + 32 * i) * LDS_PITCH + vj * 8) = rb[i];
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.Kpad / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const half_t* a = As + cur * STAGE_HALVES + (wm * (BM / 2) + frow) * LDS_PITCH + fk;
        const half_t* b = Bs + cur * STAGE_HALVES + (wn * (BN / 2) + frow) * LDS_PITCH + fk;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const half8*>(a + i * 32 * LDS_PITCH + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const half8*>(b + j * 32 * LDS_PITCH + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> fp32 LDS tile -> bias/residual/ReLU -> 16-byte stores ---
    constexpr int CP = Smem<BM, BN>::kCPitch;
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * (BN / 2) + j * 32 + (lane & 31);
                Cs[row * CP + col] = acc[i][j][r];
            }
    __syncthreads();

    constexpr int VPR = BN / 8;              // 8-wide vectors per tile row
    constexpr int RPP = 256 / VPR;           // rows per pass
    const int c8 = (tid % VPR) * 8;
    const int n = n0 + c8;
    const bool vec_ok = (p.Cout & 7) == 0;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (p.bias && n + e < p.Cout) ? p.bias[n + e] : 0.f;

    for (int r = tid / VPR; r < BM; r += RPP) {
        const int m = m0 + r;
        if (m >= p.M || n >= p.Cout) continue;
        float v[8];
        const float4v lo = *reinterpret_cast<const float4v*>(Cs + r * CP + c8);
        const float4v hi = *reinterpret_cast<const float4v*>(Cs + r * CP + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e] + bias8[e];
            v[e + 4] = hi[e] + bias8[e + 4];
        }
        if (p.res_mode) {
            long ridx;
            if (p.res_mode == 1) {
                ridx = (long)m * p.Cout + n;
            } else {  // nearest x2 upsample of a [N, Ho/2, Wo/2, Cout] tensor (FPN top-down)
                const int ox = m % p.Wo;
                const int t = m / p.Wo;
                const int oy = t % p.Ho;
                const int img = t / p.Ho;
                ridx = ((long)(img * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + n;
            }
            if (p.res_f32) {
                const float* rp = reinterpret_cast<const float*>(p.res) + ridx;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.Cout) v[e] += rp[e];
            } else if (vec_ok) {
                const half8 rv = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.res) + ridx);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            } else {
                const half_t* rp = reinterpret_cast<const half_t*>(p.res) + ridx;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.Cout) v[e] += (float)rp[e];
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const long oidx = (long)m * p.ldc + n;
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.out) + oidx;
            if (vec_ok && (p.ldc & 3) == 0) {
                *reinterpret_cast<float4v*>(op) = (float4v){v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4v*>(op + 4) = (float4v){v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.Cout) op[e] = v[e];
            }
        } else {
            half_t* op = reinterpret_cast<half_t*>(p.out) + oidx;
            if (vec_ok && (p.ldc & 7) == 0) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                *reinterpret_cast<half8*>(op) = hv;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.Cout) op[e] = (half_t)v[e];
            }
        }
    }
}

template <int BM, int BN, bool SMALLC>
int launch(const IgemmParams& p0, hipStream_t s) {
    IgemmParams p = p0;
    p.tiles_m = ceil_div(p.M, BM);
    p.tiles_n = ceil_div(p.Cout, BN);
    constexpr int smem = Smem<BM, BN>::kBytes;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, SMALLC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_kernel<BM, BN, SMALLC>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

int dvid_igemm_launch(const IgemmParams& p, hipStream_t s) {
    // v2 (direct-to-LDS staging, igemm2.hip) is the product kernel; DVID_IGEMM_V1=1 selects the
    // register-staged round-1 kernel below for A/B measurements.
    static const bool use_v1 = getenv("DVID_IGEMM_V1") != nullptr;
    if (!use_v1 || p.splitk > 1 || p.relu == 2) return dvid_igemm2_launch(p, s);
    if (p.M <= 0 || p.Cout <= 0) return DVID_OK;
    if (p.Kpad % BK != 0 || p.Kpad < BK) return DVID_ERR_ARG;
    const bool smallc = (p.Cin == 8 && p.KH * p.KW > 1);
    if (!smallc && (p.Cin % BK != 0)) return DVID_ERR_UNSUPPORTED;
    if (p.res_mode == 2 && ((p.Ho | p.Wo) & 1)) return DVID_ERR_ARG;
    if (smallc) return launch<128, 64, true>(p, s);
    // tile choice: big tiles only when they still fill the chip (256 CUs, 2 blocks/CU)
    const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.Cout, 128);
    if (t128 >= 384 && p.Cout >= 128) return launch<128, 128, false>(p, s);
    const long t12864 = (long)ceil_div(p.M, 128) * ceil_div(p.Cout, 64);
    if (t12864 >= 384) return launch<128, 64, false>(p, s);
    return launch<64, 64, false>(p, s);
}
