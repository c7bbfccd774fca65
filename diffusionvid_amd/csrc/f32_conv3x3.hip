// DTYPE float32, split operands: 3x3 / stride-1 / pad-1 convolution (every bottleneck conv2 but the three strided ones, the FPN output
// convolutions) with the A operand staged -- and split -- ONCE per input-channel chunk instead of once per filter tap.
//
// Why: on 128 x 128 x 32 tiles (csrc/f32.hip: f32x3_igemm_kernel) a K step pulls 16 KB of fp32 im2col rows and 16 KB of (hi, lo) weights
// through the L2 -> CU path and converts the 16 KB, and that path, not the matrix pipe (28 % busy), sets the pace: res4 conv2 runs at
// 253 TFLOP/s of fp32-grade work against the 833 roof (profiles/r06_layers_r101_x1_float32.csv).  A 3x3 convolution reads every input
// pixel nine times, once per tap.  Here -- the geometry of csrc/conv3x3.hip -- a workgroup of 8 waves owns an 8 x 32 patch of output
// pixels (256 GEMM rows) x BN output channels, stages the 10 x 34 halo of the patch for 32 input channels as two fp16 planes (global
// fp32 -> registers -> split -> LDS, pixels at a pitch of 40 halves: the ds_read_b128 fragments of 32 consecutive pixels are
// conflict-free) and takes the A fragments of all nine taps from it by shifting the LDS read address.  Per 256 x 128 x 32 step the
// fill path carries 16 KB of weights + 43.5 / 9 KB of pixels = 20.8 KB instead of 48, and one split per pixel value instead of nine.
// Weights: the pre-split (hi, lo) planes, the three [BN x 32] tiles of one filter row (ky; kx = 0, 1, 2) per stage, 64 bytes per row
// with an XOR swizzle of its four 16-byte pieces (key (row >> 2) & 3: conflict-free ds_read_b128 without padding), double-buffered:
// one barrier per 72 (BN 128) MFMAs of a wave.
//
// K order: channel chunk outermost, then tap, then channel inside the chunk (the tiled kernel: tap, then channel) -- the same products
// (the same split, the same three passes per 16-deep step: a_lo w_hi, a_hi w_lo, a_hi w_hi), another fp32 summation order; which of the
// two kernels a layer runs on is a function of its shape and the option table only (never of a timing), so results stay reproducible.
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "options.h"

namespace {

constexpr int C3_TH = 8, C3_TW = 32;                          // output patch
constexpr int C3_HH = C3_TH + 2, C3_HW = C3_TW + 2;           // halo
constexpr int C3_PX = C3_HH * C3_HW;                          // 340 halo pixels
constexpr int C3_PITCH = 40;                                  // halves per pixel / weight row in LDS (32 used)
constexpr int C3_HALO_PLANE = C3_PX * C3_PITCH;               // halves
constexpr int C3_HALO_ITERS = (C3_PX * 8 + 511) / 512;        // float4 loads per thread and chunk

template <int BN, int TAPS>
struct C3Smem {
    static constexpr int kHalo = 2 * C3_HALO_PLANE * 2;       // bytes: hi | lo
    static constexpr int kBTap = BN * 32;                     // halves of one tap's [BN x 32] tile (one plane)
    static constexpr int kBPlane = TAPS * kBTap;              // the taps of one step
    static constexpr int kBStage = 2 * kBPlane * 2;           // bytes: hi | lo
    static constexpr int kBytes = kHalo + 2 * kBStage;
};

// TAPS: filter taps per step (= per barrier): 3 (one filter row; 150 KB of LDS at BN 128, one workgroup per CU) for the long-K layers,
// 1 (70 KB at BN 64, two workgroups per CU: a tile's prologue and epilogue overlap the other's products) for the Cout = 64 layers of res2.
template <int BN, int TAPS>
__global__ __launch_bounds__(512) void f32x3_conv3x3_kernel(F32GemmParams p, int tiles_x, int tiles_y, int tiles_n) {
    using SM = C3Smem<BN, TAPS>;
    constexpr int NB = BN / 64;                  // 32-column blocks per wave (waves: 4 along M x 2 along N)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* const Hhi = reinterpret_cast<half_t*>(smem);
    half_t* const Hlo = Hhi + C3_HALO_PLANE;
    half_t* const Bbase = reinterpret_cast<half_t*>(smem + SM::kHalo);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    const int ntiles = (p.M / (p.Ho * p.Wo)) * tiles_y * tiles_x * tiles_n;
    int t = igemm_xcd_remap((int)blockIdx.x, ntiles);          // an XCD owns a contiguous run of patches: neighbours share halo lines and weights in its L2
    const int tn = t % tiles_n;
    t /= tiles_n;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int img = t / tiles_y;
    const int y0 = ty * C3_TH, x0 = tx * C3_TW, n0 = tn * BN;
    const float* const in_img = p.in + (long)img * p.H * p.W * p.Cin;

    // ---- halo staging: float4 idx = tid + 512 i -> (halo pixel, 4-channel group); out-of-image pixels are the convolution's zero padding
    const float* hsrc[C3_HALO_ITERS];
    int hdst[C3_HALO_ITERS];
    bool hok[C3_HALO_ITERS], hin[C3_HALO_ITERS];
#pragma unroll
    for (int i = 0; i < C3_HALO_ITERS; ++i) {
        const int idx = tid + 512 * i;
        const int hp = idx >> 3, c4 = (idx & 7) * 4;
        hin[i] = hp < C3_PX;
        const int hy = hp / C3_HW, hx = hp - hy * C3_HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        hok[i] = hin[i] && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        hsrc[i] = hok[i] ? in_img + ((long)y * p.W + x) * p.Cin + c4 : p.in;
        hdst[i] = hp * C3_PITCH + c4;
    }
    float4v hreg[C3_HALO_ITERS];
    float range_max = 0.f;
    auto halo_fetch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < C3_HALO_ITERS; ++i) hreg[i] = *reinterpret_cast<const float4v*>(hsrc[i] + (hok[i] ? chunk * 32 : 0));
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < C3_HALO_ITERS; ++i) {
            float4v v = hreg[i];
            if (!hok[i]) v = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) range_max = fmaxf(range_max, __builtin_fabsf(v[e]));
            const half4 h = __builtin_convertvector(v, half4);                                        // round to nearest even
            const half4 l = __builtin_convertvector(v - __builtin_convertvector(h, float4v), half4);   // v - hi is exact in fp32
            if (hin[i]) {
                *reinterpret_cast<half4*>(Hhi + hdst[i]) = h;
                *reinterpret_cast<half4*>(Hlo + hdst[i]) = l;
            }
        }
    };

    // ---- weight staging: 16-byte piece = tid + 512 i -> (tap kx, row, 8-half group) of a filter row's three [BN x 32] tiles
    constexpr int NPIECE = TAPS * BN * 4;
    constexpr int BP = (NPIECE + 511) / 512;
    long wsrc[BP];
    int wdst[BP], wkx[BP];
    bool wok[BP], win[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int piece = tid + 512 * i;
        win[i] = piece < NPIECE;
        const int kx = piece / (BN * 4), rem = piece - kx * (BN * 4);
        const int row = rem >> 2, g = rem & 3;
        wok[i] = win[i] && n0 + row < p.Cout;
        wkx[i] = kx;
        wsrc[i] = (long)(wok[i] ? n0 + row : 0) * p.Kpad + g * 8;
        wdst[i] = (win[i] ? kx : 0) * SM::kBTap + row * 32 + ((g ^ ((row >> 2) & 3)) * 8);
    }
    half8 wrh[BP], wrl[BP];
    auto w_fetch = [&](int chunk, int tg) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const long k = wok[i] ? (long)(tg * TAPS + wkx[i]) * p.Cin + chunk * 32 : 0;
            wrh[i] = *reinterpret_cast<const half8*>(p.w_hi + wsrc[i] + k);
            wrl[i] = *reinterpret_cast<const half8*>(p.w_lo + wsrc[i] + k);
        }
    };
    auto w_store = [&](int stage) {
        half_t* const bh = Bbase + stage * (2 * SM::kBPlane);
        half_t* const bl = bh + SM::kBPlane;
        const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < BP; ++i)
            if (win[i]) {
                *reinterpret_cast<half8*>(bh + wdst[i]) = wok[i] ? wrh[i] : hz;
                *reinterpret_cast<half8*>(bl + wdst[i]) = wok[i] ? wrl[i] : hz;
            }
    };

    float16v acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr int TG = 9 / TAPS;                                   // steps per channel chunk
    const int nchunks = p.Cin / 32, nsteps = nchunks * TG;
    halo_fetch(0);
    w_fetch(0, 0);
    halo_store();
    w_store(0);
    __syncthreads();
    int chunk = 0, tg = 0;
    // fragment addresses that do not depend on the step
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = wn * (BN / 2) + nb * 32 + fr;
        boff[nb] = row * 32;
    }
    const int bkey = (fr >> 2) & 3;          // (row >> 2) & 3: the wave's row offsets are multiples of 32
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        const int ntg = tg == TG - 1 ? 0 : tg + 1, nchunk = tg == TG - 1 ? chunk + 1 : chunk;
        if (more) w_fetch(nchunk, ntg);                                    // in flight under this step's MFMAs
        const bool next_halo = chunk + 1 < nchunks;                        // workgroup-uniform
        if (tg == 0 && next_halo) halo_fetch(chunk + 1);                   // held in registers until this chunk's last step
        {
            const half_t* const bh0 = Bbase + (s & 1) * (2 * SM::kBPlane);
#pragma unroll
            for (int j = 0; j < TAPS; ++j) {
                const int tap = tg * TAPS + j;
                const int ky = TAPS == 3 ? tg : tap / 3, kx = TAPS == 3 ? j : tap - 3 * ky;
                const half_t* const bh = bh0 + j * SM::kBTap;
                const half_t* const bl = bh + SM::kBPlane;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8 ah[2], al[2], wh[NB], wl[NB];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int off = ((2 * wm + mb + ky) * C3_HW + fr + kx) * C3_PITCH + ks * 16 + fk;
                        ah[mb] = *reinterpret_cast<const half8*>(Hhi + off);
                        al[mb] = *reinterpret_cast<const half8*>(Hlo + off);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int off = boff[nb] + (((2 * ks + (lane >> 5)) ^ bkey) * 8);
                        wh[nb] = *reinterpret_cast<const half8*>(bh + off);
                        wl[nb] = *reinterpret_cast<const half8*>(bl + off);
                    }
                    // the two small terms first, then the leading one (f32x3_igemm_kernel's order)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], wh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], wl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], wh[nb], acc[mb][nb], 0, 0, 0);
                }
            }
        }
        if (tg == TG - 1 && next_halo) {
            __syncthreads();                     // every wave has read this chunk's halo
            halo_store();
        }
        if (more) w_store((s + 1) & 1);
        __syncthreads();                         // the next step's weights (and halo) are visible; nobody reads this step's weights any more
        tg = ntg;
        chunk = nchunk;
    }
    if (p.range_flag && range_max > 65504.f) atomicOr(p.range_flag, 1);          // as f32x3_igemm_kernel: reported, never a silent inf

    // ---- epilogue straight from the accumulator layout: register r of block (mb, nb) = pixel column (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)
    // of patch row 2 wm + mb, channel lane & 31 of the block -- 32 lanes cover 128 contiguous bytes of one pixel
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + wn * (BN / 2) + nb * 32 + fr;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float ws = p.wscale ? p.wscale[n] : 1.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int y = y0 + 2 * wm + mb;
            if (y >= p.H) continue;
            float* const orow = p.out + ((long)(img * p.H + y) * p.W) * p.ldc + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                if (x >= p.W) continue;
                float v = acc[mb][nb][r] * ws + bias;
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = gelu_erf(v);
                orow[(long)x * p.ldc] = v;
            }
        }
    }
}

template <int BN, int TAPS>
int c3_launch(const F32GemmParams& p, hipStream_t s) {
    constexpr int smem = C3Smem<BN, TAPS>::kBytes;
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_set{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr_set)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&f32x3_conv3x3_kernel<BN, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        mark_on_device(attr_set);
    }
    const int tiles_x = ceil_div(p.W, C3_TW), tiles_y = ceil_div(p.H, C3_TH), tiles_n = ceil_div(p.Cout, BN);
    const long n_img = p.M / ((long)p.Ho * p.Wo);
    const long grid = n_img * tiles_y * tiles_x * tiles_n;
    if (grid > 0x7fffffffL) return DVID_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((f32x3_conv3x3_kernel<BN, TAPS>), dim3((unsigned)grid), dim3(512), smem, s, p, tiles_x, tiles_y, tiles_n);
    LAUNCH_CHECK();
    return DVID_OK;
}

}  // namespace

// the layer type fits: split operands present, 3x3 / stride 1 / pad 1, input channels a multiple of 32 with un-padded weight rows, no residual
bool dvid_f32_conv3x3_supported(const F32GemmParams& p) {
    if (!p.w_hi || !p.w_lo) return false;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Ho != p.H || p.Wo != p.W) return false;
    if (p.Cin % 32 || p.K != 9 * p.Cin || p.Kpad != p.K) return false;
    if (p.res_mode != 0 || p.relu > 2) return false;
    if (p.M % ((long)p.Ho * p.Wo)) return false;
    return true;
}

int dvid_f32_conv3x3_launch(const F32GemmParams& p, hipStream_t s) {
    if (!dvid_f32_conv3x3_supported(p)) return DVID_ERR_UNSUPPORTED;
    return p.Cout <= 64 ? c3_launch<64, 1>(p, s) : c3_launch<128, 3>(p, s);
}
