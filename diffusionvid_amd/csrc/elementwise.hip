// HBM-bound row / pixel kernels: image prep, max-pool, layout converts, fused
// residual+LayerNorm, time/cond modulation, SiLU.  All loads/stores are 8-16 bytes per lane.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "options.h"

namespace {

// fp32 NCHW [n,3,h,w] in [0,1] -> normalised fp16 NHWC8 (channels 3..7 zero).
// diffusion_det.py:301-303 normalizer fused with the layout change the stem conv wants.
// Frames arrive as a table of per-frame pointers (kernel argument, up to FrameTable::kMax frames per launch): the caller's
// frames need not be one contiguous [n, 3, h, w] tensor -- the reference hands the detector a list of per-frame tensors
// (diffusion_det.py:418-421 concatenates them; here nothing is copied).
__global__ void prep_images_kernel(FrameTable in, half_t* __restrict__ out, long npix, long hw, float m0, float m1,
                                   float m2, float s0, float s1, float s2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const long img = i / hw, pix = i - img * hw;
    const float* p = in.p[img] + pix;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    v[0] = (half_t)((p[0] - m0) * s0);
    v[1] = (half_t)((p[hw] - m1) * s1);
    v[2] = (half_t)((p[2 * hw] - m2) * s2);
    *reinterpret_cast<half8*>(out + i * 8) = v;
}

// The same normaliser with a 2x2 space-to-depth layout: out[n][Y][X][(dy*2 + dx)*3 + c] (12 channels + 4 zero = 32 bytes per
// 2x2 pixel block).  The 7x7 / stride-2 stem convolution over 3 channels is then a 4x4 / stride-1 convolution over these 16
// channels (csrc/model.hip: make_stem_s2d): K = 256 instead of 448 padded columns and half the input bytes.
__global__ void prep_images_s2d_kernel(FrameTable in, half_t* __restrict__ out, long nblk, int h2, int w2, float m0,
                                       float m1, float m2, float s0, float s1, float s2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    const long per = (long)h2 * w2;
    const long img = i / per, b = i - img * per;
    const int Y = (int)(b / w2), X = (int)(b - (long)Y * w2);
    const long w = 2L * w2, hw = 4 * per;
    const float* p = in.p[img] + (2L * Y) * w + 2 * X;
    half8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {0, 0, 0, 0, 0, 0, 0, 0};
    half_t v[12];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float* q = p + dy * w + dx;
            v[(dy * 2 + dx) * 3 + 0] = (half_t)((q[0] - m0) * s0);
            v[(dy * 2 + dx) * 3 + 1] = (half_t)((q[hw] - m1) * s1);
            v[(dy * 2 + dx) * 3 + 2] = (half_t)((q[2 * hw] - m2) * s2);
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) lo[e] = v[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = v[8 + e];
    *reinterpret_cast<half8*>(out + i * 16) = lo;
    *reinterpret_cast<half8*>(out + i * 16 + 8) = hi;
}

// 3x3 stride-2 pad-1 max pool, NHWC fp16, one lane per 8-channel vector of an output pixel.
__global__ void maxpool_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int n, int h, int w, int c, int ho, int wo) {
    const int cv = c >> 3;
    const long total = (long)n * ho * wo * cv;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int v = i % cv;
    long t = i / cv;
    const int ox = t % wo;
    t /= wo;
    const int oy = t % ho;
    const int img = t / ho;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if ((unsigned)ix >= (unsigned)w) continue;
            const half8 x = *reinterpret_cast<const half8*>(in + (((long)img * h + iy) * w + ix) * c + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], (float)x[e]);
        }
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)best[e];
    *reinterpret_cast<half8*>(out + i * 8) = o;
}

// NHWC fp16 -> NCHW fp32 through an LDS transpose tile (32 pixels x 32 channels).
__global__ void nchw_from_nhwc_kernel(const half_t* __restrict__ in, float* __restrict__ out, int hw, int c) {
    __shared__ float tile[32][33];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, ch = c0 + tx;
        tile[r][tx] = (p < hw && ch < c) ? (float)in[((long)img * hw + p) * c + ch] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ch = c0 + r, p = p0 + tx;
        if (p < hw && ch < c) out[((long)img * c + ch) * hw + p] = tile[tx][r];
    }
}

__global__ void nhwc_from_nchw_kernel(const float* __restrict__ in, half_t* __restrict__ out, int hw, int c) {
    __shared__ float tile[32][33];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ch = c0 + r, p = p0 + tx;
        tile[r][tx] = (p < hw && ch < c) ? in[((long)img * c + ch) * hw + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, ch = c0 + tx;
        if (p < hw && ch < c) out[((long)img * hw + p) * c + ch] = (half_t)tile[tx][r];
    }
}

// y = LayerNorm(x + r) (eps 1e-5), optional ReLU; one wave per row, D = 64 * VPL * 4 / ... generic loop.
// Statistics in fp32 (two-pass on registers), as apex amp keeps layer_norm in fp32.
template <int MAXV>
__global__ void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ g,
                                     const float* __restrict__ b, float* __restrict__ y32, half_t* __restrict__ y16, int rows, int d,
                                     int relu, int nsplit, long split_stride, const float* __restrict__ xbias) {
    // no mul + add contraction: add_layernorm_rows_kernel below must round exactly like this kernel, and which products the
    // compiler fuses otherwise depends on the code around them
#pragma clang fp contract(off)
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int nv = d >> 2;  // float4 vectors per row
    float4v v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + i * 64;
        if (j < nv) {
            v[i] = *reinterpret_cast<const float4v*>(x + (long)row * d + j * 4);
            for (int sp = 1; sp < nsplit; ++sp)      // split-K partial slabs of the producing GEMM
                v[i] += *reinterpret_cast<const float4v*>(x + sp * split_stride + (long)row * d + j * 4);
            if (xbias) v[i] += *reinterpret_cast<const float4v*>(xbias + j * 4);
            if (r) v[i] += *reinterpret_cast<const float4v*>(r + (long)row * d + j * 4);
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(sum) / d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + i * 64;
        if (j < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                sq += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / d + 1e-5f);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int j = lane + i * 64;
        if (j < nv) {
            const float4v gg = *reinterpret_cast<const float4v*>(g + j * 4);
            const float4v bb = *reinterpret_cast<const float4v*>(b + j * 4);
            float4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
            if (y32) *reinterpret_cast<float4v*>(y32 + (long)row * d + j * 4) = o;
            if (y16) {
                half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                *reinterpret_cast<half4*>(y16 + (long)row * d + j * 4) = h;
            }
        }
    }
}

// The same LayerNorm for the short rows (d = 4 * LPR exactly: 128 or 256, one slab): R * (64 / LPR) rows per wave, every lane one
// float4 of each of its R rows and all of a wave's loads in flight before its first reduction.  One row per wave keeps a single
// 512-byte / 1-KB load per wave outstanding (32 waves per CU: ~2 TB/s at the memory latency; at d = 128 half of the lanes idle).
// Per-row arithmetic is add_layernorm_kernel's operation by operation -- lane sums in the same order, the same xor butterfly
// (its 32-lane step adds the idle half's zeros when d = 128), no contraction in either kernel, division by the power of two d exact
// either way -- so the results are bit-identical (tests/test_gpu_kernels.py::test_add_layernorm_rows_per_wave_bit_identical).
template <int LPR, int R>
__global__ __launch_bounds__(256) void add_layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                 const float* __restrict__ g, const float* __restrict__ b,
                                                                 float* __restrict__ y32, half_t* __restrict__ y16, int rows, int relu,
                                                                 const float* __restrict__ xbias) {
#pragma clang fp contract(off)
    constexpr int SUB = 64 / LPR, RPW = R * SUB, D = 4 * LPR;
    const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int lane = threadIdx.x & 63, l = lane % LPR, sub = lane / LPR;
    float4v v[R], rv[R];
    long off[R];
    bool ok[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const long row = row0 + i * SUB + sub;
        ok[i] = row < rows;
        off[i] = (ok[i] ? row : (long)rows - 1) * D + l * 4;          // rows past the end re-read the last row and store nothing
        v[i] = *reinterpret_cast<const float4v*>(x + off[i]);
    }
    if (r) {
#pragma unroll
        for (int i = 0; i < R; ++i) rv[i] = *reinterpret_cast<const float4v*>(r + off[i]);
    }
    const float4v gg = *reinterpret_cast<const float4v*>(g + l * 4);
    const float4v bb = *reinterpret_cast<const float4v*>(b + l * 4);
    if (xbias) {
        const float4v xb = *reinterpret_cast<const float4v*>(xbias + l * 4);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] += xb;
    }
    if (r) {
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] += rv[i];
    }
    float sum[R], sq[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        sum[i] = 0.f;
        sum[i] += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < R; ++i) sum[i] += __shfl_xor(sum[i], o, 64);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float mean = sum[i] / D;
        sum[i] = mean;
        sq[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[i][e] - mean;
            sq[i] += t * t;
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < R; ++i) sq[i] += __shfl_xor(sq[i], o, 64);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float mean = sum[i];
        const float rstd = rsqrtf(sq[i] / D + 1e-5f);
        float4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            if (relu) o[e] = fmaxf(o[e], 0.f);
        }
        if (ok[i]) {
            if (y32) *reinterpret_cast<float4v*>(y32 + off[i]) = o;
            if (y16) {
                half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                *reinterpret_cast<half4*>(y16 + off[i]) = h;
            }
        }
    }
}

// Swin PatchMerging front half (swintransformer.py:296-319): gather the 2x2 neighbourhood
// [x(2i,2j) | x(2i+1,2j) | x(2i,2j+1) | x(2i+1,2j+1)] (zeros beyond an odd H/W), LayerNorm over 4C, fp16 out.
// One wave per output token; the bias-free reduction Linear(4C -> 2C) that follows is an igemm launch.
// C <= 512: every lane holds up to 2 float4 of each of the 4 source tokens (no per-element division, 32 live registers)
__global__ __launch_bounds__(256) void patch_merge_ln_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ b, half_t* __restrict__ y16, float* __restrict__ y32, int B, int H,
                                                             int W, int C) {
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const long tok = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tok >= (long)B * H2 * W2) return;
    const int lane = threadIdx.x & 63;
    const int ox = tok % W2;
    const long t2 = tok / W2;
    const int oy = t2 % H2;
    const int img = t2 / H2;
    const int cv = C >> 2;         // float4 vectors per source token (<= 128)
    float4v v[4][2];
    float sum = 0.f;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        const int y = 2 * oy + (part & 1), xx = 2 * ox + (part >> 1);
        const bool inside = y < H && xx < W;
        const float* src = x + (((long)img * H + (inside ? y : 0)) * W + (inside ? xx : 0)) * C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c4 = lane + i * 64;
            v[part][i] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (inside && c4 < cv) v[part][i] = *reinterpret_cast<const float4v*>(src + c4 * 4);
            sum += v[part][i][0] + v[part][i][1] + v[part][i][2] + v[part][i][3];
        }
    }
    const int d = 4 * C;
    const float mean = wave_sum(sum) / d;
    float sq = 0.f;
#pragma unroll
    for (int part = 0; part < 4; ++part)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + i * 64 < cv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[part][i][e] - mean;
                    sq += t * t;
                }
            }
    const float rstd = rsqrtf(wave_sum(sq) / d + 1e-5f);
#pragma unroll
    for (int part = 0; part < 4; ++part)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < cv) {
                const int j = part * cv + c4;
                const float4v gg = *reinterpret_cast<const float4v*>(g + j * 4);
                const float4v bb = *reinterpret_cast<const float4v*>(b + j * 4);
                float4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[part][i][e] - mean) * rstd * gg[e] + bb[e];
                if (y32) *reinterpret_cast<float4v*>(y32 + tok * d + j * 4) = o;          // DTYPE float32
                if (y16) *reinterpret_cast<half4*>(y16 + tok * d + j * 4) = (half4){(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            }
        }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4*>(y + i * 4) = h;
}

__global__ void silu_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    half4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (half_t)(v[e] / (1.f + expf(-v[e])));
    *reinterpret_cast<half4*>(y + i * 4) = h;
}

// box_head.py:533-536 / :643-647: fc = x * (scale + 1) + shift
__global__ void modulate_kernel(const float* __restrict__ x, const float* __restrict__ scale, int scale_ld,
                                const float* __restrict__ shift, int shift_per_row, int shift_ld, half_t* __restrict__ y16, long n4,
                                int rows_per_frame, int d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int dv = d >> 2;
    const long row = i / dv;
    const int col = (int)(i - row * dv) * 4;
    const long frame = row / rows_per_frame;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    const float4v sc = *reinterpret_cast<const float4v*>(scale + frame * scale_ld + col);
    const float4v sh = *reinterpret_cast<const float4v*>(shift + (shift_per_row ? row : frame) * shift_ld + col);
    half4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (half_t)(v[e] * (sc[e] + 1.f) + sh[e]);
    *reinterpret_cast<half4*>(y16 + i * 4) = h;
}

}  // namespace

int dvid_prep_images_launch(const float* const* frames, half_t* nhwc8, int n, int h, int w, const float* mean, const float* inv_std,
                            hipStream_t s) {
    const long hw = (long)h * w;
    for (int f0 = 0; f0 < n; f0 += FrameTable::kMax) {
        const int nf = n - f0 < FrameTable::kMax ? n - f0 : FrameTable::kMax;
        FrameTable tab;
        for (int i = 0; i < FrameTable::kMax; ++i) tab.p[i] = frames[f0 + (i < nf ? i : 0)];
        const long npix = hw * nf;
        hipLaunchKernelGGL(prep_images_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, tab, nhwc8 + (long)f0 * hw * 8, npix, hw,
                           mean[0], mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2]);
        LAUNCH_CHECK();
    }
    return DVID_OK;
}

int dvid_prep_images_s2d_launch(const float* const* frames, half_t* s2d16, int n, int h, int w, const float* mean, const float* inv_std,
                                hipStream_t s) {
    if ((h | w) & 1) return DVID_ERR_ARG;
    const long per = (long)(h / 2) * (w / 2);
    for (int f0 = 0; f0 < n; f0 += FrameTable::kMax) {
        const int nf = n - f0 < FrameTable::kMax ? n - f0 : FrameTable::kMax;
        FrameTable tab;
        for (int i = 0; i < FrameTable::kMax; ++i) tab.p[i] = frames[f0 + (i < nf ? i : 0)];
        const long nblk = per * nf;
        hipLaunchKernelGGL(prep_images_s2d_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s, tab, s2d16 + (long)f0 * per * 16, nblk,
                           h / 2, w / 2, mean[0], mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2]);
        LAUNCH_CHECK();
    }
    return DVID_OK;
}

int dvid_maxpool3x3s2_launch(const half_t* in, half_t* out, int n, int h, int w, int c, hipStream_t s) {
    if (c % 8) return DVID_ERR_ARG;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, n, h, w, c, ho, wo);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_nchw_from_nhwc_launch(const half_t* in, float* out, int n, int h, int w, int c, hipStream_t s) {
    const int hw = h * w;
    hipLaunchKernelGGL(nchw_from_nhwc_kernel, dim3(ceil_div(hw, 32), ceil_div(c, 32), n), dim3(256), 0, s, in, out, hw, c);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_nhwc_from_nchw_launch(const float* in, half_t* out, int n, int h, int w, int c, hipStream_t s) {
    const int hw = h * w;
    hipLaunchKernelGGL(nhwc_from_nchw_kernel, dim3(ceil_div(hw, 32), ceil_div(c, 32), n), dim3(256), 0, s, in, out, hw, c);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_add_layernorm_launch(const float* x, const float* r, const float* g, const float* b, float* y32, half_t* y16, int rows,
                              int d, int relu, hipStream_t s, int nsplit, long split_stride, const float* xbias) {
    if (d % 4 || d > 1024) return DVID_ERR_ARG;
    const int wpb = 4;
    if (g_opt.ln_rows && nsplit == 1 && rows > 0 && (d == 128 || d == 256)) {
        constexpr int R = 4;
        const dim3 blk(64 * wpb);
        if (d == 128)
            hipLaunchKernelGGL((add_layernorm_rows_kernel<32, R>), dim3(ceil_div(rows, 2 * R * wpb)), blk, 0, s, x, r, g, b, y32, y16, rows, relu, xbias);
        else
            hipLaunchKernelGGL((add_layernorm_rows_kernel<64, R>), dim3(ceil_div(rows, R * wpb)), blk, 0, s, x, r, g, b, y32, y16, rows, relu, xbias);
        LAUNCH_CHECK();
        return DVID_OK;
    }
    const dim3 grid(ceil_div(rows, wpb)), block(64 * wpb);
    if (d <= 256)
        hipLaunchKernelGGL(add_layernorm_kernel<1>, grid, block, 0, s, x, r, g, b, y32, y16, rows, d, relu, nsplit, split_stride, xbias);
    else
        hipLaunchKernelGGL(add_layernorm_kernel<4>, grid, block, 0, s, x, r, g, b, y32, y16, rows, d, relu, nsplit, split_stride, xbias);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_f32_to_f16_launch(const float* x, half_t* y, long n, hipStream_t s) {
    if (n % 4) return DVID_ERR_ARG;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, y, n / 4);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_silu_f16_launch(const float* x, half_t* y, long n, hipStream_t s) {
    if (n % 4) return DVID_ERR_ARG;
    hipLaunchKernelGGL(silu_f16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, y, n / 4);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_modulate_launch(const float* x, const float* scale, int scale_ld, const float* shift, int shift_per_row, int shift_ld,
                         half_t* y16, int rows, int rows_per_frame, int d, hipStream_t s) {
    if (d % 4) return DVID_ERR_ARG;
    const long n4 = (long)rows * d / 4;
    hipLaunchKernelGGL(modulate_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, scale, scale_ld, shift, shift_per_row,
                       shift_ld, y16, n4, rows_per_frame, d);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_patch_merge_ln_launch(const float* x, const float* g, const float* b, half_t* y16, int B, int H, int W, int C, hipStream_t s, float* y32) {
    if (C % 4 || C > 512) return DVID_ERR_UNSUPPORTED;
    const long ntok = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
    const int wpb = 4;
    hipLaunchKernelGGL(patch_merge_ln_kernel, dim3((unsigned)((ntok + wpb - 1) / wpb)), dim3(64 * wpb), 0, s, x, g, b, y16, y32, B, H, W, C);
    LAUNCH_CHECK();
    return DVID_OK;
}
