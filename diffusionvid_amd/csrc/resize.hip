// Pillow-identical BILINEAR resize of uint8 RGB frames on the GPU + ToTensor (/255) + zero padding.
//
// Replaces, for frames that arrive as uint8 at their native size, the host-side `Resize` + `ToTensor` of the reference's
// test pipeline (mega_core/data/transforms/build.py:89-97, transforms.py:31-70: torchvision F.resize on a PIL image =
// PIL.Image.resize(BILINEAR)) and the bottom/right zero padding of to_image_list (structures/image_list.py:54-61).
// Pillow's resample is two separable integer passes (src/libImaging/Resample.c): weights in 22-bit fixed point, an int32
// dot product over the filter support, + 2^21, >> 22, clamp to uint8 -- with the uint8 rounding BETWEEN the passes, which
// is why this is two kernels and why the result is bit-identical to the CPU library.  The weight tables are built on the
// host in double precision (diffusionvid_amd/data/transforms.py: resample_tables).
// HBM-bound byte work: pass 1 reads H*W*3 and writes H*ow*3 bytes, pass 2 reads that and writes 3*PH*PW floats
// (12 bytes per output pixel dominate).  One thread per output pixel (3 channels); consecutive threads walk along x, so
// pass-2 stores are coalesced per channel plane and the loads of neighbouring threads share cache lines.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int kBits = 22;

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= kBits;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [h][w][3] -> dst [h][ow][3]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int h,
                                                          int w, int ow, const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          int ksize) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
    const int* k = kk + (long)x * ksize;
    const unsigned char* p = src + ((long)y * w + x0) * 3;
    int r = 1 << (kBits - 1), g = r, b = r;
    for (int i = 0; i < n; ++i) {
        const int c = k[i];
        r += p[3 * i] * c;
        g += p[3 * i + 1] * c;
        b += p[3 * i + 2] * c;
    }
    unsigned char* o = dst + ((long)y * ow + x) * 3;
    o[0] = clip8(r);
    o[1] = clip8(g);
    o[2] = clip8(b);
}

// src [h][ow][3] -> out fp32 [3][ph][pw]: rows < oh resampled (or copied when bounds == nullptr), scaled by 1/255; the rest zero
__global__ __launch_bounds__(256) void resample_v_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ out, int h, int ow,
                                                              int oh, int ph, int pw, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pw) return;
    float v[3] = {0.f, 0.f, 0.f};
    if (x < ow && y < oh) {
        if (bounds) {
            const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
            const int* k = kk + (long)y * ksize;
            const unsigned char* p = src + ((long)y0 * ow + x) * 3;
            int r = 1 << (kBits - 1), g = r, b = r;
            for (int i = 0; i < n; ++i) {
                const int c = k[i];
                r += p[0] * c;
                g += p[1] * c;
                b += p[2] * c;
                p += (long)ow * 3;
            }
            v[0] = (float)clip8(r) / 255.f;
            v[1] = (float)clip8(g) / 255.f;
            v[2] = (float)clip8(b) / 255.f;
        } else {
            const unsigned char* p = src + ((long)y * ow + x) * 3;
            v[0] = (float)p[0] / 255.f;
            v[1] = (float)p[1] / 255.f;
            v[2] = (float)p[2] / 255.f;
        }
    }
    const long plane = (long)ph * pw;
    float* o = out + (long)y * pw + x;
    o[0] = v[0];
    o[plane] = v[1];
    o[2 * plane] = v[2];
}

}  // namespace

int dvid_resize_u8_launch(const unsigned char* src, int h, int w, unsigned char* tmp, float* out, int oh, int ow, int ph, int pw,
                          const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize, hipStream_t s) {
    if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || ph < oh || pw < ow) return DVID_ERR_ARG;
    if ((xbounds == nullptr) != (ow == w) || (ybounds == nullptr) != (oh == h)) return DVID_ERR_ARG;
    const unsigned char* mid = src;
    if (xbounds) {
        if (!tmp) return DVID_ERR_ARG;
        hipLaunchKernelGGL(resample_h_kernel, dim3(ceil_div(ow, 256), h), dim3(256), 0, s, src, tmp, h, w, ow, xbounds, xk, xksize);
        LAUNCH_CHECK();
        mid = tmp;
    }
    hipLaunchKernelGGL(resample_v_f32_kernel, dim3(ceil_div(pw, 256), ph), dim3(256), 0, s, mid, out, h, ow, oh, ph, pw, ybounds, yk,
                       yksize);
    LAUNCH_CHECK();
    return DVID_OK;
}
