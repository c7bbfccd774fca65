// The per-box walk of multi-level RoIAlignV2 (aligned = True, 7 x 7 bins, 2 x 2 samples; detectron2 ROIPooler(ROIAlignV2) as called at
// box_head.py:507/:617, restated in oracle/roi_align.py), shared by csrc/roialign.hip (tile -> global memory) and csrc/dynconv.hip (tile ->
// the LDS image DynamicConv's first product reads: the fused launch).  One workgroup of 256 threads per box: 32 lanes x 8 channels
// (16-byte loads) cover the 256-channel vector of one tap, the 8 lane groups walk the 49 bins.  Coordinate math is fp32.  ONE copy of the
// arithmetic, so that both launches produce the same fp16 tile bit for bit (tests/test_gpu_kernels.py::test_roi_fused_dynconv_bit_identical).
#pragma once

#include "common.h"
#include "kernels.h"

namespace roi_taps {

constexpr int P = 7;       // pooler resolution
constexpr int G = 2;       // sampling ratio
constexpr int CV = 32;     // 8-channel vectors per pixel (C = 256)

// acc + w * float(h): one v_fma_mix_f32 per channel (the fp16 feature promoted exactly inside the FMA) instead of a convert and an
// FMA -- the gather is VALU-bound (784 taps x 256 channels per box), so this halves its instruction count
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float mix_fma_lo(unsigned int h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float mix_fma_hi(unsigned int h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ void tap_fma8(float (&acc)[8], half8 v, float w) {
    const uint4v u = __builtin_bit_cast(uint4v, v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[2 * q] = mix_fma_lo(u[q], w, acc[2 * q]);
        acc[2 * q + 1] = mix_fma_hi(u[q], w, acc[2 * q + 1]);
    }
}

struct Tap {
    int lo, hi;
    float wl, wh;
    bool ok;
};

// torchvision bilinear_interpolate, one axis
__device__ __forceinline__ Tap axis_tap(float y, int limit) {
    Tap t;
    t.ok = !(y < -1.0f || y > (float)limit);
    if (y <= 0.f) y = 0.f;
    int lo = (int)y;
    int hi;
    if (lo >= limit - 1) {
        hi = lo = limit - 1;
        y = (float)lo;
    } else {
        hi = lo + 1;
    }
    const float l = y - (float)lo;
    t.lo = lo;
    t.hi = hi;
    t.wl = 1.f - l;  // weight of lo
    t.wh = l;        // weight of hi
    return t;
}

// Bins p = grp, grp + 8, ... of box `box` for the 8 channels [8 ln, 8 ln + 8): sink(p, half8) receives each finished bin; macc
// accumulates the fp32 bin values (the caller divides by 49 for the mean over the bins).
template <typename Sink>
__device__ __forceinline__ void gather_box(const RoiLevels& lv, const float* __restrict__ boxes, int boxes_per_img, int box, int grp, int ln,
                                           float (&macc)[8], Sink&& sink) {
    const int img = box / boxes_per_img;
    const float bx1 = boxes[box * 4 + 0], by1 = boxes[box * 4 + 1], bx2 = boxes[box * 4 + 2], by2 = boxes[box * 4 + 3];
    // detectron2 assign_boxes_to_levels (canonical 224 / level 4, levels 3..5)
    const float area = (bx2 - bx1) * (by2 - by1);
    const bool valid_box = area >= 0.f;  // NaN / negative area: upstream matches no level -> zeros
    float lvf = floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 3.f), 5.f);
    const int level = valid_box ? (int)lvf - 3 : 0;

    const half_t* feat = lv.feat[level];
    const int H = lv.h[level], W = lv.w[level];
    const float sc = lv.scale[level];
    feat += (long)img * H * W * (CV * 8);

    const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f;
    const float x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
    const float bin_w = (x2 - x1) / P, bin_h = (y2 - y1) / P;

#pragma unroll
    for (int e = 0; e < 8; ++e) macc[e] = 0.f;

    for (int p = grp; p < P * P; p += 8) {
        const int ph = p / P, pw = p - ph * P;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (valid_box) {
#pragma unroll
            for (int iy = 0; iy < G; ++iy) {
                const float y = y1 + ph * bin_h + (iy + 0.5f) * bin_h / G;
                const Tap ty = axis_tap(y, H);
#pragma unroll
                for (int ix = 0; ix < G; ++ix) {
                    const float x = x1 + pw * bin_w + (ix + 0.5f) * bin_w / G;
                    const Tap tx = axis_tap(x, W);
                    if (!(ty.ok && tx.ok)) continue;
                    const half8 v1 = *reinterpret_cast<const half8*>(feat + ((long)ty.lo * W + tx.lo) * (CV * 8) + ln * 8);
                    const half8 v2 = *reinterpret_cast<const half8*>(feat + ((long)ty.lo * W + tx.hi) * (CV * 8) + ln * 8);
                    const half8 v3 = *reinterpret_cast<const half8*>(feat + ((long)ty.hi * W + tx.lo) * (CV * 8) + ln * 8);
                    const half8 v4 = *reinterpret_cast<const half8*>(feat + ((long)ty.hi * W + tx.hi) * (CV * 8) + ln * 8);
                    const float w1 = ty.wl * tx.wl, w2 = ty.wl * tx.wh, w3 = ty.wh * tx.wl, w4 = ty.wh * tx.wh;
                    tap_fma8(acc, v1, w1);
                    tap_fma8(acc, v2, w2);
                    tap_fma8(acc, v3, w3);
                    tap_fma8(acc, v4, w4);
                }
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] *= 1.f / (G * G);
            macc[e] += acc[e];
            o[e] = (half_t)acc[e];
        }
        sink(p, o);
    }
}

}  // namespace roi_taps
