// Multi-level RoIAlignV2 (aligned=True, 7x7 bins, 2x2 samples) over NHWC fp16 pyramids.
//
// Replaces detectron2 ROIPooler(ROIAlignV2) as called at box_head.py:507/:617 (restated in
// oracle/roi_align.py).  HBM/L2-bound gather: one workgroup per box; 32 lanes x 8 channels
// (16-byte loads) cover the 256-channel vector of one tap, the 8 lane-groups of the block
// walk the 49 bins.  Output is [box][bin][channel] fp16 -- exactly the A operand of the
// DynamicConv batched matmul -- plus the optional fp32 mean over the 49 bins that RCNNHead
// uses as initial proposal features (box_head.py:509-510).  Coordinate math is fp32.
#include <stdlib.h>

#include "common.h"
#include "igemm_epilogue.h"
#include "kernels.h"
#include "roi_taps.h"

namespace {

using namespace roi_taps;

__global__ __launch_bounds__(256) void roialign_kernel(RoiLevels lv, const float* __restrict__ boxes, int boxes_per_img,
                                                        half_t* __restrict__ roi_out, float* __restrict__ mean_out, int nbox,
                                                        int xcd_major) {
    __shared__ float red[8][256];
    // an XCD takes one contiguous run of boxes, i.e. whole images: the boxes that gather from one image's pyramid meet in one L2
    // instead of pulling that image's lines into all eight (xcd_major = 0: round-robin, for A/B runs)
    const int box = xcd_major ? igemm_xcd_remap((int)blockIdx.x, nbox) : (int)blockIdx.x;
    const int tid = threadIdx.x;
    const int grp = tid >> 5, ln = tid & 31;
    float macc[8];
    gather_box(lv, boxes, boxes_per_img, box, grp, ln, macc,
               [&](int p, half8 o) { *reinterpret_cast<half8*>(roi_out + ((long)box * (P * P) + p) * (CV * 8) + ln * 8) = o; });
    if (mean_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[grp][ln * 8 + e] = macc[e];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][tid];
        mean_out[(long)box * 256 + tid] = s / (P * P);
    }
}

}  // namespace

int dvid_roialign_launch(const RoiLevels& lv, int channels, const float* boxes, int n_img, int boxes_per_img, half_t* roi_out,
                         float* mean_out, hipStream_t s) {
    if (channels != 256) return DVID_ERR_UNSUPPORTED;
    const int nbox = n_img * boxes_per_img;
    if (nbox == 0) return DVID_OK;
    hipLaunchKernelGGL(roialign_kernel, dim3(nbox), dim3(256), 0, s, lv, boxes, boxes_per_img, roi_out, mean_out, nbox, /*xcd_major=*/1);
    LAUNCH_CHECK();
    return DVID_OK;
}
