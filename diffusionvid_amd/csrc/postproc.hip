// Detection post-processing on device (diffusion_det.py:754-839 `inference`, :607-627 ensemble):
//
//   kernel A  topk_candidates: per (frame, candidate set) sigmoid over [M, C] logits and the
//             top-M of the M*C scores, ordered by (score desc, flat index asc)
//   kernel B  nms_frame: per frame, merge the sets (stable: score desc, position asc), class-aware
//             NMS with torchvision's coordinate trick (boxes + label * (max_coord + 1), IoU
//             without +1, `>` threshold, fp32, same operation order as torchvision's kernels),
//             clip_to_image (bounding_box.py:214-224), compacted outputs + count.
//
// Integer work (ordering, suppression) is exact given the scores/boxes; no D2H copies.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned long long u64;

// ascending bitonic sort of n (power of two) keys in LDS by all threads of the block
__device__ void bitonic_sort_u64(u64* keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int p = i | j;
                const bool up = (i & k) == 0;
                const u64 a = keys[i], b = keys[p];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ unsigned f2u(float f) { return __float_as_uint(f); }

// grid (frames, sets); logits/boxes for set s of frame f at ((s * n_img + f) * m) rows.
__global__ __launch_bounds__(1024) void topk_candidates_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                                int n_img, int m, int c, int npad, float* __restrict__ cand_boxes,
                                                                float* __restrict__ cand_scores, int* __restrict__ cand_labels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int f = blockIdx.x, set = blockIdx.y;
    const int nsets = gridDim.y;
    const long base = ((long)set * n_img + f) * m;
    const int total = m * c;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        u64 key = ~0ull;
        if (i < total) {
            const float x = logits[base * c + i];
            const float sc = 1.f / (1.f + expf(-x));                 // torch.sigmoid
            key = ((u64)(~f2u(sc)) << 32) | (unsigned)i;             // score desc, index asc (scores > 0)
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_u64(keys, npad);
    const long obase = ((long)f * nsets + set) * m;
    for (int r = threadIdx.x; r < m; r += blockDim.x) {
        const u64 key = keys[r];
        const unsigned idx = (unsigned)key;
        const float sc = __uint_as_float(~(unsigned)(key >> 32));
        cand_scores[obase + r] = sc;
        cand_labels[obase + r] = (int)(idx % c) + 1;
        const float4v b = *reinterpret_cast<const float4v*>(boxes + (base + idx / c) * 4);
        *reinterpret_cast<float4v*>(cand_boxes + (obase + r) * 4) = b;
    }
}

// The same selection without sorting all M*C keys: a three-pass radix select (11 + 11 + 10 bits, LDS histograms) finds the M-th
// smallest 32-bit key (= bit pattern of the M-th largest score), the keys below it and -- in index order, by a block-wide prefix sum
// over per-thread contiguous ranges -- as many keys equal to it as complete the M are compacted, and only those M keys are sorted
// (512 instead of 16384 keys for 300 x 30: 45 compare-exchange rounds instead of 105 over 32x fewer keys).  Exactly the set and
// order of the full sort: (score desc, flat index asc).  One workgroup of 1024 threads per (frame, set).
__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, int n_img,
                                                            int m, int c, int mpad, float* __restrict__ cand_boxes,
                                                            float* __restrict__ cand_scores, int* __restrict__ cand_labels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = m * c;
    u64* sel = reinterpret_cast<u64*>(smem);                              // [mpad] selected keys
    unsigned* k32 = reinterpret_cast<unsigned*>(sel + mpad);             // [total] ~bits(score)
    int* hist = reinterpret_cast<int*>(k32 + total);                     // [2048]
    __shared__ int s_bin, s_need, s_cnt, s_wave[16];
    const int f = blockIdx.x, set = blockIdx.y, nsets = gridDim.y, tid = threadIdx.x;
    const long base = ((long)set * n_img + f) * m;
    for (int i = tid; i < total; i += 1024) {
        const float x = logits[base * c + i];
        k32[i] = ~f2u(1.f / (1.f + expf(-x)));                           // torch.sigmoid; smaller key = larger score (scores > 0)
    }
    for (int i = tid; i < mpad; i += 1024) sel[i] = ~0ull;
    if (tid == 0) {
        s_need = m;
        s_cnt = 0;
    }
    unsigned prefix = 0;
    int decided = 0;                                                     // leading bits of the threshold known so far
    const int bits_of[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int bits = bits_of[pass], shift = 32 - decided - bits;
        for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < total; i += 1024) {
            const unsigned k = k32[i];
            if (decided == 0 || (k >> (32 - decided)) == prefix) atomicAdd(&hist[(k >> shift) & ((1u << bits) - 1)], 1);
        }
        __syncthreads();
        if (tid < 64) {                                                  // first wave: the bin that holds the s_need-th key
            const int nb = 1 << bits, per = nb / 64;
            int sum = 0;
            for (int b = 0; b < per; ++b) sum += hist[tid * per + b];
            int inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if (tid >= o) inc += v;
            }
            const int need = s_need, exc = inc - sum;
            if (exc < need && need <= inc) {
                int run = exc;
                for (int b = 0; b < per; ++b) {
                    const int h = hist[tid * per + b];
                    if (need <= run + h) {
                        s_bin = tid * per + b;
                        s_need = need - run;
                        break;
                    }
                    run += h;
                }
            }
        }
        __syncthreads();
        prefix = (prefix << bits) | (unsigned)s_bin;
        decided += bits;
        __syncthreads();
    }
    const unsigned thr = prefix;                                         // the m-th smallest key; s_need of the keys equal to it belong to the top-m
    const int need_eq = s_need;
    // keys below the threshold: any order (they are sorted afterwards)
    for (int i = tid; i < total; i += 1024) {
        const unsigned k = k32[i];
        if (k < thr) sel[atomicAdd(&s_cnt, 1)] = ((u64)k << 32) | (unsigned)i;
    }
    // keys equal to it: the first need_eq in index order (per-thread contiguous ranges + block-wide exclusive prefix sum)
    const int per = (total + 1023) / 1024, i0 = tid * per, i1 = min(total, i0 + per);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += k32[i] == thr;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += v;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = inc;
    __syncthreads();
    int rank = inc - mine;
    for (int w = 0; w < (tid >> 6); ++w) rank += s_wave[w];
    const int below = s_cnt;                                             // final: every k < thr has been counted before the barrier above
    for (int i = i0; i < i1; ++i)
        if (k32[i] == thr) {
            if (rank < need_eq) sel[below + rank] = ((u64)thr << 32) | (unsigned)i;
            ++rank;
        }
    __syncthreads();
    bitonic_sort_u64(sel, mpad);
    const long obase = ((long)f * nsets + set) * m;
    for (int r = tid; r < m; r += 1024) {
        const u64 key = sel[r];
        const unsigned idx = (unsigned)key;
        cand_scores[obase + r] = __uint_as_float(~(unsigned)(key >> 32));
        cand_labels[obase + r] = (int)(idx % c) + 1;
        *reinterpret_cast<float4v*>(cand_boxes + (obase + r) * 4) = *reinterpret_cast<const float4v*>(boxes + (base + idx / c) * 4);
    }
}

// one workgroup (1024 threads) per frame; n = nsets * m candidates (<= 1024)
__global__ __launch_bounds__(1024) void nms_frame_kernel(const float* __restrict__ cand_boxes, const float* __restrict__ cand_scores,
                                                          const int* __restrict__ cand_labels, int n, int npad, float img_w,
                                                          float img_h, float iou_thr, int use_nms, int out_cap,
                                                          float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                          int* __restrict__ out_labels, int* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int words = (n + 63) >> 6;
    u64* keys = reinterpret_cast<u64*>(smem);                          // [npad]
    float* bx = reinterpret_cast<float*>(keys + npad);                 // [n][4] offset boxes, sorted order
    float* area = bx + 4 * n;                                          // [n]
    int* order = reinterpret_cast<int*>(area + n);                     // [n]
    float* redf = reinterpret_cast<float*>(order + n);                 // [16]
    int* keep_slot = reinterpret_cast<int*>(redf + 16);                // [n]
    u64* mask = reinterpret_cast<u64*>(keep_slot + ((n + 1) & ~1));    // [n][words]
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* cb = cand_boxes + (long)f * n * 4;
    const float* cs = cand_scores + (long)f * n;
    const int* cl = cand_labels + (long)f * n;

    float mx = -INFINITY;
    for (int i = tid; i < npad; i += blockDim.x) {
        u64 key = ~0ull;
        if (i < n) {
            key = ((u64)(~f2u(cs[i])) << 32) | (unsigned)i;
            const float4v b = *reinterpret_cast<const float4v*>(cb + i * 4);
            mx = fmaxf(mx, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
        }
        keys[i] = key;
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    bitonic_sort_u64(keys, npad);
    float max_coord = redf[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) max_coord = fmaxf(max_coord, redf[w]);
    const float off_unit = max_coord + 1.0f;
    for (int r = tid; r < n; r += blockDim.x) {
        const int i = (int)(unsigned)keys[r];
        order[r] = i;
        const float4v b = *reinterpret_cast<const float4v*>(cb + i * 4);
        const float off = (float)cl[i] * off_unit;
        const float x1 = b[0] + off, y1 = b[1] + off, x2 = b[2] + off, y2 = b[3] + off;
        bx[r * 4 + 0] = x1;
        bx[r * 4 + 1] = y1;
        bx[r * 4 + 2] = x2;
        bx[r * 4 + 3] = y2;
        area[r] = (x2 - x1) * (y2 - y1);
    }
    __syncthreads();
    if (use_nms) {
        // suppression bit matrix: mask[i][w] bit j = IoU(i, 64w + j) > thr for 64w + j > i
        for (int t = tid; t < n * words; t += blockDim.x) {
            const int i = t / words, w = t - i * words;
            const float ix1 = bx[i * 4], iy1 = bx[i * 4 + 1], ix2 = bx[i * 4 + 2], iy2 = bx[i * 4 + 3];
            const float ia = area[i];
            u64 bits = 0;
            const int j0 = w << 6;
            const int jend = min(64, n - j0);
            for (int jj = 0; jj < jend; ++jj) {
                const int j = j0 + jj;
                if (j <= i) continue;
                const float xx1 = fmaxf(ix1, bx[j * 4]), yy1 = fmaxf(iy1, bx[j * 4 + 1]);
                const float xx2 = fminf(ix2, bx[j * 4 + 2]), yy2 = fminf(iy2, bx[j * 4 + 3]);
                const float ww = fmaxf(0.f, xx2 - xx1), hh = fmaxf(0.f, yy2 - yy1);
                const float inter = ww * hh;
                const float ovr = inter / (ia + area[j] - inter);
                if (ovr > iou_thr) bits |= 1ull << jj;
            }
            mask[(long)i * words + w] = bits;
        }
    }
    __syncthreads();
    // greedy sweep by wave 0: lane w owns word w of the removed set
    if (tid < 64) {
        u64 removed = 0;
        int nkeep = 0;
        for (int i = 0; i < n; ++i) {
            const u64 wrd = __shfl(removed, i >> 6, 64);
            const bool alive = !((wrd >> (i & 63)) & 1ull);
            if (alive) {
                if (tid == 0) keep_slot[nkeep] = i;
                ++nkeep;
                if (use_nms && tid < words) removed |= mask[(long)i * words + tid];
            }
        }
        if (tid == 0) out_counts[f] = nkeep;
        redf[0] = __int_as_float(nkeep);
    }
    __syncthreads();
    const int nkeep = __float_as_int(redf[0]);
    for (int s = tid; s < out_cap; s += blockDim.x) {
        float4v b = {0.f, 0.f, 0.f, 0.f};
        float sc = 0.f;
        int lb = 0;
        if (s < nkeep) {
            const int i = order[keep_slot[s]];
            b = *reinterpret_cast<const float4v*>(cb + i * 4);
            b[0] = fminf(fmaxf(b[0], 0.f), img_w - 1.f);
            b[1] = fminf(fmaxf(b[1], 0.f), img_h - 1.f);
            b[2] = fminf(fmaxf(b[2], 0.f), img_w - 1.f);
            b[3] = fminf(fmaxf(b[3], 0.f), img_h - 1.f);
            sc = cs[i];
            lb = cl[i];
        }
        *reinterpret_cast<float4v*>(out_boxes + ((long)f * out_cap + s) * 4) = b;
        out_scores[(long)f * out_cap + s] = sc;
        out_labels[(long)f * out_cap + s] = lb;
    }
}

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

int dvid_topk_candidates_launch(const float* logits, const float* boxes, int n_img, int nsets, int m, int c, float* cand_boxes,
                                float* cand_scores, int* cand_labels, hipStream_t s) {
    if (n_img == 0) return DVID_OK;
    {
        // radix select + sort of the M selected keys (topk_select_kernel); shapes whose keys do not fit its LDS take the full sort below
        const int mpad = next_pow2(m);
        const size_t smem2 = (size_t)mpad * 8 + (size_t)m * c * 4 + 2048 * 4;
        if (m * c > m && smem2 <= 150 * 1024) {
            static std::atomic<unsigned long long> attr2{0};          // one bit per device: the attribute belongs to (function, device)
            if (first_on_device(attr2)) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            150 * 1024));
                mark_on_device(attr2);
            }
            hipLaunchKernelGGL(topk_select_kernel, dim3(n_img, nsets), dim3(1024), smem2, s, logits, boxes, n_img, m, c, mpad, cand_boxes,
                               cand_scores, cand_labels);
            LAUNCH_CHECK();
            return DVID_OK;
        }
    }
    const int npad = next_pow2(m * c);
    const size_t smem = (size_t)npad * 8;
    if (smem > 160 * 1024) return DVID_ERR_UNSUPPORTED;
    static std::atomic<unsigned long long> attr{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_candidates_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        mark_on_device(attr);
    }
    hipLaunchKernelGGL(topk_candidates_kernel, dim3(n_img, nsets), dim3(1024), smem, s, logits, boxes, n_img, m, c, npad, cand_boxes,
                       cand_scores, cand_labels);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_nms_frames_launch(const float* cand_boxes, const float* cand_scores, const int* cand_labels, int n_img, int n, float img_w,
                           float img_h, float iou, int use_nms, int out_cap, float* out_boxes, float* out_scores, int* out_labels,
                           int* out_counts, hipStream_t s) {
    if (n_img == 0) return DVID_OK;
    if (n > 1024 || out_cap < n) return DVID_ERR_UNSUPPORTED;
    const int npad = next_pow2(n);
    const int words = (n + 63) / 64;
    const size_t smem = (size_t)npad * 8 + (size_t)n * (16 + 4 + 4) + 64 + (size_t)((n + 1) & ~1) * 4 + (size_t)n * words * 8;
    if (smem > 160 * 1024) return DVID_ERR_UNSUPPORTED;
    static std::atomic<unsigned long long> attr{0};          // one bit per device: the attribute belongs to (function, device)
    if (first_on_device(attr)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_frame_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        mark_on_device(attr);
    }
    hipLaunchKernelGGL(nms_frame_kernel, dim3(n_img), dim3(1024), smem, s, cand_boxes, cand_scores, cand_labels, n, npad, img_w, img_h,
                       iou, use_nms, out_cap, out_boxes, out_scores, out_labels, out_counts);
    LAUNCH_CHECK();
    return DVID_OK;
}
