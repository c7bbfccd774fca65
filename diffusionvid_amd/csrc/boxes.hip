// Small box-space kernels: noise->boxes (diffusion_det.py:657-660), apply_deltas
// (box_head.py:550-590), top-75/25 memory-feature selection (box_head.py:304-317),
// x_start / pred_noise (diffusion_det.py:649-653, :666-672) and the DDIM + box-renewal step
// (diffusion_det.py:559-596).  fp32 throughout; no host synchronisation.
#include "common.h"
#include "kernels.h"

namespace {

__global__ void noise_to_boxes_kernel(const float* __restrict__ x, float* __restrict__ boxes, int n, float scale, float w, float h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4v v = *reinterpret_cast<const float4v*>(x + i * 4);
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = ((fminf(fmaxf(v[e], -scale), scale) / scale) + 1.f) / 2.f;
    float4v o;
    o[0] = (c[0] - 0.5f * c[2]) * w;
    o[1] = (c[1] - 0.5f * c[3]) * h;
    o[2] = (c[0] + 0.5f * c[2]) * w;
    o[3] = (c[1] + 0.5f * c[3]) * h;
    *reinterpret_cast<float4v*>(boxes + i * 4) = o;
}

// Counter-based N(0, 1) draws (the reference draws with torch.randn on the device, diffusion_det.py:449,:542,:587,:595; a draw here
// is a pure function of (key, element index), so calls need no generator state, ranks need no stream position, and the CPU oracle
// regenerates the same values -- oracle/noise.py).  Philox4x32-10 (Salmon et al., SC'11): counter = (element / 4, 0, 0, 0), key =
// the 64-bit draw key of the image; its four 32-bit outputs make two Box-Muller pairs in fp64 (u = (x + 0.5) 2^-32: never 0 or 1),
// rounded once to fp32: element 4 q + {0, 1} = r cos / r sin of (x0, x1), element 4 q + {2, 3} of (x2, x3).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void counter_normal_kernel(float* __restrict__ out, long per_image, int n_images, uint64_t key0) {
    const long quads = (per_image + 3) / 4;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int img = blockIdx.y;
    if (q >= quads) return;
    const uint64_t key = key0 + (uint64_t)img;
    uint32_t x[4];
    philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)q >> 32), 0u, 0u, (uint32_t)key, (uint32_t)(key >> 32), x);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)x[2 * h] + 0.5) * (1.0 / 4294967296.0), u2 = ((double)x[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
        const double r = sqrt(-2.0 * log(u1)), th = 6.283185307179586476925286766559 * u2;
        z[2 * h] = (float)(r * cos(th));
        z[2 * h + 1] = (float)(r * sin(th));
    }
    float* dst = out + (long)img * per_image + q * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (q * 4 + e < per_image) dst[e] = z[e];
}

__global__ void apply_deltas_kernel(const float* __restrict__ deltas, int delta_ld, const float* __restrict__ boxes,
                                    float* __restrict__ out, int n, float wx, float wy, float ww, float wh, float clamp,
                                    int* __restrict__ bad_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4v b = *reinterpret_cast<const float4v*>(boxes + i * 4);
    const float* d = deltas + (long)i * delta_ld;
    const float widths = b[2] - b[0], heights = b[3] - b[1];
    const float ctr_x = b[0] + 0.5f * widths, ctr_y = b[1] + 0.5f * heights;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, clamp), dh = fminf(d[3] / wh, clamp);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    float4v o;
    o[0] = pcx - 0.5f * pw;
    o[1] = pcy - 0.5f * ph;
    o[2] = pcx + 0.5f * pw;
    o[3] = pcy + 0.5f * ph;
    *reinterpret_cast<float4v*>(out + i * 4) = o;
    // box_head.py:588 `assert (pred_boxes[:, 2:] >= pred_boxes[:, :2]).all()` without a host sync
    if (bad_flag && !(o[2] >= o[0] && o[3] >= o[1])) atomicOr(bad_flag, 1);
}

// One workgroup per frame.  rank by (max logit desc, index asc); emit rows of the top-k1 / top-k2
// sets in ascending box-index order ("mask order", box_head.py:315-317).
__global__ __launch_bounds__(256) void topk_mask_kernel(const float* __restrict__ logits, int m, int c, int k1, int k2,
                                                         const float* __restrict__ feats, int d, float* __restrict__ out1,
                                                         float* __restrict__ out2) {
    extern __shared__ float sm[];
    float* val = sm;                                   // [m]
    int* sel1 = reinterpret_cast<int*>(sm + m);        // [m] output slot or -1
    int* sel2 = sel1 + m;
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < m; i += blockDim.x) {
        const float* lp = logits + ((long)f * m + i) * c;
        float mx = lp[0];
        for (int j = 1; j < c; ++j) mx = fmaxf(mx, lp[j]);
        val[i] = mx;
    }
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        const float vi = val[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const float vj = val[j];
            rank += (vj > vi) || (vj == vi && j < i);
        }
        sel1[i] = rank < k1;
        sel2[i] = rank < k2;
    }
    __syncthreads();
    if (tid == 0) {  // m <= ~1000: a serial exclusive scan is negligible here
        int p1 = 0, p2 = 0;
        for (int i = 0; i < m; ++i) {
            const int s1 = sel1[i], s2 = sel2[i];
            sel1[i] = s1 ? p1 : -1;
            sel2[i] = s2 ? p2 : -1;
            p1 += s1;
            p2 += s2;
        }
    }
    __syncthreads();
    const int dv = d >> 2;
    for (int i = tid; i < m * dv; i += blockDim.x) {
        const int row = i / dv, v = i - row * dv;
        const int s1 = sel1[row], s2 = sel2[row];
        if (s1 < 0) continue;
        const float4v x = *reinterpret_cast<const float4v*>(feats + ((long)f * m + row) * d + v * 4);
        *reinterpret_cast<float4v*>(out1 + ((long)f * k1 + s1) * d + v * 4) = x;
        if (s2 >= 0) *reinterpret_cast<float4v*>(out2 + ((long)f * k2 + s2) * d + v * 4) = x;
    }
}


// One workgroup per frame.  diffusion_det.py:559-596 (box renewal + DDIM step, eta = 1) with
// :666-672 (x_start from the predicted boxes; NB divided by the frame size for every frame) and
// :649-653 (predict_noise_from_start).  Kept boxes are compacted in index order; the j-th kept box
// consumes noise row j; the tail is replenished with fresh N(0,1) rows.
__global__ __launch_bounds__(256) void ddim_renew_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                          const float* __restrict__ xt, const float* __restrict__ noise,
                                                          const float* __restrict__ fresh, float* __restrict__ out, int m, int c,
                                                          float w, float h, float scale, float sra, float srm1, float sqrt_an,
                                                          float cc, float sigma, float thr) {
    extern __shared__ int pos[];   // [m] compacted slot or -1
    __shared__ int s_remain;
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < m; i += blockDim.x) {
        const float* lp = logits + ((long)f * m + i) * c;
        float mx = -INFINITY;
        for (int j = 0; j < c; ++j) mx = fmaxf(mx, 1.f / (1.f + expf(-lp[j])));
        pos[i] = mx > thr ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int p = 0;
        for (int i = 0; i < m; ++i) {
            const int k = pos[i];
            pos[i] = k ? p : -1;
            p += k;
        }
        s_remain = p;
    }
    __syncthreads();
    const int remain = s_remain;
    const float whwh[4] = {w, h, w, h};
    for (int i = tid; i < m; i += blockDim.x) {
        const int slot = pos[i];
        if (slot >= 0) {
            const float4v b = *reinterpret_cast<const float4v*>(boxes + ((long)f * m + i) * 4);
            const float4v x = *reinterpret_cast<const float4v*>(xt + ((long)f * m + i) * 4);
            const float4v nz = *reinterpret_cast<const float4v*>(noise + ((long)f * m + slot) * 4);
            const float n0 = b[0] / whwh[0], n1 = b[1] / whwh[1], n2 = b[2] / whwh[2], n3 = b[3] / whwh[3];
            float xs[4] = {(n0 + n2) / 2.f, (n1 + n3) / 2.f, n2 - n0, n3 - n1};
            float4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fminf(fmaxf((xs[e] * 2.f - 1.f) * scale, -scale), scale);
                const float pn = (sra * x[e] - v) / srm1;
                o[e] = v * sqrt_an + cc * pn + sigma * nz[e];
            }
            *reinterpret_cast<float4v*>(out + ((long)f * m + slot) * 4) = o;
        }
        if (i >= remain)
            *reinterpret_cast<float4v*>(out + ((long)f * m + i) * 4) =
                *reinterpret_cast<const float4v*>(fresh + ((long)f * m + (i - remain)) * 4);
    }
}

}  // namespace

int dvid_noise_to_boxes_launch(const float* x, float* boxes, int n, float scale, float w, float h, hipStream_t s) {
    if (n == 0) return DVID_OK;
    hipLaunchKernelGGL(noise_to_boxes_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, x, boxes, n, scale, w, h);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_counter_normal_launch(float* out, long per_image, int n_images, uint64_t key0, hipStream_t s) {
    if (per_image <= 0 || n_images <= 0) return DVID_OK;
    const long quads = (per_image + 3) / 4;
    const long nblk = ceil_div(quads, 256L);
    if (nblk > 0x7fffffffL || n_images > 65535) return DVID_ERR_ARG;          // grid limits: x < 2^31, y < 2^16
    hipLaunchKernelGGL(counter_normal_kernel, dim3((unsigned)nblk, (unsigned)n_images), dim3(256), 0, s, out, per_image, n_images, key0);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_apply_deltas_launch(const float* deltas, int delta_ld, const float* boxes, float* out, int n, float wx, float wy, float ww,
                             float wh, float clamp, int* bad_flag, hipStream_t s) {
    if (n == 0) return DVID_OK;
    hipLaunchKernelGGL(apply_deltas_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, deltas, delta_ld, boxes, out, n, wx, wy, ww, wh,
                       clamp, bad_flag);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_topk_mask_launch(const float* logits, int n_img, int m, int c, int k1, int k2, const float* feats, int d, float* out1,
                          float* out2, hipStream_t s) {
    if (n_img == 0) return DVID_OK;
    if (d % 4 || k2 > k1 || k1 > m) return DVID_ERR_ARG;
    const size_t smem = (size_t)m * 12;
    hipLaunchKernelGGL(topk_mask_kernel, dim3(n_img), dim3(256), smem, s, logits, m, c, k1, k2, feats, d, out1, out2);
    LAUNCH_CHECK();
    return DVID_OK;
}

int dvid_ddim_renew_launch(const float* logits, const float* boxes, const float* xt, const float* noise, const float* fresh,
                           float* out, int n_img, int m, int c, float w, float h, float scale, float sra, float srm1, float sqrt_an,
                           float cc, float sigma, float thr, hipStream_t s) {
    if (n_img == 0) return DVID_OK;
    hipLaunchKernelGGL(ddim_renew_kernel, dim3(n_img), dim3(256), (size_t)m * 4, s, logits, boxes, xt, noise, fresh, out, m, c, w, h,
                       scale, sra, srm1, sqrt_an, cc, sigma, thr);
    LAUNCH_CHECK();
    return DVID_OK;
}
