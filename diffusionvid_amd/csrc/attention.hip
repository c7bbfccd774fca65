// Multi-head attention core (softmax(Q K^T / sqrt(dh)) V), head_dim 32, fp32 math.
//
// Used for the 300x300 per-frame self-attention of RCNNHead (box_head.py:514-518) and the
// 2400x900 global-memory cross-attention of DynamicHead (box_head.py:366-380); the in/out
// projections are igemm launches.  Round-1 kernel: VALU flash-style -- one workgroup =
// 64 queries x 4 key partitions, K/V tiles of 32 keys staged in LDS (row pitch 33 floats so the
// four partitions land on distinct banks), online softmax per 8-key chunk, partitions merged
// with wave shuffles.  Softmax statistics are fp32 as apex amp keeps them.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DH = 32;
constexpr int KT = 32;      // keys per LDS tile
constexpr int KP = DH + 1;  // LDS row pitch (floats)

__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out,
                                                        half_t* __restrict__ out16, int lq, int lk, int q_ld, int kv_ld, int out_ld,
                                                        long q_bs, long kv_bs, long out_bs, float scaling) {
    __shared__ float Ks[KT * KP];
    __shared__ float Vs[KT * KP];
    const int tid = threadIdx.x;
    const int part = tid & 3, ql = tid >> 2;
    const int h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 64 + ql;
    const bool q_ok = qi < lq;

    float qr[DH], acc[DH];
    {
        const float* qp = q + b * q_bs + (long)(q_ok ? qi : 0) * q_ld + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4v t = *reinterpret_cast<const float4v*>(qp + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) qr[d + e] = t[e] * scaling;
        }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    float m = -1e30f, l = 0.f;

    const float* kb = k + b * kv_bs + h * DH;
    const float* vb = v + b * kv_bs + h * DH;
    for (int k0 = 0; k0 < lk; k0 += KT) {
        __syncthreads();
        // stage 32 keys x 32 dims of K and V: 2048 floats, 8 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;           // 0..1023
            const int key = idx >> 5, d = idx & 31;
            const bool ok = k0 + key < lk;
            const long off = (long)(ok ? k0 + key : 0) * kv_ld + d;
            Ks[key * KP + d] = ok ? kb[off] : 0.f;
            Vs[key * KP + d] = ok ? vb[off] : 0.f;
        }
        __syncthreads();
        const int kbase = part * 8;
        float sc[8];
        float cmax = -1e30f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* kr = Ks + (kbase + j) * KP;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s += qr[d] * kr[d];
            const bool ok = k0 + kbase + j < lk;
            sc[j] = ok ? s : -1e30f;
            cmax = fmaxf(cmax, sc[j]);
        }
        const float mn = fmaxf(m, cmax);
        const float f = __expf(m - mn);
        float psum = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] *= f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = k0 + kbase + j < lk;
            const float pj = ok ? __expf(sc[j] - mn) : 0.f;
            psum += pj;
            const float* vr = Vs + (kbase + j) * KP;
#pragma unroll
            for (int d = 0; d < DH; ++d) acc[d] += pj * vr[d];
        }
        l = l * f + psum;
        m = mn;
    }
    // merge the 4 key partitions (adjacent lanes)
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64);
        const float l2 = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, m2);
        const float f1 = __expf(m - mn), f2 = __expf(m2 - mn);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float a2 = __shfl_xor(acc[d], o, 64);
            acc[d] = acc[d] * f1 + a2 * f2;
        }
        l = l * f1 + l2 * f2;
        m = mn;
    }
    if (!q_ok) return;
    const float inv = 1.f / l;
    // each partition writes 8 of the 32 output dims
    const long o_off = b * out_bs + (long)qi * out_ld + h * DH + part * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // static register indexing: select the partition's slice without dynamic indexing
        float val = 0.f;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) val = (part == pp) ? acc[pp * 8 + e] : val;
        val *= inv;
        if (out) out[o_off + e] = val;
        if (out16) out16[o_off + e] = (half_t)val;
    }
}

}  // namespace

int dvid_mha_core_launch(const float* q, const float* k, const float* v, float* out, int batch, int lq, int lk, int nheads,
                         int head_dim, int q_ld, int kv_ld, int out_ld, long q_bs, long kv_bs, long out_bs, half_t* out16,
                         hipStream_t s) {
    if (head_dim != DH) return DVID_ERR_UNSUPPORTED;
    if ((q_ld & 3) || lk <= 0) return DVID_ERR_ARG;
    if (lq == 0 || batch == 0) return DVID_OK;
    const float scaling = 1.0f / sqrtf((float)DH);
    hipLaunchKernelGGL(mha_core_kernel, dim3(ceil_div(lq, 64), nheads, batch), dim3(256), 0, s, q, k, v, out, out16, lq, lk, q_ld,
                       kv_ld, out_ld, q_bs, kv_bs, out_bs, scaling);
    LAUNCH_CHECK();
    return DVID_OK;
}
