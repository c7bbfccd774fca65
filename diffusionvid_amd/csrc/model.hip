// libdvid_hip runtime: weight ingest/repack, workspace, stage orchestration, C ABI.
// See include/dvid_hip.h for the contract of every exported symbol.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dvid_hip.h"
#include "common.h"
#include "kernels.h"
#include "options.h"

DvidOptions g_opt;          // csrc/options.h: the library-wide option table (defaults = the benchmarked configuration)

namespace {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define FAIL(code, ...)       \
    do {                      \
        set_err(__VA_ARGS__); \
        return (code);        \
    } while (0)
#define TRY(expr)                       \
    do {                                \
        int _rc = (expr);               \
        if (_rc != DVID_OK) {           \
            if (!g_err[0]) set_err("%s failed (%d) at %s:%d", #expr, _rc, __FILE__, __LINE__); \
            return _rc;                 \
        }                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// profiling of igemm launches (bench.py roofline): events recorded on the launch stream
// ---------------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    double flop, bytes;
    int M, N, K, taps, stride, res_mode;
    const char* kind;      // kernel family the launch ran on (dump / bench.py's top_kernels)
    bool family;           // counts towards the implicit-GEMM family's sums (dvid_profile_read); the other records are the heads' and the
                           // backbone's non-GEMM kernels, timed for the per-kernel table only
};
bool g_prof_on = false;
std::mutex g_prof_mu;                 // models on different host threads may launch concurrently
std::vector<ProfRec> g_prof;
std::vector<ProfRec> g_prof_pool;

int prof_take(ProfRec* r) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_pool.empty()) {
        *r = g_prof_pool.back();
        g_prof_pool.pop_back();
        return DVID_OK;
    }
    HIP_TRY(hipEventCreate(&r->a));
    HIP_TRY(hipEventCreate(&r->b));
    return DVID_OK;
}
void prof_push(const ProfRec& r) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.push_back(r);
}

int igemm(const IgemmParams& p, hipStream_t s) {
    if (!g_prof_on) return dvid_igemm_launch(p, s);
    ProfRec r;
    if (prof_take(&r) != DVID_OK) return DVID_ERR_HIP;
    r.flop = 2.0 * p.M * (double)p.Cout * (double)p.alg_k;
    // algorithmic HBM bytes: every operand touched once (input pixels, packed weights, output, residual)
    const double in_px = (double)p.M * (p.ntaps > 1 ? p.stride * p.stride : 1);
    r.bytes = in_px * p.Cin * 2.0 + (double)p.Cout * p.Kpad * 2.0 +
              (double)p.M * p.Cout * (p.out_f32 ? 4.0 : 2.0) * (p.splitk > 1 ? p.splitk : 1) +
              (p.res_mode == 1 ? (double)p.M * p.Cout * (p.res_f32 ? 4.0 : 2.0) : p.res_mode == 2 ? (double)p.M * p.Cout * 0.5 : 0.0);
    r.M = p.M;
    r.N = p.Cout;
    r.K = p.Kpad;
    r.taps = p.ntaps;
    r.stride = p.stride;
    r.res_mode = p.res_mode;
    r.family = true;
    // the kernel the shape rules of dvid_igemm_launch pick (runs with a forced tile configuration or with the wstat / conv3x3 options off are labelled by the rule)
    r.kind = dvid_wstat_preferred(p) ? (p.res_mode == 1 ? "wstat2" : "wstat") : dvid_conv3x3_halo_preferred(p) ? (p.Cin == 16 ? "conv4x4_s2d" : p.Cout == 64 ? "conv3x3_c64" : "conv3x3_halo") : "igemm2";
    HIP_TRY(hipEventRecord(r.a, s));
    const int rc = dvid_igemm_launch(p, s);
    HIP_TRY(hipEventRecord(r.b, s));
    prof_push(r);
    return rc;
}

// The fused tail of a res2 bottleneck block (bneck.hip) as one record of the implicit-GEMM family: its algorithmic work is the sum
// of the products it computes (conv2 + conv3 [+ shortcut] [+ next conv1]), its bytes what the launch touches once.
int bneck_tail(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res, const half_t* ws,
               const float* bs, const half_t* w1n, const float* b1n, int n_next, half_t* out, half_t* t1n, int n, int H, int W, hipStream_t s) {
    if (!g_prof_on) return dvid_bneck64_tail_launch(t1, w2, b2, w3, b3, res, ws, bs, w1n, b1n, n_next, out, t1n, n, H, W, s);
    ProfRec r;
    if (prof_take(&r) != DVID_OK) return DVID_ERR_HIP;
    const double M = (double)n * H * W;
    const int nn = w1n ? n_next : 0;
    const int kk = 576 + 256 + (ws ? 256 : 0) + 4 * nn;               // MACs per pixel / 64
    r.flop = 2.0 * M * 64.0 * kk;
    r.bytes = M * 2.0 * (64 + (ws ? 64 : 256) + 256 + nn) + 2.0 * 64 * kk;
    r.M = (int)M;
    r.N = 256;
    r.K = kk;
    r.taps = 9;
    r.family = true;
    r.kind = "bneck64_tail";
    r.stride = 1;
    r.res_mode = ws ? 4 : 3;                  // CSV marker: 3 = fused block tail, 4 = with the shortcut convolution
    HIP_TRY(hipEventRecord(r.a, s));
    const int rc = dvid_bneck64_tail_launch(t1, w2, b2, w3, b3, res, ws, bs, w1n, b1n, n_next, out, t1n, n, H, W, s);
    HIP_TRY(hipEventRecord(r.b, s));
    prof_push(r);
    return rc;
}

int bneck128_tail(const half_t* t1, const half_t* w2, const float* b2, const half_t* w3, const float* b3, const half_t* res, const half_t* w1n,
                  const float* b1n, half_t* out, half_t* t1n, int n, int H, int W, hipStream_t s) {
    if (!g_prof_on) return dvid_bneck128_tail_launch(t1, w2, b2, w3, b3, res, w1n, b1n, out, t1n, n, H, W, s);
    ProfRec r;
    if (prof_take(&r) != DVID_OK) return DVID_ERR_HIP;
    const double M = (double)n * H * W;
    const int kk = (w2 ? 1152 : 0) + 512 + (w1n ? 512 : 0);            // MACs per pixel / 128
    r.flop = 2.0 * M * 128.0 * kk;
    r.bytes = M * 2.0 * (128 + 512 + 512 + (w1n ? 128 : 0)) + 2.0 * 128 * kk;
    r.M = (int)M;
    r.N = 512;
    r.K = kk;
    r.taps = w2 ? 9 : 1;
    r.family = true;
    r.kind = "bneck128_tail";
    r.stride = 1;
    r.res_mode = 3;
    HIP_TRY(hipEventRecord(r.a, s));
    const int rc = dvid_bneck128_tail_launch(t1, w2, b2, w3, b3, res, w1n, b1n, out, t1n, n, H, W, s);
    HIP_TRY(hipEventRecord(r.b, s));
    prof_push(r);
    return rc;
}

// A launch outside the implicit-GEMM family (RoIAlign, DynamicConv, attention, the head tail, max pool) as a record of the per-kernel
// table: `rows` units, algorithmic FLOP and bytes of the whole launch.  Not part of the family's sums unless `family`.
template <typename F>
int prof_other(const char* kind, long rows, int n, int k, double flop, double bytes, hipStream_t s, F&& launch, bool family = false) {
    if (!g_prof_on) return launch();
    ProfRec r;
    if (prof_take(&r) != DVID_OK) return DVID_ERR_HIP;
    r.flop = flop;
    r.bytes = bytes;
    r.M = (int)rows;
    r.N = n;
    r.K = k;
    r.taps = 0;
    r.stride = 0;
    r.res_mode = 0;
    r.kind = kind;
    r.family = family;
    HIP_TRY(hipEventRecord(r.a, s));
    const int rc = launch();
    HIP_TRY(hipEventRecord(r.b, s));
    prof_push(r);
    return rc;
}

// ---------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};

// A captured launch sequence holds raw pointers into the workspace: every model counts the moves of ITS buffers (`gen`, bumped when a
// buffer that already existed is re-allocated -- a first allocation cannot have been captured), the detector compares the counter with the
// value it saw at capture time and drops its graphs when it differs (dvid_workspace_generation).  Per model: another model's growth, or
// this model's first allocations, no longer throw this model's graphs away.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n, std::atomic<unsigned long long>* gen = nullptr) {
        if (n <= bytes) return DVID_OK;
        if (gen && p) gen->fetch_add(1, std::memory_order_relaxed);
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        // DVID_POISON_WORKSPACE=1 (diagnostics): fresh workspace starts as 0xFF bytes (NaN as fp16 / fp32), so a kernel that reads
        // workspace nothing has written shows up in the results instead of depending on what the allocation held before
        static const bool poison = getenv("DVID_POISON_WORKSPACE") && atoi(getenv("DVID_POISON_WORKSPACE")) != 0;
        if (poison) HIP_TRY(hipMemset(p, 0xFF, n));
        return DVID_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct ConvW {   // conv or linear weights in MFMA-operand layout
    half_t* w = nullptr;
    float* bias = nullptr;
    int cin = 0, cout = 0, kh = 1, kw = 1, stride = 1, pad = 0, kpad = 0;
    int cin_real = 0;  // un-padded input channels (algorithmic FLOP count)
    bool same_size = false;    // output spatial size = input size whatever (kh, pad) say (the space-to-depth stem: pad 2 before, 1 after)
    int alg_k = 0;             // algorithmic K for the FLOP count when the packed layout carries structural zeros (0: kh*kw*cin_real)
    // DTYPE float32 (csrc/f32.hip): the same rows un-rounded, [cout][kpad32] fp32 with k = (ky*kw + kx)*cin32 + c, cin32 = cin rounded up to 4
    float* w32 = nullptr;
    int* range_flag = nullptr;          // the model's f32_range_flag (null for stand-alone launches)
    half_t *w16hi = nullptr, *w16lo = nullptr;          // the scaled rows as fp16 (hi, lo) planes for the split-operand kernel (csrc/f32.hip)
    float* wscale32 = nullptr;          // [cout]: 2^-e of the power-of-two scaling that puts each packed row's largest magnitude in [0.5, 1) (exact; undone in the epilogue)
    int cin32 = 0, kpad32 = 0;
};
// fragment-order copies of the head-tail weights (csrc/headtail.hip)
struct HeadFrags {
    half_t *w1 = nullptr, *w2 = nullptr, *wc = nullptr, *wlog = nullptr, *wdel = nullptr;
    half_t* cls[4] = {nullptr, nullptr, nullptr, nullptr};
    half_t* reg[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = false;
};
struct LNW {
    float* g = nullptr;
    float* b = nullptr;
    int d = 0;
};
struct HeadW {
    ConvW in_proj, out_proj, dynamic_layer, out_layer, linear1, linear2, class_logits, bboxes_delta, c_mlp;
    std::vector<ConvW> cls, reg;
    std::vector<LNW> cls_ln, reg_ln;
    LNW norm1, norm2, norm3, dc_norm1, dc_norm2, dc_norm3;
    // host copies for the time conditioning (block_time_mlp.1)
    std::vector<float> bt_w, bt_b;
    int bt_out = 0;
    bool cond = false;
    HeadFrags frag;
};
struct Block {
    ConvW c1, c2, c3, sc;
    bool has_sc = false;
};
struct SwinBlockW {
    LNW norm1, norm2;
    ConvW qkv, proj, fc1, fc2;
    half_t* qkv_bias16 = nullptr;   // q/k/v of a padded window position = the qkv bias
    float* relbias = nullptr;       // [heads][49][SWIN_RELBIAS_PITCH] relative-position bias, gathered from the table at load
};
struct SwinStageW {
    std::vector<SwinBlockW> blocks;
    int dim = 0, heads = 0;
    bool has_down = false, has_out = false;
    LNW down_norm, out_norm;
    ConvW down_red;
};

half_t f2h(float f) { return (half_t)f; }

}  // namespace

struct dvid_model {
    dvid_config cfg;
    std::map<std::string, HostTensor> raw;
    bool finalized = false;
    std::atomic<unsigned long long> ws_gen{0};          // moves of this model's workspace buffers (DevBuf::ensure)
    int* f32_range_flag = nullptr;                      // DTYPE float32, split operands: set by a kernel that met |activation| > 65504 (dvid_model_take_range_flag)
    int precision = 0;         // 0: fp16 storage / fp16 MFMA (DTYPE float16), 1: fp32 storage / fp32 MFMA (DTYPE float32); dvid_model_set_precision
    std::vector<void*> owned;  // device allocations of weights

    // backbone
    bool has_backbone = false;
    ConvW stem, stem_s2d;      // NHWC8 7x7/2 form and the 2x2 space-to-depth 4x4/1 form of the same layer
    bool use_s2d = true;       // dvid_set_stem_layout(m, 0): the NHWC8 form
    std::vector<Block> blocks[4];
    ConvW lateral[3], output[3];  // index 0 -> level 3
    // Swin backbone (backbone_type 1)
    ConvW swin_patch;
    LNW swin_patch_norm;
    SwinStageW swin[4];
    // head
    std::vector<HeadW> heads;       // head_series
    std::vector<HeadW> heads_cond;  // head_series_cond
    ConvW gq, gkv, gout;            // global attention projections
    std::vector<float> tm1_w, tm1_b, tm3_w, tm3_b;  // time_mlp host copies
    std::map<int64_t, std::vector<float>> time_cache;  // t -> time_mlp(t) [4*hidden]

    // workspace
    int ws_frames = 0, ws_h = 0, ws_w = 0, ws_boxes = 0;
    DevBuf img8, bufX, bufY, bufT1, bufT2, bufSC, c3, c4, c5, lat[3];
    DevBuf sw_x, sw_x2, sw_ln16, sw_qkv16, sw_attn16, sw_h16;   // Swin token buffers
    DevBuf roi, params, dyn, qkv, attn16, f32a, f32b, f32c, f32d, h16a, h16b, hid16, ss, deltas, kvproj, mem16, splitk, vt;
    // (head slot, t) -> device scale/shift row [bt_out], sub-allocated from slabs of kSsSlabRows rows (a sampler that walks all
    // 1000 time steps on 4 head slots ends at 16 allocations of 512 KB, not 4000 of 2 KB; rows live until the model is destroyed)
    std::map<std::pair<int, int64_t>, float*> ss_rows;
    std::vector<DevBuf> ss_slabs;
    size_t ss_slab_used = 0;          // rows taken from the last slab

    int mem_lk = 0;       // rows of the global memory whose K/V projections sit in kvproj (0: none)

    // sub-batch chains (see dvid_backbone_resnet_fpn)
    int nchain = 2;
    hipStream_t cs[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool streams_ready = false;
    int ensure_streams() {
        if (streams_ready) return DVID_OK;
        for (int c = 0; c < 4; ++c) {
            HIP_TRY(hipStreamCreateWithFlags(&cs[c], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ev_join[c], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        streams_ready = true;
        return DVID_OK;
    }

    int upload(const void* host, size_t bytes, void** dev) {
        HIP_TRY(hipMalloc(dev, bytes));
        owned.push_back(*dev);
        HIP_TRY(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
        return DVID_OK;
    }
    const HostTensor* get(const std::string& name) const {
        auto it = raw.find(name);
        return it == raw.end() ? nullptr : &it->second;
    }
};

namespace {

#define NEED(var, name)                                              \
    const HostTensor* var = m->get(name);                            \
    if (!var) FAIL(DVID_ERR_STATE, "missing tensor '%s'", std::string(name).c_str())

int upload_f32(dvid_model* m, const std::vector<float>& v, float** dev) {
    return m->upload(v.data(), v.size() * sizeof(float), reinterpret_cast<void**>(dev));
}

// weights [cout][cin][kh][kw] (OIHW; Linear: [out][in]) -> fp16 [cout][kpad], k = (ky*kw + kx)*cin_pad + c.
// `scale` (per cout, may be empty) is folded in before rounding to fp16; row_perm maps dst row -> src row.
int make_conv(dvid_model* m, const HostTensor& w, const std::vector<float>& scale, const std::vector<float>& bias, int stride, int pad,
              int cin_pad, const std::vector<int>* row_perm, ConvW* out) {
    const int cout = (int)w.shape[0], cin = (int)w.shape[1];
    const int kh = w.shape.size() == 4 ? (int)w.shape[2] : 1, kw = w.shape.size() == 4 ? (int)w.shape[3] : 1;
    const int cp = cin_pad > 0 ? cin_pad : cin;
    const int kreal = kh * kw * cp;
    const int kpad = (kreal + 63) / 64 * 64;
    std::vector<half_t> packed((size_t)cout * kpad, f2h(0.f));
    for (int o = 0; o < cout; ++o) {
        const int so = row_perm ? (*row_perm)[o] : o;
        const float sc = scale.empty() ? 1.f : scale[so];
        for (int c = 0; c < cin; ++c)
            for (int y = 0; y < kh; ++y)
                for (int x = 0; x < kw; ++x) {
                    const float v = w.v[(((size_t)so * cin + c) * kh + y) * kw + x] * sc;
                    packed[(size_t)o * kpad + (size_t)(y * kw + x) * cp + c] = f2h(v);
                }
    }
    TRY(m->upload(packed.data(), packed.size() * sizeof(half_t), reinterpret_cast<void**>(&out->w)));
    if (m->precision == 1) {          // the un-rounded rows for the fp32 kernels
        const int c4 = (cin + 3) / 4 * 4, k32 = (kh * kw * c4 + 15) / 16 * 16;
        std::vector<float> p32((size_t)cout * k32, 0.f), ws(cout, 1.f);
        for (int o = 0; o < cout; ++o) {
            const int so = row_perm ? (*row_perm)[o] : o;
            const float sc = scale.empty() ? 1.f : scale[so];
            float mx = 0.f;
            for (int c = 0; c < cin; ++c)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x) {
                        const float v = w.v[(((size_t)so * cin + c) * kh + y) * kw + x] * sc;
                        p32[(size_t)o * k32 + (size_t)(y * kw + x) * c4 + c] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
            // the row times 2^e with its largest magnitude in [0.5, 1): the split-operand kernel keeps (hi, lo) fp16 parts of every value, and lo
            // is a full-precision fp16 number only while |v| >= 2^-3; the epilogue multiplies the sums by 2^-e (both exact)
            if (mx > 0.f && std::isfinite(mx)) {
                int e = 0;
                (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
                const float up = ldexpf(1.f, -e);
                for (int k = 0; k < k32; ++k) p32[(size_t)o * k32 + k] *= up;
                ws[o] = ldexpf(1.f, e);
            }
        }
        TRY(upload_f32(m, p32, &out->w32));
        TRY(upload_f32(m, ws, &out->wscale32));
        std::vector<half_t> hi(p32.size()), lo(p32.size());
        for (size_t i = 0; i < p32.size(); ++i) {
            hi[i] = f2h(p32[i]);
            lo[i] = f2h(p32[i] - (float)hi[i]);
        }
        TRY(m->upload(hi.data(), hi.size() * sizeof(half_t), reinterpret_cast<void**>(&out->w16hi)));
        TRY(m->upload(lo.data(), lo.size() * sizeof(half_t), reinterpret_cast<void**>(&out->w16lo)));
        if (!m->f32_range_flag) {
            const int zero = 0;
            TRY(m->upload(&zero, sizeof(int), reinterpret_cast<void**>(&m->f32_range_flag)));
        }
        out->range_flag = m->f32_range_flag;
        out->cin32 = c4;
        out->kpad32 = k32;
    }
    out->bias = nullptr;
    if (!bias.empty()) {
        std::vector<float> b(cout);
        for (int o = 0; o < cout; ++o) b[o] = bias[row_perm ? (*row_perm)[o] : o];
        TRY(upload_f32(m, b, &out->bias));
    }
    out->cin = cp;
    out->cin_real = cin;
    out->cout = cout;
    out->kh = kh;
    out->kw = kw;
    out->stride = stride;
    out->pad = pad;
    out->kpad = kpad;
    return DVID_OK;
}

// conv + FrozenBN (eps 1e-5) folded:  y = conv(x, w * s) + (beta - mean * s),  s = gamma * rsqrt(var + eps)
int make_conv_bn(dvid_model* m, const std::string& name, int stride, int pad, int cin_pad, ConvW* out) {
    NEED(w, name + ".weight");
    NEED(g, name + ".norm.weight");
    NEED(b, name + ".norm.bias");
    NEED(mu, name + ".norm.running_mean");
    NEED(var, name + ".norm.running_var");
    const int cout = (int)w->shape[0];
    std::vector<float> s(cout), bb(cout);
    for (int o = 0; o < cout; ++o) {
        s[o] = g->v[o] / sqrtf(var->v[o] + 1e-5f);
        bb[o] = b->v[o] - mu->v[o] * s[o];
    }
    return make_conv(m, *w, s, bb, stride, pad, cin_pad, nullptr, out);
}

// The 7x7 / stride-2 / pad-3 stem over 3 channels as a 4x4 / stride-1 convolution over the 2x2 space-to-depth image (16 channels:
// (dy*2 + dx)*3 + c, 4 zero): output pixel (oy, ox) reads input rows 2oy-3 .. 2oy+3 = s2d rows oy-2 .. oy+1 (pad 2 before; the row
// after is inside or beyond the image), and original tap ky lives in s2d tap ty = (ky + 1) >> 1 at sub-row dy = (ky + 1) & 1.  FrozenBN
// folded as in make_conv_bn.  K = 16 taps x 16 channels = 256 packed columns against 49 x 8 -> 448 of the NHWC8 form.
int make_stem_s2d(dvid_model* m, const std::string& name, ConvW* out) {
    NEED(w, name + ".weight");
    NEED(g, name + ".norm.weight");
    NEED(b, name + ".norm.bias");
    NEED(mu, name + ".norm.running_mean");
    NEED(var, name + ".norm.running_var");
    const int cout = (int)w->shape[0];
    if (w->shape.size() != 4 || w->shape[1] != 3 || w->shape[2] != 7 || w->shape[3] != 7) FAIL(DVID_ERR_UNSUPPORTED, "stem must be 3 -> C, 7x7");
    const int kpad = 256;
    std::vector<half_t> packed((size_t)cout * kpad, f2h(0.f));
    std::vector<float> bias(cout);
    for (int o = 0; o < cout; ++o) {
        const float sc = g->v[o] / sqrtf(var->v[o] + 1e-5f);
        bias[o] = b->v[o] - mu->v[o] * sc;
        for (int c = 0; c < 3; ++c)
            for (int ky = 0; ky < 7; ++ky)
                for (int kx = 0; kx < 7; ++kx) {
                    const int ty = (ky + 1) >> 1, dy = (ky + 1) & 1, tx = (kx + 1) >> 1, dx = (kx + 1) & 1;
                    const float v = w->v[(((size_t)o * 3 + c) * 7 + ky) * 7 + kx] * sc;
                    packed[(size_t)o * kpad + (size_t)(ty * 4 + tx) * 16 + (dy * 2 + dx) * 3 + c] = f2h(v);
                }
    }
    TRY(m->upload(packed.data(), packed.size() * sizeof(half_t), reinterpret_cast<void**>(&out->w)));
    TRY(upload_f32(m, bias, &out->bias));
    out->cin = 16;
    out->cin_real = 3;
    out->cout = cout;
    out->kh = out->kw = 4;
    out->stride = 1;
    out->pad = 2;
    out->kpad = kpad;
    out->same_size = true;
    out->alg_k = 147;
    return DVID_OK;
}

int make_linear(dvid_model* m, const std::string& name, bool has_bias, ConvW* out, const std::vector<int>* perm = nullptr,
                int row0 = 0, int rows = -1) {
    NEED(w, name + (name.find("in_proj") != std::string::npos ? "_weight" : ".weight"));
    const HostTensor* b = nullptr;
    if (has_bias) {
        const std::string bn = name + (name.find("in_proj") != std::string::npos ? "_bias" : ".bias");
        b = m->get(bn);
        if (!b) FAIL(DVID_ERR_STATE, "missing tensor '%s'", bn.c_str());
    }
    HostTensor sub;
    const HostTensor* src = w;
    std::vector<float> bias;
    if (rows >= 0) {  // row slice (in_proj q / kv parts)
        const int in = (int)w->shape[1];
        sub.shape = {rows, in};
        sub.v.assign(w->v.begin() + (size_t)row0 * in, w->v.begin() + (size_t)(row0 + rows) * in);
        src = &sub;
        if (b) bias.assign(b->v.begin() + row0, b->v.begin() + row0 + rows);
    } else if (b) {
        bias = b->v;
    }
    if (src->shape[1] % 64) FAIL(DVID_ERR_UNSUPPORTED, "linear '%s': in_features %lld not a multiple of 64", name.c_str(),
                                 (long long)src->shape[1]);
    return make_conv(m, *src, {}, bias, 1, 0, 0, perm, out);
}

int make_ln(dvid_model* m, const std::string& name, LNW* out) {
    NEED(g, name + ".weight");
    NEED(b, name + ".bias");
    out->d = (int)g->numel();
    TRY(upload_f32(m, g->v, &out->g));
    TRY(upload_f32(m, b->v, &out->b));
    return DVID_OK;
}

// [cout][kpad] (K contiguous, on the device) -> MFMA fragment order for v_mfma_f32_32x32x16_f16 with the weights as first operand:
// block (n-tile of 32 rows, K step of 16) = 64 lanes x 8 halves, lane l = row (l & 31), k = 8 (l >> 5) .. + 8 -- one contiguous
// 1-KiB wave load per fragment (csrc/headtail.hip).  Rows are zero-padded to a whole number of tiles.
int make_frags(dvid_model* m, const ConvW& w, half_t** out) {
    if (w.kh != 1 || w.kw != 1 || w.kpad % 16) FAIL(DVID_ERR_UNSUPPORTED, "fragment order needs a 1x1 layer with K %% 16 == 0");
    const int ntile = (w.cout + 31) / 32, ks_n = w.kpad / 16;
    std::vector<half_t> src((size_t)w.cout * w.kpad), dst((size_t)ntile * 32 * w.kpad, f2h(0.f));
    HIP_TRY(hipMemcpy(src.data(), w.w, src.size() * sizeof(half_t), hipMemcpyDeviceToHost));
    for (int nt = 0; nt < ntile; ++nt)
        for (int ks = 0; ks < ks_n; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int row = nt * 32 + (l & 31);
                if (row >= w.cout) continue;
                for (int e = 0; e < 8; ++e)
                    dst[(((size_t)nt * ks_n + ks) * 64 + l) * 8 + e] = src[(size_t)row * w.kpad + ks * 16 + (l >> 5) * 8 + e];
            }
    return m->upload(dst.data(), dst.size() * sizeof(half_t), reinterpret_cast<void**>(out));
}

int make_head(dvid_model* m, const std::string& pfx, bool cond, HeadW* h) {
    const dvid_config& c = m->cfg;
    const int d = c.hidden_dim, dd = c.dim_dynamic;
    h->cond = cond;
    TRY(make_linear(m, pfx + ".self_attn.in_proj", true, &h->in_proj));
    TRY(make_linear(m, pfx + ".self_attn.out_proj", true, &h->out_proj));
    // dynamic_layer rows re-ordered so that the generated parameters come out as P1T[j][c], P2T[c][j]
    // (box_head.py:695-696 views them as param1[c][j] at c*dd + j and param2[j][c] at d*dd + j*d + c)
    std::vector<int> perm(2 * d * dd);
    for (int j = 0; j < dd; ++j)
        for (int ch = 0; ch < d; ++ch) perm[j * d + ch] = ch * dd + j;
    for (int ch = 0; ch < d; ++ch)
        for (int j = 0; j < dd; ++j) perm[d * dd + ch * dd + j] = d * dd + j * d + ch;
    TRY(make_linear(m, pfx + ".inst_interact.dynamic_layer", true, &h->dynamic_layer, &perm));
    TRY(make_linear(m, pfx + ".inst_interact.out_layer", true, &h->out_layer));
    TRY(make_ln(m, pfx + ".inst_interact.norm1", &h->dc_norm1));
    TRY(make_ln(m, pfx + ".inst_interact.norm2", &h->dc_norm2));
    TRY(make_ln(m, pfx + ".inst_interact.norm3", &h->dc_norm3));
    TRY(make_linear(m, pfx + ".linear1", true, &h->linear1));
    TRY(make_linear(m, pfx + ".linear2", true, &h->linear2));
    TRY(make_ln(m, pfx + ".norm1", &h->norm1));
    TRY(make_ln(m, pfx + ".norm2", &h->norm2));
    TRY(make_ln(m, pfx + ".norm3", &h->norm3));
    h->cls.resize(c.num_cls);
    h->cls_ln.resize(c.num_cls);
    for (int i = 0; i < c.num_cls; ++i) {
        TRY(make_linear(m, pfx + ".cls_module." + std::to_string(3 * i), false, &h->cls[i]));
        TRY(make_ln(m, pfx + ".cls_module." + std::to_string(3 * i + 1), &h->cls_ln[i]));
    }
    h->reg.resize(c.num_reg);
    h->reg_ln.resize(c.num_reg);
    for (int i = 0; i < c.num_reg; ++i) {
        TRY(make_linear(m, pfx + ".reg_module." + std::to_string(3 * i), false, &h->reg[i]));
        TRY(make_ln(m, pfx + ".reg_module." + std::to_string(3 * i + 1), &h->reg_ln[i]));
    }
    TRY(make_linear(m, pfx + ".class_logits", true, &h->class_logits));
    TRY(make_linear(m, pfx + ".bboxes_delta", true, &h->bboxes_delta));
    NEED(btw, pfx + ".block_time_mlp.1.weight");
    NEED(btb, pfx + ".block_time_mlp.1.bias");
    h->bt_w = btw->v;
    h->bt_b = btb->v;
    h->bt_out = (int)btw->shape[0];
    if (h->bt_out != (cond ? d : 2 * d)) FAIL(DVID_ERR_ARG, "%s.block_time_mlp.1: unexpected out dim %d", pfx.c_str(), h->bt_out);
    if (cond) TRY(make_linear(m, pfx + ".c_mlp.1", true, &h->c_mlp));
    if (dvid_head_tail_supported(d, c.dim_feedforward, c.num_cls, c.num_reg, c.num_classes)) {
        TRY(make_frags(m, h->linear1, &h->frag.w1));
        TRY(make_frags(m, h->linear2, &h->frag.w2));
        if (cond) TRY(make_frags(m, h->c_mlp, &h->frag.wc));
        for (int i = 0; i < c.num_cls; ++i) TRY(make_frags(m, h->cls[i], &h->frag.cls[i]));
        for (int i = 0; i < c.num_reg; ++i) TRY(make_frags(m, h->reg[i], &h->frag.reg[i]));
        TRY(make_frags(m, h->class_logits, &h->frag.wlog));
        TRY(make_frags(m, h->bboxes_delta, &h->frag.wdel));
        h->frag.ok = true;
    }
    return DVID_OK;
}

// every block of the stage is a 64-wide stride-1 bottleneck with 256 outputs; block 0 has a shortcut convolution over 64 channels
// (R-50 / R-101 res2), the others take the block input as the residual
bool bneck64_stage(const std::vector<Block>& blocks) {
    if (blocks.empty()) return false;
    for (size_t b = 0; b < blocks.size(); ++b) {
        const Block& k = blocks[b];
        const int cin = b == 0 ? 64 : 256;
        if (k.c1.kh != 1 || k.c1.stride != 1 || k.c1.cin != cin || k.c1.cout != 64 || k.c1.kpad != cin || !k.c1.bias) return false;
        if (k.c2.kh != 3 || k.c2.kw != 3 || k.c2.stride != 1 || k.c2.pad != 1 || k.c2.cin != 64 || k.c2.cout != 64 || k.c2.kpad != 576 ||
            !k.c2.bias)
            return false;
        if (k.c3.kh != 1 || k.c3.stride != 1 || k.c3.cin != 64 || k.c3.cout != 256 || k.c3.kpad != 64 || !k.c3.bias) return false;
        if (k.has_sc != (b == 0)) return false;
        if (k.has_sc && (k.sc.kh != 1 || k.sc.stride != 1 || k.sc.cin != 64 || k.sc.cout != 256 || k.sc.kpad != 64 || !k.sc.bias)) return false;
    }
    return true;
}

// res3 of R-50 / R-101: 128-wide bottlenecks with 512 outputs; the first block has the stride and a shortcut convolution, the others
// are stride-1 identity blocks
bool bneck128_stage(const std::vector<Block>& blocks) {
    if (blocks.size() < 2) return false;
    for (size_t b = 0; b < blocks.size(); ++b) {
        const Block& k = blocks[b];
        if (k.c3.kh != 1 || k.c3.stride != 1 || k.c3.cin != 128 || k.c3.cout != 512 || k.c3.kpad != 128 || !k.c3.bias) return false;
        if (k.c2.kh != 3 || k.c2.kw != 3 || k.c2.pad != 1 || k.c2.cin != 128 || k.c2.cout != 128 || k.c2.kpad != 1152 || !k.c2.bias) return false;
        if (k.has_sc != (b == 0)) return false;
        if (b == 0) {
            if (k.sc.cout != 512) return false;
            continue;
        }
        if (k.c1.kh != 1 || k.c1.stride != 1 || k.c1.cin != 512 || k.c1.cout != 128 || k.c1.kpad != 512 || !k.c1.bias) return false;
        if (k.c2.stride != 1) return false;
    }
    return true;
}

int conv_run(const ConvW& w, const half_t* in, int n, int h, int wd, void* out, int relu, int out_f32, const void* res,
             int res_mode, int res_f32, hipStream_t s, int* ho_out = nullptr, int* wo_out = nullptr, int ldc = 0, int splitk = 1,
             bool pooled = false) {
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.in = in;
    p.w = w.w;
    p.bias = w.bias;
    p.res = res;
    p.out = out;
    p.H = h;
    p.W = wd;
    p.Cin = w.cin;
    p.KH = w.kh;
    p.KW = w.kw;
    p.stride = w.stride;
    p.pad = w.pad;
    p.Ho = w.same_size ? h : (h + 2 * w.pad - w.kh) / w.stride + 1;
    p.Wo = w.same_size ? wd : (wd + 2 * w.pad - w.kw) / w.stride + 1;
    p.Cout = w.cout;
    p.M = n * p.Ho * p.Wo;
    p.Kpad = w.kpad;
    p.ntaps = w.kh * w.kw;
    p.alg_k = w.alg_k ? w.alg_k : w.kh * w.kw * (w.cin_real ? w.cin_real : w.cin);
    p.ldc = ldc ? ldc : w.cout;
    p.relu = relu;
    p.out_f32 = out_f32;
    p.res_mode = res_mode;
    p.res_f32 = res_f32;
    if (splitk > 1) {               // fp32 partial slabs; bias/activation are applied by the consumer
        p.splitk = splitk;
        p.split_stride = (long)p.M * p.ldc;
        p.bias = nullptr;
    }
    if (ho_out) *ho_out = p.Ho;
    if (wo_out) *wo_out = p.Wo;
    if (pooled) {          // the space-to-depth stem with its max pool in the same launch: `out` is the pooled map; ho / wo stay the stem's
        if (!dvid_stem_pool_supported(p)) return DVID_ERR_UNSUPPORTED;
        // algorithmic bytes: the space-to-depth image in, the pooled map out, the weights
        return prof_other("stem_pool", p.M, 64, p.Kpad, 2.0 * p.M * 64.0 * p.alg_k, (double)p.M * 32.0 + (double)p.M / 4 * 128.0 + 64.0 * p.Kpad * 2.0, s,
                          [&] { return dvid_stem_pool_launch(p, s); }, /*family=*/true);          // an implicit-GEMM launch like the stem it replaces
    }
    return igemm(p, s);
}

// Linear on [rows, in] fp16
int linear_run(const ConvW& w, const half_t* in, int rows, void* out, int relu, int out_f32, hipStream_t s) {
    return conv_run(w, in, rows, 1, 1, out, relu, out_f32, nullptr, 0, 0, s);
}

// ---- DTYPE float32: the same layers on csrc/f32.hip (fp32 NHWC activations, un-rounded weights) ---------------------------------
int conv_run32(const ConvW& w, const float* in, int n, int h, int wd, float* out, int relu, const float* res, int res_mode, hipStream_t s,
               int* ho_out = nullptr, int* wo_out = nullptr, int ldc = 0) {
    if (!w.w32) return DVID_ERR_STATE;
    F32GemmParams p;
    memset(&p, 0, sizeof(p));
    p.in = in;
    p.w = w.w32;
    p.w_hi = w.w16hi;
    p.w_lo = w.w16lo;
    p.range_flag = w.range_flag;
    p.bias = w.bias;
    p.wscale = w.wscale32;
    p.res = res;
    p.out = out;
    p.H = h;
    p.W = wd;
    p.Cin = w.cin32;
    p.KH = w.kh;
    p.KW = w.kw;
    p.stride = w.stride;
    p.pad = w.pad;
    p.Ho = (h + 2 * w.pad - w.kh) / w.stride + 1;
    p.Wo = (wd + 2 * w.pad - w.kw) / w.stride + 1;
    p.Cout = w.cout;
    p.M = n * p.Ho * p.Wo;
    p.K = w.kh * w.kw * w.cin32;
    p.Kpad = w.kpad32;
    p.ldc = ldc ? ldc : w.cout;
    p.relu = relu;
    p.res_mode = res_mode;
    if (ho_out) *ho_out = p.Ho;
    if (wo_out) *wo_out = p.Wo;
    const double alg_k = (double)w.kh * w.kw * (w.cin_real ? w.cin_real : w.cin32);
    const double in_px = (double)p.M * (w.kh * w.kw > 1 ? w.stride * w.stride : 1);
    const double bytes = 4.0 * (in_px * p.Cin + (double)p.Cout * p.Kpad + (double)p.M * p.Cout * (res_mode == 1 ? 2.0 : res_mode == 2 ? 1.25 : 1.0));
    return prof_other("igemm_f32", p.M, p.Cout, p.Kpad, 2.0 * p.M * (double)p.Cout * alg_k, bytes, s, [&] { return dvid_f32_igemm_launch(p, s); },
                      /*family=*/true);
}
int linear_run32(const ConvW& w, const float* in, int rows, float* out, int relu, hipStream_t s, int ldc = 0) {
    return conv_run32(w, in, rows, 1, 1, out, relu, nullptr, 0, s, nullptr, nullptr, ldc);
}

// detectron2 FPN.forward over three levels (strides 8/16/32): lateral 1x1 (+ nearest-x2 top-down sum fused in the
// epilogue), 3x3 output conv.  Inputs: m->c3/c4/c5 fp16 NHWC; sh/sw = their heights/widths.
int run_fpn(dvid_model* m, int n, const int* sh, const int* sw, void* p3, void* p4, void* p5, hipStream_t s) {
    void* pout[3] = {p3, p4, p5};
    const half_t* cin[3] = {m->c3.as<half_t>(), m->c4.as<half_t>(), m->c5.as<half_t>()};
    for (int l = 2; l >= 0; --l) {
        const void* res = (l < 2) ? m->lat[l + 1].p : nullptr;
        TRY(conv_run(m->lateral[l], cin[l], n, sh[l], sw[l], m->lat[l].p, 0, 0, res, res ? 2 : 0, 0, s));
        TRY(conv_run(m->output[l], m->lat[l].as<half_t>(), n, sh[l], sw[l], pout[l], 0, 0, nullptr, 0, 0, s));
    }
    return DVID_OK;
}

// One RCNNHead / RCNNHead_cond pass over frames [f0, f0 + nf) on stream `s`; `wrow` = first workspace row of this
// chain's slice of the [rows, *] buffers, `vt_off` = its offset (halves) in the V^T scratch.
int rcnn_head_chain(dvid_model* m, const HeadW& hw, int is_cond, const void* p3, const void* p4, const void* p5, int f0, int nf,
                    int height, int width, int M, const float* boxes_all, const float* pro_all, const float* cond_all,
                    float* logits_all, float* boxes_out_all, float* obj_all, int* bad_box_flag, const float* ss_all, int ss_stride,
                    size_t wrow, size_t vt_off, hipStream_t s) {
    const int d = m->cfg.hidden_dim, R = nf * M;
    const size_t r0 = (size_t)f0 * M;
    const float* boxes = boxes_all + r0 * 4;
    const float* pro_features = pro_all ? pro_all + r0 * d : nullptr;
    const float* cond = cond_all ? cond_all + r0 * d : nullptr;
    float* logits = logits_all + r0 * m->cfg.num_classes;
    float* boxes_out = boxes_out_all + r0 * 4;
    float* obj_features = obj_all + r0 * d;
    const float* ss_dev = ss_all + (size_t)f0 * ss_stride;          // ss_stride 0: every frame reads the one (head, t) row
    // workspace slices
    half_t* roi16 = m->roi.as<half_t>() + wrow * 49 * d;
    half_t* dyn16 = m->dyn.as<half_t>() + wrow * 49 * d;
    half_t* params16 = m->params.as<half_t>() + wrow * 2 * d * m->cfg.dim_dynamic;
    half_t* qkv16 = m->qkv.as<half_t>() + wrow * 3 * d * 2;        // buffer is sized in fp32 units; fp16 use needs half of it
    half_t* attn16 = m->attn16.as<half_t>() + wrow * d;
    float* f32a = m->f32a.as<float>() + wrow * d;
    float* f32b = m->f32b.as<float>() + wrow * d;
    float* f32c = m->f32c.as<float>() + wrow * d;
    float* f32d = m->f32d.as<float>() + wrow * d;
    half_t* h16a = m->h16a.as<half_t>() + wrow * d;
    half_t* h16b = m->h16b.as<half_t>() + wrow * d;
    half_t* hid16 = m->hid16.as<half_t>() + wrow * m->cfg.dim_feedforward;
    float* deltas = m->deltas.as<float>() + wrow * 4;
    float* splitk = m->splitk.as<float>() + wrow * d * 8;
    half_t* vt = m->vt.as<half_t>() + vt_off;

    // --- RoIAlign ---
    RoiLevels lv;
    const void* pl[3] = {p3, p4, p5};
    for (int l = 0; l < 3; ++l) {
        lv.h[l] = height >> (3 + l);
        lv.w[l] = width >> (3 + l);
        lv.feat[l] = reinterpret_cast<const half_t*>(pl[l]) + (size_t)f0 * lv.h[l] * lv.w[l] * d;
        lv.scale[l] = 1.f / (float)(8 << l);
    }
    float* pro32 = f32a;
    // A pass that gets its proposal features from the caller needs nothing of the tile before DynamicConv: the gather then runs INSIDE the
    // DynamicConv launch (csrc/dynconv.hip, FUSED_ROI) and the fp16 tile never reaches memory.  A pass without them takes the tile's mean
    // over the bins as its features (box_head.py:509-510) ahead of the self-attention: the two launches.
    const bool roi_fused = g_opt.roi_fuse && pro_features != nullptr && d == 256;
    double map_px = 0;
    for (int l = 0; l < 3; ++l) map_px += (double)lv.h[l] * lv.w[l];
    if (!roi_fused) {
        // algorithmic bytes: the three maps of the launch's frames once + one 49 x d tile per box (the 784 taps per box go through L1)
        TRY(prof_other("roialign", R, d, 49, 0.0, (double)nf * map_px * d * 2.0 + (double)R * 49 * d * 2.0, s,
                       [&] { return dvid_roialign_launch(lv, d, boxes, nf, M, roi16, pro_features ? nullptr : pro32, s); }));
    }
    const float* pro = pro_features ? pro_features : pro32;
    // --- self attention + norm1 ---
    TRY(dvid_f32_to_f16_launch(pro, h16a, (long)R * d, s));
    TRY(linear_run(hw.in_proj, h16a, R, qkv16, 0, 0, s));          // fp16 q|k|v, MFMA operands
    TRY(prof_other("mha_mfma", R, d, M, 4.0 * R * (double)M * d, (double)R * d * 2.0 * 4.0, s, [&] {
        return dvid_mha_mfma_launch(qkv16, qkv16 + d, qkv16 + 2 * d, attn16, vt, nf, M, M, m->cfg.nheads, 3 * d, 3 * d, d, (long)M * 3 * d,
                                    (long)M * 3 * d, (long)M * d, s);
    }));
    TRY(linear_run(hw.out_proj, attn16, R, f32b, 0, 1, s));
    float* x1 = f32c;
    TRY(dvid_add_layernorm_launch(pro, f32b, hw.norm1.g, hw.norm1.b, x1, h16a, R, d, 0, s));
    // --- DynamicConv ---
    // dynamic_layer writes 64 KB of parameters per box that DynamicConv reads straight back (csrc/dynconv.hip)
    TRY(linear_run(hw.dynamic_layer, h16a, R, params16, 0, 0, s));
    {
        const int dd = m->cfg.dim_dynamic;
        if (roi_fused) {
            // algorithmic bytes: the maps once + the parameters + the output tile per box
            TRY(prof_other("dynconv_roi", R, d, dd, 2.0 * R * 49.0 * d * dd * 2.0, (double)nf * map_px * d * 2.0 + (double)R * (49 * d * 2.0 + 2.0 * d * dd * 2.0), s, [&] {
                return dvid_dynconv_roi_launch(lv, d, boxes, nf, M, params16, hw.dc_norm1.g, hw.dc_norm1.b, hw.dc_norm2.g, hw.dc_norm2.b, dyn16, s);
            }));
        } else {
            TRY(prof_other("dynconv", R, d, dd, 2.0 * R * 49.0 * d * dd * 2.0, (double)R * (2.0 * 49 * d * 2.0 + 2.0 * d * dd * 2.0), s,
                           [&] { return dvid_dynconv_launch(roi16, params16, hw.dc_norm1.g, hw.dc_norm1.b, hw.dc_norm2.g, hw.dc_norm2.b, dyn16, R, s); }));
        }
    }
    // out_layer: K = 49*d = 12544 on only R x d outputs -> split K over 7 workgroups per tile; the partial slabs
    // and the bias are summed inside the norm3 kernel that consumes them.
    const int osplit = ((hw.out_layer.kpad / 64) % 7 == 0) ? 7 : 1;
    if (osplit > 1) {
        TRY(conv_run(hw.out_layer, dyn16, R, 1, 1, splitk, 0, 1, nullptr, 0, 0, s, nullptr, nullptr, 0, osplit));
        TRY(dvid_add_layernorm_launch(splitk, nullptr, hw.dc_norm3.g, hw.dc_norm3.b, f32b, nullptr, R, d, 1, s, osplit, (long)R * d,
                                      hw.out_layer.bias));
    } else {
        TRY(linear_run(hw.out_layer, dyn16, R, f32b, 0, 1, s));
        TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.dc_norm3.g, hw.dc_norm3.b, f32b, nullptr, R, d, 1, s));
    }
    float* obj = f32d;
    TRY(dvid_add_layernorm_launch(x1, f32b, hw.norm2.g, hw.norm2.b, obj, h16a, R, d, 0, s));
    // --- FFN + norm3 + modulation + towers + class_logits + bboxes_delta + apply_deltas: one row-tile kernel (csrc/headtail.hip);
    // option head_tail = 0 (the fused-vs-layerwise parity test) or an unsupported shape takes the layer-by-layer launches below
    if (g_opt.head_tail && hw.frag.ok) {
        HeadTailParams q;
        memset(&q, 0, sizeof(q));
        q.x16 = h16a;
        q.obj32 = obj;
        q.w1f = hw.frag.w1;
        q.b1 = hw.linear1.bias;
        q.w2f = hw.frag.w2;
        q.b2 = hw.linear2.bias;
        q.n3g = hw.norm3.g;
        q.n3b = hw.norm3.b;
        q.scale = ss_dev;
        q.ss_stride = ss_stride;
        q.rows_per_frame = M;
        q.cond32 = is_cond ? cond : nullptr;
        q.wcf = hw.frag.wc;
        q.bc = hw.c_mlp.bias;
        q.num_cls = (int)hw.cls.size();
        q.num_reg = (int)hw.reg.size();
        q.num_classes = m->cfg.num_classes;
        q.dff = m->cfg.dim_feedforward;
        for (size_t i = 0; i < hw.cls.size(); ++i) {
            q.clsf[i] = hw.frag.cls[i];
            q.clsg[i] = hw.cls_ln[i].g;
            q.clsb[i] = hw.cls_ln[i].b;
        }
        for (size_t i = 0; i < hw.reg.size(); ++i) {
            q.regf[i] = hw.frag.reg[i];
            q.regg[i] = hw.reg_ln[i].g;
            q.regb[i] = hw.reg_ln[i].b;
        }
        q.wlogf = hw.frag.wlog;
        q.blog = hw.class_logits.bias;
        q.wdelf = hw.frag.wdel;
        q.bdel = hw.bboxes_delta.bias;
        q.boxes = boxes;
        q.obj_out = obj_features;
        q.logits = logits;
        q.boxes_out = boxes_out;
        q.bad_flag = bad_box_flag;
        q.R = R;
        q.wx = 2.f;
        q.wy = 2.f;
        q.ww = 1.f;
        q.wh = 1.f;
        q.clamp = logf(100000.f / 16.f);
        {
            const double dff = m->cfg.dim_feedforward, nt = (double)hw.cls.size() + (double)hw.reg.size() + (is_cond ? 1.0 : 0.0);
            return prof_other("head_tail", R, d, (int)dff, 2.0 * R * d * (2.0 * dff + nt * d + 64.0),
                              (double)R * (d * 6.0 + d * 4.0 + m->cfg.num_classes * 4.0 + 32.0), s, [&] { return dvid_head_tail_launch(q, s); });
        }
    }
    // --- FFN + norm3 ---
    TRY(linear_run(hw.linear1, h16a, R, hid16, 1, 0, s));
    TRY(linear_run(hw.linear2, hid16, R, f32b, 0, 1, s));
    TRY(dvid_add_layernorm_launch(obj, f32b, hw.norm3.g, hw.norm3.b, obj_features, nullptr, R, d, 0, s));
    // --- time / cond modulation ---
    half_t* fc16 = h16a;
    if (!is_cond) {
        TRY(dvid_modulate_launch(obj_features, ss_dev, ss_stride, ss_dev + d, 0, ss_stride, fc16, R, M, d, s));
    } else {
        TRY(dvid_silu_f16_launch(cond, h16b, (long)R * d, s));
        TRY(linear_run(hw.c_mlp, h16b, R, f32b, 0, 1, s));
        TRY(dvid_modulate_launch(obj_features, ss_dev, ss_stride, f32b, 1, d, fc16, R, M, d, s));
    }
    // --- cls tower ---
    const half_t* cur = fc16;
    for (size_t i = 0; i < hw.cls.size(); ++i) {
        TRY(linear_run(hw.cls[i], cur, R, f32b, 0, 1, s));
        TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.cls_ln[i].g, hw.cls_ln[i].b, nullptr, h16b, R, d, 1, s));
        cur = h16b;
    }
    TRY(conv_run(hw.class_logits, cur, R, 1, 1, logits, 0, 1, nullptr, 0, 0, s, nullptr, nullptr, m->cfg.num_classes));
    // --- reg tower ---
    cur = fc16;
    half_t* regbuf[2] = {h16b, attn16};
    for (size_t i = 0; i < hw.reg.size(); ++i) {
        TRY(linear_run(hw.reg[i], cur, R, f32b, 0, 1, s));
        TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.reg_ln[i].g, hw.reg_ln[i].b, nullptr, regbuf[i & 1], R, d, 1, s));
        cur = regbuf[i & 1];
    }
    TRY(conv_run(hw.bboxes_delta, cur, R, 1, 1, deltas, 0, 1, nullptr, 0, 0, s, nullptr, nullptr, 4));
    TRY(dvid_apply_deltas_launch(deltas, 4, boxes, boxes_out, R, 2.f, 2.f, 1.f, 1.f, logf(100000.f / 16.f), bad_box_flag, s));
    return DVID_OK;
}

// The same pass with DTYPE float32 (csrc/f32.hip): fp32 RoI tiles, q / k / v, dynamic parameters, hidden layers; layer by layer.
int rcnn_head_chain_f32(dvid_model* m, const HeadW& hw, int is_cond, const void* p3, const void* p4, const void* p5, int f0, int nf,
                        int height, int width, int M, const float* boxes_all, const float* pro_all, const float* cond_all,
                        float* logits_all, float* boxes_out_all, float* obj_all, int* bad_box_flag, const float* ss_all, int ss_stride,
                        size_t wrow, hipStream_t s) {
    const int d = m->cfg.hidden_dim, R = nf * M, dd = m->cfg.dim_dynamic, dff = m->cfg.dim_feedforward;
    const size_t r0 = (size_t)f0 * M;
    const float* boxes = boxes_all + r0 * 4;
    const float* pro_features = pro_all ? pro_all + r0 * d : nullptr;
    const float* cond = cond_all ? cond_all + r0 * d : nullptr;
    float* logits = logits_all + r0 * m->cfg.num_classes;
    float* boxes_out = boxes_out_all + r0 * 4;
    float* obj_features = obj_all + r0 * d;
    const float* ss_dev = ss_all + (size_t)f0 * ss_stride;
    float* roi = m->roi.as<float>() + wrow * 49 * d;
    float* dyn = m->dyn.as<float>() + wrow * 49 * d;
    float* params = m->params.as<float>() + wrow * 2 * d * dd;
    float* qkv = m->qkv.as<float>() + wrow * 3 * d;
    float* attn = m->attn16.as<float>() + wrow * d;
    float* f32a = m->f32a.as<float>() + wrow * d;
    float* f32b = m->f32b.as<float>() + wrow * d;
    float* f32c = m->f32c.as<float>() + wrow * d;
    float* f32d = m->f32d.as<float>() + wrow * d;
    float* ha = m->h16a.as<float>() + wrow * d;
    float* hb = m->h16b.as<float>() + wrow * d;
    float* hid = m->hid16.as<float>() + wrow * dff;
    float* deltas = m->deltas.as<float>() + wrow * 4;

    RoiLevels32 lv;
    const void* pl[3] = {p3, p4, p5};
    double map_px = 0;
    for (int l = 0; l < 3; ++l) {
        lv.h[l] = height >> (3 + l);
        lv.w[l] = width >> (3 + l);
        lv.feat[l] = reinterpret_cast<const float*>(pl[l]) + (size_t)f0 * lv.h[l] * lv.w[l] * d;
        lv.scale[l] = 1.f / (float)(8 << l);
        map_px += (double)lv.h[l] * lv.w[l];
    }
    float* pro32 = f32a;
    TRY(prof_other("roialign_f32", R, d, 49, 0.0, (double)nf * map_px * d * 4.0 + (double)R * 49 * d * 4.0, s,
                   [&] { return dvid_f32_roialign_launch(lv, d, boxes, nf, M, roi, pro_features ? nullptr : pro32, s); }));
    const float* pro = pro_features ? pro_features : pro32;
    // --- self attention + norm1 (box_head.py:512-517)
    TRY(linear_run32(hw.in_proj, pro, R, qkv, 0, s));
    TRY(prof_other("mha_f32", R, d, M, 4.0 * R * (double)M * d, (double)R * d * 4.0 * 4.0, s, [&] {
        return dvid_f32_mha_launch(qkv, qkv + d, qkv + 2 * d, attn, nf, M, M, m->cfg.nheads, 3 * d, 3 * d, d, (long)M * 3 * d, (long)M * 3 * d,
                                   (long)M * d, s);
    }));
    TRY(linear_run32(hw.out_proj, attn, R, f32b, 0, s));
    float* x1 = f32c;
    TRY(dvid_add_layernorm_launch(pro, f32b, hw.norm1.g, hw.norm1.b, x1, nullptr, R, d, 0, s));
    // --- DynamicConv (box_head.py:687-711)
    TRY(linear_run32(hw.dynamic_layer, x1, R, params, 0, s));
    TRY(prof_other("dynconv_f32", R, d, dd, 2.0 * R * 49.0 * d * dd * 2.0, (double)R * (2.0 * 49 * d * 4.0 + 2.0 * d * dd * 4.0), s,
                   [&] { return dvid_f32_dynconv_launch(roi, params, hw.dc_norm1.g, hw.dc_norm1.b, hw.dc_norm2.g, hw.dc_norm2.b, dyn, R, m->f32_range_flag, s); }));
    TRY(linear_run32(hw.out_layer, dyn, R, f32b, 0, s));
    TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.dc_norm3.g, hw.dc_norm3.b, f32b, nullptr, R, d, 1, s));
    float* obj = f32d;
    TRY(dvid_add_layernorm_launch(x1, f32b, hw.norm2.g, hw.norm2.b, obj, nullptr, R, d, 0, s));
    // --- FFN + norm3
    TRY(linear_run32(hw.linear1, obj, R, hid, 1, s));
    TRY(linear_run32(hw.linear2, hid, R, f32b, 0, s));
    TRY(dvid_add_layernorm_launch(obj, f32b, hw.norm3.g, hw.norm3.b, obj_features, nullptr, R, d, 0, s));
    // --- time / cond modulation
    float* fc = ha;
    if (!is_cond) {
        TRY(dvid_f32_modulate_launch(obj_features, ss_dev, ss_stride, ss_dev + d, 0, ss_stride, fc, R, M, d, s));
    } else {
        TRY(dvid_f32_silu_launch(cond, hb, (long)R * d, s));
        TRY(linear_run32(hw.c_mlp, hb, R, f32b, 0, s));
        TRY(dvid_f32_modulate_launch(obj_features, ss_dev, ss_stride, f32b, 1, d, fc, R, M, d, s));
    }
    // --- cls tower
    const float* cur = fc;
    for (size_t i = 0; i < hw.cls.size(); ++i) {
        TRY(linear_run32(hw.cls[i], cur, R, f32b, 0, s));
        TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.cls_ln[i].g, hw.cls_ln[i].b, hb, nullptr, R, d, 1, s));
        cur = hb;
    }
    TRY(linear_run32(hw.class_logits, cur, R, logits, 0, s, m->cfg.num_classes));
    // --- reg tower
    cur = fc;
    float* regbuf[2] = {hb, attn};
    for (size_t i = 0; i < hw.reg.size(); ++i) {
        TRY(linear_run32(hw.reg[i], cur, R, f32b, 0, s));
        TRY(dvid_add_layernorm_launch(f32b, nullptr, hw.reg_ln[i].g, hw.reg_ln[i].b, regbuf[i & 1], nullptr, R, d, 1, s));
        cur = regbuf[i & 1];
    }
    TRY(linear_run32(hw.bboxes_delta, cur, R, deltas, 0, s, 4));
    TRY(dvid_apply_deltas_launch(deltas, 4, boxes, boxes_out, R, 2.f, 2.f, 1.f, 1.f, logf(100000.f / 16.f), bad_box_flag, s));
    return DVID_OK;
}

// detectron2 build_resnet_fpn_backbone with DTYPE float32: normaliser -> NHWC4, BasicStem (7x7 / 2 + FrozenBN folded + ReLU + max pool),
// the bottleneck stages layer by layer, FPN; every tensor fp32 (csrc/f32.hip)
int backbone_resnet_f32(dvid_model* m, const float* const* frames, int n, int height, int width, float* p3, float* p4, float* p5, hipStream_t s) {
    float mean[3], stdv[3];
    for (int i = 0; i < 3; ++i) {
        mean[i] = m->cfg.pixel_mean[i] / 255.f;
        stdv[i] = m->cfg.pixel_std[i] / 255.f;
    }
    // sub-batch chains on separate streams, as the fp16 backbone runs them (frames are independent; every chain works in its own slice of
    // the workspace): the HBM-paced short-K layers of one chain run beside the operand-stream-paced 3x3 layers of the other
    const int nchain = (m->nchain > 1 && n >= 16 * m->nchain) ? m->nchain : 1;
    if (nchain > 1) {
        TRY(m->ensure_streams());
        HIP_TRY(hipEventRecord(m->ev_fork, s));
        for (int c = 0; c < nchain; ++c) HIP_TRY(hipStreamWaitEvent(m->cs[c], m->ev_fork, 0));
    }
    const int per = (n + nchain - 1) / nchain;
    const size_t px = (size_t)height * width, px4 = px / 16;
    for (int c = 0; c < nchain; ++c) {
        const int f0 = c * per, nf = (f0 + per <= n) ? per : n - f0;
        if (nf <= 0) continue;
        hipStream_t cs = nchain > 1 ? m->cs[c] : s;
        const size_t fo = (size_t)f0, big = fo * px4 * 256;
        float* img = m->img8.as<float>() + fo * px * 4;
        float* bx = m->bufX.as<float>() + big;
        float* by = m->bufY.as<float>() + big;
        float* t1 = m->bufT1.as<float>() + big;
        float* t2 = m->bufT2.as<float>() + big;
        float* sc = m->bufSC.as<float>() + big;
        float* stage_out[4] = {nullptr, m->c3.as<float>() + fo * (px4 / 4) * 512, m->c4.as<float>() + fo * (px4 / 16) * 1024,
                               m->c5.as<float>() + fo * (px4 / 64) * 2048};
        float* lat[3];
        for (int l = 0; l < 3; ++l) lat[l] = m->lat[l].as<float>() + fo * (px4 / (4 << (2 * l))) * 256;
        TRY(dvid_f32_prep_images_launch(frames + f0, img, nf, height, width, mean, stdv, cs));
        int h = height, w = width;
        TRY(conv_run32(m->stem, img, nf, h, w, t1, 1, nullptr, 0, cs, &h, &w));
        TRY(prof_other("maxpool_f32", (long)nf * h * w, 64, 9, 0.0, (double)nf * h * w * 64 * 4.0 * 1.25, cs,
                       [&] { return dvid_f32_maxpool3x3s2_launch(t1, bx, nf, h, w, 64, cs); }));
        h = (h + 2 - 3) / 2 + 1;
        w = (w + 2 - 3) / 2 + 1;
        float* cur = bx;
        int sh[4], sw[4];
        for (int st = 0; st < 4; ++st) {
            const int nb = (int)m->blocks[st].size();
            for (int b = 0; b < nb; ++b) {
                const Block& blk = m->blocks[st][b];
                int h2 = h, w2 = w;
                TRY(conv_run32(blk.c1, cur, nf, h, w, t1, 1, nullptr, 0, cs));
                TRY(conv_run32(blk.c2, t1, nf, h, w, t2, 1, nullptr, 0, cs, &h2, &w2));
                const float* res = cur;
                if (blk.has_sc) {
                    TRY(conv_run32(blk.sc, cur, nf, h, w, sc, 0, nullptr, 0, cs));
                    res = sc;
                }
                float* dst = (b == nb - 1 && stage_out[st]) ? stage_out[st] : (cur == bx ? by : bx);
                TRY(conv_run32(blk.c3, t2, nf, h2, w2, dst, 1, res, 1, cs));
                h = h2;
                w = w2;
                cur = dst;
            }
            sh[st] = h;
            sw[st] = w;
        }
        float* pout[3] = {p3 + (size_t)f0 * sh[1] * sw[1] * 256, p4 + (size_t)f0 * sh[2] * sw[2] * 256, p5 + (size_t)f0 * sh[3] * sw[3] * 256};
        for (int l = 2; l >= 0; --l) {
            const float* res = (l < 2) ? lat[l + 1] : nullptr;
            TRY(conv_run32(m->lateral[l], stage_out[l + 1], nf, sh[l + 1], sw[l + 1], lat[l], 0, res, res ? 2 : 0, cs));
            TRY(conv_run32(m->output[l], lat[l], nf, sh[l + 1], sw[l + 1], pout[l], 0, nullptr, 0, cs));
        }
        if (nchain > 1) {
            HIP_TRY(hipEventRecord(m->ev_join[c], cs));
            HIP_TRY(hipStreamWaitEvent(s, m->ev_join[c], 0));
        }
    }
    return DVID_OK;
}

// Swin-Transformer + FPN with DTYPE float32 (swintransformer.py:464-751): the fp16 path's launch sequence with fp32 operands everywhere
int backbone_swin_f32(dvid_model* m, const float* const* frames, int n, int height, int width, float* p3, float* p4, float* p5, hipStream_t s) {
    float mean[3], stdv[3];
    for (int i = 0; i < 3; ++i) {
        mean[i] = m->cfg.pixel_mean[i] / 255.f;
        stdv[i] = m->cfg.pixel_std[i] / 255.f;
    }
    float* img = m->img8.as<float>();
    TRY(dvid_f32_prep_images_launch(frames, img, n, height, width, mean, stdv, s));
    int H = height, W = width;
    float* x = m->sw_x.as<float>();
    float* x2 = m->sw_x2.as<float>();
    TRY(conv_run32(m->swin_patch, img, n, H, W, x, 0, nullptr, 0, s, &H, &W));
    TRY(dvid_add_layernorm_launch(x, nullptr, m->swin_patch_norm.g, m->swin_patch_norm.b, x, nullptr, n * H * W, m->swin[0].dim, 0, s));
    float* ln = m->sw_ln16.as<float>();
    float* qkv = m->sw_qkv16.as<float>();
    float* attn = m->sw_attn16.as<float>();
    float* hid = m->sw_h16.as<float>();
    float* stage_out[4] = {nullptr, m->c3.as<float>(), m->c4.as<float>(), m->c5.as<float>()};
    int sh[4], sw[4];
    for (int st = 0; st < 4; ++st) {
        const SwinStageW& S = m->swin[st];
        const int C = S.dim, M = n * H * W;
        for (size_t b = 0; b < S.blocks.size(); ++b) {
            const SwinBlockW& B = S.blocks[b];
            const int shift = (b % 2 == 0) ? 0 : 3;
            TRY(dvid_add_layernorm_launch(x, nullptr, B.norm1.g, B.norm1.b, ln, nullptr, M, C, 0, s));
            TRY(linear_run32(B.qkv, ln, M, qkv, 0, s));
            TRY(prof_other("swin_attn_f32", M, C, 49, 4.0 * M * 49.0 * C, (double)M * C * 4.0 * 4.0, s,
                           [&] { return dvid_f32_swin_window_attn_launch(qkv, B.qkv.bias, B.relbias, attn, n, H, W, C, S.heads, shift, s); }));
            TRY(conv_run32(B.proj, attn, M, 1, 1, x, 0, x, 1, s));                                // x += proj(attn)
            TRY(dvid_add_layernorm_launch(x, nullptr, B.norm2.g, B.norm2.b, ln, nullptr, M, C, 0, s));
            TRY(linear_run32(B.fc1, ln, M, hid, 2, s));                                           // exact GELU
            TRY(conv_run32(B.fc2, hid, M, 1, 1, x, 0, x, 1, s));                                  // x += fc2(...)
        }
        sh[st] = H;
        sw[st] = W;
        if (S.has_out) TRY(dvid_add_layernorm_launch(x, nullptr, S.out_norm.g, S.out_norm.b, stage_out[st], nullptr, M, C, 0, s));
        if (S.has_down) {
            TRY(dvid_patch_merge_ln_launch(x, S.down_norm.g, S.down_norm.b, nullptr, n, H, W, C, s, hid));
            H = (H + 1) / 2;
            W = (W + 1) / 2;
            TRY(linear_run32(S.down_red, hid, n * H * W, x2, 0, s));
            float* t = x;
            x = x2;
            x2 = t;
        }
    }
    float* pout[3] = {p3, p4, p5};
    for (int l = 2; l >= 0; --l) {
        const float* res = (l < 2) ? m->lat[l + 1].as<float>() : nullptr;
        TRY(conv_run32(m->lateral[l], stage_out[l + 1], n, sh[l + 1], sw[l + 1], m->lat[l].as<float>(), 0, res, res ? 2 : 0, s));
        TRY(conv_run32(m->output[l], m->lat[l].as<float>(), n, sh[l + 1], sw[l + 1], pout[l], 0, nullptr, 0, s));
    }
    return DVID_OK;
}

float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// box_head.py:218-223 + :734-741 on the host (a handful of distinct t values per config)
const std::vector<float>& time_embedding(dvid_model* m, int64_t t) {
    auto it = m->time_cache.find(t);
    if (it != m->time_cache.end()) return it->second;
    const int d = m->cfg.hidden_dim, td = 4 * d, half = d / 2;
    std::vector<float> emb(d), h1(td), out(td);
    const float e = logf(10000.f) / (half - 1);
    for (int i = 0; i < half; ++i) {
        const float a = (float)t * expf((float)i * -e);
        emb[i] = sinf(a);
        emb[half + i] = cosf(a);
    }
    for (int o = 0; o < td; ++o) {
        double acc = m->tm1_b[o];
        for (int i = 0; i < d; ++i) acc += (double)m->tm1_w[(size_t)o * d + i] * emb[i];
        h1[o] = gelu_exact((float)acc);
    }
    for (int o = 0; o < td; ++o) {
        double acc = m->tm3_b[o];
        for (int i = 0; i < td; ++i) acc += (double)m->tm3_w[(size_t)o * td + i] * h1[i];
        out[o] = (float)acc;
    }
    return m->time_cache.emplace(t, std::move(out)).first->second;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* dvid_last_error(void) { return g_err; }
int dvid_version(void) { return 1; }

int dvid_model_create(const dvid_config* cfg, dvid_model** out) {
    g_err[0] = 0;
    if (!cfg || !out) FAIL(DVID_ERR_ARG, "null argument");
    if (cfg->hidden_dim != 256 || cfg->nheads != 8 || cfg->dim_dynamic != 64 || cfg->pooler_resolution != 7 ||
        cfg->sampling_ratio != 2)
        FAIL(DVID_ERR_UNSUPPORTED,
             "kernels are specialised for HIDDEN_DIM 256, NHEADS 8, DIM_DYNAMIC 64, POOLER_RESOLUTION 7, SAMPLING_RATIO 2");
    if (cfg->dim_feedforward % 64 || cfg->num_classes < 1 || cfg->num_classes > 64)
        FAIL(DVID_ERR_UNSUPPORTED, "DIM_FEEDFORWARD must be a multiple of 64 and 1 <= NUM_CLASSES <= 64");
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) FAIL(DVID_ERR_HIP, "no HIP device available");
    dvid_model* m = new dvid_model();
    m->cfg = *cfg;
    if (const char* e = getenv("DVID_CHAINS")) m->nchain = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
    *out = m;
    return DVID_OK;
}

int dvid_model_destroy(dvid_model* m) {
    if (!m) return DVID_OK;
    for (void* p : m->owned) (void)hipFree(p);
    DevBuf* bufs[] = {&m->img8, &m->bufX, &m->bufY, &m->bufT1, &m->bufT2, &m->bufSC, &m->c3, &m->c4, &m->c5, &m->lat[0], &m->sw_x, &m->sw_x2, &m->sw_ln16, &m->sw_qkv16, &m->sw_attn16, &m->sw_h16,
                      &m->lat[1], &m->lat[2], &m->roi, &m->params, &m->dyn, &m->qkv, &m->attn16, &m->f32a, &m->f32b, &m->f32c,
                      &m->f32d, &m->h16a, &m->h16b, &m->hid16, &m->ss, &m->deltas, &m->kvproj, &m->mem16, &m->splitk, &m->vt};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : m->ss_slabs) b.release();
    delete m;
    return DVID_OK;
}

int dvid_model_set_tensor(dvid_model* m, const char* name, const float* data, const int64_t* shape, int ndim) {
    g_err[0] = 0;
    if (!m || !name || !data || ndim < 0 || ndim > 8) FAIL(DVID_ERR_ARG, "bad argument");
    if (m->finalized) FAIL(DVID_ERR_STATE, "model already finalized");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.v.assign(data, data + t.numel());
    m->raw[name] = std::move(t);
    return DVID_OK;
}

int dvid_model_finalize(dvid_model* m) {
    g_err[0] = 0;
    if (!m) FAIL(DVID_ERR_ARG, "null model");
    if (m->finalized) return DVID_OK;
    const dvid_config& c = m->cfg;
    m->has_backbone = c.backbone_type == 1 ? c.swin_depths[0] > 0 : c.res_blocks[0] > 0;
    if (m->has_backbone && c.backbone_type == 0) {
        const std::string bu = "backbone.bottom_up.";
        TRY(make_conv_bn(m, bu + "stem.conv1", 2, 3, 8, &m->stem));
        TRY(make_stem_s2d(m, bu + "stem.conv1", &m->stem_s2d));
        for (int s = 0; s < 4; ++s) {
            m->blocks[s].resize(c.res_blocks[s]);
            for (int b = 0; b < c.res_blocks[s]; ++b) {
                const std::string p = bu + "res" + std::to_string(s + 2) + "." + std::to_string(b);
                Block& blk = m->blocks[s][b];
                const int stride = (b == 0 && s > 0) ? 2 : 1;  // STRIDE_IN_1X1: False -> stride on the 3x3
                TRY(make_conv_bn(m, p + ".conv1", 1, 0, 0, &blk.c1));
                TRY(make_conv_bn(m, p + ".conv2", stride, 1, 0, &blk.c2));
                TRY(make_conv_bn(m, p + ".conv3", 1, 0, 0, &blk.c3));
                blk.has_sc = (b == 0);
                if (blk.has_sc) TRY(make_conv_bn(m, p + ".shortcut", stride, 0, 0, &blk.sc));
            }
        }
    }
    if (m->has_backbone && c.backbone_type == 1) {
        if (c.swin_window != 7) FAIL(DVID_ERR_UNSUPPORTED, "Swin window size %d (only 7 is built)", c.swin_window);
        const std::string bu = "backbone.bottom_up.";
        {
            NEED(pw, bu + "patch_embed.proj.weight");
            NEED(pb, bu + "patch_embed.proj.bias");
            if (pw->shape[2] != 4 || pw->shape[3] != 4) FAIL(DVID_ERR_UNSUPPORTED, "patch size must be 4");
            TRY(make_conv(m, *pw, {}, pb->v, 4, 0, 8, nullptr, &m->swin_patch));
            TRY(make_ln(m, bu + "patch_embed.norm", &m->swin_patch_norm));
        }
        // relative position index of a 7x7 window (swintransformer.py:122-131)
        int relidx[49][49];
        for (int i = 0; i < 49; ++i)
            for (int j = 0; j < 49; ++j) relidx[i][j] = ((i / 7 - j / 7) + 6) * 13 + ((i % 7 - j % 7) + 6);
        for (int st = 0; st < 4; ++st) {
            SwinStageW& S = m->swin[st];
            S.dim = c.swin_embed_dim << st;
            S.heads = c.swin_heads[st];
            if (S.dim != S.heads * 32) FAIL(DVID_ERR_UNSUPPORTED, "Swin stage %d: head dim %d != 32", st, S.dim / S.heads);
            S.blocks.resize(c.swin_depths[st]);
            for (int b = 0; b < c.swin_depths[st]; ++b) {
                const std::string p = bu + "layers." + std::to_string(st) + ".blocks." + std::to_string(b);
                SwinBlockW& B = S.blocks[b];
                TRY(make_ln(m, p + ".norm1", &B.norm1));
                TRY(make_ln(m, p + ".norm2", &B.norm2));
                TRY(make_linear(m, p + ".attn.qkv", true, &B.qkv));
                TRY(make_linear(m, p + ".attn.proj", true, &B.proj));
                TRY(make_linear(m, p + ".mlp.fc1", true, &B.fc1));
                TRY(make_linear(m, p + ".mlp.fc2", true, &B.fc2));
                NEED(qb, p + ".attn.qkv.bias");
                std::vector<half_t> qb16(qb->v.size());
                for (size_t i = 0; i < qb16.size(); ++i) qb16[i] = f2h(qb->v[i]);
                TRY(m->upload(qb16.data(), qb16.size() * sizeof(half_t), reinterpret_cast<void**>(&B.qkv_bias16)));
                NEED(tb, p + ".attn.relative_position_bias_table");
                if (tb->shape[0] != 169 || tb->shape[1] != S.heads) FAIL(DVID_ERR_ARG, "%s: bad bias table shape", p.c_str());
                // one 256-byte row per (head, query): a lane fetches the bias of its 16 keys as four aligned 16-byte loads
                std::vector<float> rb((size_t)S.heads * 49 * SWIN_RELBIAS_PITCH, 0.f);
                for (int h = 0; h < S.heads; ++h)
                    for (int i = 0; i < 49; ++i)
                        for (int j = 0; j < 49; ++j)
                            rb[((size_t)h * 49 + i) * SWIN_RELBIAS_PITCH + j] = tb->v[(size_t)relidx[i][j] * S.heads + h];
                TRY(upload_f32(m, rb, &B.relbias));
            }
            S.has_down = st < 3;
            if (S.has_down) {
                const std::string p = bu + "layers." + std::to_string(st) + ".downsample";
                TRY(make_ln(m, p + ".norm", &S.down_norm));
                TRY(make_linear(m, p + ".reduction", false, &S.down_red));
            }
            S.has_out = st >= 1;                       // out_indices (1, 2, 3)
            if (S.has_out) TRY(make_ln(m, bu + "norm" + std::to_string(st), &S.out_norm));
        }
    }
    if (m->has_backbone) {
        for (int l = 0; l < 3; ++l) {
            const std::string lat = "backbone.fpn_lateral" + std::to_string(l + 3);
            const std::string outn = "backbone.fpn_output" + std::to_string(l + 3);
            NEED(lw, lat + ".weight");
            NEED(lb, lat + ".bias");
            NEED(ow, outn + ".weight");
            NEED(ob, outn + ".bias");
            TRY(make_conv(m, *lw, {}, lb->v, 1, 0, 0, nullptr, &m->lateral[l]));
            TRY(make_conv(m, *ow, {}, ob->v, 1, 1, 0, nullptr, &m->output[l]));
        }
    }
    m->heads.resize(c.num_heads);
    for (int i = 0; i < c.num_heads; ++i) TRY(make_head(m, "head.head_series." + std::to_string(i), false, &m->heads[i]));
    m->heads_cond.resize(c.num_heads_cond);
    for (int i = 0; i < c.num_heads_cond; ++i)
        TRY(make_head(m, "head.head_series_cond." + std::to_string(i), true, &m->heads_cond[i]));
    if (m->get("head.global_attention.0.0.in_proj_weight")) {
        const int d = c.hidden_dim;
        TRY(make_linear(m, "head.global_attention.0.0.in_proj", true, &m->gq, nullptr, 0, d));
        TRY(make_linear(m, "head.global_attention.0.0.in_proj", true, &m->gkv, nullptr, d, 2 * d));
        TRY(make_linear(m, "head.global_attention.0.0.out_proj", true, &m->gout));
    }
    {
        NEED(w1, "head.time_mlp.1.weight");
        NEED(b1, "head.time_mlp.1.bias");
        NEED(w3, "head.time_mlp.3.weight");
        NEED(b3, "head.time_mlp.3.bias");
        m->tm1_w = w1->v;
        m->tm1_b = b1->v;
        m->tm3_w = w3->v;
        m->tm3_b = b3->v;
    }
    m->raw.clear();
    m->finalized = true;
    return DVID_OK;
}

int dvid_model_set_precision(dvid_model* m, int precision) {
    g_err[0] = 0;
    if (!m || (precision != 0 && precision != 1)) FAIL(DVID_ERR_ARG, "precision must be 0 (float16) or 1 (float32)");
    if (m->finalized) FAIL(DVID_ERR_STATE, "dvid_model_set_precision must precede dvid_model_finalize (the weights are packed for one precision)");
    m->precision = precision;
    return DVID_OK;
}

namespace {
struct OptEntry {
    const char* name;
    int DvidOptions::*field;
    int lo, hi;
};
const OptEntry kOptions[] = {
    {"conv3x3", &DvidOptions::conv3x3, 0, 2},       {"wstat", &DvidOptions::wstat, 0, 2},           {"bneck_fuse", &DvidOptions::bneck_fuse, 0, 2},
    {"stem_pool", &DvidOptions::stem_pool, 0, 1},   {"head_tail", &DvidOptions::head_tail, 0, 1},   {"roi_fuse", &DvidOptions::roi_fuse, 0, 1},   {"ln_rows", &DvidOptions::ln_rows, 0, 1},
    {"igemm_cfg", &DvidOptions::igemm_cfg, -1, 255}, {"igemm_tune", &DvidOptions::igemm_tune, -1, 1}, {"igemm_generic", &DvidOptions::igemm_generic, 0, 1},
    {"f32_split", &DvidOptions::f32_split, 0, 1},    {"f32_wstat", &DvidOptions::f32_wstat, 0, 2},    {"f32_conv3x3", &DvidOptions::f32_conv3x3, 0, 1},
    {"bneck_lds", &DvidOptions::bneck_lds, 0, 160 * 1024},
};
}  // namespace

int dvid_set_option(const char* name, int value) {
    g_err[0] = 0;
    if (!name) FAIL(DVID_ERR_ARG, "null option name");
    for (const OptEntry& e : kOptions)
        if (!strcmp(e.name, name)) {
            if (value < e.lo || value > e.hi) FAIL(DVID_ERR_ARG, "option %s: %d outside [%d, %d]", name, value, e.lo, e.hi);
            if (e.field == &DvidOptions::igemm_cfg && value >= dvid_igemm_num_configs()) FAIL(DVID_ERR_ARG, "igemm_cfg %d: the table has %d entries", value, dvid_igemm_num_configs());
            g_opt.*(e.field) = value;
            return DVID_OK;
        }
    FAIL(DVID_ERR_ARG, "unknown option '%s'", name);
}

int dvid_get_option(const char* name, int* value) {
    g_err[0] = 0;
    if (!name || !value) FAIL(DVID_ERR_ARG, "null argument");
    for (const OptEntry& e : kOptions)
        if (!strcmp(e.name, name)) {
            *value = g_opt.*(e.field);
            return DVID_OK;
        }
    FAIL(DVID_ERR_ARG, "unknown option '%s'", name);
}

int dvid_reset_options(void) {
    g_opt = DvidOptions();
    return DVID_OK;
}

// "name=value ..." of every option, then the environment switches the library still reads, as they are set
int dvid_effective_config(char* buf, int cap) {
    g_err[0] = 0;
    if (!buf || cap <= 0) FAIL(DVID_ERR_ARG, "no buffer");
    std::string out;
    for (const OptEntry& e : kOptions) out += std::string(e.name) + "=" + std::to_string(g_opt.*(e.field)) + " ";
    for (const char* env : {"DVID_IGEMM_TUNE", "DVID_IGEMM_TUNE_CACHE", "DVID_CHAINS", "DVID_POISON_WORKSPACE"}) {
        const char* v = getenv(env);
        out += std::string(env) + "=" + (v ? v : "") + " ";
    }
    if (!out.empty()) out.pop_back();
    if ((int)out.size() + 1 > cap) FAIL(DVID_ERR_ARG, "buffer of %d bytes, need %d", cap, (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return DVID_OK;
}

// DTYPE float32 with split operands: 1 if any launch since the last call staged an activation whose magnitude exceeds the fp16 range (its
// products are then inf / NaN where fp32 arithmetic would be finite); reads 4 bytes from the device behind `stream` and clears the flag
int dvid_model_take_range_flag(dvid_model* m, int* exceeded, void* stream) {
    g_err[0] = 0;
    if (!m || !exceeded) FAIL(DVID_ERR_ARG, "null argument");
    *exceeded = 0;
    if (!m->f32_range_flag) return DVID_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(exceeded, m->f32_range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*exceeded) HIP_TRY(hipMemsetAsync(m->f32_range_flag, 0, sizeof(int), s));
    return DVID_OK;
}

int dvid_set_stem_pool(int mode) {
    if (mode < -1 || mode > 1) return DVID_ERR_ARG;
    g_opt.stem_pool = mode < 0 ? DvidOptions().stem_pool : mode;
    return DVID_OK;
}

unsigned long long dvid_workspace_generation(const dvid_model* m) { return m ? m->ws_gen.load(std::memory_order_relaxed) : 0ull; }

int dvid_set_chains(dvid_model* m, int nchain) {
    g_err[0] = 0;
    if (!m || nchain < 1 || nchain > 4) FAIL(DVID_ERR_ARG, "nchain must be 1..4");
    m->nchain = nchain;
    return DVID_OK;
}

int dvid_set_stem_layout(dvid_model* m, int space_to_depth) {
    g_err[0] = 0;
    if (!m) FAIL(DVID_ERR_ARG, "null model");
    m->use_s2d = space_to_depth != 0;
    return DVID_OK;
}

int dvid_workspace_reserve(dvid_model* m, int max_frames, int height, int width, int boxes_per_frame) {
    g_err[0] = 0;
    if (!m || max_frames <= 0 || boxes_per_frame <= 0) FAIL(DVID_ERR_ARG, "bad argument");
    if (height % 32 || width % 32) FAIL(DVID_ERR_ARG, "height/width must be multiples of 32 (got %dx%d)", height, width);
    const size_t n = ((size_t)max_frames + 3) / 4 * 4;      // room for up to 4 equal sub-batch chains
    const size_t px4 = (size_t)(height / 4) * (width / 4);
    const size_t es = m->precision == 1 ? 2 : 1;            // DTYPE float32: every fp16 buffer below holds fp32 values instead
    if (m->has_backbone && m->cfg.backbone_type == 1) {
        const size_t C0 = m->cfg.swin_embed_dim, M0 = n * px4;      // stage-0 tokens; M*C halves per stage
        TRY(m->img8.ensure(n * height * width * 8 * 2, &m->ws_gen));
        TRY(m->sw_x.ensure(M0 * C0 * 4, &m->ws_gen));
        TRY(m->sw_x2.ensure(M0 * C0 * 4 / 2, &m->ws_gen));
        TRY(m->sw_ln16.ensure(M0 * C0 * 2 * es, &m->ws_gen));
        TRY(m->sw_qkv16.ensure(M0 * C0 * 3 * 2 * es, &m->ws_gen));
        TRY(m->sw_attn16.ensure(M0 * C0 * 2 * es, &m->ws_gen));
        TRY(m->sw_h16.ensure(M0 * C0 * 4 * 2 * es, &m->ws_gen));
        TRY(m->c3.ensure(n * (px4 / 4) * (C0 * 2) * 2 * es, &m->ws_gen));
        TRY(m->c4.ensure(n * (px4 / 16) * (C0 * 4) * 2 * es, &m->ws_gen));
        TRY(m->c5.ensure(n * (px4 / 64) * (C0 * 8) * 2 * es, &m->ws_gen));
        for (int l = 0; l < 3; ++l) TRY(m->lat[l].ensure(n * (px4 / (4 << (2 * l))) * 256 * 2 * es, &m->ws_gen));
    }
    if (m->has_backbone && m->cfg.backbone_type == 0) {
        TRY(m->img8.ensure(n * height * width * 8 * 2, &m->ws_gen));          // (fp32: NHWC4 = the same bytes)
        const size_t big = n * px4 * 256 * 2 * es;  // largest activation: res2 output (also >= stem output)
        TRY(m->bufX.ensure(big, &m->ws_gen));
        TRY(m->bufY.ensure(big, &m->ws_gen));
        TRY(m->bufT1.ensure(big, &m->ws_gen));
        TRY(m->bufT2.ensure(big, &m->ws_gen));
        TRY(m->bufSC.ensure(big, &m->ws_gen));
        TRY(m->c3.ensure(n * (px4 / 4) * 512 * 2 * es, &m->ws_gen));
        TRY(m->c4.ensure(n * (px4 / 16) * 1024 * 2 * es, &m->ws_gen));
        TRY(m->c5.ensure(n * (px4 / 64) * 2048 * 2 * es, &m->ws_gen));
        for (int l = 0; l < 3; ++l) TRY(m->lat[l].ensure(n * (px4 / (4 << (2 * l))) * 256 * 2 * es, &m->ws_gen));
    }
    const size_t R = n * boxes_per_frame;
    const int d = m->cfg.hidden_dim;
    TRY(m->roi.ensure(R * 49 * d * 2 * es, &m->ws_gen));
    TRY(m->dyn.ensure(R * 49 * d * 2 * es, &m->ws_gen));
    TRY(m->params.ensure(R * 2 * d * m->cfg.dim_dynamic * 2 * es, &m->ws_gen));
    TRY(m->qkv.ensure(R * 3 * d * 4, &m->ws_gen));
    TRY(m->attn16.ensure(R * d * 2 * es, &m->ws_gen));
    TRY(m->f32a.ensure(R * d * 4, &m->ws_gen));
    TRY(m->f32b.ensure(R * d * 4, &m->ws_gen));
    TRY(m->f32c.ensure(R * d * 4, &m->ws_gen));
    TRY(m->f32d.ensure(R * d * 4, &m->ws_gen));
    TRY(m->h16a.ensure(R * d * 2 * es, &m->ws_gen));
    TRY(m->h16b.ensure(R * d * 2 * es, &m->ws_gen));
    TRY(m->hid16.ensure(R * m->cfg.dim_feedforward * 2 * es, &m->ws_gen));
    TRY(m->ss.ensure((size_t)(m->cfg.num_heads + m->cfg.num_heads_cond) * n * 2 * d * 4, &m->ws_gen));
    TRY(m->deltas.ensure(R * 4 * 4, &m->ws_gen));
    TRY(m->splitk.ensure(R * d * 4 * 8, &m->ws_gen));
    TRY(m->vt.ensure((size_t)n * m->cfg.nheads * 32 * (((size_t)boxes_per_frame + 31) / 32 * 32 + 32) * 2, &m->ws_gen));          // up to 8 split-K slabs of an [R, d] fp32 output
    m->ws_frames = max_frames;
    m->ws_h = height;
    m->ws_w = width;
    m->ws_boxes = boxes_per_frame;
    return DVID_OK;
}

int dvid_backbone_resnet_fpn(dvid_model* m, const float* images, int n, int height, int width, void* p3, void* p4, void* p5,
                             void* stream) {
    if (!images || n <= 0) {
        g_err[0] = 0;
        FAIL(DVID_ERR_ARG, "no images");
    }
    std::vector<const float*> frames(n);
    for (int i = 0; i < n; ++i) frames[i] = images + (size_t)i * 3 * height * width;
    return dvid_backbone_resnet_fpn_frames(m, frames.data(), n, height, width, p3, p4, p5, stream);
}

int dvid_backbone_resnet_fpn_frames(dvid_model* m, const float* const* frames, int n, int height, int width, void* p3, void* p4, void* p5,
                                    void* stream) {
    g_err[0] = 0;
    if (!frames || n <= 0) FAIL(DVID_ERR_ARG, "no frames");
    if (!m || !m->finalized || !m->has_backbone || m->cfg.backbone_type != 0)
        FAIL(DVID_ERR_STATE, "model not finalized or built without a ResNet backbone");
    // capacity, not equality: a set mixes frame sizes (ImageNet-VID has 16:9 and 4:3 videos) and the workspace only grows
    if (n > m->ws_frames || height > m->ws_h || width > m->ws_w || height % 32 || width % 32)
        FAIL(DVID_ERR_STATE, "workspace reserved for %d frames of up to %dx%d, got %d of %dx%d", m->ws_frames, m->ws_h, m->ws_w, n,
             height, width);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (m->precision == 1)          // DTYPE float32: p3 / p4 / p5 are fp32 NHWC
        return backbone_resnet_f32(m, frames, n, height, width, reinterpret_cast<float*>(p3), reinterpret_cast<float*>(p4), reinterpret_cast<float*>(p5), s);
    float mean[3], inv_std[3];
    for (int i = 0; i < 3; ++i) {
        mean[i] = m->cfg.pixel_mean[i] / 255.f;
        inv_std[i] = 1.f / (m->cfg.pixel_std[i] / 255.f);
    }
    // Frames are independent through the backbone.  They are processed as `nchain` sub-batches on separate HIP
    // streams: a layer of one sub-batch rarely fills 256 CUs evenly (e.g. res4: 304-608 tiles), and with two chains
    // in flight the blocks of one chain's next kernel start on the CUs the other chain's tail leaves idle.  (A two-stream
    // front / back software pipeline of HBM-bound early layers beside MFMA-bound late ones measured no gain,
    // profiles/r02_backbone_pipeline_sweep.txt; it lives in the history of this file.)
    // Small launch sequences stay on one stream: at 8 frames two 4-frame chains are slower than one 8-frame sequence (1250 vs 1273
    // frames/s with the reference's one-batch-per-call protocol, 980 with four chains; profiles/r03c_chains_at_lookahead1.txt) --
    // the layers are then bound by how few workgroups a launch has, and halving the rows halves them again.
    const int nchain = (m->nchain > 1 && n >= 16 * m->nchain) ? m->nchain : 1;
    if (nchain > 1) TRY(m->ensure_streams());
    const int per = (n + nchain - 1) / nchain;
    const size_t px = (size_t)height * width, px4 = px / 16;
    if (nchain > 1) {
        HIP_TRY(hipEventRecord(m->ev_fork, s));
        for (int c = 0; c < nchain; ++c) HIP_TRY(hipStreamWaitEvent(m->cs[c], m->ev_fork, 0));
    }
    for (int c = 0; c < nchain; ++c) {
        const int f0 = c * per, nf = (f0 + per <= n) ? per : n - f0;
        if (nf <= 0) continue;
        hipStream_t cs = nchain > 1 ? m->cs[c] : s;
        // this chain's slice of every workspace buffer starts at its first frame (f0 + nf <= n <= ws_frames for any
        // chain count, so no slice can run past the end)
        const size_t fo = (size_t)f0;
        half_t* img8 = m->img8.as<half_t>() + fo * px * 8;
        const size_t big = fo * px4 * 256;
        half_t* bx = m->bufX.as<half_t>() + big;
        half_t* by = m->bufY.as<half_t>() + big;
        half_t* t1 = m->bufT1.as<half_t>() + big;
        half_t* t2 = m->bufT2.as<half_t>() + big;
        half_t* sc = m->bufSC.as<half_t>() + big;
        half_t* stage_out[4] = {nullptr, m->c3.as<half_t>() + fo * (px4 / 4) * 512,
                                m->c4.as<half_t>() + fo * (px4 / 16) * 1024,
                                m->c5.as<half_t>() + fo * (px4 / 64) * 2048};
        half_t* lat[3];
        for (int l = 0; l < 3; ++l) lat[l] = m->lat[l].as<half_t>() + fo * (px4 / (4 << (2 * l))) * 256;

        int h = height, w = width;
        bool pooled = false;
        if (m->use_s2d) {
            // normalise + 2x2 space-to-depth (16 halves per block: the same bytes per frame as half an NHWC8 image), then the stem
            // as a 4x4 / stride-1 convolution on the half-resolution grid
            TRY(dvid_prep_images_s2d_launch(frames + f0, img8, nf, height, width, mean, inv_std, cs));
            // (only while the patch kernels are on and no tile configuration is forced: "all layers on igemm2" runs -- conv3x3 = 0,
            // dvid_igemm_set_config -- then include the stem, whose fused kernel sums in the patch kernels' order)
            if (g_opt.stem_pool && g_opt.conv3x3 && g_opt.igemm_cfg < 0) {
                // stem + ReLU + max pool as one launch (csrc/conv3x3.hip: stem_pool_kernel): the half-resolution 64-channel map never exists
                TRY(conv_run(m->stem_s2d, img8, nf, h / 2, w / 2, bx, 1, 0, nullptr, 0, 0, cs, &h, &w, 0, 1, /*pooled=*/true));
                pooled = true;
            } else {
                TRY(conv_run(m->stem_s2d, img8, nf, h / 2, w / 2, t1, 1, 0, nullptr, 0, 0, cs, &h, &w));
            }
        } else {
            TRY(dvid_prep_images_launch(frames + f0, img8, nf, height, width, mean, inv_std, cs));
            TRY(conv_run(m->stem, img8, nf, h, w, t1, 1, 0, nullptr, 0, 0, cs, &h, &w));
        }
        if (!pooled) TRY(prof_other("maxpool", (long)nf * h * w, 64, 9, 0.0, (double)nf * h * w * 64 * 2.0 * 1.25, cs, [&] { return dvid_maxpool3x3s2_launch(t1, bx, nf, h, w, 64, cs); }));
        h = (h + 2 - 3) / 2 + 1;
        w = (w + 2 - 3) / 2 + 1;
        half_t* cur = bx;  // block input
        half_t* res3_t1 = nullptr;
        int sh[4], sw[4];
        for (int st = 0; st < 4; ++st) {
            const int nb = (int)m->blocks[st].size();
            // res2 (64-wide bottlenecks, 256 out): one launch per block for everything behind conv1 -- conv2, conv3 + shortcut / residual
            // + ReLU and the next block's conv1 (csrc/bneck.hip; bit-identical to the launches below)
            if (st == 0 && bneck64_stage(m->blocks[0]) && dvid_bneck64_tail_preferred(h, w)) {
                half_t* ta = t1;
                half_t* tb = t2;
                TRY(conv_run(m->blocks[0][0].c1, cur, nf, h, w, ta, 1, 0, nullptr, 0, 0, cs));
                // the last block's launch also computes res3's first conv1 (1x1 / stride 1 over this stage's output, 256 -> 128) when
                // res3 takes the fused path too: that layer alone re-read the 512 B per pixel this launch has in registers
                const Block* r3 = nullptr;
                if (nb > 1 && bneck128_stage(m->blocks[1])) {
                    const Block& b0 = m->blocks[1][0];
                    auto osz = [](const ConvW& c, int v) { return (v + 2 * c.pad - c.kh) / c.stride + 1; };
                    if (b0.c1.kh == 1 && b0.c1.stride == 1 && b0.c1.pad == 0 && b0.c1.cin == 256 && b0.c1.cout == 128 && b0.c1.kpad == 256 &&
                        b0.c1.bias && dvid_bneck64_tail_preferred(osz(b0.c2, h), osz(b0.c2, w)))
                        r3 = &b0;
                }
                for (int b = 0; b < nb; ++b) {
                    const Block& blk = m->blocks[0][b];
                    const Block* nxt = b + 1 < nb ? &m->blocks[0][b + 1] : r3;
                    half_t* dst = cur == bx ? by : bx;
                    TRY(bneck_tail(ta, blk.c2.w, blk.c2.bias, blk.c3.w, blk.c3.bias, cur, blk.has_sc ? blk.sc.w : nullptr,
                                   blk.has_sc ? blk.sc.bias : nullptr, nxt ? nxt->c1.w : nullptr, nxt ? nxt->c1.bias : nullptr,
                                   nxt ? nxt->c1.cout : 0, dst, nxt ? tb : nullptr, nf, h, w, cs));
                    std::swap(ta, tb);
                    cur = dst;
                }
                if (r3) res3_t1 = ta;                     // res3's first conv1 output, already computed
                sh[st] = h;
                sw[st] = w;
                continue;
            }
            // res3 (128-wide): the first block's conv1 / strided conv2 / shortcut as their own launches, then one launch per block for
            // conv3 + residual + ReLU + the next block's conv1 (+ the next block's conv2 in front of them)
            if (st == 1 && bneck128_stage(m->blocks[1])) {
                const Block& b0 = m->blocks[1][0];
                auto osz = [](const ConvW& c, int v) { return (v + 2 * c.pad - c.kh) / c.stride + 1; };
                if (dvid_bneck64_tail_preferred(osz(b0.c2, osz(b0.c1, h)), osz(b0.c2, osz(b0.c1, w)))) {
                    int h2 = h, w2 = w;
                    half_t* c1out = res3_t1 ? res3_t1 : t1;           // (res2's last launch may have computed it)
                    half_t* c2out = c1out == t1 ? t2 : t1;
                    if (!res3_t1) TRY(conv_run(b0.c1, cur, nf, h, w, c1out, 1, 0, nullptr, 0, 0, cs, &h2, &w2));
                    const int h1 = h2, w1 = w2;
                    TRY(conv_run(b0.c2, c1out, nf, h1, w1, c2out, 1, 0, nullptr, 0, 0, cs, &h2, &w2));
                    TRY(conv_run(b0.sc, cur, nf, h, w, sc, 0, 0, nullptr, 0, 0, cs));
                    h = h2;
                    w = w2;
                    half_t* ta = c1out;                   // free again: conv2 has consumed it
                    half_t* tb = c2out;
                    half_t* dst = cur == bx ? by : bx;
                    TRY(bneck128_tail(c2out, nullptr, nullptr, b0.c3.w, b0.c3.bias, sc, m->blocks[1][1].c1.w, m->blocks[1][1].c1.bias, dst, ta, nf, h,
                                      w, cs));
                    cur = dst;
                    for (int b = 1; b < nb; ++b) {
                        const Block& blk = m->blocks[1][b];
                        const Block* nxt = b + 1 < nb ? &m->blocks[1][b + 1] : nullptr;
                        dst = (b == nb - 1 && stage_out[st]) ? stage_out[st] : (cur == bx ? by : bx);
                        TRY(bneck128_tail(ta, blk.c2.w, blk.c2.bias, blk.c3.w, blk.c3.bias, cur, nxt ? nxt->c1.w : nullptr,
                                          nxt ? nxt->c1.bias : nullptr, dst, nxt ? tb : nullptr, nf, h, w, cs));
                        std::swap(ta, tb);
                        cur = dst;
                    }
                    sh[st] = h;
                    sw[st] = w;
                    continue;
                }
            }
            for (int b = 0; b < nb; ++b) {
                const Block& blk = m->blocks[st][b];
                int h2 = h, w2 = w;
                TRY(conv_run(blk.c1, cur, nf, h, w, t1, 1, 0, nullptr, 0, 0, cs));
                TRY(conv_run(blk.c2, t1, nf, h, w, t2, 1, 0, nullptr, 0, 0, cs, &h2, &w2));
                const half_t* res = cur;
                if (blk.has_sc) {
                    TRY(conv_run(blk.sc, cur, nf, h, w, sc, 0, 0, nullptr, 0, 0, cs));
                    res = sc;
                }
                // res3..res5 outputs persist for the FPN; everything else ping-pongs between bufX/bufY
                half_t* dst = (b == nb - 1 && stage_out[st]) ? stage_out[st] : (cur == bx ? by : bx);
                TRY(conv_run(blk.c3, t2, nf, h2, w2, dst, 1, 0, res, 1, 0, cs));
                h = h2;
                w = w2;
                cur = dst;
            }
            sh[st] = h;
            sw[st] = w;
        }
        // FPN: outputs go to the caller's [n, ...] tensors at this chain's frame offset
        void* pout[3] = {reinterpret_cast<half_t*>(p3) + (size_t)f0 * sh[1] * sw[1] * 256,
                         reinterpret_cast<half_t*>(p4) + (size_t)f0 * sh[2] * sw[2] * 256,
                         reinterpret_cast<half_t*>(p5) + (size_t)f0 * sh[3] * sw[3] * 256};
        for (int l = 2; l >= 0; --l) {
            const void* res = (l < 2) ? lat[l + 1] : nullptr;
            TRY(conv_run(m->lateral[l], stage_out[l + 1], nf, sh[l + 1], sw[l + 1], lat[l], 0, 0, res, res ? 2 : 0, 0, cs));
            TRY(conv_run(m->output[l], lat[l], nf, sh[l + 1], sw[l + 1], pout[l], 0, 0, nullptr, 0, 0, cs));
        }
        if (nchain > 1) {
            HIP_TRY(hipEventRecord(m->ev_join[c], cs));
            HIP_TRY(hipStreamWaitEvent(s, m->ev_join[c], 0));
        }
    }
    return DVID_OK;
}

int dvid_backbone_swin_fpn(dvid_model* m, const float* images, int n, int height, int width, void* p3, void* p4, void* p5,
                           void* stream) {
    if (!images || n <= 0) {
        g_err[0] = 0;
        FAIL(DVID_ERR_ARG, "no images");
    }
    std::vector<const float*> frames(n);
    for (int i = 0; i < n; ++i) frames[i] = images + (size_t)i * 3 * height * width;
    return dvid_backbone_swin_fpn_frames(m, frames.data(), n, height, width, p3, p4, p5, stream);
}

int dvid_backbone_swin_fpn_frames(dvid_model* m, const float* const* frames, int n, int height, int width, void* p3, void* p4, void* p5,
                                  void* stream) {
    g_err[0] = 0;
    if (!frames || n <= 0) FAIL(DVID_ERR_ARG, "no frames");
    if (!m || !m->finalized || !m->has_backbone || m->cfg.backbone_type != 1) FAIL(DVID_ERR_STATE, "model has no Swin backbone");
    // capacity, not equality: a set mixes frame sizes (ImageNet-VID has 16:9 and 4:3 videos) and the workspace only grows
    if (n > m->ws_frames || height > m->ws_h || width > m->ws_w || height % 32 || width % 32)
        FAIL(DVID_ERR_STATE, "workspace reserved for %d frames of up to %dx%d, got %d of %dx%d", m->ws_frames, m->ws_h, m->ws_w, n,
             height, width);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (m->precision == 1)          // DTYPE float32: p3 / p4 / p5 are fp32 NHWC
        return backbone_swin_f32(m, frames, n, height, width, reinterpret_cast<float*>(p3), reinterpret_cast<float*>(p4), reinterpret_cast<float*>(p5), s);
    float mean[3], inv_std[3];
    for (int i = 0; i < 3; ++i) {
        mean[i] = m->cfg.pixel_mean[i] / 255.f;
        inv_std[i] = 1.f / (m->cfg.pixel_std[i] / 255.f);
    }
    TRY(dvid_prep_images_launch(frames, m->img8.as<half_t>(), n, height, width, mean, inv_std, s));
    // patch embedding: 4x4/4 conv (implicit GEMM on NHWC8) -> fp32 tokens -> LayerNorm  (swintransformer.py:441-458)
    int H = height, W = width;
    float* x = m->sw_x.as<float>();
    float* x2 = m->sw_x2.as<float>();
    TRY(conv_run(m->swin_patch, m->img8.as<half_t>(), n, H, W, x, 0, 1, nullptr, 0, 0, s, &H, &W));
    TRY(dvid_add_layernorm_launch(x, nullptr, m->swin_patch_norm.g, m->swin_patch_norm.b, x, nullptr, n * H * W, m->swin[0].dim, 0, s));
    half_t* ln16 = m->sw_ln16.as<half_t>();
    half_t* qkv16 = m->sw_qkv16.as<half_t>();
    half_t* attn16 = m->sw_attn16.as<half_t>();
    half_t* h16 = m->sw_h16.as<half_t>();
    half_t* stage_out[4] = {nullptr, m->c3.as<half_t>(), m->c4.as<half_t>(), m->c5.as<half_t>()};
    int sh[4], sw[4];
    for (int st = 0; st < 4; ++st) {
        const SwinStageW& S = m->swin[st];
        const int C = S.dim, M = n * H * W;
        for (size_t b = 0; b < S.blocks.size(); ++b) {
            const SwinBlockW& B = S.blocks[b];
            const int shift = (b % 2 == 0) ? 0 : 3;                                             // window_size // 2
            TRY(dvid_add_layernorm_launch(x, nullptr, B.norm1.g, B.norm1.b, nullptr, ln16, M, C, 0, s));
            TRY(linear_run(B.qkv, ln16, M, qkv16, 0, 0, s));
            TRY(dvid_swin_window_attn_launch(qkv16, B.qkv_bias16, B.relbias, attn16, n, H, W, C, S.heads, shift, s));
            TRY(conv_run(B.proj, attn16, M, 1, 1, x, 0, 1, x, 1, 1, s));                         // x += proj(attn)   (fp32 stream)
            TRY(dvid_add_layernorm_launch(x, nullptr, B.norm2.g, B.norm2.b, nullptr, ln16, M, C, 0, s));
            TRY(linear_run(B.fc1, ln16, M, h16, 2, 0, s));                                       // GELU epilogue
            TRY(conv_run(B.fc2, h16, M, 1, 1, x, 0, 1, x, 1, 1, s));                             // x += fc2(...)
        }
        sh[st] = H;
        sw[st] = W;
        if (S.has_out) TRY(dvid_add_layernorm_launch(x, nullptr, S.out_norm.g, S.out_norm.b, nullptr, stage_out[st], M, C, 0, s));
        if (S.has_down) {
            TRY(dvid_patch_merge_ln_launch(x, S.down_norm.g, S.down_norm.b, h16, n, H, W, C, s));
            H = (H + 1) / 2;
            W = (W + 1) / 2;
            TRY(linear_run(S.down_red, h16, n * H * W, x2, 0, 1, s));
            float* t = x;
            x = x2;
            x2 = t;
        }
    }
    return run_fpn(m, n, sh + 1, sw + 1, p3, p4, p5, s);
}

int dvid_rcnn_head(dvid_model* m, int head_index, int is_cond, const void* p3, const void* p4, const void* p5, int n_frames,
                   int height, int width, int boxes_per_frame, const float* boxes, const float* pro_features,
                   const int64_t* t, const float* cond, float* logits, float* boxes_out, float* obj_features,
                   int* bad_box_flag, void* stream) {
    g_err[0] = 0;
    if (!m || !m->finalized) FAIL(DVID_ERR_STATE, "model not finalized");
    const std::vector<HeadW>& hv = is_cond ? m->heads_cond : m->heads;
    if (head_index < 0 || head_index >= (int)hv.size()) FAIL(DVID_ERR_ARG, "head_index %d out of range", head_index);
    if (is_cond && !cond) FAIL(DVID_ERR_ARG, "RCNNHead_cond needs cond");
    if (n_frames > m->ws_frames || boxes_per_frame > m->ws_boxes) FAIL(DVID_ERR_STATE, "workspace too small; call dvid_workspace_reserve");
    if (height % 32 || width % 32) FAIL(DVID_ERR_ARG, "height/width must be multiples of 32");
    const HeadW& hw = hv[head_index];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int d = m->cfg.hidden_dim, M = boxes_per_frame;

    // --- time conditioning (box_head.py:533-536 / :645): a scale/shift row is a function of (head, t) only.  Every distinct
    // (head slot, t value) keeps ONE device row, computed on the host the first time it is seen; a call whose frames share
    // one t (every call of the reference's sampler) reads that row with frame stride 0, so the steady state -- including the 4
    // alternating time steps of the x4 sampler and any ragged tail length -- does no host math, no upload and no stream sync.
    const int slot = (is_cond ? m->cfg.num_heads : 0) + head_index;
    auto ss_row = [&](int64_t tv, const float** dev) -> int {
        auto key = std::make_pair(slot, tv);
        auto it = m->ss_rows.find(key);
        if (it == m->ss_rows.end()) {
            const int td = 4 * d;
            const std::vector<float>& te = time_embedding(m, tv);
            std::vector<float> sl(td), row(hw.bt_out);
            for (int i = 0; i < td; ++i) sl[i] = te[i] / (1.f + expf(-te[i]));  // SiLU
            for (int o = 0; o < hw.bt_out; ++o) {
                double acc = hw.bt_b[o];
                for (int i = 0; i < td; ++i) acc += (double)hw.bt_w[(size_t)o * td + i] * sl[i];
                row[o] = (float)acc;
            }
            constexpr size_t kSsSlabRows = 256, kSsRowFloats = 512;          // bt_out = 2 d <= 512 floats
            if ((size_t)hw.bt_out > kSsRowFloats) return DVID_ERR_UNSUPPORTED;
            if (m->ss_slabs.empty() || m->ss_slab_used == kSsSlabRows) {
                m->ss_slabs.emplace_back();
                TRY(m->ss_slabs.back().ensure(kSsSlabRows * kSsRowFloats * sizeof(float), &m->ws_gen));
                m->ss_slab_used = 0;
            }
            float* dst = m->ss_slabs.back().as<float>() + (m->ss_slab_used++) * kSsRowFloats;
            HIP_TRY(hipMemcpy(dst, row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice));   // once per distinct (head, t)
            it = m->ss_rows.emplace(key, dst).first;
        }
        *dev = it->second;
        return DVID_OK;
    };
    const float* ss_dev = nullptr;
    int ss_stride = 0;              // floats between the rows of consecutive frames (0: one shared row)
    bool same_t = true;
    for (int f = 1; f < n_frames; ++f) same_t = same_t && t[f] == t[0];
    if (same_t) {
        TRY(ss_row(t[0], &ss_dev));
    } else {
        // frames with different time steps (not produced by the reference's sampler): the rows are laid out per frame in
        // the workspace by device-to-device copies on the launch stream
        float* tab = m->ss.as<float>() + (size_t)slot * (((size_t)m->ws_frames + 3) / 4 * 4) * 2 * d;          // slot stride of dvid_workspace_reserve
        for (int f = 0; f < n_frames; ++f) {
            const float* row = nullptr;
            TRY(ss_row(t[f], &row));
            HIP_TRY(hipMemcpyAsync(tab + (size_t)f * hw.bt_out, row, (size_t)hw.bt_out * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        ss_dev = tab;
        ss_stride = hw.bt_out;
    }

    // One launch sequence on the caller's stream.  (Frames are independent inside a head, but two sub-batch chains on two streams
    // measured slower -- 0.53 against 0.49 ms per pass, tools/bench_head.py -- the switch that kept that path is gone.)
    if (m->precision == 1)
        return rcnn_head_chain_f32(m, hw, is_cond, p3, p4, p5, 0, n_frames, height, width, M, boxes, pro_features, cond, logits, boxes_out, obj_features,
                                   bad_box_flag, ss_dev, ss_stride, 0, s);
    return rcnn_head_chain(m, hw, is_cond, p3, p4, p5, 0, n_frames, height, width, M, boxes, pro_features, cond, logits, boxes_out, obj_features,
                           bad_box_flag, ss_dev, ss_stride, 0, 0, s);
}

// K/V projections of the global memory (box_head.py:366-380 recomputes them on every call; they depend on the per-video
// memory only): projected once per memory update and kept until the next one.
int dvid_global_memory_project(dvid_model* m, const float* memory, int lk, void* stream) {
    g_err[0] = 0;
    if (!m || !m->finalized || !m->gq.w) FAIL(DVID_ERR_STATE, "model not finalized or has no global attention");
    if (!memory || lk <= 0) FAIL(DVID_ERR_ARG, "empty memory");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int d = m->cfg.hidden_dim;
    m->mem_lk = 0;
    TRY(m->kvproj.ensure((size_t)lk * 2 * d * 4, &m->ws_gen));
    if (m->precision == 1) {          // fp32 K | V rows
        TRY(linear_run32(m->gkv, memory, lk, m->kvproj.as<float>(), 0, s));
        m->mem_lk = lk;
        return DVID_OK;
    }
    TRY(m->mem16.ensure((size_t)lk * d * 2, &m->ws_gen));
    TRY(dvid_f32_to_f16_launch(memory, m->mem16.as<half_t>(), (long)lk * d, s));
    TRY(linear_run(m->gkv, m->mem16.as<half_t>(), lk, m->kvproj.p, 0, 0, s));
    m->mem_lk = lk;
    return DVID_OK;
}

int dvid_global_xattn(dvid_model* m, const float* query, int rows, const float* memory, int lk, float* out, void* stream) {
    g_err[0] = 0;
    if (!m || !m->finalized || !m->gq.w) FAIL(DVID_ERR_STATE, "model not finalized or has no global attention");
    if (rows > m->ws_frames * m->ws_boxes) FAIL(DVID_ERR_STATE, "workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int d = m->cfg.hidden_dim;
    if (memory) {
        TRY(dvid_global_memory_project(m, memory, lk, stream));
    } else if (m->mem_lk <= 0 || (lk > 0 && lk != m->mem_lk)) {
        FAIL(DVID_ERR_STATE, "no projected global memory of %d rows (call dvid_global_memory_project)", lk);
    }
    lk = m->mem_lk;
    if (m->precision == 1) {
        float* qp = m->h16a.as<float>();
        float* at = m->attn16.as<float>();
        const float* kv32 = m->kvproj.as<float>();
        TRY(linear_run32(m->gq, query, rows, qp, 0, s));
        TRY(dvid_f32_mha_launch(qp, kv32, kv32 + d, at, 1, rows, lk, m->cfg.nheads, d, 2 * d, d, 0, 0, 0, s));
        TRY(linear_run32(m->gout, at, rows, out, 0, s));
        return DVID_OK;
    }
    TRY(dvid_f32_to_f16_launch(query, m->h16a.as<half_t>(), (long)rows * d, s));
    TRY(linear_run(m->gq, m->h16a.as<half_t>(), rows, m->h16b.p, 0, 0, s));
    const half_t* kv = m->kvproj.as<half_t>();
    TRY(m->vt.ensure((size_t)m->cfg.nheads * 32 * (((size_t)lk + 31) / 32 * 32 + 32) * 2, &m->ws_gen));
    TRY(dvid_mha_mfma_launch(m->h16b.as<half_t>(), kv, kv + d, m->attn16.as<half_t>(), m->vt.as<half_t>(), 1, rows, lk, m->cfg.nheads,
                             d, 2 * d, d, 0, 0, 0, s));
    TRY(linear_run(m->gout, m->attn16.as<half_t>(), rows, out, 0, 1, s));
    return DVID_OK;
}

// ---- stand-alone ops ---------------------------------------------------------------------------
int dvid_roialign_v2_multilevel(const void* p3, const void* p4, const void* p5, int n_frames, int height, int width, int channels,
                                const float* boxes, int boxes_per_frame, void* roi_out, float* mean_out, void* stream) {
    g_err[0] = 0;
    if (height % 32 || width % 32) FAIL(DVID_ERR_ARG, "height/width must be multiples of 32");
    RoiLevels lv;
    const void* pl[3] = {p3, p4, p5};
    for (int l = 0; l < 3; ++l) {
        lv.feat[l] = reinterpret_cast<const half_t*>(pl[l]);
        lv.h[l] = height >> (3 + l);
        lv.w[l] = width >> (3 + l);
        lv.scale[l] = 1.f / (float)(8 << l);
    }
    TRY(dvid_roialign_launch(lv, channels, boxes, n_frames, boxes_per_frame, reinterpret_cast<half_t*>(roi_out), mean_out,
                             reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_roialign_v2_multilevel_f32(const float* p3, const float* p4, const float* p5, int n_frames, int height, int width, int channels,
                                    const float* boxes, int boxes_per_frame, float* roi_out, float* mean_out, void* stream) {
    g_err[0] = 0;
    if (height % 32 || width % 32) FAIL(DVID_ERR_ARG, "height/width must be multiples of 32");
    RoiLevels32 lv;
    const float* pl[3] = {p3, p4, p5};
    for (int l = 0; l < 3; ++l) {
        lv.feat[l] = pl[l];
        lv.h[l] = height >> (3 + l);
        lv.w[l] = width >> (3 + l);
        lv.scale[l] = 1.f / (float)(8 << l);
    }
    TRY(dvid_f32_roialign_launch(lv, channels, boxes, n_frames, boxes_per_frame, roi_out, mean_out, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_conv2d_nhwc_f32(const float* in, const float* w, const void* w_hi, const void* w_lo, const float* bias, const float* row_scale, const float* residual,
                         float* out, int n, int h, int wd, int cin, int cout, int kh, int kw, int stride, int pad, int kpad, int relu, int residual_mode,
                         void* stream) {
    g_err[0] = 0;
    if (cin % 4 || kpad % 16 || kpad < kh * kw * cin || pad < 0) FAIL(DVID_ERR_ARG, "fp32 conv: cin %% 4 == 0, kpad %% 16 == 0, kpad >= kh*kw*cin, pad >= 0");
    ConvW cw;
    cw.w32 = const_cast<float*>(w);
    cw.w16hi = reinterpret_cast<half_t*>(const_cast<void*>(w_hi));
    cw.w16lo = reinterpret_cast<half_t*>(const_cast<void*>(w_lo));
    cw.bias = const_cast<float*>(bias);
    cw.wscale32 = const_cast<float*>(row_scale);
    cw.cin32 = cin;
    cw.cin_real = cin;
    cw.cout = cout;
    cw.kh = kh;
    cw.kw = kw;
    cw.stride = stride;
    cw.pad = pad;
    cw.kpad32 = kpad;
    TRY(conv_run32(cw, in, n, h, wd, out, relu, residual, residual_mode, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_mha_f32(const float* q, const float* k, const float* v, float* out, int batch, int lq, int lk, int nheads, int q_ld, int kv_ld, int out_ld,
                 int64_t q_bs, int64_t kv_bs, int64_t out_bs, void* stream) {
    g_err[0] = 0;
    TRY(dvid_f32_mha_launch(q, k, v, out, batch, lq, lk, nheads, q_ld, kv_ld, out_ld, (long)q_bs, (long)kv_bs, (long)out_bs,
                            reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_dynconv_f32(const float* roi, const float* params, const float* g1, const float* b1, const float* g2, const float* b2, float* out,
                     int rows, void* stream) {
    g_err[0] = 0;
    TRY(dvid_f32_dynconv_launch(roi, params, g1, b1, g2, b2, out, rows, nullptr, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_select_topk_features(const float* logits, int n_frames, int mm, int num_classes, int k1, int k2, const float* feats,
                              int hidden, float* out_k1, float* out_k2, void* stream) {
    g_err[0] = 0;
    TRY(dvid_topk_mask_launch(logits, n_frames, mm, num_classes, k1, k2, feats, hidden, out_k1, out_k2,
                              reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_counter_normal(float* out, int64_t per_image, int n_images, uint64_t key0, void* stream) {
    g_err[0] = 0;
    if (!out || per_image < 0 || n_images < 0 || n_images > 65535) FAIL(DVID_ERR_ARG, "dvid_counter_normal: bad arguments");
    TRY(dvid_counter_normal_launch(out, (long)per_image, n_images, key0, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_noise_to_boxes(const float* x, float* boxes, int n, float snr_scale, float img_w, float img_h, void* stream) {
    g_err[0] = 0;
    TRY(dvid_noise_to_boxes_launch(x, boxes, n, snr_scale, img_w, img_h, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_ddim_renew_step(const float* logits, const float* boxes, const float* x_t, const float* noise, const float* fresh,
                         float* x_next, int n_frames, int mm, int c, float img_w, float img_h, float snr_scale,
                         float sqrt_recip_ac, float sqrt_recipm1_ac, float sqrt_ac_next, float coef_c, float sigma, float keep_thr,
                         void* stream) {
    g_err[0] = 0;
    TRY(dvid_ddim_renew_launch(logits, boxes, x_t, noise, fresh, x_next, n_frames, mm, c, img_w, img_h, snr_scale, sqrt_recip_ac,
                               sqrt_recipm1_ac, sqrt_ac_next, coef_c, sigma, keep_thr, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_postproc_topk_nms(const float* logits, const float* boxes, int nsets, int n_frames, int mm, int c, float img_w, float img_h,
                           float iou_threshold, int use_nms, float* out_boxes, float* out_scores, int* out_labels, int* out_counts,
                           void* scratch, void* stream) {
    g_err[0] = 0;
    if (!scratch) FAIL(DVID_ERR_ARG, "scratch required");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t ncand = (size_t)n_frames * nsets * mm;
    float* cb = reinterpret_cast<float*>(scratch);
    float* cs = cb + ncand * 4;
    int* cl = reinterpret_cast<int*>(cs + ncand);
    TRY(dvid_topk_candidates_launch(logits, boxes, n_frames, nsets, mm, c, cb, cs, cl, s));
    TRY(dvid_nms_frames_launch(cb, cs, cl, n_frames, nsets * mm, img_w, img_h, iou_threshold, use_nms, nsets * mm, out_boxes,
                               out_scores, out_labels, out_counts, s));
    return DVID_OK;
}

int dvid_cdist(const float* x, int n, int d, float* dist, void* stream) {
    g_err[0] = 0;
    TRY(dvid_cdist_launch(x, n, d, dist, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}
int dvid_fps_greedy(const float* dist, int n, int mm, int bs_emul, int* idx, void* stream) {
    g_err[0] = 0;
    TRY(dvid_fps_launch(dist, n, mm, bs_emul, idx, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}
int dvid_gather_rows(const float* x, const int* idx, float* y, int mm, int d, void* stream) {
    g_err[0] = 0;
    TRY(dvid_gather_rows_launch(x, idx, y, mm, d, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_conv2d_nhwc_f16(const void* in, const void* w, const float* bias, const void* residual, void* out, int n, int h, int wd,
                         int cin, int cout, int kh, int kw, int stride, int pad, int kpad, int relu, int out_f32, int residual_mode,
                         void* stream) {
    g_err[0] = 0;
    ConvW cw;
    cw.w = reinterpret_cast<half_t*>(const_cast<void*>(w));
    cw.bias = const_cast<float*>(bias);
    cw.cin = cin;
    cw.cout = cout;
    cw.kh = kh;
    cw.kw = kw;
    cw.stride = stride;
    cw.pad = pad < 0 ? -pad : pad;
    cw.same_size = pad < 0;          // pad < 0: |pad| before, as many after as keep the output at the input's size (stride 1)
    cw.kpad = kpad;
    if (pad < 0 && stride != 1) FAIL(DVID_ERR_ARG, "same-size padding needs stride 1");
    TRY(conv_run(cw, reinterpret_cast<const half_t*>(in), n, h, wd, out, relu, out_f32, residual, residual_mode, 0,
                 reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_bottleneck64_tail_f16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* residual,
                               const void* w_shortcut, const float* b_shortcut, const void* w1_next, const float* b1_next, int next_channels,
                               void* out, void* t1_next, int n, int h, int wd, void* stream) {
    g_err[0] = 0;
    const int rc = bneck_tail(reinterpret_cast<const half_t*>(t1), reinterpret_cast<const half_t*>(w2), b2, reinterpret_cast<const half_t*>(w3),
                              b3, reinterpret_cast<const half_t*>(residual), reinterpret_cast<const half_t*>(w_shortcut), b_shortcut,
                              reinterpret_cast<const half_t*>(w1_next), b1_next, next_channels, reinterpret_cast<half_t*>(out),
                              reinterpret_cast<half_t*>(t1_next), n, h, wd, reinterpret_cast<hipStream_t>(stream));
    if (rc != DVID_OK) FAIL(rc, "bottleneck tail: bad argument (n %d, %d x %d, next conv1 with %d channels)", n, h, wd, next_channels);
    return DVID_OK;
}

int dvid_bottleneck128_tail_f16(const void* t1, const void* w2, const float* b2, const void* w3, const float* b3, const void* residual,
                                const void* w1_next, const float* b1_next, void* out, void* t1_next, int n, int h, int wd, void* stream) {
    g_err[0] = 0;
    const int rc = bneck128_tail(reinterpret_cast<const half_t*>(t1), reinterpret_cast<const half_t*>(w2), b2, reinterpret_cast<const half_t*>(w3),
                                 b3, reinterpret_cast<const half_t*>(residual), reinterpret_cast<const half_t*>(w1_next), b1_next,
                                 reinterpret_cast<half_t*>(out), reinterpret_cast<half_t*>(t1_next), n, h, wd,
                                 reinterpret_cast<hipStream_t>(stream));
    if (rc != DVID_OK) FAIL(rc, "bottleneck tail (128): bad argument (n %d, %d x %d)", n, h, wd);
    return DVID_OK;
}

int dvid_mha_f16(const void* q, const void* k, const void* v, void* out, void* vt_scratch, int batch, int lq, int lk, int nheads,
                 int q_ld, int kv_ld, int out_ld, int64_t q_bs, int64_t kv_bs, int64_t out_bs, void* stream) {
    g_err[0] = 0;
    TRY(dvid_mha_mfma_launch(reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k), reinterpret_cast<const half_t*>(v),
                             reinterpret_cast<half_t*>(out), reinterpret_cast<half_t*>(vt_scratch), batch, lq, lk, nheads, q_ld, kv_ld,
                             out_ld, q_bs, kv_bs, out_bs, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_dynconv(const void* roi, const void* params, const float* g1, const float* b1, const float* g2, const float* b2, void* out,
                 int rows, void* stream) {
    g_err[0] = 0;
    TRY(dvid_dynconv_launch(reinterpret_cast<const half_t*>(roi), reinterpret_cast<const half_t*>(params), g1, b1, g2, b2,
                            reinterpret_cast<half_t*>(out), rows, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_add_layernorm(const float* x, const float* r, const float* g, const float* b, float* y, int rows, int d, int relu,
                       void* stream) {
    g_err[0] = 0;
    TRY(dvid_add_layernorm_launch(x, r, g, b, y, nullptr, rows, d, relu, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_nhwc_from_nchw(const float* in, void* out_f16, int n, int h, int w, int c, void* stream) {
    g_err[0] = 0;
    TRY(dvid_nhwc_from_nchw_launch(in, reinterpret_cast<half_t*>(out_f16), n, h, w, c, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}
int dvid_nchw_from_nhwc(const void* in_f16, float* out, int n, int h, int w, int c, void* stream) {
    g_err[0] = 0;
    TRY(dvid_nchw_from_nhwc_launch(reinterpret_cast<const half_t*>(in_f16), out, n, h, w, c, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}
int dvid_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
    g_err[0] = 0;
    TRY(dvid_f32_to_f16_launch(x, reinterpret_cast<half_t*>(y), (long)n, reinterpret_cast<hipStream_t>(stream)));
    return DVID_OK;
}

int dvid_resize_u8_to_f32(const void* src_hwc, int h, int w, void* tmp, float* out_chw, int oh, int ow, int ph, int pw,
                          const int* xbounds, const int* xk, int xksize, const int* ybounds, const int* yk, int yksize, void* stream) {
    g_err[0] = 0;
    if (!src_hwc || !out_chw) FAIL(DVID_ERR_ARG, "null image");
    const int rc = dvid_resize_u8_launch(reinterpret_cast<const unsigned char*>(src_hwc), h, w, reinterpret_cast<unsigned char*>(tmp),
                                         out_chw, oh, ow, ph, pw, xbounds, xk, xksize, ybounds, yk, yksize,
                                         reinterpret_cast<hipStream_t>(stream));
    if (rc != DVID_OK) FAIL(rc, "resize %dx%d -> %dx%d (padded %dx%d): bad sizes or missing tables / scratch", h, w, oh, ow, ph, pw);
    return DVID_OK;
}

// ---- measurement -------------------------------------------------------------------------------
int dvid_profile_enable(int on) {
    g_prof_on = on != 0;
    return DVID_OK;
}
int dvid_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) g_prof_pool.push_back(r);
    g_prof.clear();
    return DVID_OK;
}
int dvid_profile_read_bytes(double* igemm_alg_bytes) {
    g_err[0] = 0;
    double b = 0;
    for (auto& r : g_prof)
        if (r.family) b += r.bytes;
    if (igemm_alg_bytes) *igemm_alg_bytes = b;
    return DVID_OK;
}

static int profile_sum(double* ms_out, double* flop_out, double* bytes_out, int64_t* n_out) {
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (!r.family) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        fl += r.flop;
        by += r.bytes;
        ++n;
    }
    if (ms_out) *ms_out = ms;
    if (flop_out) *flop_out = fl;
    if (bytes_out) *bytes_out = by;
    if (n_out) *n_out = n;
    return DVID_OK;
}

int dvid_profile_read(double* igemm_ms, double* igemm_flop, int64_t* igemm_launches) {
    g_err[0] = 0;
    return profile_sum(igemm_ms, igemm_flop, nullptr, igemm_launches);
}

// one CSV line per recorded igemm launch: M,N,K,taps,stride,res_mode,ms,tflops
int dvid_profile_dump(const char* path) {
    g_err[0] = 0;
    FILE* f = fopen(path, "w");
    if (!f) FAIL(DVID_ERR_ARG, "cannot open %s", path);
    fprintf(f, "kernel,family,M,N,K,taps,stride,res_mode,ms,tflops,alg_mbytes,alg_gbs\n");
    for (auto& r : g_prof) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%d,%.5f,%.2f,%.3f,%.1f\n", r.kind, (int)r.family, r.M, r.N, r.K, r.taps, r.stride, r.res_mode, t, r.flop / (t * 1e-3) / 1e12,
                r.bytes / 1e6, r.bytes / (t * 1e-3) / 1e9);
    }
    fclose(f);
    return DVID_OK;
}

}  // extern "C"
