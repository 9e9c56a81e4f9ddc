// The detector: weight preparation (BatchNorm folding, GEMM packing) and the forward pass of
// faster_rcnn50 (reference: code/detection/symbol/faster_rcnn.py:107-179) as a fixed sequence
// of kernel launches on one stream.  Nothing in the forward pass synchronises with the host:
// candidate / keep counts stay on the device and every grid is sized from shape bounds.
#include "fd_boxes.h"
#include "fd_conv.h"
#include "fd_conv_split.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace fd {

constexpr float BN_EPS = 2e-5f;
constexpr int UNITS[4] = {3, 4, 6, 3};
constexpr int FILTERS[5] = {64, 256, 512, 1024, 2048};
constexpr int POOLED = 7;

struct Unit {
    ConvParams conv1, conv2, conv3, sc;
    SplitConvParams s_conv1, s_conv2, s_conv3, s_sc;   // split (f16x3) operands
    SplitConvParams f_conv1, f_conv2, f_conv3, f_sc;   // float32 operands on the same kernels
    WinoConvParams w_conv2;                            // conv2 in Winograd form (wide stride-1 layers only; both precisions)
    const float* tailw_u = nullptr;                    // 64-channel units: conv2's F(2x2,3x3) weights for the fused float32 tail
    const float* tailw_w3sc = nullptr;                 // ... of the stride-1 projection unit: [cout][conv3's 64 | shortcut's 64] weights
    const float* c3sc_w = nullptr;                     // projection units of the later stages: [cout][conv3's mid | shortcut's cin] weights
    const float* pair_w = nullptr;                     // stride-1 projection units: shortcut rows + conv1 rows, one float32 GEMM
    size_t pair_w_floats = 0;
    const float *bn1_scale = nullptr, *bn1_shift = nullptr;   // the unit's input BatchNorm (device)
    bool has_sc = false;
};

struct Tap {
    std::string name;
    const float* ptr;
    int64_t dims[4];
};

enum NetBuf {
    NB_STAGED = 0, NB_CONV0, NB_XA, NB_XB, NB_T1, NB_T2, NB_SC, NB_RPN_HID, NB_RPN_OUT, NB_ROIS, NB_ROI_SCORES,
    NB_POOLED, NB_X4A, NB_X4B, NB_FLAT, NB_CLS, NB_BOX, NB_COUNTS, NB_ACT_A, NB_ACT_B, NB_SPLITK, NB_WINO_V, NB_WINO_M, NB_COUNT
};

}  // namespace fd

struct fd_detector {
    fd_ctx* ctx = nullptr;
    fd_proposal_cfg prop;
    std::vector<void*> owned;
    fd::InputNorm norm;
    fd::ConvParams conv0;
    std::vector<fd::Unit> stage[4];
    fd::ConvParams rpn_conv, rpn_out;
    fd::SplitConvParams s_conv0, s_rpn_conv, s_rpn_out;
    fd::SplitConvParams f_rpn_conv, f_rpn_out, f_conv0;
    fd::WinoConvParams w_rpn_conv;       // the RPN's 3x3 convolution in Winograd form (both precisions)
    int winograd = 4;                    // Winograd tile of the wide 3x3 layers, 4 = F(4x4,3x3),
                                         // 2 = F(2x2,3x3), 0 = direct (environment variable FD_WINOGRAD)
    bool winograd_split = true;          // f16x3 precision: F(4x4,3x3) for the RPN and the 256-channel stage when
                                         // `winograd` is 4 (environment variable FD_WINOGRAD_SPLIT=0: direct)
    int precision = FD_PRECISION_F32;      // the reference's arithmetic unless the caller opts into another form
    int b3_balance = 63;                   // FD_PRECISION_BF16X3: see SplitConvOut (FD_B3_BALANCE overrides; 0 = off)
    const float *bn1_scale = nullptr, *bn1_shift = nullptr;
    const float *w_cls = nullptr, *b_cls = nullptr, *w_box = nullptr, *b_box = nullptr;
    int32_t* nan_flag = nullptr;         // status block on the device: [0] NaN met by the proposal stage,
    unsigned* clamped = nullptr;         // [1] values outside the fp16 range met while splitting (this pass)
    std::vector<fd::Tap> taps;
    fd::Buf net[fd::NB_COUNT];
    int keep_mode = 0;                   // snapshot tapped tensors: 0 no; 1 every named tensor, which takes the
                                         // layer-by-layer kernels (the fused ones do not materialise them all);
                                         // 2 the tensors the production kernels do materialise, kernel choice unchanged
    bool keeping() const { return keep_mode != 0; }
    bool layerwise() const { return keep_mode == 1; }
    int fuse_units = 3;                  // whole-unit kernels of the 64-channel stage: bit 0 identity units, bit 1 the
                                         // projection unit, bit 2 also while tensors are kept for inspection (the
                                         // unit outputs then come from the fused kernels; t1/t2 are not produced);
                                         // the environment variable FD_FUSE_UNITS overrides
    std::vector<void*> tap_copies;
};

namespace fd {

// ----------------------------------------------------------------------------- weights
struct ParamTable {
    std::map<std::string, const fd_param*> m;
    const fd_param* get(const std::string& name, int64_t numel) const {
        auto it = m.find(name);
        if (it == m.end()) {
            set_error("missing parameter '%s'", name.c_str());
            return nullptr;
        }
        if (it->second->numel != numel || !it->second->data_host) {
            set_error("parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second->numel,
                      (long long)numel);
            return nullptr;
        }
        // A NaN or Inf weight would make every score NaN (the reference then raises ValueError in its
        // proposal stage on every frame); the split format cannot even carry it.  Refuse it here.
        const float* v = it->second->data_host;
        for (int64_t i = 0; i < numel; ++i)
            if (!std::isfinite(v[i])) {
                set_error("parameter '%s' holds a non-finite value at element %lld", name.c_str(), (long long)i);
                return nullptr;
            }
        return it->second;
    }
};

static int upload(fd_detector* d, const float* host, size_t n, float** out) {
    void* p = nullptr;
    FD_HIP(hipMalloc(&p, n * sizeof(float)));
    d->owned.push_back(p);
    FD_HIP(hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<float*>(p);
    return FD_OK;
}

// Split-precision form of a Winograd layer's transformed weights (F(4x4,3x3) only): the float32 set `u_dev`
// (already on the device) scaled by a power of two that brings the largest magnitude to [2^13, 2^14), as for
// ordinary convolution weights, and packed per GEMM.
static int make_wino_split(fd_detector* d, const std::vector<float>& u_host, const float* u_dev, WinoConvParams* p) {
    if (p->tile != 4 || p->cout % 32 != 0) return FD_OK;
    double umax = 0.0;
    for (float v : u_host) umax = std::max(umax, std::fabs((double)v));
    float wscale = 1.f;
    if (umax > 0.0 && std::isfinite(umax)) wscale = (float)std::ldexp(1.0, 13 - (int)std::floor(std::log2(umax)));
    const size_t set_floats = wino_split_weight_floats(p->cin, p->cout);
    void* packed = nullptr;
    FD_HIP(hipMalloc(&packed, 36 * set_floats * sizeof(float)));
    d->owned.push_back(packed);
    FD_TRY(wino_pack_weights_split(u_dev, p->cin, p->cout, p->cout_pad, wscale, packed, nullptr));
    FD_HIP(hipDeviceSynchronize());
    p->u_split = packed;
    p->u_inv_scale = 1.0f / wscale;
    p->u_split_bytes = (unsigned)(set_floats * sizeof(float));
    return FD_OK;
}

// Inference BatchNorm as y = x * scale + shift:  scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
static int bn_fold(const ParamTable& t, const std::string& name, int ch, bool fix_gamma, std::vector<float>& scale,
                   std::vector<float>& shift) {
    const fd_param* g = t.get(name + "_gamma", ch);
    const fd_param* b = t.get(name + "_beta", ch);
    const fd_param* mu = t.get(name + "_moving_mean", ch);
    const fd_param* var = t.get(name + "_moving_var", ch);
    if (!g || !b || !mu || !var) return FD_ERR_PARAM;
    scale.resize(ch);
    shift.resize(ch);
    for (int c = 0; c < ch; ++c) {
        const double s = (fix_gamma ? 1.0 : (double)g->data_host[c]) / std::sqrt((double)var->data_host[c] + (double)BN_EPS);
        scale[c] = (float)s;
        shift[c] = (float)((double)b->data_host[c] - (double)mu->data_host[c] * s);
    }
    return FD_OK;
}

// Uploads OIHW weights, packs them for the GEMM (scaled per output channel when out_scale is given).
static int make_conv(fd_detector* d, const float* w_host, int cin_w, ConvShape s, const std::vector<float>* out_scale,
                     const std::vector<float>* bias, int relu, ConvParams* out, SplitConvParams* sout,
                     SplitConvParams* fout = nullptr) {
    float *raw = nullptr, *scale_d = nullptr, *packed = nullptr, *bias_d = nullptr;
    const size_t nraw = (size_t)s.cout * cin_w * s.kh * s.kw;
    FD_HIP(hipMalloc((void**)&raw, nraw * sizeof(float)));
    FD_HIP(hipMemcpy(raw, w_host, nraw * sizeof(float), hipMemcpyHostToDevice));
    if (out_scale) {
        FD_HIP(hipMalloc((void**)&scale_d, s.cout * sizeof(float)));
        FD_HIP(hipMemcpy(scale_d, out_scale->data(), s.cout * sizeof(float), hipMemcpyHostToDevice));
    }
    FD_HIP(hipMalloc((void**)&packed, s.packed_floats() * sizeof(float)));
    d->owned.push_back(packed);
    FD_TRY(pack_conv_weights(raw, cin_w, s, scale_d, packed, nullptr));
    // split form: weights (with the folded BatchNorm scale) times a power of two that brings the
    // largest magnitude to [2^13, 2^14), so that their lo halves stay clear of fp16 subnormals
    void* packed_split = nullptr;
    float wscale = 1.f;
    {
        const size_t per_out = (size_t)cin_w * s.kh * s.kw;
        double wmax = 0.0;
        for (int o = 0; o < s.cout; ++o) {
            const double sc = out_scale ? std::fabs((double)(*out_scale)[o]) : 1.0;
            for (size_t i = 0; i < per_out; ++i) {
                const double v = std::fabs((double)w_host[o * per_out + i]) * sc;
                if (v > wmax) wmax = v;
            }
        }
        if (wmax > 0.0 && std::isfinite(wmax)) wscale = (float)std::ldexp(1.0, 13 - (int)std::floor(std::log2(wmax)));
        FD_HIP(hipMalloc(&packed_split, s.packed_floats() * sizeof(float)));
        d->owned.push_back(packed_split);
        FD_TRY(pack_conv_weights_split(raw, cin_w, s, scale_d, wscale, packed_split, nullptr));
    }
    FD_HIP(hipDeviceSynchronize());
    FD_HIP(hipFree(raw));
    if (scale_d) FD_HIP(hipFree(scale_d));
    if (bias) FD_TRY(upload(d, bias->data(), bias->size(), &bias_d));
    out->s = s;
    out->w = packed;
    out->bias = bias_d;
    out->relu = relu;
    sout->s = s;
    sout->w = packed_split;
    sout->w_inv_scale = 1.0f / wscale;
    sout->bias = bias_d;
    sout->relu = relu;
    if (fout) {
        fout->s = s;
        fout->w = packed;
        fout->w_inv_scale = 1.0f;
        fout->bias = bias_d;
        fout->relu = relu;
        fout->f32 = true;
    }
    return FD_OK;
}

static ConvShape shape(int cin, int cout, int k, int stride, int pad) {
    ConvShape s;
    s.cin = cin; s.cout = cout; s.kh = k; s.kw = k; s.kwp = k; s.stride = stride; s.pad = pad;
    return s;
}

static int build_unit(fd_detector* d, const ParamTable& t, const std::string& name, int cin, int cout, int stride,
                      bool dim_match, Unit* u) {
    const int mid = cout / 4;
    std::vector<float> s1, h1, s2, h2, s3, h3;
    FD_TRY(bn_fold(t, name + "_bn1", cin, false, s1, h1));
    FD_TRY(bn_fold(t, name + "_bn2", mid, false, s2, h2));
    FD_TRY(bn_fold(t, name + "_bn3", mid, false, s3, h3));
    const fd_param* w1 = t.get(name + "_conv1_weight", (int64_t)mid * cin);
    const fd_param* w2 = t.get(name + "_conv2_weight", (int64_t)mid * mid * 9);
    const fd_param* w3 = t.get(name + "_conv3_weight", (int64_t)cout * mid);
    if (!w1 || !w2 || !w3) return FD_ERR_PARAM;
    // bn1 + relu run inside conv1's (and the shortcut's) input gather; bn2 / bn3 fold into the
    // convolution that feeds them: scale into the weights, shift into the bias, relu in the epilogue
    float *pre_s = nullptr, *pre_h = nullptr;
    FD_TRY(upload(d, s1.data(), s1.size(), &pre_s));
    FD_TRY(upload(d, h1.data(), h1.size(), &pre_h));
    FD_TRY(make_conv(d, w1->data_host, cin, shape(cin, mid, 1, 1, 0), &s2, &h2, 1, &u->conv1, &u->s_conv1, &u->f_conv1));
    u->conv1.pre_scale = pre_s;
    u->conv1.pre_shift = pre_h;
    u->bn1_scale = pre_s;
    u->bn1_shift = pre_h;
    FD_TRY(make_conv(d, w2->data_host, mid, shape(mid, mid, 3, stride, 1), &s3, &h3, 1, &u->conv2, &u->s_conv2, &u->f_conv2));
    FD_TRY(make_conv(d, w3->data_host, mid, shape(mid, cout, 1, 1, 0), nullptr, nullptr, 0, &u->conv3, &u->s_conv3, &u->f_conv3));
    // Winograd form of the 3x3 where it pays.  With F(4x4,3x3) (a quarter of the multiplies, Winograd-domain
    // tensors 2.25x the plain ones) that is from 128 channels on: the 128- and 256-channel stages and the
    // stride-1 units of the ROI head (1080p, float32: 376 -> 413 frames/s with the 256-channel stage and the RPN,
    // 427 with the 128-channel stage as well); with F(2x2,3x3) (4/9 of the multiplies, tensors 4x) the
    // 128-channel stage would lose.  At 64 channels the GEMMs are two K-steps long and the transforms cost more
    // than the multiplies saved.  FD_WINO_MIN_MID overrides the channel count (measurements).
    const int wino_min_mid = getenv("FD_WINO_MIN_MID") ? atoi(getenv("FD_WINO_MIN_MID")) : (d->winograd == 4 ? 128 : 256);
    if (stride == 1 && mid >= wino_min_mid) {
        std::vector<float> uw;
        int cout_pad = 0;
        FD_TRY(wino_pack_weights(w2->data_host, mid, mid, s3.data(), d->winograd ? d->winograd : 2, &uw, &cout_pad));
        float* ud = nullptr;
        FD_TRY(upload(d, uw.data(), uw.size(), &ud));
        u->w_conv2.cin = mid; u->w_conv2.cout = mid; u->w_conv2.cout_pad = cout_pad; u->w_conv2.tile = d->winograd ? d->winograd : 2;
        u->w_conv2.u = ud; u->w_conv2.bias = u->f_conv2.bias; u->w_conv2.relu = 1;
        if (mid >= 256 && d->winograd_split) FD_TRY(make_wino_split(d, uw, ud, &u->w_conv2));
    }
    if (stride == 1 && mid == 64) {
        std::vector<float> uw;
        FD_TRY(tail_wino_pack(w2->data_host, s3.data(), &uw));
        float* ud = nullptr;
        FD_TRY(upload(d, uw.data(), uw.size(), &ud));
        u->tailw_u = ud;
    }
    u->has_sc = !dim_match;
    if (u->has_sc) {
        const fd_param* ws = t.get(name + "_sc_weight", (int64_t)cout * cin);
        if (!ws) return FD_ERR_PARAM;
        FD_TRY(make_conv(d, ws->data_host, cin, shape(cin, cout, 1, stride, 0), nullptr, nullptr, 0, &u->sc, &u->s_sc, &u->f_sc));
        u->sc.pre_scale = pre_s;
        u->sc.pre_shift = pre_h;
        if (stride == 1 && mid == 64 && cin == 64 && u->tailw_u && u->f_sc.w && u->f_conv3.w && !u->f_sc.bias) {
            // the shortcut inside the float32 tail: conv3's and the shortcut's weight rows side by side (K = 64 + 64)
            float* pw = nullptr;
            FD_HIP(hipMalloc((void**)&pw, (size_t)cout * 128 * sizeof(float)));
            d->owned.push_back(pw);
            FD_HIP(hipMemcpy2D(pw, 128 * sizeof(float), u->f_conv3.w, 64 * sizeof(float), 64 * sizeof(float), cout, hipMemcpyDeviceToDevice));
            FD_HIP(hipMemcpy2D(pw + 64, 128 * sizeof(float), u->f_sc.w, 64 * sizeof(float), 64 * sizeof(float), cout, hipMemcpyDeviceToDevice));
            u->tailw_w3sc = pw;
        }
        if (mid >= 128 && mid <= 256 && u->f_sc.w && u->f_conv3.w && !u->f_sc.bias) {
            // conv3 and the shortcut as one two-source GEMM (float32): the weight rows side by side, K = mid + cin
            const size_t kc = (size_t)mid + cin;
            float* pw = nullptr;
            FD_HIP(hipMalloc((void**)&pw, (size_t)cout * kc * sizeof(float)));
            d->owned.push_back(pw);
            FD_HIP(hipMemcpy2D(pw, kc * sizeof(float), u->f_conv3.w, mid * sizeof(float), mid * sizeof(float), cout, hipMemcpyDeviceToDevice));
            FD_HIP(hipMemcpy2D(pw + mid, kc * sizeof(float), u->f_sc.w, cin * sizeof(float), cin * sizeof(float), cout, hipMemcpyDeviceToDevice));
            u->c3sc_w = pw;
        }
        if (stride == 1 && mid <= 64 && cout % 64 == 0 && u->f_sc.w && u->f_conv1.w) {
            // the shortcut and conv1 read the same tensor: their float32 weight rows in one buffer, conv1's padded to 64
            const size_t k = (size_t)cin;
            float* pw = nullptr;
            FD_HIP(hipMalloc((void**)&pw, ((size_t)cout + 64) * k * sizeof(float)));
            d->owned.push_back(pw);
            FD_HIP(hipMemset(pw, 0, ((size_t)cout + 64) * k * sizeof(float)));
            FD_HIP(hipMemcpy(pw, u->f_sc.w, (size_t)cout * k * sizeof(float), hipMemcpyDeviceToDevice));
            FD_HIP(hipMemcpy(pw + (size_t)cout * k, u->f_conv1.w, (size_t)mid * k * sizeof(float), hipMemcpyDeviceToDevice));
            u->pair_w = pw;
            u->pair_w_floats = ((size_t)cout + 64) * k;
        }
    }
    return FD_OK;
}

static int build(fd_detector* d, const ParamTable& t) {
    std::vector<float> sc, sh;
    // bn_data (gamma fixed to 1) is applied while the uint8 frame is widened to float
    FD_TRY(bn_fold(t, "bn_data", 3, true, sc, sh));
    for (int c = 0; c < 3; ++c) { d->norm.scale[c] = sc[c]; d->norm.shift[c] = sh[c]; }
    // conv0 7x7/2 on the 4-channel staged image, kernel width padded to 8; bn0 + relu folded in
    FD_TRY(bn_fold(t, "bn0", FILTERS[0], false, sc, sh));
    const fd_param* w0 = t.get("conv0_weight", (int64_t)FILTERS[0] * 3 * 49);
    if (!w0) return FD_ERR_PARAM;
    ConvShape s0 = shape(4, FILTERS[0], 7, 2, 3);
    s0.kwp = 8;
    FD_TRY(make_conv(d, w0->data_host, 3, s0, &sc, &sh, 1, &d->conv0, &d->s_conv0));
    d->f_conv0.s = s0; d->f_conv0.w = d->conv0.w; d->f_conv0.w_inv_scale = 1.f; d->f_conv0.bias = d->conv0.bias;
    d->f_conv0.relu = 1; d->f_conv0.f32 = true;
    int cin = FILTERS[0];
    for (int s = 0; s < 4; ++s) {
        const int cout = FILTERS[s + 1];
        d->stage[s].resize(UNITS[s]);
        for (int u = 0; u < UNITS[s]; ++u) {
            char name[64];
            snprintf(name, sizeof(name), "stage%d_unit%d", s + 1, u + 1);
            FD_TRY(build_unit(d, t, name, cin, cout, (u == 0 && s > 0) ? 2 : 1, u != 0, &d->stage[s][u]));
            cin = cout;
        }
    }
    // RPN: 3x3 conv + relu, then the two 1x1 heads fused into one 72-row GEMM (24 scores | 48 deltas)
    {
        const fd_param* w = t.get("rpn_conv_3x3_weight", 512ll * 1024 * 9);
        const fd_param* b = t.get("rpn_conv_3x3_bias", 512);
        if (!w || !b) return FD_ERR_PARAM;
        std::vector<float> bias(b->data_host, b->data_host + 512);
        FD_TRY(make_conv(d, w->data_host, 1024, shape(1024, 512, 3, 1, 1), nullptr, &bias, 1, &d->rpn_conv, &d->s_rpn_conv, &d->f_rpn_conv));
        {
            std::vector<float> u;
            int cout_pad = 0;
            FD_TRY(wino_pack_weights(w->data_host, 512, 1024, nullptr, d->winograd ? d->winograd : 2, &u, &cout_pad));
            float* ud = nullptr;
            FD_TRY(upload(d, u.data(), u.size(), &ud));
            d->w_rpn_conv.cin = 1024; d->w_rpn_conv.cout = 512; d->w_rpn_conv.cout_pad = cout_pad;
            d->w_rpn_conv.tile = d->winograd ? d->winograd : 2;
            d->w_rpn_conv.u = ud; d->w_rpn_conv.bias = d->f_rpn_conv.bias; d->w_rpn_conv.relu = 1;
            if (d->winograd_split) FD_TRY(make_wino_split(d, u, ud, &d->w_rpn_conv));
        }
        const int A = d->prop.num_ratios * d->prop.num_scales;
        const fd_param* wc = t.get("rpn_cls_score_weight", 2ll * A * 512);
        const fd_param* bc = t.get("rpn_cls_score_bias", 2 * A);
        const fd_param* wb = t.get("rpn_bbox_pred_weight", 4ll * A * 512);
        const fd_param* bb = t.get("rpn_bbox_pred_bias", 4 * A);
        if (!wc || !bc || !wb || !bb) return FD_ERR_PARAM;
        std::vector<float> wcat(6ull * A * 512), bcat(6 * A);
        memcpy(wcat.data(), wc->data_host, sizeof(float) * 2 * A * 512);
        memcpy(wcat.data() + 2ull * A * 512, wb->data_host, sizeof(float) * 4 * A * 512);
        memcpy(bcat.data(), bc->data_host, sizeof(float) * 2 * A);
        memcpy(bcat.data() + 2 * A, bb->data_host, sizeof(float) * 4 * A);
        FD_TRY(make_conv(d, wcat.data(), 512, shape(512, 6 * A, 1, 1, 0), nullptr, &bcat, 0, &d->rpn_out, &d->s_rpn_out, &d->f_rpn_out));
    }
    // head: bn1 + relu + global average, then the two FullyConnected layers
    FD_TRY(bn_fold(t, "bn1", FILTERS[4], false, sc, sh));
    float* p = nullptr;
    FD_TRY(upload(d, sc.data(), sc.size(), &p)); d->bn1_scale = p;
    FD_TRY(upload(d, sh.data(), sh.size(), &p)); d->bn1_shift = p;
    const fd_param* wcls = t.get("cls_score_weight", 2ll * FILTERS[4]);
    const fd_param* bcls = t.get("cls_score_bias", 2);
    const fd_param* wbox = t.get("bbox_pred_weight", 8ll * FILTERS[4]);
    const fd_param* bbox = t.get("bbox_pred_bias", 8);
    if (!wcls || !bcls || !wbox || !bbox) return FD_ERR_PARAM;
    FD_TRY(upload(d, wcls->data_host, 2 * FILTERS[4], &p)); d->w_cls = p;
    FD_TRY(upload(d, bcls->data_host, 2, &p)); d->b_cls = p;
    FD_TRY(upload(d, wbox->data_host, 8 * FILTERS[4], &p)); d->w_box = p;
    FD_TRY(upload(d, bbox->data_host, 8, &p)); d->b_box = p;
    void* flag = nullptr;
    FD_HIP(hipMalloc(&flag, 256));
    FD_HIP(hipMemset(flag, 0, 256));
    d->owned.push_back(flag);
    d->nan_flag = static_cast<int32_t*>(flag);
    d->clamped = reinterpret_cast<unsigned*>(d->nan_flag + 1);
    return FD_OK;
}

// ----------------------------------------------------------------------------- forward
struct Plan {
    fd_detector* d;
    bool dry;
    bool f32 = false;          // float32 operands (FD_PRECISION_F32) on the graph of run_graph_split
    size_t need[NB_COUNT];
    hipStream_t stream;

    float* buf(NetBuf id, size_t nfloats) {
        if (dry) {
            if (nfloats > need[id]) need[id] = nfloats;
            return nullptr;
        }
        return static_cast<float*>(d->net[id].ptr);
    }
    // output descriptor of a split-precision convolution, carrying the split-K workspace
    static constexpr size_t SPLITK_FLOATS = size_t(32) << 20;   // 128 MB: the ROI head of 24 frames x 10 rois
    SplitConvOut out() {
        SplitConvOut o;
        o.splitk_ws = buf(NB_SPLITK, SPLITK_FLOATS);
        o.splitk_ws_floats = SPLITK_FLOATS;
        o.clamped = d->clamped;
        o.b3 = d->precision == FD_PRECISION_BF16X3;
        o.b3_balance = d->b3_balance;
        return o;
    }
    void tap(const char* name, const float* p, int64_t a, int64_t b, int64_t c, int64_t e) {
        if (dry) return;
        if (d->keeping()) {
            // activation buffers are recycled between layers; keep a private copy of this one
            void* copy = nullptr;
            const size_t bytes = (size_t)(a * b * c * e) * sizeof(float);
            if (hipMalloc(&copy, bytes) == hipSuccess) {
                (void)hipMemcpyAsync(copy, p, bytes, hipMemcpyDeviceToDevice, stream);
                d->tap_copies.push_back(copy);
                p = static_cast<const float*>(copy);
            }
        }
        Tap t;
        t.name = name; t.ptr = p; t.dims[0] = a; t.dims[1] = b; t.dims[2] = c; t.dims[3] = e;
        d->taps.push_back(t);
    }
};

static int run_unit(Plan& pl, const Unit& u, const char* name, const float* x, int n, int h, int w, float* y,
                    int* ho_out, int* wo_out) {
    const int mid = u.conv1.s.cout, cout = u.conv3.s.cout;
    const int ho = u.conv2.s.out_h(h), wo = u.conv2.s.out_w(w);
    float* t1 = pl.buf(NB_T1, (size_t)n * h * w * mid);
    float* t2 = pl.buf(NB_T2, (size_t)n * ho * wo * mid);
    const float* res = x;
    if (u.has_sc) {
        float* s = pl.buf(NB_SC, (size_t)n * ho * wo * cout);
        if (!pl.dry) FD_TRY(conv_forward(u.sc, x, n, h, w, nullptr, s, pl.stream));
        res = s;
    }
    if (!pl.dry) {
        FD_TRY(conv_forward(u.conv1, x, n, h, w, nullptr, t1, pl.stream));
        FD_TRY(conv_forward(u.conv2, t1, n, h, w, nullptr, t2, pl.stream));
        FD_TRY(conv_forward(u.conv3, t2, n, ho, wo, res, y, pl.stream));
    }
    pl.tap(name, y, n, ho, wo, cout);
    *ho_out = ho;
    *wo_out = wo;
    return FD_OK;
}

// frames: uint8 [B][H][W][3].  Outputs (device, may be NULL to keep them internal):
// rois [B*R][5], cls_prob [B*R][2], bbox [B*R][8].
static int run_graph(Plan& pl, const uint8_t* frames, int B, int H, int W, float im_scale, float** rois_io,
                     float** cls_io, float** box_io) {
    fd_detector* d = pl.d;
    const int R = d->prop.post_nms_top_n;
    float* staged = pl.buf(NB_STAGED, (size_t)B * H * W * 4);
    if (!pl.dry) {
        FD_HIP(hipMemsetAsync(d->clamped, 0, sizeof(unsigned), pl.stream));     // float32 operands: nothing is clamped
        FD_TRY(stage_input(frames, B, H, W, d->norm, staged, pl.stream));
    }
    const int h1 = d->conv0.s.out_h(H), w1 = d->conv0.s.out_w(W);
    float* c0 = pl.buf(NB_CONV0, (size_t)B * h1 * w1 * FILTERS[0]);
    if (!pl.dry) FD_TRY(conv_forward(d->conv0, staged, B, H, W, nullptr, c0, pl.stream));
    pl.tap("conv0", c0, B, h1, w1, FILTERS[0]);
    int h = (h1 + 2 - 3) / 2 + 1, w = (w1 + 2 - 3) / 2 + 1;
    // ping-pong activations; sized for the widest stage
    size_t xmax = (size_t)B * h * w * FILTERS[0];
    {
        int hh = h, ww = w;
        for (int s = 0; s < 3; ++s) {
            if (s > 0) { hh = (hh + 2 - 3) / 2 + 1; ww = (ww + 2 - 3) / 2 + 1; }
            const size_t v = (size_t)B * hh * ww * FILTERS[s + 1];
            if (v > xmax) xmax = v;
        }
    }
    float* xa = pl.buf(NB_XA, xmax);
    float* xb = pl.buf(NB_XB, xmax);
    if (!pl.dry) FD_TRY(maxpool_3x3s2(c0, B, h1, w1, FILTERS[0], xa, pl.stream));
    pl.tap("pool0", xa, B, h, w, FILTERS[0]);
    float *cur = xa, *nxt = xb;
    for (int s = 0; s < 3; ++s)
        for (size_t u = 0; u < d->stage[s].size(); ++u) {
            char name[32];
            snprintf(name, sizeof(name), "stage%d_unit%d", s + 1, (int)u + 1);
            FD_TRY(run_unit(pl, d->stage[s][u], name, cur, B, h, w, nxt, &h, &w));
            float* t = cur; cur = nxt; nxt = t;
        }
    const float* feat = cur;
    pl.tap("feat", feat, B, h, w, FILTERS[3]);
    const int fh = h, fw = w;
    const int A = d->prop.num_ratios * d->prop.num_scales;
    float* hid = pl.buf(NB_RPN_HID, (size_t)B * fh * fw * 512);
    float* rout = pl.buf(NB_RPN_OUT, (size_t)B * fh * fw * 6 * A);
    if (!pl.dry) {
        FD_TRY(conv_forward(d->rpn_conv, feat, B, fh, fw, nullptr, hid, pl.stream));
        FD_TRY(conv_forward(d->rpn_out, hid, B, fh, fw, nullptr, rout, pl.stream));
    }
    pl.tap("rpn_out", rout, B, fh, fw, 6 * A);
    float* rois = pl.buf(NB_ROIS, (size_t)B * R * 5);
    float* roi_scores = pl.buf(NB_ROI_SCORES, (size_t)B * R);
    int32_t* counts = reinterpret_cast<int32_t*>(pl.buf(NB_COUNTS, (size_t)B * 4));
    if (!pl.dry) {
        if (rois_io && *rois_io) rois = *rois_io;
        RpnLayout lay;   // channels-last fused head: [pixel][bg(A) | fg(A) | deltas(4A)]
        lay.cls_batch = (long long)fh * fw * 6 * A; lay.cls_anchor = 1; lay.cls_h = (long long)fw * 6 * A;
        lay.cls_w = 6 * A; lay.cls_bg_off = 0; lay.cls_fg_off = A;
        lay.box_batch = lay.cls_batch; lay.box_chan = 1; lay.box_h = lay.cls_h; lay.box_w = lay.cls_w;
        lay.box_off = 2 * A;
        const float info[3] = {(float)H, (float)W, im_scale};
        FD_TRY(proposal_run(d->ctx, &d->prop, rout, rout, lay, 1, nullptr, info, B, fh, fw, rois, roi_scores, counts,
                            d->nan_flag, pl.stream));
    }
    pl.tap("rois", rois, B, R, 5, 1);
    pl.tap("roi_scores", roi_scores, B, R, 1, 1);
    const int N = B * R;
    float* pooled = pl.buf(NB_POOLED, (size_t)N * POOLED * POOLED * FILTERS[3]);
    if (!pl.dry)
        FD_TRY(roipool_nhwc(feat, B, FILTERS[3], fh, fw, rois, N, POOLED, POOLED, 1.0f / 16.0f, pooled, pl.stream));
    pl.tap("pooled", pooled, N, POOLED, POOLED, FILTERS[3]);
    // stage 4 treats every roi as a 7x7 image
    const size_t x4 = (size_t)N * POOLED * POOLED * FILTERS[4];
    float* x4a = pl.buf(NB_X4A, x4);
    float* x4b = pl.buf(NB_X4B, x4);
    const float* c4 = pooled;
    float* n4 = x4a;
    h = POOLED; w = POOLED;
    for (size_t u = 0; u < d->stage[3].size(); ++u) {
        char name[32];
        snprintf(name, sizeof(name), "stage4_unit%d", (int)u + 1);
        FD_TRY(run_unit(pl, d->stage[3][u], name, c4, N, h, w, n4, &h, &w));
        c4 = n4;
        n4 = (n4 == x4a) ? x4b : x4a;
    }
    float* flat = pl.buf(NB_FLAT, (size_t)N * FILTERS[4]);
    float* cls = pl.buf(NB_CLS, (size_t)N * 2);
    float* box = pl.buf(NB_BOX, (size_t)N * 8);
    if (!pl.dry) {
        if (cls_io && *cls_io) cls = *cls_io;
        if (box_io && *box_io) box = *box_io;
        FD_TRY(bn_relu_avgpool(c4, N, h * w, FILTERS[4], d->bn1_scale, d->bn1_shift, flat, pl.stream));
        FD_TRY(fc_head(flat, N, FILTERS[4], d->w_cls, d->b_cls, d->w_box, d->b_box, cls, box, pl.stream));
        if (rois_io) *rois_io = rois;
        if (cls_io) *cls_io = cls;
        if (box_io) *box_io = box;
    }
    pl.tap("flat", flat, N, FILTERS[4], 1, 1);
    pl.tap("cls_prob", cls, N, 2, 1, 1);
    pl.tap("bbox_pred", box, N, 8, 1, 1);
    return FD_OK;
}

// ----------------------------------------------------------------------------- split-precision graph
// Same layer sequence with every convolution on the f16 matrix cores (fd_conv_split.hip).
// A projection-shortcut unit consumes act = split(relu(bn1(x))), written by whoever produced
// x.  An identity-shortcut unit gets the raw float32 x only: its first convolution applies
// bn1+relu and splits while loading, so the producer writes x once instead of in both forms.
static int run_unit_split(Plan& pl, const Unit& u, const char* name, const void* act, const float* xraw, int n, int h,
                          int w, float* y32, void* ysplit, const float* next_scale, const float* next_shift,
                          int* ho_out, int* wo_out) {
    const SplitConvParams& c1 = pl.f32 ? u.f_conv1 : u.s_conv1;
    const SplitConvParams& c2 = pl.f32 ? u.f_conv2 : u.s_conv2;
    const SplitConvParams& c3 = pl.f32 ? u.f_conv3 : u.s_conv3;
    const SplitConvParams& csc = pl.f32 ? u.f_sc : u.s_sc;
    const int mid = c1.s.cout, cout = c3.s.cout;
    const int ho = c2.s.out_h(h), wo = c2.s.out_w(w);
    float* t1 = pl.buf(NB_T1, (size_t)n * h * w * mid);
    float* t2 = pl.buf(NB_T2, (size_t)n * ho * wo * mid);
    const float* res = xraw;
    // (on the ROI head's 4x4 maps only with many rois: a few dozen rois leave the batched GEMMs a handful of tiles,
    // and the split-K direct kernel is faster)
    const bool wino2 = pl.f32 && !pl.d->layerwise() && pl.d->winograd && u.w_conv2.u && wino_fits(u.w_conv2, n, h, w) &&
                       (h * w > 49 || (long long)n * h * w >= 16384);
    // split precision: the feature-map layers only (the ROI head's maps give the 36 GEMMs too few rows)
    const bool wino2s = !pl.f32 && !pl.d->layerwise() && u.w_conv2.u_split && h * w > 49 && wino_fits_split(u.w_conv2, n, h, w);
    float *wino_v = nullptr, *wino_m = nullptr;
    if (wino2 || wino2s) {
        const size_t tp = (size_t)wino_patches_padded(n, h, w, u.w_conv2.tile, wino2s ? 256 : 128);
        const size_t nb = (size_t)(u.w_conv2.tile + 2) * (u.w_conv2.tile + 2);
        wino_v = pl.buf(NB_WINO_V, nb * tp * u.w_conv2.cin);
        wino_m = pl.buf(NB_WINO_M, nb * tp * u.w_conv2.cout);
    }
    // the projection unit of the 64-channel stage as one kernel: shortcut and main branch are
    // accumulated together, so neither the shortcut tensor nor t1/t2 exist
    bool whole_proj = false;
    if (u.has_sc && !pl.dry && (!pl.d->layerwise() || (pl.d->fuse_units & 4)) && (pl.d->fuse_units & 2) && y32 && !ysplit)
        FD_TRY(unit_proj_fused_forward(c1, c2, c3, csc, act, n, h, w, y32, pl.d->clamped,
                                       pl.stream, &whole_proj));
    bool pair_done = false;
    // float32, the stride-1 projection unit: the shortcut convolution runs inside the Winograd tail as 64 more channels
    // of its 1x1 (FD_TAIL_SC=0, read at every call: the shortcut as a launch of its own and a residual read)
    const char* tw_env = getenv("FD_TAIL_WINO");
    const bool tail_wino = !(tw_env && atoi(tw_env) == 0) && !pl.d->layerwise() && pl.f32 && pl.d->winograd && u.tailw_u;
    const char* tsc_env = getenv("FD_TAIL_SC");
    const bool sc_in_tail = u.has_sc && tail_wino && pl.d->precision == FD_PRECISION_F32 && u.tailw_w3sc &&
                            !(tsc_env && atoi(tsc_env) == 0) && !whole_proj &&
                            (unsigned long long)n * h * w * 64ull * 4ull < (1ull << 31);     // (the kernel's 32-bit byte offsets)
    // ... and conv1 as well, on the input patch in LDS: the whole unit is one launch and t1 does not exist
    // (FD_TAIL_C1=0, read at every call: conv1 as a launch of its own)
    const char* tc1_env = getenv("FD_TAIL_C1");
    const bool c1_in_tail = sc_in_tail && !(tc1_env && atoi(tc1_env) == 0) && c1.f32 && c1.s.cin == 64 && c1.s.cout == 64;
    // ... of the 128- and 256-channel stages: the shortcut as cin more channels of conv3's K, a two-source GEMM
    // (FD_SC_IN_C3=0, read at every call: the shortcut as a launch of its own and a residual read)
    const char* sc3_env = getenv("FD_SC_IN_C3");
    const bool sc_in_c3 = u.has_sc && u.c3sc_w && pl.f32 && pl.d->precision == FD_PRECISION_F32 && !pl.d->layerwise() &&
                          !sc_in_tail && !whole_proj && !(sc3_env && atoi(sc3_env) == 0) &&
                          (unsigned long long)n * h * w * u.s_sc.s.cin * 4ull < (1ull << 31) &&
                          (unsigned long long)n * ho * wo * (unsigned long long)mid * 4ull < (1ull << 31);
    if (u.has_sc) {
        float* s = pl.buf(NB_SC, (size_t)n * ho * wo * cout);
        if (!pl.dry && !whole_proj && !sc_in_tail && !sc_in_c3) {
            // float32, stride 1: the shortcut and conv1 as one launch (both read act)
            if (pl.f32 && pl.d->precision == FD_PRECISION_F32 && !pl.d->layerwise() && u.pair_w)
                FD_TRY(split_conv_forward_pair(csc, c1, u.pair_w, u.pair_w_floats, static_cast<const float*>(act), n, h, w, s, t1,
                                               pl.stream, &pair_done));
            if (!pair_done) {
                SplitConvOut o = pl.out();
                o.y32 = s;
                FD_TRY(split_conv_forward(csc, act, n, h, w, nullptr, o, pl.stream));
            }
        }
        res = s;
    }
    if (!pl.dry && !whole_proj) {
        SplitConvOut o1 = pl.out(), o2 = pl.out(), o3 = pl.out();
        o1.ysplit = t1;
        o2.ysplit = t2;
        o3.y32 = y32;
        o3.ysplit = ysplit;
        o3.post_scale = next_scale;
        o3.post_shift = next_shift;
        // identity units of the 64-channel stage run as one kernel (nothing between the unit's
        // input and output goes to HBM); the intermediate tensors exist only for inspection
        bool whole = false;
        if (!u.has_sc && (!pl.d->layerwise() || (pl.d->fuse_units & 4)) && (pl.d->fuse_units & 1))
            FD_TRY(unit_fused_forward(c1, c2, c3, xraw, u.bn1_scale, u.bn1_shift, n, h, w, o3,
                                      pl.stream, &whole));
        if (!whole) {
            if (u.has_sc) {
                if (!pair_done && !c1_in_tail) FD_TRY(split_conv_forward(c1, act, n, h, w, nullptr, o1, pl.stream));
            } else {
                FD_TRY(split_conv_forward_f32(c1, xraw, u.bn1_scale, u.bn1_shift, n, h, w, nullptr, o1,
                                              pl.stream));
            }
            bool fused = false;
            // float32: the tail with its 3x3 in Winograd F(2x2,3x3) form (FD_TAIL_WINO=0, read at every call so that a
            // test can run both: the direct-form tail below; FD_WINOGRAD=0 keeps every layer direct as well)
            if (sc_in_tail) {
                FD_TRY(conv_tail_wino_forward(u.tailw_u, c2.bias, c3, t1, n, h, w, nullptr, o3, pl.stream, &fused,
                                              static_cast<const float*>(act), u.tailw_w3sc, c1_in_tail ? &c1 : nullptr));
                FD_REQUIRE(fused, "float32 tail with the shortcut inside: the kernel refused a layer it was built for");
            } else if (tail_wino) {
                FD_TRY(conv_tail_wino_forward(u.tailw_u, c2.bias, c3, t1, n, h, w, res, o3, pl.stream, &fused));
            }
            if (!fused && !pl.d->layerwise())
                FD_TRY(conv_tail_fused_forward(c2, c3, t1, n, h, w, res, o3, pl.stream, &fused));
            if (!fused) {
                if (wino2) FD_TRY(wino_conv_forward(u.w_conv2, t1, n, h, w, wino_v, wino_m, t2, pl.stream, o2.b3, o2.b3_balance));
                else if (wino2s) FD_TRY(wino_conv_forward_split(u.w_conv2, t1, n, h, w, wino_v, wino_m, t2, pl.d->clamped, pl.stream));
                else FD_TRY(split_conv_forward(c2, t1, n, h, w, nullptr, o2, pl.stream));
                bool two = false;
                if (sc_in_c3) {
                    FD_TRY(split_conv_forward_two(c3, u.c3sc_w, t2, n, ho, wo, static_cast<const float*>(act), h, w, csc.s.cin,
                                                  csc.s.stride, o3, pl.stream, &two));
                    FD_REQUIRE(two, "two-source convolution refused a layer it was built for");
                }
                if (!two) FD_TRY(split_conv_forward(c3, t2, n, ho, wo, res, o3, pl.stream));
            }
        }
    }
    if (y32) pl.tap(name, y32, n, ho, wo, cout);
    *ho_out = ho;
    *wo_out = wo;
    return FD_OK;
}

static int run_graph_split(Plan& pl, const uint8_t* frames, int B, int H, int W, float im_scale, float** rois_io,
                           float** cls_io, float** box_io) {
    fd_detector* d = pl.d;
    const int R = d->prop.post_nms_top_n;
    const bool keep = d->keeping();          // unit outputs are also written as float32 and snapshotted
    const bool layerwise = d->layerwise();   // ... and the stem, pooling and ROI pooling run as separate kernels
    (void)pl.out();                        // sizes the split-K workspace during the dry run
    const int h1 = d->s_conv0.s.out_h(H), w1 = d->s_conv0.s.out_w(W);
    // Stem and max-pool run as one kernel on a planar staged image.  When intermediate tensors are
    // kept for inspection the two-kernel form is used instead, because only it materialises the
    // convolution output.
    const size_t planar_floats = (size_t)B * H * staged_planar_width(W) * 4;
    const bool f32 = pl.f32;                 // float32 operands: same graph, "split" tensors are plain float32
    const bool fused_stem = !layerwise && planar_floats * 4 < (1ull << 31);
    float* staged = pl.buf(NB_STAGED, fused_stem ? planar_floats : (size_t)B * H * W * 4);
    if (!pl.dry) {
        FD_HIP(hipMemsetAsync(d->clamped, 0, sizeof(unsigned), pl.stream));
        if (f32 && fused_stem) FD_TRY(stage_input_padded_f32(frames, B, H, W, d->norm, staged, pl.stream));
        else if (f32) FD_TRY(stage_input(frames, B, H, W, d->norm, staged, pl.stream));
        else if (fused_stem) FD_TRY(stage_input_planar(frames, B, H, W, d->norm, staged, d->clamped, pl.stream));
        else FD_TRY(stage_input_split(frames, B, H, W, d->norm, staged, d->clamped, pl.stream));
    }
    float* c0 = fused_stem ? nullptr : pl.buf(NB_CONV0, (size_t)B * h1 * w1 * FILTERS[0]);
    if (!pl.dry && !fused_stem) {
        if (f32) {
            // the 7x7 stem on the float32 staged image: the register-staged fp32 kernel
            FD_TRY(conv_forward(d->conv0, staged, B, H, W, nullptr, c0, pl.stream));
        } else {
            SplitConvOut o = pl.out();
            o.y32 = c0;
            FD_TRY(split_conv_forward(d->s_conv0, staged, B, H, W, nullptr, o, pl.stream));
        }
    }
    if (!fused_stem) pl.tap("conv0", c0, B, h1, w1, FILTERS[0]);
    int h = (h1 + 2 - 3) / 2 + 1, w = (w1 + 2 - 3) / 2 + 1;
    size_t xmax = (size_t)B * h * w * FILTERS[0];
    {
        int hh = h, ww = w;
        for (int s = 0; s < 3; ++s) {
            if (s > 0) { hh = (hh + 2 - 3) / 2 + 1; ww = (ww + 2 - 3) / 2 + 1; }
            const size_t v = (size_t)B * hh * ww * FILTERS[s + 1];
            if (v > xmax) xmax = v;
        }
    }
    float* xbuf[2] = {pl.buf(NB_XA, xmax), pl.buf(NB_XB, xmax)};
    float* abuf[2] = {pl.buf(NB_ACT_A, xmax), pl.buf(NB_ACT_B, xmax)};
    // max pool: float32 copy only for inspection; the first unit (projection shortcut) needs act only
    if (!pl.dry) {
        if (fused_stem)
            FD_TRY(stem_pool_forward(f32 ? d->f_conv0 : d->s_conv0, staged, B, H, W, d->stage[0][0].bn1_scale,
                                     d->stage[0][0].bn1_shift, nullptr, abuf[0], d->clamped, pl.stream));
        else
            FD_TRY(maxpool_3x3s2_split(c0, B, h1, w1, FILTERS[0], d->stage[0][0].bn1_scale,
                                       d->stage[0][0].bn1_shift, layerwise ? xbuf[0] : nullptr, abuf[0], d->clamped,
                                       pl.stream, f32));
    }
    if (layerwise && !fused_stem) pl.tap("pool0", xbuf[0], B, h, w, FILTERS[0]);
    int cur = 0;
    const float* xraw = nullptr;           // raw input of the next unit (NULL when it does not need one)
    for (int s = 0; s < 3; ++s)
        for (size_t u = 0; u < d->stage[s].size(); ++u) {
            char name[32];
            snprintf(name, sizeof(name), "stage%d_unit%d", s + 1, (int)u + 1);
            const bool last = (s == 2 && u + 1 == d->stage[s].size());
            const Unit* next = last ? nullptr : (u + 1 < d->stage[s].size() ? &d->stage[s][u + 1] : &d->stage[s + 1][0]);
            // the raw sum is needed by an identity-shortcut successor, by ROI pooling, or for inspection
            const bool want32 = last || keep || (next && !next->has_sc);
            float* y32 = want32 ? xbuf[cur ^ 1] : nullptr;
            // the split form is for a projection-shortcut successor and, after the last unit, the RPN
            // (with float32 operands the RPN reads the raw feature map itself)
            const bool want_split = (last && !f32) || (next && next->has_sc);
            FD_TRY(run_unit_split(pl, d->stage[s][u], name, abuf[cur], d->stage[s][u].has_sc ? nullptr : xraw, B, h, w,
                                  y32, want_split ? abuf[cur ^ 1] : nullptr, next ? next->bn1_scale : nullptr,
                                  next ? next->bn1_shift : nullptr, &h, &w));
            xraw = y32;
            cur ^= 1;
        }
    const float* feat = xbuf[cur];
    const void* feat_split = f32 ? (const void*)xbuf[cur] : (const void*)abuf[cur];   // the RPN's input: the raw map (split form: no BatchNorm)
    pl.tap("feat", feat, B, h, w, FILTERS[3]);
    const int fh = h, fw = w;
    const int A = d->prop.num_ratios * d->prop.num_scales;
    float* hid = pl.buf(NB_RPN_HID, (size_t)B * fh * fw * 512);
    float* rout = pl.buf(NB_RPN_OUT, (size_t)B * fh * fw * 6 * A);
    // the 3x3 convolution runs in Winograd form (a quarter of the multiplies with F(4x4,3x3)) unless tensors
    // are inspected layer by layer
    const bool wino = f32 && !layerwise && d->winograd && wino_fits(d->w_rpn_conv, B, fh, fw);
    const bool winos = !f32 && !layerwise && d->w_rpn_conv.u_split && wino_fits_split(d->w_rpn_conv, B, fh, fw);
    float *wino_v = nullptr, *wino_m = nullptr;
    if (wino || winos) {
        const size_t tp = (size_t)wino_patches_padded(B, fh, fw, d->w_rpn_conv.tile, winos ? 256 : 128);
        const size_t nb = (size_t)(d->w_rpn_conv.tile + 2) * (d->w_rpn_conv.tile + 2);
        wino_v = pl.buf(NB_WINO_V, nb * tp * d->w_rpn_conv.cin);
        wino_m = pl.buf(NB_WINO_M, nb * tp * d->w_rpn_conv.cout);
    }
    if (!pl.dry) {
        SplitConvOut o1 = pl.out(), o2 = pl.out();
        o1.ysplit = hid;
        o2.y32 = rout;
        if (wino) FD_TRY(wino_conv_forward(d->w_rpn_conv, feat, B, fh, fw, wino_v, wino_m, hid, pl.stream, o1.b3, o1.b3_balance));
        else if (winos) FD_TRY(wino_conv_forward_split(d->w_rpn_conv, feat_split, B, fh, fw, wino_v, wino_m, hid, d->clamped, pl.stream));
        else FD_TRY(split_conv_forward(f32 ? d->f_rpn_conv : d->s_rpn_conv, feat_split, B, fh, fw, nullptr, o1, pl.stream));
        FD_TRY(split_conv_forward(f32 ? d->f_rpn_out : d->s_rpn_out, hid, B, fh, fw, nullptr, o2, pl.stream));
    }
    pl.tap("rpn_out", rout, B, fh, fw, 6 * A);
    float* rois = pl.buf(NB_ROIS, (size_t)B * R * 5);
    float* roi_scores = pl.buf(NB_ROI_SCORES, (size_t)B * R);
    int32_t* counts = reinterpret_cast<int32_t*>(pl.buf(NB_COUNTS, (size_t)B * 4));
    if (!pl.dry) {
        if (rois_io && *rois_io) rois = *rois_io;
        RpnLayout lay;
        lay.cls_batch = (long long)fh * fw * 6 * A; lay.cls_anchor = 1; lay.cls_h = (long long)fw * 6 * A;
        lay.cls_w = 6 * A; lay.cls_bg_off = 0; lay.cls_fg_off = A;
        lay.box_batch = lay.cls_batch; lay.box_chan = 1; lay.box_h = lay.cls_h; lay.box_w = lay.cls_w;
        lay.box_off = 2 * A;
        const float info[3] = {(float)H, (float)W, im_scale};
        FD_TRY(proposal_run(d->ctx, &d->prop, rout, rout, lay, 1, nullptr, info, B, fh, fw, rois, roi_scores, counts,
                            d->nan_flag, pl.stream));
    }
    pl.tap("rois", rois, B, R, 5, 1);
    pl.tap("roi_scores", roi_scores, B, R, 1, 1);
    const int N = B * R;
    float* pooled = layerwise ? pl.buf(NB_POOLED, (size_t)N * POOLED * POOLED * FILTERS[3]) : nullptr;
    const size_t x4 = (size_t)N * POOLED * POOLED * FILTERS[4];
    float* x4buf[2] = {pl.buf(NB_X4A, x4), pl.buf(NB_X4B, x4)};
    float* a4buf[2] = {pl.buf(NB_ACT_A, x4), pl.buf(NB_ACT_B, x4)};
    if (!pl.dry) {
        // feat_split lives in one of the act buffers and is no longer needed: reuse them for stage 4.
        // The float32 pooled tensor is only materialised for inspection.
        if (layerwise) {
            FD_TRY(roipool_nhwc(feat, B, FILTERS[3], fh, fw, rois, N, POOLED, POOLED, 1.0f / 16.0f, pooled, pl.stream));
            FD_TRY(to_split(pooled, (long long)N * POOLED * POOLED, FILTERS[3], d->stage[3][0].bn1_scale,
                            d->stage[3][0].bn1_shift, a4buf[0], d->clamped, pl.stream, f32));
        } else {
            FD_TRY(roipool_nhwc_split(feat, B, FILTERS[3], fh, fw, rois, N, POOLED, POOLED, 1.0f / 16.0f,
                                      d->stage[3][0].bn1_scale, d->stage[3][0].bn1_shift, a4buf[0], d->clamped,
                                      pl.stream, f32));
        }
    }
    if (layerwise) pl.tap("pooled", pooled, N, POOLED, POOLED, FILTERS[3]);
    h = POOLED; w = POOLED;
    cur = 0;
    xraw = nullptr;
    for (size_t u = 0; u < d->stage[3].size(); ++u) {
        char name[32];
        snprintf(name, sizeof(name), "stage4_unit%d", (int)u + 1);
        const bool last = u + 1 == d->stage[3].size();
        const Unit* next = last ? nullptr : &d->stage[3][u + 1];
        FD_TRY(run_unit_split(pl, d->stage[3][u], name, a4buf[cur], d->stage[3][u].has_sc ? nullptr : xraw, N, h, w,
                              x4buf[cur ^ 1], (next && next->has_sc) ? a4buf[cur ^ 1] : nullptr,
                              next ? next->bn1_scale : nullptr,
                              next ? next->bn1_shift : nullptr, &h, &w));
        xraw = x4buf[cur ^ 1];
        cur ^= 1;
    }
    const float* c4 = x4buf[cur];
    float* flat = pl.buf(NB_FLAT, (size_t)N * FILTERS[4]);
    float* cls = pl.buf(NB_CLS, (size_t)N * 2);
    float* box = pl.buf(NB_BOX, (size_t)N * 8);
    if (!pl.dry) {
        if (cls_io && *cls_io) cls = *cls_io;
        if (box_io && *box_io) box = *box_io;
        FD_TRY(bn_relu_avgpool(c4, N, h * w, FILTERS[4], d->bn1_scale, d->bn1_shift, flat, pl.stream));
        FD_TRY(fc_head(flat, N, FILTERS[4], d->w_cls, d->b_cls, d->w_box, d->b_box, cls, box, pl.stream));
        if (rois_io) *rois_io = rois;
        if (cls_io) *cls_io = cls;
        if (box_io) *box_io = box;
    }
    pl.tap("flat", flat, N, FILTERS[4], 1, 1);
    pl.tap("cls_prob", cls, N, 2, 1, 1);
    pl.tap("bbox_pred", box, N, 8, 1, 1);
    return FD_OK;
}

static int forward(fd_detector* d, const uint8_t* frames, int B, int H, int W, float im_scale, float** rois,
                   float** cls, float** box, hipStream_t stream) {
    FD_REQUIRE(d && frames, "forward: NULL argument");
    FD_REQUIRE(B > 0 && H >= 32 && W >= 32, "forward: frames must be at least 32x32 (got %d x %d x %d)", B, H, W);
    FD_REQUIRE(d->prop.post_nms_top_n > 0, "forward: post_nms_top_n must be > 0");
    FD_HIP(hipSetDevice(d->ctx->device));
    Plan pl;
    pl.d = d; pl.stream = stream; pl.dry = true;
    memset(pl.need, 0, sizeof(pl.need));
    // Both precisions run the graph of run_graph_split; FD_F32_LAYERWISE=1 selects the older float32 pass
    // that launches every layer with the register-staged fp32 kernel (comparisons only).
    static const bool legacy_f32 = getenv("FD_F32_LAYERWISE") && atoi(getenv("FD_F32_LAYERWISE")) != 0;
    pl.f32 = d->precision != FD_PRECISION_F16X3;      // float32 operand tensors (bf16x3 forms its products from them)
    const bool split = !(pl.f32 && legacy_f32);
    FD_TRY((split ? run_graph_split : run_graph)(pl, frames, B, H, W, im_scale, nullptr, nullptr, nullptr));
    for (int i = 0; i < NB_COUNT; ++i) {
        Buf& b = d->net[i];
        const size_t bytes = pl.need[i] * sizeof(float);
        if (b.cap < bytes) {
            void* p = nullptr;
            const size_t want = (bytes + 255) & ~size_t(255);
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess) {
                set_error("activation buffer %d: hipMalloc(%zu) failed: %s", i, want, hipGetErrorString(e));
                return FD_ERR_HIP;
            }
            if (b.ptr) FD_TRY(park(d->ctx, b.ptr, b.cap));   // kernels in flight may still read it
            b.ptr = p;
            b.cap = want;
            d->ctx->bytes += (int64_t)want;
        }
    }
    pl.dry = false;
    d->taps.clear();
    if (!d->tap_copies.empty()) {
        FD_HIP(hipStreamSynchronize(stream));
        for (void* p : d->tap_copies) (void)hipFree(p);
        d->tap_copies.clear();
    }
    return (split ? run_graph_split : run_graph)(pl, frames, B, H, W, im_scale, rois, cls, box);
}

// detection.py:145-148: int(round(coordinate / scale)) for the four coordinates of each row.
__global__ __launch_bounds__(256) void boxes_to_source_kernel(const float* __restrict__ dets, int total, double scale,
                                                             int32_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    out[i] = (int32_t)rint((double)dets[(size_t)(i >> 2) * 5 + (i & 3)] / scale);
}

// Frames one pass can take.  The float32 kernels address their tensors through buffer descriptors with
// 32-bit byte offsets below 2^31; the largest tensors of a pass are the stem output, the stage-1 tensors
// and, with many proposals per frame, the ROI head's.  The f16x3 kernels have pointer-addressed forms for
// larger tensors and take any batch.  FD_F32_PASS_FRAMES lowers the limit (tests).
static int frames_per_pass(const fd_detector* d, int batch, int H, int W) {
    if (d->precision == FD_PRECISION_F16X3) return batch;
    const long long h2 = (H - 1) / 2 + 1, w2 = (W - 1) / 2 + 1;      // stem (7x7, stride 2, pad 3)
    const long long h4 = (h2 - 1) / 2 + 1, w4 = (w2 - 1) / 2 + 1;    // max-pool (3x3, stride 2, pad 1)
    long long per_frame = std::max(h2 * w2 * FILTERS[0], h4 * w4 * FILTERS[1]) * 4;
    // ROI head: the pooled maps (7x7, 1024 channels) are its largest tensor; 4x4 maps of 2048 channels follow
    per_frame = std::max(per_frame, (long long)d->prop.post_nms_top_n * POOLED * POOLED * FILTERS[3] * 4);
    long long lim = std::max(1ll, ((1ll << 31) - 1) / per_frame);
    if (const char* e = getenv("FD_F32_PASS_FRAMES"))
        if (atoi(e) > 0) lim = std::min<long long>(lim, atoi(e));
    return (int)std::min<long long>(batch, lim);
}

// status block of a call that took several passes: NaN flags OR-ed, clamp counts summed
__global__ void status_merge_kernel(int32_t* __restrict__ acc, const int32_t* __restrict__ cur, int first) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    acc[0] = first ? cur[0] : (acc[0] | cur[0]);
    const unsigned sum = (first ? 0u : (unsigned)acc[1]) + (unsigned)cur[1];
    acc[1] = (int32_t)min(sum, 0x7fffffffu);
}

// forward + tail over `batch` frames in passes of at most frames_per_pass frames; a frame's result does not
// depend on which frames share its pass
static int detect_passes(fd_detector* d, const uint8_t* frames, int batch, int height, int width, float im_scale,
                         float thresh, float nms_thresh, int do_round, float* dets, int32_t* n_dets, hipStream_t s) {
    FD_REQUIRE(batch > 0 && height > 0 && width > 0, "detect: bad frame shape %d x %d x %d", batch, height, width);
    const int per = frames_per_pass(d, batch, height, width);
    const int R = d->prop.post_nms_top_n;
    for (int b0 = 0; b0 < batch; b0 += per) {
        const int nb = std::min(per, batch - b0);
        float *r = nullptr, *c = nullptr, *b = nullptr;
        FD_TRY(forward(d, frames + (size_t)b0 * height * width * 3, nb, height, width, im_scale, &r, &c, &b, s));
        FD_TRY(postprocess_run(d->ctx, r, 5, c, b, nb, R, height, width, thresh, nms_thresh, do_round,
                               dets + (size_t)b0 * R * 5, n_dets + b0, s));
        if (per < batch) {
            hipLaunchKernelGGL(status_merge_kernel, dim3(1), dim3(64), 0, s, d->nan_flag + 2, d->nan_flag, b0 == 0 ? 1 : 0);
            FD_LAUNCH_CHECK();
        }
    }
    if (per < batch)
        FD_HIP(hipMemcpyAsync(d->nan_flag, d->nan_flag + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return FD_OK;
}

}  // namespace fd

// =================================================================================
// C ABI
// =================================================================================
extern "C" {

int fd_detector_create(fd_ctx* ctx, const fd_param* params, int n_params, const fd_proposal_cfg* cfg,
                       fd_detector** out) {
    FD_REQUIRE(ctx && params && n_params > 0 && cfg && out, "fd_detector_create: bad arguments");
    FD_REQUIRE(cfg->num_ratios > 0 && cfg->num_scales > 0 && cfg->num_ratios <= 8 && cfg->num_scales <= 8 &&
                   cfg->num_ratios * cfg->num_scales <= 32,
               "fd_detector_create: unsupported anchor configuration");
    FD_HIP(hipSetDevice(ctx->device));
    fd::ParamTable t;
    for (int i = 0; i < n_params; ++i)
        if (params[i].name) t.m[params[i].name] = &params[i];
    fd_detector* d = new fd_detector();
    d->ctx = ctx;
    if (const char* e = getenv("FD_FUSE_UNITS")) d->fuse_units = atoi(e);
    if (const char* e = getenv("FD_WINOGRAD")) {
        const int v = atoi(e);
        d->winograd = v == 0 ? 0 : (v == 4 ? 4 : 2);          // 1 and 2: F(2x2,3x3)
    }
    if (const char* e = getenv("FD_WINOGRAD_SPLIT")) d->winograd_split = atoi(e) != 0;
    if (d->winograd != 4) d->winograd_split = false;
    d->prop = *cfg;
    const int rc = fd::build(d, t);
    if (rc != FD_OK) {
        for (void* p : d->owned) (void)hipFree(p);
        delete d;
        return rc;
    }
    *out = d;
    return FD_OK;
}

void fd_detector_destroy(fd_detector* d) {
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : d->owned) (void)hipFree(p);
    for (void* p : d->tap_copies) (void)hipFree(p);
    for (auto& b : d->net)
        if (b.ptr) (void)hipFree(b.ptr);
    delete d;
}

int fd_detector_set_proposal(fd_detector* d, const fd_proposal_cfg* cfg) {
    FD_REQUIRE(d && cfg, "fd_detector_set_proposal: NULL argument");
    FD_REQUIRE(cfg->num_ratios == d->prop.num_ratios && cfg->num_scales == d->prop.num_scales,
               "fd_detector_set_proposal: the anchor count is fixed by the weights");
    FD_REQUIRE(cfg->post_nms_top_n > 0, "fd_detector_set_proposal: post_nms_top_n must be > 0");
    d->prop = *cfg;
    return FD_OK;
}

int fd_detector_set_precision(fd_detector* d, int precision) {
    FD_REQUIRE(d, "fd_detector_set_precision: NULL detector");
    FD_REQUIRE(precision == FD_PRECISION_F32 || precision == FD_PRECISION_F16X3 || precision == FD_PRECISION_BF16X3,
               "fd_detector_set_precision: unknown precision %d", precision);
    d->precision = precision;
    if (const char* e = getenv("FD_B3_BALANCE")) d->b3_balance = atoi(e);
    return FD_OK;
}

int fd_detector_keep_tensors(fd_detector* d, int keep) {
    FD_REQUIRE(d, "fd_detector_keep_tensors: NULL detector");
    FD_REQUIRE(keep >= 0 && keep <= 2, "fd_detector_keep_tensors: keep must be 0, 1 or 2");
    d->keep_mode = keep;
    return FD_OK;
}

int fd_forward(fd_detector* d, const uint8_t* frames, int batch, int height, int width, float im_scale, float* rois,
               float* cls_prob, float* bbox_pred, void* stream) {
    float *r = rois, *c = cls_prob, *b = bbox_pred;
    return fd::forward(d, frames, batch, height, width, im_scale, &r, &c, &b, fd::as_stream(stream));
}

int fd_detect(fd_detector* d, const uint8_t* frames, int batch, int height, int width, float thresh,
              float nms_thresh, int do_round, float* dets, int32_t* n_dets, int32_t* nan_flag, void* stream) {
    FD_REQUIRE(d && frames && dets && n_dets, "fd_detect: NULL argument");
    hipStream_t s = fd::as_stream(stream);
    FD_TRY(fd::detect_passes(d, frames, batch, height, width, 1.0f, thresh, nms_thresh, do_round, dets, n_dets, s));
    if (nan_flag) FD_HIP(hipMemcpyAsync(nan_flag, d->nan_flag, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return FD_OK;
}

int fd_detect_scaled(fd_detector* d, const uint8_t* frames, int batch, int height, int width, float im_scale,
                     float thresh, float nms_thresh, float* dets, int32_t* n_dets, int32_t* boxes_src,
                     int32_t* nan_flag, void* stream) {
    FD_REQUIRE(d && frames && dets && n_dets && im_scale > 0.0f, "fd_detect_scaled: bad arguments");
    hipStream_t s = fd::as_stream(stream);
    FD_TRY(fd::detect_passes(d, frames, batch, height, width, im_scale, thresh, nms_thresh, 0, dets, n_dets, s));
    if (boxes_src) {
        const int total = batch * d->prop.post_nms_top_n * 4;
        hipLaunchKernelGGL(fd::boxes_to_source_kernel, dim3(fd::ceil_div(total, 256)), dim3(256), 0, s, dets, total,
                           (double)im_scale, boxes_src);
        FD_LAUNCH_CHECK();
    }
    if (nan_flag) FD_HIP(hipMemcpyAsync(nan_flag, d->nan_flag, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// A detection call recorded once and replayed: the ~70-150 launches of a pass become one graph launch, which
// takes the host's launch cost and the gaps between short kernels out of a single frame's latency.
struct fd_detect_graph {
    fd_detector* det = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t capture_stream = nullptr;
    uint64_t epoch = 0;
};

static void graph_release(fd_detect_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->capture_stream) (void)hipStreamDestroy(g->capture_stream);
    delete g;
}

int fd_detect_graph_create(fd_detector* d, const uint8_t* frames, int batch, int height, int width, float im_scale,
                           float thresh, float nms_thresh, int do_round, float* dets, int32_t* n_dets,
                           int32_t* boxes_src, int32_t* status, fd_detect_graph** out) {
    FD_REQUIRE(d && frames && dets && n_dets && out && im_scale > 0.0f, "fd_detect_graph_create: bad arguments");
    FD_REQUIRE(!fd::prof_active(), "fd_detect_graph_create: not while a profiling session is open");
    FD_REQUIRE(d->keep_mode == 0, "fd_detect_graph_create: not while intermediate tensors are kept");
    FD_HIP(hipSetDevice(d->ctx->device));
    fd_detect_graph* g = new fd_detect_graph();
    g->det = d;
    if (hipStreamCreateWithFlags(&g->capture_stream, hipStreamNonBlocking) != hipSuccess) {
        graph_release(g);
        fd::set_error("fd_detect_graph_create: no stream");
        return FD_ERR_HIP;
    }
    hipStream_t s = g->capture_stream;
    auto body = [&]() -> int {
        FD_TRY(fd::detect_passes(d, frames, batch, height, width, im_scale, thresh, nms_thresh, do_round, dets, n_dets, s));
        if (boxes_src) {
            const int total = batch * d->prop.post_nms_top_n * 4;
            hipLaunchKernelGGL(fd::boxes_to_source_kernel, dim3(fd::ceil_div(total, 256)), dim3(256), 0, s, dets, total,
                               (double)im_scale, boxes_src);
            FD_LAUNCH_CHECK();
        }
        if (status) FD_HIP(hipMemcpyAsync(status, d->nan_flag, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        return FD_OK;
    };
    // an ordinary call first: every activation and scratch buffer reaches the size this shape needs, so the
    // recorded call allocates nothing (an allocation would fail the capture) and records final addresses
    int rc = body();
    if (rc == FD_OK && hipStreamSynchronize(s) != hipSuccess) {
        fd::set_error("fd_detect_graph_create: the warm-up call failed");
        rc = FD_ERR_HIP;
    }
    if (rc != FD_OK) {
        graph_release(g);
        return rc;
    }
    const uint64_t epoch0 = d->ctx->epoch;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        graph_release(g);
        fd::set_error("fd_detect_graph_create: hipStreamBeginCapture failed");
        return FD_ERR_HIP;
    }
    rc = body();
    const hipError_t e = hipStreamEndCapture(s, &g->graph);      // also ends a capture that a failed call broke
    if (rc != FD_OK || e != hipSuccess || !g->graph || d->ctx->epoch != epoch0) {
        if (rc == FD_OK)
            fd::set_error("fd_detect_graph_create: capture failed (%s)",
                          e != hipSuccess ? hipGetErrorString(e) : "a buffer was re-allocated while recording");
        (void)hipGetLastError();
        graph_release(g);
        return rc != FD_OK ? rc : FD_ERR_HIP;
    }
    const hipError_t ei = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        fd::set_error("fd_detect_graph_create: hipGraphInstantiate failed: %s", hipGetErrorString(ei));
        graph_release(g);
        return FD_ERR_HIP;
    }
    g->epoch = epoch0;
    *out = g;
    return FD_OK;
}

int fd_detect_graph_launch(fd_detect_graph* g, void* stream) {
    FD_REQUIRE(g && g->exec, "fd_detect_graph_launch: NULL graph");
    FD_REQUIRE(g->epoch == g->det->ctx->epoch,
               "fd_detect_graph_launch: the library's buffers were re-allocated after this call was recorded "
               "(a larger shape ran since): record it again");
    FD_HIP(hipSetDevice(g->det->ctx->device));
    FD_HIP(hipGraphLaunch(g->exec, fd::as_stream(stream)));
    return FD_OK;
}

void fd_detect_graph_destroy(fd_detect_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->det->ctx->device);
    (void)hipDeviceSynchronize();
    graph_release(g);
}

int fd_detector_status(fd_detector* d, int32_t* status, void* stream) {
    FD_REQUIRE(d && status, "fd_detector_status: NULL argument");
    FD_HIP(hipMemcpyAsync(status, d->nan_flag, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, fd::as_stream(stream)));
    return FD_OK;
}

int fd_detector_tensor(fd_detector* d, const char* name, float* out, int64_t capacity, int64_t* dims4, void* stream) {
    FD_REQUIRE(d && name && dims4, "fd_detector_tensor: NULL argument");
    for (const fd::Tap& t : d->taps) {
        if (t.name != name) continue;
        int64_t n = 1;
        for (int i = 0; i < 4; ++i) { dims4[i] = t.dims[i]; n *= t.dims[i]; }
        if (out) {
            FD_REQUIRE(capacity >= n, "fd_detector_tensor: '%s' needs %lld floats, buffer holds %lld", name,
                       (long long)n, (long long)capacity);
            FD_HIP(hipMemcpyAsync(out, t.ptr, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice,
                                  fd::as_stream(stream)));
        }
        return FD_OK;
    }
    fd::set_error("fd_detector_tensor: no tensor named '%s' in the last forward pass", name);
    return FD_ERR_ARG;
}

}  // extern "C"
