// ore_engine: owns packed weights + every intermediate NHWC buffer of the eval hot path and enqueues
// the whole layer sequence (optionally as one replayed hipGraph):
//
//   image -> stem_1 (fused preprocess) -> stem_2 -> stem_3 -> 4 x OSA stage [maxpool(gated) ->
//   3 x conv3x3 writing channel slices of ONE concat buffer -> 1x1 concat conv -> eSE gate] ->
//   FPN (gate folded into the lateral 1x1, top-down add in its epilogue) -> depthwise correlation ->
//   shared 1x1 conv3 over [attn|q] -> CenterNet tower conv -> GroupNorm stats -> fused (reg|hm) conv
//   with GN+ReLU on its input -> sigmoid/top-k/decode/NMS.
//
// The OSA concat, the eSE product, the FPN upsample+add, torch.cat((attn,q)) and the GroupNorm output
// are never materialised.  Reference call sites: include/ore_hip.h.
#include <map>
#include <string>
#include <vector>
#include <math.h>
#include <string.h>
#include "ore_common.h"
#include "ore_conv_internal.h"
#include <algorithm>


namespace {

struct HostTensor { std::vector<float> v; std::vector<int64_t> shape; };

struct Conv {
    float* w = nullptr; float* scale = nullptr; float* shift = nullptr;
    uint16_t* wh = nullptr;              // bf16 packed weights (engines created in ORE_CONV_BF16S mode)
    float* wino = nullptr;               // Winograd F(2x2,3x3) form of w where ore_winograd_covers() (else null)
    int Cin = 0, Cout = 0, k = 1, stride = 1, pad = 0, relu_cout = 0;
};

struct Buf { float* p = nullptr; int ld = 0; size_t rows = 0; };

struct Geo { int B, H, W, Hp, Wp; int h[6], w[6]; };  // h[1]=Hp/2 (stem), h[2]=Hp/4 ... h[5]=Hp/32

}  // namespace

// conv operand precision (ore_conv_set_precision) for the lifetime of a scope
struct PrecisionScope {
    int saved;
    explicit PrecisionScope(int mode) : saved(ore_conv_get_precision()) { ore_conv_set_precision(mode); }
    ~PrecisionScope() { ore_conv_set_precision(saved); }
};

struct ore_engine {
    ore_model_cfg cfg{};
    int device = 0;
    int conv_precision = ORE_CONV_FP32;                   // mode in force at ore_engine_create; every launch of this engine uses it
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;
    // weights
    float* stem1_w = nullptr; float* stem1_scale = nullptr; float* stem1_shift = nullptr;
    Conv stem2, stem3;
    struct Stage { Conv layer[8]; Conv concat; float* fc_w = nullptr; float* fc_b = nullptr; int in_ch, conv_ch, out_ch, cat_ch; } stage[4];
    Conv lateral[3], output[3], conv3, tower, pred;      // pred.scale/shift are [3 levels][16]
    float* out_wino3 = nullptr; float* out_shift3 = nullptr; size_t out_wino_stride = 0;   // the three FPN output convs as ONE per-level Winograd launch: [3][U], [3][F] bias
    uint16_t* out_wh3 = nullptr; size_t out_wh_stride = 0;                                 // bf16 storage: [3][packed bf16 weights] for the weight-stationary kernel
    float* gn_gamma = nullptr; float* gn_beta = nullptr;
    float* k11 = nullptr, *k13 = nullptr, *k31 = nullptr; // level-major [3][C], [3][C][3], [3][C][3]
    HostTensor support[3];
    bool support_set[3] = {false, false, false};
    // buffers
    void* img_in = nullptr; size_t img_bytes = 0;
    Buf s1, s2, cat[4], sout[4], lat_all;                 // lat_all: the three inner (top-down) FPN maps, level-major rows
    Buf pcat, pos, tow, head;                    // all pyramid levels in ONE level-major matrix each ([level][b][y][x])
    bool head_pred_valu = true;                  // k_head_pred (VALU) instead of the MFMA conv for the head's last step (A/B: ORE_HEAD_MFMA=1)
    Buf tn;                                      // bf16 storage only: GroupNorm + ReLU of the tower, materialised (the DMA-fed kernels cannot touch their A operand)
    bool sb() const { return conv_precision == ORE_CONV_BF16S; }
    // element offset into an ACTIVATION buffer (fp32, or bf16 in ORE_CONV_BF16S engines)
    float* at(float* base, size_t elems) const { return sb() ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(base) + elems) : base + elems; }
    float* gate[4] = {};
    float* lat_scaled[3] = {};                   // bs = 1: the lateral weights times the stage's eSE gate (written by the gate kernel)
    float* gn_mul = nullptr, *gn_add = nullptr;  // [3*B][C]
    float* ws = nullptr; size_t ws_floats = 0;   // conv split-K counters + slabs (include/ore_hip.h workspace contract)
    float* ese_ws = nullptr;
    float* gn_ws = nullptr;
    float* colsum = nullptr; size_t colsum_floats = 0;
    // detect
    float* pre_boxes = nullptr; float* pre_scores = nullptr; int64_t* pre_loc = nullptr; int32_t* pre_level = nullptr;
    int64_t* keep_idx = nullptr; int32_t* counts = nullptr; float* out_boxes = nullptr; float* out_scores = nullptr;
    void* det_ws = nullptr; size_t det_ws_bytes = 0;
    // second stage (optional: ore_engine_set_roi_head)
    bool roi_set = false;
    int roi_cap = ORE_DET_RECORD_ROWS, roi_fc = 0, roi_pooled = 8, roi_topk = 100;   // roi_cap IS the record's row count (k_roi_tail, Engine.detect)
    float roi_score_thresh = 0.f, roi_nms_thresh = 0.9f, roi_reg_w[4] = {10.f, 10.f, 5.f, 5.f};
    float* roi_W = nullptr; float* roi_b = nullptr; float* roi_cls_w = nullptr; float* roi_cls_b = nullptr;
    float* roi_box_w = nullptr; float* roi_box_b = nullptr;
    float* roi_feat = nullptr; float* roi_h = nullptr;
    float* roi_hp = nullptr; int roi_ksplit = 1;       // K-split partial sums of the second-stage GEMM [roi_ksplit][roi_cap][roi_fc]
    float* det_boxes = nullptr; float* det_scores = nullptr; int64_t* det_src = nullptr; int32_t* det_count = nullptr;
    void* roi_ws = nullptr; size_t roi_ws_bytes = 0;
    // detector_postprocess inside the graph (ore_engine_detect_fwd): device {sx, sy, out_w, out_h} per image slot, the postprocessed
    // detections, and a pinned host word the count is copied into behind the graph
    float* post = nullptr; float post_host[4] = {1.f, 1.f, 0.f, 0.f};
    char* fin_pack = nullptr; int32_t* fin_count = nullptr;
    float* fin_boxes_of(size_t b) const { return (float*)(fin_pack + b * (size_t)roi_cap * 28); }
    float* fin_scores_of(size_t b) const { return fin_boxes_of(b) + (size_t)roi_cap * 4; }
    int32_t* pin_count = nullptr; int32_t* pin_count_dev = nullptr; float* pin_post = nullptr;
    // graph cache
    hipStream_t cap_stream = nullptr;
    struct GraphKey { int u8, H, W, B; hipGraphExec_t exec; };
    std::vector<GraphKey> graphs;
    Geo last{};
    double last_flops = 0.0;
    // optional per-launch HIP-event timing of the conv kernels (eager mode only; bench.py roofline leg)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct EvSpan { size_t a, b; double flops; double flops_exec; };   // algorithmic / executed (Winograd: 2.25x fewer multiplies)
    double last_profile_exec_flops = 0.0;
    std::vector<EvSpan> spans;
    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; ev_pool.push_back(e); }
        return ev_pool[ev_used++];
    }

    template <typename T> int dalloc(T** p, size_t n) {
        void* q = nullptr;
        ORE_HIP(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
        allocs.push_back(q);
        *p = (T*)q;
        return ORE_OK;
    }
    int upload(float** dst, const std::vector<float>& v) {
        int rc = dalloc(dst, v.size());
        if (rc) return rc;
        ORE_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return ORE_OK;
    }
    const HostTensor* get(const std::string& n) const {
        auto it = host.find(n);
        return it == host.end() ? nullptr : &it->second;
    }
};

namespace {

int need(const ore_engine* e, const std::string& name, const HostTensor** out, std::vector<int64_t> shape) {
    const HostTensor* t = e->get(name);
    if (!t) { ore_set_error("missing tensor '%s'", name.c_str()); return ORE_ENOENT; }
    if (!shape.empty() && t->shape != shape) {
        std::string a, b;
        for (auto d : t->shape) a += std::to_string(d) + ",";
        for (auto d : shape) b += std::to_string(d) + ",";
        ore_set_error("tensor '%s' has shape [%s], expected [%s]", name.c_str(), a.c_str(), b.c_str());
        return ORE_EINVAL;
    }
    *out = t;
    return ORE_OK;
}

// 3x3 stride-1 layers the Winograd kernel covers (ore_conv_wino.hip) also get their transformed weights, once
int make_bf16_w(ore_engine* e, Conv* c, const float* w_oihw) {
    if (!e->sb()) return ORE_OK;
    std::vector<uint16_t> h(ore_packed_weight_bf16_elems(c->Cout, c->Cin, c->k, c->k));
    int rc = ore_pack_conv_weight_bf16_host(w_oihw, c->Cout, c->Cin, c->k, c->k, h.data());
    if (rc) return rc;
    if ((rc = e->dalloc(&c->wh, h.size()))) return rc;
    ORE_HIP(hipMemcpy(c->wh, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return ORE_OK;
}

int make_wino(ore_engine* e, Conv* c) {
    if (e->sb()) return ORE_OK;
    if (c->k != 3 || c->stride != 1 || !ore_winograd_covers(c->Cout, c->Cin)) return ORE_OK;
    int rc = e->dalloc(&c->wino, ore_winograd_weight_floats(c->Cout, c->Cin));
    if (rc) return rc;
    if ((rc = ore_winograd_weight_fwd(c->w, c->Cout, c->Cin, c->wino, nullptr))) return rc;
    ORE_HIP(hipDeviceSynchronize());
    return ORE_OK;
}

// conv (no bias) + FrozenBN folded to (scale, shift) applied in the epilogue; weights stay unscaled.
int make_conv_bn(ore_engine* e, const std::string& name, int Cin, int Cout, int k, int stride, Conv* c) {
    const HostTensor *w, *g, *b, *m, *v;
    int rc;
    if ((rc = need(e, name + "/conv.weight", &w, {Cout, Cin, k, k}))) return rc;
    if ((rc = need(e, name + "/norm.weight", &g, {Cout}))) return rc;
    if ((rc = need(e, name + "/norm.bias", &b, {Cout}))) return rc;
    if ((rc = need(e, name + "/norm.running_mean", &m, {Cout}))) return rc;
    if ((rc = need(e, name + "/norm.running_var", &v, {Cout}))) return rc;
    std::vector<float> packed(ore_packed_weight_floats(Cout, Cin, k, k));
    ore_pack_conv_weight_host(w->v.data(), Cout, Cin, k, k, packed.data());
    std::vector<float> sc(Cout), sh(Cout);
    for (int n = 0; n < Cout; ++n) {  // d2z:layers/batch_norm.py:48-49
        sc[n] = g->v[n] * (1.0f / sqrtf(v->v[n] + 1e-5f));
        sh[n] = b->v[n] - m->v[n] * sc[n];
    }
    if ((rc = e->upload(&c->w, packed))) return rc;
    if ((rc = e->upload(&c->scale, sc))) return rc;
    if ((rc = e->upload(&c->shift, sh))) return rc;
    c->Cin = Cin; c->Cout = Cout; c->k = k; c->stride = stride; c->pad = k / 2; c->relu_cout = Cout;
    if ((rc = make_bf16_w(e, c, w->v.data()))) return rc;
    return make_wino(e, c);
}

int make_conv_bias(ore_engine* e, const std::string& name, int Cin, int Cout, int k, int relu, Conv* c) {
    const HostTensor *w, *b;
    int rc;
    if ((rc = need(e, name + ".weight", &w, {Cout, Cin, k, k}))) return rc;
    if ((rc = need(e, name + ".bias", &b, {Cout}))) return rc;
    std::vector<float> packed(ore_packed_weight_floats(Cout, Cin, k, k));
    ore_pack_conv_weight_host(w->v.data(), Cout, Cin, k, k, packed.data());
    if ((rc = e->upload(&c->w, packed))) return rc;
    if ((rc = e->upload(&c->shift, b->v))) return rc;
    c->scale = nullptr;
    c->Cin = Cin; c->Cout = Cout; c->k = k; c->stride = 1; c->pad = k / 2; c->relu_cout = relu ? Cout : 0;
    if ((rc = make_bf16_w(e, c, w->v.data()))) return rc;
    return make_wino(e, c);
}

int alloc_buf(ore_engine* e, Buf* b, size_t rows, int ld, bool activation = true) {
    b->ld = ld; b->rows = rows;
    // bf16 storage: half the bytes, plus slack for the 32-byte over-read of a row's last K chunk when Cin % 32 == 16 (zero weights)
    const size_t floats = (activation && e->sb()) ? (rows * ld + 1) / 2 + 16 : rows * ld;
    int rc = e->dalloc(&b->p, floats);
    if (rc) return rc;
    if (activation && e->sb()) ORE_HIP(hipMemset(b->p, 0, floats * sizeof(float)));
    return ORE_OK;
}

int pool_out(int n) { int o = (n - 3 + 1) / 2 + 1; if (n < 3) o = 1; if ((o - 1) * 2 >= n) --o; return o < 1 ? 1 : o; }

Geo make_geo(int B, int H, int W) {
    Geo g{};
    g.B = B; g.H = H; g.W = W;
    g.Hp = round_up(H, 32); g.Wp = round_up(W, 32);
    g.h[1] = g.Hp / 2; g.w[1] = g.Wp / 2;
    g.h[2] = (g.h[1] - 1) / 2 + 1; g.w[2] = (g.w[1] - 1) / 2 + 1;  // stem_3: 3x3 s2 p1
    for (int k = 3; k <= 5; ++k) { g.h[k] = pool_out(g.h[k - 1]); g.w[k] = pool_out(g.w[k - 1]); }
    return g;
}

struct Run {
    ore_engine* e; hipStream_t st; double flops = 0.0; int rc = ORE_OK; bool prof = false;
    // one launch over all pyramid levels (level-major rows)
    void conv_levels(const Conv& c, const float* in, int in_ld, int in_coff, int B, const int* H, const int* W, float* out,
                     int out_ld, int out_coff, int ep_stride = 0, const float* in_mul = nullptr, const float* in_add = nullptr,
                     int in_relu = 0, bool out_f32 = false, size_t wino_level_stride = 0, size_t w_level_stride = 0) {
        if (rc) return;
        ore_conv_desc d{};
        d.in = in; d.in_ld = in_ld; d.in_coff = in_coff; d.B = B; d.Cin = c.Cin;
        d.w = c.w; d.Cout = c.Cout; d.kh = d.kw = c.k; d.stride = 1; d.pad = c.pad;
        d.scale = c.scale; d.shift = c.shift; d.relu_cout = c.relu_cout;
        d.in_mul = in_mul; d.in_add = in_add; d.in_relu = in_relu;
        d.out = out; d.out_ld = out_ld; d.out_coff = out_coff;
        d.splitk = 0; d.workspace = e->ws; d.workspace_floats = e->ws_floats;
        d.w_wino = c.wino; d.w_wino_level_stride = (int64_t)wino_level_stride; d.w_level_stride = (int64_t)w_level_stride;
        if (e->sb()) { d.storage = out_f32 ? ORE_ST_BF16_F32OUT : ORE_ST_BF16; d.w = reinterpret_cast<const float*>(c.wh); }
        double rows = 0;
        for (int l = 0; l < 3; ++l) rows += (double)B * H[l] * W[l];
        const double fl = 2.0 * rows * (double)c.Cout * c.Cin * c.k * c.k;
        hipEvent_t ea = nullptr, eb = nullptr;
        size_t ia = 0;
        if (prof) { ia = e->ev_used; ea = e->next_event(); eb = e->next_event(); if (ea) (void)hipEventRecord(ea, st); }
        rc = ore_conv2d_levels_fwd(&d, 3, H, W, ep_stride, st);
        if (prof && ea && eb) { (void)hipEventRecord(eb, st); e->spans.push_back({ia, ia + 1, fl, ore_last_wino() ? fl / 2.25 : fl}); }
        flops += fl;
    }
    void conv(const Conv& c, const float* in, int in_ld, int in_coff, int B, int H, int W, float* out, int out_ld,
              int out_coff, const float* in_mul = nullptr, const float* in_add = nullptr, int in_relu = 0,
              const float* add = nullptr, int add_ld = 0, int add_coff = 0, float* colsum = nullptr, int* colsum_rows = nullptr) {
        if (rc) return;
        ore_conv_desc d{};
        d.in = in; d.in_ld = in_ld; d.in_coff = in_coff; d.B = B; d.H = H; d.W = W; d.Cin = c.Cin;
        d.w = c.w; d.Cout = c.Cout; d.kh = d.kw = c.k; d.stride = c.stride; d.pad = c.pad;
        d.scale = c.scale; d.shift = c.shift; d.relu_cout = c.relu_cout;
        d.in_mul = in_mul; d.in_add = in_add; d.in_relu = in_relu;
        d.add = add; d.add_ld = add_ld; d.add_coff = add_coff;
        d.out = out; d.out_ld = out_ld; d.out_coff = out_coff;
        d.splitk = 0; d.workspace = e->ws; d.workspace_floats = e->ws_floats;
        d.w_wino = c.wino;
        if (e->sb() && c.wh) { d.storage = ORE_ST_BF16; d.w = reinterpret_cast<const float*>(c.wh); }
        if (colsum) {
            const int rows = ore_conv_colsum_rows(&d);
            if ((size_t)rows * round_up(c.Cout, 16) <= e->colsum_floats) { d.colsum = colsum; *colsum_rows = rows; }
        }
        const int Ho = (H + 2 * c.pad - c.k) / c.stride + 1, Wo = (W + 2 * c.pad - c.k) / c.stride + 1;
        const double fl = 2.0 * B * Ho * Wo * (double)c.Cout * c.Cin * c.k * c.k;
        hipEvent_t ea = nullptr, eb = nullptr;
        size_t ia = 0;
        if (prof) { ia = e->ev_used; ea = e->next_event(); eb = e->next_event(); if (ea) (void)hipEventRecord(ea, st); }
        rc = ore_conv2d_fwd(&d, st);
        if (prof && ea && eb) { (void)hipEventRecord(eb, st); e->spans.push_back({ia, ia + 1, fl, ore_last_wino() ? fl / 2.25 : fl}); }
        flops += fl;
    }
};

int lvl_row0(const Geo& g, int l) {   // first row of pyramid level l (0 = p3) in the level-major matrices
    int r = 0;
    for (int j = 0; j < l; ++j) r += g.B * g.h[j + 3] * g.w[j + 3];
    return r;
}

int run_backbone(ore_engine* e, const void* img, int is_u8, const Geo& g, hipStream_t st, double* flops) {
    const ore_model_cfg& c = e->cfg;
    Run r{e, st};
    r.prof = e->profiling;
    const bool sb = e->sb();
    int rc = sb ? ore_stem1_bf16_fwd(img, is_u8, g.B, g.H, g.W, g.Hp, g.Wp, c.pixel_mean, c.pixel_std, e->stem1_w, e->stem1_scale,
                                     e->stem1_shift, c.stem_ch[0], reinterpret_cast<uint16_t*>(e->s1.p), e->s1.ld, 0, st)
                : ore_stem1_fwd(img, is_u8, g.B, g.H, g.W, g.Hp, g.Wp, c.pixel_mean, c.pixel_std, e->stem1_w, e->stem1_scale,
                                e->stem1_shift, c.stem_ch[0], e->s1.p, e->s1.ld, 0, st);
    if (rc) return rc;
    r.flops += 2.0 * g.B * g.h[1] * g.w[1] * 27.0 * c.stem_ch[0];
    r.conv(e->stem2, e->s1.p, e->s1.ld, 0, g.B, g.h[1], g.w[1], e->s2.p, e->s2.ld, 0);
    r.conv(e->stem3, e->s2.p, e->s2.ld, 0, g.B, g.h[1], g.w[1], e->cat[0].p, e->cat[0].ld, 0);
    bool lat_scaled_ok[3] = {false, false, false};         // this forward wrote lat_scaled[l] (bs = 1 with the fused column sums)
    bool pooled_next = false;                              // the gate kernel of the previous stage already pooled into this stage's buffer
    for (int s = 0; s < 4 && !r.rc; ++s) {
        auto& S = e->stage[s];
        const int k = s + 2;
        Buf& cat = e->cat[s];
        if (s > 0 && !pooled_next) {
            r.rc = sb ? ore_maxpool3x3s2_bf16_fwd(reinterpret_cast<const uint16_t*>(e->sout[s - 1].p), e->sout[s - 1].ld, 0, g.B, g.h[k - 1],
                                                  g.w[k - 1], S.in_ch, e->gate[s - 1], reinterpret_cast<uint16_t*>(cat.p), cat.ld, 0, st)
                      : ore_maxpool3x3s2_fwd(e->sout[s - 1].p, e->sout[s - 1].ld, 0, g.B, g.h[k - 1], g.w[k - 1], S.in_ch,
                                             e->gate[s - 1], cat.p, cat.ld, 0, st);
            if (r.rc) break;
        }
        int src = 0, dst = S.in_ch;
        for (int i = 0; i < c.layers_per_block; ++i) {
            r.conv(S.layer[i], cat.p, cat.ld, src, g.B, g.h[k], g.w[k], cat.p, cat.ld, dst);
            src = dst; dst += S.conv_ch;
        }
        int cs_rows = 0;
        r.conv(S.concat, cat.p, cat.ld, 0, g.B, g.h[k], g.w[k], e->sout[s].p, e->sout[s].ld, 0, nullptr, nullptr, 0, nullptr, 0, 0,
               g.B == 1 ? e->colsum : nullptr, &cs_rows);   // eSE average pool fused into the concat conv's epilogue
        if (r.rc) break;
        pooled_next = false;
        if (cs_rows > 0 && s < 3 && (s == 0 || e->lat_scaled[s - 1])) {
            // bs = 1, a stage followed by the max-pool: gate + the gate-scaled lateral weight of this stage + the pooled, gated input of
            // the next stage (written into its concat buffer) in one launch
            Buf& nx = e->cat[s + 1];
            const float* lw = s >= 1 ? e->lateral[s - 1].w : nullptr;
            const int lrows = s >= 1 ? round_up(e->lateral[s - 1].Cout, 16) : 0;
            if (s >= 1) lat_scaled_ok[s - 1] = true;
            r.rc = sb ? ore_ese_gate_pool_bf16_fwd(e->colsum, cs_rows, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s], lw, lrows,
                                                   s >= 1 ? reinterpret_cast<uint16_t*>(e->lat_scaled[s - 1]) : nullptr,
                                                   reinterpret_cast<const uint16_t*>(e->sout[s].p), e->sout[s].ld, 0, g.h[k], g.w[k],
                                                   reinterpret_cast<uint16_t*>(nx.p), nx.ld, 0, st)
                      : ore_ese_gate_pool_fwd(e->colsum, cs_rows, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s], lw, lrows,
                                              s >= 1 ? e->lat_scaled[s - 1] : nullptr, e->sout[s].p, e->sout[s].ld, 0, g.h[k], g.w[k], nx.p,
                                              nx.ld, 0, st);
            pooled_next = true;
        } else if (cs_rows > 0 && s >= 1 && e->lat_scaled[s - 1]) { // bs = 1, last stage: gate + the gate-scaled lateral weight in one launch
            lat_scaled_ok[s - 1] = true;
            r.rc = sb ? ore_ese_gate_scaled_weight_bf16_fwd(e->colsum, cs_rows, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s], e->ese_ws,
                                                            e->lateral[s - 1].w, round_up(e->lateral[s - 1].Cout, 16),
                                                            reinterpret_cast<uint16_t*>(e->lat_scaled[s - 1]), st)
                      : ore_ese_gate_scaled_weight_fwd(e->colsum, cs_rows, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s], e->ese_ws,
                                                       e->lateral[s - 1].w, round_up(e->lateral[s - 1].Cout, 16), e->lat_scaled[s - 1], st);
        } else if (cs_rows > 0)
            r.rc = ore_ese_gate_from_colsum_fwd(e->colsum, cs_rows, 1, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s], e->ese_ws, st);
        else if (sb) { ore_set_error("bf16-storage engine: the eSE pool must come fused from the concat conv (one image per pass)"); r.rc = ORE_EINVAL; }
        else
            r.rc = ore_ese_gate_fwd(e->sout[s].p, e->sout[s].ld, 0, g.B, g.h[k] * g.w[k], S.out_ch, S.fc_w, S.fc_b, e->gate[s],
                                    e->ese_ws, st);
    }
    // FPN top-down: level index 2 = p5, 1 = p4, 0 = p3; outputs land in the q half of pcat
    const int F = c.fpn_ch;
    float* lat[3];
    for (int l = 0; l < 3; ++l) lat[l] = e->at(e->lat_all.p, (size_t)lvl_row0(g, l) * F);
    const bool grouped = e->out_wino3 != nullptr || e->out_wh3 != nullptr;   // the three output convs run as one launch behind the lateral chain
    for (int l = 2; l >= 0 && !r.rc; --l) {
        const int k = l + 3, s = l + 1;
        const float* add = l < 2 ? lat[l + 1] : nullptr;
        if (lat_scaled_ok[l]) {                             // x * (g W): plain conv on the gate-scaled weights (k_conv_kw)
            Conv lc = e->lateral[l];
            lc.w = e->lat_scaled[l];
            lc.wh = reinterpret_cast<uint16_t*>(e->lat_scaled[l]);
            r.conv(lc, e->sout[s].p, e->sout[s].ld, 0, g.B, g.h[k], g.w[k], lat[l], F, 0, nullptr, nullptr, 0, add, F, 0);
        } else if (sb) {
            ore_set_error("bf16-storage engine: the lateral needs the gate-scaled weights (one image per pass)"); r.rc = ORE_EINVAL;
        } else {
            r.conv(e->lateral[l], e->sout[s].p, e->sout[s].ld, 0, g.B, g.h[k], g.w[k], lat[l], F, 0, e->gate[s], nullptr, 0,
                   add, F, 0);
        }
        if (!grouped)
            r.conv(e->output[l], lat[l], F, 0, g.B, g.h[k], g.w[k], e->at(e->pcat.p, (size_t)lvl_row0(g, l) * 2 * F), 2 * F, F);
    }
    if (grouped && !r.rc) {
        // fpn_output3/4/5 (d2z:modeling/backbone/fpn.py:139-145): three layers of one shape -> ONE Winograd launch, every block keeps the
        // weights of its level in registers (the levels' 100 / 25 / 7 batches of a 640 x 640 image in 3 rounds instead of 3 launches)
        const int H[3] = {g.h[3], g.h[4], g.h[5]}, W[3] = {g.w[3], g.w[4], g.w[5]};
        Conv oc = e->output[0];
        oc.wino = e->out_wino3; oc.shift = e->out_shift3;
        if (sb) oc.wh = e->out_wh3;
        r.conv_levels(oc, e->lat_all.p, F, 0, g.B, H, W, e->pcat.p, 2 * F, F, F, nullptr, nullptr, 0, false, sb ? 0 : e->out_wino_stride,
                      sb ? e->out_wh_stride : 0);
    }
    *flops = r.flops;
    return r.rc;
}

int run_heads(ore_engine* e, const Geo& g, hipStream_t st, double* flops) {
    const ore_model_cfg& c = e->cfg;
    const int F = c.fpn_ch;
    Run r{e, st};
    r.prof = e->profiling;
    const int H[3] = {g.h[3], g.h[4], g.h[5]}, W[3] = {g.w[3], g.w[4], g.w[5]};
    const int HW[3] = {H[0] * W[0], H[1] * W[1], H[2] * W[2]};
    for (int l = 0; l < 3; ++l)
        if (!e->support_set[l]) { ore_set_error("support prototype for level %d not set", l + 3); return ORE_EINVAL; }
    // query<->support depthwise correlation, all levels: attn -> channels [0,F) of pcat (q sits in [F,2F))
    const bool sb = e->sb();
    r.rc = sb ? ore_correlation_levels_bf16_fwd(reinterpret_cast<const uint16_t*>(e->pcat.p), 2 * F, F, g.B, 3, H, W, F, e->k11, e->k13, e->k31,
                                                reinterpret_cast<uint16_t*>(e->pcat.p), 2 * F, 0, st)
              : ore_correlation_levels_fwd(e->pcat.p, 2 * F, F, g.B, 3, H, W, F, e->k11, e->k13, e->k31, e->pcat.p, 2 * F, 0, st);
    if (r.rc) return r.rc;
    const double rows = (double)g.B * (HW[0] + HW[1] + HW[2]);
    r.flops += 2.0 * 8.0 * rows * F;
    r.conv_levels(e->conv3, e->pcat.p, 2 * F, 0, g.B, H, W, e->pos.p, F, 0);            // relu(conv3(cat(attn, q)))
    r.conv_levels(e->tower, e->pos.p, F, 0, g.B, H, W, e->tow.p, F, 0);                 // bbox_tower conv (+bias)
    if (r.rc) return r.rc;
    // GroupNorm statistics (fp32, from the fp32 or bf16 tower output), then GroupNorm + ReLU + the (l,t,r,b | hm) conv + per-level Scale
    // in ONE VALU kernel (k_head_pred: N = 5 is not MFMA work; it was 20 us on the matrix cores, + a materialised bf16 tensor in the
    // storage mode); the detection tail downstream is fp32 as always
    if (F == 128 && e->head_pred_valu) {
        // statistics, then ONE kernel for GroupNorm fold + ReLU + the (l,t,r,b | hm) conv + per-level Scale: its blocks combine the chunk
        // statistics of their own (level, image) under their tower loads
        r.rc = sb ? ore_head_pred_gn_bf16_fwd(reinterpret_cast<const uint16_t*>(e->tow.p), F, g.B, 3, H, W, 32, 1e-5f, e->gn_gamma, e->gn_beta,
                                              e->pred.w, e->pred.scale, e->pred.shift, 16, e->head.p, 8, e->gn_ws, st)
                  : ore_head_pred_gn_fwd(e->tow.p, F, g.B, 3, H, W, 32, 1e-5f, e->gn_gamma, e->gn_beta, e->pred.w, e->pred.scale,
                                         e->pred.shift, 16, e->head.p, 8, e->gn_ws, st);
        r.flops += 2.0 * rows * 5.0 * F * 9.0;
        *flops = r.flops;
        return r.rc;
    }
    r.rc = sb ? ore_groupnorm_affine_levels_bf16_fwd(reinterpret_cast<const uint16_t*>(e->tow.p), F, 0, g.B, 3, HW, F, 32, 1e-5f, e->gn_gamma,
                                                     e->gn_beta, e->gn_mul, e->gn_add, e->gn_ws, st)
              : ore_groupnorm_affine_levels_fwd(e->tow.p, F, 0, g.B, 3, HW, F, 32, 1e-5f, e->gn_gamma, e->gn_beta, e->gn_mul, e->gn_add,
                                                e->gn_ws, st);
    if (r.rc) return r.rc;
    if (sb) {
        if (!r.rc) r.rc = ore_groupnorm_apply_levels_bf16_fwd(reinterpret_cast<const uint16_t*>(e->tow.p), F, 0, g.B, 3, HW, F, e->gn_mul, e->gn_add, 1,
                                                              reinterpret_cast<uint16_t*>(e->tn.p), st);
        if (r.rc) return r.rc;
        r.conv_levels(e->pred, e->tn.p, F, 0, g.B, H, W, e->head.p, 8, 0, 16, nullptr, nullptr, 0, true);
        *flops = r.flops;
        return r.rc;
    }
    // (l,t,r,b | hm) conv with GN affine + ReLU on its input, per-level Scale in the epilogue
    r.conv_levels(e->pred, e->tow.p, F, 0, g.B, H, W, e->head.p, 8, 0, 16, e->gn_mul, e->gn_add, 1);
    *flops = r.flops;
    return r.rc;
}

int run_detect(ore_engine* e, const Geo& g, hipStream_t st, int b = 0) {        // image b of the batch
    const ore_model_cfg& c = e->cfg;
    const size_t cap = (size_t)3 * c.pre_topk;
    ore_detect_desc d{};
    d.n_levels = 3; d.head_ld = 8;
    for (int l = 0; l < 3; ++l) {
        d.head[l] = e->head.p + ((size_t)lvl_row0(g, l) + (size_t)b * g.h[l + 3] * g.w[l + 3]) * 8;
        d.H[l] = g.h[l + 3]; d.W[l] = g.w[l + 3]; d.stride[l] = c.strides[l];
    }
    d.score_thresh = c.score_thresh; d.pre_topk = c.pre_topk; d.nms_thresh = c.nms_thresh; d.post_topk = c.post_topk;
    d.pre_boxes = e->pre_boxes + b * cap * 4; d.pre_scores = e->pre_scores + b * cap; d.pre_loc = e->pre_loc + b * cap;
    d.pre_level = e->pre_level + b * cap; d.keep_idx = e->keep_idx + b * cap; d.counts = e->counts + b * 4;
    d.out_boxes = e->out_boxes + b * cap * 4; d.out_scores = e->out_scores + b * cap;
    d.workspace = e->det_ws; d.workspace_bytes = e->det_ws_bytes;
    return ore_detect_fwd(&d, st);
}

int run_roi(ore_engine* e, const Geo& g, hipStream_t st, double* flops, int b = 0) {   // image b (scratch roi_feat / roi_h / roi_ws is shared: stream order)
    const ore_model_cfg& c = e->cfg;
    const int F = c.fpn_ch;
    const size_t cap = (size_t)3 * c.pre_topk, rcap = (size_t)e->roi_cap;
    float* out_boxes = e->out_boxes + b * cap * 4;
    int32_t* counts = e->counts + b * 4;
    const float* feat[3]; int ld[3], coff[3], H[3], W[3]; float sc[3];
    for (int l = 0; l < 3; ++l) {
        feat[l] = e->at(e->pcat.p, ((size_t)lvl_row0(g, l) + (size_t)b * g.h[l + 3] * g.w[l + 3]) * 2 * F); ld[l] = 2 * F; coff[l] = F;
        H[l] = g.h[l + 3]; W[l] = g.w[l + 3]; sc[l] = 1.0f / (float)c.strides[l];
    }
    int rc = e->sb() ? ore_roi_align_bf16_fwd(reinterpret_cast<const uint16_t* const*>(feat), ld, coff, H, W, sc, 3, 3, F, e->roi_pooled, out_boxes,
                                              counts + 1, 0, e->roi_cap, e->roi_feat, st)
                     : ore_roi_align_fwd(feat, ld, coff, H, W, sc, 3, 3, F, e->roi_pooled, out_boxes, counts + 1, 0, e->roi_cap, e->roi_feat, st);
    if (rc) return rc;
    const int K = e->roi_pooled * e->roi_pooled * F;
    Run r{e, st};
    r.prof = e->profiling;
    Conv fc{};
    fc.w = e->roi_W; fc.scale = nullptr; fc.shift = e->roi_b; fc.Cin = K; fc.Cout = e->roi_fc; fc.k = 1; fc.stride = 1; fc.pad = 0;
    fc.relu_cout = e->roi_fc;
    // pre-composed DSA mix + flatten + fc1 (+ bias, ReLU).  The second-stage GEMM stays fp32 in every mode (include/ore_hip.h).
    const int S = e->roi_ksplit;
    if (S > 1) {
        // K split over blocks (oreconv::conv_gd_splitk): raw partial sums, added in slice order with bias and ReLU by the predictor kernel
        const double fl = 2.0 * (double)e->roi_cap * K * e->roi_fc;
        hipEvent_t ea = nullptr, eb = nullptr;
        size_t ia = 0;
        if (r.prof) { ia = e->ev_used; ea = e->next_event(); eb = e->next_event(); if (ea) (void)hipEventRecord(ea, st); }
        rc = oreconv::conv_gd_splitk(e->roi_feat, K, e->roi_W, e->roi_cap, K, e->roi_fc, e->roi_hp, S, st);
        if (r.prof && ea && eb) { (void)hipEventRecord(eb, st); e->spans.push_back({ia, ia + 1, fl, fl}); }
        if (rc) return rc;
        *flops = fl;
    } else {
        PrecisionScope fp32(ORE_CONV_FP32);
        r.conv(fc, e->roi_feat, K, 0, 1, 1, e->roi_cap, e->roi_h, e->roi_fc, 0);
        if (r.rc) return r.rc;
        *flops = r.flops;
    }
    return oreroi::roi_predict_post(S > 1 ? e->roi_hp : e->roi_h, e->roi_fc, S > 1 ? S : 0, S > 1 ? e->roi_b : nullptr, e->roi_cls_w, e->roi_cls_b,
                                    e->roi_box_w, e->roi_box_b, out_boxes, counts + 1, 0,
                                    e->roi_cap, e->roi_reg_w, (float)g.H, (float)g.W, e->roi_score_thresh, e->roi_nms_thresh, e->roi_topk,
                                    e->det_boxes + b * rcap * 4, e->det_scores + b * rcap, e->det_src + b * rcap, e->det_count + b * 4,
                                    e->post, e->fin_boxes_of(b), e->fin_scores_of(b), e->fin_count + b * 4,
                                    b == 0 ? e->pin_count_dev : nullptr, e->roi_ws, e->roi_ws_bytes, st);
}

}  // namespace

extern "C" int ore_engine_set_roi_head(ore_engine* e, const float* W_host, const float* b_host, int32_t fc_dim, int32_t pooled,
                                       const float* cls_w_host, const float* cls_b_host, const float* box_w_host,
                                       const float* box_b_host, const float* reg_weights4_host, float score_thresh, float nms_thresh,
                                       int32_t topk) {
    ORE_CHECK_ARG(e && e->finalized && W_host && b_host && cls_w_host && cls_b_host && box_w_host && box_b_host && reg_weights4_host,
                  "ore_engine_set_roi_head: engine must be finalized, pointers non-null");
    ORE_CHECK_ARG(fc_dim % 16 == 0 && fc_dim > 0 && pooled >= 1 && pooled <= 16 && nms_thresh > 0.f, "ore_engine_set_roi_head: bad args");
    // the engine's second stage always ends in the fused predict + tail kernels (they write the result record and the polled count):
    // refuse here what they do not cover instead of failing every image later (ore_roi_predict_post_fwd has the same bound)
    ORE_CHECK_ARG((size_t)fc_dim * 4 * 71 + 6 * 64 * 4 <= 60 * 1024,
                  "ore_engine_set_roi_head: fc width %d: the fused second-stage kernels cover fc widths up to 208 (FC_DIM/8 = 128 in finetune_vovnet.yaml)", fc_dim);
    static_assert(ORE_DET_RECORD_ROWS <= 512, "k_roi_tail<1024> sorts at most 512 candidates");
    ORE_HIP(hipSetDevice(e->device));
    const size_t K = (size_t)pooled * pooled * e->cfg.fpn_ch;
    int rc;
    auto up = [&](float** dst, const float* src, size_t n) -> int {
        if (!*dst && (rc = e->dalloc(dst, n))) return rc;
        ORE_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
        return ORE_OK;
    };
    if (e->roi_set && (e->roi_fc != fc_dim || e->roi_pooled != pooled)) { ore_set_error("ore_engine_set_roi_head: shape change not supported"); return ORE_EINVAL; }
    if ((rc = up(&e->roi_W, W_host, (size_t)fc_dim * K)) || (rc = up(&e->roi_b, b_host, fc_dim)) || (rc = up(&e->roi_cls_w, cls_w_host, 2 * (size_t)fc_dim)) ||
        (rc = up(&e->roi_cls_b, cls_b_host, 2)) || (rc = up(&e->roi_box_w, box_w_host, 4 * (size_t)fc_dim)) || (rc = up(&e->roi_box_b, box_b_host, 4)))
        return rc;
    if (!e->roi_set) {
        const size_t cap = (size_t)e->roi_cap;
        const size_t MB = (size_t)e->cfg.max_batch;
        // K split of the second-stage GEMM: 32 slices when the shape allows it (k_conv_gd's 80 x 64 tiles, whole 16-channel chunks per slice)
        e->roi_ksplit = (fc_dim % 64 == 0 && K % (16 * 32) == 0 && K >= 2048 && cap <= 512) ? 32 : 1;
        if (const char* ks = getenv("ORE_ROI_KSPLIT")) { const int v = atoi(ks); if (e->roi_ksplit > 1 && v >= 2 && (K / 16) % v == 0) e->roi_ksplit = v; }   // A/B aid
        if (e->roi_ksplit > 1 && (rc = e->dalloc(&e->roi_hp, (size_t)e->roi_ksplit * cap * fc_dim))) return rc;
        if ((rc = e->dalloc(&e->roi_feat, cap * K)) || (rc = e->dalloc(&e->roi_h, cap * fc_dim)) || (rc = e->dalloc(&e->det_boxes, MB * cap * 4)) ||
            (rc = e->dalloc(&e->det_scores, MB * cap)) || (rc = e->dalloc(&e->det_src, MB * cap)) || (rc = e->dalloc(&e->det_count, MB * 4))) return rc;
        e->roi_ws_bytes = ore_roi_predict_workspace_bytes(e->roi_cap);
        char* w = nullptr;
        if ((rc = e->dalloc(&w, e->roi_ws_bytes))) return rc;
        e->roi_ws = w;
        ORE_HIP(hipMemset(e->det_count, 0, MB * 4 * sizeof(int32_t)));
        // packed result record per image: [cap][4] boxes | [cap] scores | [cap] int64 classes (always 0: one foreground class), so that
        // ore_engine_detect_fwd hands the caller everything with ONE device-to-device copy
        char* pack = nullptr;
        if ((rc = e->dalloc(&e->post, 8)) || (rc = e->dalloc(&pack, MB * cap * 28)) || (rc = e->dalloc(&e->fin_count, MB * 4))) return rc;
        ORE_HIP(hipMemset(pack, 0, MB * cap * 28));
        e->fin_pack = pack;
        ORE_HIP(hipMemset(e->fin_count, 0, MB * 4 * sizeof(int32_t)));
        ORE_HIP(hipHostMalloc((void**)&e->pin_count, 64, hipHostMallocMapped));
        ORE_HIP(hipHostGetDevicePointer((void**)&e->pin_count_dev, e->pin_count, 0));
        e->pin_count[0] = 0; e->pin_count[2] = 0; e->pin_count[3] = 0;   // [0] count (device -> host), [2..3] result record address (host -> device)
        ORE_HIP(hipHostMalloc((void**)&e->pin_post, 64, hipHostMallocDefault));
        e->post_host[0] = e->post_host[1] = 1.f; e->post_host[2] = e->post_host[3] = 3.0e38f;   // identity until a size is requested
        ORE_HIP(hipMemcpy(e->post, e->post_host, 4 * sizeof(float), hipMemcpyHostToDevice));
    }
    e->roi_fc = fc_dim; e->roi_pooled = pooled; e->roi_topk = topk; e->roi_score_thresh = score_thresh; e->roi_nms_thresh = nms_thresh;
    for (int i = 0; i < 4; ++i) e->roi_reg_w[i] = reg_weights4_host[i];
    e->roi_set = true;
    for (auto& gk : e->graphs) (void)hipGraphExecDestroy(gk.exec);   // captured graphs did not contain the second stage
    e->graphs.clear();
    ORE_HIP(hipDeviceSynchronize());
    return ORE_OK;
}

extern "C" int ore_engine_create(const ore_model_cfg* cfg, int32_t device, ore_engine** out) {
    ORE_CHECK_ARG(cfg && out, "ore_engine_create: null");
    ORE_CHECK_ARG(cfg->layers_per_block >= 1 && cfg->layers_per_block <= 8, "layers_per_block");
    ORE_CHECK_ARG(cfg->fpn_ch % 16 == 0 && cfg->max_batch >= 1 && cfg->max_h >= 32 && cfg->max_w >= 32, "bad cfg");
    for (int i = 0; i < 3; ++i) ORE_CHECK_ARG(cfg->stem_ch[i] % 16 == 0, "stem channels must be multiples of 16");
    for (int i = 0; i < 4; ++i)
        ORE_CHECK_ARG(cfg->stage_conv_ch[i] % 16 == 0 && cfg->stage_out_ch[i] % 16 == 0, "stage channels must be multiples of 16");
    ORE_HIP(hipSetDevice(device));
    ore_engine* e = new ore_engine();
    e->cfg = *cfg; e->device = device;
    e->conv_precision = ore_conv_get_precision();
    { const char* hm = getenv("ORE_HEAD_MFMA"); e->head_pred_valu = !(hm && hm[0] == '1'); }
    *out = e;
    return ORE_OK;
}

extern "C" void ore_engine_destroy(ore_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    for (auto& g : e->graphs) hipGraphExecDestroy(g.exec);
    if (e->cap_stream) hipStreamDestroy(e->cap_stream);
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    for (void* p : e->allocs) hipFree(p);
    if (e->pin_count) hipHostFree(e->pin_count);
    if (e->pin_post) hipHostFree(e->pin_post);
    delete e;
}

extern "C" int ore_engine_set_tensor(ore_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    ORE_CHECK_ARG(e && name && data && ndim >= 0 && ndim <= 8, "ore_engine_set_tensor: bad args");
    ORE_CHECK_ARG(!e->finalized, "ore_engine_set_tensor: engine already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.v.assign(data, data + n);
    e->host[name] = std::move(t);
    return ORE_OK;
}

extern "C" int ore_engine_set_support(ore_engine* e, int32_t level, const float* proto, int32_t C, int32_t s) {
    ORE_CHECK_ARG(e && proto && level >= 3 && level <= 5 && C == e->cfg.fpn_ch && s >= 1, "ore_engine_set_support: bad args");
    const int l = level - 3;
    e->support[l].v.assign(proto, proto + (size_t)C * s * s);
    e->support[l].shape = {C, s, s};
    if (e->finalized) {
        ORE_HIP(hipSetDevice(e->device));
        float* tmp = nullptr;
        ORE_HIP(hipMalloc((void**)&tmp, (size_t)C * s * s * sizeof(float)));
        hipError_t err = hipMemcpy(tmp, proto, (size_t)C * s * s * sizeof(float), hipMemcpyHostToDevice);
        int rc = err == hipSuccess ? ore_support_kernels_fwd(tmp, C, s, e->k11 + (size_t)l * C, e->k13 + (size_t)l * C * 3,
                                                             e->k31 + (size_t)l * C * 3, nullptr) : ORE_EHIP;
        hipDeviceSynchronize();
        hipFree(tmp);
        if (rc) return rc;
    }
    e->support_set[l] = true;
    return ORE_OK;
}

extern "C" int ore_engine_finalize(ore_engine* e) {
    ORE_CHECK_ARG(e && !e->finalized, "ore_engine_finalize: bad state");
    ORE_CHECK_ARG(!e->sb() || e->cfg.max_batch == 1, "ore_engine_finalize: a bf16-storage engine takes one image per pass (max_batch = 1)");
    ORE_HIP(hipSetDevice(e->device));
    const ore_model_cfg& c = e->cfg;
    int rc;
    const std::string bu = "backbone.bottom_up.";
    {   // stem_1 stays OIHW-flat [Cout][27]
        const HostTensor *w, *g, *b, *m, *v;
        const int C0 = c.stem_ch[0];
        if ((rc = need(e, bu + "stem.stem_1/conv.weight", &w, {C0, 3, 3, 3}))) return rc;
        if ((rc = need(e, bu + "stem.stem_1/norm.weight", &g, {C0}))) return rc;
        if ((rc = need(e, bu + "stem.stem_1/norm.bias", &b, {C0}))) return rc;
        if ((rc = need(e, bu + "stem.stem_1/norm.running_mean", &m, {C0}))) return rc;
        if ((rc = need(e, bu + "stem.stem_1/norm.running_var", &v, {C0}))) return rc;
        std::vector<float> sc(C0), sh(C0);
        for (int n = 0; n < C0; ++n) { sc[n] = g->v[n] * (1.0f / sqrtf(v->v[n] + 1e-5f)); sh[n] = b->v[n] - m->v[n] * sc[n]; }
        if ((rc = e->upload(&e->stem1_w, w->v)) || (rc = e->upload(&e->stem1_scale, sc)) || (rc = e->upload(&e->stem1_shift, sh))) return rc;
    }
    if ((rc = make_conv_bn(e, bu + "stem.stem_2", c.stem_ch[0], c.stem_ch[1], 3, 1, &e->stem2))) return rc;
    if ((rc = make_conv_bn(e, bu + "stem.stem_3", c.stem_ch[1], c.stem_ch[2], 3, 2, &e->stem3))) return rc;
    int cin = c.stem_ch[2];
    for (int s = 0; s < 4; ++s) {
        auto& S = e->stage[s];
        const int k = s + 2;
        const std::string mod = "OSA" + std::to_string(k) + "_1";
        const std::string pre = bu + "stage" + std::to_string(k) + "." + mod + ".";
        S.in_ch = cin; S.conv_ch = c.stage_conv_ch[s]; S.out_ch = c.stage_out_ch[s];
        S.cat_ch = cin + c.layers_per_block * S.conv_ch;
        int ci = cin;
        for (int i = 0; i < c.layers_per_block; ++i) {
            if ((rc = make_conv_bn(e, pre + "layers." + std::to_string(i) + "." + mod + "_" + std::to_string(i), ci, S.conv_ch, 3, 1, &S.layer[i]))) return rc;
            ci = S.conv_ch;
        }
        if ((rc = make_conv_bn(e, pre + "concat." + mod + "_concat", S.cat_ch, S.out_ch, 1, 1, &S.concat))) return rc;
        const HostTensor *fw, *fb;
        if ((rc = need(e, pre + "ese.fc.weight", &fw, {S.out_ch, S.out_ch, 1, 1}))) return rc;
        if ((rc = need(e, pre + "ese.fc.bias", &fb, {S.out_ch}))) return rc;
        if ((rc = e->upload(&S.fc_w, fw->v)) || (rc = e->upload(&S.fc_b, fb->v))) return rc;
        cin = S.out_ch;
    }
    const int F = c.fpn_ch;
    for (int l = 0; l < 3; ++l) {
        const std::string st = std::to_string(l + 3);
        if ((rc = make_conv_bias(e, "backbone.fpn_lateral" + st, c.stage_out_ch[l + 1], F, 1, 0, &e->lateral[l]))) return rc;
        if ((rc = make_conv_bias(e, "backbone.fpn_output" + st, F, F, 3, 0, &e->output[l]))) return rc;
    }
    if (!e->sb() && (F == 64 || F == 128) && ore_winograd_covers(F, F)) {
        // the three output convs are one shape: their Winograd weights and biases side by side, one launch over the three levels
        e->out_wino_stride = ore_winograd_weight_floats(F, F);
        if ((rc = e->dalloc(&e->out_wino3, 3 * e->out_wino_stride)) || (rc = e->dalloc(&e->out_shift3, (size_t)3 * F))) return rc;
        for (int l = 0; l < 3; ++l) {
            if ((rc = ore_winograd_weight_fwd(e->output[l].w, F, F, e->out_wino3 + l * e->out_wino_stride, nullptr))) return rc;
            ORE_HIP(hipMemcpy(e->out_shift3 + (size_t)l * F, e->output[l].shift, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice));
        }
        ORE_HIP(hipDeviceSynchronize());
    } else if (e->sb() && (F == 64 || F == 128)) {
        // bf16 storage: the same grouping on the weight-stationary kernel (one tile per block, the block loads its level's weights)
        e->out_wh_stride = ore_packed_weight_bf16_elems(F, F, 3, 3);
        if ((rc = e->dalloc(&e->out_wh3, 3 * e->out_wh_stride)) || (rc = e->dalloc(&e->out_shift3, (size_t)3 * F))) return rc;
        for (int l = 0; l < 3; ++l) {
            ORE_HIP(hipMemcpy(e->out_wh3 + l * e->out_wh_stride, e->output[l].wh, e->out_wh_stride * sizeof(uint16_t), hipMemcpyDeviceToDevice));
            ORE_HIP(hipMemcpy(e->out_shift3 + (size_t)l * F, e->output[l].shift, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice));
        }
    }
    if ((rc = make_conv_bias(e, "conv3", 2 * F, F, 1, 1, &e->conv3))) return rc;
    const std::string hp = "proposal_generator.centernet_head.";
    if ((rc = make_conv_bias(e, hp + "bbox_tower.0", F, F, 3, 0, &e->tower))) return rc;
    {
        const HostTensor *g, *b, *wr, *br, *wh, *bh;
        if ((rc = need(e, hp + "bbox_tower.1.weight", &g, {F})) || (rc = need(e, hp + "bbox_tower.1.bias", &b, {F}))) return rc;
        if ((rc = e->upload(&e->gn_gamma, g->v)) || (rc = e->upload(&e->gn_beta, b->v))) return rc;
        if ((rc = need(e, hp + "bbox_pred.weight", &wr, {4, F, 3, 3})) || (rc = need(e, hp + "bbox_pred.bias", &br, {4}))) return rc;
        if ((rc = need(e, hp + "agn_hm.weight", &wh, {1, F, 3, 3})) || (rc = need(e, hp + "agn_hm.bias", &bh, {1}))) return rc;
        std::vector<float> w5(wr->v);
        w5.insert(w5.end(), wh->v.begin(), wh->v.end());
        std::vector<float> packed(ore_packed_weight_floats(5, F, 3, 3));
        ore_pack_conv_weight_host(w5.data(), 5, F, 3, 3, packed.data());
        Conv& p = e->pred;
        if ((rc = e->upload(&p.w, packed))) return rc;
        p.Cin = F; p.Cout = 5; p.k = 3; p.stride = 1; p.pad = 1; p.relu_cout = 4;
        if ((rc = make_bf16_w(e, &p, w5.data()))) return rc;
        std::vector<float> scale(3 * 16, 0.f), shift(3 * 16, 0.f);
        for (int l = 0; l < 3; ++l) {
            const HostTensor* sc;
            if ((rc = need(e, hp + "scales." + std::to_string(l) + ".scale", &sc, {1}))) return rc;
            const float s = sc->v[0];
            // reg = relu(scale_l * (conv + bias)); hm = conv + bias   (centernet_head.py:150-159)
            for (int j = 0; j < 4; ++j) { scale[l * 16 + j] = s; shift[l * 16 + j] = br->v[j] * s; }
            scale[l * 16 + 4] = 1.0f; shift[l * 16 + 4] = bh->v[0];
        }
        if ((rc = e->upload(&p.scale, scale)) || (rc = e->upload(&p.shift, shift))) return rc;
    }
    // ---- buffers for the largest padded geometry
    const Geo g = make_geo(c.max_batch, c.max_h, c.max_w);
    const size_t B = g.B;
    e->img_bytes = B * 3 * (size_t)c.max_h * c.max_w * sizeof(float);
    ORE_HIP(hipMalloc(&e->img_in, e->img_bytes));
    e->allocs.push_back(e->img_in);
    if ((rc = alloc_buf(e, &e->s1, B * g.h[1] * g.w[1], c.stem_ch[0]))) return rc;
    if ((rc = alloc_buf(e, &e->s2, B * g.h[1] * g.w[1], c.stem_ch[1]))) return rc;
    size_t cmax = 0, cs_need = 0;
    for (int s = 0; s < 4; ++s) {
        const int k = s + 2;
        const size_t M = B * g.h[k] * g.w[k];
        if ((rc = alloc_buf(e, &e->cat[s], M, e->stage[s].cat_ch))) return rc;
        if ((rc = alloc_buf(e, &e->sout[s], M, e->stage[s].out_ch))) return rc;
        if ((rc = e->dalloc(&e->gate[s], B * e->stage[s].out_ch))) return rc;
        if (s >= 1 && (rc = e->dalloc(&e->lat_scaled[s - 1], (size_t)round_up(e->cfg.fpn_ch, 16) * e->stage[s].out_ch))) return rc;
        if ((size_t)e->stage[s].out_ch > cmax) cmax = e->stage[s].out_ch;
        const size_t cs = (M / 16 + 1) * (size_t)e->stage[s].out_ch;     // worst case: 16-row tiles
        if (cs > cs_need) cs_need = cs;
    }
    size_t rows_all = 0;
    for (int l = 0; l < 3; ++l) rows_all += (size_t)B * g.h[l + 3] * g.w[l + 3];
    if ((rc = alloc_buf(e, &e->lat_all, rows_all, F))) return rc;        // the three inner (top-down) maps, level-major rows like pcat
    if ((rc = alloc_buf(e, &e->pcat, rows_all, 2 * F)) || (rc = alloc_buf(e, &e->pos, rows_all, F)) ||
        (rc = alloc_buf(e, &e->tow, rows_all, F)) || (rc = alloc_buf(e, &e->head, rows_all, 8, false))) return rc;
    if (e->sb() && (rc = alloc_buf(e, &e->tn, rows_all, F))) return rc;
    ORE_HIP(hipMemset(e->head.p, 0, rows_all * 8 * sizeof(float)));
    if ((rc = e->dalloc(&e->gn_mul, 3 * B * F)) || (rc = e->dalloc(&e->gn_add, 3 * B * F))) return rc;
    if ((rc = e->dalloc(&e->k11, (size_t)3 * F)) || (rc = e->dalloc(&e->k13, (size_t)3 * F * 3)) || (rc = e->dalloc(&e->k31, (size_t)3 * F * 3))) return rc;
    e->ws_floats = ore_conv_workspace_floats();
    if ((rc = e->dalloc(&e->ws, e->ws_floats))) return rc;
    ORE_HIP(hipMemset(e->ws, 0, e->ws_floats * sizeof(float)));   // arrival counters start at zero and reset themselves
    if ((rc = e->dalloc(&e->ese_ws, B * (ORE_ESE_PARTS + 1) * cmax))) return rc;
    e->colsum_floats = cs_need;
    if ((rc = e->dalloc(&e->colsum, cs_need))) return rc;
    if ((rc = e->dalloc(&e->gn_ws, (rows_all / 64 + 3 * B + 8) * 64 * 2))) return rc;
    const size_t cap = (size_t)3 * c.pre_topk;
    const size_t MB = (size_t)c.max_batch;                  // detection outputs: one set per image of a batch, image b at b * stride
    if ((rc = e->dalloc(&e->pre_boxes, MB * cap * 4)) || (rc = e->dalloc(&e->pre_scores, MB * cap)) || (rc = e->dalloc(&e->pre_loc, MB * cap)) ||
        (rc = e->dalloc(&e->pre_level, MB * cap)) || (rc = e->dalloc(&e->keep_idx, MB * cap)) || (rc = e->dalloc(&e->counts, MB * 4)) ||
        (rc = e->dalloc(&e->out_boxes, MB * cap * 4)) || (rc = e->dalloc(&e->out_scores, MB * cap))) return rc;
    ORE_HIP(hipMemset(e->counts, 0, MB * 4 * sizeof(int32_t)));
    e->det_ws_bytes = ore_detect_workspace_bytes(3, c.pre_topk);
    char* dws = nullptr;
    if ((rc = e->dalloc(&dws, e->det_ws_bytes))) return rc;
    e->det_ws = dws;
    ORE_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    e->finalized = true;
    for (int l = 0; l < 3; ++l)
        if (!e->support[l].v.empty()) {
            e->support_set[l] = false;
            if ((rc = ore_engine_set_support(e, l + 3, e->support[l].v.data(), (int)e->support[l].shape[0], (int)e->support[l].shape[1]))) return rc;
        }
    e->host.clear();
    ORE_HIP(hipDeviceSynchronize());
    return ORE_OK;
}

static int check_geo(ore_engine* e, int B, int H, int W) {
    ORE_CHECK_ARG(e && e->finalized, "engine not finalized");
    ORE_CHECK_ARG(B >= 1 && B <= e->cfg.max_batch && H >= 1 && W >= 1 && round_up(H, 32) <= round_up(e->cfg.max_h, 32) &&
                      round_up(W, 32) <= round_up(e->cfg.max_w, 32) &&
                      (size_t)B * round_up(H, 32) * round_up(W, 32) <= (size_t)e->cfg.max_batch * round_up(e->cfg.max_h, 32) * round_up(e->cfg.max_w, 32),
                  "input %dx%dx%d exceeds the engine's max %dx%dx%d", B, H, W, e->cfg.max_batch, e->cfg.max_h, e->cfg.max_w);
    return ORE_OK;
}

extern "C" int ore_engine_backbone_fwd(ore_engine* e, const void* img, int32_t is_u8, int32_t B, int32_t H, int32_t W, void* stream) {
    int rc = check_geo(e, B, H, W);
    if (rc) return rc;
    ORE_CHECK_ARG(img, "null image");
    const Geo g = make_geo(B, H, W);
    e->last = g;
    PrecisionScope ps(e->sb() ? ORE_CONV_FP32 : e->conv_precision);
    return run_backbone(e, img, is_u8, g, (hipStream_t)stream, &e->last_flops);
}

extern "C" int ore_engine_eval_fwd(ore_engine* e, const void* img, int32_t is_u8, int32_t H, int32_t W, int32_t use_graph,
                                   void* stream) {
    return ore_engine_eval_batch_fwd(e, img, is_u8, 1, H, W, use_graph, stream);
}

extern "C" int ore_engine_detect_begin(ore_engine* e, const void* img, int32_t is_u8, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                                       void* out_record, void* stream) {
    ORE_CHECK_ARG(e && e->roi_set && out_record && out_h >= 1 && out_w >= 1, "ore_engine_detect_begin: needs the second stage (ore_engine_set_roi_head)");
    hipStream_t st = (hipStream_t)stream;
    // detector_postprocess parameters: sx = out_w / W, sy = out_h / H evaluated like the reference's Python floats, rounded once to
    // fp32 (what `boxes *= scale` does to a float32 tensor); uploaded only when the requested size changes
    const float pp[4] = {(float)((double)out_w / (double)W), (float)((double)out_h / (double)H), (float)out_w, (float)out_h};
    if (memcmp(pp, e->post_host, sizeof(pp)) != 0) {
        ORE_HIP(hipStreamSynchronize(st));                   // the pinned staging word may still be in flight from the previous change
        memcpy(e->pin_post, pp, sizeof(pp));
        ORE_HIP(hipMemcpyAsync(e->post, e->pin_post, sizeof(pp), hipMemcpyHostToDevice, st));
        memcpy(e->post_host, pp, sizeof(pp));
    }
    // the caller's freshly allocated result record is filled by the LAST KERNEL of the graph: its address travels in the pinned,
    // device-mapped word behind the count (k_roi_tail reads it with a system-scope load), so nothing is queued behind the replay
    volatile unsigned long long* rec_word = reinterpret_cast<volatile unsigned long long*>(e->pin_count) + 1;
    ORE_CHECK_ARG(*rec_word == 0ull, "ore_engine_detect_begin: a pass is already pending on this engine (ore_engine_detect_end first)");
    *rec_word = (unsigned long long)(uintptr_t)out_record;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    volatile int32_t* cnt_word = e->pin_count;
    *cnt_word = -1;                                          // sentinel: the last kernel of the graph overwrites it with the count (>= 0)
    __atomic_thread_fence(__ATOMIC_RELEASE);
    const int rc = ore_engine_eval_batch_fwd(e, img, is_u8, 1, H, W, 1, stream);
    if (rc) *rec_word = 0ull;
    return rc;
}

extern "C" int ore_engine_detect_end(ore_engine* e, void* stream, int32_t* n_det) {
    ORE_CHECK_ARG(e && e->roi_set && n_det, "ore_engine_detect_end: null");
    hipStream_t st = (hipStream_t)stream;
    volatile unsigned long long* rec_word = reinterpret_cast<volatile unsigned long long*>(e->pin_count) + 1;
    volatile int32_t* cnt_word = e->pin_count;
    ORE_CHECK_ARG(*rec_word != 0ull, "ore_engine_detect_end: no ore_engine_detect_begin is pending on this engine");
    // Wait for the count word instead of the stream: the kernel writes it as its very last action (after its result stores), and a
    // poll of pinned memory sees it a wake-up latency earlier than hipStreamSynchronize returns.  Everything the caller does with the
    // record afterwards is stream-ordered behind the kernel anyway.  Bounded spin, then the ordinary synchronise (also the error path).
    int32_t n = -1;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        n = *cnt_word;
        if (n >= 0) break;
        __builtin_ia32_pause();
    }
    if (n < 0) {
        const hipError_t se = hipStreamSynchronize(st);
        if (se != hipSuccess) { *rec_word = 0ull; ore_set_error("ore_engine_detect_end: %s", hipGetErrorString(se)); return ORE_EHIP; }
        n = *cnt_word;
        if (n < 0) { *rec_word = 0ull; ore_set_error("ore_engine_detect_end: the pass finished without writing the detection count"); return ORE_EHIP; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    *rec_word = 0ull;                                        // no later replay of this engine may write into the caller's tensor
    *n_det = n;
    return ORE_OK;
}

extern "C" int ore_engine_detect_fwd(ore_engine* e, const void* img, int32_t is_u8, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                                     void* out_record, void* stream, int32_t* n_det) {
    ORE_CHECK_ARG(n_det, "ore_engine_detect_fwd: null count");
    const int rc = ore_engine_detect_begin(e, img, is_u8, H, W, out_h, out_w, out_record, stream);
    return rc ? rc : ore_engine_detect_end(e, stream, n_det);
}

extern "C" int ore_engine_eval_batch_fwd(ore_engine* e, const void* img, int32_t is_u8, int32_t B, int32_t H, int32_t W,
                                         int32_t use_graph, void* stream) {
    int rc = check_geo(e, B, H, W);
    if (rc) return rc;
    ORE_CHECK_ARG(img, "null image");
    hipStream_t st = (hipStream_t)stream;
    const Geo g = make_geo(B, H, W);
    e->last = g;
    PrecisionScope ps(e->sb() ? ORE_CONV_FP32 : e->conv_precision);
    const size_t bytes = (size_t)B * 3 * H * W * (is_u8 ? 1 : 4);
    ORE_CHECK_ARG(bytes <= e->img_bytes, "image batch exceeds the engine's input buffer");
    if (img != e->img_in) ORE_HIP(hipMemcpyAsync(e->img_in, img, bytes, hipMemcpyDefault, st));   // device or (pinned / pageable) host source
    auto body = [&](hipStream_t s, double* fl) -> int {
        double f1 = 0, f2 = 0;
        int r = run_backbone(e, e->img_in, is_u8, g, s, &f1);
        double f3 = 0;
        if (!r) r = run_heads(e, g, s, &f2);
        for (int b = 0; b < B && !r; ++b) {                 // the detection tail and the second stage are per image
            double fb = 0;
            r = run_detect(e, g, s, b);
            if (!r && e->roi_set) r = run_roi(e, g, s, &fb, b);
            f3 += fb;
        }
        *fl = f1 + f2 + f3;
        return r;
    };
    if (!use_graph) return body(st, &e->last_flops);
    for (auto& k : e->graphs)
        if (k.u8 == is_u8 && k.H == H && k.W == W && k.B == B) {
            ORE_HIP(hipGraphLaunch(k.exec, st));
            return ORE_OK;
        }
    // capture once on the engine's own stream, replay on the caller's
    hipGraph_t graph = nullptr;
    const bool was_prof = e->profiling;
    e->profiling = false;
    ORE_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
    rc = body(e->cap_stream, &e->last_flops);
    e->profiling = was_prof;
    hipError_t ce = hipStreamEndCapture(e->cap_stream, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) { ore_set_error("hipStreamEndCapture -> %s", hipGetErrorString(ce)); return ORE_EHIP; }
    hipGraphExec_t exec = nullptr;
    ORE_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    e->graphs.push_back({is_u8, H, W, B, exec});
    ORE_HIP(hipGraphLaunch(exec, st));
    return ORE_OK;
}

extern "C" double ore_engine_last_flops(ore_engine* e) { return e ? e->last_flops : 0.0; }
extern "C" double ore_engine_profile_executed_flops(ore_engine* e) { return e ? e->last_profile_exec_flops : 0.0; }

extern "C" int ore_engine_set_profiling(ore_engine* e, int32_t enable) {
    ORE_CHECK_ARG(e, "ore_engine_set_profiling: null");
    e->profiling = enable != 0;
    e->ev_used = 0;
    e->spans.clear();
    return ORE_OK;
}

extern "C" int ore_engine_read_profile(ore_engine* e, double* conv_ms, double* conv_flops, int32_t* n_launches) {
    ORE_CHECK_ARG(e && conv_ms && conv_flops && n_launches, "ore_engine_read_profile: null");
    ORE_HIP(hipDeviceSynchronize());
    double ms = 0.0, fl = 0.0, fe = 0.0;
    for (auto& s : e->spans) {
        float t = 0.f;
        ORE_HIP(hipEventElapsedTime(&t, e->ev_pool[s.a], e->ev_pool[s.b]));
        ms += t; fl += s.flops; fe += s.flops_exec;
    }
    *conv_ms = ms; *conv_flops = fl; *n_launches = (int32_t)e->spans.size();
    e->last_profile_exec_flops = fe;
    e->ev_used = 0;
    e->spans.clear();
    return ORE_OK;
}

__global__ void k_noop() {}

// Cost of bracketing one launch with hipEvents, beyond the launch itself: median time of (event, 1 empty launch, event) minus the
// marginal cost of one more empty launch inside the bracket, i.e. 2*T(1) - T(2).  bench.py subtracts launches x this value so that
// the event-based kernel time agrees with the durations rocprofv3 reports for the same kernels.
extern "C" int ore_event_pair_overhead_us(void* stream, int32_t reps, double* median_us) {
    ORE_CHECK_ARG(median_us && reps >= 1 && reps <= 4096, "ore_event_pair_overhead_us: bad args");
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)reps);
    for (auto& e : ev) ORE_HIP(hipEventCreate(&e));
    double T[2] = {0.0, 0.0};
    for (int nk = 1; nk <= 2; ++nk) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, st);
        for (int i = 0; i < reps; ++i) {
            ORE_HIP(hipEventRecord(ev[2 * i], st));
            for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, st);
            ORE_HIP(hipEventRecord(ev[2 * i + 1], st));
        }
        ORE_HIP(hipStreamSynchronize(st));
        std::vector<float> t((size_t)reps);
        for (int i = 0; i < reps; ++i) ORE_HIP(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
        std::sort(t.begin(), t.end());
        T[nk - 1] = 1e3 * (double)t[t.size() / 2];
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    const double o = 2.0 * T[0] - T[1];
    *median_us = o > 0.0 ? o : 0.0;
    return ORE_OK;
}

extern "C" int ore_engine_buffer(ore_engine* e, const char* name, void** ptr, int64_t dims[4]) {
    ORE_CHECK_ARG(e && e->finalized && name && ptr && dims, "ore_engine_buffer: bad args");
    const Geo& g = e->last;
    const std::string n(name);
    const int F = e->cfg.fpn_ch;
    auto set = [&](void* p, int64_t rows, int64_t ch, int64_t ld, int64_t coff) {
        *ptr = p; dims[0] = rows; dims[1] = ch; dims[2] = ld; dims[3] = coff; return ORE_OK;
    };
    const int64_t cap = (int64_t)3 * e->cfg.pre_topk;
    if (n == "stem1") return set(e->s1.p, (int64_t)g.B * g.h[1] * g.w[1], e->cfg.stem_ch[0], e->s1.ld, 0);
    if (n == "stem2") return set(e->s2.p, (int64_t)g.B * g.h[1] * g.w[1], e->cfg.stem_ch[1], e->s2.ld, 0);
    if (n == "stem3") return set(e->cat[0].p, (int64_t)g.B * g.h[2] * g.w[2], e->cfg.stem_ch[2], e->cat[0].ld, 0);
    for (int s = 0; s < 4; ++s) {
        const int64_t rows = (int64_t)g.B * g.h[s + 2] * g.w[s + 2];
        if (n == "stage" + std::to_string(s + 2)) return set(e->sout[s].p, rows, e->stage[s].out_ch, e->sout[s].ld, 0);  // pre-gate
        if (n == "gate" + std::to_string(s + 2)) return set(e->gate[s], g.B, e->stage[s].out_ch, e->stage[s].out_ch, 0);
        if (n == "cat" + std::to_string(s + 2)) return set(e->cat[s].p, rows, e->stage[s].cat_ch, e->cat[s].ld, 0);
    }
    for (int l = 0; l < 3; ++l) {
        const std::string k = std::to_string(l + 3);
        const int64_t rows = (int64_t)g.B * g.h[l + 3] * g.w[l + 3];
        const size_t r0 = (size_t)lvl_row0(g, l);
        if (n == "p" + k) return set(e->at(e->pcat.p, r0 * 2 * F), rows, F, 2 * F, F);
        if (n == "attn" + k) return set(e->at(e->pcat.p, r0 * 2 * F), rows, F, 2 * F, 0);
        if (n == "lat" + k) return set(e->at(e->lat_all.p, r0 * F), rows, F, F, 0);
        if (n == "pos" + k) return set(e->at(e->pos.p, r0 * F), rows, F, F, 0);
        if (n == "tower" + k) return set(e->at(e->tow.p, r0 * F), rows, F, F, 0);
        if (n == "head" + k) return set(e->head.p + r0 * 8, rows, 5, 8, 0);
    }
    // per-image detection outputs: "name" = image 0, "name#b" = image b of the last batch
    size_t ib = 0;
    std::string base = n;
    { const size_t h = n.find('#'); if (h != std::string::npos) { base = n.substr(0, h); ib = (size_t)std::atoi(n.c_str() + h + 1); } }
    if (ib >= (size_t)e->cfg.max_batch) { ore_set_error("ore_engine_buffer: image index in '%s' exceeds max_batch", name); return ORE_EINVAL; }
    const size_t ucap = (size_t)cap, rcap = (size_t)e->roi_cap;
    if (base == "pre_boxes") return set(e->pre_boxes + ib * ucap * 4, cap, 4, 4, 0);
    if (base == "pre_scores") return set(e->pre_scores + ib * ucap, cap, 1, 1, 0);
    if (base == "pre_loc") return set(e->pre_loc + ib * ucap, cap, 1, 1, 0);
    if (base == "pre_level") return set(e->pre_level + ib * ucap, cap, 1, 1, 0);
    if (base == "keep_idx") return set(e->keep_idx + ib * ucap, cap, 1, 1, 0);
    if (base == "counts") return set(e->counts + ib * 4, 4, 1, 1, 0);
    if (base == "out_boxes") return set(e->out_boxes + ib * ucap * 4, cap, 4, 4, 0);
    if (base == "out_scores") return set(e->out_scores + ib * ucap, cap, 1, 1, 0);
    if (e->roi_set) {
        if (base == "det_boxes") return set(e->det_boxes + ib * rcap * 4, e->roi_cap, 4, 4, 0);
        if (base == "det_scores") return set(e->det_scores + ib * rcap, e->roi_cap, 1, 1, 0);
        if (base == "det_src") return set(e->det_src + ib * rcap, e->roi_cap, 1, 1, 0);
        if (base == "det_count") return set(e->det_count + ib * 4, 4, 1, 1, 0);
        if (base == "final_boxes") return set(e->fin_boxes_of(ib), e->roi_cap, 4, 4, 0);
        if (base == "final_scores") return set(e->fin_scores_of(ib), e->roi_cap, 1, 1, 0);
        if (base == "final_count") return set(e->fin_count + ib * 4, 4, 1, 1, 0);
        if (n == "roi_h") return set(e->roi_h, e->roi_cap, e->roi_fc, e->roi_fc, 0);
        // the predictor's per-ROI rows BEFORE the score filter / NMS (row r = proposal r of the image the second stage ran last on)
        const oreroi::PredictWs L = oreroi::predict_ws_layout(e->roi_cap);
        if (n == "roi_raw_boxes") return set((float*)((char*)e->roi_ws + L.raw_boxes), e->roi_cap, 4, 4, 0);
        if (n == "roi_raw_scores") return set((float*)((char*)e->roi_ws + L.raw_scores), e->roi_cap, 1, 1, 0);
        if (n == "roi_ok") return set((float*)((char*)e->roi_ws + L.ok), e->roi_cap, 1, 1, 0);
    }
    ore_set_error("ore_engine_buffer: unknown buffer '%s'", name);
    return ORE_ENOENT;
}

extern "C" int32_t ore_engine_buffer_is_bf16(ore_engine* e, const char* name) {
    if (!e || !name || !e->sb()) return 0;
    const std::string n(name);
    for (const char* pre : {"stem", "stage", "cat", "attn", "lat", "pos", "tower"})
        if (n.rfind(pre, 0) == 0) return 1;
    return n.size() == 2 && n[0] == 'p' ? 1 : 0;                 // "p3" .. "p5"
}
