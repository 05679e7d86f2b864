// MuseTalk per-frame generator on the HIP engine: conditional U-Net + VAE decoder as a static launch program.
//
// Reference call sites: avatars/musetalk_avatar.py:130-152 (MuseReal.inference_batch),
// avatars/musetalk/models/unet.py:12-46 (PositionalEncoding, diffusers UNet2DConditionModel),
// avatars/musetalk/models/vae.py:96-108 (decode_latents).  The diffusers graph itself is restated in
// oracle/musetalk_oracle.py (the checker); this file is its device program: every conv / linear is one launch of
// the MFMA conv kernels (conv3_mfma.hip / conv_mfma.hip: a Linear over channels is a 1x1 conv on the token map),
// GroupNorm / LayerNorm / attention / GEGLU are nn_kernels.hip.  Load-time folding:
//   * the timestep is the constant 0 (musetalk_avatar.py:61,148-150): time_embedding and every resnet's
//     time_emb_proj collapse into the bias of that resnet's conv1;
//   * the attention scale d^-0.5 goes into to_q; heads of 40 channels are padded to 48 (three channel blocks)
//     by permuting to_q/to_k/to_v rows and to_out columns;
//   * 1/scaling_factor (vae.py:103) goes into post_quant_conv;
//   * torch.cat([h, skip]) of the up path is a channel-block range of one buffer.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/ltk.h"
#include "conv_mfma.h"
#include "tune.h"
#include "musetalk.h"
#include "nn_kernels.h"
#include "misc_kernels.h"

namespace ltk {

namespace {

struct SD {
    const ltk_named_tensor* t;
    int n;
    std::string err;
    const float* get(const std::string& name, size_t expect) {
        for (int i = 0; i < n; ++i)
            if (name == t[i].name) {
                size_t cnt = 1;
                for (int d = 0; d < t[i].ndim; ++d) cnt *= (size_t)t[i].shape[d];
                if (cnt != expect) { err = "tensor " + name + " has " + std::to_string(cnt) + " elements, expected " + std::to_string(expect); return nullptr; }
                return t[i].data;
            }
        err = "state_dict is missing " + name;
        return nullptr;
    }
    bool has(const std::string& name) const {
        for (int i = 0; i < n; ++i) if (name == t[i].name) return true;
        return false;
    }
};

int up16(int c) { return (c + 15) / 16 * 16; }

}  // namespace

// ------------------------------------------------------------------------------------------ graph
struct MtTensor {
    int buf = -1;
    int C = 0;        // channels of this view (multiple of 16)
    int ld = 0;       // channels of the underlying buffer
    int coff = 0;     // first channel of the view
    int H = 1, W = 1;
    bool q8 = false;  // e4m3 bytes, [N][C/32][P][32]: C, ld, coff count real channels (multiples of 32), one byte each
    int P() const { return H * W; }
};

enum MtOpType { OP_CONV, OP_GN, OP_LN, OP_ATTN, OP_GEGLU, OP_ADDPOS, OP_VT };

struct MtOp {
    MtOpType type;
    std::string name;
    MtTensor x, y, r, k, v;     // r: residual; attention: x = q, k, v
    int plan = -1;
    int rplan = -1;             // index into MtGraph::rplans: the same layer as a weight-streaming GEMM over gathered rows (rowgemm.hip rowconv)
    int ksz = 1;                // its kernel size (1 or 3)
    bool unet3x3 = false;       // 3x3 stride-1 fp16 conv of the U-Net (maps of <= 32 x 32): measured per-level tile choice (mt_graph_run)
    int act = 0, ups = 0;
    int gamma = -1, beta = -1;  // indices into MtGraph::vecs
    int groups = 32, silu = 0;
    float eps = 1e-5f;
    int heads = 1, d16 = 0;
    int Tk = 0;
    // LayerNorm fold (MT_FUSE bit 2; conv3_mfma.hip K3Args::ln_*): ln_out_buf = this linear layer also writes its output's per-token
    // partial sums there; ln_in_buf = it consumes a LayerNorm'ed tensor but reads the raw one, statistics from that buffer
    int ln_out_buf = -1, ln_in_buf = -1, ln_in_tiles = 0;
    float ln_eps = 1e-5f;
    int vt_buf = -1;            // OP_ATTN: the values are already transposed in this buffer (written by the pass's OP_VT), else the shared scratch
    long long gn_slot_off = -1; // OP_GN on a map gn_coop_kernel serves: first word of this op's exchange slots in MtGraph::gn_slots
};

// one cross-attention's share of the hoisted k | v projection (MtGraph::kv_all): its value view and where the transposed values go
struct MtVtItem { MtTensor v; int heads, d16, Tk, vt_buf; };

struct MtGraph {
    std::vector<size_t> buf_halfs;           // per frame
    std::vector<f16*> bufs;
    std::vector<ConvPlan> plans;
    std::vector<RowGemmPlan> rplans;         // small-map layers (<= 64 pixels / tokens per frame) also as rowconv plans (add_conv2)
    std::vector<float*> vecs;                // device fp32 vectors (norm affine)
    std::vector<MtOp> ops;
    std::map<std::string, MtTensor> named;
    float* gn_partial = nullptr;
    size_t gn_partial_floats = 0;
    unsigned* gn_slots = nullptr;            // gn_coop_kernel's exchange slots of every GroupNorm op it serves (reset to the sentinel at the head of a pass)
    size_t gn_slot_words = 0;
    unsigned* gn_err_host = nullptr;         // host-mapped word a block sets when its wait for its set ran out; gn_err_dev = the device's view of it
    unsigned* gn_err_dev = nullptr;
    f16* vt = nullptr;                        // transposed values scratch
    size_t vt_halfs = 0;                      // per frame
    struct KvPre { MtTensor k, v; int vt_buf; };
    std::map<std::string, KvPre> kv_pre;      // cross-attention name -> its views of the hoisted projection
    std::vector<MtVtItem> vt_items;           // OP_VT: every cross-attention's values, transposed by ONE launch at the head of the pass
    int frames = 0;
    // fp8 conv path (BASELINE configs[4]): the GroupNorm+SiLU in front of every ResnetBlock2D 3x3 conv writes e4m3
    // (x * fp8_ascale, saturating) and the conv runs on fp8 operands; everything else stays fp16
    bool fp8 = false;
    float fp8_ascale = 8.f;
    double macs = 0;                          // conv / linear MACs per frame (attention excluded)
    double macs_fp8 = 0;                      // ... of which on fp8 operands
    std::string err;
    unsigned long long* sat_ctr = nullptr;    // debug (knob SAT_CHECK): saturation counters every op's output is scanned into
    MtTensor *t_latent = nullptr, *t_ctx = nullptr, *t_unet_out = nullptr, *t_vae_out = nullptr;
    MtTensor* whisper_states = nullptr;

    MtTensor alloc(int C, int H, int W) {
        MtTensor t;
        t.buf = (int)buf_halfs.size();
        t.C = up16(C); t.ld = t.C; t.coff = 0; t.H = H; t.W = W;
        buf_halfs.push_back((size_t)t.C * H * W);
        return t;
    }
    MtTensor alloc_q8(int C, int H, int W) {          // C % 32 == 0
        MtTensor t;
        t.buf = (int)buf_halfs.size();
        t.C = C; t.ld = C; t.coff = 0; t.H = H; t.W = W; t.q8 = true;
        buf_halfs.push_back((size_t)(C / 2) * H * W);
        return t;
    }
    static MtTensor view(const MtTensor& b, int coff, int C) {
        MtTensor t = b;
        t.coff = b.coff + coff; t.C = C;
        return t;
    }
    int add_vec(const float* host, int n, int pad_to = 0) {
        const int m = std::max(n, pad_to);
        std::vector<float> tmp(m, 0.f);
        memcpy(tmp.data(), host, n * sizeof(float));
        float* d = nullptr;
        if (hipMalloc((void**)&d, m * sizeof(float)) != hipSuccess) { err = "hipMalloc failed"; return -1; }
        (void)hipMemcpy(d, tmp.data(), m * sizeof(float), hipMemcpyHostToDevice);
        vecs.push_back(d);
        return (int)vecs.size() - 1;
    }
    // conv / linear: weight [Cout][Cin][k][k] fp32 host, bias [Cout] or null
    // `scale` (or null = 1): per-output-channel factor of the epilogue (the LayerNorm fold passes sum_ci W'[co][ci] here)
    int add_conv(const std::string& name, const float* w, const float* bias, int Cin, int Cout, int k, int stride, int pad,
                 const MtTensor& x, const MtTensor& y, const MtTensor* res, int act, int ups, const float* scale = nullptr) {
        return add_conv2(name, w, bias, Cin, Cout, k, k, stride, stride, pad, pad, x, y, res, act, ups, 0, scale);
    }
    // per-token partial statistics of a C-channel tensor on an H x W map: [tokens][C / 32] float2 (LayerNorm fold)
    int alloc_ln_stats(int C, int H, int W) {
        const int b = (int)buf_halfs.size();
        buf_halfs.push_back((size_t)H * W * (C / 32) * 4);          // float2 = 4 halfs
        return b;
    }
    // diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + Conv2d(k3, s2, p0)  (AutoencoderKL encoder)
    int add_conv_down_asym(const std::string& name, const float* w, const float* bias, int C, const MtTensor& x, const MtTensor& y) {
        return add_conv2(name, w, bias, C, C, 3, 3, 2, 2, 0, 0, x, y, nullptr, 0, 0, 1);
    }
    // rectangular kernel / stride (Conv1d over a [T][1] token map: kh x 1)
    int add_conv2(const std::string& name, const float* w, const float* bias, int Cin, int Cout, int kh, int kw, int sh, int sw,
                  int ph, int pw, const MtTensor& x, const MtTensor& y, const MtTensor* res, int act, int ups, int pad_br = 0,
                  const float* scale = nullptr) {
        const int k = kh, stride = sh;
        const int kk = kh * kw;
        (void)k;
        const int CoutP = up16(Cout);
        const int CinR = Cin;
        Cin = up16(Cin);                    // whole channel blocks on both sides (zero weights for the padding)
        std::vector<float> wp;
        const float* wuse = w;
        if (CoutP != Cout || Cin != CinR) {
            wp.assign((size_t)CoutP * Cin * kk, 0.f);
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < CinR; ++ci)
                    memcpy(&wp[((size_t)co * Cin + ci) * kk], &w[((size_t)co * CinR + ci) * kk], (size_t)kk * sizeof(float));
            wuse = wp.data();
        }
        macs += (double)CinR * Cout * kk * (stride == 2 ? y.P() : (ups ? x.P() : y.P()));
        if (x.q8) macs_fp8 += (double)CinR * Cout * kk * y.P();
        std::vector<float> sc(CoutP, 1.f), sf(CoutP, 0.f);
        if (bias) memcpy(sf.data(), bias, Cout * sizeof(float));
        if (scale) memcpy(sc.data(), scale, Cout * sizeof(float));
        ConvPlan p;
        std::string e;
        int rc = conv_plan_create(&p, wuse, Cin, CoutP, kh, kw, sh, sw, ph, pw, false, pad_br, sc.data(), sf.data(), &e, x.P(),
                                  x.q8 ? ((CinR % 64 == 0 && (knob(K_FP8_MX) == 2 || (knob(K_FP8_MX) == 1 && CinR >= 512))) ? 2 : 1) : 0, fp8_ascale, ups ? 1 : 0);
        if (rc) { err = name + ": " + e; return -1; }
        plans.push_back(p);
        MtOp op;
        op.type = OP_CONV; op.name = name; op.x = x; op.y = y; op.plan = (int)plans.size() - 1; op.act = act; op.ups = ups;
        op.unet3x3 = !x.q8 && !ups && kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && pad_br == 0 && x.P() <= 1024 && Cin >= 320 &&
                     name.rfind("decoder.", 0) != 0 && name.rfind("encoder.", 0) != 0;
        // The 1x1 / linear layers on maps of <= 64 pixels or tokens per frame (the U-Net's 8x8 and 4x4 levels: projections of the transformer
        // blocks, resnet shortcuts, the 50-token context projections; 1..13 MB of weights each behind 1024 / 256 rows of a 16-frame pass) also get
        // a rowconv plan: conv3 runs them as ~160 items of 40..160 chunks each behind a two-stage DMA pipe (30 us for a 1280 x 1280 linear layer
        // whose weights stream in 0.5 us); the weight-streaming GEMM over gathered rows pays one round trip per trip instead (mt_graph_run picks
        // it by the launch's row count).
        // Only the 1x1 / linear layers with <= 2560 outputs: a row block re-gathers its rows for every 32-output slab and every weight slab is
        // re-read by every row group, so the 3x3 layers (K = 11 520..23 040: ~1.4 GB of L2 -> CU traffic per layer at 1024 rows) and the
        // 10 240-output GEGLU projection would lose to conv3.
        if (knob(K_MT_ROWCONV) > 0 && !x.q8 && !ups && act == 0 && sh == 1 && sw == 1 && kh == 1 && kw == 1 && ph == 0 && pw == 0 && pad_br == 0 &&
            Cin % 32 == 0 && Cin <= 5120 && CoutP % 256 == 0 && CoutP <= 2560 && x.P() <= 64 && x.P() == y.P()) {
            const size_t K = (size_t)kk * Cin;
            std::vector<float> we((size_t)CoutP * K);
            for (int co = 0; co < CoutP; ++co) {
                const float* src = wuse + (size_t)co * Cin * kk;
                float* dst = we.data() + (size_t)co * K;
                for (int t = 0; t < kk; ++t)
                    for (int ci = 0; ci < Cin; ++ci) dst[(size_t)t * Cin + ci] = src[(size_t)ci * kk + t];
            }
            RowGemmPlan rg;
            rc = rowgemm_plan_create(&rg, we.data(), CoutP, (int)K, sc.data(), sf.data(), &e);
            if (rc) { err = name + ": " + e; return -1; }
            rplans.push_back(rg);
            op.rplan = (int)rplans.size() - 1;
            op.ksz = kh;
        }
        if (res) op.r = *res;
        // act 4 = GEGLU in the epilogue (conv3_mfma.hip): the output is half as wide as the projection
        if (up16(Cin) != x.C || (act == 4 ? CoutP / 2 : CoutP) != y.C) { err = name + ": channel mismatch (" + std::to_string(Cin) + "->" + std::to_string(Cout) + ")"; return -1; }
        ops.push_back(op);
        named[name] = y;
        return 0;
    }
    int add_gn(const std::string& name, SD& sd, const std::string& prefix, const MtTensor& x, const MtTensor& y, float eps, int silu) {
        const float* g = sd.get(prefix + ".weight", x.C);
        const float* b = sd.get(prefix + ".bias", x.C);
        if (!g || !b) { err = sd.err; return -1; }
        MtOp op;
        op.type = OP_GN; op.name = name; op.x = x; op.y = y; op.eps = eps; op.silu = silu; op.groups = 32;
        op.gamma = add_vec(g, x.C); op.beta = add_vec(b, x.C);
        ops.push_back(op);
        if (!y.q8) named[name] = y;
        return 0;
    }
    int add_ln(const std::string& name, SD& sd, const std::string& prefix, const MtTensor& x, const MtTensor& y, float eps) {
        const float* g = sd.get(prefix + ".weight", x.C);
        const float* b = sd.get(prefix + ".bias", x.C);
        if (!g || !b) { err = sd.err; return -1; }
        MtOp op;
        op.type = OP_LN; op.name = name; op.x = x; op.y = y; op.eps = eps;
        op.gamma = add_vec(g, x.C); op.beta = add_vec(b, x.C);
        ops.push_back(op);
        named[name] = y;
        return 0;
    }
    // `vt_buf` >= 0: the values were transposed into that buffer by the pass's OP_VT (hoisted cross-attention k | v, mt_build_unet)
    void add_attn(const std::string& name, const MtTensor& q, const MtTensor& k, const MtTensor& v, const MtTensor& o, int heads, int d16,
                  int vt_buf = -1) {
        MtOp op;
        op.type = OP_ATTN; op.name = name; op.x = q; op.k = k; op.v = v; op.y = o; op.heads = heads; op.d16 = d16; op.Tk = k.P();
        op.vt_buf = vt_buf;
        if (vt_buf < 0) vt_halfs = std::max(vt_halfs, (size_t)heads * attn_dv32(d16) * attn_tkp(k.P()));
        ops.push_back(op);
        named[name + ".attn"] = o;
    }
    void add_geglu(const std::string& name, const MtTensor& x, const MtTensor& y) {
        MtOp op;
        op.type = OP_GEGLU; op.name = name; op.x = x; op.y = y;
        ops.push_back(op);
        named[name] = y;
    }
};

namespace {

// ------------------------------------------------------------------------------------------ weight transforms
// Linear [Cout][Cin] whose OUTPUT channels are per-head slices of d channels -> heads padded to d16 (zero rows)
std::vector<float> pad_heads_rows(const float* w, int C, int Cin, int heads, int d, int d16, float scale) {
    std::vector<float> o((size_t)heads * d16 * Cin, 0.f);
    for (int h = 0; h < heads; ++h)
        for (int j = 0; j < d; ++j) {
            const float* src = w + (size_t)(h * d + j) * Cin;
            float* dst = o.data() + (size_t)(h * d16 + j) * Cin;
            for (int i = 0; i < Cin; ++i) dst[i] = src[i] * scale;
        }
    (void)C;
    return o;
}
std::vector<float> pad_heads_vec(const float* b, int heads, int d, int d16, float scale) {
    std::vector<float> o((size_t)heads * d16, 0.f);
    if (b)
        for (int h = 0; h < heads; ++h)
            for (int j = 0; j < d; ++j) o[h * d16 + j] = b[h * d + j] * scale;
    return o;
}
// Linear [Cout][C] whose INPUT channels are per-head slices -> padded input columns
std::vector<float> pad_heads_cols(const float* w, int Cout, int heads, int d, int d16) {
    std::vector<float> o((size_t)Cout * heads * d16, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int h = 0; h < heads; ++h)
            for (int j = 0; j < d; ++j) o[(size_t)co * heads * d16 + h * d16 + j] = w[(size_t)co * heads * d + h * d + j];
    return o;
}

// LayerNorm folded into the linear layer behind it (MT_FUSE bit 2): y = W LN(x) + b = rstd * (W' x - mean * scale) + shift with
//   W'[co][ci] = W[co][ci] gamma[ci]   (what is packed as the layer's fp16 weights),
//   scale[co]  = sum_ci fp16(W'[co][ci])   (of the ROUNDED weights, so that the mean term cancels exactly what the MFMAs summed),
//   shift[co]  = sum_ci W[co][ci] beta[ci] + b[co];
// mean / rstd per token come from the partial sums the producer of x wrote (stats_buf).
struct LnFold { const float* gamma; const float* beta; int C; int stats_buf; float eps; };
void ln_fold(const float* w, const float* bias, int Cout, const LnFold& ln, std::vector<float>* wf, std::vector<float>* scale,
             std::vector<float>* shift) {
    const int C = ln.C;
    wf->resize((size_t)Cout * C); scale->assign(Cout, 0.f); shift->assign(Cout, 0.f);
    for (int co = 0; co < Cout; ++co) {
        double ssum = 0.0, bsum = bias ? (double)bias[co] : 0.0;
        for (int ci = 0; ci < C; ++ci) {
            const float v = w[(size_t)co * C + ci] * ln.gamma[ci];
            (*wf)[(size_t)co * C + ci] = v;
            ssum += (double)(float)(f16)v;
            bsum += (double)w[(size_t)co * C + ci] * ln.beta[ci];
        }
        (*scale)[co] = (float)ssum;
        (*shift)[co] = (float)bsum;
    }
}

float silu_h(float v) { return v / (1.f + expf(-v)); }

// UNet2DConditionModel time_proj(flip_sin_to_cos=True, freq_shift=0) + time_embedding at timestep t, then SiLU
// (what every ResnetBlock2D feeds its time_emb_proj): fp32 [1280]
int time_embedding_silu(SD& sd, float t, std::vector<float>* out) {
    const int dim = 320, td = 1280;
    const float* w1 = sd.get("time_embedding.linear_1.weight", (size_t)td * dim);
    const float* b1 = sd.get("time_embedding.linear_1.bias", td);
    const float* w2 = sd.get("time_embedding.linear_2.weight", (size_t)td * td);
    const float* b2 = sd.get("time_embedding.linear_2.bias", td);
    if (!w1 || !b1 || !w2 || !b2) return -1;
    std::vector<float> emb(dim);
    const int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        const float f = expf(-logf(10000.f) * (float)i / (float)half);
        emb[i] = cosf(t * f);            // flip_sin_to_cos: cos first
        emb[half + i] = sinf(t * f);
    }
    std::vector<float> h1(td), h2(td);
    for (int o = 0; o < td; ++o) {
        double a = b1[o];
        for (int i = 0; i < dim; ++i) a += (double)w1[(size_t)o * dim + i] * emb[i];
        h1[o] = silu_h((float)a);
    }
    for (int o = 0; o < td; ++o) {
        double a = b2[o];
        for (int i = 0; i < td; ++i) a += (double)w2[(size_t)o * td + i] * h1[i];
        h2[o] = silu_h((float)a);
    }
    *out = h2;
    return 0;
}

// ------------------------------------------------------------------------------------------ blocks
// diffusers ResnetBlock2D; temb_silu (or null) is folded into conv1's bias
int build_resnet(MtGraph& g, SD& sd, const std::string& p, const MtTensor& x, const MtTensor& out, int Cin, int Cout,
                 const std::vector<float>* temb_silu, float eps) {
    const int H = x.H, W = x.W;
    MtTensor t1 = (g.fp8 && Cin % 32 == 0) ? g.alloc_q8(Cin, H, W) : g.alloc(Cin, H, W);
    if (g.add_gn(p + ".norm1", sd, p + ".norm1", x, t1, eps, 1)) return -1;
    const float* w1 = sd.get(p + ".conv1.weight", (size_t)Cout * Cin * 9);
    const float* b1 = sd.get(p + ".conv1.bias", Cout);
    if (!w1 || !b1) { g.err = sd.err; return -1; }
    std::vector<float> bias1(b1, b1 + Cout);
    if (temb_silu) {
        const int td = (int)temb_silu->size();
        const float* wt = sd.get(p + ".time_emb_proj.weight", (size_t)Cout * td);
        const float* bt = sd.get(p + ".time_emb_proj.bias", Cout);
        if (!wt || !bt) { g.err = sd.err; return -1; }
        for (int o = 0; o < Cout; ++o) {
            double a = bt[o];
            for (int i = 0; i < td; ++i) a += (double)wt[(size_t)o * td + i] * (*temb_silu)[i];
            bias1[o] += (float)a;
        }
    }
    MtTensor h = g.alloc(Cout, H, W);
    if (g.add_conv(p + ".conv1", w1, bias1.data(), Cin, Cout, 3, 1, 1, t1, h, nullptr, 0, 0)) return -1;
    MtTensor t2 = (g.fp8 && Cout % 32 == 0) ? g.alloc_q8(Cout, H, W) : g.alloc(Cout, H, W);
    if (g.add_gn(p + ".norm2", sd, p + ".norm2", h, t2, eps, 1)) return -1;
    MtTensor skip = x;
    if (sd.has(p + ".conv_shortcut.weight")) {
        const float* ws = sd.get(p + ".conv_shortcut.weight", (size_t)Cout * Cin);
        const float* bs = sd.get(p + ".conv_shortcut.bias", Cout);
        if (!ws || !bs) { g.err = sd.err; return -1; }
        skip = g.alloc(Cout, H, W);
        if (g.add_conv(p + ".conv_shortcut", ws, bs, Cin, Cout, 1, 1, 0, x, skip, nullptr, 0, 0)) return -1;
    } else if (Cin != Cout) {
        g.err = p + ": missing conv_shortcut"; return -1;
    }
    const float* w2 = sd.get(p + ".conv2.weight", (size_t)Cout * Cout * 9);
    const float* b2 = sd.get(p + ".conv2.bias", Cout);
    if (!w2 || !b2) { g.err = sd.err; return -1; }
    return g.add_conv(p + ".conv2", w2, b2, Cout, Cout, 3, 1, 1, t2, out, &skip, 0, 0);
}

// diffusers Attention (to_q / to_k / to_v / to_out.0), q from x (C channels), k/v from ctx (Cctx channels);
// out = to_out(attn) + res
// `kv_pre` (or null): {k view, v view} of a projection that already ran (the hoisted, stacked k | v projection of every
// cross-attention, mt_build_unet), `vt_pre` the buffer its transposed values are in
int build_attention(MtGraph& g, SD& sd, const std::string& p, const MtTensor& x, const MtTensor& ctx, int C, int Cctx, int heads,
                    bool qkv_bias, const MtTensor& res, const MtTensor& out, const MtTensor* kv_pre = nullptr, int vt_pre = -1,
                    const LnFold* ln = nullptr, int ln_out_buf = -1) {
    // `ln`: x is the RAW tensor of a LayerNorm in front of this attention; the projections that read it are built folded (ln_fold).
    // `ln_out_buf` >= 0: to_out.0 (+ residual) also writes the per-token partial statistics of its output there (the next LayerNorm's)
    const int d = C / heads, d16 = up16(d), Cp = heads * d16;
    const float scale = 1.0f / sqrtf((float)d);
    const float* wq = sd.get(p + ".to_q.weight", (size_t)C * C);
    const float* wk = sd.get(p + ".to_k.weight", (size_t)C * Cctx);
    const float* wv = sd.get(p + ".to_v.weight", (size_t)C * Cctx);
    const float* wo = sd.get(p + ".to_out.0.weight", (size_t)C * C);
    const float* bo = sd.get(p + ".to_out.0.bias", C);
    if (!wq || !wk || !wv || !wo || !bo) { g.err = sd.err; return -1; }
    const float *bq = nullptr, *bk = nullptr, *bv = nullptr;
    if (qkv_bias) {
        bq = sd.get(p + ".to_q.bias", C); bk = sd.get(p + ".to_k.bias", C); bv = sd.get(p + ".to_v.bias", C);
        if (!bq || !bk || !bv) { g.err = sd.err; return -1; }
    }
    const std::vector<float> wqp = pad_heads_rows(wq, C, C, heads, d, d16, scale), bqp = pad_heads_vec(bq, heads, d, d16, scale);
    const std::vector<float> wkp = pad_heads_rows(wk, C, Cctx, heads, d, d16, 1.f), bkp = pad_heads_vec(bk, heads, d, d16, 1.f);
    const std::vector<float> wvp = pad_heads_rows(wv, C, Cctx, heads, d, d16, 1.f), bvp = pad_heads_vec(bv, heads, d, d16, 1.f);
    const std::vector<float> wop = pad_heads_cols(wo, C, heads, d, d16);
    // Projections that read the same tensor are ONE launch (rows of the weight matrices stacked, outputs = channel-block
    // ranges of one buffer): q|k|v for self-attention, k|v for cross-attention.  These are 1-3 GFLOP GEMMs whose cost is
    // the launch, not the math.
    auto stack = [](std::initializer_list<const std::vector<float>*> parts) {
        std::vector<float> o;
        for (const std::vector<float>* v : parts) o.insert(o.end(), v->begin(), v->end());
        return o;
    };
    const bool fuse = !knob(K_MT_NO_QKV_FUSE);      // A/B switch
    const bool self = fuse && x.buf == ctx.buf && x.coff == ctx.coff && x.C == ctx.C;
    MtTensor q, k, v, o = g.alloc(Cp, x.H, x.W);
    // a projection of x: plain, or with the LayerNorm in front of it folded in
    auto proj_x = [&](const std::string& name, const std::vector<float>& w, const std::vector<float>& b, int Cout, const MtTensor& y) -> int {
        if (!ln) return g.add_conv(name, w.data(), b.data(), C, Cout, 1, 1, 0, x, y, nullptr, 0, 0);
        std::vector<float> wf, sc, sf;
        ln_fold(w.data(), b.data(), Cout, *ln, &wf, &sc, &sf);
        if (g.add_conv(name, wf.data(), sf.data(), C, Cout, 1, 1, 0, x, y, nullptr, 0, 0, sc.data())) return -1;
        MtOp& op = g.ops.back();
        op.ln_in_buf = ln->stats_buf; op.ln_in_tiles = ln->C / 32; op.ln_eps = ln->eps;
        return 0;
    };
    if (ln && !(kv_pre || self)) { g.err = p + ": the LayerNorm fold needs the stacked q|k|v projection or a hoisted k|v"; return -1; }
    if (kv_pre) {
        q = g.alloc(Cp, x.H, x.W);
        if (proj_x(p + ".to_q", wqp, bqp, Cp, q)) return -1;
        k = kv_pre[0]; v = kv_pre[1];
    } else if (self) {
        MtTensor qkv = g.alloc(3 * Cp, x.H, x.W);
        const std::vector<float> w3 = stack({&wqp, &wkp, &wvp}), b3 = stack({&bqp, &bkp, &bvp});
        if (proj_x(p + ".to_qkv", w3, b3, 3 * Cp, qkv)) return -1;
        q = MtGraph::view(qkv, 0, Cp); k = MtGraph::view(qkv, Cp, Cp); v = MtGraph::view(qkv, 2 * Cp, Cp);
    } else if (!fuse) {
        q = g.alloc(Cp, x.H, x.W); k = g.alloc(Cp, ctx.H, ctx.W); v = g.alloc(Cp, ctx.H, ctx.W);
        if (g.add_conv(p + ".to_q", wqp.data(), bqp.data(), C, Cp, 1, 1, 0, x, q, nullptr, 0, 0)) return -1;
        if (g.add_conv(p + ".to_k", wkp.data(), bkp.data(), Cctx, Cp, 1, 1, 0, ctx, k, nullptr, 0, 0)) return -1;
        if (g.add_conv(p + ".to_v", wvp.data(), bvp.data(), Cctx, Cp, 1, 1, 0, ctx, v, nullptr, 0, 0)) return -1;
    } else {
        q = g.alloc(Cp, x.H, x.W);
        MtTensor kv = g.alloc(2 * Cp, ctx.H, ctx.W);
        const std::vector<float> w2 = stack({&wkp, &wvp}), b2 = stack({&bkp, &bvp});
        if (g.add_conv(p + ".to_q", wqp.data(), bqp.data(), C, Cp, 1, 1, 0, x, q, nullptr, 0, 0)) return -1;
        if (g.add_conv(p + ".to_kv", w2.data(), b2.data(), Cctx, 2 * Cp, 1, 1, 0, ctx, kv, nullptr, 0, 0)) return -1;
        k = MtGraph::view(kv, 0, Cp); v = MtGraph::view(kv, Cp, Cp);
    }
    g.named[p + ".to_q"] = q; g.named[p + ".to_k"] = k; g.named[p + ".to_v"] = v;
    g.add_attn(p, q, k, v, o, heads, d16, kv_pre ? vt_pre : -1);
    if (g.add_conv(p + ".to_out.0", wop.data(), bo, Cp, C, 1, 1, 0, o, out, &res, 0, 0)) return -1;
    g.ops.back().ln_out_buf = ln_out_buf;
    return 0;
}

// diffusers Transformer2DModel (conv proj_in/out) with one BasicTransformerBlock (GEGLU feed-forward)
int build_transformer(MtGraph& g, SD& sd, const std::string& p, const MtTensor& x, const MtTensor& ctx, const MtTensor& out, int C) {
    const int H = x.H, W = x.W;
    MtTensor t = g.alloc(C, H, W);
    if (g.add_gn(p + ".norm", sd, p + ".norm", x, t, 1e-6f, 0)) return -1;
    const float* wi = sd.get(p + ".proj_in.weight", (size_t)C * C);
    const float* bi = sd.get(p + ".proj_in.bias", C);
    if (!wi || !bi) { g.err = sd.err; return -1; }
    MtTensor h0 = g.alloc(C, H, W);
    if (g.add_conv(p + ".proj_in", wi, bi, C, C, 1, 1, 0, t, h0, nullptr, 0, 0)) return -1;
    const std::string b = p + ".transformer_blocks.0";
    // MT_FUSE bit 2: the three LayerNorms disappear into the linear layers around them - the producer of a normalised tensor
    // (proj_in, attn1.to_out.0, attn2.to_out.0) writes per-token partial sums beside its output, the consumer (to_qkv, attn2.to_q,
    // ff.net.0.proj) reads the raw tensor with gamma folded into its weights and normalises in its epilogue (ln_fold).  48 launches
    // and as many tensor round trips of a pass; needs the stacked q|k|v projection and the hoisted cross-attention k|v (bit 1).
    const bool fold = (knob(K_MT_FUSE) & 4) && (knob(K_MT_FUSE) & 2) && !knob(K_MT_NO_QKV_FUSE) && C % 32 == 0 &&
                      g.kv_pre.find(b + ".attn2") != g.kv_pre.end();
    LnFold l1{nullptr, nullptr, C, -1, 1e-5f}, l2 = l1, l3 = l1;
    if (fold) {
        const char* nm[3] = {".norm1", ".norm2", ".norm3"};
        LnFold* ls[3] = {&l1, &l2, &l3};
        for (int i = 0; i < 3; ++i) {
            ls[i]->gamma = sd.get(b + nm[i] + ".weight", C);
            ls[i]->beta = sd.get(b + nm[i] + ".bias", C);
            if (!ls[i]->gamma || !ls[i]->beta) { g.err = sd.err; return -1; }
            ls[i]->stats_buf = g.alloc_ln_stats(C, H, W);
        }
        g.ops.back().ln_out_buf = l1.stats_buf;           // proj_in
    }
    MtTensor n1, h1 = g.alloc(C, H, W);
    if (fold) {
        if (build_attention(g, sd, b + ".attn1", h0, h0, C, C, 8, false, h0, h1, nullptr, -1, &l1, l2.stats_buf)) return -1;
    } else {
        n1 = g.alloc(C, H, W);
        if (g.add_ln(b + ".norm1", sd, b + ".norm1", h0, n1, 1e-5f)) return -1;
        if (build_attention(g, sd, b + ".attn1", n1, n1, C, C, 8, false, h0, h1)) return -1;
    }
    MtTensor n2, h2 = g.alloc(C, H, W);
    if (fold) {
        auto kv = g.kv_pre.find(b + ".attn2");
        const MtTensor pre[2] = {kv->second.k, kv->second.v};
        if (build_attention(g, sd, b + ".attn2", h1, ctx, C, 384, 8, false, h1, h2, pre, kv->second.vt_buf, &l2, l3.stats_buf)) return -1;
    } else {
        n2 = g.alloc(C, H, W);
        if (g.add_ln(b + ".norm2", sd, b + ".norm2", h1, n2, 1e-5f)) return -1;
        auto kv = g.kv_pre.find(b + ".attn2");
        if (kv != g.kv_pre.end()) {
            const MtTensor pre[2] = {kv->second.k, kv->second.v};
            if (build_attention(g, sd, b + ".attn2", n2, ctx, C, 384, 8, false, h1, h2, pre, kv->second.vt_buf)) return -1;
        } else if (build_attention(g, sd, b + ".attn2", n2, ctx, C, 384, 8, false, h1, h2)) return -1;
    }
    MtTensor n3, f1, gg = g.alloc(4 * C, H, W), h3 = g.alloc(C, H, W);
    if (!(knob(K_MT_FUSE) & 1)) f1 = g.alloc(8 * C, H, W);
    if (!fold) {
        n3 = g.alloc(C, H, W);
        if (g.add_ln(b + ".norm3", sd, b + ".norm3", h2, n3, 1e-5f)) return -1;
    }
    const float* w1 = sd.get(b + ".ff.net.0.proj.weight", (size_t)8 * C * C);
    const float* b1 = sd.get(b + ".ff.net.0.proj.bias", 8 * C);
    const float* w2 = sd.get(b + ".ff.net.2.weight", (size_t)C * 4 * C);
    const float* b2 = sd.get(b + ".ff.net.2.bias", C);
    if (!w1 || !b1 || !w2 || !b2) { g.err = sd.err; return -1; }
    // the projection as the device runs it: LayerNorm folded in (fold), rows permuted for the GEGLU epilogue (bit 0)
    std::vector<float> wff, scf, sff;
    const float *wp1 = w1, *bp1 = b1, *sp1 = nullptr;
    if (fold) { ln_fold(w1, b1, 8 * C, l3, &wff, &scf, &sff); wp1 = wff.data(); bp1 = sff.data(); sp1 = scf.data(); }
    const MtTensor& ffin = fold ? h2 : n3;
    if (knob(K_MT_FUSE) & 1) {
        // GEGLU in the projection's epilogue: rows permuted so that every 32-row tile is [16 value rows | their 16 gate rows]
        // (value = rows [0, 4C), gate = rows [4C, 8C) of ff.net.0.proj: diffusers GEGLU chunks the projection in that order)
        const int F = 4 * C;
        std::vector<float> wp((size_t)8 * C * C), bp((size_t)8 * C), sp((size_t)8 * C, 1.f);
        for (int tl = 0; tl < F / 16; ++tl)
            for (int r = 0; r < 32; ++r) {
                const int src = (r < 16) ? tl * 16 + r : F + tl * 16 + (r - 16);
                memcpy(&wp[(size_t)(tl * 32 + r) * C], &wp1[(size_t)src * C], (size_t)C * sizeof(float));
                bp[tl * 32 + r] = bp1[src];
                if (sp1) sp[tl * 32 + r] = sp1[src];
            }
        if (g.add_conv(b + ".ff.net.0.proj", wp.data(), bp.data(), C, 8 * C, 1, 1, 0, ffin, gg, nullptr, 4, 0, sp1 ? sp.data() : nullptr)) return -1;
        g.named.erase(b + ".ff.net.0.proj");          // the projection itself is never materialised
        g.named[b + ".ff.geglu"] = gg;
    } else {
        if (g.add_conv(b + ".ff.net.0.proj", wp1, bp1, C, 8 * C, 1, 1, 0, ffin, f1, nullptr, 0, 0, sp1)) return -1;
    }
    if (fold) { MtOp& op = g.ops[g.ops.size() - 1]; op.ln_in_buf = l3.stats_buf; op.ln_in_tiles = C / 32; op.ln_eps = l3.eps; }
    if (!(knob(K_MT_FUSE) & 1)) g.add_geglu(b + ".ff.geglu", f1, gg);
    if (g.add_conv(b + ".ff.net.2", w2, b2, 4 * C, C, 1, 1, 0, gg, h3, &h2, 0, 0)) return -1;
    const float* wo = sd.get(p + ".proj_out.weight", (size_t)C * C);
    const float* bo = sd.get(p + ".proj_out.bias", C);
    if (!wo || !bo) { g.err = sd.err; return -1; }
    return g.add_conv(p + ".proj_out", wo, bo, C, C, 1, 1, 0, h3, out, &x, 0, 0);
}

}  // namespace

// ------------------------------------------------------------------------------------------ U-Net
int mt_build_unet(MtGraph& g, const ltk_named_tensor* t, int n, MtTensor* latent_in, MtTensor* ctx_in, MtTensor* out) {
    SD sd{t, n, ""};
    const int ch[4] = {320, 640, 1280, 1280};
    const bool down_attn[4] = {true, true, true, false};
    const bool up_attn[4] = {false, true, true, true};
    std::vector<float> temb;
    if (time_embedding_silu(sd, 0.f, &temb)) { g.err = sd.err; return -1; }

    *latent_in = g.alloc(8, 32, 32);         // 16-channel block, 8 real
    *ctx_in = g.alloc(384, 50, 1);
    g.named["latent_in"] = *latent_in;

    // Hoisted cross-attention k | v (MT_FUSE bit 1): the k and v projections of the 16 cross-attentions read the audio context
    // only, so they are ONE stacked 384 -> 25 600 projection at the head of the pass (rows [k_0 | v_0 | k_1 | v_1 ...], heads padded
    // as in build_attention) and ONE launch that transposes the 16 value tensors - instead of 16 + 16 launches on the critical
    // path of their blocks (320 us of a 16-frame pass).
    if (knob(K_MT_FUSE) & 2) {
        struct Blk { std::string p; int C; };
        std::vector<Blk> blks;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) blks.push_back({"down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), ch[i]});
        blks.push_back({"mid_block.attentions.0", 1280});
        const int revc[4] = {1280, 1280, 640, 320};
        for (int i = 1; i < 4; ++i) for (int j = 0; j < 3; ++j) blks.push_back({"up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), revc[i]});
        int total = 0;
        for (const Blk& b : blks) total += 2 * 8 * up16(b.C / 8);
        std::vector<float> wall((size_t)total * 384);
        size_t row = 0;
        std::vector<int> offs;
        for (const Blk& b : blks) {
            const std::string a = b.p + ".transformer_blocks.0.attn2";
            const int d = b.C / 8, d16 = up16(d);
            const float* wk = sd.get(a + ".to_k.weight", (size_t)b.C * 384);
            const float* wv = sd.get(a + ".to_v.weight", (size_t)b.C * 384);
            if (!wk || !wv) { g.err = sd.err; return -1; }
            const std::vector<float> wkp = pad_heads_rows(wk, b.C, 384, 8, d, d16, 1.f), wvp = pad_heads_rows(wv, b.C, 384, 8, d, d16, 1.f);
            offs.push_back((int)row);
            memcpy(&wall[row * 384], wkp.data(), wkp.size() * sizeof(float)); row += (size_t)8 * d16;
            memcpy(&wall[row * 384], wvp.data(), wvp.size() * sizeof(float)); row += (size_t)8 * d16;
        }
        MtTensor kv_all = g.alloc(total, 50, 1);
        if (g.add_conv("attn2.to_kv_all", wall.data(), nullptr, 384, total, 1, 1, 0, *ctx_in, kv_all, nullptr, 0, 0)) return -1;
        MtOp vt;
        vt.type = OP_VT; vt.name = "attn2.v_transpose_all";
        for (size_t bi = 0; bi < blks.size(); ++bi) {
            const int d16 = up16(blks[bi].C / 8), Cp = 8 * d16;
            MtGraph::KvPre pre;
            pre.k = MtGraph::view(kv_all, offs[bi], Cp);
            pre.v = MtGraph::view(kv_all, offs[bi] + Cp, Cp);
            pre.vt_buf = (int)g.buf_halfs.size();
            g.buf_halfs.push_back((size_t)8 * attn_dv32(d16) * attn_tkp(50));
            g.kv_pre[blks[bi].p + ".transformer_blocks.0.attn2"] = pre;
            g.vt_items.push_back({pre.v, 8, d16, 50, pre.vt_buf});
        }
        g.ops.push_back(vt);
    }

    // skip tensors live inside the cat buffers of the up path: plan those first (pop order of down_block_res_samples)
    struct SkipSpec { int C, HW; };
    const SkipSpec skips[12] = {{320, 32}, {320, 32}, {320, 32}, {320, 16}, {640, 16}, {640, 16}, {640, 8}, {1280, 8}, {1280, 8}, {1280, 4}, {1280, 4}, {1280, 4}};
    // channels of h entering each up resnet (block i, resnet j)
    const int up_h[4][3] = {{1280, 1280, 1280}, {1280, 1280, 1280}, {1280, 640, 640}, {640, 320, 320}};
    MtTensor cat[4][3], skip_dst[12];
    {
        int si = 11;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j, --si) {
                cat[i][j] = g.alloc(up_h[i][j] + skips[si].C, skips[si].HW, skips[si].HW);
                skip_dst[si] = MtGraph::view(cat[i][j], up_h[i][j], skips[si].C);
            }
    }
    int si = 0;
    const float* wci = sd.get("conv_in.weight", (size_t)320 * 8 * 9);
    const float* bci = sd.get("conv_in.bias", 320);
    if (!wci || !bci) { g.err = sd.err; return -1; }
    if (g.add_conv("conv_in", wci, bci, 8, 320, 3, 1, 1, *latent_in, skip_dst[si], nullptr, 0, 0)) return -1;
    g.named["conv_in"] = skip_dst[si];
    MtTensor h = skip_dst[si++];
    int C = 320;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) {
            const std::string rp = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            MtTensor dst = skip_dst[si++];
            if (down_attn[i]) {
                MtTensor r = g.alloc(ch[i], h.H, h.W);
                if (build_resnet(g, sd, rp, h, r, C, ch[i], &temb, 1e-5f)) return -1;
                if (build_transformer(g, sd, "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), r, *ctx_in, dst, ch[i])) return -1;
            } else {
                if (build_resnet(g, sd, rp, h, dst, C, ch[i], &temb, 1e-5f)) return -1;
            }
            C = ch[i];
            h = dst;
            g.named["down_blocks." + std::to_string(i) + "." + std::to_string(j)] = h;
        }
        if (i < 3) {
            const std::string dp = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
            const float* w = sd.get(dp + ".weight", (size_t)C * C * 9);
            const float* b = sd.get(dp + ".bias", C);
            if (!w || !b) { g.err = sd.err; return -1; }
            MtTensor dst = skip_dst[si++];
            if (g.add_conv(dp, w, b, C, C, 3, 2, 1, h, dst, nullptr, 0, 0)) return -1;
            h = dst;
            g.named["down_blocks." + std::to_string(i) + ".down"] = h;
        }
    }
    {
        MtTensor r0 = g.alloc(C, h.H, h.W), a0 = g.alloc(C, h.H, h.W);
        if (build_resnet(g, sd, "mid_block.resnets.0", h, r0, C, C, &temb, 1e-5f)) return -1;
        if (build_transformer(g, sd, "mid_block.attentions.0", r0, *ctx_in, a0, C)) return -1;
        MtTensor dst = MtGraph::view(cat[0][0], 0, up_h[0][0]);
        if (build_resnet(g, sd, "mid_block.resnets.1", a0, dst, C, C, &temb, 1e-5f)) return -1;
        g.named["mid_block"] = dst;
    }
    const int rev[4] = {1280, 1280, 640, 320};
    MtTensor hup;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) {
            const MtTensor& in = cat[i][j];                    // [h | skip]
            // where this resnet(+attention) writes: the h half of the next cat buffer, or a fresh tensor
            MtTensor dst;
            const bool last_in_block = (j == 2);
            if (!last_in_block) dst = MtGraph::view(cat[i][j + 1], 0, up_h[i][j + 1]);
            else dst = g.alloc(rev[i], in.H, in.W);
            const std::string rp = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            if (up_attn[i]) {
                MtTensor r = g.alloc(rev[i], in.H, in.W);
                if (build_resnet(g, sd, rp, in, r, in.C, rev[i], &temb, 1e-5f)) return -1;
                if (build_transformer(g, sd, "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), r, *ctx_in, dst, rev[i])) return -1;
            } else {
                if (build_resnet(g, sd, rp, in, dst, in.C, rev[i], &temb, 1e-5f)) return -1;
            }
            g.named["up_blocks." + std::to_string(i) + "." + std::to_string(j)] = dst;
            hup = dst;
        }
        if (i < 3) {
            const std::string up = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
            const float* w = sd.get(up + ".weight", (size_t)rev[i] * rev[i] * 9);
            const float* b = sd.get(up + ".bias", rev[i]);
            if (!w || !b) { g.err = sd.err; return -1; }
            MtTensor dst = MtGraph::view(cat[i + 1][0], 0, up_h[i + 1][0]);
            MtTensor xin = hup;
            xin.H *= 2; xin.W *= 2;                               // logical (upsampled) size the conv runs on
            if (g.add_conv(up, w, b, rev[i], rev[i], 3, 1, 1, xin, dst, nullptr, 0, 1)) return -1;
            g.named["up_blocks." + std::to_string(i) + ".up"] = dst;
        }
    }
    MtTensor tn = g.alloc(320, 32, 32);
    if (g.add_gn("conv_norm_out", sd, "conv_norm_out", hup, tn, 1e-5f, 1)) return -1;
    const float* wo = sd.get("conv_out.weight", (size_t)4 * 320 * 9);
    const float* bo = sd.get("conv_out.bias", 4);
    if (!wo || !bo) { g.err = sd.err; return -1; }
    *out = g.alloc(4, 32, 32);
    if (g.add_conv("conv_out", wo, bo, 320, 4, 3, 1, 1, tn, *out, nullptr, 0, 0)) return -1;
    g.named["conv_out"] = *out;
    return 0;
}

// ------------------------------------------------------------------------------------------ VAE decoder
int mt_build_vae(MtGraph& g, const ltk_named_tensor* t, int n, const MtTensor& z, MtTensor* out) {
    SD sd{t, n, ""};
    const float kScaling = 0.18215f;       // AutoencoderKL config.scaling_factor of sd-vae-ft-mse (vae.py:35,103)
    const float* wq = sd.get("post_quant_conv.weight", 16);
    const float* bq = sd.get("post_quant_conv.bias", 4);
    if (!wq || !bq) { g.err = sd.err; return -1; }
    std::vector<float> wqs(16);
    for (int i = 0; i < 16; ++i) wqs[i] = wq[i] / kScaling;
    MtTensor pq = g.alloc(4, 32, 32);
    if (g.add_conv("post_quant_conv", wqs.data(), bq, 4, 4, 1, 1, 0, z, pq, nullptr, 0, 0)) return -1;
    const float* wi = sd.get("decoder.conv_in.weight", (size_t)512 * 4 * 9);
    const float* bi = sd.get("decoder.conv_in.bias", 512);
    if (!wi || !bi) { g.err = sd.err; return -1; }
    MtTensor h = g.alloc(512, 32, 32);
    if (g.add_conv("decoder.conv_in", wi, bi, 4, 512, 3, 1, 1, pq, h, nullptr, 0, 0)) return -1;
    g.named["decoder.conv_in"] = h;
    {
        MtTensor r0 = g.alloc(512, 32, 32);
        if (build_resnet(g, sd, "decoder.mid_block.resnets.0", h, r0, 512, 512, nullptr, 1e-6f)) return -1;
        const std::string a = "decoder.mid_block.attentions.0";
        MtTensor gn = g.alloc(512, 32, 32), ao = g.alloc(512, 32, 32);
        if (g.add_gn(a + ".group_norm", sd, a + ".group_norm", r0, gn, 1e-6f, 0)) return -1;
        if (build_attention(g, sd, a, gn, gn, 512, 512, 1, true, r0, ao)) return -1;
        MtTensor r1 = g.alloc(512, 32, 32);
        if (build_resnet(g, sd, "decoder.mid_block.resnets.1", ao, r1, 512, 512, nullptr, 1e-6f)) return -1;
        h = r1;
        g.named["decoder.mid_block"] = h;
    }
    const int chs[4] = {512, 512, 256, 128};
    int C = 512;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) {
            MtTensor r = g.alloc(chs[i], h.H, h.W);
            if (build_resnet(g, sd, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h, r, C, chs[i], nullptr, 1e-6f)) return -1;
            C = chs[i];
            h = r;
        }
        g.named["decoder.up_blocks." + std::to_string(i)] = h;
        if (i < 3) {
            const std::string up = "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
            const float* w = sd.get(up + ".weight", (size_t)C * C * 9);
            const float* b = sd.get(up + ".bias", C);
            if (!w || !b) { g.err = sd.err; return -1; }
            MtTensor xin = h;
            xin.H *= 2; xin.W *= 2;
            MtTensor dst = g.alloc(C, xin.H, xin.W);
            if (g.add_conv(up, w, b, C, C, 3, 1, 1, xin, dst, nullptr, 0, 1)) return -1;
            h = dst;
        }
    }
    MtTensor tn = g.alloc(128, 256, 256);
    if (g.add_gn("decoder.conv_norm_out", sd, "decoder.conv_norm_out", h, tn, 1e-6f, 1)) return -1;
    const float* wo = sd.get("decoder.conv_out.weight", (size_t)3 * 128 * 9);
    const float* bo = sd.get("decoder.conv_out.bias", 3);
    if (!wo || !bo) { g.err = sd.err; return -1; }
    *out = g.alloc(3, 256, 256);
    if (g.add_conv("decoder.conv_out", wo, bo, 128, 3, 3, 1, 1, tn, *out, nullptr, 0, 0)) return -1;
    g.named["decoder.conv_out"] = *out;
    return 0;
}

// ------------------------------------------------------------------------------------------ VAE encoder
// AutoencoderKL.encode of sd-vae (avatars/musetalk/models/vae.py:84-94 encode_latents; avatar preparation,
// avatars/musetalk/genavatar.py:116-128): conv_in, 4 down blocks x 2 resnets with asymmetric-pad stride-2 convs,
// mid (resnet, attention, resnet), GN-SiLU-conv_out (8 = mean | logvar), quant_conv.
int mt_build_vae_encoder(MtGraph& g, const ltk_named_tensor* t, int n, MtTensor* img_in, MtTensor* moments) {
    SD sd{t, n, ""};
    *img_in = g.alloc(3, 256, 256);
    const float* wi = sd.get("encoder.conv_in.weight", (size_t)128 * 3 * 9);
    const float* bi = sd.get("encoder.conv_in.bias", 128);
    if (!wi || !bi) { g.err = sd.err; return -1; }
    MtTensor h = g.alloc(128, 256, 256);
    if (g.add_conv("encoder.conv_in", wi, bi, 3, 128, 3, 1, 1, *img_in, h, nullptr, 0, 0)) return -1;
    const int chs[4] = {128, 256, 512, 512};
    int C = 128;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) {
            MtTensor r = g.alloc(chs[i], h.H, h.W);
            if (build_resnet(g, sd, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), h, r, C, chs[i], nullptr, 1e-6f)) return -1;
            C = chs[i];
            h = r;
        }
        if (i < 3) {
            const std::string dp = "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
            const float* w = sd.get(dp + ".weight", (size_t)C * C * 9);
            const float* b = sd.get(dp + ".bias", C);
            if (!w || !b) { g.err = sd.err; return -1; }
            MtTensor d = g.alloc(C, h.H / 2, h.W / 2);
            if (g.add_conv_down_asym(dp, w, b, C, h, d)) return -1;
            h = d;
        }
        g.named["encoder.down_blocks." + std::to_string(i)] = h;
    }
    {
        MtTensor r0 = g.alloc(512, 32, 32);
        if (build_resnet(g, sd, "encoder.mid_block.resnets.0", h, r0, 512, 512, nullptr, 1e-6f)) return -1;
        const std::string a = "encoder.mid_block.attentions.0";
        MtTensor gn = g.alloc(512, 32, 32), ao = g.alloc(512, 32, 32), r1 = g.alloc(512, 32, 32);
        if (g.add_gn(a + ".group_norm", sd, a + ".group_norm", r0, gn, 1e-6f, 0)) return -1;
        if (build_attention(g, sd, a, gn, gn, 512, 512, 1, true, r0, ao)) return -1;
        if (build_resnet(g, sd, "encoder.mid_block.resnets.1", ao, r1, 512, 512, nullptr, 1e-6f)) return -1;
        h = r1;
        g.named["encoder.mid_block"] = h;
    }
    MtTensor tn = g.alloc(512, 32, 32), co = g.alloc(8, 32, 32);
    if (g.add_gn("encoder.conv_norm_out", sd, "encoder.conv_norm_out", h, tn, 1e-6f, 1)) return -1;
    const float* wo = sd.get("encoder.conv_out.weight", (size_t)8 * 512 * 9);
    const float* bo = sd.get("encoder.conv_out.bias", 8);
    const float* wq = sd.get("quant_conv.weight", 64);
    const float* bq = sd.get("quant_conv.bias", 8);
    if (!wo || !bo || !wq || !bq) { g.err = sd.err; return -1; }
    if (g.add_conv("encoder.conv_out", wo, bo, 512, 8, 3, 1, 1, tn, co, nullptr, 0, 0)) return -1;
    *moments = g.alloc(8, 32, 32);
    if (g.add_conv("quant_conv", wq, bq, 8, 8, 1, 1, 0, co, *moments, nullptr, 0, 0)) return -1;
    return 0;
}

// ------------------------------------------------------------------------------------------ Whisper encoder
// transformers WhisperEncoder (whisper-tiny: d 384, 4 layers, 6 heads, ffn 1536), the Audio2Feature model
// (avatars/musetalk/whisper/audio2feature.py:15-23,106-117).  In-tree statement of the same encoder:
// avatars/musetalk/whisper/whisper/model.py (AudioEncoder, ResidualAttentionBlock).  state_dict = model.encoder's.
// x: log-mel [80][3000] as a [3000][1] token map.  states[5] = hidden_states (embeddings, layers 0-2, final LN).
int mt_build_whisper(MtGraph& g, const ltk_named_tensor* t, int n, MtTensor* mel_in, MtTensor states[5], int* pos_vec) {
    SD sd{t, n, ""};
    const int D = 384, L = 4, HEADS = 6, FF = 1536, T0 = 3000, T = 1500;
    *mel_in = g.alloc(80, T0, 1);
    g.named["input_features"] = *mel_in;
    const float* w1 = sd.get("conv1.weight", (size_t)D * 80 * 3);
    const float* b1 = sd.get("conv1.bias", D);
    const float* w2 = sd.get("conv2.weight", (size_t)D * D * 3);
    const float* b2 = sd.get("conv2.bias", D);
    const float* pos = sd.get("embed_positions.weight", (size_t)T * D);
    if (!w1 || !b1 || !w2 || !b2 || !pos) { g.err = sd.err; return -1; }
    MtTensor c1 = g.alloc(D, T0, 1), h = g.alloc(D, T, 1);
    if (g.add_conv2("conv1", w1, b1, 80, D, 3, 1, 1, 1, 1, 0, *mel_in, c1, nullptr, 2, 0)) return -1;     // GELU
    if (g.add_conv2("conv2", w2, b2, D, D, 3, 1, 2, 1, 1, 0, c1, h, nullptr, 2, 0)) return -1;            // stride 2, GELU
    *pos_vec = g.add_vec(pos, T * D);
    {
        MtOp op;
        op.type = OP_ADDPOS; op.name = "embed_positions"; op.x = h; op.y = h; op.gamma = *pos_vec;
        g.ops.push_back(op);
        g.named["embed_positions"] = h;
    }
    states[0] = h;
    for (int l = 0; l < L; ++l) {
        const std::string p = "layers." + std::to_string(l);
        MtTensor n1 = g.alloc(D, T, 1), h1 = g.alloc(D, T, 1);
        if (g.add_ln(p + ".self_attn_layer_norm", sd, p + ".self_attn_layer_norm", h, n1, 1e-5f)) return -1;
        // WhisperAttention: q (bias, scaled by d^-0.5), k (no bias), v (bias), out (bias)
        const int d = D / HEADS;
        const float scale = 1.0f / sqrtf((float)d);
        const float* wq = sd.get(p + ".self_attn.q_proj.weight", (size_t)D * D);
        const float* bq = sd.get(p + ".self_attn.q_proj.bias", D);
        const float* wk = sd.get(p + ".self_attn.k_proj.weight", (size_t)D * D);
        const float* wv = sd.get(p + ".self_attn.v_proj.weight", (size_t)D * D);
        const float* bv = sd.get(p + ".self_attn.v_proj.bias", D);
        const float* wo = sd.get(p + ".self_attn.out_proj.weight", (size_t)D * D);
        const float* bo = sd.get(p + ".self_attn.out_proj.bias", D);
        if (!wq || !bq || !wk || !wv || !bv || !wo || !bo) { g.err = sd.err; return -1; }
        std::vector<float> wqs((size_t)D * D), bqs(D);
        for (size_t i = 0; i < wqs.size(); ++i) wqs[i] = wq[i] * scale;
        for (int i = 0; i < D; ++i) bqs[i] = bq[i] * scale;
        MtTensor q = g.alloc(D, T, 1), k = g.alloc(D, T, 1), v = g.alloc(D, T, 1), o = g.alloc(D, T, 1);
        if (g.add_conv(p + ".self_attn.q_proj", wqs.data(), bqs.data(), D, D, 1, 1, 0, n1, q, nullptr, 0, 0)) return -1;
        if (g.add_conv(p + ".self_attn.k_proj", wk, nullptr, D, D, 1, 1, 0, n1, k, nullptr, 0, 0)) return -1;
        if (g.add_conv(p + ".self_attn.v_proj", wv, bv, D, D, 1, 1, 0, n1, v, nullptr, 0, 0)) return -1;
        g.add_attn(p + ".self_attn", q, k, v, o, HEADS, d);
        if (g.add_conv(p + ".self_attn.out_proj", wo, bo, D, D, 1, 1, 0, o, h1, &h, 0, 0)) return -1;
        MtTensor n2 = g.alloc(D, T, 1), f1 = g.alloc(FF, T, 1), h2 = g.alloc(D, T, 1);
        if (g.add_ln(p + ".final_layer_norm", sd, p + ".final_layer_norm", h1, n2, 1e-5f)) return -1;
        const float* wf1 = sd.get(p + ".fc1.weight", (size_t)FF * D);
        const float* bf1 = sd.get(p + ".fc1.bias", FF);
        const float* wf2 = sd.get(p + ".fc2.weight", (size_t)D * FF);
        const float* bf2 = sd.get(p + ".fc2.bias", D);
        if (!wf1 || !bf1 || !wf2 || !bf2) { g.err = sd.err; return -1; }
        if (g.add_conv(p + ".fc1", wf1, bf1, D, FF, 1, 1, 0, n2, f1, nullptr, 2, 0)) return -1;               // GELU
        if (g.add_conv(p + ".fc2", wf2, bf2, FF, D, 1, 1, 0, f1, h2, &h1, 0, 0)) return -1;
        h = h2;
        if (l + 1 < L) states[l + 1] = h;
    }
    MtTensor fin = g.alloc(D, T, 1);
    if (g.add_ln("layer_norm", sd, "layer_norm", h, fin, 1e-5f)) return -1;
    states[4] = fin;
    for (int i = 0; i < 5; ++i) g.named["hidden_states." + std::to_string(i)] = states[i];
    return 0;
}

// ------------------------------------------------------------------------------------------ allocate / run / free
int mt_graph_alloc(MtGraph& g, int frames) {
    g.frames = frames;
    g.bufs.assign(g.buf_halfs.size(), nullptr);
    for (size_t i = 0; i < g.buf_halfs.size(); ++i) {
        const size_t bytes = g.buf_halfs[i] * frames * sizeof(f16) + 256;
        if (hipMalloc((void**)&g.bufs[i], bytes) != hipSuccess) { g.err = "activation allocation failed"; return -4; }
        (void)hipMemset(g.bufs[i], 0, bytes);
    }
    // GroupNorm partial stats: [N][C/16][segs][32] floats, C <= 2560, segs <= 256 -> bound by the op list
    size_t need = 0;
    for (const MtOp& op : g.ops)
        if (op.type == OP_GN) need = std::max(need, (size_t)frames * (op.x.C / 16) * gn_segments(frames, op.x.C, op.x.P()) * 32);
    // the segment count is chosen per launch from the launch's frame count: size for the worst case (1 frame)
    for (const MtOp& op : g.ops)
        if (op.type == OP_GN) need = std::max(need, (size_t)frames * (op.x.C / 16) * 256 * 32);
    g.gn_partial_floats = need;
    if (hipMalloc((void**)&g.gn_partial, need * sizeof(float)) != hipSuccess) { g.err = "allocation failed"; return -4; }
    // gn_coop_kernel: 8 words per block, a region per op (a launch of nf <= frames images uses the head of its region)
    g.gn_slot_words = 0;
    for (MtOp& op : g.ops) {
        op.gn_slot_off = -1;
        if (op.type != OP_GN || gn_group_fits(op.x.C, op.x.P(), op.groups)) continue;
        const int M = gn_coop_members(op.x.C, op.x.P(), op.groups);
        if (!M) continue;
        op.gn_slot_off = (long long)g.gn_slot_words;
        g.gn_slot_words += (size_t)frames * (op.x.C / 16) * (op.x.P() / 2048) * 8;      // sized for 2048-pixel slices, the smallest the kernel uses
    }
    if (g.gn_slot_words) {
        if (hipMalloc((void**)&g.gn_slots, g.gn_slot_words * sizeof(unsigned)) != hipSuccess) { g.err = "allocation failed"; return -4; }
        if (hipHostMalloc((void**)&g.gn_err_host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) { g.err = "allocation failed"; return -4; }
        *g.gn_err_host = 0u;
        if (hipHostGetDevicePointer((void**)&g.gn_err_dev, g.gn_err_host, 0) != hipSuccess) { g.err = "allocation failed"; return -4; }
    }
    if (g.vt_halfs) {
        if (hipMalloc((void**)&g.vt, g.vt_halfs * frames * sizeof(f16)) != hipSuccess) { g.err = "allocation failed"; return -4; }
    }
    return 0;
}

void mt_graph_free(MtGraph& g) {
    for (f16* b : g.bufs) if (b) (void)hipFree(b);
    for (ConvPlan& p : g.plans) conv_plan_destroy(&p);
    for (RowGemmPlan& p : g.rplans) rowgemm_plan_destroy(&p);
    for (float* v : g.vecs) if (v) (void)hipFree(v);
    if (g.gn_partial) (void)hipFree(g.gn_partial);
    if (g.gn_slots) (void)hipFree(g.gn_slots);
    if (g.gn_err_host) (void)hipHostFree(g.gn_err_host);
    g.gn_slots = nullptr; g.gn_err_host = nullptr; g.gn_err_dev = nullptr;
    if (g.vt) (void)hipFree(g.vt);
    g.bufs.clear(); g.plans.clear(); g.rplans.clear(); g.vecs.clear();
}

f16* mt_ptr(const MtGraph& g, const MtTensor& t) { return g.bufs[t.buf]; }

// one op on stream `s`
static int mt_run_op_body(MtGraph& g, const MtOp& op, int nf, float* partial, size_t partial_cap, hipStream_t s) {
    switch (op.type) {
        case OP_CONV: {
            ConvIO io;
            io.x = g.bufs[op.x.buf]; io.N = nf; io.H = op.x.H; io.W = op.x.W; io.x_ld = op.x.ld; io.x_coff = op.x.coff;
            if (op.x.q8) { io.x_ld /= 2; io.x_coff /= 2; }      // 16-bit units of the fp8 tensor (conv_mfma.h: ConvPlan::q8)
            io.y = g.bufs[op.y.buf]; io.y_ld = op.y.ld; io.y_coff = op.y.coff;
            io.res = op.r.buf >= 0 ? g.bufs[op.r.buf] : nullptr; io.res_ld = op.r.ld; io.res_coff = op.r.coff;
            io.relu = 0; io.act = op.act; io.ups = op.ups;
            io.partial = partial; io.partial_cap = partial_cap;
            if (op.ln_out_buf >= 0) { io.ln_out = reinterpret_cast<float*>(g.bufs[op.ln_out_buf]); io.ln_out_tiles = op.y.C / 32; }
            if (op.ln_in_buf >= 0) { io.ln_in = reinterpret_cast<const float*>(g.bufs[op.ln_in_buf]); io.ln_in_tiles = op.ln_in_tiles; io.ln_eps = op.ln_eps; }
            // U-Net resnet convs of a <= 16-frame pass: conv3's items-per-CU rule settles on tiles that re-read weights (8x8, 32x32 levels) or
            // under-fill the chip (16x16 level); measured per level with the tile forced for every 3x3 launch (profiles/r02_mt_tile_force_ab.txt:
            // 8x8 1280..2560 ch 116 -> 89 us at 256-px tiles, 16x16 640 ch 67 -> 49 us at 128-px tiles, 32x32 320 ch 55 -> 44 us at 256-px tiles).
            // The VAE decoder keeps the rule (it wants its 512-px tiles).
            // (The 4x4 level keeps the rule: forced 256-px tiles measured 31 -> 37 us there, profiles/r04_mt_rowconv_tile_ab.txt.)
            if (op.unet3x3 && nf <= 16 && knob(K_MT_TILE_TABLE) && op.x.P() >= 64) io.force_pxw = op.x.P() == 256 ? 1 : 2;
            std::string e;
            int rc;
            // (K = 1280 linear layers from LIN_FK_MIN_ROWS rows on - the 8^2 level of a 16-frame pass - take conv3_launch's lin_fk route)
            const bool lin_fk = knob(K_LIN_FK) && op.ksz == 1 && (long long)nf * op.y.P() >= knob(K_LIN_FK_MIN_ROWS) &&
                                (conv3_lin_fk_k(g.plans[op.plan].Cin) || conv3_lin_mp_nsl(g.plans[op.plan].Cin, (long long)nf * op.y.P(), g.plans[op.plan].lCout) > 0);
            if (op.rplan >= 0 && !lin_fk && (long long)nf * op.y.P() <= std::min(knob(K_MT_ROWCONV), kRowConvMaxRows)) {
                RowConvIO rio;
                rio.x = io.x; rio.x_ld = op.x.ld; rio.x_coff = op.x.coff; rio.H = op.x.H; rio.W = op.x.W;
                rio.y = io.y; rio.y_ld = op.y.ld; rio.y_coff = op.y.coff; rio.Ho = op.y.H; rio.Wo = op.y.W;
                rio.res = io.res; rio.res_ld = io.res_ld; rio.res_coff = io.res_coff;
                rio.N = nf; rio.KW = op.ksz; rio.stride = 1; rio.pad = op.ksz / 2; rio.relu = 0;
                rio.ln_out = io.ln_out; rio.ln_out_tiles = io.ln_out_tiles; rio.ln_in = io.ln_in; rio.ln_in_tiles = io.ln_in_tiles; rio.ln_eps = io.ln_eps;
                rc = rowconv_launch(g.rplans[op.rplan], rio, s, &e);
            } else {
                rc = conv_launch(g.plans[op.plan], io, s, &e);
            }
            if (rc) { g.err = op.name + ": " + e; return rc; }
            break;
        }
        case OP_GN: {
            const int P = op.x.P();
            if (knob(K_MT_GN1) && gn_group_fits(op.x.C, P, op.groups)) {      // one launch: block = (image, group)
                launch_gn_group(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, P, op.groups, op.eps, g.vecs[op.gamma],
                                g.vecs[op.beta], op.silu, g.bufs[op.y.buf], op.y.q8 ? op.y.ld / 32 : op.y.ld / 16,
                                op.y.q8 ? op.y.coff / 32 : op.y.coff / 16, op.y.q8 ? 1 : 0, g.fp8_ascale, s);
                break;
            }
            if (knob(K_GN_COOP) && op.gn_slot_off >= 0 && g.gn_slots) {           // one tensor pass: slices in registers, partial sums exchanged between blocks
                launch_gn_coop(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, P, op.groups, op.eps, g.gn_slots + op.gn_slot_off, g.gn_err_dev,
                               g.vecs[op.gamma], g.vecs[op.beta], op.silu, g.bufs[op.y.buf], op.y.q8 ? op.y.ld / 32 : op.y.ld / 16,
                               op.y.q8 ? op.y.coff / 32 : op.y.coff / 16, op.y.q8 ? 1 : 0, g.fp8_ascale, s);
                break;
            }
            const int segs = gn_segments(nf, op.x.C, P);
            launch_gn_stats(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, P, segs, g.gn_partial, s);
            if (op.y.q8)
                launch_gn_apply_fp8(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, P, op.groups, op.eps, g.gn_partial, segs,
                                    g.vecs[op.gamma], g.vecs[op.beta], op.silu, g.fp8_ascale, (unsigned char*)g.bufs[op.y.buf],
                                    op.y.ld / 32, op.y.coff / 32, s);
            else
                launch_gn_apply(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, P, op.groups, op.eps, g.gn_partial, segs,
                                g.vecs[op.gamma], g.vecs[op.beta], op.silu, g.bufs[op.y.buf], op.y.ld / 16, op.y.coff / 16, s);
            break;
        }
        case OP_LN:
            launch_layernorm(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, op.x.P(), op.eps, g.vecs[op.gamma],
                             g.vecs[op.beta], g.bufs[op.y.buf], op.y.ld / 16, op.y.coff / 16, s);
            break;
        case OP_VT: {
            VtMulti m;
            m.n = (int)g.vt_items.size(); m.Tk = g.vt_items.empty() ? 0 : g.vt_items[0].Tk; m.Tkp = 0;
            if (m.n > 16) { g.err = "too many hoisted value tensors"; return -1; }
            for (int i = 0; i < m.n; ++i) {
                const MtVtItem& it = g.vt_items[i];
                m.it[i] = {g.bufs[it.v.buf], g.bufs[it.vt_buf], it.v.ld / 16, it.v.coff / 16, it.heads, it.d16, 0, 0};
            }
            launch_v_transpose_multi(m, nf, s);
            break;
        }
        case OP_ATTN: {
            f16* vt = g.vt;
            if (op.vt_buf >= 0) vt = g.bufs[op.vt_buf];
            else launch_v_transpose(g.bufs[op.v.buf], nf, op.v.ld / 16, op.v.coff / 16, op.heads, op.d16, op.Tk, g.vt, s);
            const int rc = launch_attention(g.bufs[op.x.buf], op.x.ld / 16, op.x.coff / 16, op.x.P(), g.bufs[op.k.buf], op.k.ld / 16,
                                            op.k.coff / 16, op.Tk, vt, g.bufs[op.y.buf], op.y.ld / 16, op.y.coff / 16, nf, op.heads,
                                            op.d16, s);
            if (rc) { g.err = op.name + ": attention launch failed (head dim " + std::to_string(op.d16) + ")"; return rc; }
            break;
        }
        case OP_ADDPOS:
            launch_add_pos(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.x.C, op.x.P(), g.vecs[op.gamma], s);
            break;
        case OP_GEGLU:
            launch_geglu(g.bufs[op.x.buf], nf, op.x.ld / 16, op.x.coff / 16, op.y.C, op.x.P(), g.bufs[op.y.buf], op.y.ld / 16,
                         op.y.coff / 16, s);
            break;
    }
    return 0;
}

static int mt_run_op(MtGraph& g, const MtOp& op, int nf, float* partial, size_t partial_cap, hipStream_t s) {
    const int rc = mt_run_op_body(g, op, nf, partial, partial_cap, s);
    if (!rc && g.sat_ctr && op.y.buf >= 0 && knob(K_SAT_CHECK)) {      // debug: what this op clamped to (or pushed past) the limit of its output type
        const int gran = op.y.q8 ? 32 : 16;
        launch_sat_scan(g.bufs[op.y.buf], nf, op.y.ld / gran, op.y.coff / gran, op.y.C / gran, op.y.P(), op.y.q8 ? 1 : 0, g.sat_ctr, s);
    }
    return rc;
}

// `evs` (measurement): one event in front of every op and one behind the last
int mt_graph_run(MtGraph& g, int nf, float* partial, size_t partial_cap, hipStream_t s, int op_begin, int op_end,
                 std::vector<hipEvent_t>* evs = nullptr) {
    if (nf > g.frames) { g.err = "more frames than the graph was sized for"; return -1; }
    if (op_end < 0) op_end = (int)g.ops.size();
    if (g.gn_slots && knob(K_GN_COOP)) {                 // the exchange slots of this range's cooperative GroupNorms back to the sentinel (one fill per pass)
        long long lo = -1, hi = -1;
        for (int oi = op_begin; oi < op_end; ++oi) {
            const MtOp& op = g.ops[oi];
            if (op.type != OP_GN || op.gn_slot_off < 0) continue;
            const long long words = (long long)g.frames * (op.x.C / 16) * (op.x.P() / 2048) * 8;
            if (lo < 0) lo = op.gn_slot_off;
            hi = op.gn_slot_off + words;
        }
        if (lo >= 0) launch_gn_coop_reset(g.gn_slots + lo, (size_t)(hi - lo), s);
    }
    for (int oi = op_begin; oi < op_end; ++oi) {
        if (evs) (void)hipEventRecord((*evs)[oi - op_begin], s);
        const int rc = mt_run_op(g, g.ops[oi], nf, partial, partial_cap, s);
        if (rc) return rc;
    }
    if (evs) (void)hipEventRecord((*evs)[op_end - op_begin], s);
    if (hipGetLastError() != hipSuccess) { g.err = "a MuseTalk kernel launch failed"; return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------ public wrappers
int mt_op_count(MtGraph* g) { return (int)g->ops.size(); }
const char* mt_op_name(MtGraph* g, int i, int* type) {
    if (i < 0 || i >= (int)g->ops.size()) return nullptr;
    if (type) *type = (int)g->ops[i].type;
    return g->ops[i].name.c_str();
}
int mt_run_timed(MtGraph* g, int nf, float* partial, size_t partial_cap, hipStream_t s, std::vector<hipEvent_t>* evs) {
    return mt_graph_run(*g, nf, partial, partial_cap, s, 0, -1, evs);
}
MtGraph* mt_graph_new() { return new MtGraph(); }
void mt_graph_delete(MtGraph* g) {
    if (!g) return;
    mt_graph_free(*g);
    delete g->t_latent; delete g->t_ctx; delete g->t_unet_out; delete g->t_vae_out;
    delete[] g->whisper_states;
    delete g;
}
const char* mt_graph_error(const MtGraph* g) { return g->err.c_str(); }
int mt_gn_error(MtGraph* g) {
    if (!g || !g->gn_err_host || !*reinterpret_cast<volatile unsigned*>(g->gn_err_host)) return 0;
    *g->gn_err_host = 0u;
    g->err = "a cooperative GroupNorm block gave up waiting for its set (gn_coop_kernel): the pass's frames are invalid; LTK_GN_COOP=0 selects the two-pass kernels";
    return 1;
}

void mt_set_sat_counter(MtGraph* g, unsigned long long* d_ctr) { g->sat_ctr = d_ctr; }
void mt_set_fp8(MtGraph* g, int on, float act_scale) { g->fp8 = on != 0; if (act_scale > 0.f) g->fp8_ascale = act_scale; }
double mt_macs_fp8_per_frame(const MtGraph* g) { return g->macs_fp8; }

int mt_build(MtGraph* g, const ltk_named_tensor* unet_sd, int n_unet, const ltk_named_tensor* vae_sd, int n_vae, int frames) {
    g->t_latent = new MtTensor(); g->t_ctx = new MtTensor(); g->t_unet_out = new MtTensor(); g->t_vae_out = new MtTensor();
    if (mt_build_unet(*g, unet_sd, n_unet, g->t_latent, g->t_ctx, g->t_unet_out)) return -1;
    if (mt_build_vae(*g, vae_sd, n_vae, *g->t_unet_out, g->t_vae_out)) return -1;
    return mt_graph_alloc(*g, frames);
}
static f16* tptr(MtGraph* g, const MtTensor* t, int* cbt) { if (cbt) *cbt = t->ld / 16; return g->bufs[t->buf]; }
f16* mt_latent_in(MtGraph* g, int* cbt) { return tptr(g, g->t_latent, cbt); }
f16* mt_ctx_in(MtGraph* g, int* cbt) { return tptr(g, g->t_ctx, cbt); }
f16* mt_unet_out(MtGraph* g, int* cbt) { return tptr(g, g->t_unet_out, cbt); }
f16* mt_vae_out(MtGraph* g, int* cbt) { return tptr(g, g->t_vae_out, cbt); }
int mt_run(MtGraph* g, int nf, float* partial, size_t partial_cap, hipStream_t s) { return mt_graph_run(*g, nf, partial, partial_cap, s, 0, -1); }
f16* mt_named(MtGraph* g, const char* name, int* C, int* ld, int* coff, int* H, int* W) {
    auto it = g->named.find(name);
    if (it == g->named.end()) return nullptr;
    const MtTensor& t = it->second;
    *C = t.C; *ld = t.ld; *coff = t.coff; *H = t.H; *W = t.W;
    return g->bufs[t.buf];
}
int mt_build_whisper_graph(MtGraph* g, const ltk_named_tensor* sd, int n) {
    g->t_latent = new MtTensor();                       // reused as the log-mel input tensor
    g->whisper_states = new MtTensor[5];
    int pos_vec = -1;
    if (mt_build_whisper(*g, sd, n, g->t_latent, g->whisper_states, &pos_vec)) return -1;
    return mt_graph_alloc(*g, 1);
}
f16* mt_whisper_state(MtGraph* g, int i, int* cbt, int* cb0) {
    const MtTensor& t = g->whisper_states[i];
    *cbt = t.ld / 16; *cb0 = t.coff / 16;
    return g->bufs[t.buf];
}
int mt_build_vae_encoder_graph(MtGraph* g, const ltk_named_tensor* sd, int n, int frames) {
    g->t_latent = new MtTensor();          // image input
    g->t_unet_out = new MtTensor();        // moments output
    if (mt_build_vae_encoder(*g, sd, n, g->t_latent, g->t_unet_out)) return -1;
    return mt_graph_alloc(*g, frames);
}
double mt_macs_per_frame(const MtGraph* g) { return g->macs; }

}  // namespace ltk
