// rowgemm: the bottleneck layers of the Wav2Lip generator whose maps are ONE pixel per frame
// (avatars/wav2lip/models/wav2lip_v2.py:36-39 face_encoder_blocks.7 = Conv2d(512,512,4,1,0) on the 4x4 map + Conv2d(512,512,1);
//  :60 face_decoder_blocks.0 = Conv2d(512,512,1) on the audio embedding; :62 face_decoder_blocks.1.0 = ConvTranspose2d(1024,512,4,1,0)
//  on the 1x1 map), i.e. plain GEMMs  Y[frame][j] = act(scale[j] * sum_k X[frame][k] * W[j][k] + shift[j])  with as many rows as
// there are frames in the launch: 16 for a session's step.  conv3 ran them as 1x1 convs with a split-K finish launch each: 70 us of a
// 16-frame pass for 26 MB of weights that stream in ~5 us.  Here a block owns 16 output channels and ALL of K: its 8 waves split K,
// every wave streams its weight fragments straight from HBM into registers (they are used once: no LDS staging), feeds
// v_mfma_f32_16x16x32_f16 with the frames as the other operand, and the 8 partial tiles meet in LDS in a fixed order - no
// split-K slabs, no finish launch, deterministic.  Used for launches of <= 32 frames; larger ones have enough rows for conv3.
#include "conv_mfma.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <vector>

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RowGemmArgs {
    const f16* x; int x_ld, x_coff;      // X[frame][x_coff + k], row pitch x_ld halfs (a one-pixel CB16 map is exactly that)
    f16* y; int y_ld, y_coff;
    const f16* w;                         // packed [J/16][K/32][64 lanes][8]: lane (i = l & 15, g = l >> 4) holds W[jt*16 + i][kt*32 + g*8 .. +8]
    const float* scale; const float* shift;
    int M, K, J, relu;
};

// FT: 16-frame tiles per block (M <= 16 * FT)
template <int FT>
__global__ __launch_bounds__(512) void rowgemm_kernel(const RowGemmArgs a) {
    __shared__ f32x4 red[8][FT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jt = blockIdx.x;
    const int KT = a.K >> 5;                                   // k-steps of 32
    const int per = (KT + 7) >> 3;
    const int k0 = wave * per, k1 = min(KT, k0 + per);
    const int i16 = lane & 15, g = lane >> 4;
    const f16x8* wp = reinterpret_cast<const f16x8*>(a.w) + ((size_t)jt * KT) * 64 + lane;
    const f16* xrow[FT];
    bool live[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        const int fr = ft * 16 + i16;
        live[ft] = fr < a.M;
        xrow[ft] = a.x + (size_t)(live[ft] ? fr : 0) * a.x_ld + a.x_coff + g * 8;
    }
    f32x4 acc[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) acc[ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    int kt = k0;
    // U weight fragments (U KiB per wave, 8 waves per CU) in flight per trip: hipcc waits for a trip's loads before its MFMAs and
    // issues the next trip's loads behind them, so a wave pays one HBM round trip per trip - the K = 8192 layer (8.4 MB behind
    // only 32 blocks, 32 steps per wave) takes 16 us with 4, 8 or 16 steps per trip alike, so the trip depth is not what bounds it
    constexpr int U = FT == 1 ? 8 : 4;
    for (; kt + U <= k1; kt += U) {
        f16x8 wa[U], xb[U][FT];
#pragma unroll
        for (int u = 0; u < U; ++u) wa[u] = __builtin_nontemporal_load(wp + (size_t)(kt + u) * 64);      // streamed once
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) xb[u][ft] = live[ft] ? *reinterpret_cast<const f16x8*>(xrow[ft] + (kt + u) * 32) : zero;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u], xb[u][ft], acc[ft], 0, 0, 0);
    }
    for (; kt < k1; ++kt) {
        const f16x8 wa = __builtin_nontemporal_load(wp + (size_t)kt * 64);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            const f16x8 xb = live[ft] ? *reinterpret_cast<const f16x8*>(xrow[ft] + kt * 32) : zero;
            acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb, acc[ft], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) red[wave][ft][lane] = acc[ft];
    __syncthreads();
    if (wave >= FT) return;                                    // wave ft finishes frame tile ft
    const int ft = wave;
    f32x4 s = red[0][ft][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) {                              // fixed order: deterministic
        const f32x4 t = red[w][ft][lane];
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    // D layout of the 16x16 MFMA: lane holds rows (couts) 4g .. 4g+3 of column (frame) i16
    const int j0 = jt * 16 + 4 * g, fr = ft * 16 + i16;
    if (fr >= a.M) return;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + j0), sf = *reinterpret_cast<const f32x4*>(a.shift + j0);
    f16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = s[r] * sc[r] + sf[r];
        v = a.relu ? __builtin_amdgcn_fmed3f(v, 0.f, 65504.f) : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
        o[r] = (f16)v;
    }
    *reinterpret_cast<f16x4*>(a.y + (size_t)fr * a.y_ld + a.y_coff + j0) = o;
}

// ------------------------------------------------------------------------------------------
// rowconv: the same weight-streaming GEMM for the 3x3 convolutions on the 4x4 and 8x8 maps (face_encoder_blocks.5.* / 6.*,
// face_decoder_blocks.1.1 / 2.1: wav2lip_v2.py:32-36,63-66), whose 16-frame launches have only 256 / 1024 output pixels behind
// 4.7 MB of weights each: rows = output pixels (frame, oy, ox), K = (tap, channel) with the im2col row of a pixel GATHERED from the
// channel-blocked input map on the fly (16 bytes = 8 channels of one input pixel per lane and k-step, zeros outside the map: the
// activations of these layers are 0.26 - 1 MB and live in L2).  A block owns 32 output channels x 16*FT rows and all of K; its 8
// waves split K, stream the weight fragments HBM -> registers and meet in LDS in wave order (deterministic).  Blocks that share
// weights (same 32 channels, different rows) sit on ONE XCD, so every weight byte leaves HBM once.  conv3 ran these layers as
// 128..256 items of 4..8 chunks each behind a two-stage DMA pipe + a split-K finish launch: 18-22 us for ~2 us of work.
struct RowConvArgs {
    const f16* x; int x_cbt, x_cb0;       // input  [N][x_cbt][H*W][16], this tensor's first channel block
    f16* y; int y_cbt, y_cb0;             // output [N][y_cbt][Ho*Wo][16]
    const f16* res; int res_cbt, res_cb0; // residual (same geometry as y) or nullptr
    const f16* w;                         // packed [J/16][K/32][64][8], K = taps * C ordered (tap, channel)
    const float* scale; const float* shift;
    int N, H, W, Ho, Wo, S, Sx, pad, KW;  // S / Sx: row / column stride; KW: kernel width (taps = KW * KW)
    int cpt;                              // C / 32: k-steps per tap (any C % 32 == 0)
    unsigned cmagic;                      // ceil(2^32 / cpt): tap = umulhi(k, cmagic) for cpt > 1
    int KT;                               // k-steps: taps * C / 32
    int M;                                // rows: N * Ho * Wo
    int NR;                               // row groups (16 * FT rows each)
    int JB;                               // J / 32 channel pairs
    int relu;
    // Sub-pixel phases of a stride-2 transposed conv (nph = 4, gridDim.y = phase; nph = 1: a plain conv, the fields above).  Rows are
    // SOURCE pixels (n, y, x) of the H x W = Ho x Wo map; phase (py, px) owns output pixel (2y + py, 2x + px) of the 2H x 2W map and
    // contracts its own 1 / 2 / 2 / 4 taps (dy, dx) = (t / kw, t % kw) of the source window at (y + dy, x + dx).
    int nph;
    int Wout, HWout;                      // output map row pitch and plane size used for addressing (= Wo, Ho * Wo when nph == 1)
    struct Phase { const f16* w; const float* scale; const float* shift; int KT, KW, oy_add, ox_add; } ph[4];
    // LayerNorm fold (1x1 plans; conv3_mfma.hip K3Args has the algebra): ln_out = [rows][ln_out_tiles] float2 partial sums of the stored
    // output (this block's 32 channels = tile jb), ln_in = the same of the input rows
    float* ln_out; const float* ln_in;
    int ln_out_tiles, ln_in_tiles;
    float ln_eps;
};

// FT: 16-row tiles per block; UB: k-steps in flight per trip.  Measured (profiles/r03_rowconv_ab.txt): deeper trips (9 / 6 steps), 32-row
// blocks at every size and two resident blocks per CU (<= 128 VGPRs) are all equal or slower than <2, 6> up to 512 row tiles x channel
// pairs and <4, 4> above.
template <int FT, int UB>
__global__ __launch_bounds__(512) void rowconv_kernel(const RowConvArgs a) {
    constexpr int JT = 2;                                      // 32 output channels per block
    __shared__ f32x4 red[8][FT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: hardware block b runs on XCD b & 7; all row groups of a channel pair share their weights through that XCD's L2
    // (JB = J / 32 channel pairs, a multiple of 8); layers with fewer pairs (JB = 4: the audio encoder's 128-channel layers) take the
    // plain order - their weights are a few hundred KB
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int jb = (a.JB & 7) ? (int)(blockIdx.x % (unsigned)a.JB) : xcd + 8 * (slot / a.NR);
    const int rg = (a.JB & 7) ? (int)(blockIdx.x / (unsigned)a.JB) : slot % a.NR;
    const bool phased = a.nph > 1;
    const int pz = phased ? (int)blockIdx.y : 0;
    const int KT = phased ? a.ph[pz].KT : a.KT, KW = phased ? a.ph[pz].KW : a.KW;
    const f16* const wbase = phased ? a.ph[pz].w : a.w;
    const float* const scale = phased ? a.ph[pz].scale : a.scale;
    const float* const shift = phased ? a.ph[pz].shift : a.shift;
    const int omul = phased ? 2 : 1, oy_add = phased ? a.ph[pz].oy_add : 0, ox_add = phased ? a.ph[pz].ox_add : 0;
    const int per = (KT + 7) >> 3;
    const int k0 = wave * per, k1 = min(KT, k0 + per);
    const int i16 = lane & 15, g = lane >> 4;
    const f16x8* wp = reinterpret_cast<const f16x8*>(wbase) + ((size_t)(jb * JT) * KT) * 64 + lane;
    const size_t wj = (size_t)KT * 64;                         // fragments between the block's two 16-channel slabs
    const int HWi = a.H * a.W, HWo = a.Ho * a.Wo;
    // this lane's row of each 16-row tile: frame, top-left input pixel of its window
    int iy0[FT], ix0[FT];
    const f16* xn[FT];
    bool live[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        const int r = (rg * FT + ft) * 16 + i16;
        live[ft] = r < a.M;
        const int rr = live[ft] ? r : 0;
        const int n = rr / HWo, pix = rr - n * HWo;
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        iy0[ft] = oy * a.S - a.pad; ix0[ft] = ox * a.Sx - a.pad;
        // channel block (g >> 1) and half (g & 1) of the k-step's 32 channels belong to this lane
        xn[ft] = a.x + (((size_t)n * a.x_cbt + a.x_cb0 + (g >> 1)) * HWi) * 16 + (g & 1) * 8;
    }
    f32x4 acc[JT][FT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) acc[jt][ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // operand of k-step k for row tile ft: tap = k / cpt (wave-uniform), 32 channels from (k - tap * cpt) * 32
    auto xload = [&](int k, int ft) -> f16x8 {
        const int tap = a.cpt == 1 ? k : (int)__umulhi((unsigned)k, a.cmagic);
        const int cc = k - tap * a.cpt;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int iy = iy0[ft] + ky, ix = ix0[ft] + kx;
        const bool ok = live[ft] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        return ok ? *reinterpret_cast<const f16x8*>(xn[ft] + ((size_t)(cc * 2) * HWi + iy * a.W + ix) * 16) : zero;
    };
    // One trip = U k-steps: all of its loads in flight, then its MFMAs.  hipcc waits for a trip's loads before its MFMAs and issues
    // the next trip's loads behind them, so a wave pays one round trip per trip.  A two-set software pipeline (the loads of trip
    // t+1 issued before the MFMAs of trip t; every load unconditional - zero page for taps outside the map - so that the counted
    // vmcnt waits really leave the newer set outstanding, checked in the ISA) was built and is 1.5 - 4 us SLOWER per layer
    // (profiles/r03_rowconv_ab.txt): with ~200 VGPRs and dead trips it loses more than the overlap returns.
    auto trip = [&](int kt, auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        f16x8 wa[U][JT], xb[U][FT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) wa[u][jt] = wp[jt * wj + (size_t)(kt + u) * 64];      // shared with the XCD's other row groups: cached
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) xb[u][ft] = xload(kt + u, ft);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int ft = 0; ft < FT; ++ft)
                    acc[jt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][jt], xb[u][ft], acc[jt][ft], 0, 0, 0);
    };
    int kt = k0;
    for (; kt + UB <= k1; kt += UB) trip(kt, std::integral_constant<int, UB>{});
    for (; kt + 2 <= k1; kt += 2) trip(kt, std::integral_constant<int, 2>{});
    for (; kt < k1; ++kt) trip(kt, std::integral_constant<int, 1>{});

    // LayerNorm fold: the finishing lane's row statistics (consumer) / running sums of what it stores (producer); row index = token index
    float lmean = 0.f, lrstd = 1.f, lsum = 0.f, lsq = 0.f;
    if (a.ln_in && wave < FT) {
        const int r = (rg * FT + wave) * 16 + i16;
        const float2* pp = reinterpret_cast<const float2*>(a.ln_in) + (size_t)(r < a.M ? r : 0) * a.ln_in_tiles;
        float su = 0.f, sq = 0.f;
        for (int t = g; t < a.ln_in_tiles; t += 4) { const float2 v = pp[t]; su += v.x; sq += v.y; }
        su += __shfl_xor(su, 16); sq += __shfl_xor(sq, 16);
        su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
        const float invC = 1.f / (float)(a.ln_in_tiles * 32);
        lmean = su * invC;
        lrstd = rsqrtf(fmaxf(sq * invC - lmean * lmean, 0.f) + a.ln_eps);
    }
    // ---- the 8 partial tiles meet in LDS, one 16-channel slab at a time; wave ft finishes row tile ft in wave order (deterministic)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        if (jt) __syncthreads();                               // every finishing wave has read the previous slab
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) red[wave][ft][lane] = acc[jt][ft];
        __syncthreads();
        if (wave < FT) {
            const int ft = wave;
            f32x4 s = red[0][ft][lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const f32x4 t = red[w][ft][lane];
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            }
            // D layout of the 16x16 MFMA: lane holds output channels 4g .. 4g+3 of row i16
            const int r = (rg * FT + ft) * 16 + i16;
            if (r < a.M) {
                const int n = r / HWo, rpix = r - n * HWo;
                const int ry = rpix / a.Wo, rx = rpix - ry * a.Wo;
                const int pix = (ry * omul + oy_add) * a.Wout + rx * omul + ox_add;      // output pixel inside its plane
                const int cb = jb * JT + jt;                   // 16-channel block of the output
                const int j0 = cb * 16 + 4 * g;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + j0), sf = *reinterpret_cast<const f32x4*>(shift + j0);
                float v[4];
                if (a.ln_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = lrstd * (s[q] - lmean * sc[q]) + sf[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = s[q] * sc[q] + sf[q];
                }
                if (a.res) {
                    const f16x4 rv = *reinterpret_cast<const f16x4*>(a.res + (((size_t)n * a.res_cbt + a.res_cb0 + cb) * a.HWout + pix) * 16 + 4 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += (float)rv[q];
                }
                f16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = (f16)(a.relu ? __builtin_amdgcn_fmed3f(v[q], 0.f, 65504.f) : __builtin_amdgcn_fmed3f(v[q], -65504.f, 65504.f));
                *reinterpret_cast<f16x4*>(a.y + (((size_t)n * a.y_cbt + a.y_cb0 + cb) * a.HWout + pix) * 16 + 4 * g) = o;
                if (a.ln_out) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float f = (float)o[q]; lsum += f; lsq += f * f; }
                }
            }
        }
    }
    if (a.ln_out && wave < FT) {          // this block's 32 channels of the row: the four channel-group lanes of a row meet, lane group 0 writes
        lsum += __shfl_xor(lsum, 16); lsq += __shfl_xor(lsq, 16);
        lsum += __shfl_xor(lsum, 32); lsq += __shfl_xor(lsq, 32);
        const int r = (rg * FT + wave) * 16 + i16;
        if (g == 0 && r < a.M) reinterpret_cast<float2*>(a.ln_out)[(size_t)r * a.ln_out_tiles + jb] = make_float2(lsum, lsq);
    }
}

int rowgemm_plan_create(RowGemmPlan* p, const float* w_eff, int J, int K, const float* scale, const float* shift, std::string* err) {
    *p = RowGemmPlan();
    if (J % 16 || K % 32) { if (err) *err = "rowgemm: J % 16 == 0 and K % 32 == 0"; return -1; }
    const int KT = K / 32;
    std::vector<f16> packed((size_t)J * K);
    for (int jt = 0; jt < J / 16; ++jt)
        for (int kt = 0; kt < KT; ++kt)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e)
                    packed[(((size_t)jt * KT + kt) * 64 + l) * 8 + e] = (f16)w_eff[(size_t)(jt * 16 + (l & 15)) * K + kt * 32 + (l >> 4) * 8 + e];
    if (hipMalloc((void**)&p->d_w, packed.size() * sizeof(f16)) != hipSuccess ||
        hipMalloc((void**)&p->d_scale, (size_t)2 * J * sizeof(float)) != hipSuccess) {
        rowgemm_plan_destroy(p);
        if (err) *err = "rowgemm: device allocation failed";
        return -2;
    }
    p->d_shift = p->d_scale + J;
    if (hipMemcpy(p->d_w, packed.data(), packed.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_scale, scale, (size_t)J * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_shift, shift, (size_t)J * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        rowgemm_plan_destroy(p);
        if (err) *err = "rowgemm: upload failed";
        return -2;
    }
    p->J = J; p->K = K;
    return 0;
}

void rowgemm_plan_destroy(RowGemmPlan* p) {
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_scale) (void)hipFree(p->d_scale);
    *p = RowGemmPlan();
}

int rowgemm_launch(const RowGemmPlan& p, const f16* x, int x_ld, int x_coff, f16* y, int y_ld, int y_coff, int M, int relu,
                   hipStream_t stream, std::string* err) {
    if (!p.d_w || M <= 0 || M > kRowGemmMaxFrames) { if (err) *err = "rowgemm: no plan, or more frames than it is built for"; return -1; }
    if (((x_ld | x_coff) & 7) || ((y_ld | y_coff) & 3)) { if (err) *err = "rowgemm: operand pitch / offset alignment"; return -1; }
    if (x_coff < 0 || y_coff < 0 || x_coff + p.K > x_ld || y_coff + p.J > y_ld) { if (err) *err = "rowgemm: channel range outside the row pitch"; return -1; }
    RowGemmArgs a{x, x_ld, x_coff, y, y_ld, y_coff, p.d_w, p.d_scale, p.d_shift, M, p.K, p.J, relu};
    if (M <= 16) hipLaunchKernelGGL(rowgemm_kernel<1>, dim3((unsigned)(p.J / 16)), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL(rowgemm_kernel<2>, dim3((unsigned)(p.J / 16)), dim3(512), 0, stream, a);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "rowgemm: launch failed"; return -2; }
    return 0;
}

int rowconv_launch(const RowGemmPlan& p, const RowConvIO& io, hipStream_t stream, std::string* err) {
    const int taps = io.KW * io.KW;
    if (!p.d_w || io.N <= 0 || taps <= 0 || p.K % taps) { if (err) *err = "rowconv: no plan / bad tap count"; return -1; }
    const int C = p.K / taps;
    if (C % 32 || p.J % 32 || ((p.J / 32) % 8 != 0 && p.J > 128)) { if (err) *err = "rowconv: channels must be a multiple of 32, output channels a multiple of 256 (or <= 128)"; return -1; }
    if (((io.x_ld | io.x_coff | io.y_ld | io.y_coff | io.res_ld | io.res_coff) & 15) || io.x_coff + C > io.x_ld || io.y_coff + p.J > io.y_ld) {
        if (err) *err = "rowconv: channel pitch / offset"; return -1;
    }
    const long long M = (long long)io.N * io.Ho * io.Wo;
    if (M > kRowConvMaxRows) { if (err) *err = "rowconv: more rows than it is built for"; return -1; }
    RowConvArgs a;
    a.x = io.x; a.x_cbt = io.x_ld >> 4; a.x_cb0 = io.x_coff >> 4;
    a.y = io.y; a.y_cbt = io.y_ld >> 4; a.y_cb0 = io.y_coff >> 4;
    a.res = io.res; a.res_cbt = io.res_ld >> 4; a.res_cb0 = io.res_coff >> 4;
    a.w = p.d_w; a.scale = p.d_scale; a.shift = p.d_shift;
    a.N = io.N; a.H = io.H; a.W = io.W; a.Ho = io.Ho; a.Wo = io.Wo; a.S = io.stride; a.Sx = io.stride_w > 0 ? io.stride_w : io.stride; a.pad = io.pad; a.KW = io.KW;
    a.cpt = C / 32; a.cmagic = (unsigned)((0x100000000ull + (unsigned)a.cpt - 1) / (unsigned)a.cpt); a.KT = p.K / 32; a.M = (int)M; a.relu = io.relu;
    a.nph = 1; a.Wout = io.Wo; a.HWout = io.Ho * io.Wo; a.JB = p.J / 32;
    a.ln_out = io.ln_out; a.ln_in = io.ln_in; a.ln_out_tiles = io.ln_out_tiles; a.ln_in_tiles = io.ln_in_tiles; a.ln_eps = io.ln_eps;
    if ((io.ln_out || io.ln_in) && (taps != 1 || io.stride != 1 || (io.ln_out && io.ln_out_tiles != p.J / 32))) {
        if (err) *err = "rowconv: the LayerNorm fold is a 1x1-layer feature"; return -1;
    }
    const int tiles = (int)((M + 15) / 16);
    const int FT = tiles * (p.J / 32) <= 512 ? 2 : 4;          // ~2 blocks per CU's worth of row groups before the tiles grow
    a.NR = (tiles + FT - 1) / FT;
    const unsigned grid = (unsigned)((p.J / 32) * a.NR);       // (J / 32) % 8 == 0: the XCD mapping of the kernel covers it exactly
    if (FT == 2) hipLaunchKernelGGL((rowconv_kernel<2, 6>), dim3(grid), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((rowconv_kernel<4, 4>), dim3(grid), dim3(512), 0, stream, a);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "rowconv: launch failed"; return -2; }
    return 0;
}

// ConvTranspose2d(k3, s2, p1, op1) on a map of <= 8 x 8 pixels as four weight-streaming GEMMs, one per output sub-pixel phase, in ONE
// launch (gridDim.y = phase): phase (py, px) has ny * nx taps, ny = 1 + py, nx = 1 + px (wav2lip_v2.py:63,65: face_decoder_blocks.2.0 /
// 3.0; conv3 ran them as 192..256 merged-phase items behind a split-K finish launch).  `p[g]`, g = py * 2 + px: plan over
// W_eff[j][t * C + c], t = dy * nx + dx.  io.H x io.W is the source map, io.Ho x io.Wo = 2H x 2W the output map.
int rowconvT_launch(const RowGemmPlan* p, const RowConvIO& io, hipStream_t stream, std::string* err) {
    if (!p || !p[0].d_w || io.N <= 0 || io.Ho != 2 * io.H || io.Wo != 2 * io.W) { if (err) *err = "rowconvT: no plan / bad geometry"; return -1; }
    const int C = p[0].K, J = p[0].J;                         // phase 0 has one tap
    if (C % 32 || J % 256) { if (err) *err = "rowconvT: channels must be a multiple of 32, output channels a multiple of 256"; return -1; }
    if (((io.x_ld | io.x_coff | io.y_ld | io.y_coff) & 15) || io.x_coff + C > io.x_ld || io.y_coff + J > io.y_ld) {
        if (err) *err = "rowconvT: channel pitch / offset"; return -1;
    }
    const long long M = (long long)io.N * io.H * io.W;
    if (M > kRowConvMaxRows) { if (err) *err = "rowconvT: more rows than it is built for"; return -1; }
    RowConvArgs a;
    a.x = io.x; a.x_cbt = io.x_ld >> 4; a.x_cb0 = io.x_coff >> 4;
    a.y = io.y; a.y_cbt = io.y_ld >> 4; a.y_cb0 = io.y_coff >> 4;
    a.res = nullptr; a.res_cbt = 0; a.res_cb0 = 0;
    a.w = p[0].d_w; a.scale = p[0].d_scale; a.shift = p[0].d_shift;
    a.N = io.N; a.H = io.H; a.W = io.W; a.Ho = io.H; a.Wo = io.W; a.S = 1; a.Sx = 1; a.pad = 0; a.KW = 1;      // the ROW map is the source map
    a.cpt = C / 32; a.cmagic = (unsigned)((0x100000000ull + (unsigned)a.cpt - 1) / (unsigned)a.cpt); a.KT = C / 32; a.M = (int)M; a.relu = io.relu;
    a.nph = 4; a.Wout = io.Wo; a.HWout = io.Ho * io.Wo; a.JB = J / 32;
    a.ln_out = nullptr; a.ln_in = nullptr; a.ln_out_tiles = 0; a.ln_in_tiles = 0; a.ln_eps = 0.f;
    for (int g = 0; g < 4; ++g) {
        const int py = g >> 1, px = g & 1;
        if (!p[g].d_w || p[g].J != J || p[g].K != (1 + py) * (1 + px) * C) { if (err) *err = "rowconvT: phase plan mismatch"; return -1; }
        a.ph[g].w = p[g].d_w; a.ph[g].scale = p[g].d_scale; a.ph[g].shift = p[g].d_shift;
        a.ph[g].KT = p[g].K / 32; a.ph[g].KW = 1 + px; a.ph[g].oy_add = py; a.ph[g].ox_add = px;
    }
    const int tiles = (int)((M + 15) / 16);
    const int FT = tiles * (J / 32) <= 512 ? 2 : 4;
    a.NR = (tiles + FT - 1) / FT;
    const dim3 grid((unsigned)((J / 32) * a.NR), 4u);
    if (FT == 2) hipLaunchKernelGGL((rowconv_kernel<2, 6>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((rowconv_kernel<4, 4>), grid, dim3(512), 0, stream, a);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "rowconvT: launch failed"; return -2; }
    return 0;
}

}  // namespace ltk
