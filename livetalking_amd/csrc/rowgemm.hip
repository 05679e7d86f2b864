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
    RowGemmArgs a{x, x_ld, x_coff, y, y_ld, y_coff, p.d_w, p.d_scale, p.d_shift, M, p.K, p.J, relu};
    if (M <= 16) hipLaunchKernelGGL(rowgemm_kernel<1>, dim3((unsigned)(p.J / 16)), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL(rowgemm_kernel<2>, dim3((unsigned)(p.J / 16)), dim3(512), 0, stream, a);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "rowgemm: launch failed"; return -2; }
    return 0;
}

}  // namespace ltk
