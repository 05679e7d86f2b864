// Non-convolution kernels of the MuseTalk path (gfx950).  See nn_kernels.h.
//
// Reference behaviour restated by these kernels (diffusers is a third-party dependency of the reference,
// requirements.txt:41; call sites avatars/musetalk/models/unet.py:36-46, vae.py:96-108):
//   GroupNorm + SiLU   ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out
//   LayerNorm          BasicTransformerBlock.norm1/2/3
//   attention          Attention / AttnProcessor2_0 (scaled_dot_product_attention)
//   GEGLU              FeedForward.net[0]
// In-tree statements of the same blocks: avatars/musetalk/models/syncnet.py:71-181.
#include "nn_kernels.h"
#include "tune.h"

#include <hip/hip_fp16.h>

#include <algorithm>

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kLog2e = 1.4426950408889634f;

// =============================================================================================== GroupNorm
int gn_segments(int N, int C, int P) {
    const long long blocks = (long long)N * (C / 16);
    // a bandwidth kernel needs thousands of blocks in flight (each thread keeps only a few 16-byte loads outstanding):
    // 512 blocks ran the 128-channel 256^2 maps at 1.6 TB/s
    int segs = 1;
    while (blocks * segs < 4096 && P / (segs * 2) >= 512 && segs < 256) segs *= 2;
    return segs;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ x, int cbt, int cb0, int CB, int P, int segs,
                                                        float* __restrict__ partial) {
    __shared__ float red[4][2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = tid & 1, pl = tid >> 1;
    const int n = blockIdx.x / CB, cb = blockIdx.x - n * CB, seg = blockIdx.y;
    const int seglen = (P + segs - 1) / segs;
    const int p0 = seg * seglen, p1 = min(P, p0 + seglen);
    const f16* base = x + ((size_t)(n * cbt + cb0 + cb) * P) * 16 + half * 8;
    float s[8], q[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { s[c] = 0.f; q[c] = 0.f; }
    int p = p0 + pl;
    for (; p + 384 < p1; p += 512) {          // four independent loads in flight per thread
        const f16x8 v0 = *reinterpret_cast<const f16x8*>(base + (size_t)p * 16);
        const f16x8 v1 = *reinterpret_cast<const f16x8*>(base + (size_t)(p + 128) * 16);
        const f16x8 v2 = *reinterpret_cast<const f16x8*>(base + (size_t)(p + 256) * 16);
        const f16x8 v3 = *reinterpret_cast<const f16x8*>(base + (size_t)(p + 384) * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float f0 = (float)v0[c], f1 = (float)v1[c], f2 = (float)v2[c], f3 = (float)v3[c];
            s[c] += (f0 + f1) + (f2 + f3);
            q[c] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        }
    }
    for (; p < p1; p += 128) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(base + (size_t)p * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float f = (float)v[c]; s[c] += f; q[c] += f * f; }
    }
    // reduce over the 32 lanes of this wave that share `half` (lane bit 0)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int m = 2; m < 64; m <<= 1) { s[c] += __shfl_xor(s[c], m); q[c] += __shfl_xor(q[c], m); }
    }
    if (lane < 2) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { red[wave][0][lane * 8 + c] = s[c]; red[wave][1][lane * 8 + c] = q[c]; }
    }
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, c = tid & 15;
        const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        partial[(((size_t)blockIdx.x * segs + seg) * 2 + which) * 16 + c] = v;
    }
}

void launch_gn_stats(const f16* x, int N, int cbt, int cb0, int C, int P, int segs, float* partial, hipStream_t s) {
    const int CB = C / 16;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(N * CB, segs), dim3(256), 0, s, x, cbt, cb0, CB, P, segs, partial);
}

// v * rcp(1 + exp(-v)): v_rcp_f32 (1 ulp) instead of the correctly rounded division (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup: 11 more
// VALU instructions per element, 38 % of gn_coop_kernel's instruction stream; MuseTalk pass -1.1 .. -1.5 % in-job, profiles/r06_gn_coop_ab.txt)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

template <bool FP8>
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x, int x_cbt, int x_cb0, int CB, int P, int cpg,
                                                        float eps, const float* __restrict__ partial, int segs,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                        f16* __restrict__ y, int y_cbt, int y_cb0, float out_scale) {
    __shared__ float ab[2][16];
    __shared__ float red[2][16][17];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / CB, cb = blockIdx.x - n * CB;
    {   // group statistics of this block's 16 channels: 16 threads per channel share the cpg x segs partial sums
        const int c = cb * 16 + (tid & 15), sl = tid >> 4;
        const int g0 = (c / cpg) * cpg;           // first channel of this channel's group
        const int terms = cpg * segs;
        float S = 0.f, Q = 0.f;
        for (int i = sl; i < terms; i += 16) {
            const int cc = g0 + i / segs, sg = i - (i / segs) * segs;
            const float* pp = partial + ((size_t)(n * CB + (cc >> 4)) * segs + sg) * 32 + (cc & 15);
            S += pp[0]; Q += pp[16];
        }
        red[0][tid & 15][sl] = S;
        red[1][tid & 15][sl] = Q;
    }
    __syncthreads();
    if (tid < 16) {
        const int c = cb * 16 + tid;
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { S += red[0][tid][i]; Q += red[1][tid][i]; }
        const float cnt = (float)cpg * (float)P;
        const float mean = S / cnt;
        const float var = fmaxf(Q / cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        const float a = gamma[c] * rstd;
        ab[0][tid] = a;
        ab[1][tid] = beta[c] - mean * a;
    }
    __syncthreads();
    if constexpr (FP8) {
        // e4m3 output: a thread takes all 16 channels of its pixel (32 bytes in, ONE 16-byte store into its half of the pixel's 32-byte
        // granule; the 8-byte stores of the (pixel, half) mapping below made this pass 15 % slower than the fp16 one, r06 fp8 table)
        float a16[16], b16[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { a16[c] = ab[0][c]; b16[c] = ab[1][c]; }
        const f16* xb2 = x + ((size_t)(n * x_cbt + x_cb0 + cb) * P) * 16;
        unsigned char* yq2 = reinterpret_cast<unsigned char*>(y) + ((size_t)(n * y_cbt + y_cb0 + (cb >> 1)) * P) * 32 + (cb & 1) * 16;
        const int q0 = blockIdx.y * 1024, q1 = min(P, q0 + 1024);
        for (int p = q0 + tid; p < q1; p += 256) {
            const f16x8 v0 = *reinterpret_cast<const f16x8*>(xb2 + (size_t)p * 16), v1 = *reinterpret_cast<const f16x8*>(xb2 + (size_t)p * 16 + 8);
            float f[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float t = (float)(c < 8 ? v0[c] : v1[c - 8]) * a16[c] + b16[c];
                if (silu) t = silu_f(t);
                f[c] = fminf(fmaxf(t * out_scale, -448.f), 448.f);
            }
            int w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * k], f[4 * k + 1], w[k], false);
                w[k] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * k + 2], f[4 * k + 3], w[k], true);
            }
            *reinterpret_cast<int4*>(yq2 + (size_t)p * 32) = make_int4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    const int half = tid & 1;
    float a8[8], b8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { a8[c] = ab[0][half * 8 + c]; b8[c] = ab[1][half * 8 + c]; }
    const f16* xb = x + ((size_t)(n * x_cbt + x_cb0 + cb) * P) * 16 + half * 8;
    f16* yb = y + ((size_t)(n * y_cbt + y_cb0 + cb) * P) * 16 + half * 8;
    // fp8: 16-channel block cb is the (cb & 1) half of the 32-byte pixel granule of 32-channel block cb >> 1
    unsigned char* yq = reinterpret_cast<unsigned char*>(y) + ((size_t)(n * y_cbt + y_cb0 + (cb >> 1)) * P) * 32 + (cb & 1) * 16 + half * 8;
    const int p0 = blockIdx.y * 1024;
    const int p1 = min(P, p0 + 1024);
    auto emit = [&](const f16x8 v, int p) {
        f16x8 o;
        float f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = (float)v[c] * a8[c] + b8[c];
            if (silu) t = silu_f(t);
            o[c] = (f16)t;
            f[c] = fminf(fmaxf(t * out_scale, -448.f), 448.f);
        }
        if (FP8) {
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            *reinterpret_cast<int2*>(yq + (size_t)p * 32) = make_int2(w0, w1);
        } else {
            *reinterpret_cast<f16x8*>(yb + (size_t)p * 16) = o;
        }
    };
    int p = p0 + (tid >> 1);
    if (p0 + 1024 <= P) {                     // whole segment: all 8 loads of a thread in flight before the first store
        f16x8 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f16x8*>(xb + (size_t)(p + i * 128) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) emit(v[i], p + i * 128);
        return;
    }
    for (; p < p1; p += 128) emit(*reinterpret_cast<const f16x8*>(xb + (size_t)p * 16), p);
}

void launch_gn_apply(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* partial,
                     int segs, const float* gamma, const float* beta, int silu, f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    const int CB = C / 16;
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(N * CB, (P + 1023) / 1024), dim3(256), 0, s, x, x_cbt, x_cb0, CB, P, C / groups, eps,
                       partial, segs, gamma, beta, silu, y, y_cbt, y_cb0, 1.f);
}

void launch_gn_apply_fp8(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* partial,
                         int segs, const float* gamma, const float* beta, int silu, float out_scale, unsigned char* y, int y_cbt,
                         int y_cb0, hipStream_t s) {
    const int CB = C / 16;
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(N * CB, (P + 1023) / 1024), dim3(256), 0, s, x, x_cbt, x_cb0, CB, P, C / groups, eps,
                       partial, segs, gamma, beta, silu, reinterpret_cast<f16*>(y), y_cbt, y_cb0, out_scale);
}

// One launch per GroupNorm for the maps whose whole (image, group) fits the registers of one block (the U-Net's 32^2 .. 4^2 maps,
// the VAE's 512-channel 32^2 maps): block = (image, group).  A thread owns ONE channel pair of the group (one dword per pixel:
// groups of C/32 = 10, 20, 30 ... channels start at any even channel of a 16-channel block) and every (256 / L)-th pixel, L = the
// power of two >= pairs per group; it keeps its <= KMAX dwords from the single read pass (fixed base pointer, constant stride),
// the block reduces sum / sum of squares in a fixed order, and the same registers are normalised, activated and stored - one read
// instead of two, one launch instead of two (61 of the 495 launches of a 16-frame U-Net pass were gn_stats).  Statistics are
// summed per thread, per wave, per block: another fp32 order than gn_stats + gn_apply (per channel, then per group), same formula.
template <int KMAX, int NT, bool FP8>
__global__ __launch_bounds__(NT) void gn_group_kernel(const f16* __restrict__ x, int x_cbt, int x_cb0, int P, int cpg, int log2L,
                                                       float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int silu, f16* __restrict__ y, int y_cbt, int y_cb0, float out_scale, int groups) {
    constexpr int NW = NT / 64;
    __shared__ float red[NW][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int j = tid & ((1 << log2L) - 1), slot = tid >> log2L;
    const int ppi = NT >> log2L;                        // pixels per iteration
    const bool live = 2 * j < cpg && slot < P;
    const int c = g * cpg + 2 * (live ? j : 0);
    const unsigned* xp = reinterpret_cast<const unsigned*>(x + ((size_t)(n * x_cbt + x_cb0 + (c >> 4)) * P + (live ? slot : 0)) * 16 + (c & 15));
    const int stride = ppi * 8;                         // dwords between this thread's pixels
    const int iters = live ? (P - slot + ppi - 1) / ppi : 0;
    unsigned v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = (k < iters) ? xp[k * stride] : 0u;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const __half2 h = *reinterpret_cast<const __half2*>(&v[k]);
        const float f0 = __low2float(h), f1 = __high2float(h);
        s += f0 + f1;
        q += f0 * f0 + f1 * f1;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { s += __shfl_xor(s, m); q += __shfl_xor(q, m); }
    if (lane == 0) { red[wave][0] = s; red[wave][1] = q; }
    __syncthreads();
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { S += red[w][0]; Q += red[w][1]; }          // fixed order
    const float cnt = (float)cpg * (float)P;
    const float mean = S / cnt;
    const float rstd = rsqrtf(fmaxf(Q / cnt - mean * mean, 0.f) + eps);
    const float a0 = gamma[c] * rstd, a1 = gamma[c + 1] * rstd;
    const float b0 = beta[c] - mean * a0, b1 = beta[c + 1] - mean * a1;
    const int slot0 = live ? slot : 0;
    unsigned* yp = reinterpret_cast<unsigned*>(y + ((size_t)(n * y_cbt + y_cb0 + (c >> 4)) * P + slot0) * 16 + (c & 15));
    unsigned short* yq = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(y) + ((size_t)(n * y_cbt + y_cb0 + (c >> 5)) * P + slot0) * 32 + (c & 31));
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < iters) {
            const __half2 h = *reinterpret_cast<const __half2*>(&v[k]);
            float t0 = __low2float(h) * a0 + b0;
            float t1 = __high2float(h) * a1 + b1;
            if (silu) { t0 = silu_f(t0); t1 = silu_f(t1); }
            if (FP8) {
                const float f0 = fminf(fmaxf(t0 * out_scale, -448.f), 448.f), f1 = fminf(fmaxf(t1 * out_scale, -448.f), 448.f);
                const int w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
                yq[(size_t)k * ppi * 16] = (unsigned short)(w & 0xffff);
            } else {
                const __half2 o = __floats2half2_rn(t0, t1);
                yp[k * stride] = *reinterpret_cast<const unsigned*>(&o);
            }
        }
    }
}

static int gn_group_log2L(int cpg) { int l = 0; while ((1 << l) < cpg / 2) ++l; return l; }

bool gn_group_fits(int C, int P, int groups) {
    const int cpg = C / groups;
    if (C % groups || (cpg & 1) || cpg > 128) return false;
    return ((long long)P << gn_group_log2L(cpg)) <= 16 * 1024;        // <= 16 dwords per thread of a 1024-thread block
}

void launch_gn_group(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* gamma, const float* beta,
                     int silu, f16* y, int y_cbt, int y_cb0, int fp8, float out_scale, hipStream_t s) {
    const int cpg = C / groups, l2 = gn_group_log2L(cpg);
    const long long work = (long long)P << l2;           // thread slots one (image, group) needs
    const dim3 grid((unsigned)(N * groups));
#define GNG(K, NT, F) hipLaunchKernelGGL((gn_group_kernel<K, NT, F>), grid, dim3(NT), 0, s, x, x_cbt, x_cb0, P, cpg, l2, eps, gamma, beta, silu, y, y_cbt, y_cb0, out_scale, groups)
    // small maps: 256-thread blocks (<= 8 dwords per thread); the 32^2 / 16^2 maps: 1024-thread blocks, so that a thread's chain of
    // loads stays <= 16 deep (256-thread blocks with 32..64 loads per thread measured 17-24 us per launch on the 32^2 level)
    if (fp8) { if (work <= 2048) GNG(8, 256, true); else if (work <= 8192) GNG(8, 1024, true); else GNG(16, 1024, true); }
    else { if (work <= 2048) GNG(8, 256, false); else if (work <= 8192) GNG(8, 1024, false); else GNG(16, 1024, false); }
#undef GNG
}

// ---- GroupNorm of the large maps in ONE tensor pass (round 6): gn_coop_kernel
// The (image, group)s of the VAE's 64^2 .. 256^2 maps (128 KB .. 2 MB per 16-channel block) fit no single block, so gn_stats + gn_apply read the
// tensor twice.  Here a block owns a 2048-pixel slice of one (image, 16-channel block) - 64 KB, kept in REGISTERS from its single read -, the M
// blocks of a (image, channel block) exchange their partial sums through global memory and every block normalises, activates and stores its own
// registers: one read + one write instead of two reads + one write.
//   * exchange without fences: a block publishes its 8 partial sums (sum / sum of squares of the channel quarters 0-3, 4-7, 8-11, 12-15) as 8
//     self-validating words - relaxed agent-scope atomic stores into slots that the pass's prologue filled with the sentinel 0xFFFFFFFF (no fp32
//     sum has that bit pattern) - and polls the 8 M words of its set with relaxed agent-scope atomic loads until none is the sentinel.  No
//     release / acquire fence anywhere (an agent-scope fence writes back / invalidates the XCD's whole L2: what sank the in-kernel split-K
//     reduction of round 3).
//   * forward progress: the members of a set have consecutive block ids; workgroups are dispatched in id order (per XCD), so the members of
//     the lowest unfinished set are resident or done whatever the later blocks wait for.  A poll that does not complete in ~2^16 tries sets
//     *err (host-mapped) and goes on with what it has: loud on the host, never a hang.
//   * fixed summation order (thread, wave tree, waves, members in id order): deterministic, every member derives the same statistics.
// Channels per group 4 / 8 / 16 (whole groups inside a 16-channel block), P a multiple of 2048 with <= 32 slices.
// Measured (profiles/r06_gn_coop_ab.txt, 16 frames): 128 ch @ 256^2 155 -> 132 us (117 us with the exchange switched off: the wait is exposed once per
// round of resident blocks; 512-thread blocks with 16 members per set: 134 us), 256 ch @ 128^2 80 -> 67 us, 512 ch @ 64^2 50 -> 33 us.
int gn_coop_members(int C, int P, int groups) {
    if (groups <= 0 || C % groups) return 0;
    const int cpg = C / groups;
    if (!(cpg == 4 || cpg == 8 || cpg == 16) || C % 16 || P % 2048) return 0;
    const int M = P / 2048;
    return (M >= 2 && M <= 32) ? M : 0;
}

__global__ __launch_bounds__(256) void gn_fill_kernel(unsigned* __restrict__ p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
void launch_gn_coop_reset(unsigned* slots, size_t words, hipStream_t s) {
    if (!words) return;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 1024);
    hipLaunchKernelGGL(gn_fill_kernel, dim3(blocks), dim3(256), 0, s, slots, words, 0xFFFFFFFFu);
}

template <bool FP8>
__global__ __launch_bounds__(256) void gn_coop_kernel(const f16* __restrict__ x, int x_cbt, int x_cb0, int CB, int P, int cpg, int M, float eps,
                                                       unsigned* __restrict__ slots, unsigned* __restrict__ err,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                       f16* __restrict__ y, int y_cbt, int y_cb0, float out_scale) {
    constexpr int NT = 256, NW = NT / 64, SL = NT * 8;  // threads, waves, pixels of a block's slice
    __shared__ float red[NW][8];
    __shared__ float part[32][8];
    __shared__ float ab[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int set = blockIdx.x / M, mem = blockIdx.x - set * M;
    const int n = set / CB, cb = set - n * CB;
    const int p0 = mem * SL + tid;                         // this thread's pixels: p0 + NT k, all 16 channels of each
    const f16* xb = x + ((size_t)(n * x_cbt + x_cb0 + cb) * P + p0) * 16;
    f16x8 v[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k][0] = *reinterpret_cast<const f16x8*>(xb + (size_t)k * NT * 16);
        v[k][1] = *reinterpret_cast<const f16x8*>(xb + (size_t)k * NT * 16 + 8);
    }
    float sq[8];                                           // [0..3] sums of the channel quarters, [4..7] sums of squares
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = 0.f;
    // v_dot2c_f32_f16: sum and sum of squares of a channel PAIR in one instruction each (exact fp16 products, fp32 accumulation) - a third of the
    // convert / add / fma sequence; the kernel's VALU time is of the order of its memory time (r06_gn_coop_ab.txt)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const h2 pr = {v[k][h][2 * c2], v[k][h][2 * c2 + 1]};
                sq[2 * h + (c2 >> 1)] = __builtin_amdgcn_fdot2(pr, ones, sq[2 * h + (c2 >> 1)], false);
                sq[4 + 2 * h + (c2 >> 1)] = __builtin_amdgcn_fdot2(pr, pr, sq[4 + 2 * h + (c2 >> 1)], false);
            }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sq[i] += __shfl_xor(sq[i], m);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][i] = sq[i];
    }
    __syncthreads();
    unsigned* const set_slots = slots + (size_t)set * M * 8;
    if (tid < 8) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][tid];
        __hip_atomic_store(set_slots + mem * 8 + tid, __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 8 * M) {
        unsigned w = __hip_atomic_load(set_slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int tries = 0;
        while (w == 0xFFFFFFFFu && ++tries < (1 << 16)) {
            __builtin_amdgcn_s_sleep(2);
            w = __hip_atomic_load(set_slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (w == 0xFFFFFFFFu) { *err = 1u; w = 0u; }
        part[tid >> 3][tid & 7] = __uint_as_float(w);
    }
    __syncthreads();
    if (tid < 16) {
        const int q0 = (tid / cpg) * cpg / 4, nq = cpg / 4;       // this channel's group = quarters [q0, q0 + nq)
        float S = 0.f, Q = 0.f;
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < nq; ++i) { S += part[m][q0 + i]; Q += part[m][4 + q0 + i]; }
        const float cnt = (float)cpg * (float)P;
        const float mean = S / cnt;
        const float rstd = rsqrtf(fmaxf(Q / cnt - mean * mean, 0.f) + eps);
        const float a = gamma[cb * 16 + tid] * rstd;
        ab[0][tid] = a;
        ab[1][tid] = beta[cb * 16 + tid] - mean * a;
    }
    __syncthreads();
    // (the slice stays PACKED across the exchange: without this the compiler keeps the fp32 copies of the statistics loop alive - 168 registers)
#pragma unroll
    for (int k = 0; k < 8; ++k) { asm volatile("" : "+v"(v[k][0])); asm volatile("" : "+v"(v[k][1])); }
    float a16[16], b16[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { a16[c] = ab[0][c]; b16[c] = ab[1][c]; }
    if constexpr (FP8) {
        unsigned char* yq = reinterpret_cast<unsigned char*>(y) + ((size_t)(n * y_cbt + y_cb0 + (cb >> 1)) * P + p0) * 32 + (cb & 1) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float t = (float)v[k][c >> 3][c & 7] * a16[c] + b16[c];
                if (silu) t = silu_f(t);
                f[c] = fminf(fmaxf(t * out_scale, -448.f), 448.f);
            }
            int w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j], f[4 * j + 1], w[j], false);
                w[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j + 2], f[4 * j + 3], w[j], true);
            }
            *reinterpret_cast<int4*>(yq + (size_t)k * NT * 32) = make_int4(w[0], w[1], w[2], w[3]);
        }
    } else {
        f16* yb = y + ((size_t)(n * y_cbt + y_cb0 + cb) * P + p0) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float t = (float)v[k][h][c] * a16[8 * h + c] + b16[8 * h + c];
                    if (silu) t = silu_f(t);
                    o[c] = (f16)t;
                }
                *reinterpret_cast<f16x8*>(yb + (size_t)k * NT * 16 + 8 * h) = o;
            }
    }
}

void launch_gn_coop(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, unsigned* slots, unsigned* err,
                    const float* gamma, const float* beta, int silu, f16* y, int y_cbt, int y_cb0, int fp8, float out_scale, hipStream_t s) {
    const int CB = C / 16, M = gn_coop_members(C, P, groups);
    const dim3 grid((unsigned)((size_t)N * CB * M));
    if (fp8) hipLaunchKernelGGL(gn_coop_kernel<true>, grid, dim3(256), 0, s, x, x_cbt, x_cb0, CB, P, C / groups, M, eps, slots, err, gamma, beta, silu, y, y_cbt, y_cb0, out_scale);
    else hipLaunchKernelGGL(gn_coop_kernel<false>, grid, dim3(256), 0, s, x, x_cbt, x_cb0, CB, P, C / groups, M, eps, slots, err, gamma, beta, silu, y, y_cbt, y_cb0, 1.f);
}

// =============================================================================================== LayerNorm
// 16 tokens x 16 channel slices per block: the transformer maps are small (1024 / 256 / 64 tokens per image), so the grid
// needs short blocks to cover 256 CUs (64-token blocks left the 32^2 level at one block per CU, 1 TB/s)
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, int cbt, int cb0, int C, int P, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         f16* __restrict__ y, int y_cbt, int y_cb0) {
    __shared__ float red[2][16][17];
    const int tid = threadIdx.x;
    const int tok = tid & 15, part = tid >> 4;
    const int n = blockIdx.y;
    const int p = blockIdx.x * 16 + tok;
    const bool ok = p < P;
    const int items = C >> 3;
    const f16* xb = x + ((size_t)(n * cbt + cb0) * P) * 16;
    float s = 0.f, q = 0.f;
    if (ok) {
        for (int i = part; i < items; i += 16) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xb + ((size_t)(i >> 1) * P + p) * 16 + (i & 1) * 8);
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float f = (float)v[c]; s += f; q += f * f; }
        }
    }
    red[0][tok][part] = s;
    red[1][tok][part] = q;
    __syncthreads();
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S += red[0][tok][i]; Q += red[1][tok][i]; }
    const float mean = S / (float)C;
    const float rstd = rsqrtf(fmaxf(Q / (float)C - mean * mean, 0.f) + eps);
    if (!ok) return;
    f16* yb = y + ((size_t)(n * y_cbt + y_cb0) * P) * 16;
    for (int i = part; i < items; i += 16) {
        const size_t off = ((size_t)(i >> 1) * P + p) * 16 + (i & 1) * 8;
        const f16x8 v = *reinterpret_cast<const f16x8*>(xb + off);
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (f16)(((float)v[c] - mean) * rstd * gamma[i * 8 + c] + beta[i * 8 + c]);
        *reinterpret_cast<f16x8*>(yb + off) = o;
    }
}

void launch_layernorm(const f16* x, int N, int cbt, int cb0, int C, int P, float eps, const float* gamma, const float* beta,
                      f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((P + 15) / 16, N), dim3(256), 0, s, x, cbt, cb0, C, P, eps, gamma, beta, y, y_cbt, y_cb0);
}

// =============================================================================================== attention
int attn_dv32(int d16) { return (d16 + 31) / 32 * 32; }
int attn_tkp(int Tk) { return (Tk + 63) / 64 * 64; }

// V [N][cbt][Tk][16] (head h at blocks cb0 + h*d16/16) -> VT [N][heads][dv32][Tkp], channel-major, the 16 keys of
// every group in MFMA B-operand slot order {0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15}; padding rows / keys are zero.
__device__ __forceinline__ void v_transpose_body(const f16* __restrict__ v, int cbt, int cb0, int heads, int d16, int dv32, int Tk, int Tkp,
                                                 f16* __restrict__ vt, int key0, int h, int n, f16* tile) {
    const int tid = threadIdx.x;
    const int ncb = d16 >> 4;
    const f16* vb = v + ((size_t)(n * cbt + cb0 + h * ncb) * Tk) * 16;
    for (int i = tid; i < ncb * 128; i += 256) {
        const int j = i >> 7, r = i & 127, key = key0 + (r >> 1), half = r & 1;
        f16x8 val;
#pragma unroll
        for (int c = 0; c < 8; ++c) val[c] = (f16)0.f;
        if (key < Tk) val = *reinterpret_cast<const f16x8*>(vb + ((size_t)j * Tk + key) * 16 + half * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) tile[(j * 16 + half * 8 + c) * 66 + (r >> 1)] = val[c];
    }
    __syncthreads();
    f16* out = vt + ((size_t)(n * heads + h) * dv32) * Tkp + key0;
    for (int i = tid; i < dv32 * 8; i += 256) {          // 8 x 16-byte pieces per channel row
        const int ch = i >> 3, piece = i & 7;             // piece: positions piece*8 .. +7 of the 64-key tile
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int pos = piece * 8 + c;
            const int w = pos & 15;
            const int key = (pos & ~15) + ((w < 4) ? w : (w < 8) ? w + 4 : (w < 12) ? w - 4 : w);
            o[c] = (ch < d16) ? tile[ch * 66 + key] : (f16)0.f;
        }
        *reinterpret_cast<f16x8*>(out + (size_t)ch * Tkp + piece * 8) = o;
    }
}

__global__ __launch_bounds__(256) void v_transpose_kernel(const f16* __restrict__ v, int cbt, int cb0, int heads, int d16, int dv32,
                                                           int Tk, int Tkp, f16* __restrict__ vt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_vt[];
    v_transpose_body(v, cbt, cb0, heads, d16, dv32, Tk, Tkp, vt, blockIdx.x * 64, blockIdx.y, blockIdx.z, reinterpret_cast<f16*>(smem_vt));   // tile [d16][66]
}

// the value tensors of several attentions over the SAME keys (the 16 cross-attentions of MuseTalk's U-Net: views of one stacked
// k | v projection of the audio context) in one launch: blockIdx.y walks the heads of all of them
__global__ __launch_bounds__(256) void v_transpose_multi_kernel(const VtMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_vt[];
    int it = 0;
    while (it + 1 < m.n && (int)blockIdx.y >= m.it[it + 1].h0) ++it;
    const VtMulti::Item& a = m.it[it];
    v_transpose_body(a.v, a.cbt, a.cb0, a.heads, a.d16, a.dv32, m.Tk, m.Tkp, a.vt, blockIdx.x * 64, blockIdx.y - a.h0, blockIdx.z,
                     reinterpret_cast<f16*>(smem_vt));
}

void launch_v_transpose_multi(VtMulti m, int N, hipStream_t s) {
    int heads = 0, dmax = 0;
    for (int i = 0; i < m.n; ++i) { m.it[i].h0 = heads; heads += m.it[i].heads; m.it[i].dv32 = attn_dv32(m.it[i].d16); dmax = std::max(dmax, m.it[i].d16); }
    m.Tkp = attn_tkp(m.Tk);
    hipLaunchKernelGGL(v_transpose_multi_kernel, dim3(m.Tkp / 64, heads, N), dim3(256), (size_t)dmax * 66 * sizeof(f16), s, m);
}

void launch_v_transpose(const f16* v, int N, int cbt, int cb0, int heads, int d16, int Tk, f16* vt, hipStream_t s) {
    const int dv32 = attn_dv32(d16), Tkp = attn_tkp(Tk);
    hipLaunchKernelGGL(v_transpose_kernel, dim3(Tkp / 64, heads, N), dim3(256), (size_t)d16 * 66 * sizeof(f16), s, v, cbt, cb0, heads,
                       d16, dv32, Tk, Tkp, vt);
}

struct AttnArgs {
    const f16* q; const f16* k; const f16* vt; f16* o;
    int q_cbt, q_cb0, k_cbt, k_cb0, o_cbt, o_cb0;
    int Tq, Tk, Tkp, heads, dv32;
};

// One wave = 32 queries of one (image, head).  S^T = K Q^T (keys as MFMA rows, queries as columns): a lane then
// holds 16 keys of ONE query, so the softmax statistics are per lane (+ one exchange with the partner lane), and
// the probabilities already sit in B-operand order for O^T += V^T P^T.  SHARE: the 4 waves of a block take the
// same query tile and a quarter of the value channels each (head dim 512 of the VAE mid-block attention).
template <int DT, int DVT, bool SHARE, bool PF = false>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, n = blockIdx.z;
    const int qtile = SHARE ? blockIdx.x : blockIdx.x * 4 + wave;
    const int q0 = qtile * 32;
    if (q0 >= a.Tq) return;
    const int dvt0 = SHARE ? wave * DVT : 0;
    const int Tq = a.Tq, Tk = a.Tk, Tkp = a.Tkp;
    const f16* qb = a.q + ((size_t)(n * a.q_cbt + a.q_cb0 + h * DT) * Tq) * 16 + hh * 8;
    const f16* kb = a.k + ((size_t)(n * a.k_cbt + a.k_cb0 + h * DT) * Tk) * 16 + hh * 8;
    const f16* vtb = a.vt + ((size_t)(n * a.heads + h) * a.dv32 + dvt0 * 32 + l31) * Tkp + hh * 8;
    const int qrow = min(q0 + l31, Tq - 1);

    f16x8 qf[SHARE ? 1 : DT];
    if constexpr (!SHARE) {
#pragma unroll
        for (int j = 0; j < DT; ++j) qf[j] = *reinterpret_cast<const f16x8*>(qb + ((size_t)j * Tq + qrow) * 16);
    }
    f32x16 acc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float m = -1e30f, l = 0.f;

    // Round 6: (a) PF: the K and V^T fragments of key tile t+1 are loaded while tile t is computed (register double buffer; they were
    // loaded right in front of their MFMAs: one exposed L2 round trip per tile and contraction; knob ATTN_PF, head dims <= 80); (b) the key-range mask is a
    // wave-uniform branch taken on a ragged last tile only; (c) the running maximum rarely moves after the first tiles: the
    // accumulators are rescaled only when some lane's maximum did (wave-uniform test), with alpha == 1 exactly otherwise.
    // Values are identical to the straight loop: (a) and (b) change no arithmetic, (c) skips multiplications by exactly 1.0f.
    constexpr int KF = SHARE ? 1 : DT;
    struct Frag { f16x8 k[KF], v[2][DVT]; };
    auto load_tile = [&](int key0, Frag& f) {
        const int krow = min(key0 + l31, Tk - 1);
        if constexpr (!SHARE) {
#pragma unroll
            for (int j = 0; j < DT; ++j) f.k[j] = *reinterpret_cast<const f16x8*>(kb + ((size_t)j * Tk + krow) * 16);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < DVT; ++t) f.v[s2][t] = *reinterpret_cast<const f16x8*>(vtb + (size_t)t * 32 * Tkp + key0 + s2 * 16);
    };
    auto tile = [&](int key0, const Frag& f) {
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        if constexpr (SHARE) {
            const int krow = min(key0 + l31, Tk - 1);
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const f16x8 kj = *reinterpret_cast<const f16x8*>(kb + ((size_t)j * Tk + krow) * 16);
                const f16x8 qj = *reinterpret_cast<const f16x8*>(qb + ((size_t)j * Tq + qrow) * 16);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kj, qj, st, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < DT; ++j) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.k[j], qf[j], st, 0, 0, 0);
        }
        if (key0 + 32 > Tk) {                   // ragged last tile (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 8 * (r >> 2) + 4 * hh + (r & 3);
                if (key >= Tk) st[r] = -1e30f;
            }
        }
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        float ls = 0.f;
        float p[16];
        const float mneg = -m_new * kLog2e;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(st[r], kLog2e, mneg)); ls += p[r]; }      // exp(st - m_new): one fma in front of v_exp_f32 instead of sub + mul
        if (__any(m_new > m)) {
            const float alpha = __expf(m - m_new);
            l = l * alpha;
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
        }
        l += ls;
        m = m_new;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pf[r] = (f16)p[8 * s + r]; pf[4 + r] = (f16)p[8 * s + 4 + r]; }
#pragma unroll
            for (int t = 0; t < DVT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v[s][t], pf, acc[t], 0, 0, 0);
        }
    };
    // two tiles per trip, so that the double buffer is two NAMED register sets (a runtime buffer index would send them to scratch)
    if constexpr (PF) {
        Frag fa, fb;
        load_tile(0, fa);
        for (int key0 = 0; key0 < Tk; key0 += 64) {
            const bool second = key0 + 32 < Tk;
            if (second) load_tile(key0 + 32, fb);
            tile(key0, fa);
            if (second) {
                if (key0 + 64 < Tk) load_tile(key0 + 64, fa);
                tile(key0 + 32, fb);
            }
        }
    } else {
        for (int key0 = 0; key0 < Tk; key0 += 32) {
            Frag f;
            load_tile(key0, f);
            tile(key0, f);
        }
    }
    const float inv = 1.f / (l + __shfl_xor(l, 32));
    const bool qok = (q0 + l31) < Tq;
    f16* ob = a.o + ((size_t)(n * a.o_cbt + a.o_cb0 + h * DT) * Tq + q0 + l31) * 16 + hh * 8;
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int cbl = (dvt0 + t) * 2 + pr;       // channel block of the head
            unsigned pk[2][2];
#pragma unroll
            for (int eo = 0; eo < 2; ++eo) {
                const int g = 2 * pr + eo;
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (f16)(acc[t][4 * g + r] * inv);
                const uint2 u = *reinterpret_cast<const uint2*>(&o);
                pk[eo][0] = u.x; pk[eo][1] = u.y;
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (qok && cbl < DT) *reinterpret_cast<uint4*>(ob + (size_t)cbl * Tq * 16) = out;
        }
    }
}

// Round 6: self-attention with the key / value tiles staged in LDS (the 32^2 and 16^2 levels of the U-Net: 1024 / 256 tokens, head
// dims 40 / 80).  attn_kernel's waves each pull all of K and V^T through their own registers from L2 (4096 waves x 160 KB per 32^2
// attention; 190 registers with the two-tile prefetch: two waves per SIMD); here the four waves of a block - four query tiles of one
// (image, head) - share 64-key tiles that the block copies with global_load_lds into a three-stage ring (K: [channel block][key][2 x 16 B]
// with the halves swapped where bit 3 of the key is set; V^T: [channel row][8 x 16 B] with the chunk index xor (row >> 1) & 7: both
// fragment reads are conflict-free ds_read_b128), one barrier per tile.  Arithmetic and order are attn_kernel's: same values.
#define NN_GLDS16(gptr, lptr)                                                                                \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                  \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

template <int DT, int DVT>
__global__ __launch_bounds__(256) void attn_lds_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_at[];
    constexpr int KBY = DT * 2048, VBY = DVT * 32 * 128, STG = KBY + VBY;      // bytes per stage: K tile, V^T tile
    constexpr int NP = 2 * DT + 4 * DVT;                                       // 1-KiB DMA pieces per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const int Tq = a.Tq, Tk = a.Tk, Tkp = a.Tkp;
    const f16* qb = a.q + ((size_t)(n * a.q_cbt + a.q_cb0 + h * DT) * Tq) * 16 + hh * 8;
    const f16* kb = a.k + ((size_t)(n * a.k_cbt + a.k_cb0 + h * DT) * Tk) * 16;
    const f16* vtb = a.vt + ((size_t)(n * a.heads + h) * a.dv32) * Tkp;
    const int qrow = min(q0 + l31, Tq - 1);

    // every wave issues PPW copies per tile (the last ones wrap round and repeat a piece: same bytes to the same place), so that ONE
    // counted wait - all but the newest tile's copies - holds for the whole block
    constexpr int PPW = (NP + 3) / 4;
    auto stage = [&](int t, int buf) {
        unsigned char* const base = smem_at + buf * STG;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            int pi = wave + 4 * k;                               // wave-uniform
            if (pi >= NP) pi -= 4;
            if (pi < 2 * DT) {
                const int j = pi >> 1, key_l = (pi & 1) * 32 + (lane >> 1);
                const int half = (lane & 1) ^ ((key_l >> 3) & 1);
                NN_GLDS16(kb + ((size_t)j * Tk + t * 64 + key_l) * 16 + half * 8, base + pi * 1024);
            } else {
                const int pv = pi - 2 * DT, row = pv * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                NN_GLDS16(vtb + (size_t)row * Tkp + t * 64 + c * 8, base + KBY + pv * 1024);
            }
        }
    };
    const int ntiles = Tk >> 6;
    stage(0, 0);
    if (ntiles > 1) stage(1, 1);

    f16x8 qf[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) qf[j] = *reinterpret_cast<const f16x8*>(qb + ((size_t)j * Tq + qrow) * 16);
    f32x16 acc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float m = -1e30f, l = 0.f;

    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        // tile t landed (this wave's copies - all but the PPW of tile t+1 -; the barrier then covers the other waves'), and every wave left
        // the stage of tile t-1, which tile t+2 now overwrites.  The waits are explicit: hipcc's own s_waitcnt in front of this barrier
        // covered lgkmcnt only (it had hoisted the vmcnt(0) out of the loop), and a second call with the same inputs gave other frames.
        // (No other vector-memory operation is outstanding inside the loop: the counter counts tile copies only.)
        if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < ntiles) stage(t + 2, buf == 0 ? 2 : buf - 1);
        const unsigned char* const Kb = smem_at + buf * STG;
        const unsigned char* const Vb = Kb + KBY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int key_l = u * 32 + l31;
            const unsigned char* kp = Kb + key_l * 32 + ((hh ^ ((key_l >> 3) & 1)) << 4);
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int j = 0; j < DT; ++j) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(kp + j * 2048), qf[j], st, 0, 0, 0);
            // the value fragments of this sub-tile: requested in front of the softmax arithmetic
            f16x8 vf[2][DVT];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tt = 0; tt < DVT; ++tt) {
                    const int row = tt * 32 + l31;
                    vf[s2][tt] = *reinterpret_cast<const f16x8*>(Vb + row * 128 + (((u * 4 + s2 * 2 + hh) ^ ((row >> 1) & 7)) << 4));
                }
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m, mx);
            float ls = 0.f;
            float p[16];
            const float mneg = -m_new * kLog2e;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(st[r], kLog2e, mneg)); ls += p[r]; }      // exp(st - m_new): one fma in front of v_exp_f32 instead of sub + mul
            if (__any(m_new > m)) {
                const float alpha = __expf(m - m_new);
                l = l * alpha;
#pragma unroll
                for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tt][r] *= alpha;
            }
            l += ls;
            m = m_new;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pf[r] = (f16)p[8 * s2 + r]; pf[4 + r] = (f16)p[8 * s2 + 4 + r]; }
#pragma unroll
                for (int tt = 0; tt < DVT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s2][tt], pf, acc[tt], 0, 0, 0);
            }
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    const float inv = 1.f / (l + __shfl_xor(l, 32));
    const bool qok = (q0 + l31) < Tq;
    f16* ob = a.o + ((size_t)(n * a.o_cbt + a.o_cb0 + h * DT) * Tq + min(q0 + l31, Tq - 1)) * 16 + hh * 8;
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int cbl = t * 2 + pr;
            unsigned pk[2][2];
#pragma unroll
            for (int eo = 0; eo < 2; ++eo) {
                const int g = 2 * pr + eo;
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (f16)(acc[t][4 * g + r] * inv);
                const uint2 u = *reinterpret_cast<const uint2*>(&o);
                pk[eo][0] = u.x; pk[eo][1] = u.y;
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (qok && cbl < DT) *reinterpret_cast<uint4*>(ob + (size_t)cbl * Tq * 16) = out;
        }
    }
}

// Wide single head (VAE mid-block attention: 1 head of 512 channels, 1024 tokens; vae.py:96-108 -> AutoencoderKL mid block).
// The 4 waves of a block take the SAME 32 queries and split BOTH contractions:
//   S^T = K Q^T over the channel dimension: wave w contracts channel blocks [8w, 8w+8) (8 MFMAs per 32-key tile instead of
//         32), the four partial S^T tiles meet in LDS (one 16-byte-strided exchange area per key-tile parity, one barrier
//         per key tile) and every wave then holds the complete tile in the usual register layout;
//   O^T += V^T P^T over the value channels: wave w owns value channels [128w, 128w+128) as before.
// Per key tile a wave issues 16 MFMAs (attn_kernel<32,4,true> issued 40: every wave recomputed the whole S^T), its 8 Q
// fragments stay in registers for the whole kernel (they were re-read from memory for every key tile), and the K / V^T
// fragments of tile t+1 are loaded while tile t is computed.
__global__ __launch_bounds__(256, 2) void attn_wide_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float xch[2][4][4][64][4];        // [parity][wave][register quad][lane][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = blockIdx.x * 32;
    const int Tq = a.Tq, Tk = a.Tk, Tkp = a.Tkp;
    constexpr int DTW = 8, DVT = 4;                      // channel blocks / value tiles per wave (512 channels over 4 waves)
    const int j0 = wave * DTW;
    const f16* qb = a.q + ((size_t)(n * a.q_cbt + a.q_cb0 + h * 32 + j0) * Tq) * 16 + hh * 8;
    const f16* kb = a.k + ((size_t)(n * a.k_cbt + a.k_cb0 + h * 32 + j0) * Tk) * 16 + hh * 8;
    const f16* vtb = a.vt + ((size_t)(n * a.heads + h) * a.dv32 + wave * DVT * 32 + l31) * Tkp + hh * 8;
    const int qrow = min(q0 + l31, Tq - 1);

    f16x8 qf[DTW];
#pragma unroll
    for (int j = 0; j < DTW; ++j) qf[j] = *reinterpret_cast<const f16x8*>(qb + ((size_t)j * Tq + qrow) * 16);
    f32x16 acc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float m = -1e30f, l = 0.f;

    f16x8 kf[DTW];
    auto load_k = [&](int key0) {
        const int krow = min(key0 + l31, Tk - 1);
#pragma unroll
        for (int j = 0; j < DTW; ++j) kf[j] = *reinterpret_cast<const f16x8*>(kb + ((size_t)j * Tk + krow) * 16);
    };
    load_k(0);
    int par = 0;
    for (int key0 = 0; key0 < Tk; key0 += 32, par ^= 1) {
        // this tile's V^T fragments: needed only after the softmax, their latency hides behind the S^T work
        f16x8 vcur[2][DVT];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < DVT; ++t) vcur[s2][t] = *reinterpret_cast<const f16x8*>(vtb + (size_t)t * 32 * Tkp + key0 + s2 * 16);
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int j = 0; j < DTW; ++j) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j], qf[j], st, 0, 0, 0);
        if (key0 + 32 < Tk) load_k(key0 + 32);            // next tile's K fragments fly while this one is reduced and applied
        // partial S^T tiles of the four waves -> LDS -> full tile in every wave
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<f32x4*>(&xch[par][wave][q4][lane][0]) = (f32x4){st[4 * q4], st[4 * q4 + 1], st[4 * q4 + 2], st[4 * q4 + 3]};
        __syncthreads();
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const int ow = (wave + w) & 3;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(&xch[par][ow][q4][lane][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[4 * q4 + r] += o[r];
            }
        }
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 8 * (r >> 2) + 4 * hh + (r & 3);
            if (key >= Tk) st[r] = -1e30f;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx);
        const float alpha = __expf(m - m_new);
        float ls = 0.f;
        float p[16];
        const float mneg = -m_new * kLog2e;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(st[r], kLog2e, mneg)); ls += p[r]; }      // exp(st - m_new): one fma in front of v_exp_f32 instead of sub + mul
        l = l * alpha + ls;
        m = m_new;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pf[r] = (f16)p[8 * s2 + r]; pf[4 + r] = (f16)p[8 * s2 + 4 + r]; }
#pragma unroll
            for (int t = 0; t < DVT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vcur[s2][t], pf, acc[t], 0, 0, 0);
        }
    }
    const float inv = 1.f / (l + __shfl_xor(l, 32));
    const bool qok = (q0 + l31) < Tq;
    f16* ob = a.o + ((size_t)(n * a.o_cbt + a.o_cb0 + h * 32) * Tq + q0 + l31) * 16 + hh * 8;
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int cbl = (wave * DVT + t) * 2 + pr;       // channel block of the head
            unsigned pk[2][2];
#pragma unroll
            for (int eo = 0; eo < 2; ++eo) {
                const int g = 2 * pr + eo;
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (f16)(acc[t][4 * g + r] * inv);
                const uint2 u = *reinterpret_cast<const uint2*>(&o);
                pk[eo][0] = u.x; pk[eo][1] = u.y;
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (qok) *reinterpret_cast<uint4*>(ob + (size_t)cbl * Tq * 16) = out;
        }
    }
}

int launch_attention(const f16* q, int q_cbt, int q_cb0, int Tq, const f16* k, int k_cbt, int k_cb0, int Tk, const f16* vt,
                     f16* o, int o_cbt, int o_cb0, int N, int heads, int d16, hipStream_t s) {
    AttnArgs a;
    a.q = q; a.k = k; a.vt = vt; a.o = o;
    a.q_cbt = q_cbt; a.q_cb0 = q_cb0; a.k_cbt = k_cbt; a.k_cb0 = k_cb0; a.o_cbt = o_cbt; a.o_cb0 = o_cb0;
    a.Tq = Tq; a.Tk = Tk; a.Tkp = attn_tkp(Tk); a.heads = heads; a.dv32 = attn_dv32(d16);
    const int qtiles = (Tq + 31) / 32;
    const dim3 grid4((qtiles + 3) / 4, heads, N), grid1(qtiles, heads, N);
    // self-attention over whole 64-key tiles: K / V^T tiles shared by a block's four query tiles through LDS (knob ATTN_LDS)
    if (knob(K_ATTN_LDS) && Tk % 64 == 0 && Tk >= 128 && (d16 == 48 || d16 == 80)) {
        if (d16 == 48) hipLaunchKernelGGL((attn_lds_kernel<3, 2>), grid4, dim3(256), (size_t)3 * (3 * 2048 + 2 * 32 * 128), s, a);
        else {
            constexpr int lds53 = 3 * (5 * 2048 + 3 * 32 * 128);          // 66 KiB: above the 64-KiB default
            if (ensure_dyn_lds((const void*)attn_lds_kernel<5, 3>, lds53)) return -2;
            hipLaunchKernelGGL((attn_lds_kernel<5, 3>), grid4, dim3(256), (size_t)lds53, s, a);
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    switch (d16) {
        case 48:
            if (knob(K_ATTN_PF)) hipLaunchKernelGGL((attn_kernel<3, 2, false, true>), grid4, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_kernel<3, 2, false>), grid4, dim3(256), 0, s, a);
            break;
        case 64:
            if (knob(K_ATTN_PF)) hipLaunchKernelGGL((attn_kernel<4, 2, false, true>), grid4, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_kernel<4, 2, false>), grid4, dim3(256), 0, s, a);
            break;
        case 80:
            if (knob(K_ATTN_PF)) hipLaunchKernelGGL((attn_kernel<5, 3, false, true>), grid4, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_kernel<5, 3, false>), grid4, dim3(256), 0, s, a);
            break;
        case 160: hipLaunchKernelGGL((attn_kernel<10, 5, false>), grid4, dim3(256), 0, s, a); break;
        case 512:
            if (knob(K_ATTN_WIDE)) hipLaunchKernelGGL(attn_wide_kernel, grid1, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_kernel<32, 4, true>), grid1, dim3(256), 0, s, a);
            break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// =============================================================================================== elementwise
__global__ __launch_bounds__(256) void geglu_kernel(const f16* __restrict__ x, int x_cbt, int x_cb0, int CB, int P,
                                                     f16* __restrict__ y, int y_cbt, int y_cb0, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (n, cb, p, half)
    if (i >= total) return;
    const int half = (int)(i & 1);
    const long long r = i >> 1;
    const int p = (int)(r % P);
    const long long r2 = r / P;
    const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
    const f16x8 av = *reinterpret_cast<const f16x8*>(x + ((size_t)(n * x_cbt + x_cb0 + cb) * P + p) * 16 + half * 8);
    const f16x8 gv = *reinterpret_cast<const f16x8*>(x + ((size_t)(n * x_cbt + x_cb0 + CB + cb) * P + p) * 16 + half * 8);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (f16)((float)av[c] * gelu_as((float)gv[c]));
    *reinterpret_cast<f16x8*>(y + ((size_t)(n * y_cbt + y_cb0 + cb) * P + p) * 16 + half * 8) = o;
}

void launch_geglu(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    const int CB = C / 16;
    const long long total = (long long)N * CB * P * 2;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x_cbt, x_cb0, CB, P, y, y_cbt, y_cb0, total);
}

__global__ __launch_bounds__(256) void act_kernel(const f16* __restrict__ x, int cbt, int cb0, int CB, int P, int act,
                                                   f16* __restrict__ y, int y_cbt, int y_cb0, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int half = (int)(i & 1);
    const long long r = i >> 1;
    const int p = (int)(r % P);
    const long long r2 = r / P;
    const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((size_t)(n * cbt + cb0 + cb) * P + p) * 16 + half * 8);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float f = (float)v[c];
        o[c] = (f16)(act == 2 ? gelu_f(f) : act == 3 ? silu_f(f) : f);
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)(n * y_cbt + y_cb0 + cb) * P + p) * 16 + half * 8) = o;
}

void launch_act(const f16* x, int N, int cbt, int cb0, int C, int P, int act, f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    const int CB = C / 16;
    const long long total = (long long)N * CB * P * 2;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, cbt, cb0, CB, P, act, y, y_cbt, y_cb0, total);
}

__global__ __launch_bounds__(256) void add_pos_kernel(f16* __restrict__ x, int cbt, int cb0, int CB, int P, int C,
                                                       const float* __restrict__ pos, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int half = (int)(i & 1);
    const long long r = i >> 1;
    const int p = (int)(r % P);
    const long long r2 = r / P;
    const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
    f16* ptr = x + ((size_t)(n * cbt + cb0 + cb) * P + p) * 16 + half * 8;
    f16x8 v = *reinterpret_cast<const f16x8*>(ptr);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cb * 16 + half * 8 + c;
        if (ch < C) v[c] = (f16)((float)v[c] + pos[(size_t)p * C + ch]);
    }
    *reinterpret_cast<f16x8*>(ptr) = v;
}

void launch_add_pos(f16* x, int N, int cbt, int cb0, int C, int P, const float* pos, hipStream_t s) {
    const int CB = (C + 15) / 16;
    const long long total = (long long)N * CB * P * 2;
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, cbt, cb0, CB, P, C, pos, total);
}

// =============================================================================================== layout bridges
__global__ __launch_bounds__(256) void nchw_to_cb16_kernel(const float* __restrict__ x, int C, int P, f16* __restrict__ y,
                                                            int y_cbt, int y_cb0, int CB, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (n, cb, p, half)
    if (i >= total) return;
    const int half = (int)(i & 1);
    const long long r = i >> 1;
    const int p = (int)(r % P);
    const long long r2 = r / P;
    const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cb * 16 + half * 8 + c;
        o[c] = (ch < C) ? (f16)x[((size_t)n * C + ch) * P + p] : (f16)0.f;
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)(n * y_cbt + y_cb0 + cb) * P + p) * 16 + half * 8) = o;
}

void launch_nchw_to_cb16(const float* x, int N, int C, int P, f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    const int CB = (C + 15) / 16;
    const long long total = (long long)N * CB * P * 2;
    hipLaunchKernelGGL(nchw_to_cb16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, C, P, y, y_cbt, y_cb0, CB, total);
}

__global__ __launch_bounds__(256) void tokens_to_cb16_kernel(const float* __restrict__ x, int P, int C, const float* __restrict__ add,
                                                              f16* __restrict__ y, int y_cbt, int y_cb0, int CB, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int half = (int)(i & 1);
    const long long r = i >> 1;
    const int p = (int)(r % P);
    const long long r2 = r / P;
    const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cb * 16 + half * 8 + c;
        float v = 0.f;
        if (ch < C) {
            v = x[((size_t)n * P + p) * C + ch];
            if (add) v += add[(size_t)p * C + ch];
        }
        o[c] = (f16)v;
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)(n * y_cbt + y_cb0 + cb) * P + p) * 16 + half * 8) = o;
}

// per-frame token blocks gathered by pointer (cross-session batches: every request has its own feature tensor)
__global__ __launch_bounds__(256) void tokens_gather_to_cb16_kernel(const PtrList64 src, int P, int C, const float* __restrict__ add,
                                                                     f16* __restrict__ y, int y_cbt, int CB) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;       // (cb, p, half)
    if (i >= CB * P * 2) return;
    const int half = i & 1, p = (i >> 1) % P, cb = (i >> 1) / P;
    const float* x = reinterpret_cast<const float*>(src.p[n]);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cb * 16 + half * 8 + c;
        float v = 0.f;
        if (ch < C) {
            v = x[(size_t)p * C + ch];
            if (add) v += add[(size_t)p * C + ch];
        }
        o[c] = (f16)v;
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)(n * y_cbt + cb) * P + p) * 16 + half * 8) = o;
}

void launch_tokens_gather_to_cb16(const PtrList64& src, int N, int P, int C, const float* add, f16* y, int y_cbt, hipStream_t s) {
    const int CB = (C + 15) / 16;
    hipLaunchKernelGGL(tokens_gather_to_cb16_kernel, dim3((CB * P * 2 + 255) / 256, N), dim3(256), 0, s, src, P, C, add, y, y_cbt, CB);
}

void launch_tokens_to_cb16(const float* x, int N, int P, int C, const float* add, f16* y, int y_cbt, int y_cb0, hipStream_t s) {
    const int CB = (C + 15) / 16;
    const long long total = (long long)N * CB * P * 2;
    hipLaunchKernelGGL(tokens_to_cb16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, P, C, add, y, y_cbt, y_cb0, CB, total);
}

__global__ __launch_bounds__(256) void gather_latents_kernel(const PtrList64 src, int C, int P, f16* __restrict__ y, int y_cbt, int CB) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;       // (cb, p, half)
    if (i >= CB * P * 2) return;
    const int half = i & 1, p = (i >> 1) % P, cb = (i >> 1) / P;
    const float* x = reinterpret_cast<const float*>(src.p[f]);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = cb * 16 + half * 8 + c;
        o[c] = (ch < C) ? (f16)x[(size_t)ch * P + p] : (f16)0.f;
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)(f * y_cbt + cb) * P + p) * 16 + half * 8) = o;
}

void launch_gather_latents(const PtrList64& src, int nframes, int C, int P, f16* y, int y_cbt, hipStream_t s) {
    const int CB = (C + 15) / 16;
    hipLaunchKernelGGL(gather_latents_kernel, dim3((CB * P * 2 + 255) / 256, nframes), dim3(256), 0, s, src, C, P, y, y_cbt, CB);
}

__global__ __launch_bounds__(256) void vae_post_kernel(const f16* __restrict__ x, int x_cbt, int P, const OutList64 out,
                                                        float* __restrict__ out_f32) {
    __shared__ unsigned obytes[192];
    const int f = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float rgb[3] = {0.f, 0.f, 0.f};
    if (p < P) {
        const f16x4 v = *reinterpret_cast<const f16x4*>(x + ((size_t)f * x_cbt * P + p) * 16);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (image / 2 + 0.5).clamp(0, 1) in the VAE's fp16, then float * 255 and round-half-even (vae.py:104-106)
            const f16 t = (f16)((float)v[c] * 0.5f + 0.5f);
            rgb[c] = fminf(fmaxf((float)t, 0.f), 1.f);
            if (out_f32) out_f32[((size_t)f * 3 + c) * P + p] = (float)v[c];
        }
    }
    unsigned char* ob = reinterpret_cast<unsigned char*>(obytes) + threadIdx.x * 3;
    ob[0] = (unsigned char)rintf(rgb[2] * 255.f);   // image[..., ::-1]: RGB -> BGR (vae.py:107)
    ob[1] = (unsigned char)rintf(rgb[1] * 255.f);
    ob[2] = (unsigned char)rintf(rgb[0] * 255.f);
    __syncthreads();
    if (out.p[f] && threadIdx.x < 192 && (blockIdx.x * 256 + 255) < P)
        reinterpret_cast<unsigned*>(out.p[f] + (size_t)blockIdx.x * 768)[threadIdx.x] = obytes[threadIdx.x];
}

void launch_vae_post(const f16* x, int x_cbt, int nframes, int P, const OutList64& out, float* out_f32_nchw, hipStream_t s) {
    hipLaunchKernelGGL(vae_post_kernel, dim3((P + 255) / 256, nframes), dim3(256), 0, s, x, x_cbt, P, out, out_f32_nchw);
}

// =============================================================================================== Whisper front end
// transformers WhisperFeatureExtractor (the Audio2Feature.feature_extractor of avatars/musetalk/whisper/
// audio2feature.py:20,107-111; same arithmetic as the vendored avatars/musetalk/whisper/whisper/audio.py:92-127):
// zero-pad to 30 s, STFT n_fft 400 / hop 160 / periodic Hann / centre reflect padding, power spectrum, 80 slaney
// mel bins, log10(max(1e-10,.)), drop the last frame, max(x, global_max - 8), (x + 4) / 4  ->  (80, 3000).
// Frames whose window lies entirely in the zero padding are the constant -10 before the global step.
constexpr int kWNfft = 400, kWHop = 160, kWBins = 201, kWMels = 80, kWFrames = 3000;

__global__ __launch_bounds__(256) void whisper_logmel_kernel(const float* __restrict__ pcm, int n_samples,
                                                              const float* __restrict__ basis, float* __restrict__ logspec,
                                                              int* __restrict__ gmax_bits) {
    __shared__ double frame[kWNfft];
    __shared__ double tw_c[kWNfft];
    __shared__ double tw_s[kWNfft];
    __shared__ double power[kWBins + 7];
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    for (int n = tid; n < kWNfft; n += 256) {
        int i = t * kWHop - kWNfft / 2 + n;
        if (i < 0) i = -i;                                   // reflect (np.pad / torch.stft pad_mode="reflect")
        const double x = (i < n_samples) ? (double)pcm[i] : 0.0;
        double sn, cs;
        sincospi(2.0 * (double)n / (double)kWNfft, &sn, &cs);
        frame[n] = x * (0.5 - 0.5 * cs);
        tw_c[n] = cs;
        tw_s[n] = sn;
    }
    __syncthreads();
    if (tid < kWBins) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < kWNfft; ++n) {
            const double v = frame[n];
            re += v * tw_c[idx];
            im -= v * tw_s[idx];
            idx += tid;
            if (idx >= kWNfft) idx -= kWNfft;
        }
        power[tid] = re * re + im * im;
    }
    __syncthreads();
    if (tid < kWMels) {
        double acc = 0.0;
        for (int k = 0; k < kWBins; ++k) acc += (double)basis[tid * kWBins + k] * power[k];
        const float v = (float)log10(fmax(acc, 1e-10));
        logspec[(size_t)tid * kWFrames + t] = v;
        // float max through an int atomic: v >= -10, so v + 16 is positive and its bit pattern is monotonic
        atomicMax(gmax_bits, __float_as_int(v + 16.0f));
    }
}

__global__ __launch_bounds__(256) void whisper_logmel_finish_kernel(const float* __restrict__ logspec, int n_active,
                                                                     const int* __restrict__ gmax_bits, f16* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // (cb, frame, half)
    if (i >= 5 * kWFrames * 2) return;
    const int half = i & 1, t = (i >> 1) % kWFrames, cb = (i >> 1) / kWFrames;
    const float gmax = fmaxf(__int_as_float(*gmax_bits) - 16.0f, -10.0f);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int m = cb * 16 + half * 8 + c;
        float v = (t < n_active) ? logspec[(size_t)m * kWFrames + t] : -10.0f;
        v = fmaxf(v, gmax - 8.0f);
        o[c] = (f16)((v + 4.0f) / 4.0f);
    }
    *reinterpret_cast<f16x8*>(y + ((size_t)cb * kWFrames + t) * 16 + half * 8) = o;
}

void launch_whisper_logmel(const float* d_pcm, int n_samples, const float* d_basis, float* d_logspec, int* d_gmax, f16* y,
                           hipStream_t s) {
    int n_active = (n_samples + kWNfft / 2 + kWHop - 1) / kWHop + 1;
    if (n_active > kWFrames) n_active = kWFrames;
    (void)hipMemsetAsync(d_gmax, 0, sizeof(int), s);      // bits of +0.0f == "-16" in the shifted domain
    hipLaunchKernelGGL(whisper_logmel_kernel, dim3(n_active), dim3(256), 0, s, d_pcm, n_samples, d_basis, d_logspec, d_gmax);
    hipLaunchKernelGGL(whisper_logmel_finish_kernel, dim3((5 * kWFrames * 2 + 255) / 256), dim3(256), 0, s, d_logspec, n_active, d_gmax, y);
}

// avatars/audio_features/whisper.py:35-56 + base_asr.py:91-133: frame i takes encoder rows
// [first_row + i*row_step, +rows) (clamped), each row = the 5 hidden states -> out fp32 [batch][rows*5][384]
__global__ __launch_bounds__(256) void whisper_chunks_kernel(const WhisperStates st, int T, int batch, int first_row, int row_step,
                                                              int rows, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (frame, row, state, c8)
    const int total = batch * rows * 5 * 48;
    if (i >= total) return;
    const int c8 = i % 48, s5 = (i / 48) % 5, r = (i / 240) % rows, f = i / (240 * rows);
    int row = first_row + f * row_step + r;
    row = min(max(row, 0), T - 1);
    const f16x8 v = *reinterpret_cast<const f16x8*>(st.p[s5] + ((size_t)(st.cb0[s5] + (c8 >> 1)) * T + row) * 16 + (c8 & 1) * 8);
    float* o = out + (((size_t)f * rows + r) * 5 + s5) * 384 + c8 * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (float)v[c];
}

void launch_whisper_chunks(const WhisperStates& st, int T, int batch, int first_row, int row_step, int rows, float* out, hipStream_t s) {
    const int total = batch * rows * 5 * 48;
    hipLaunchKernelGGL(whisper_chunks_kernel, dim3((total + 255) / 256), dim3(256), 0, s, st, T, batch, first_row, row_step, rows, out);
}

// =============================================================================================== VAE encode bridges
// avatars/musetalk/models/vae.py:55-82 preprocess_img on an array input: BGR->RGB, /255., lower-half mask
// (x * (mask > 0.5), mask = 1 on rows < 128), Normalize(0.5, 0.5) -> [-1, 1]; image 2i = masked, 2i+1 = full.
__global__ __launch_bounds__(256) void vae_pre_kernel(const uint8_t* __restrict__ bgr, int nfaces, f16* __restrict__ y) {
    const int img = blockIdx.y;                     // 0 .. 2*nfaces-1
    const int p = blockIdx.x * 256 + threadIdx.x;   // pixel
    const uint8_t* src = bgr + ((size_t)(img >> 1) * 65536 + p) * 3;
    const bool masked = !(img & 1) && (p >> 8) >= 128;
    f16x8 o0, o1;
#pragma unroll
    for (int c = 0; c < 8; ++c) { o0[c] = (f16)0.f; o1[c] = (f16)0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (float)((double)src[2 - c] / 255.0);      // RGB order
        if (masked) v = 0.f;
        o0[c] = (f16)((v - 0.5f) / 0.5f);
    }
    f16* dst = y + ((size_t)img * 65536 + p) * 16;
    *reinterpret_cast<f16x8*>(dst) = o0;
    *reinterpret_cast<f16x8*>(dst + 8) = o1;
}

void launch_vae_pre(const uint8_t* d_bgr, int nfaces, f16* y, hipStream_t s) {
    hipLaunchKernelGGL(vae_pre_kernel, dim3(256, 2 * nfaces), dim3(256), 0, s, d_bgr, nfaces, y);
}

// vae.py:84-94,110-122: latents = scaling_factor * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) per image,
// then cat([masked, reference], dim=1) -> fp32 [nfaces][8][32][32].  noise fp32 [2*nfaces][4][1024] or null (= mean).
__global__ __launch_bounds__(256) void vae_latents_kernel(const f16* __restrict__ moments, int nfaces, const float* __restrict__ noise,
                                                           float scaling, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;       // (image, pixel)
    if (i >= 2 * nfaces * 1024) return;
    const int img = i >> 10, p = i & 1023;
    const f16x8 m = *reinterpret_cast<const f16x8*>(moments + ((size_t)img * 1024 + p) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = (float)m[c];
        if (noise) {
            const float lv = fminf(fmaxf((float)m[4 + c], -30.f), 20.f);
            v += __expf(0.5f * lv) * noise[((size_t)img * 4 + c) * 1024 + p];
        }
        out[(((size_t)(img >> 1) * 8) + (img & 1) * 4 + c) * 1024 + p] = scaling * v;
    }
}

void launch_vae_latents(const f16* moments, int nfaces, const float* d_noise, float scaling, float* out, hipStream_t s) {
    hipLaunchKernelGGL(vae_latents_kernel, dim3((2 * nfaces * 1024 + 255) / 256), dim3(256), 0, s, moments, nfaces, d_noise, scaling, out);
}

}  // namespace ltk
