// conv7: the generator's first layer, Conv2d(6, 16, 7, stride 1, pad 3) + BN + ReLU on the 256x256 face crop
// (avatars/wav2lip/models/wav2lip_v2.py:13, conv.py:5-19), fused with the input pack of LipReal.inference_batch
// (avatars/wav2lip_avatar.py:119-134: bank gather, lower-half mask, 6-channel concat, /255).
//
// Why its own kernel: K = 6 channels x 49 taps.  The generic kernels pad it to 8 x 49 rows of a 32-cout MFMA tile whose
// upper 16 rows are zero and stage 7x7-halo patches through registers; in a pass it ran at 0.2 PFLOP/s (52 us for 16
// frames, 716 us for 256, against 6 / 130 us of HBM time) and the separate pack launch wrote and re-read 1 MB per frame.
// Here:
//   * v_mfma_f32_16x16x32_f16 with the WEIGHTS as the row operand (16 couts = the whole layer) and 16 pixels as columns;
//     k = 4 taps x 8 channels, so a lane's 8-half B fragment is ONE pixel's 8 packed channels: a single 16-byte LDS read
//     at (x + tap) with no shuffling, conflict-free (a 16-lane group reads 16 consecutive 16-byte slots).  A 7-tap kernel
//     row is two MFMAs (taps 0-3, taps 4-6 + a zero tap): 14 MFMAs per 16 pixels, 66 % of them useful work;
//   * the 14 weight fragments (56 VGPRs) stay in registers for the life of the block (persistent over tiles);
//   * BANK mode reads the uint8 bank crop directly (3 B per pixel instead of the 16-B packed item), builds the
//     {masked b,g,r, b,g,r, 0, 0} / 255 item in registers and writes it to LDS: the pack kernel and its 1 MB per frame
//     round trip disappear; PACKED mode (test hooks, the float face6 input) reads the packed fp16 item;
//   * the epilogue stores a lane's 4 consecutive output channels (8 B); the 4 lanes of a pixel cover its 32-byte CB16 cell.
#include <hip/hip_fp16.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "conv_mfma.h"
#include "misc_kernels.h"

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C7_TW = 64, C7_TH = 16;                 // output pixels per tile
constexpr int C7_PW = C7_TW + 6, C7_PH = C7_TH + 6;   // input patch (pad 3 each side)
constexpr int C7_PWP = 72;                            // patch row pitch in pixels (16-byte slots)
constexpr int C7_LDS = C7_PH * C7_PWP * 16;           // 25 344 B

struct C7Args {
    const f16* x;                 // PACKED mode: fp16 [N][256][256][8]; BANK mode: unused
    const f16x8* w;               // [14][64] A fragments: (ky*2 + h) x lane
    const float* scale;           // [16]
    const float* shift;           // [16]
    f16* y;                       // CB16 output buffer
    int N, y_cbt, y_cb0;
    int ntiles;                   // N * 64 tiles
};

template <bool BANK>
__global__ __launch_bounds__(256, 2) void conv7_kernel(const C7Args a, const FacePtrs* __restrict__ faces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;

    f16x8 wf[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) wf[i] = a.w[i * 64 + lane];
    f32x4 sc, sf;                                      // this lane's 4 couts: 4g .. 4g+3
    sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * g);
    sf = *reinterpret_cast<const f32x4*>(a.shift + 4 * g);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n = tile >> 6, t = tile & 63;
        const int ty0 = (t >> 2) * C7_TH, tx0 = (t & 3) * C7_TW;
        if (tile != (int)blockIdx.x) __syncthreads();           // everyone is out of the previous patch
        // ---- stage the (TH+6) x (TW+6) input patch, zero outside the image
        const uint8_t* __restrict__ bank = BANK ? faces->p[n] : nullptr;
        // (all C7_PWP columns: the zero eighth tap of the second MFMA reads up to column 71; 0 x garbage could be NaN)
        for (int i = tid; i < C7_PH * C7_PWP; i += 256) {
            const int py = i / C7_PWP, px = i - py * C7_PWP;
            const int iy = ty0 - 3 + py, ix = px < C7_PW ? tx0 - 3 + px : -1;
            f16x8 v;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (f16)0.f;
            if ((unsigned)iy < 256u && (unsigned)ix < 256u) {
                if constexpr (BANK) {
                    const uint8_t* s = bank + (iy * 256 + ix) * 3;
                    const float k = 1.0f / 255.0f;                 // wav2lip_avatar.py:129 (/255.), same rounding as pack_faces_kernel
                    const float b = s[0] * k, gg = s[1] * k, r = s[2] * k;
                    const bool keep = iy < 128;                     // img_masked[:, 128:] = 0 (rows), wav2lip_avatar.py:127-128
                    v[0] = (f16)(keep ? b : 0.f); v[1] = (f16)(keep ? gg : 0.f); v[2] = (f16)(keep ? r : 0.f);
                    v[3] = (f16)b; v[4] = (f16)gg; v[5] = (f16)r;
                } else {
                    v = *reinterpret_cast<const f16x8*>(a.x + ((size_t)n * 65536 + iy * 256 + ix) * 8);
                }
            }
            *reinterpret_cast<f16x8*>(smem + (py * C7_PWP + px) * 16) = v;
        }
        __syncthreads();
        // ---- wave w: output rows 4w .. 4w+3 of the tile, 4 column groups of 16 pixels each
        f32x4 acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // input row (relative to the patch) py = 4w + r + ky serves output row r through kernel row ky: walk the 10 input
        // rows once, each fragment read feeds every (r, ky) pair with r + ky = row
#pragma unroll
        for (int row = 0; row < 10; ++row) {
            const unsigned char* rowp = smem + ((wave * 4 + row) * C7_PWP + n16 + g) * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 xb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[c] = *reinterpret_cast<const f16x8*>(rowp + (c * 16 + 4 * h) * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ky = row - r;
                    if (ky < 0 || ky > 6) continue;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ky * 2 + h], xb[c], acc[r][c], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: BN + ReLU, lane holds couts 4g..4g+3 of pixel (row 4w+r, column 16c + n16)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = ty0 + wave * 4 + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ox = tx0 + c * 16 + n16;
                f16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (f16)__builtin_amdgcn_fmed3f(acc[r][c][q] * sc[q] + sf[q], 0.f, 65504.f);
                *reinterpret_cast<f16x4*>(a.y + (((size_t)n * a.y_cbt + a.y_cb0) * 65536 + oy * 256 + ox) * 16 + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
struct Conv7Plan {
    f16x8* d_w = nullptr;
    float* d_scale = nullptr;
    float* d_shift = nullptr;
};

int conv7_plan_create(Conv7Plan** out, const float* weight /*[16][6][7][7]*/, const float* scale, const float* shift, std::string* err) {
    std::vector<f16> w((size_t)14 * 64 * 8, (f16)0.f);
    for (int ky = 0; ky < 7; ++ky)
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = lane & 15, kx = 4 * h + (lane >> 4);
                if (kx > 6) continue;
                for (int ci = 0; ci < 6; ++ci)
                    w[(((size_t)(ky * 2 + h)) * 64 + lane) * 8 + ci] = (f16)weight[(((size_t)co * 6 + ci) * 7 + ky) * 7 + kx];
            }
    Conv7Plan* p = new Conv7Plan();
    auto fail = [&](const char* m) { if (err) *err = m; delete p; return -2; };
    if (hipMalloc((void**)&p->d_w, w.size() * sizeof(f16)) != hipSuccess) return fail("conv7: weight allocation failed");
    if (hipMalloc((void**)&p->d_scale, 16 * sizeof(float)) != hipSuccess) return fail("conv7: allocation failed");
    if (hipMalloc((void**)&p->d_shift, 16 * sizeof(float)) != hipSuccess) return fail("conv7: allocation failed");
    if (hipMemcpy(p->d_w, w.data(), w.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_scale, scale, 16 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_shift, shift, 16 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail("conv7: upload failed");
    *out = p;
    return 0;
}

void conv7_plan_destroy(Conv7Plan* p) {
    if (!p) return;
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_scale) (void)hipFree(p->d_scale);
    if (p->d_shift) (void)hipFree(p->d_shift);
    delete p;
}

// faces != nullptr: BANK mode (uint8 crops; `faces` is a DEVICE table); else PACKED mode from x0 (fp16 [N][256][256][8]).
int conv7_launch(const Conv7Plan* p, const FacePtrs* faces, const f16* x0, int N, f16* y, int y_ld, int y_coff, hipStream_t stream,
                 std::string* err) {
    if (!p || N <= 0 || N > kPackMaxFrames || ((y_ld | y_coff) & 15)) { if (err) *err = "conv7: bad arguments"; return -1; }
    C7Args a;
    a.x = x0; a.w = p->d_w; a.scale = p->d_scale; a.shift = p->d_shift; a.y = y;
    a.N = N; a.y_cbt = y_ld >> 4; a.y_cb0 = y_coff >> 4; a.ntiles = N * 64;
    const int grid = std::min(a.ntiles, 512);            // 2 resident blocks per CU (180 VGPRs) walk the tile list
    if (faces) {
        hipLaunchKernelGGL(conv7_kernel<true>, dim3(grid), dim3(256), C7_LDS, stream, a, faces);
    } else {
        hipLaunchKernelGGL(conv7_kernel<false>, dim3(grid), dim3(256), C7_LDS, stream, a, (const FacePtrs*)nullptr);
    }
    if (hipGetLastError() != hipSuccess) { if (err) *err = "conv7: launch failed"; return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------ audio0
// audio_encoder.0: Conv2d(1, 32, 3, stride 1, pad 1) + BN + ReLU on the 80 x 16 mel window (wav2lip_v2.py:42, conv.py:5-19), with the
// mel pack of LipReal.inference_batch fused (avatars/wav2lip_avatar.py:131,134: the float32 windows go to the network as [B][1][80][16]).
//
// Why its own kernel (round 6): ONE input channel x 9 taps.  The generic path was pack_mel_kernel (float32 -> an 8-channel fp16 cell per
// pixel, 7 of them zero) + the first-generation MFMA kernel on a K of 8 x 10 rows of which 9 carry anything: 4.8 + 12.0 us at the head of
// a 16-frame call's dependent chain (the audio encoder heads the critical path under knob PREFETCH) and 10 + 243 us at 256 frames, for
// 5.9 MFLOP per 16 frames.  As VALU work it is 288 fma per pixel: a thread owns one pixel and all 32 output channels, reads its 3 x 3
// neighbourhood from the frame's float32 window (rounded to fp16 exactly as the pack did, so the operands are the MFMA path's operands),
// takes the fp16-rounded weights, the BN scale and shift (352 floats: one coalesced load into LDS, then broadcast ds_read_b128; as
// wave-uniform scalar loads they were a chain of 27 s_load round trips, 10.4 us for the launch in profiles' first timeline) and stores
// the two 32-byte CB16 cells of its pixel.  fp32 products of fp16 operands are exact; the nine-term fp32 sum
// differs from the MFMA's in summation order only.
__global__ __launch_bounds__(256) void audio0_kernel(const MelPtrs* __restrict__ mel, const float* __restrict__ wsf, f16* __restrict__ y,
                                                     const int y_cbt, const int y_cb0) {
    __shared__ __attribute__((aligned(16))) float sw[352];
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;          // pixel of the 80 x 16 map (5 blocks x 256 threads = 1280)
    const float4 wv = reinterpret_cast<const float4*>(wsf)[min((int)threadIdx.x, 87)];      // (written to LDS behind the mel loads' issue)
    // (a global-address-space pointer and unconditional loads at clamped indices: nine loads in flight instead of nine flat-load round trips)
    const float __attribute__((address_space(1)))* const m = (const float __attribute__((address_space(1)))*)mel->p[f];
    const int r = i >> 4, c = i & 15;
    float v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int rr = r + ky - 1, cc = c + kx - 1;
            v[ky * 3 + kx] = m[min(max(rr, 0), 79) * 16 + min(max(cc, 0), 15)];
        }
    if (threadIdx.x < 88) reinterpret_cast<float4*>(sw)[threadIdx.x] = wv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int rr = r + ky - 1, cc = c + kx - 1;
            const bool in = rr >= 0 && rr < 80 && cc >= 0 && cc < 16;
            v[ky * 3 + kx] = in ? (float)(f16)v[ky * 3 + kx] : 0.f;
        }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        f16* const yp = y + (((size_t)f * y_cbt + y_cb0 + cb) * 1280 + i) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f16x8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int co = cb * 16 + h * 8 + q;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(v[t], sw[co * 9 + t], acc);
                o[q] = (f16)__builtin_amdgcn_fmed3f(fmaf(acc, sw[288 + co], sw[320 + co]), 0.f, 65504.f);
            }
            *reinterpret_cast<f16x8*>(yp + h * 8) = o;
        }
    }
}

struct Audio0Plan {
    float* d_wsf = nullptr;        // [32][9] fp16-rounded weights, [32] scale, [32] shift
};

int audio0_plan_create(Audio0Plan** out, const float* weight /*[32][1][3][3]*/, const float* scale, const float* shift, std::string* err) {
    std::vector<float> wsf(352);
    for (int k = 0; k < 288; ++k) wsf[k] = (float)(f16)weight[k];
    for (int co = 0; co < 32; ++co) { wsf[288 + co] = scale[co]; wsf[320 + co] = shift[co]; }
    Audio0Plan* p = new Audio0Plan();
    if (hipMalloc((void**)&p->d_wsf, wsf.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->d_wsf, wsf.data(), wsf.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_wsf) (void)hipFree(p->d_wsf);
        delete p;
        if (err) *err = "audio0: allocation / upload failed";
        return -2;
    }
    *out = p;
    return 0;
}

void audio0_plan_destroy(Audio0Plan* p) {
    if (!p) return;
    if (p->d_wsf) (void)hipFree(p->d_wsf);
    delete p;
}

// `mels` is a DEVICE table (entries [0, N)); y = the layer's output buffer [N][y_ld / 16][80][16][16], written at channels [y_coff, y_coff + 32)
int audio0_launch(const Audio0Plan* p, const MelPtrs* mels, int N, f16* y, int y_ld, int y_coff, hipStream_t stream, std::string* err) {
    if (!p || !mels || !y || N <= 0 || N > kPackMaxFrames || ((y_ld | y_coff) & 15)) { if (err) *err = "audio0: bad arguments"; return -1; }
    hipLaunchKernelGGL(audio0_kernel, dim3(5, N), dim3(256), 0, stream, mels, p->d_wsf, y, y_ld >> 4, y_coff >> 4);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "audio0: launch failed"; return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------ audio3
// audio_encoder.3: Conv2d(32, 64, 3, stride (3, 1), pad 1) + BN + ReLU, 80 x 16 -> 27 x 16 (wav2lip_v2.py:46, conv.py:5-19).
//
// The only stride-(3, 1) layer of the network ran on the first-generation kernel: 64 blocks, each staging a 48-row patch through its
// registers in two chunks - 15.4 us of a 16-frame call's head for 0.25 GFLOP.  Its output rows read DISJOINT input row triples, so a
// patch in LDS shares nothing but the three columns of a tap row.  Here a WAVE owns 32 consecutive output pixels (over all frames) and
// all 64 output channels, and feeds v_mfma_f32_32x32x16_f16 straight from global memory: the pixel operand of tap (ky, kx) and channel
// block cb is ONE 16-byte load per lane (lane & 31 = pixel, lane >> 5 = which 8 of the block's 16 channels; zero outside the map), the
// weight operand one 16-byte load per lane from a host-packed [cout tile][tap][cb][lane] image (1 KB per wave and MFMA, L2 resident):
// 18 + 36 independent loads and 36 MFMAs per wave, no LDS, no barrier; one-wave blocks, 216 of them at 16 frames.  Two variants
// with v_dot2_f32_f16 on the VALU were slower than the MFMA launch they replaced (profiles/r06_audio0_ab.txt).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void audio3_kernel(const f16* __restrict__ x, const int x_cbt, const int x_cb0, const int npix_total,
                                                    const f16x8* __restrict__ wq, const float* __restrict__ ss, f16* __restrict__ y,
                                                    const int y_cbt, const int y_cb0) {
    const int lane = threadIdx.x, l31 = lane & 31, kh = lane >> 5;
    const int Pr = blockIdx.x * 32 + l31;                  // output pixel over all frames (432 per frame)
    const bool live = Pr < npix_total;
    const int P = live ? Pr : npix_total - 1;
    const int f = P / 432, p = P - f * 432;
    const int oy = p >> 4, ox = p & 15;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nt][v] = 0.f;
    const f16x8 zero = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    // all 18 + 36 operand loads in flight before the first MFMA (216 VGPRs; left to itself the compiler pairs every few loads with their
    // MFMAs: nine serial round trips, as slow as the launch this kernel replaces)
    f16x8 bv[18], av[36];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = 3 * oy - 1 + t / 3, ix = ox - 1 + t % 3;      // iy <= 79 always
        const bool in = iy >= 0 && ix >= 0 && ix < 16;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f16x8 xv = *reinterpret_cast<const f16x8*>(x + ((((size_t)f * x_cbt + x_cb0 + cb) * 80 + (in ? iy : 0)) * 16 + (in ? ix : 0)) * 16 + 8 * kh);
            bv[t * 2 + cb] = in ? xv : zero;
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) av[k] = wq[k * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(nt * 9 + t) * 2 + cb], bv[t * 2 + cb], acc[nt], 0, 0, 0);
    // accumulator v of a lane: output channel 32 nt + 8 (v / 4) + 4 kh + (v % 4) of pixel l31
    if (live) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = 32 * nt + 8 * q + 4 * kh;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + co), sf = *reinterpret_cast<const f32x4*>(ss + 64 + co);
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (f16)__builtin_amdgcn_fmed3f(fmaf(acc[nt][4 * q + r], sc[r], sf[r]), 0.f, 65504.f);
                *reinterpret_cast<f16x4*>(y + ((((size_t)f * y_cbt + y_cb0 + (co >> 4)) * 432 + p) * 16 + (co & 15))) = o;
            }
    }
}

struct Audio3Plan {
    f16x8* d_wq = nullptr;         // [2 cout tiles][9 taps][2 channel blocks][64 lanes] x 8 halfs: the A operand of every MFMA
    float* d_ss = nullptr;         // [64] scale, [64] shift
};

int audio3_plan_create(Audio3Plan** out, const float* weight /*[64][32][3][3]*/, const float* scale, const float* shift, std::string* err) {
    std::vector<f16> wq((size_t)2 * 9 * 2 * 64 * 8);
    for (int nt = 0; nt < 2; ++nt)
        for (int t = 0; t < 9; ++t)
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * nt + (lane & 31), ci = 16 * cb + 8 * (lane >> 5) + j;
                        wq[((((size_t)nt * 9 + t) * 2 + cb) * 64 + lane) * 8 + j] = (f16)weight[((size_t)co * 32 + ci) * 9 + t];
                    }
    std::vector<float> ss(128);
    for (int co = 0; co < 64; ++co) { ss[co] = scale[co]; ss[64 + co] = shift[co]; }
    Audio3Plan* p = new Audio3Plan();
    if (hipMalloc((void**)&p->d_wq, wq.size() * sizeof(f16)) != hipSuccess || hipMalloc((void**)&p->d_ss, ss.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->d_wq, wq.data(), wq.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_ss, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_wq) (void)hipFree(p->d_wq);
        if (p->d_ss) (void)hipFree(p->d_ss);
        delete p;
        if (err) *err = "audio3: allocation / upload failed";
        return -2;
    }
    *out = p;
    return 0;
}

void audio3_plan_destroy(Audio3Plan* p) {
    if (!p) return;
    if (p->d_wq) (void)hipFree(p->d_wq);
    if (p->d_ss) (void)hipFree(p->d_ss);
    delete p;
}

// x: [N][x_ld / 16][80][16][16] (channels [x_coff, x_coff + 32)), y: [N][y_ld / 16][27][16][16] (channels [y_coff, y_coff + 64))
int audio3_launch(const Audio3Plan* p, const f16* x, int x_ld, int x_coff, int N, f16* y, int y_ld, int y_coff, hipStream_t stream, std::string* err) {
    if (!p || !x || !y || N <= 0 || N > kPackMaxFrames || ((x_ld | x_coff | y_ld | y_coff) & 15)) { if (err) *err = "audio3: bad arguments"; return -1; }
    const int npix = N * 432;
    hipLaunchKernelGGL(audio3_kernel, dim3((npix + 31) / 32), dim3(64), 0, stream, x, x_ld >> 4, x_coff >> 4, npix, p->d_wq, p->d_ss, y,
                       y_ld >> 4, y_coff >> 4);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "audio3: launch failed"; return -2; }
    return 0;
}

// ------------------------------------------------------------------------------------------ convs2d
// The shallow stride-2 layers of the face encoder, face_encoder_blocks.1.0 (16 -> 32, 256^2 -> 128^2) and 2.0 (32 -> 64, 128^2 -> 64^2)
// (wav2lip_v2.py:15,19: Conv2d(k3, s2, p1) + BN + ReLU), the last Wav2Lip layers on the first-generation kernel: K = 9 x 16 / 9 x 32 is
// one or two 16-channel chunks, so an LDS-staged tile pays a patch copy and two barriers for 9 - 18 MFMAs per wave - 251 + 198 us of a
// 256-frame pass at 0.15 PFLOP/s, and every LDS route off that kernel measured slower (rounds 3, 4, 6).  Same idea as audio3_kernel: no
// LDS.  A wave owns one output row of one frame and one 32-channel tile of the output channels; its 9 x CB weight operands stay in
// registers for the whole row; per 32-pixel tile of the row the pixel operand of (tap, channel block) is one 16-byte load per lane
// (lane & 31 = pixel, lane >> 5 = which 8 of the block's 16 channels; the 2.25x tap redundancy of a stride-2 3x3 window is served by the
// vector L1, which is what bounds the kernel), all 9 x CB loads in flight before the first MFMA; the other waves of the SIMD hide the
// round trip.  Same fp16 operands and fp32 accumulation as the generic route, another summation order.
template <int CB>
__global__ __launch_bounds__(256) void convs2d_kernel(const f16* __restrict__ x, const int x_cbt, const int x_cb0, const int H, const int W,
                                                      const int ntasks, const int NT, const f16x8* __restrict__ wq, const float* __restrict__ ss,
                                                      const int Cout, f16* __restrict__ y, const int y_cbt, const int y_cb0) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);      // (frame, output row, cout tile)
    if (task >= ntasks) return;                                 // (no barriers below)
    const int Ho = H >> 1, Wo = W >> 1;
    const int nt = task % NT, row = task / NT, f = row / Ho, oy = row - f * Ho;
    f16x8 av[9 * CB];
#pragma unroll
    for (int k = 0; k < 9 * CB; ++k) av[k] = wq[(nt * 9 * CB + k) * 64 + lane];
    f32x4 sc[4], sf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = *reinterpret_cast<const f32x4*>(ss + 32 * nt + 8 * q + 4 * kh);
        sf[q] = *reinterpret_cast<const f32x4*>(ss + Cout + 32 * nt + 8 * q + 4 * kh);
    }
    const f16x8 zero = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const size_t HW = (size_t)H * W;
    const f16* const xf = x + ((size_t)f * x_cbt + x_cb0) * HW * 16 + 8 * kh;
    f16* const yf = y + (((size_t)f * y_cbt + y_cb0 + 2 * nt) * Ho + oy) * Wo * 16 + 4 * kh;
    for (int ox0 = 0; ox0 < Wo; ox0 += 32) {
        const int ox = ox0 + l31;
        f16x8 bv[9 * CB];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = 2 * oy - 1 + t / 3, ix = 2 * ox - 1 + t % 3;      // iy <= H - 1, ix <= W - 1 always
            const bool in = iy >= 0 && ix >= 0;
            const f16* const px = xf + ((size_t)(in ? iy : 0) * W + (in ? ix : 0)) * 16;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f16x8 xv = *reinterpret_cast<const f16x8*>(px + cb * HW * 16);
                bv[t * CB + cb] = in ? xv : zero;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int k = 0; k < 9 * CB; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[k], bv[k], acc, 0, 0, 0);
        // accumulator v of a lane: output channel 32 nt + 8 (v / 4) + 4 kh + (v % 4) of pixel ox
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)__builtin_amdgcn_fmed3f(fmaf(acc[4 * q + r], sc[q][r], sf[q][r]), 0.f, 65504.f);
            *reinterpret_cast<f16x4*>(yf + ((size_t)(q >> 1) * Ho * Wo + ox) * 16 + 8 * (q & 1)) = o;
        }
    }
}

struct ConvS2dPlan {
    f16x8* d_wq = nullptr;         // [Cout / 32][9 taps][Cin / 16][64 lanes] x 8 halfs: the weight operand of every MFMA
    float* d_ss = nullptr;         // [Cout] scale, [Cout] shift
    int Cin = 0, Cout = 0;
};

int convs2d_plan_create(ConvS2dPlan** out, const float* weight /*[Cout][Cin][3][3]*/, int Cin, int Cout, const float* scale, const float* shift,
                        std::string* err) {
    if (!(Cin == 16 || Cin == 32) || Cout % 32 || Cout <= 0) { if (err) *err = "convs2d: 16 or 32 input channels, output channels in tiles of 32"; return -1; }
    const int CB = Cin / 16, NT = Cout / 32;
    std::vector<f16> wq((size_t)NT * 9 * CB * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int t = 0; t < 9; ++t)
            for (int cb = 0; cb < CB; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * nt + (lane & 31), ci = 16 * cb + 8 * (lane >> 5) + j;
                        wq[((((size_t)nt * 9 + t) * CB + cb) * 64 + lane) * 8 + j] = (f16)weight[((size_t)co * Cin + ci) * 9 + t];
                    }
    std::vector<float> ss((size_t)2 * Cout);
    for (int co = 0; co < Cout; ++co) { ss[co] = scale[co]; ss[Cout + co] = shift[co]; }
    ConvS2dPlan* p = new ConvS2dPlan();
    p->Cin = Cin; p->Cout = Cout;
    if (hipMalloc((void**)&p->d_wq, wq.size() * sizeof(f16)) != hipSuccess || hipMalloc((void**)&p->d_ss, ss.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->d_wq, wq.data(), wq.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_ss, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_wq) (void)hipFree(p->d_wq);
        if (p->d_ss) (void)hipFree(p->d_ss);
        delete p;
        if (err) *err = "convs2d: allocation / upload failed";
        return -2;
    }
    *out = p;
    return 0;
}

void convs2d_plan_destroy(ConvS2dPlan* p) {
    if (!p) return;
    if (p->d_wq) (void)hipFree(p->d_wq);
    if (p->d_ss) (void)hipFree(p->d_ss);
    delete p;
}

// x: [N][x_ld / 16][H][W][16] (channels [x_coff, x_coff + Cin)), y: [N][y_ld / 16][H / 2][W / 2][16] (channels [y_coff, y_coff + Cout)); H, W even, W / 2 a
// multiple of 32
int convs2d_launch(const ConvS2dPlan* p, const f16* x, int x_ld, int x_coff, int N, int H, int W, f16* y, int y_ld, int y_coff, hipStream_t stream,
                   std::string* err) {
    if (!p || !x || !y || N <= 0 || H <= 0 || (H & 1) || W <= 0 || (W & 63) || ((x_ld | x_coff | y_ld | y_coff) & 15)) { if (err) *err = "convs2d: bad arguments"; return -1; }
    const int NT = p->Cout / 32;
    const long long ntasks = (long long)N * (H / 2) * NT;
    if (ntasks > (1ll << 30)) { if (err) *err = "convs2d: launch too large"; return -1; }
    const dim3 grid((unsigned)((ntasks + 3) / 4));
    if (p->Cin == 16)
        hipLaunchKernelGGL(convs2d_kernel<1>, grid, dim3(256), 0, stream, x, x_ld >> 4, x_coff >> 4, H, W, (int)ntasks, NT, p->d_wq, p->d_ss, p->Cout, y, y_ld >> 4, y_coff >> 4);
    else
        hipLaunchKernelGGL(convs2d_kernel<2>, grid, dim3(256), 0, stream, x, x_ld >> 4, x_coff >> 4, H, W, (int)ntasks, NT, p->d_wq, p->d_ss, p->Cout, y, y_ld >> 4, y_coff >> 4);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "convs2d: launch failed"; return -2; }
    return 0;
}

}  // namespace ltk
