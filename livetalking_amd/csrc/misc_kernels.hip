// HBM-bound helper kernels of the render hot path (gfx950).  Byte/short
// streaming work: coalesced 8/16-byte accesses, no LDS reuse to exploit except
// in the mel DFT (frame + twiddle table in LDS).
#include "misc_kernels.h"

#include <algorithm>

#include <hip/hip_fp16.h>

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// input pack: wav2lip_avatar.py:119-134
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_faces_kernel(const FacePtrs* __restrict__ faces, f16* __restrict__ x0) {
    const int f = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;  // 0..65535
    const uint8_t* __restrict__ src = faces->p[f] + pix * 3;
    const float k = 1.0f / 255.0f;
    const float b = src[0] * k, g = src[1] * k, r = src[2] * k;
    const bool keep = (pix >> 8) < 128;  // img_masked[:, 128:] = 0  (rows)
    f16x8 o;
    o[0] = (f16)(keep ? b : 0.f); o[1] = (f16)(keep ? g : 0.f); o[2] = (f16)(keep ? r : 0.f);
    o[3] = (f16)b; o[4] = (f16)g; o[5] = (f16)r; o[6] = (f16)0.f; o[7] = (f16)0.f;
    *reinterpret_cast<f16x8*>(x0 + ((size_t)f * 65536 + pix) * 8) = o;
}

void launch_pack_faces(const FacePtrs* faces, int nframes, f16* x0, hipStream_t s) {
    hipLaunchKernelGGL(pack_faces_kernel, dim3(256, nframes), dim3(256), 0, s, faces, x0);
}

__global__ __launch_bounds__(256) void pack_mel_kernel(const MelPtrs* __restrict__ mel, f16* __restrict__ out) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // 0..1279
    if (i >= 1280) return;
    f16x8 o;
#pragma unroll
    for (int j = 1; j < 8; ++j) o[j] = (f16)0.f;
    o[0] = (f16)mel->p[f][i];
    *reinterpret_cast<f16x8*>(out + ((size_t)f * 1280 + i) * 8) = o;
}

void launch_pack_mel(const MelPtrs* mel, int nframes, f16* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_mel_kernel, dim3(5, nframes), dim3(256), 0, s, mel, out);
}

__global__ __launch_bounds__(256) void pack_face6_nchw_kernel(const float* __restrict__ face6, f16* __restrict__ x0) {
    const int f = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const float* src = face6 + (size_t)f * 6 * 65536 + pix;
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = (f16)src[(size_t)c * 65536];
    o[6] = (f16)0.f; o[7] = (f16)0.f;
    *reinterpret_cast<f16x8*>(x0 + ((size_t)f * 65536 + pix) * 8) = o;
}

void launch_pack_face6_nchw(const float* face6, int nframes, f16* x0, hipStream_t s) {
    hipLaunchKernelGGL(pack_face6_nchw_kernel, dim3(256, nframes), dim3(256), 0, s, face6, x0);
}

// ---------------------------------------------------------------------------------------
// output head: nn.Conv2d(32,3,1) + Sigmoid (wav2lip_v2.py:90-91), *255 + uint8 truncation
// (wav2lip_avatar.py:138,145).  One thread = 4 consecutive pixels -> 12 output bytes.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const f16* __restrict__ x, int x_cbt,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    const OutPtrs* __restrict__ outs, float* __restrict__ out_f32) {
    constexpr int hw = 65536;
    __shared__ float sw[3 * 32 + 3];
    __shared__ unsigned obytes[192];          // 256 pixels x 3 bytes
    if (threadIdx.x < 96) sw[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 3) sw[96 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const int f = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;     // one pixel per thread: a wave reads 64 x 32 B contiguous per block
    float acc0 = sw[96], acc1 = sw[97], acc2 = sw[98];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        // channel-blocked [N][cb][H*W][16]: channels 8v..8v+7 of pixel r sit in block v/2, half v&1
        const f16x8 h = *reinterpret_cast<const f16x8*>(x + (((size_t)f * x_cbt + (v >> 1)) * hw + r) * 16 + (v & 1) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (float)h[j];
            acc0 += xv * sw[v * 8 + j];
            acc1 += xv * sw[32 + v * 8 + j];
            acc2 += xv * sw[64 + v * 8 + j];
        }
    }
    const float s0 = 1.f / (1.f + __expf(-acc0));
    const float s1 = 1.f / (1.f + __expf(-acc1));
    const float s2 = 1.f / (1.f + __expf(-acc2));
    if (out_f32) {
        float* o = out_f32 + (size_t)f * 3 * hw + r;
        o[0] = s0; o[(size_t)hw] = s1; o[(size_t)2 * hw] = s2;
    }
    // float32 * 255 then truncation toward zero, as numpy astype(uint8) on [0,255]
    unsigned char* ob = reinterpret_cast<unsigned char*>(obytes) + threadIdx.x * 3;
    ob[0] = (unsigned char)(unsigned)(s0 * 255.f);
    ob[1] = (unsigned char)(unsigned)(s1 * 255.f);
    ob[2] = (unsigned char)(unsigned)(s2 * 255.f);
    __syncthreads();
    uint8_t* const of = outs ? outs->p[f] : nullptr;
    if (of && threadIdx.x < 192)
        reinterpret_cast<unsigned*>(of + (size_t)blockIdx.x * 768)[threadIdx.x] = obytes[threadIdx.x];
}

void launch_head(const f16* x32, int x_ld, int nframes, const float* w3x32, const float* b3,
                 const OutPtrs* out_u8, float* out_f32_nchw, hipStream_t s) {
    hipLaunchKernelGGL(head_kernel, dim3(256, nframes), dim3(256), 0, s, x32, x_ld >> 4, w3x32, b3, out_u8, out_f32_nchw);
}

// ---------------------------------------------------------------------------------------
// per-call pointer tables -> device memory.  One launch carries up to 128 entries of each table as kernel arguments (3 KB: HIP
// guarantees 4 KB of kernel arguments), so a 256-frame pass needs two.
// ---------------------------------------------------------------------------------------
constexpr int kTabChunk = 128;
struct TabChunk {
    const void* p[3][kTabChunk];
};
__global__ __launch_bounds__(kTabChunk) void upload_tables_kernel(const TabChunk c, int mask, int first, int n, DevTables* __restrict__ t) {
    const int i = threadIdx.x;
    if (i >= n) return;
    if (mask & 1) t->faces.p[first + i] = (const uint8_t*)c.p[0][i];
    if (mask & 2) t->mels.p[first + i] = (const float*)c.p[1][i];
    if (mask & 4) t->outs.p[first + i] = (uint8_t*)c.p[2][i];
}

void launch_upload_tables(const FacePtrs* faces, const MelPtrs* mels, const OutPtrs* outs, int nframes, DevTables* d_tab, hipStream_t s) {
    const int mask = (faces ? 1 : 0) | (mels ? 2 : 0) | (outs ? 4 : 0);
    if (!mask) return;
    for (int f0 = 0; f0 < nframes; f0 += kTabChunk) {
        const int n = nframes - f0 < kTabChunk ? nframes - f0 : kTabChunk;
        TabChunk c;
        for (int i = 0; i < n; ++i) {
            c.p[0][i] = faces ? faces->p[f0 + i] : nullptr;
            c.p[1][i] = mels ? mels->p[f0 + i] : nullptr;
            c.p[2][i] = outs ? outs->p[f0 + i] : nullptr;
        }
        for (int i = n; i < kTabChunk; ++i) c.p[0][i] = c.p[1][i] = c.p[2][i] = nullptr;
        hipLaunchKernelGGL(upload_tables_kernel, dim3(1), dim3(kTabChunk), 0, s, c, mask, f0, n, d_tab);
    }
}

// face-encoder skip cache <-> concat buffers (misc_kernels.h FeatGeom): blockIdx.y = frame, 16 bytes per thread and trip
__global__ __launch_bounds__(256) void feat_copy_kernel(const FacePtrs* __restrict__ recs, const FeatGeom g, int dir) {
    const int f = blockIdx.y;
    uint4* const rec = reinterpret_cast<uint4*>(const_cast<uint8_t*>(recs->p[f]));
    const unsigned total = g.off[8];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) k += (i >= g.off[q]) ? 1 : 0;
        uint4* const cat = reinterpret_cast<uint4*>(g.cat[k] + (size_t)f * g.cat_stride[k]) + (i - g.off[k]);
        if (dir == 0) *cat = rec[i];
        else rec[i] = *cat;
    }
}

void launch_feat_copy(const FacePtrs* recs, int nframes, const FeatGeom& g, int dir, hipStream_t s) {
    const unsigned blocks = (g.off[8] + 256u * 4u - 1u) / (256u * 4u);       // four items per thread
    hipLaunchKernelGGL(feat_copy_kernel, dim3(blocks, (unsigned)nframes), dim3(256), 0, s, recs, g, dir);
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* __restrict__ x, int N, int HW, int ld, int coff, int C,
                                                            float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * C * HW;
    if (i >= total) return;
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int n = (int)(i / ((size_t)HW * C));
    const int cc = coff + c;   // channel-blocked [N][ld/16][HW][16]
    out[i] = (float)x[(((size_t)n * (ld >> 4) + (cc >> 4)) * HW + p) * 16 + (cc & 15)];
}

__global__ __launch_bounds__(256) void sat_scan_kernel(const f16* __restrict__ x, int cbt, int cb0, int CB, long long P, long long total,
                                                        int q8, unsigned long long* __restrict__ ctr) {
    unsigned hit = 0, bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int half = (int)(i & 1);
        const long long r = i >> 1;
        const long long p = r % P;
        const long long r2 = r / P;
        const int cb = (int)(r2 % CB), n = (int)(r2 / CB);
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(x) + (((size_t)n * cbt + cb0 + cb) * P + p) * 32 + half * 16);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (q8) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned m = (w[k] >> (8 * b)) & 0x7fu;
                    hit += m == 0x7eu; bad += m == 0x7fu;          // e4m3fn: 0x7e = 448, 0x7f = NaN
                }
            } else {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const unsigned m = (w[k] >> (16 * b)) & 0x7fffu;
                    hit += m == 0x7bffu; bad += m >= 0x7c00u;      // fp16: 0x7bff = 65504, >= 0x7c00 = inf / NaN
                }
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { hit += __shfl_xor(hit, m); bad += __shfl_xor(bad, m); }
    if ((threadIdx.x & 63) == 0) {
        if (hit) atomicAdd(ctr, (unsigned long long)hit);
        if (bad) atomicAdd(ctr + 1, (unsigned long long)bad);
    }
}

void launch_sat_scan(const f16* x, int N, int cbt, int cb0, int CB, long long P, int q8, unsigned long long* ctr, hipStream_t s) {
    const long long total = (long long)N * CB * P * 2;
    if (total <= 0 || !ctr) return;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(sat_scan_kernel, dim3(grid), dim3(256), 0, s, x, cbt, cb0, CB, P, total, q8, ctr);
}

void launch_nhwc_to_nchw_f32(const f16* x, int N, int H, int W, int ld, int coff, int C, float* out, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, N, H * W, ld, coff, C, out);
}

// ---------------------------------------------------------------------------------------
// mel-spectrogram: avatars/wav2lip/audio.py:45-51 with hparams.py:33-73 constants, and the
// (80,16) window gather of avatars/audio_features/mel.py:56-63.
// One workgroup = one STFT column: pre-emphasis + periodic Hann window into LDS, 401-bin
// real DFT in fp64 against an LDS twiddle table (the reference computes in float64), mel
// projection with the float32 Slaney basis, dB + symmetric normalisation.
// ---------------------------------------------------------------------------------------
constexpr int kNfft = 800, kHop = 200, kBins = 401, kMels = 80, kMelStep = 16;

__global__ __launch_bounds__(448) void mel_kernel(const float* __restrict__ pcm, int n_samples,
                                                   const int32_t* __restrict__ win_start, int n_win, int col_min,
                                                   const float* __restrict__ basis, const int32_t* __restrict__ lohi,
                                                   float* __restrict__ out) {
    __shared__ double frame[kNfft];
    __shared__ double tw_c[kNfft];
    __shared__ double tw_s[kNfft];
    __shared__ double mag[kBins + 7];
    __shared__ float melcol[kMels];
    const int tid = threadIdx.x;
    const int col = col_min + blockIdx.x;
    const int base = col * kHop - kNfft / 2;  // librosa center=True: frame t covers [t*hop - n_fft/2, +n_fft)
    for (int n = tid; n < kNfft; n += blockDim.x) {
        const int i = base + n;
        double y = 0.0;
        if (i >= 0 && i < n_samples) {
            // audio.py:20-23 lfilter([1,-0.97],[1]): y[i] = x[i] - 0.97*x[i-1], y[0] = x[0]
            const double xi = (double)pcm[i];
            const double xm = (i > 0) ? (double)pcm[i - 1] : 0.0;
            y = xi - 0.97 * xm;
        }
        double s, c;
        sincospi(2.0 * (double)n / (double)kNfft, &s, &c);
        frame[n] = y * (0.5 - 0.5 * c);  // periodic Hann
        tw_c[n] = c;
        tw_s[n] = s;
    }
    __syncthreads();
    if (tid < kBins) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < kNfft; ++n) {
            const double v = frame[n];
            re += v * tw_c[idx];
            im -= v * tw_s[idx];
            idx += tid;
            if (idx >= kNfft) idx -= kNfft;
        }
        mag[tid] = sqrt(re * re + im * im);
    }
    __syncthreads();
    if (tid < kMels) {
        const int lo = lohi[2 * tid], hi = lohi[2 * tid + 1];
        double acc = 0.0;
        for (int k = lo; k < hi; ++k) acc += (double)basis[tid * kBins + k] * mag[k];
        // audio.py:103-105 / :47 / :110-115
        const double min_level = 1e-5;  // exp(-100/20*ln 10)
        double S = 20.0 * log10(fmax(min_level, acc)) - 20.0;
        double v = 8.0 * ((S + 100.0) / 100.0) - 4.0;
        v = fmin(fmax(v, -4.0), 4.0);
        melcol[tid] = (float)v;
    }
    __syncthreads();
    // scatter this column into every window that contains it
    for (int w = 0; w < n_win; ++w) {
        const int k = col - win_start[w];
        if (k >= 0 && k < kMelStep && tid < kMels) out[((size_t)w * kMels + tid) * kMelStep + k] = melcol[tid];
    }
}

void launch_mel(const float* pcm, int n_samples, const int32_t* win_start, int n_win, int col_min, int n_cols,
                const float* basis, const int32_t* lohi, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mel_kernel, dim3(n_cols), dim3(448), 0, s, pcm, n_samples, win_start, n_win, col_min, basis, lohi, out);
}

// ---------------------------------------------------------------------------------------
// paste-back composite: avatars/wav2lip_avatar.py:141-147.  cv2.resize(INTER_LINEAR) 8-bit
// semantics: float32 half-pixel source coordinates, 11-bit fixed-point weights, horizontal
// then vertical pass with OpenCV's (>>4, >>16, +2, >>2) rounding; exact 2x shrink = 2x2 box.
// One thread = 4 consecutive output bytes of the flat H*W*3 frame.
// ---------------------------------------------------------------------------------------
struct AxisTap { int s0, s1, a0, a1; };

__device__ __forceinline__ AxisTap axis_tap(int d, int dst, int src, bool clamp_f) {
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_f) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    AxisTap t;
    t.a0 = __float2int_rn((1.0f - f) * 2048.0f);
    t.a1 = __float2int_rn(f * 2048.0f);
    t.s0 = min(max(s, 0), src - 1);
    t.s1 = min(max(s + 1, 0), src - 1);
    return t;
}

__device__ __forceinline__ void paste_body(const uint8_t* __restrict__ full, int H, int W,
                                           const uint8_t* __restrict__ pred, int y1, int y2, int x1, int x2,
                                           uint8_t* __restrict__ out) {
    const size_t total = (size_t)H * W * 3;
    const size_t b0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (b0 >= total) return;
    const int dh = y2 - y1, dw = x2 - x1;
    const int rowbytes = W * 3;
    unsigned word = 0;
    const int nb = (int)min((size_t)4, total - b0);
    for (int k = 0; k < nb; ++k) {
        const size_t bi = b0 + k;
        const int y = (int)(bi / rowbytes);
        const int rb = (int)(bi - (size_t)y * rowbytes);
        const int x = rb / 3, c = rb - x * 3;
        unsigned v;
        if (y >= y1 && y < y2 && x >= x1 && x < x2) {
            const int dy = y - y1, dx = x - x1;
            if (dw == 256 && dh == 256) {
                v = pred[(dy * 256 + dx) * 3 + c];
            } else if (dw == 128 && dh == 128) {
                const uint8_t* p = pred + ((2 * dy) * 256 + 2 * dx) * 3 + c;
                v = (p[0] + p[3] + p[768] + p[771] + 2) >> 2;
            } else {
                const AxisTap tx = axis_tap(dx, dw, 256, true);
                const AxisTap ty = axis_tap(dy, dh, 256, false);
                const uint8_t* r0 = pred + (size_t)ty.s0 * 768 + c;
                const uint8_t* r1 = pred + (size_t)ty.s1 * 768 + c;
                const int S0 = r0[tx.s0 * 3] * tx.a0 + r0[tx.s1 * 3] * tx.a1;
                const int S1 = r1[tx.s0 * 3] * tx.a0 + r1[tx.s1 * 3] * tx.a1;
                const int o = (((ty.a0 * (S0 >> 4)) >> 16) + ((ty.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
                v = (unsigned)min(max(o, 0), 255);
            }
        } else {
            v = full[bi];
        }
        word |= v << (8 * k);
    }
    // (frame f of a batched launch starts at out0 + f * H*W*3: dword-aligned only when H*W*3 is a multiple of 4)
    if (nb == 4 && (reinterpret_cast<uintptr_t>(out + b0) & 3) == 0) {
        *reinterpret_cast<unsigned*>(out + b0) = word;
    } else {
        for (int k = 0; k < nb; ++k) out[b0 + k] = (uint8_t)(word >> (8 * k));
    }
}

__global__ __launch_bounds__(256) void paste_kernel(const uint8_t* __restrict__ full, int H, int W,
                                                     const uint8_t* __restrict__ pred, int y1, int y2, int x1, int x2,
                                                     uint8_t* __restrict__ out) {
    paste_body(full, H, W, pred, y1, y2, x1, x2, out);
}

// the composites of up to kPasteBatch frames of ONE inference_batch result in one launch: blockIdx.y = frame
__global__ __launch_bounds__(256) void paste_batch_kernel(const PasteBatch b, int H, int W, const uint8_t* __restrict__ pred0,
                                                           uint8_t* __restrict__ out0, size_t out_stride) {
    const int f = blockIdx.y;
    paste_body(b.full[f], H, W, pred0 + (size_t)f * (256 * 256 * 3), b.y1[f], b.y2[f], b.x1[f], b.x2[f], out0 + (size_t)f * out_stride);
}

// ---------------------------------------------------------------------------------------
// MuseTalk paste-back: avatars/musetalk_avatar.py:154-164 + avatars/musetalk/myutil.py:4-25.
// face_large = body[crop].copy(); face_large[face box] = cv2.resize(pred, box); mask = gray(mask)/255 (the mask
// PNG has B=G=R, so gray == channel); body[crop] = cv2.blendLinear(face_large, body[crop], mask, 1-mask), i.e.
// dst = saturate_cast<uchar>((s1*w1 + s2*w2) / (w1 + w2 + 1e-5f)) in float32 with round-half-even.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned resized_pred(const uint8_t* __restrict__ pred, int dy, int dx, int dh, int dw, int c) {
    if (dw == 256 && dh == 256) return pred[(dy * 256 + dx) * 3 + c];
    if (dw == 128 && dh == 128) {
        const uint8_t* p = pred + ((2 * dy) * 256 + 2 * dx) * 3 + c;
        return (p[0] + p[3] + p[768] + p[771] + 2) >> 2;
    }
    const AxisTap tx = axis_tap(dx, dw, 256, true);
    const AxisTap ty = axis_tap(dy, dh, 256, false);
    const uint8_t* r0 = pred + (size_t)ty.s0 * 768 + c;
    const uint8_t* r1 = pred + (size_t)ty.s1 * 768 + c;
    const int S0 = r0[tx.s0 * 3] * tx.a0 + r0[tx.s1 * 3] * tx.a1;
    const int S1 = r1[tx.s0 * 3] * tx.a0 + r1[tx.s1 * 3] * tx.a1;
    const int o = (((ty.a0 * (S0 >> 4)) >> 16) + ((ty.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
    return (unsigned)min(max(o, 0), 255);
}

__global__ __launch_bounds__(256) void paste_blend_kernel(const uint8_t* __restrict__ full, int H, int W,
                                                           const uint8_t* __restrict__ pred, int x1, int y1, int x2, int y2,
                                                           int xs, int ys, int xe, int ye, const uint8_t* __restrict__ mask,
                                                           uint8_t* __restrict__ out) {
#pragma clang fp contract(off)      // blendLinear is products-then-sum; no fused multiply-add
    const size_t total = (size_t)H * W * 3;
    const size_t b0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (b0 >= total) return;
    const int dh = y2 - y1, dw = x2 - x1, mw = xe - xs;
    const int rowbytes = W * 3;
    unsigned word = 0;
    const int nb = (int)min((size_t)4, total - b0);
    for (int k = 0; k < nb; ++k) {
        const size_t bi = b0 + k;
        const int y = (int)(bi / rowbytes);
        const int rb = (int)(bi - (size_t)y * rowbytes);
        const int x = rb / 3, c = rb - x * 3;
        unsigned v = full[bi];
        if (y >= ys && y < ye && x >= xs && x < xe) {
            unsigned fl = v;
            if (y >= y1 && y < y2 && x >= x1 && x < x2) fl = resized_pred(pred, y - y1, x - x1, dh, dw, c);
            const float w1 = (float)((double)mask[((size_t)(y - ys) * mw + (x - xs)) * 3] / 255.0);
            const float w2 = __fsub_rn(1.0f, w1);
            const float den = __fadd_rn(__fadd_rn(w1, w2), 1e-5f);
            const float num = __fadd_rn(__fmul_rn((float)fl, w1), __fmul_rn((float)v, w2));
            const float q = __fdiv_rn(num, den);
            v = (unsigned)min(max((int)rintf(q), 0), 255);
        }
        word |= v << (8 * k);
    }
    // (frame f of a batched launch starts at out0 + f * H*W*3: dword-aligned only when H*W*3 is a multiple of 4)
    if (nb == 4 && (reinterpret_cast<uintptr_t>(out + b0) & 3) == 0) {
        *reinterpret_cast<unsigned*>(out + b0) = word;
    } else {
        for (int k = 0; k < nb; ++k) out[b0 + k] = (uint8_t)(word >> (8 * k));
    }
}

void launch_paste_blend(const uint8_t* full, int H, int W, const uint8_t* pred256, int x1, int y1, int x2, int y2, int xs, int ys,
                        int xe, int ye, const uint8_t* mask, uint8_t* out, hipStream_t s) {
    const size_t total = (size_t)H * W * 3;
    const size_t words = (total + 3) / 4;
    hipLaunchKernelGGL(paste_blend_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, full, H, W, pred256, x1, y1, x2, y2,
                       xs, ys, xe, ye, mask, out);
}

void launch_paste_batch(const PasteBatch& b, int n, int H, int W, const uint8_t* pred0, uint8_t* out0, size_t out_stride, hipStream_t s) {
    const size_t words = ((size_t)H * W * 3 + 3) / 4;
    hipLaunchKernelGGL(paste_batch_kernel, dim3((unsigned)((words + 255) / 256), (unsigned)n), dim3(256), 0, s, b, H, W, pred0, out0, out_stride);
}

void launch_paste(const uint8_t* full, int H, int W, const uint8_t* pred256, int y1, int y2, int x1, int x2,
                  uint8_t* out, hipStream_t s) {
    const size_t total = (size_t)H * W * 3;
    const size_t words = (total + 3) / 4;
    hipLaunchKernelGGL(paste_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, full, H, W, pred256, y1, y2, x1, x2, out);
}

}  // namespace ltk
