// Frame egress: what the reference does on the CPU between paste_back_frame and the
// encoder, done on the composite while it is still in HBM:
//   * speaking<->silent transition  cv2.addWeighted(prev, 1-alpha, cur, alpha, 0)     avatars/base_avatar.py:419-447
//   * watermark                     cv2.putText(frame, "LiveTalking", ...)            avatars/base_avatar.py:449
//   * BGR24 -> I420                 VideoFrame.from_ndarray(bgr24) + the encoder's swscale reformat to yuv420p
//                                   server/webrtc.py:190-193 (aiortc encodes yuv420p), streamout/rtmp.py:81-83
// HBM-bound byte work: one thread owns a 2-row x 4-pixel patch = 3 dwords per row of BGR in, 1 dword per row of Y
// and one 16-bit store each of U and V out; consecutive lanes own consecutive patches, so every wave instruction
// touches one contiguous span (768 B of BGR / 256 B of Y).  A 1280x720 frame is 2.8 MB in and 1.4 MB out.
#include "misc_kernels.h"

namespace ltk {

// libswscale's BT.601 limited-range integer matrix (RGB2YUV_SHIFT = 15):  RY = round(0.299*219/255 * 2^15) ...,
// rounded to nearest like swscale's generic input stage (white -> 235/128/128; the truncating ff_rgb24toyv12_c
// shortcut of some FFmpeg builds gives 234/127/127, i.e. differs by at most 1 LSB)
__device__ __forceinline__ int yuv_y(int b, int g, int r) { return ((8414 * r + 16519 * g + 3208 * b + 16384) >> 15) + 16; }
__device__ __forceinline__ int yuv_u(int b, int g, int r) { return ((-4865 * r - 9528 * g + 14392 * b + 16384) >> 15) + 128; }
__device__ __forceinline__ int yuv_v(int b, int g, int r) { return ((14392 * r - 12061 * g - 2332 * b + 16384) >> 15) + 128; }

struct EgressArgs {
    const uint8_t* src;      // BGR [H][W][3] composite / bank frame
    const uint8_t* prev;     // BGR frame cached for the other speaking state, or null (no transition)
    float w_prev, w_src;     // addWeighted weights (1-alpha, alpha)
    uint8_t* cache;          // BGR: the blended frame before the watermark (the reference's _last_*_frame copy), or null
    const uint8_t* wm;       // watermark coverage [wm_h][wm_w], non-zero = pixel takes the colour; null = none
    int wm_x, wm_y, wm_w, wm_h;
    int wm_b, wm_g, wm_r;
    uint8_t* out;            // BGR [H][W][3], or I420: Y [H][W], U [H/2][W/2], V [H/2][W/2]
    int H, W;
    int i420;                // 0: BGR24, 1: I420
    int chroma;              // 0: chroma of the top-left pixel of each 2x2 quad (ffmpeg <= 6 C bgr24toyv12), 1: 2x2 mean
};

template <bool VEC>
__device__ __forceinline__ void egress_body(const EgressArgs& a) {
    // OpenCV evaluates a*w1 + b*w2 as two products and a sum; hipcc's default -ffp-contract=fast would fuse one of them
#pragma clang fp contract(off)
    const int pw = (a.W + 3) >> 2;                       // 4-pixel patches per row pair
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int py = t / pw;
    const int x0 = (t - py * pw) * 4;
    const int y0 = py * 2;
    if (y0 >= a.H) return;
    const int nx = min(4, a.W - x0);
    const int ny = min(2, a.H - y0);
    uint8_t px[2][12];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r >= ny) {
#pragma unroll
            for (int k = 0; k < 12; ++k) px[r][k] = 0;
            continue;
        }
        const size_t off = ((size_t)(y0 + r) * a.W + x0) * 3;
        if (VEC) {
            const uint3 s = *reinterpret_cast<const uint3*>(a.src + off);
            *reinterpret_cast<uint3*>(px[r]) = s;
            if (a.prev) {
                const uint3 p = *reinterpret_cast<const uint3*>(a.prev + off);
                const uint8_t* pb = reinterpret_cast<const uint8_t*>(&p);
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    // cv2.addWeighted on 8-bit: float32 src1*alpha + src2*beta (+0), cvRound (half to even), saturate
                    const float v = __fadd_rn(__fmul_rn((float)pb[k], a.w_prev), __fmul_rn((float)px[r][k], a.w_src));
                    px[r][k] = (uint8_t)min(max((int)rintf(v), 0), 255);
                }
            }
            if (a.cache) *reinterpret_cast<uint3*>(a.cache + off) = *reinterpret_cast<const uint3*>(px[r]);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (k < nx * 3) {
                    uint8_t v = a.src[off + k];
                    if (a.prev) {
                        const float f = __fadd_rn(__fmul_rn((float)a.prev[off + k], a.w_prev), __fmul_rn((float)v, a.w_src));
                        v = (uint8_t)min(max((int)rintf(f), 0), 255);
                    }
                    if (a.cache) a.cache[off + k] = v;
                    px[r][k] = v;
                } else {
                    px[r][k] = 0;
                }
            }
        }
        if (a.wm) {
            const int wy = y0 + r - a.wm_y;
            if (wy >= 0 && wy < a.wm_h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int wx = x0 + i - a.wm_x;
                    if (i < nx && wx >= 0 && wx < a.wm_w && a.wm[wy * a.wm_w + wx]) {
                        px[r][3 * i] = (uint8_t)a.wm_b; px[r][3 * i + 1] = (uint8_t)a.wm_g; px[r][3 * i + 2] = (uint8_t)a.wm_r;
                    }
                }
            }
        }
    }
    if (!a.i420) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r >= ny) continue;
            const size_t off = ((size_t)(y0 + r) * a.W + x0) * 3;
            if (VEC) {
                *reinterpret_cast<uint3*>(a.out + off) = *reinterpret_cast<const uint3*>(px[r]);
            } else {
                for (int k = 0; k < nx * 3; ++k) a.out[off + k] = px[r][k];
            }
        }
        return;
    }
    // I420 (H and W even, checked by the host)
    uint8_t* const Y = a.out;
    uint8_t* const U = a.out + (size_t)a.H * a.W;
    uint8_t* const V = U + (size_t)(a.H >> 1) * (a.W >> 1);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        unsigned yw = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) yw |= (unsigned)yuv_y(px[r][3 * i], px[r][3 * i + 1], px[r][3 * i + 2]) << (8 * i);
        uint8_t* yd = Y + (size_t)(y0 + r) * a.W + x0;
        if (VEC) {
            *reinterpret_cast<unsigned*>(yd) = yw;
        } else {
            for (int i = 0; i < nx; ++i) yd[i] = (uint8_t)(yw >> (8 * i));
        }
    }
    unsigned uw = 0, vw = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int b, g, r;
        if (a.chroma) {
            b = (px[0][6 * q] + px[0][6 * q + 3] + px[1][6 * q] + px[1][6 * q + 3] + 2) >> 2;
            g = (px[0][6 * q + 1] + px[0][6 * q + 4] + px[1][6 * q + 1] + px[1][6 * q + 4] + 2) >> 2;
            r = (px[0][6 * q + 2] + px[0][6 * q + 5] + px[1][6 * q + 2] + px[1][6 * q + 5] + 2) >> 2;
        } else {
            b = px[0][6 * q]; g = px[0][6 * q + 1]; r = px[0][6 * q + 2];
        }
        uw |= (unsigned)yuv_u(b, g, r) << (8 * q);
        vw |= (unsigned)yuv_v(b, g, r) << (8 * q);
    }
    const size_t coff = (size_t)py * (a.W >> 1) + (x0 >> 1);
    if (VEC) {
        *reinterpret_cast<unsigned short*>(U + coff) = (unsigned short)uw;
        *reinterpret_cast<unsigned short*>(V + coff) = (unsigned short)vw;
    } else {
        for (int q = 0; q < (nx >> 1); ++q) { U[coff + q] = (uint8_t)(uw >> (8 * q)); V[coff + q] = (uint8_t)(vw >> (8 * q)); }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void egress_kernel(const EgressArgs a) {
    egress_body<VEC>(a);
}

// n frames of one batch in one launch (blockIdx.y = frame): no transition blend, no cache write
template <bool VEC>
__global__ __launch_bounds__(256) void egress_batch_kernel(EgressArgs a, size_t src_stride, size_t out_stride) {
    a.src += (size_t)blockIdx.y * src_stride;
    a.out += (size_t)blockIdx.y * out_stride;
    egress_body<VEC>(a);
}

void launch_egress_batch(const uint8_t* src0, size_t src_stride, int n, const uint8_t* wm, int wm_x, int wm_y, int wm_w, int wm_h, int wm_b,
                         int wm_g, int wm_r, uint8_t* out0, size_t out_stride, int H, int W, int i420, int chroma, hipStream_t s) {
    EgressArgs a{src0, nullptr, 0.f, 1.f, nullptr, wm, wm_x, wm_y, wm_w, wm_h, wm_b, wm_g, wm_r, out0, H, W, i420, chroma};
    const int patches = ((W + 3) >> 2) * ((H + 1) >> 1);
    const dim3 grid((unsigned)((patches + 255) / 256), (unsigned)n);
    const bool vec = (W % 4 == 0) && ((((uintptr_t)src0 | (uintptr_t)out0 | src_stride | out_stride) & 3) == 0) &&
                     (!i420 || (((size_t)H * W) % 4 == 0 && ((size_t)(H >> 1) * (W >> 1)) % 2 == 0));
    if (vec) hipLaunchKernelGGL(egress_batch_kernel<true>, grid, dim3(256), 0, s, a, src_stride, out_stride);
    else hipLaunchKernelGGL(egress_batch_kernel<false>, grid, dim3(256), 0, s, a, src_stride, out_stride);
}

void launch_egress(const uint8_t* src, const uint8_t* prev, float w_prev, float w_src, uint8_t* cache, const uint8_t* wm, int wm_x,
                   int wm_y, int wm_w, int wm_h, int wm_b, int wm_g, int wm_r, uint8_t* out, int H, int W, int i420, int chroma,
                   hipStream_t s) {
    EgressArgs a{src, prev, w_prev, w_src, cache, wm, wm_x, wm_y, wm_w, wm_h, wm_b, wm_g, wm_r, out, H, W, i420, chroma};
    const int patches = ((W + 3) >> 2) * ((H + 1) >> 1);
    const dim3 grid((unsigned)((patches + 255) / 256));
    // dword path: every row starts 4-byte aligned (W % 4 == 0; the buffers come from hipMalloc / 256-B aligned slices)
    const bool vec = (W % 4 == 0) && ((((uintptr_t)src | (uintptr_t)prev | (uintptr_t)cache | (uintptr_t)out) & 3) == 0) &&
                     (!i420 || (((size_t)H * W) % 4 == 0 && ((size_t)(H >> 1) * (W >> 1)) % 2 == 0));
    if (vec) hipLaunchKernelGGL(egress_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(egress_kernel<false>, grid, dim3(256), 0, s, a);
}

}  // namespace ltk
