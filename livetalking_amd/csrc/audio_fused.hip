// audio_mid: audio_encoder.4 .. audio_encoder.8 of the Wav2Lip generator (avatars/wav2lip/models/wav2lip_v2.py:46-51 through
// conv.py:5-19) in ONE launch, one workgroup per frame.
//
//   .4  Conv2d(64, 64, 3, 1, 1) + BN + x + ReLU (residual)            27 x 16
//   .5  the same
//   .6  Conv2d(64, 128, 3, stride 3, pad 1) + BN + ReLU               27 x 16 -> 9 x 6
//   .7  Conv2d(128, 128, 3, 1, 1) + BN + x + ReLU (residual)          9 x 6
//   .8  the same
//
// Why: since round 5 a session's consecutive calls are pipelined across calls (knob PREFETCH) and the audio encoder - 13 launches of
// 6-16 us each for 0.3 % of the MACs - heads the critical path of a call.  These six layers work on maps of 432 and 54 pixels per
// frame: their activations (55 KB and 14 KB per frame) fit a CU's LDS, every frame is independent, so a workgroup walks one frame
// through all six layers without leaving the CU: 6 launches and 5 activation round trips become 1 launch.  Only 16 CUs are busy for a
// 16-frame call; the other 240 belong to the prefetched face encoder that runs beside this.
//
// MFMA roles as everywhere in this library (v_mfma_f32_32x32x16_f16: rows = 32 output channels, columns = 32 pixels, k = one
// 16-channel block per tap); contraction order per output element = conv3's (channel block outer, tap inner), epilogue = conv3's
// (acc * scale + shift, ReLU + fp16 clamp in one v_med3, round to fp16; the identity branch of the residual layers is folded into
// the centre tap of the packed weights by the host, exactly as conv_plan_create's callers do), so the residual layers reproduce the
// unfused path bit for bit and the two strided layers differ from the first-generation kernel by summation order only.
//
// LDS: two regions of 4 channel-block planes x (29 x 18) padded pixels x 32 B = 66 816 B.  A map lives with a one-pixel zero border
// ([y + 1][x + 1]), so no tap ever tests a bound; a plane is the conv3 image [pixel][2 x 16 B] with the halves swapped where bit 3
// of the pixel index is set (conflict-free ds_read_b128 over 16 consecutive pixels).  R1: .3 out, .5 out; R2: .4 out, then (dead
// after .5 has read it) the two 9 x 6 maps of .6 / .7 as 8 planes x (11 x 8) pixels each.
#include "audio_fused.h"

#include <hip/hip_fp16.h>

#include <vector>

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HA = 27, WA = 16, WPA = WA + 2, PPA = (HA + 2) * WPA;      // level A: 27 x 16, padded 29 x 18 = 522 pixels
constexpr int HB = 9, WB = 6, WPB = WB + 2, PPB = (HB + 2) * WPB;        // level B: 9 x 6, padded 11 x 8 = 88 pixels
constexpr int PLANE_A = PPA * 32, PLANE_B = PPB * 32;                    // bytes per channel-block plane
constexpr int REGION = 4 * PLANE_A;                                      // 66 816 B
constexpr int RB_BYTES = 8 * PLANE_B;                                    // 22 528 B per level-B map
static_assert(2 * RB_BYTES <= REGION, "the two 9 x 6 maps live in R2");

// byte address of (padded pixel p, half h) inside a plane
__device__ __forceinline__ int pix_addr(int p, int h) { return p * 32 + ((h ^ ((p >> 3) & 1)) << 4); }

// One layer for this wave: NG pixel groups x ONE 32-cout tile.
//   SRC 1: input = level-A LDS region, stride 1 (layers .4 .5)
//   SRC 2: input = level-A LDS region, stride 3 -> level-B output (layer .6)
//   SRC 3: input = level-B LDS region, stride 1 (layers .7 .8)
//   DST 0: level-A LDS region; 1: level-B LDS region; 2: global CB16 [8][54][16]
// The weight fragments of channel block kb + 1 are in flight while block kb is contracted (they come from L2: ~2 us away).
template <int SRC, int DST, int KB, int NG>
__device__ __forceinline__ void layer(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, f16* __restrict__ gy,
                                      const f16* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                      int ct, int g0, int gstep, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
    constexpr bool OUT_A = (SRC == 1);                       // output pixel grid: level A (432 px) or level B (54 px)
    constexpr int P = OUT_A ? HA * WA : HB * WB;
    constexpr int WO = OUT_A ? WA : WB;
    int oy[NG], ox[NG];
    bool live[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int m = (g0 + j * gstep) * 32 + l31;
        live[j] = m < P;
        const int mm = live[j] ? m : 0;
        oy[j] = mm / WO; ox[j] = mm - oy[j] * WO;
    }
    f32x16 acc[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // weights: [ct][kb][tap][64 lanes][8 halfs]
    const f16x8* wp = reinterpret_cast<const f16x8*>(w) + ((size_t)ct * KB * 9) * 64 + lane;
    f16x8 wf[2][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[0][t] = wp[(size_t)t * 64];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < KB) {
#pragma unroll
            for (int t = 0; t < 9; ++t) wf[cur ^ 1][t] = wp[(size_t)((kb + 1) * 9 + t) * 64];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            f16x8 xb[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                if constexpr (SRC == 1) {
                    const int p = (oy[j] + ky) * WPA + ox[j] + kx;
                    xb[j] = *reinterpret_cast<const f16x8*>(src + kb * PLANE_A + pix_addr(p, hh));
                } else if constexpr (SRC == 2) {
                    const int p = (oy[j] * 3 + ky) * WPA + ox[j] * 3 + kx;
                    xb[j] = *reinterpret_cast<const f16x8*>(src + kb * PLANE_A + pix_addr(p, hh));
                } else {
                    const int p = (oy[j] + ky) * WPB + ox[j] + kx;
                    xb[j] = *reinterpret_cast<const f16x8*>(src + kb * PLANE_B + pix_addr(p, hh));
                }
            }
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][t], xb[j], acc[j], 0, 0, 0);
        }
    }
    // epilogue: lane (l31, hh) holds channels ct*32 + 8*q4 + 4*hh .. +3 of its pixel in registers 4*q4 .. 4*q4+3
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = ct * 32 + 8 * q4 + 4 * hh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c0), sf = *reinterpret_cast<const f32x4*>(shift + c0);
        const int kbo = c0 >> 4, half = (c0 >> 3) & 1;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (!live[j]) continue;
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)__builtin_amdgcn_fmed3f(acc[j][4 * q4 + r] * sc[r] + sf[r], 0.f, 65504.f);
            if constexpr (DST == 0) {
                const int p = (oy[j] + 1) * WPA + ox[j] + 1;
                *reinterpret_cast<f16x4*>(dst + kbo * PLANE_A + pix_addr(p, half) + 8 * hh) = o;
            } else if constexpr (DST == 1) {
                const int p = (oy[j] + 1) * WPB + ox[j] + 1;
                *reinterpret_cast<f16x4*>(dst + kbo * PLANE_B + pix_addr(p, half) + 8 * hh) = o;
            } else {
                const int m = oy[j] * WB + ox[j];
                *reinterpret_cast<f16x4*>(gy + ((size_t)(kbo * HB * WB + m)) * 16 + half * 8 + 4 * hh) = o;
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void audio_mid_kernel(const AudioMidArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const R1 = smem;
    unsigned char* const R2 = smem + REGION;
    unsigned char* const B1 = R2;                 // level-B maps inside R2 (after layer .5)
    unsigned char* const B2 = R2 + RB_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x;
    const f16* const gx = a.x + (size_t)f * a.x_stride;
    f16* const gy = a.y + (size_t)f * a.y_stride;
    // zero both regions once: the one-pixel borders are never written afterwards
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid * 16; i < 2 * REGION; i += 512 * 16) *reinterpret_cast<uint4*>(smem + i) = z;
    }
    __syncthreads();
    // audio_encoder.3's output (CB16 [4][432][16]) -> R1: item i = (channel block, pixel, half)
    for (int i = tid; i < 4 * HA * WA * 2; i += 512) {
        const int half = i & 1, px = (i >> 1) % (HA * WA), kb = (i >> 1) / (HA * WA);
        const int y = px / WA, x = px - y * WA;
        const uint4 v = *reinterpret_cast<const uint4*>(gx + ((size_t)(kb * HA * WA + px)) * 16 + half * 8);
        *reinterpret_cast<uint4*>(R1 + kb * PLANE_A + pix_addr((y + 1) * WPA + x + 1, half)) = v;
    }
    __syncthreads();
    // level A: 14 pixel groups x 2 cout tiles over 8 waves -> wave w: cout tile w & 1, groups (w >> 1), +4, +8, +12 (the last one of
    // waves 4..7 is past the map: those lanes compute on pixel 0 and do not store)
    const int ctA = wave & 1, gA = wave >> 1;
    layer<1, 0, 4, 4>(R1, R2, nullptr, a.w[1], a.scale[1], a.shift[1], ctA, gA, 4, lane);     // .4: R1 -> R2
    __syncthreads();
    layer<1, 0, 4, 4>(R2, R1, nullptr, a.w[2], a.scale[2], a.shift[2], ctA, gA, 4, lane);     // .5: R2 -> R1
    __syncthreads();
    {   // R2 is dead: it becomes the two level-B maps, borders zero
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid * 16; i < 2 * RB_BYTES; i += 512 * 16) *reinterpret_cast<uint4*>(R2 + i) = z;
    }
    __syncthreads();
    // level B: 2 pixel groups x 4 cout tiles over 8 waves -> wave w: cout tile w >> 1, group w & 1
    layer<2, 1, 4, 1>(R1, B1, nullptr, a.w[3], a.scale[3], a.shift[3], wave >> 1, wave & 1, 1, lane);     // .6: R1 -> B1
    __syncthreads();
    layer<3, 1, 8, 1>(B1, B2, nullptr, a.w[4], a.scale[4], a.shift[4], wave >> 1, wave & 1, 1, lane);     // .7: B1 -> B2
    __syncthreads();
    layer<3, 2, 8, 1>(B2, nullptr, gy, a.w[5], a.scale[5], a.shift[5], wave >> 1, wave & 1, 1, lane);     // .8: B2 -> global
}

}  // namespace

int audio_mid_pack(AudioMidPlan* p, const float* const weight[6], const float* const scale[6], const float* const shift[6], std::string* err) {
    static const int cin[6] = {32, 64, 64, 64, 128, 128}, cout[6] = {64, 64, 64, 128, 128, 128};
    for (int l = 0; l < 6; ++l) {
        const int KB = cin[l] / 16, CT = cout[l] / 32;
        std::vector<f16> packed((size_t)CT * KB * 9 * 64 * 8);
        for (int ct = 0; ct < CT; ++ct)
            for (int kb = 0; kb < KB; ++kb)
                for (int t = 0; t < 9; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = ct * 32 + (lane & 31), hh = lane >> 5;
                        for (int i = 0; i < 8; ++i) {
                            const int ci = kb * 16 + hh * 8 + i;
                            packed[((((size_t)ct * KB + kb) * 9 + t) * 64 + lane) * 8 + i] = (f16)weight[l][((size_t)co * cin[l] + ci) * 9 + t];
                        }
                    }
        if (hipMalloc((void**)&p->d_w[l], packed.size() * sizeof(f16)) != hipSuccess ||
            hipMalloc((void**)&p->d_scale[l], (size_t)2 * cout[l] * sizeof(float)) != hipSuccess) {
            if (err) *err = "audio_mid: allocation failed";
            audio_mid_destroy(p);
            return -2;
        }
        p->d_shift[l] = p->d_scale[l] + cout[l];
        if (hipMemcpy(p->d_w[l], packed.data(), packed.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_scale[l], scale[l], cout[l] * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->d_shift[l], shift[l], cout[l] * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            if (err) *err = "audio_mid: upload failed";
            audio_mid_destroy(p);
            return -2;
        }
    }
    p->ready = true;
    return 0;
}

void audio_mid_destroy(AudioMidPlan* p) {
    for (int l = 0; l < 6; ++l) {
        if (p->d_w[l]) (void)hipFree(p->d_w[l]);
        if (p->d_scale[l]) (void)hipFree(p->d_scale[l]);
        p->d_w[l] = nullptr; p->d_scale[l] = nullptr; p->d_shift[l] = nullptr;
    }
    p->ready = false;
}

int audio_mid_launch(const AudioMidPlan& p, const f16* x, int x_stride, f16* y, int y_stride, int nframes, hipStream_t s, std::string* err) {
    if (!p.ready || nframes <= 0) { if (err) *err = "audio_mid: no plan"; return -1; }
    AudioMidArgs a;
    a.x = x; a.x_stride = x_stride; a.y = y; a.y_stride = y_stride; a.N = nframes;
    for (int l = 0; l < 6; ++l) { a.w[l] = p.d_w[l]; a.scale[l] = p.d_scale[l]; a.shift[l] = p.d_shift[l]; }
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)audio_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * REGION) != hipSuccess) {
            if (err) *err = "audio_mid: cannot set the dynamic LDS size";
            return -2;
        }
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(audio_mid_kernel, dim3((unsigned)nframes), dim3(512), 2 * REGION, s, a);
    if (hipGetLastError() != hipSuccess) { if (err) *err = "audio_mid: launch failed"; return -2; }
    return 0;
}

}  // namespace ltk
