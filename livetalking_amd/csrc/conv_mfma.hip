// Implicit-GEMM fp16 convolution for gfx950 (CDNA4) on MFMA, first generation (register-staged); activations are
// channel-blocked CB16 ([N][C/16][H][W][16]), inputs of <= 8 channels [N][H][W][8].  Today it runs the 7x7, the
// shallow strided 3x3 and the valid-padding layers; conv3_mfma.hip runs the rest.
//
// Replaces the torch.nn.Conv2d / ConvTranspose2d + BatchNorm2d(eval) + residual
// + ReLU blocks of the reference generator (avatars/wav2lip/models/conv.py:5-44,
// instantiated at avatars/wav2lip/models/wav2lip_v2.py:12-91).
//
// Mapping (one workgroup = 4 waves = 256 output pixels x BN output channels):
//   * the input patch a tile needs ((TH-1)*s+k rows x (TW-1)*s+k cols, zero padded
//     at the image border) is staged ONCE per 16/32-channel chunk into LDS as
//     channel planes [c8][patch pixel][8 halfs]; all k*k taps then read their
//     operand from that patch at a shifted address, so a 3x3 layer reads its
//     input once from HBM/L2 instead of nine times;
//   * weights are pre-packed on the host in LDS image order
//     [cout tile][chunk][tap][c8][cout][8 halfs], so a chunk's slab is one
//     contiguous, fully coalesced copy;
//   * v_mfma_f32_32x32x16_f16 with the WEIGHTS as the row operand and the PIXELS
//     as the column operand: a lane's 16 accumulators then are 4 groups of 4
//     consecutive output channels of ONE pixel -> 8-byte stores into the pixel's channel block, and the
//     folded BN scale/shift, residual add and ReLU are applied in registers;
//   * channel-offset reads/writes (x_ld/x_coff, y_ld/y_coff) make the decoder's
//     torch.cat skip connections (wav2lip_v2.py:146) free;
//   * ConvTranspose2d(k3,s2,p1,op1) runs as its 4 sub-pixel phases (1/2/2/4
//     taps over a 2x2 input neighbourhood) in one launch: no zero-insertion.
//
// LDS (dynamic, one array - keeps hipcc from draining vmcnt per k-step):
//   [2 x A planes][2 x B slab][tap table]
#include "conv_mfma.h"
#include "tune.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace ltk {

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PhaseMeta {
    int T, ooy, oox, w_off16;
    signed char dy[kMaxTaps], dx[kMaxTaps];
};

struct KArgs {
    const f16* x; const f16* w; const float* scale; const float* shift; const f16* res; f16* y;
    const PhaseMeta* phases;
    int N, H, W, x_cbt, x_cb0;        // channel-blocked [N][C/16][H][W][16]: blocks in the buffer, first block
    int Ho, Wo, y_cbt, y_cb0, HoA, WoA, osy, osx;
    int res_cbt, res_cb0;
    int Cin8, Cout;
    int sh, sw, pad_y, pad_x;
    int PH, PW, NPIXP;
    int log2TW, log2TH, NB;
    int tiles_x, tiles_y, tiles_n, n_ntiles;
    unsigned magicPW, magicPHW;
    int nchunks, Tp, relu;
    int lds_bytes;
    int s2half;                       // > 0: column-parity split of the patch rows (stride-2 3x3 layers): patch column px sits at slot
                                      // (px >> 1) + (px & 1) * s2half of its row, so the 16 lanes of a ds_read_b128 group - 16 consecutive
                                      // output pixels, i.e. every OTHER patch pixel - read 256 contiguous bytes instead of two 16-byte
                                      // items per 32 (a 2-way bank conflict on every A read: 30.7 % of this kernel's LDS cycles in round 5)
};

// A-patch staging registers per thread (16-byte items): the largest variant a configuration has.
__host__ __device__ constexpr int max_a_items(int NC8, int NBT) {
    return NC8 == 4 ? 6 : (NC8 == 1 ? 5 : (NBT == 4 ? 3 : 10));
}
constexpr int kMaxBItems = 9;   // 16-byte weight items a thread stages per chunk
constexpr int kTapTableBytes = 128;
constexpr int kLdsLimit = 160 * 1024;

// MODE 0: double-buffered LDS (one barrier per chunk, one workgroup per CU when the slabs are large)
// MODE 1: single-buffered LDS + register prefetch (two barriers per chunk, several workgroups per CU
//         overlap each other's load / epilogue phases)
template <int NC8, int NBT, bool TT9, int MODE, int MAXA>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BN = NBT * 32;
    constexpr int MAXB = kMaxBItems;
    constexpr int LOG2NC8 = NC8 == 4 ? 2 : (NC8 == 2 ? 1 : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    // ---- block -> (phase, image tile, y tile, x tile, cout tile); consecutive
    // logical ids (same pixel tile, different cout tiles) share an XCD's L2.
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int ntile = bid % a.n_ntiles;
    int t0 = bid / a.n_ntiles;
    const int tx_t = t0 % a.tiles_x; t0 /= a.tiles_x;
    const int ty_t = t0 % a.tiles_y; t0 /= a.tiles_y;
    const int tn_t = t0 % a.tiles_n;
    const int phase = t0 / a.tiles_n;

    const PhaseMeta* __restrict__ pm = a.phases + phase;
    const int T = pm->T;
    const int ooy = pm->ooy, oox = pm->oox;

    const int A_BYTES = NC8 * a.NPIXP * 16;
    const int B_BYTES = a.Tp * NC8 * BN * 16;
    constexpr int NBUF = (MODE == 0) ? 2 : 1;
    unsigned char* const smemA = smem;
    unsigned char* const smemB = smem + NBUF * A_BYTES;
    // the tap table sits at the very end of the allocation (the epilogue reuses the front)
    short* const tapl = reinterpret_cast<short*>(smem + a.lds_bytes - kTapTableBytes);

    if (tid < kMaxTaps) {
        short v = 0;
        if (tid < T) v = (short)(pm->dy[tid] * a.PW + pm->dx[tid]);
        tapl[tid] = v;
    }

    const int TWm = (1 << a.log2TW) - 1, THm = (1 << a.log2TH) - 1;
    const int tx0 = tx_t << a.log2TW, ty0 = ty_t << a.log2TH, n0 = tn_t * a.NB;
    const int iy0 = ty0 * a.sh - a.pad_y, ix0 = tx0 * a.sw - a.pad_x;
    const int PHW = a.PH * a.PW;
    const int nitemsA = a.NB * PHW * NC8;

    // ---- A staging descriptors (chunk independent)
    int a_goff[MAXA];
    int a_slot[MAXA];                 // LDS pixel slot of the item (the identity unless s2half)
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        const int i = tid + k * 256;
        const int pix = i >> LOG2NC8;
        const int c8 = i & (NC8 - 1);
        // exact for pix < 2^16; a divisor of 1 has no 32-bit magic
        const int b = (PHW == 1) ? pix : (int)__umulhi((unsigned)pix, a.magicPHW);
        const int rem = pix - b * PHW;
        const int py = (a.PW == 1) ? rem : (int)__umulhi((unsigned)rem, a.magicPW);
        const int px = rem - py * a.PW;
        const int n = n0 + b, iy = iy0 + py, ix = ix0 + px;
        a_slot[k] = a.s2half ? b * PHW + py * a.PW + (px >> 1) + (px & 1) * a.s2half : pix;
        const bool ok = (i < nitemsA) && (n < a.N) && ((unsigned)iy < (unsigned)a.H) && ((unsigned)ix < (unsigned)a.W);
        if constexpr (NC8 == 1)   // the two network inputs (packed face crops, mel windows) are plain [N][H][W][8]
            a_goff[k] = ok ? (((n * a.H + iy) * a.W + ix) * 8) : -1;
        else
            a_goff[k] = ok ? ((((n * a.x_cbt + a.x_cb0 + (c8 >> 1)) * a.H + iy) * a.W + ix) * 16 + (c8 & 1) * 8) : -1;
    }

    // weights are packed per 32-cout sub-slab: [cout/32][chunk][tap][plane][32][8 halfs]; a block
    // with BN = NBT*32 stages NBT sub-slabs -> LDS image [sub][tap][plane][32][16 B]
    const int slab32 = a.Tp * NC8 * 32;   // 16-byte items per (sub-slab, chunk)
    const int slab = slab32 * NBT;
    const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>(a.w) + pm->w_off16 + (size_t)(ntile * NBT) * a.nchunks * slab32;

    uint4 ra[MAXA];
    uint4 rb[MAXB];

    const int HW16 = a.H * a.W * 16;
    auto load_chunk = [&](int c) {
        const int cbase = c * NC8;
        const size_t coff = (size_t)c * (NC8 / 2) * HW16;     // chunk = NC8/2 channel blocks
#pragma unroll
        for (int k = 0; k < MAXA; ++k) {
            const int c8 = (tid + k * 256) & (NC8 - 1);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (a_goff[k] >= 0 && cbase + c8 < a.Cin8)
                v = *reinterpret_cast<const uint4*>(a.x + a_goff[k] + coff);
            ra[k] = v;
        }
        const uint4* ws = wsrc + (size_t)c * slab32;
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
            const int i = tid + k * 256;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < slab) {
                int sub = 0;
                if constexpr (NBT > 1) sub = (i >= slab32) + (NBT > 2 ? ((i >= 2 * slab32) + (i >= 3 * slab32)) : 0);
                v = ws[(size_t)sub * a.nchunks * slab32 + (i - sub * slab32)];
            }
            rb[k] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* Ab = smemA + buf * A_BYTES;
        unsigned char* Bb = smemB + buf * B_BYTES;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) {
            const int i = tid + k * 256;
            if (i < nitemsA) {
                const int c8 = i & (NC8 - 1);
                *reinterpret_cast<uint4*>(Ab + (c8 * a.NPIXP + a_slot[k]) * 16) = ra[k];
            }
        }
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
            const int i = tid + k * 256;
            if (i < slab) *reinterpret_cast<uint4*>(Bb + i * 16) = rb[k];
        }
    };

    // ---- per-lane operand bases: this lane's output pixel in each 32-pixel subtile
    int pixb[2];
    int obase[2], rbase[2];  // element offset of (output pixel, first channel block of this cout tile) in y / res, -1 = none
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = wave * 64 + j * 32 + l31;
        const int tx = m & TWm;
        const int ty = (m >> a.log2TW) & THm;
        const int b = m >> (a.log2TW + a.log2TH);
        const bool inb = b < a.NB;
        pixb[j] = inb ? ((b * a.PH + ty * a.sh) * a.PW + (a.s2half ? tx : tx * a.sw)) * 16 : 0;      // s2half: column 2 tx sits at slot tx
        const int n = n0 + b, oy = ty0 + ty, ox = tx0 + tx;
        const bool rowok = inb && n < a.N && oy < a.Ho && ox < a.Wo;
        const int opx = (oy * a.osy + ooy) * a.WoA + ox * a.osx + oox;
        const int cbo = (ntile * BN) >> 4;
        obase[j] = rowok ? (((n * a.y_cbt + a.y_cb0 + cbo) * (a.HoA * a.WoA) + opx) * 16) : -1;
        rbase[j] = rowok ? (((n * a.res_cbt + a.res_cb0 + cbo) * (a.HoA * a.WoA) + opx) * 16) : -1;
    }

    f32x16 acc[NBT][2];
#pragma unroll
    for (int i = 0; i < NBT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int PS = a.NPIXP * 16;
    const int nchunks = a.nchunks;
    auto compute_chunk = [&](int c, int cur) {
        const unsigned char* Ab = smemA + cur * A_BYTES;
        const unsigned char* Bb = smemB + cur * B_BYTES;
        if constexpr (NC8 == 1) {
            // 8-channel input: one MFMA k16 step = two taps (lanes 0-31 tap 2s, lanes 32-63 tap 2s+1)
            const int nsteps = a.Tp >> 1;
            for (int s = 0; s < nsteps; ++s) {
                const int tp = 2 * s + hh;
                const int toff = (int)tapl[tp] * 16;
                f16x8 xa[2], wf[NBT];
#pragma unroll
                for (int j = 0; j < 2; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ab + pixb[j] + toff);
#pragma unroll
                for (int i = 0; i < NBT; ++i) wf[i] = *reinterpret_cast<const f16x8*>(Bb + (((i * a.Tp + tp) * 32) + l31) * 16);
#pragma unroll
                for (int i = 0; i < NBT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xa[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const int planes = min(NC8, a.Cin8 - c * NC8);
            auto tap_body = [&](int t, int toff) {
#pragma unroll
                for (int q = 0; q < NC8 / 2; ++q) {
                    if (2 * q < planes) {
                        const int plane = 2 * q + hh;
                        f16x8 xa[2], wf[NBT];
#pragma unroll
                        for (int j = 0; j < 2; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ab + plane * PS + pixb[j] + toff);
#pragma unroll
                        for (int i = 0; i < NBT; ++i)
                            wf[i] = *reinterpret_cast<const f16x8*>(Bb + ((((i * a.Tp + t) * NC8 + plane) * 32) + l31) * 16);
#pragma unroll
                        for (int i = 0; i < NBT; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xa[j], acc[i][j], 0, 0, 0);
                    }
                }
            };
            if constexpr (TT9) {
                const int dxo1 = a.s2half ? a.s2half : 1, dxo2 = a.s2half ? 1 : 2;     // slot offset of patch column +1 / +2 (wave-uniform)
#pragma unroll
                for (int t = 0; t < 9; ++t) tap_body(t, ((t / 3) * a.PW + ((t % 3) == 0 ? 0 : (t % 3) == 1 ? dxo1 : dxo2)) * 16);
            } else {
                for (int t = 0; t < T; ++t) tap_body(t, (int)tapl[t] * 16);
            }
        }

    };

    load_chunk(0);
    if constexpr (MODE == 0) {
        store_chunk(0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int cur = c & 1;
            const bool more = (c + 1) < nchunks;
            if (more) load_chunk(c + 1);
            compute_chunk(c, cur);
            if (more) store_chunk(cur ^ 1);
            __syncthreads();
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            if (c) __syncthreads();          // every wave is done reading chunk c-1
            store_chunk(0);
            __syncthreads();
            if (c + 1 < nchunks) load_chunk(c + 1);   // in flight under the MFMAs
            compute_chunk(c, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: y = relu(acc*scale + shift + res) -> fp16, through a wave-private LDS
    // transpose so the residual is READ and the result WRITTEN as whole 16-byte channel
    // segments of a pixel row (8-byte scattered accesses from the MFMA C layout were
    // issue-bound).  The main loop ended with a barrier: the front of LDS is free.
    constexpr int ROWB = BN * 2 + 16;   // bytes per pixel row in the transpose region (+16: bank spread)
    constexpr int CBN = BN / 16;        // channel blocks per block (BN = 32: 2)
    unsigned char* const wreg = smem + wave * (64 * ROWB);
    const int cout0 = ntile * BN;
    const int ncb_valid = min(CBN, (a.Cout - cout0) >> 4);
    const bool has_res = a.res != nullptr;
    const int HWo16 = a.HoA * a.WoA * 16;

    if (has_res) {
#pragma unroll
        for (int it = 0; it < 2 * CBN; ++it) {
            const int idx = it * 64 + lane;
            const int cbl = idx >> 7, px = (idx >> 1) & 63, half = idx & 1;
            const int r0 = __shfl(rbase[0], px & 31), r1 = __shfl(rbase[1], px & 31);
            const int rb = (px & 32) ? r1 : r0;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (rb >= 0 && cbl < ncb_valid) v = *reinterpret_cast<const uint4*>(a.res + rb + cbl * HWo16 + half * 8);
            *reinterpret_cast<uint4*>(wreg + px * ROWB + (cbl * 16 + half * 8) * 2) = v;
        }
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = i * 32 + 8 * g + 4 * hh;     // channel within the block's BN
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cout0 + cl);   // padded to CoutPad
            const f32x4 sf = *reinterpret_cast<const f32x4*>(a.shift + cout0 + cl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned char* p = wreg + (j * 32 + l31) * ROWB + cl * 2;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][4 * g + r] * sc[r] + sf[r];
                if (has_res) {
                    const f16x4 rr = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                }
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = v[r];
                    if (a.relu == 1) t = fmaxf(t, 0.f);
                    else if (a.relu >= 2) {      // rare (Whisper convs): keep the transcendental code out of the common path
                        if (a.relu == 2) t = 0.5f * t * (1.f + erff(t * 0.70710678118654752f));
                        else t = t / (1.f + __expf(-t));
                    }
                    t = fminf(fmaxf(t, -65504.f), 65504.f);
                    o[r] = (f16)t;
                }
                *reinterpret_cast<f16x4*>(p) = o;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 2 * CBN; ++it) {
        const int idx = it * 64 + lane;
        const int cbl = idx >> 7, px = (idx >> 1) & 63, half = idx & 1;
        const int o0 = __shfl(obase[0], px & 31), o1 = __shfl(obase[1], px & 31);
        const int ob = (px & 32) ? o1 : o0;
        const uint4 v = *reinterpret_cast<const uint4*>(wreg + px * ROWB + (cbl * 16 + half * 8) * 2);
        if (ob >= 0 && cbl < ncb_valid) *reinterpret_cast<uint4*>(a.y + ob + cbl * HWo16 + half * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

typedef void (*conv_kernel_t)(const KArgs);

template <int NC8, int NBT, bool TT9, int MAXA>
static conv_kernel_t kptr(int mode) {
    return mode == 0 ? (conv_kernel_t)conv_mfma_kernel<NC8, NBT, TT9, 0, MAXA> : (conv_kernel_t)conv_mfma_kernel<NC8, NBT, TT9, 1, MAXA>;
}

// `need` = 16-byte A items per thread this launch stages; returns the smallest instantiation that holds them.
static conv_kernel_t pick_kernel(int NC8, int NBT, bool tt9, int mode, int need) {
    if (NC8 == 1) return (NBT == 1 && need <= 5) ? kptr<1, 1, false, 5>(mode) : nullptr;
    if (NC8 == 4) {
        if (need > 6) return nullptr;
        if (NBT == 1) return tt9 ? kptr<4, 1, true, 6>(mode) : kptr<4, 1, false, 6>(mode);
        if (NBT == 2) return tt9 ? kptr<4, 2, true, 6>(mode) : kptr<4, 2, false, 6>(mode);
        if (NBT == 4) return tt9 ? kptr<4, 4, true, 6>(mode) : kptr<4, 4, false, 6>(mode);
    }
    if (NC8 == 2) {
        if (need <= 3) {
            if (NBT == 1) return tt9 ? kptr<2, 1, true, 3>(mode) : kptr<2, 1, false, 3>(mode);
            if (NBT == 2) return tt9 ? kptr<2, 2, true, 3>(mode) : kptr<2, 2, false, 3>(mode);
            if (NBT == 4) return tt9 ? kptr<2, 4, true, 3>(mode) : kptr<2, 4, false, 3>(mode);
        } else if (need <= 10) {
            if (NBT == 1) return tt9 ? kptr<2, 1, true, 10>(mode) : kptr<2, 1, false, 10>(mode);
            if (NBT == 2) return tt9 ? kptr<2, 2, true, 10>(mode) : kptr<2, 2, false, 10>(mode);
        }
    }
    return nullptr;
}

static int ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

void ConvPlan::out_dims(int H, int W, int* Ho, int* Wo) const {
    if (!transposed) {   // out_pad on a plain conv = extra zero rows/columns at the bottom/right (diffusers Downsample2D
                         // with padding=0: F.pad(x, (0,1,0,1)) then a stride-2 "valid" conv)
        *Ho = (H + 2 * ph + out_pad - kh) / sh + 1;
        *Wo = (W + 2 * pw + out_pad - kw) / sw + 1;
    } else {
        *Ho = (H - 1) * sh - 2 * ph + kh + out_pad;
        *Wo = (W - 1) * sw - 2 * pw + kw + out_pad;
    }
}

double ConvPlan::macs_per_image(int H, int W) const {
    int Ho, Wo;
    out_dims(H, W, &Ho, &Wo);
    if (!transposed) return (double)Cout * kh * kw * Ho * Wo;  // x real Cin applied by caller
    return (double)Cout * kh * kw * H * W;
}

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(_e);        \
            return -2;                                                                \
        }                                                                             \
    } while (0)

unsigned char f32_to_e4m3(float v) {
    // OCP e4m3fn: 1-4-3, bias 7, max 448 (0x7E), no infinities; subnormal step 2^-9
    unsigned char sign = 0;
    if (v < 0.f) { sign = 0x80; v = -v; }
    if (!(v == v)) return 0x7F;                      // NaN
    if (v >= 448.f) return sign | 0x7E;              // saturate (incl. the half-way case above 448)
    if (v < 0.0009765625f) return sign;              // < 2^-10: rounds to zero (2^-10 itself is a tie -> even = 0)
    int e;
    (void)std::frexp(v, &e);                         // v = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                   // v = 1.x * 2^E
    if (E < -6) E = -6;                              // subnormal range shares the exponent of the smallest normal
    const float q = std::ldexp(v, 3 - E);            // in units of the spacing 2^(E-3)
    float r = std::nearbyint(q);                     // round half to even (default rounding mode)
    int mant = (int)r;                               // normals: 8..16, subnormals: 0..8
    int be = E + 7;
    if (E == -6 && mant < 8) return sign | (unsigned char)mant;             // subnormal (biased exponent 0)
    if (mant == 16) { mant = 8; ++be; }
    if (be > 15 || (be == 15 && mant - 8 > 6)) return sign | 0x7E;
    return sign | (unsigned char)((be << 3) | (mant - 8));
}

int conv_plan_create(ConvPlan* p, const float* weight, int CinArg, int Cout, int kh, int kw,
                     int sh, int sw, int ph, int pw, bool transposed, int out_pad,
                     const float* scale, const float* shift, std::string* err, int hint_hw, int quant, float act_scale, int ups4) {
    *p = ConvPlan();
    if (quant) {
        if (transposed || kh != 3 || kw != 3 || sh != 1 || sw != 1 || ph != 1 || pw != 1 || CinArg % 32 != 0) {
            if (err) *err = "fp8 operands: 3x3 stride-1 pad-1 convs with Cin % 32 == 0 only";
            return -1;
        }
        p->q8 = true;
        if (quant == 2) {
            if (CinArg % 64 != 0) { if (err) *err = "MX fp8 operands: Cin % 64 == 0"; return -1; }
            p->mx = true;
        }
    }
    p->CinReal = CinArg;
    // fp8: two channels per 16-bit unit; everything below (chunking, pack order, launch geometry) sees CinArg / 2 "channels"
    const int CinReal = quant ? CinArg / 2 : CinArg;
    // channel-blocked layout: whole 16-channel blocks; inputs of <= 8 channels are plain [N][H][W][8]
    const int Cin = CinReal <= 8 ? 8 : (CinReal + 15) / 16 * 16;
    p->kh = kh; p->kw = kw; p->sh = sh; p->sw = sw; p->ph = ph; p->pw = pw;
    p->transposed = transposed; p->out_pad = out_pad;
    p->Cin = Cin; p->Cout = Cout;

    // logical conv the kernel executes
    int lCout = Cout;        // output channels of the executed conv
    int lsh = sh, lsw = sw;  // input stride
    struct Tap { int ky, kx, dy, dx; };
    std::vector<std::vector<Tap>> phases;
    std::vector<std::pair<int, int>> phase_off;
    const bool v3_on = knob(K_CONV_V3) != 0;
    const bool is_convT_s2 = transposed && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && out_pad == 1;
    // Upsample2D = F.interpolate(scale_factor=2, nearest) + Conv2d(3x3, pad 1).  Output pixel (2y+py, 2x+px) reads upsampled rows
    // 2y+py+ky-1, i.e. source rows y + floor((py+ky-1)/2): phase py=0 sees rows {y-1 <- ky 0, y <- ky 1,2}, phase py=1 rows
    // {y <- ky 0,1, y+1 <- ky 2} (columns alike).  So every phase is a 2x2 conv on the SOURCE map whose weights are sums of the
    // 3x3 taps: 16 (operand, phase) products per source pixel instead of 36.  Operands (dy,dx) in 0..2 address the 3x3 source
    // neighbourhood (patch origin y-1, x-1); the kernel walks them row-major and feeds phases g = (py<<1)|px in ascending order.
    std::vector<float> weff;
    const float* wsrc = weight;
    int wkh = kh, wkw = kw;
    if (ups4 && v3_on && knob(K_UPS4) && !transposed && !quant && kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 &&
        out_pad == 0 && Cin % 16 == 0) {
        p->v3 = true; p->v3_G = 4; p->v3_T = 16; p->ups4 = true;
        lsh = lsw = 1;
        std::vector<Tap> taps;
        weff.assign((size_t)Cout * CinReal * 16, 0.f);
        int t = 0;
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx)
                for (int py = 0; py < 2; ++py)
                    for (int px = 0; px < 2; ++px) {
                        if (dy != py && dy != py + 1) continue;
                        if (dx != px && dx != px + 1) continue;
                        // 3x3 taps of this phase that land on source offset (dy, dx)
                        int kys[2], nky = 0, kxs[2], nkx = 0;
                        for (int ky = 0; ky < 3; ++ky) if (1 + ((py + ky - 1) >> 1) == dy) kys[nky++] = ky;     // floor((py+ky-1)/2) + 1
                        for (int kx = 0; kx < 3; ++kx) if (1 + ((px + kx - 1) >> 1) == dx) kxs[nkx++] = kx;
                        for (int co = 0; co < Cout; ++co)
                            for (int ci = 0; ci < CinReal; ++ci) {
                                float acc = 0.f;
                                for (int a = 0; a < nky; ++a)
                                    for (int b = 0; b < nkx; ++b) acc += weight[(((size_t)co * CinReal + ci) * 3 + kys[a]) * 3 + kxs[b]];
                                weff[((size_t)co * CinReal + ci) * 16 + t] = acc;
                            }
                        taps.push_back({t >> 2, t & 3, dy, dx});
                        ++t;
                    }
        if (t != 16) { if (err) *err = "ups4 tap table"; return -1; }
        phases.push_back(taps);
        phase_off.push_back({0, 0});
        wsrc = weff.data(); wkh = 4; wkw = 4;
    } else if (v3_on && is_convT_s2 && Cin % 16 == 0) {
        // conv3 merged transposed conv: ONE phase of 9 taps over a 2x2 input neighbourhood, in the tap order the
        // kernel's static table expects (offset (0,0): phases 0..3, (0,1): 1,3, (1,0): 2,3, (1,1): 3)
        p->v3 = true; p->v3_G = 4; p->v3_T = 9;
        lsh = lsw = 1;
        phases.push_back({{1, 1, 0, 0}, {1, 2, 0, 0}, {2, 1, 0, 0}, {2, 2, 0, 0},
                          {1, 0, 0, 1}, {2, 0, 0, 1}, {0, 1, 1, 0}, {0, 2, 1, 0}, {0, 0, 1, 1}});
        phase_off.push_back({0, 0});
    } else if (!transposed) {
        std::vector<Tap> taps;
        for (int y = 0; y < kh; ++y)
            for (int x = 0; x < kw; ++x) taps.push_back({y, x, y, x});
        phases.push_back(taps);
        phase_off.push_back({0, 0});
    } else if (sh == 1 && sw == 1 && ph == 0 && pw == 0 && out_pad == 0) {
        // k x k transposed conv: only the 1x1-input case is supported (a GEMM to k*k*Cout channels)
        p->gemm_1x1_expand = true;
        lCout = kh * kw * Cout;
        phases.push_back({{0, 0, 0, 0}});
        phase_off.push_back({0, 0});
    } else if (kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && out_pad == 1) {
        lsh = lsw = 1;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                std::vector<std::pair<int, int>> ys, xs;  // (k index, input offset)
                if (py == 0) ys = {{1, 0}}; else ys = {{0, 1}, {2, 0}};
                if (px == 0) xs = {{1, 0}}; else xs = {{0, 1}, {2, 0}};
                std::vector<Tap> taps;
                for (auto& yy : ys)
                    for (auto& xx : xs) taps.push_back({yy.first, xx.first, yy.second, xx.second});
                phases.push_back(taps);
                phase_off.push_back({py, px});
            }
    } else {
        if (err) *err = "unsupported transposed conv configuration";
        return -1;
    }

    if (v3_on && !p->v3 && !transposed && Cin % 16 == 0 && kh == 3 && kw == 3 && sh == 2 && sw == 2 &&
        ((ph == 1 && pw == 1 && out_pad == 0) || (ph == 0 && pw == 0 && out_pad == 1)) && knob(K_CONV_V3_S2) &&
        // measured in the pass (profiles/r04_s2_conflict_free.txt; conv3's stride-2 LDS image is bank-conflict free since round 4), register-
        // staged kernel -> conv3, 16 / 256 frames: 64->128 @64^2 16.9 -> 16.4 / 127 -> 94 us, 128->256 @32^2 20.4 -> 17.2 / 93 -> 69 us: conv3 from
        // 64 input channels on.  The wide, shallow ones stay register-staged: 16->32 @256^2 22.7 -> 30.5 / 249 -> 389 us (one 16-channel chunk per
        // item: 18 MFMAs per wave behind a 36-KB patch copy), 32->64 @128^2 15.0 -> 16.9 at 16 frames (-52 us at 256: a per-launch choice would
        // need both weight packs).  The asymmetric-pad form exists only in conv3.
        (Cin >= 64 || out_pad == 1 || knob(K_CONV_V3_S2) == 2)) {
        p->v3 = true; p->v3_G = 1; p->v3_T = 9; p->v3_S = 2;
    }
    if (v3_on && !p->v3 && Cin % 16 == 0 && lsh == 1 && lsw == 1) {
        if (!transposed && kh == 3 && kw == 3 && ph == 1 && pw == 1) { p->v3 = true; p->v3_G = 1; p->v3_T = 9; }
        if ((!transposed && kh == 1 && kw == 1 && ph == 0 && pw == 0) || p->gemm_1x1_expand) { p->v3 = true; p->v3_G = 1; p->v3_T = 1; }
    }
    const int Cin8 = Cin / 8;
    int NC8, NBT;
    const int Tmax = (int)std::max_element(phases.begin(), phases.end(),
                                           [](const std::vector<Tap>& a, const std::vector<Tap>& b) { return a.size() < b.size(); })->size();
    const bool strided = (lsh > 1 || lsw > 1);
    if (Cin8 == 1) {
        NC8 = 1; NBT = 1;
        if (lCout > 32) { if (err) *err = "Cin<=8 layers support Cout<=32"; return -1; }
    } else {
        if (Cin8 % 2) { if (err) *err = "Cin must be 8 or a multiple of 16"; return -1; }
        NBT = lCout >= 128 ? 4 : (lCout >= 64 ? 2 : 1);
        NC8 = (Cin8 >= 4) ? 4 : 2;
        // tuning overrides (sweeps): LTK_CONV_NBT / LTK_CONV_NC8 apply where legal
        const int fn = knob(K_CONV_NBT), fc = knob(K_CONV_NC8);
        if (fn == 1 || fn == 2 || fn == 4) NBT = std::min(NBT, fn);
        if (fc == 2 || (fc == 4 && Cin8 >= 4)) NC8 = fc;
        if (strided || Tmax > 9) { NC8 = 2; NBT = std::min(NBT, 2); }   // large patches: 10 staging items, 2 planes
        // weight slab of one chunk must fit the staging registers (9 x 16 B per thread)
        while (((NC8 == 1) ? ((Tmax + 1) / 2 * 2) : Tmax) * NC8 * NBT * 32 > kMaxBItems * 256) {
            if (NC8 > 2) NC8 /= 2; else if (NBT > 1) NBT /= 2; else break;
        }
    }
    if (p->v3) {
        // 1x1: 64-channel chunks; wide outputs take 32-channel chunks so that a 128-cout block (conv3_launch) still fits two
        // resident blocks per CU
        if (p->v3_T == 1) NC8 = (Cin % 32 == 0 && lCout >= 128 && lCout % 128 == 0 && knob(K_GEMM_NC8) == 4) ? 4 : (Cin % 64 == 0) ? 8 : 2;
        else if (p->ups4) NC8 = 2;          // 16 weight matrices per chunk: 16-channel chunks keep two blocks per CU
        else if (p->v3_G == 4) NC8 = (Cin % 32 == 0) ? 4 : 2;
        else {
            // 16-channel chunks everywhere: measured (profiles/r02_conv_sweep.txt) equal or better than 32-channel chunks on
            // every 3x3 layer at 16 frames and 25-28 % better on the 256/512-channel 8^2..16^2 maps at 256 frames (the
            // smaller stage leaves room for two resident blocks and allows 512-pixel tiles)
            NC8 = knob(K_TILE_RULE) ? 2 : ((Cin % 32 != 0 || hint_hw >= 1024 || hint_hw == 0) ? 2 : 4);     // 0: round-1 rule (A/B)
            if (knob(K_CONV3_NC8) == 2 || (knob(K_CONV3_NC8) == 4 && Cin % 32 == 0)) NC8 = knob(K_CONV3_NC8);
        }
        if (p->v3_S == 2) NC8 = 2;
        if (p->mx) NC8 = 4;                 // 4 planes of 16 e4m3 channels = the 64 channels one MX MFMA contracts
        NBT = 2;
    }
    p->NC8 = NC8; p->NBT = NBT;      // NBT here = the widest block the staging registers allow
    p->tt9 = (!transposed && kh == 3 && kw == 3 && NC8 >= 2);
    p->nphase = (int)phases.size();
    p->mode = knob(K_CONV_MODE);
    const int Tp = (NC8 == 1) ? ((Tmax + 1) / 2 * 2) : Tmax;
    p->Tp = Tp;
    if (!p->v3) {
        if (Tp * NC8 * NBT * 32 > kMaxBItems * 256) { if (err) *err = "weight slab too large for staging registers"; return -1; }
        if (!pick_kernel(NC8, NBT, p->tt9, 1, 1)) { if (err) *err = "no kernel instantiation for this configuration"; return -1; }
    }

    const int CoutPad = (lCout + 127) / 128 * 128;   // any block width <= 128 tiles it evenly
    p->CoutPad = CoutPad;
    p->lCout = lCout;
    const int n_sub = CoutPad / 32;
    const int nchunks = (Cin8 + NC8 - 1) / NC8;
    const size_t slab_halfs = (size_t)Tp * NC8 * 32 * 8;
    const size_t phase_halfs = (size_t)n_sub * nchunks * slab_halfs;
    std::vector<f16> packed(phase_halfs * phases.size(), (f16)0.f);

    auto wval = [&](int co, int ci, int ky, int kx) -> float {
        if (ci >= CinReal) return 0.f;
        if (!transposed) return wsrc[(((size_t)co * CinReal + ci) * wkh + ky) * wkw + kx];
        return wsrc[(((size_t)ci * Cout + co) * wkh + ky) * wkw + kx];
    };
    // fp8: per-output-channel scale, and the 16-bit unit (two consecutive input channels, low byte first)
    std::vector<float> wq_scale;
    if (quant) {
        wq_scale.assign(Cout, 1.f);
        const size_t per = (size_t)CinArg * 9;
        for (int co = 0; co < Cout; ++co) {
            float m = 0.f;
            for (size_t i = 0; i < per; ++i) m = std::max(m, std::fabs(weight[(size_t)co * per + i]));
            if (m > 0.f) wq_scale[co] = 224.f / m;
        }
    }
    auto wbits = [&](int co, int ci, int ky, int kx) -> uint16_t {
        if (!quant) { const f16 h = (f16)wval(co, ci, ky, kx); uint16_t u; memcpy(&u, &h, 2); return u; }
        if (ci >= CinReal) return 0;
        const float s = wq_scale[co];
        const float lo = weight[(((size_t)co * CinArg + 2 * ci) * 3 + ky) * 3 + kx] * s;
        const float hi = weight[(((size_t)co * CinArg + 2 * ci + 1) * 3 + ky) * 3 + kx] * s;
        return (uint16_t)(f32_to_e4m3(lo) | (f32_to_e4m3(hi) << 8));
    };

    std::vector<PhaseMeta> metas(phases.size());
    for (size_t pi = 0; pi < phases.size(); ++pi) {
        PhaseMeta& m = metas[pi];
        memset(&m, 0, sizeof(m));
        m.T = (int)phases[pi].size();
        m.ooy = phase_off[pi].first;
        m.oox = phase_off[pi].second;
        m.w_off16 = (int)(pi * phase_halfs / 8);
        for (int t = 0; t < m.T; ++t) { m.dy[t] = (signed char)phases[pi][t].dy; m.dx[t] = (signed char)phases[pi][t].dx; }
        p->phase[pi].T = m.T; p->phase[pi].ooy = m.ooy; p->phase[pi].oox = m.oox; p->phase[pi].w_off16 = m.w_off16;
        f16* base = packed.data() + pi * phase_halfs;
        // the 32-cout sub-slabs are written to disjoint ranges: packed by a few host threads when the layer is large (a MuseTalk load
        // packs 0.9 G weights: 16 s on one thread)
        auto pack_range = [&](int nt0, int nt1) {
        for (int nt = nt0; nt < nt1; ++nt)
            for (int c = 0; c < nchunks; ++c)
                for (int t = 0; t < m.T; ++t)
                    for (int pl = 0; pl < NC8; ++pl) {
                        const int c8 = c * NC8 + pl;
                        if (c8 >= Cin8) continue;
                        for (int n = 0; n < 32; ++n) {
                            const int lco = nt * 32 + n;
                            if (lco >= lCout) continue;
                            int co = lco, ky = phases[pi][t].ky, kx = phases[pi][t].kx;
                            if (p->gemm_1x1_expand) {
                                // output channel order (cout block, position, 16): the result [N][k*k*Cout/16][1][1][16]
                                // IS the channel-blocked k x k map [N][Cout/16][k][k][16]
                                const int c16 = lco & 15, tt = lco >> 4, pos = tt % (kh * kw);
                                co = (tt / (kh * kw)) * 16 + c16; ky = pos / kw; kx = pos % kw;
                            }
                            f16* dst = base + ((((size_t)(nt * nchunks + c) * Tp + t) * NC8 + pl) * 32 + n) * 8;
                            for (int j = 0; j < 8; ++j) { const uint16_t u = wbits(co, c8 * 8 + j, ky, kx); memcpy(&dst[j], &u, 2); }
                        }
                    }
        };
        const size_t work = (size_t)n_sub * nchunks * m.T * NC8 * 256;
        const int nthr = work < (size_t)4 << 20 ? 1 : std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, n_sub}));
        if (nthr <= 1) pack_range(0, n_sub);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < nthr; ++k) th.emplace_back(pack_range, (int)((long long)n_sub * k / nthr), (int)((long long)n_sub * (k + 1) / nthr));
            for (std::thread& t : th) t.join();
        }
    }

    // folded BN parameters (padded to CoutPad; replicated per position for the 1x1-expand case)
    std::vector<float> sc(CoutPad, 0.f), sf(CoutPad, 0.f);
    for (int i = 0; i < lCout; ++i) {
        const int co = p->gemm_1x1_expand ? ((i >> 4) / (kh * kw)) * 16 + (i & 15) : i;
        sc[i] = scale ? scale[co] : 1.f; sf[i] = shift ? shift[co] : 0.f;
        if (quant) sc[i] /= wq_scale[co] * act_scale;        // acc = sum (w s_w)(x s_a)
    }

    p->w_bytes = packed.size() * sizeof(f16);
    HIPCHK(hipMalloc((void**)&p->d_w, p->w_bytes + metas.size() * sizeof(PhaseMeta) + 256));
    HIPCHK(hipMemcpy(p->d_w, packed.data(), p->w_bytes, hipMemcpyHostToDevice));
    // phase metadata lives behind the weights (16-byte aligned)
    {
        const size_t off = (p->w_bytes + 15) / 16 * 16;
        HIPCHK(hipMemcpy((char*)p->d_w + off, metas.data(), metas.size() * sizeof(PhaseMeta), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void**)&p->d_scale, CoutPad * sizeof(float) * 2));
    p->d_shift = p->d_scale + CoutPad;
    HIPCHK(hipMemcpy(p->d_scale, sc.data(), CoutPad * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_shift, sf.data(), CoutPad * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

void conv_plan_destroy(ConvPlan* p) {
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_scale) (void)hipFree(p->d_scale);
    p->d_w = nullptr; p->d_scale = nullptr; p->d_shift = nullptr;
}

static unsigned magic_u16(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }

int conv_launch(const ConvPlan& p, const ConvIO& io, hipStream_t stream, std::string* err) {
    if (p.v3) return conv3_launch(p, io, stream, err);
    KArgs a;
    memset(&a, 0, sizeof(a));
    const int NC8 = p.NC8;
    int HoA, WoA;
    p.out_dims(io.H, io.W, &HoA, &WoA);
    a.x = io.x; a.w = p.d_w; a.scale = p.d_scale; a.shift = p.d_shift; a.res = io.res; a.y = io.y;
    a.phases = reinterpret_cast<const PhaseMeta*>((const char*)p.d_w + (p.w_bytes + 15) / 16 * 16);
    a.N = io.N; a.H = io.H; a.W = io.W; a.x_cbt = io.x_ld >> 4; a.x_cb0 = io.x_coff >> 4;
    a.res_cbt = io.res_ld >> 4; a.res_cb0 = io.res_coff >> 4;
    a.Cin8 = p.Cin / 8;
    a.relu = io.relu ? 1 : io.act;
    if (io.ups) { if (err) *err = "input upsampling is a conv3 feature (3x3 s1 p1 / 1x1 layers)"; return -1; }
    int y_ld = io.y_ld;
    int kext_y, kext_x;
    if (p.gemm_1x1_expand) {
        if (io.H != 1 || io.W != 1) { if (err) *err = "k x k transposed conv only supported on 1x1 maps"; return -1; }
        if (io.y_ld != p.Cout || io.y_coff != 0) { if (err) *err = "1x1-expand output must be contiguous"; return -1; }
        a.Ho = 1; a.Wo = 1; a.HoA = 1; a.WoA = 1; a.osy = 1; a.osx = 1;
        y_ld = p.kh * p.kw * p.Cout;
        a.Cout = p.kh * p.kw * p.Cout;
        a.sh = a.sw = 1; a.pad_y = a.pad_x = 0; kext_y = kext_x = 1;
    } else if (p.transposed) {
        a.Ho = io.H; a.Wo = io.W; a.HoA = HoA; a.WoA = WoA; a.osy = 2; a.osx = 2;
        a.Cout = p.Cout; a.sh = a.sw = 1; a.pad_y = a.pad_x = 0; kext_y = kext_x = 2;
    } else {
        a.Ho = HoA; a.Wo = WoA; a.HoA = HoA; a.WoA = WoA; a.osy = 1; a.osx = 1;
        a.Cout = p.Cout; a.sh = p.sh; a.sw = p.sw; a.pad_y = p.ph; a.pad_x = p.pw; kext_y = p.kh; kext_x = p.kw;
    }
    a.y_cbt = y_ld >> 4; a.y_cb0 = io.y_coff >> 4;
    if (((NC8 == 1 ? 0 : (io.x_ld | io.x_coff)) | y_ld | io.y_coff | a.Cout) & 15 || (NC8 == 1 && (io.x_ld != 8 || io.x_coff != 0))) {
        if (err) *err = "channel counts/offsets must be multiples of 16 (channel-blocked layout)";
        return -1;
    }
    if (io.res && ((io.res_ld | io.res_coff) & 15)) { if (err) *err = "residual channel count/offset must be a multiple of 16"; return -1; }

    // tile: TW x TH output pixels x NB images, 256 rows
    int l2w = std::min(5, ceil_log2(a.Wo));
    int l2h = std::min(8 - l2w, ceil_log2(a.Ho));
    const int forced_nbt = knob(K_CONV_NBT);
    const int min_blocks = knob(K_CONV_MIN_BLOCKS);
    int NBT = std::min(p.NBT, 2);
    if (forced_nbt == 1 || forced_nbt == 2 || forced_nbt == 4) NBT = std::min(p.NBT, forced_nbt);
    const int maxpix = max_a_items(NC8, NBT) * 256 / NC8;
    int PH, PW, NB;
    for (;;) {
        PH = ((1 << l2h) - 1) * a.sh + kext_y;
        PW = ((1 << l2w) - 1) * a.sw + kext_x;
        NB = std::min(kConvBM >> (l2w + l2h), maxpix / (PH * PW));
        if (NB >= 1) break;
        if (l2h > 0) --l2h; else if (l2w > 0) --l2w; else { if (err) *err = "patch does not fit"; return -1; }
    }
    NB = std::min(NB, std::max(1, io.N));
    a.log2TW = l2w; a.log2TH = l2h; a.NB = NB; a.PH = PH; a.PW = PW;
    const int npix = NB * PH * PW;
    int NPIXP = npix;
    const int want = NC8 == 4 ? 2 : (NC8 == 2 ? 4 : 0);
    if (NC8 > 1) while ((NPIXP & 7) != want) ++NPIXP;
    a.NPIXP = NPIXP;
    a.magicPW = magic_u16(PW);
    a.magicPHW = magic_u16(PH * PW);
    a.tiles_x = (a.Wo + (1 << l2w) - 1) >> l2w;
    a.tiles_y = (a.Ho + (1 << l2h) - 1) >> l2h;
    a.tiles_n = (io.N + NB - 1) / NB;
    // block width: the widest (<= 64 couts unless forced) that still gives the chip >= min_blocks workgroups
    const long long mtiles = (long long)p.nphase * a.tiles_n * a.tiles_y * a.tiles_x;
    if (!forced_nbt)
        while (NBT > 1 && mtiles * ((p.lCout + 32 * NBT - 1) / (32 * NBT)) < min_blocks) NBT /= 2;
    const int BN = NBT * 32;
    a.n_ntiles = (p.lCout + BN - 1) / BN;
    a.nchunks = (a.Cin8 + NC8 - 1) / NC8;
    a.Tp = p.Tp;

    const size_t a_bytes = (size_t)NC8 * NPIXP * 16, b_bytes = (size_t)p.Tp * NC8 * BN * 16;
    const size_t epi_bytes = (size_t)4 * 64 * (BN * 2 + 16);
    int mode = p.mode ? 1 : 0;
    if (mode == 0 && 2 * (a_bytes + b_bytes) + kTapTableBytes > (size_t)kLdsLimit) mode = 1;
    size_t lds = (mode == 0 ? 2 : 1) * (a_bytes + b_bytes) + kTapTableBytes;
    lds = std::max(lds, epi_bytes + kTapTableBytes);
    lds = (lds + 255) / 256 * 256;
    if (lds > (size_t)kLdsLimit) { if (err) *err = "LDS budget exceeded"; return -1; }
    a.lds_bytes = (int)lds;
    // stride-2 3x3 layers whose tile rows hold >= 16 output pixels: column-parity split of the patch rows (KArgs::s2half)
    a.s2half = (p.tt9 && a.sw == 2 && l2w >= 4 && knob(K_CONV_S2SPLIT)) ? (PW + 1) / 2 : 0;
    // largest 32-bit element offset the kernel forms
    if ((double)io.N * io.H * io.W * io.x_ld >= 2147483647.0) { if (err) *err = "input tensor too large for 32-bit offsets"; return -1; }

    const int need = (npix * NC8 + 255) / 256;
    conv_kernel_t k = pick_kernel(NC8, NBT, p.tt9, mode, need);
    if (!k) { if (err) *err = "no kernel instantiation for this launch"; return -1; }
    const long long nblk = (long long)p.nphase * a.tiles_n * a.tiles_y * a.tiles_x * a.n_ntiles;
    if (nblk <= 0 || nblk > 0x7fffffffll) { if (err) *err = "bad grid"; return -1; }
    HIPCHK((hipError_t)ensure_dyn_lds((const void*)k, kLdsLimit));
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), lds, stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace ltk
