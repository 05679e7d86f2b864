// conv3: LDS-DMA staged implicit-GEMM convolution for gfx950 (CDNA4), second generation.
//
// Covers the layers that carry 97 % of the generator's MACs (avatars/wav2lip/models/
// wav2lip_v2.py:12-91 through conv.py:5-44):
//   * Conv2d 3x3 s1 p1 and 1x1 (T = 9 / 1 taps)                                  G = 1
//   * ConvTranspose2d(k3,s2,p1,op1): all four sub-pixel phases in ONE block       G = 4
//     (the (TH+1)x(TW+1) input patch is staged once per channel chunk and its four
//     shifted MFMA operands feed the 1/2/2/4 taps of the four phases)
// What changed against conv_mfma.hip (kept for the 7x7 / strided layers):
//   * staging is global_load_lds_dwordx4 straight into LDS (no staging VGPRs, no
//     ds_write pass), two LDS stages, one barrier per channel chunk;
//   * a wave owns PXW x 32 pixels x NBT x 32 output channels (G = 1: 128 px x 64 ch =
//     8 accumulator tiles, 0.75 ds_read_b128 per MFMA instead of 1.0), so a block's
//     weight slab is amortised over 512 pixels instead of 256;
//   * split-K over channel chunks (fixed per layer, so results do not depend on the
//     batch size) for the small-map layers whose grid cannot fill 256 CUs otherwise:
//     fp32 partial slabs + conv3_finish (fixed summation order -> deterministic).
// MFMA operand roles, folded-BN epilogue, channel-offset I/O and the weight pack order
// ([cout/32][chunk][tap][plane][32][8 halfs]) are those of conv_mfma.hip.
//
// Activation layout (all kernels): channel-blocked [N][C/16][H][W][16] fp16 ("CB16").  One MFMA k16 step
// consumes exactly one channel block; a patch row, an output row and a residual row are contiguous runs of
// 32 B per pixel, so the LDS-DMA, the residual read and the store all move whole 128-B lines (measured with
// scripts/ubench/glds_bw.hip: 126 GB/s/CU for whole lines against 17 GB/s/CU when a lane takes 16 B of a
// line, which is what pixel-interleaved NHWC costs a 16-channel chunk).
// A-patch image in LDS, per channel block: [patch pixel][2 x 16 B], the two halves swapped where bit 3 of the
// pixel slot is set, so the 16-lane groups of ds_read_b128 (32-B lane stride) hit 16 distinct 16-B slots.
// That holds when the 16 lanes of a group read ONE patch row (tile rows of 32 pixels).  On narrower maps (tile rows of 16, 8, 4 pixels) a group
// spans 2-8 patch rows and the column key collides across rows: 8 / 14.7 / 16 LDS cycles per ds_read_b128 instead of 4 on 16- / 8- / 4-pixel-wide
// tiles (a bank simulator over the guide's lane groups, tests/test_lds_layout.py; round 3's PMC shows it as 30 % conflict cycles on the 128-pixel-tile
// instantiation).  There the halves swap by the parity of the patch ROW instead (and 8-pixel-wide tiles pad their row pitch to 12 pixels): 4 cycles
// on 16- and 8-pixel rows, 8 on 4-pixel rows.  The key is a launch argument (swz_x / swz_row), chosen by the host from the tile width.
// Stride-2 convolutions (S = 2) read every OTHER patch pixel (64-B lane stride: lanes i and i + 4 of a 16-lane group would share a
// bank, a 4-way conflict - SQ_LDS_BANK_CONFLICT was 41 % of the LDS cycles of this instantiation in round 3).  Their image permutes the
// four 16-B units of every aligned pixel PAIR instead: unit q = 2 * (column & 1) + half sits at q ^ ((column >> 3) & 3), so the four
// 256-B windows a 16-lane group touches use four different unit slots.  The permutation stays inside 64 B of a row, so the LDS-DMA
// still moves whole lines; the patch row pitch is even for S = 2.  (Conflict-free when a lane group reads ONE patch row - 32-pixel-wide
// tiles; on 16- / 8-pixel-wide tiles a group spans several rows whose offsets the key does not see and part of the conflicts remain.)
#include "conv_mfma.h"
#include "misc_kernels.h"
#include "tune.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#ifndef LTK_ABLATE_BUILD
#define LTK_ABLATE_BUILD 0          // 1: keep the measurement branches (scripts/conv_ablate.py); production kernels carry none
#endif

namespace ltk {

#if LTK_ABLATE_BUILD
#define ABL(a, bit) ((a).ablate & (bit))
#else
#define ABL(a, bit) 0
#endif

typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef long i64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct K3Args {
    const f16* x; const f16* w; const float* scale; const float* shift; const f16* res; f16* y;
    float* partial;                  // split-K slabs [ksplit][Mtot][CoutP] or nullptr
    int N, H, W, x_cbt, x_cb0;       // input: channel blocks in the buffer, first block of this tensor
    int Ho, Wo;                      // output grid the tiles cover (= H, W except for stride 2)
    int y_cbt, y_cb0, HoA, WoA;      // output pixel (n,oy,ox) of block cb -> ((n*y_cbt + y_cb0+cb)*HoA + oy)*WoA + ox
    int res_cbt, res_cb0;
    int Cout, CoutP;                 // logical output channels; CoutP = padded (scale/shift/partial pitch)
    int pad;                         // 1 for 3x3, 0 for 1x1 / transposed
    int PH, PW, SLOTS, npix;         // SLOTS: 16-byte slots per channel-block image (2 per patch pixel, 64-padded)
    int log2TW, log2TH, NB;
    int tiles_x, tiles_y, tiles_n, n_ntiles;
    unsigned magicPW, magicPHW;
    int nchunks, ksplit, chunks_per_split;
    int relu;                        // activation: 0 none, 1 ReLU, 2 GELU (erf), 3 SiLU
    int ups;                         // 1: input is H/2 x W/2, read through a nearest 2x upsample
    int nitems;                      // work items (ksplit x pixel tiles x cout tiles); gridDim.x <= nitems
    int lds_scale_off;               // byte offset of the [2][BN] fp32 scale/shift image behind the stages
    int swz_x, swz_row;              // stride-1 LDS image: the two 16-B halves of a patch pixel swap where bit 3 of its COLUMN is set (swz_x: tile rows of 32
                                     // pixels) / where its patch ROW is odd (swz_row: narrower tiles, where a 16-lane read group spans several rows)
    // LayerNorm folded into the two linear layers around it (1x1 layers, MuseTalk transformer blocks; musetalk.hip build_transformer):
    //   ln_out: this layer's output is a tensor a LayerNorm normalises - the epilogue also writes, per token and 32-channel tile, the sum
    //           and the sum of squares of the fp16 values it stores ([Mtot][ln_out_tiles] float2; one writer per entry, no atomics);
    //   ln_in:  this layer CONSUMES a LayerNorm'ed tensor but reads the RAW one: with W' = W diag(gamma) packed as the weights,
    //           scale[co] = sum_ci W'[co][ci] and shift[co] = sum_ci W[co][ci] beta[ci] + bias[co], the epilogue computes
    //           rstd_t * (acc - mean_t * scale[co]) + shift[co] = W LN(x)_t + bias, mean_t / rstd_t from the producer's partials.
    float* ln_out; const float* ln_in;
    int ln_out_tiles, ln_in_tiles;
    float ln_eps;
    int ablate;                      // measurement builds only (make ABLATE=1, knob LTK_ABLATE): 1 no A DMA, 2 no B DMA, 4 no MFMA,
                                     // 8 no residual read, 16 no output store, 32 no LDS zero fill, 64 no epilogue, 128 epilogue math only
    long long Mtot;                  // N*HoA*WoA (slab pitch in pixels)
};

__host__ __device__ constexpr int k3_maxa(int PXW, int NC8, int S = 1, int T = 9) {
    // 16-byte A items per thread per chunk: (NC8/2) * SLOTS / 256; a stride-2 patch is ~4x the tile; a 1x1 conv has no halo
    return T == 1 ? (NC8 / 2) * PXW
                  : S == 2 ? (NC8 == 2 ? 10 : 20) : PXW == 4 ? (NC8 == 2 ? 6 : 12) : (NC8 == 2 ? 4 : (NC8 == 4 ? 8 : 16));
}
__host__ __device__ constexpr int k3_maxb(int NBT, int NC8, int T) { return (NBT * T * NC8 * 32 + 255) / 256; }

#define GLDS16(gptr, lptr)                                                                                   \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),                  \
                                     (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

// tap tables of the merged transposed conv: tap t reads input offset (dy,dx) = (o>>1, o&1) and
// accumulates into phase g = (py<<1)|px.  Host packs the weights in this tap order (k3_convT_taps).
//   t : 0 1 2 3 | 4 5 | 6 7 | 8
//   o : 0 0 0 0 | 1 1 | 2 2 | 3
//   g : 0 1 2 3 | 1 3 | 2 3 | 3
// one work item = one (k split, pixel tile, cout tile); `bid` is its logical id
// Q = 1: fp8 (OCP e4m3) operands.  A "channel block" is then 32 fp8 channels = the same 32 bytes per pixel, the LDS images,
// the DMA descriptors and the swizzle are byte-identical to the fp16 case, and a lane's 16-byte fragment feeds TWO
// v_mfma_f32_32x32x16_fp8_fp8 (bytes 0-7, then 8-15): half the LDS / DMA / HBM bytes per MAC at the same MFMA rate.
// The host packs the fp8 weights as pairs in 16-bit units (conv_plan_create, quant = 1) so that byte i of an A
// fragment and byte i of a B fragment are the same input channel.
// HEAD = 1 (output_block of the Wav2Lip generator, wav2lip_v2.py:89-91): the block's 32 output channels do not go to
// memory; the epilogue applies the 1x1 conv 32 -> 3 + bias, the sigmoid, *255 and the uint8 truncation of
// wav2lip_avatar.py:138,145 on the fp32 accumulators and writes the 3 bytes of every pixel to its frame's own output.
struct HeadArgs {
    const float* w;                  // [3][32] weights, then [3] bias (fp32)
    const OutPtrs* outs;             // DEVICE table, per frame: uint8 [256][256][3]
};

template <int G, int NBT, int PXW, int NC8, int T, int S, int Q, int HEAD = 0>
__device__ __forceinline__ void conv3_item(const K3Args& a, const int bid, unsigned char* const smem, const HeadArgs* hd = nullptr) {
    static_assert(HEAD == 0 || (G == 1 && NBT == 1 && T == 9 && S == 1 && Q == 0), "fused head: the 3x3 32-cout output conv");
    constexpr int BN = NBT * 32;
    constexpr int MAXA = k3_maxa(PXW, NC8, S, T);
    constexpr int MAXB = k3_maxb(NBT, NC8, T);
    static_assert(G == 1 || (G == 4 && (T == 9 || T == 16) && NBT == 1), "merged convT (9 taps) / upsample-conv (16): 32 couts per block");
    static_assert(NBT <= 2 || T == 1, "128-cout blocks: 1x1 convolutions only (accumulator budget)");
    static_assert(Q == 0 || G == 1, "fp8 operands: plain convolutions only");
    static_assert(Q != 2 || (NC8 == 4 && T == 9), "MX fp8: 64-channel chunks, 3x3");
    constexpr int MPP = Q == 1 ? 2 : 1;    // MFMAs per (cout subtile, pixel subtile) pair and k16 plane pair

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    // ---- item -> (k split, image tile, y tile, x tile, cout tile)
    const int ntile = bid % a.n_ntiles;
    int t0 = bid / a.n_ntiles;
    const int tx_t = t0 % a.tiles_x; t0 /= a.tiles_x;
    const int ty_t = t0 % a.tiles_y; t0 /= a.tiles_y;
    const int tn_t = t0 % a.tiles_n;
    const int ks = t0 / a.tiles_n;
    const int c_begin = ks * a.chunks_per_split;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_split);

    constexpr int NCB = NC8 / 2;           // channel blocks (k16 steps) per chunk
    const int A_BYTES = NCB * a.SLOTS * 16;
    constexpr int B_BYTES = T * NC8 * BN * 16;
    const int STAGE = A_BYTES + B_BYTES;

    const int TWm = (1 << a.log2TW) - 1, THm = (1 << a.log2TH) - 1;
    const int tx0 = tx_t << a.log2TW, ty0 = ty_t << a.log2TH, n0 = tn_t * a.NB;
    const int iy0 = ty0 * S - a.pad, ix0 = tx0 * S - a.pad;
    const int PHW = a.PH * a.PW;

    // ---- zero both A stages once: halo slots outside the image are never written by the DMA
    if (!ABL(a, 32)) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        const bool two = c_end - c_begin > 1;               // a one-chunk item (16-channel layers) never touches the second stage
        for (int i = tid * 16; i < A_BYTES; i += 256 * 16) {
            *reinterpret_cast<uint4*>(smem + i) = z;
            if (two) *reinterpret_cast<uint4*>(smem + STAGE + i) = z;
        }
    }

    // ---- A staging descriptors: item (k, wave) = 64 consecutive 16-byte slots (32 patch pixels) of one channel block
    const int p64n = a.SLOTS >> 6;
    const int HW16 = (a.H >> a.ups) * (a.W >> a.ups) * 16;          // halfs per (source) channel block plane
    // 32-bit BYTE offsets against a wave-uniform 64-bit base: the DMA then takes the SGPR-base + VGPR-offset form
    // (one VGPR per descriptor, no 64-bit VALU add per copy).  Tensors are < 2^31 elements (checked by the host).
    unsigned a_goff[MAXA];                    // byte offset of this lane's 8 channels (chunk 0), ~0u = no copy
    // LDS destination of copy (k, wave) inside a stage: the NCB channel-block images are contiguous and SLOTS is a multiple
    // of 64, so it is simply (k*4 + wave) KiB -- no table
    // The slot -> (image, row, column) decomposition below does not depend on the item.  In the persistent loop hipcc
    // hoists it out of the item loop, keeps ~2 VGPRs per descriptor alive across the whole kernel, runs out of
    // registers and reloads the spills from scratch at every item start -- with an s_waitcnt vmcnt(0) in front of
    // every DMA of the first chunk.  An opaque copy of the lane id makes it per-item work (~100 VALU ops).
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i));
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        const int s64 = k * 4 + wave;                       // wave-uniform
        const int cbj = s64 / p64n;
        const int slot = (s64 - cbj * p64n) * 64 + lane_i;
        const int pix = slot >> 1;
        const int b = (PHW == 1) ? pix : (int)__umulhi((unsigned)pix, a.magicPHW);
        const int rem = pix - b * PHW;
        const int py = (a.PW == 1) ? rem : (int)__umulhi((unsigned)rem, a.magicPW);
        const int px = rem - py * a.PW;
        int half, pxs = px;                                 // source column / half that this LDS slot holds
        if constexpr (S == 2) {
            const int q = (2 * (px & 1) + (slot & 1)) ^ ((px >> 3) & 3);      // self-inverse: slot unit -> source unit
            pxs = (px & ~1) | (q >> 1);
            half = q & 1;
        } else {
            half = (slot & 1) ^ (a.swz_x & (px >> 3) & 1) ^ (a.swz_row & (b * a.PH + py) & 1);   // column-bit-3 or row-parity key (header)
        }
        const int n = n0 + b, iy = iy0 + py, ix = ix0 + pxs;
        const bool ok = (cbj < NCB) && (pix < a.npix) && (n < a.N) && ((unsigned)iy < (unsigned)a.H) &&
                        ((unsigned)ix < (unsigned)a.W);
        a_goff[k] = ok ? (unsigned)((((n * a.x_cbt + a.x_cb0 + cbj) * (a.H >> a.ups) + (iy >> a.ups)) * (a.W >> a.ups) + (ix >> a.ups)) * 16 + half * 8) * 2u : ~0u;
    }

    // ---- B staging: NBT sub-slabs of slab32 16-byte items each; LDS image [sub][tap][plane][32]
    constexpr int slab32 = T * NC8 * 32;
    const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>(a.w) + (size_t)(ntile * NBT) * a.nchunks * slab32;
    // item tid + k*256 of the block's NBT sub-slabs; the second sub-slab starts nchunks*slab32 items after the first
    const unsigned b_sub1 = (unsigned)(a.nchunks - 1) * slab32 * 16u;      // extra byte offset of sub-slab 1 items

    auto stage = [&](int c, int buf) {
        unsigned char* const Ab = smem + buf * STAGE;
        unsigned char* const Bb = Ab + A_BYTES;
        const unsigned char* xc = reinterpret_cast<const unsigned char*>(a.x + (size_t)c * NCB * HW16);
        if (!ABL(a, 1)) {
#pragma unroll
            for (int k = 0; k < MAXA; ++k)
                if (a_goff[k] != ~0u) GLDS16(xc + a_goff[k], Ab + (k * 4 + wave) * 1024);
        }
        const unsigned char* wc = reinterpret_cast<const unsigned char*>(wsrc + (size_t)c * slab32);
        if (!ABL(a, 2)) {
            unsigned tq = (unsigned)tid;           // opaque: otherwise the per-thread offsets are hoisted as 64-bit
            asm volatile("" : "+v"(tq));           // kernel-lifetime values and spilled (see lane_i above)
#pragma unroll
            for (int k = 0; k < MAXB; ++k) {
                const unsigned i = tq + k * 256u;
                if (i < (unsigned)(NBT * slab32)) {
                    const unsigned off = i * 16u + (NBT > 1 ? (i / (unsigned)slab32) * b_sub1 : 0u);   // sub-slab s starts s*nchunks*slab32 items in
                    GLDS16(wc + off, Bb + (k * 256 + wave * 64) * 16);
                }
            }
        }
    };

    // ---- per-lane operand bases: this lane's pixel in each of the wave's PXW 32-pixel subtiles
    // Byte address (inside a channel-block image) of this lane's 16-B operand for subtile j and column offset dx; the row
    // offset dy * PW * 32 is wave-uniform.  The half swap follows the patch column only, so a tap costs ONE v_add per
    // fragment (uniform stage/row offset + aj[j][dx]); with the swap on the pixel index every tap re-derived it: 6 VALU per
    // fragment, 218 of the 385 non-MFMA instructions of a 72-MFMA chunk, which made the loop issue-bound.
    constexpr int DXN = ((T == 9 && G == 1) || (G == 4 && T == 16)) ? 3 : (G == 4 ? 2 : 1);
    int aj[PXW][DXN];
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        const int m = (wave * PXW + j) * 32 + l31;
        const int tx = m & TWm;
        const int ty = (m >> a.log2TW) & THm;
        const int b = m >> (a.log2TW + a.log2TH);
        const bool in = b < a.NB;
        const int prow = in ? (b * a.PH + ty * S) * a.PW : 0;
        const int pcol = in ? tx * S : 0;
#pragma unroll
        for (int dx = 0; dx < DXN; ++dx) {
            if constexpr (S == 2) {
                const int pc = pcol + dx;
                const int q = (2 * (pc & 1) + hh) ^ ((pc >> 3) & 3);
                aj[j][dx] = (prow + ((pc & ~1) | (q >> 1))) * 32 + ((q & 1) << 4);
            } else {
                // (row key: for tap row dy = 0; odd dy flips the half at the read, see arow())
                const int key = (a.swz_x & ((pcol + dx) >> 3) & 1) ^ (a.swz_row & (in ? b * a.PH + ty * S : 0) & 1);
                aj[j][dx] = (prow + pcol + dx) * 32 + ((key ^ (Q == 2 ? 0 : hh)) << 4);     // Q = 2 reads both halves
            }
        }
    }

    f32x16 acc[G][NBT][PXW];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < NBT; ++i)
#pragma unroll
            for (int j = 0; j < PXW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][i][j][r] = 0.f;

    const int PS = a.SLOTS * 16;
    const int rowB = a.PW * 32;            // bytes per patch row of a channel-block image
    // operand offset of subtile j / column dx for tap row dy: aj is computed for dy = 0; under the row key an odd tap row reads the other half
    const int rowflip = (S == 1) ? (a.swz_row << 4) : 0;      // wave-uniform
    auto arow = [&](int v, int dy) -> int { return (dy & 1) ? (v ^ rowflip) : v; };
    auto compute = [&](int buf) {
        const unsigned char* Ab = smem + buf * STAGE;
        const unsigned char* Bb = Ab + A_BYTES;
        // keep hipcc from hoisting the 9 x PXW swizzled operand addresses out of the chunk loop: they would
        // pin ~70 VGPRs and leave no room for the double-buffered fragments
#pragma unroll
        for (int j = 0; j < PXW; ++j)
#pragma unroll
            for (int dx = 0; dx < DXN; ++dx) asm volatile("" : "+v"(aj[j][dx]));
        if constexpr (Q == 2) {
            // MX-scaled fp8 (v_mfma_scale_f32_32x32x64_f8f6f4, E8M0 scale 127 = 1.0 on both operands): the chunk's 64 channels in
            // ONE MFMA per tap and tile pair.  Lane half hh owns the 32-byte cell hh of the chunk: both 16-byte halves of its pixel
            // (the second sits at the swizzle partner, address ^ 16) and weight planes 2hh, 2hh+1 of its cout row; byte j of
            // both operands is the same input channel, which is all the contraction needs.
            const unsigned char* Ap = Ab + hh * PS;
            i32x8 xa[2][PXW], wf[2][NBT];
            auto load_tap = [&](int t, int sl) {
                const unsigned char* Ar = Ap + (t / 3) * rowB;
#pragma unroll
                for (int i = 0; i < NBT; ++i) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(Bb + ((((i * T + t) * NC8 + 2 * hh) * 32) + l31) * 16);
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(Bb + ((((i * T + t) * NC8 + 2 * hh + 1) * 32) + l31) * 16);
                    wf[sl][i] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(Ar + arow(aj[j][t % 3], t / 3));
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(Ar + (arow(aj[j][t % 3], t / 3) ^ 16));
                    xa[sl][j] = (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            };
            load_tap(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NBT + PXW), 0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int sl = t & 1;
                if (t + 1 < T) load_tap(t + 1, sl ^ 1);
#pragma unroll
                for (int i = 0; i < NBT; ++i)
#pragma unroll
                    for (int j = 0; j < PXW; ++j)
                        acc[0][i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[sl][i], xa[sl][j], acc[0][i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                                       0x7f7f7f7f);
                if (t + 1 < T) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NBT + PXW), 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NBT * PXW, 0);
                }
            }
        } else
#pragma unroll
        for (int q = 0; q < NCB; ++q) {
            const int plane = 2 * q + hh;
            const unsigned char* Ap = Ab + q * PS;
            if constexpr (G == 1) {
                // software pipeline over the taps: the fragments of tap t+1 are read from LDS while the MFMAs of
                // tap t run (left to itself hipcc reads a fragment right before its first use and waits
                // lgkmcnt(0): one exposed LDS round trip per four MFMAs)
                f16x8 xa[2][PXW], wf[2][NBT];
                auto load_tap = [&](int t, int sl) {
                    const unsigned char* Ar = Ap + ((T == 9) ? (t / 3) * rowB : 0);      // wave-uniform
#pragma unroll
                    for (int i = 0; i < NBT; ++i)
                        wf[sl][i] = *reinterpret_cast<const f16x8*>(Bb + ((((i * T + t) * NC8 + plane) * 32) + l31) * 16);
#pragma unroll
                    for (int j = 0; j < PXW; ++j) xa[sl][j] = *reinterpret_cast<const f16x8*>(Ar + arow(aj[j][(T == 9) ? t % 3 : 0], (T == 9) ? t / 3 : 0));
                };
                load_tap(0, 0);
                if (T > 1) __builtin_amdgcn_sched_group_barrier(0x100, NBT + PXW, 0);       // reads of tap 0
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int sl = t & 1;
                    if (t + 1 < T) load_tap(t + 1, sl ^ 1);
#pragma unroll
                    for (int i = 0; i < NBT; ++i)
#pragma unroll
                        for (int j = 0; j < PXW; ++j) {
                            if constexpr (Q == 0) {
                                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[sl][i], xa[sl][j], acc[0][i][j], 0, 0, 0);
                            } else {
                                const i64x2 wq = __builtin_bit_cast(i64x2, wf[sl][i]);
                                const i64x2 xq = __builtin_bit_cast(i64x2, xa[sl][j]);
                                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wq[0], xq[0], acc[0][i][j], 0, 0, 0);
                                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wq[1], xq[1], acc[0][i][j], 0, 0, 0);
                            }
                        }
                    if (t + 1 < T) {
                        __builtin_amdgcn_sched_group_barrier(0x100, NBT + PXW, 0);         // DS reads of tap t+1 first
                        __builtin_amdgcn_sched_group_barrier(0x008, NBT * PXW * MPP, 0);   // then the MFMAs of tap t
                    }
                }
            } else if constexpr (T == 16) {
                // nearest-2x upsample + 3x3 conv as four 2x2-tap phases (ConvPlan::ups4): the nine source offsets (dy,dx) of
                // the (TH+2)x(TW+2) patch, each feeding the phases g = (py<<1)|px with dy in {py, py+1}, dx in {px, px+1};
                // weight matrix t follows that walk (conv_plan_create)
                auto wfrag = [&](int t) {
                    return *reinterpret_cast<const f16x8*>(Bb + (((t * NC8 + plane) * 32) + l31) * 16);
                };
                int t = 0;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        f16x8 xa[PXW];
#pragma unroll
                        for (int j = 0; j < PXW; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ap + dy * rowB + arow(aj[j][dx], dy));
#pragma unroll
                        for (int py = 0; py < 2; ++py)
#pragma unroll
                            for (int px = 0; px < 2; ++px) {
                                if ((dy != py && dy != py + 1) || (dx != px && dx != px + 1)) continue;
                                const f16x8 wf = wfrag(t++);
#pragma unroll
                                for (int j = 0; j < PXW; ++j)
                                    acc[py * 2 + px][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[j], acc[py * 2 + px][0][j], 0, 0, 0);
                            }
                    }
            } else {
                auto wfrag = [&](int t) {
                    return *reinterpret_cast<const f16x8*>(Bb + (((t * NC8 + plane) * 32) + l31) * 16);
                };
                f16x8 xa[PXW];
                // offset (0,0): taps 0..3 -> phases 0..3
#pragma unroll
                for (int j = 0; j < PXW; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ap + aj[j][0]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f16x8 wf = wfrag(t);
#pragma unroll
                    for (int j = 0; j < PXW; ++j)
                        acc[t][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xa[j], acc[t][0][j], 0, 0, 0);
                }
                // offset (0,1): taps 4,5 -> phases 1,3
#pragma unroll
                for (int j = 0; j < PXW; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ap + aj[j][G == 4 ? 1 : 0]);
                {
                    const f16x8 w4 = wfrag(4), w5 = wfrag(5);
#pragma unroll
                    for (int j = 0; j < PXW; ++j) {
                        acc[1][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4, xa[j], acc[1][0][j], 0, 0, 0);
                        acc[3][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w5, xa[j], acc[3][0][j], 0, 0, 0);
                    }
                }
                // offset (1,0): taps 6,7 -> phases 2,3
#pragma unroll
                for (int j = 0; j < PXW; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ap + rowB + arow(aj[j][0], 1));
                {
                    const f16x8 w6 = wfrag(6), w7 = wfrag(7);
#pragma unroll
                    for (int j = 0; j < PXW; ++j) {
                        acc[2][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w6, xa[j], acc[2][0][j], 0, 0, 0);
                        acc[3][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w7, xa[j], acc[3][0][j], 0, 0, 0);
                    }
                }
                // offset (1,1): tap 8 -> phase 3
#pragma unroll
                for (int j = 0; j < PXW; ++j) xa[j] = *reinterpret_cast<const f16x8*>(Ap + rowB + arow(aj[j][G == 4 ? 1 : 0], 1));
                {
                    const f16x8 w8 = wfrag(8);
#pragma unroll
                    for (int j = 0; j < PXW; ++j)
                        acc[3][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8, xa[j], acc[3][0][j], 0, 0, 0);
                }
            }
        }
    };

    __syncthreads();                       // zero fill done before any DMA lands
    stage(c_begin, 0);
    // folded-BN scale/shift of this block's BN output channels -> LDS by DMA, behind the first chunk (a register
    // round trip here would put one global-load latency in front of every block's first DMA)
    // (1x1 layers read them from global in the epilogue instead: their two 40-KiB stage pairs are exactly half of
    // the CU's LDS and 512 more bytes would cost the second resident block)
    if (T != 1 && wave < 2 && lane < BN / 4)
        GLDS16((wave ? a.shift : a.scale) + ntile * BN + lane * 4, smem + a.lds_scale_off + wave * (BN * 4));
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        __syncthreads();                   // vmcnt(0): chunk c landed; every wave left stage cur^1
        if (c + 1 < c_end) stage(c + 1, cur ^ 1);
        if (!ABL(a, 4)) compute(cur);
    }

#if LTK_ABLATE_BUILD
    if (a.ablate & 64) return;             // measurement: no epilogue at all
    if (a.ablate & 128) {                  // measurement: epilogue arithmetic only keeps acc alive
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < NBT; ++i)
#pragma unroll
                for (int j = 0; j < PXW; ++j) t += acc[g][i][j][0] + acc[g][i][j][7];
        if (t == 12345.678f) a.y[0] = (f16)t;
        return;
    }
#endif
    // ---- epilogue
    const int cout0 = ntile * BN;
    const int HWo = a.HoA * a.WoA;
    // output pixel of this lane in subtile j / phase g: image n and pixel index inside the output plane
    int l31_e = l31;                       // same reason as lane_i: keep the epilogue's per-lane pixel arithmetic in the epilogue
    asm volatile("" : "+v"(l31_e));
    auto out_px = [&](int j, int g, int* n_out, bool* ok) -> int {
        const int m = (wave * PXW + j) * 32 + l31_e;
        const int tx = m & TWm;
        const int ty = (m >> a.log2TW) & THm;
        const int b = m >> (a.log2TW + a.log2TH);
        const int n = n0 + b, y = ty0 + ty, x = tx0 + tx;
        *ok = (b < a.NB) && (n < a.N) && (y < a.Ho) && (x < a.Wo);
        *n_out = n;
        const int oy = (G == 4) ? 2 * y + (g >> 1) : y;
        const int ox = (G == 4) ? 2 * x + (g & 1) : x;
        return oy * a.WoA + ox;
    };

    if constexpr (HEAD) {
        // lane (l31, hh) holds channels 8*q4 + 4*hh + r (q4, r = 0..3) of pixel l31 of each subtile: 16 of the 32 channels;
        // the partner lane (l31, hh^1) holds the other 16
        const float* const sb = reinterpret_cast<const float*>(smem + a.lds_scale_off);   // [2][32] staged in the prologue
        float p[PXW][3];
#pragma unroll
        for (int j = 0; j < PXW; ++j) p[j][0] = p[j][1] = p[j][2] = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {         // four channels at a time: the 80 scale/shift/weight values never live at once
            const int cl = 8 * q4 + 4 * hh;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sb + cl), sf = *reinterpret_cast<const f32x4*>(sb + 32 + cl);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(hd->w + cl), w1 = *reinterpret_cast<const f32x4*>(hd->w + 32 + cl),
                        w2 = *reinterpret_cast<const f32x4*>(hd->w + 64 + cl);
#pragma unroll
            for (int j = 0; j < PXW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = __builtin_amdgcn_fmed3f(acc[0][0][j][4 * q4 + r] * sc[r] + sf[r], 0.f, 65504.f);   // BN + ReLU (conv.py:12-19)
                    p[j][0] += v * w0[r]; p[j][1] += v * w1[r]; p[j][2] += v * w2[r];
                }
        }
        const float b0 = hd->w[96], b1 = hd->w[97], b2 = hd->w[98];
#pragma unroll
        for (int j = 0; j < PXW; ++j) {
            bool ok;
            int n;
            const int opx = out_px(j, 0, &n, &ok);
            const float t0 = p[j][0] + __shfl_xor(p[j][0], 32), t1 = p[j][1] + __shfl_xor(p[j][1], 32),
                        t2 = p[j][2] + __shfl_xor(p[j][2], 32);
            const float s0 = 1.f / (1.f + __expf(-(t0 + b0)));
            const float s1 = 1.f / (1.f + __expf(-(t1 + b1)));
            const float s2 = 1.f / (1.f + __expf(-(t2 + b2)));
            unsigned char* const o = ok ? hd->outs->p[n] : nullptr;
            if (o && hh == 0) {      // float32 * 255 then truncation toward zero, as numpy astype(uint8) on [0,255]
                unsigned char* q = o + (size_t)opx * 3;
                q[0] = (unsigned char)(unsigned)(s0 * 255.f);
                q[1] = (unsigned char)(unsigned)(s1 * 255.f);
                q[2] = (unsigned char)(unsigned)(s2 * 255.f);
            }
        }
        return;
    }
    if (a.partial) {
        // split-K: raw fp32 partial sums, slab ks; lane holds 4 consecutive couts per register group
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                bool ok;
                int n;
                const int opx = out_px(j, g, &n, &ok);
                if (!ok) continue;
                float* dst = a.partial + ((size_t)ks * a.Mtot + (size_t)n * HWo + opx) * a.CoutP + cout0;
#pragma unroll
                for (int i = 0; i < NBT; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int cl = i * 32 + 8 * q4 + 4 * hh;
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[g][i][j][4 * q4 + r];
                        *reinterpret_cast<f32x4*>(dst + cl) = v;
                    }
            }
        return;
    }

    // y = relu(acc*scale + shift + res) -> fp16, stored straight from the MFMA accumulator layout: a lane holds
    // 4 consecutive output channels of one pixel per register group; v_permlane32_swap pairs the two lane
    // halves so that every lane owns 8 consecutive channels (16 B) of its pixel, lanes 0-31 the first half of a
    // channel block and lanes 32-63 the second: one wave store = 32 pixels x 32 B = one contiguous KiB of the
    // channel-blocked output.  No LDS round trip (the ds_write_b64 transposes cost more than the MFMAs on the
    // 64-channel layers).
    const int ncb_valid = min(BN / 16, (a.Cout - cout0) >> 4);
    const bool has_res = a.res != nullptr && !ABL(a, 8);
    const bool do_store = !ABL(a, 16);
    const int cbo = cout0 >> 4;
    const int HWo16 = HWo * 16;
    const float* const sbase = reinterpret_cast<const float*>(smem + a.lds_scale_off);   // [2][BN] staged in the prologue

    // the activation is a compile-time parameter of the epilogue body (one wave-uniform branch selects the copy):
    // as a per-value runtime test hipcc if-converted it and every value paid for erff and exp
    // LayerNorm fold, consumer side: mean / rstd of this lane's token in subtile j from the producer's per-tile partial sums (the two
    // lanes of a pixel read alternate tiles, fixed order: deterministic)
    auto ln_token_stats = [&](int tok, float* mean, float* rstd) {
        const float2* pp = reinterpret_cast<const float2*>(a.ln_in) + (size_t)tok * a.ln_in_tiles;
        float su = 0.f, sq = 0.f;
        for (int t = hh; t < a.ln_in_tiles; t += 2) { const float2 v = pp[t]; su += v.x; sq += v.y; }
        su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
        const float invC = 1.f / (float)(a.ln_in_tiles * 32);
        const float m = su * invC;
        *mean = m;
        *rstd = rsqrtf(fmaxf(sq * invC - m * m, 0.f) + a.ln_eps);
    };
    auto epilogue = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        // Loop order: (phase, cout block) outside, the wave's PXW pixel subtiles inside.  The folded-BN scale/shift of a cout
        // block is read ONCE (4 ds_read_b128) and serves all subtiles; with the subtile loop outside every 16-byte store
        // waited for its own LDS round trip (64 reads per item, each followed by s_waitcnt lgkmcnt(0)): on the
        // 64-channel layers, whose K loop is only 4 chunks, the epilogue was a third of the item.
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int obase[PXW], rbase[PXW];
            bool okj[PXW];
            int tok[PXW];                               // T == 1, LayerNorm fold: token index (image, pixel) of this lane's pixel
            float lmean[PXW], lrstd[PXW];
            const bool ln_cons = T == 1 && a.ln_in != nullptr, ln_prod = T == 1 && a.ln_out != nullptr;
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                int n;
                const int opx = out_px(j, g, &n, &okj[j]);
                obase[j] = ((n * a.y_cbt + a.y_cb0 + cbo) * HWo + opx) * 16 + hh * 8;
                rbase[j] = ((n * a.res_cbt + a.res_cb0 + cbo) * HWo + opx) * 16 + hh * 4;
                if constexpr (T == 1) {
                    tok[j] = okj[j] ? n * HWo + opx : 0;
                    lmean[j] = 0.f; lrstd[j] = 1.f;
                    if (ln_cons) ln_token_stats(tok[j], &lmean[j], &lrstd[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                float ls[PXW], lq[PXW];                 // producer side: sums of this lane's 16 stored values of tile i
#pragma unroll
                for (int j = 0; j < PXW; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {        // channel block 2i+pr of this block's BN
                    if ((2 * i + pr) >= ncb_valid) continue;         // wave-uniform
                    f32x4 sc[2], sf[2];
#pragma unroll
                    for (int eo = 0; eo < 2; ++eo) {    // q4 = 2pr+eo: channels 8*q4 + 4*hh .. +3 of the 32-cout tile i
                        const int cl = i * 32 + 8 * (2 * pr + eo) + 4 * hh;
                        if constexpr (T == 1) {
                            sc[eo] = *reinterpret_cast<const f32x4*>(a.scale + cout0 + cl);
                            sf[eo] = *reinterpret_cast<const f32x4*>(a.shift + cout0 + cl);
                        } else {
                            sc[eo] = *reinterpret_cast<const f32x4*>(sbase + cl);
                            sf[eo] = *reinterpret_cast<const f32x4*>(sbase + BN + cl);
                        }
                    }
                    f16x4 rr[PXW][2];
                    if (has_res) {
#pragma unroll
                        for (int j = 0; j < PXW; ++j)
#pragma unroll
                            for (int eo = 0; eo < 2; ++eo)
                                if (okj[j]) rr[j][eo] = *reinterpret_cast<const f16x4*>(a.res + rbase[j] + (2 * i + pr) * HWo16 + eo * 8);
                    }
#pragma unroll
                    for (int j = 0; j < PXW; ++j) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int eo = 0; eo < 2; ++eo) {
                            const int q4 = 2 * pr + eo;
                            float v[4];
                            if constexpr (T == 1) {
                                if (ln_cons) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = lrstd[j] * (acc[g][i][j][4 * q4 + r] - lmean[j] * sc[eo][r]) + sf[eo][r];
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] = acc[g][i][j][4 * q4 + r] * sc[eo][r] + sf[eo][r];
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = acc[g][i][j][4 * q4 + r] * sc[eo][r] + sf[eo][r];
                            }
                            if (has_res && okj[j]) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += (float)rr[j][eo][r];
                            }
                            f16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float t = v[r];
                                if constexpr (ACT == 1) {
                                    t = __builtin_amdgcn_fmed3f(t, 0.f, 65504.f);          // ReLU and the fp16 range in one op
                                } else {
                                    if constexpr (ACT == 2) t = 0.5f * t * (1.f + erff(t * 0.70710678118654752f));
                                    else if constexpr (ACT == 3) t = t / (1.f + __expf(-t));
                                    t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
                                }
                                o[r] = (f16)t;
                            }
                            if constexpr (T == 1) {
                                if (ln_prod) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; ls[j] += f; lq[j] += f * f; }
                                }
                            }
                            const uint2 u = *reinterpret_cast<const uint2*>(&o);
                            pk[eo][0] = u.x; pk[eo][1] = u.y;
                        }
                        // lanes 0-31 keep eo=0 (channels 0-3) and receive the partner's eo=0 (channels 4-7);
                        // lanes 32-63 end with the eo=1 pair
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                        const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        if (okj[j] && do_store) *reinterpret_cast<uint4*>(a.y + obase[j] + (2 * i + pr) * HWo16) = out;
                    }
                }
                if constexpr (T == 1) {
                    if (ln_prod && 2 * i < ncb_valid) {
#pragma unroll
                        for (int j = 0; j < PXW; ++j) {
                            const float su = ls[j] + __shfl_xor(ls[j], 32), sq = lq[j] + __shfl_xor(lq[j], 32);
                            if (okj[j] && hh == 0)
                                reinterpret_cast<float2*>(a.ln_out)[(size_t)tok[j] * a.ln_out_tiles + (cout0 >> 5) + i] = make_float2(su, sq);
                        }
                    }
                }
            }
        }
    };
    // GEGLU (relu == 4; diffusers FeedForward.net[0] = GEGLU: proj -> value * gelu(gate)): the host packs the projection's rows so
    // that every 32-cout tile is [16 value channels | their 16 gate channels]; a lane holds value channel 8*eo + 4*hh + r in
    // register group eo and its gate in group 2 + eo, so the product needs no exchange and tile i becomes ONE 16-channel block
    // (cout0 / 32 + i) of the half-as-wide output.  1x1 layers only (scale / shift from global memory), no residual.
    if constexpr (T == 1 && G == 1 && Q == 0) {
        if (a.relu == 4) {
            const int cbo2 = cout0 >> 5;
            int obase[PXW];
            bool okj[PXW];
            float lmean[PXW], lrstd[PXW];
            const bool ln_cons = a.ln_in != nullptr;
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                int n;
                const int opx = out_px(j, 0, &n, &okj[j]);
                obase[j] = ((n * a.y_cbt + a.y_cb0 + cbo2) * HWo + opx) * 16 + hh * 8;
                lmean[j] = 0.f; lrstd[j] = 1.f;
                if (ln_cons) ln_token_stats(okj[j] ? n * HWo + opx : 0, &lmean[j], &lrstd[j]);
            }
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                if (2 * i >= ncb_valid) continue;                    // wave-uniform
                f32x4 scv[2], sfv[2], scg[2], sfg[2];
#pragma unroll
                for (int eo = 0; eo < 2; ++eo) {
                    const int cl = cout0 + i * 32 + 8 * eo + 4 * hh;
                    scv[eo] = *reinterpret_cast<const f32x4*>(a.scale + cl);      sfv[eo] = *reinterpret_cast<const f32x4*>(a.shift + cl);
                    scg[eo] = *reinterpret_cast<const f32x4*>(a.scale + cl + 16); sfg[eo] = *reinterpret_cast<const f32x4*>(a.shift + cl + 16);
                }
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int eo = 0; eo < 2; ++eo) {
                        f16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v, gt;
                            if (ln_cons) {
                                v = lrstd[j] * (acc[0][i][j][4 * eo + r] - lmean[j] * scv[eo][r]) + sfv[eo][r];
                                gt = lrstd[j] * (acc[0][i][j][4 * (2 + eo) + r] - lmean[j] * scg[eo][r]) + sfg[eo][r];
                            } else {
                                v = acc[0][i][j][4 * eo + r] * scv[eo][r] + sfv[eo][r];
                                gt = acc[0][i][j][4 * (2 + eo) + r] * scg[eo][r] + sfg[eo][r];
                            }
                            const float t = v * gelu_as(gt);
                            o[r] = (f16)__builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
                        }
                        const uint2 u = *reinterpret_cast<const uint2*>(&o);
                        pk[eo][0] = u.x; pk[eo][1] = u.y;
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    if (okj[j] && do_store) *reinterpret_cast<uint4*>(a.y + obase[j] + i * HWo16) = out;
                }
            }
            return;
        }
    }
    if (a.relu == 1) epilogue(std::integral_constant<int, 1>{});
    else if (a.relu == 2) epilogue(std::integral_constant<int, 2>{});
    else if (a.relu == 3) epilogue(std::integral_constant<int, 3>{});
    else epilogue(std::integral_constant<int, 0>{});
}

// Persistent launch: the grid is at most `2 x CUs` blocks (what is resident at once) and every block walks the item
// list with stride gridDim.x.  With thousands of items per layer (64/128-channel layers on 128^2..256^2 maps) the
// per-block dispatch, kernarg fetch and descriptor set-up were a fixed ~16 us per launch; a non-persistent launch
// (gridDim.x == nitems) is the same code with one trip.  Logical ids are XCD-contiguous: consecutive ids (same
// pixel tile, different cout tiles) run on one XCD and share its L2.
template <int G, int NBT, int PXW, int NC8, int T, int S = 1, int Q = 0>
__global__ __launch_bounds__(256, 2) void conv3_kernel(const K3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    for (int item = first; item < a.nitems; item += nblk) {
        if (item != first) __syncthreads();      // every wave is out of the previous item's LDS stages
        conv3_item<G, NBT, PXW, NC8, T, S, Q>(a, item, smem);
    }
}

template <int PXW>
__global__ __launch_bounds__(256, 2) void conv3_head_kernel(const K3Args a, const HeadArgs h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    for (int item = first; item < a.nitems; item += nblk) {
        if (item != first) __syncthreads();
        conv3_item<1, 1, PXW, 2, 9, 1, 0, 1>(a, item, smem, &h);
    }
}

// split-K finish: y = relu((sum_s partial[s]) * scale + shift + res) -> fp16.  One thread = one pixel x 8 couts;
// consecutive threads = the two halves of a channel block, then consecutive pixels.
__global__ __launch_bounds__(256) void conv3_finish_kernel(const float* __restrict__ partial, int ksplit, long long Mtot, int HWo,
                                                            int CoutP, int Cout, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const f16* __restrict__ res,
                                                            int res_cbt, int res_cb0, f16* __restrict__ y, int y_cbt, int y_cb0,
                                                            int relu) {
    const int ncb = Cout >> 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Mtot * ncb * 2) return;
    const int half = (int)(i & 1);
    const long long r2 = i >> 1;
    const long long row = r2 % Mtot;
    const int cb = (int)(r2 / Mtot);
    const int n = (int)(row / HWo), opx = (int)(row - (long long)n * HWo);
    const int c0 = cb * 16 + half * 8;
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float* p = partial + ((size_t)s * Mtot + row) * CoutP + c0;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += a0[r]; v[4 + r] += a1[r]; }
    }
    f16x8 rr;
    if (res) rr = *reinterpret_cast<const f16x8*>(res + (((size_t)n * res_cbt + res_cb0 + cb) * HWo + opx) * 16 + half * 8);
    f16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float t = v[r] * scale[c0 + r] + shift[c0 + r];
        if (res) t += (float)rr[r];
        if (relu == 1) t = fmaxf(t, 0.f);
        else if (relu >= 2) t = (relu == 2) ? 0.5f * t * (1.f + erff(t * 0.70710678118654752f)) : t / (1.f + __expf(-t));
        t = fminf(fmaxf(t, -65504.f), 65504.f);
        o[r] = (f16)t;
    }
    *reinterpret_cast<f16x8*>(y + (((size_t)n * y_cbt + y_cb0 + cb) * HWo + opx) * 16 + half * 8) = o;
}


// ------------------------------------------------------------------------------------------
// lin_fk_kernel: the short-K linear / 1x1 layers of MuseTalk's transformer blocks (K = 320 / 640 on the 32^2 / 16^2 levels; round 6).
// As conv3 1x1 launches these layers are latency chains: a 256-pixel x 64-cout item walks its 5-10 channel chunks one LDS-DMA
// round trip at a time (16 MFMAs per wave behind each), pays a zero fill, a first round trip and an epilogue per item, and a
// 3.4-GFLOP projection takes 22-28 us whatever the level.  Here the roles are turned round for a K that fits the register file:
//   * a wave keeps the WHOLE K of its 32 pixels in registers as MFMA B-operand fragments (KB x 16 B per lane, straight from
//     global memory: one contiguous KiB per channel block and wave, no LDS image, no zero fill);
//   * the block streams the weights through LDS in FULL-K slabs of 32 output channels (KB KiB, contiguous in the packed
//     weights: [cout/32][K/8][32][8]), two stages, one barrier per slab; the four waves read the same slab (one ds_read_b128
//     per MFMA);
//   * every slab ends in the epilogue of its 32 x 128 outputs (conv3's 1x1 epilogue: bias / folded LayerNorm consumer and
//     producer sides / residual / activation / GEGLU), whose global loads and stores overlap the next slab's DMA.
// A block = 128 consecutive pixels (tokens) x a group of `cpg` slabs; blocks of one pixel tile are adjacent (its A rows come from L2).
// Summation order per output = conv3's unsplit order (channel blocks ascending): the two kernels agree bit for bit.
// KS = 2 (K = 1280: the 8^2 level's projections, ff.net.2 of the 32^2 level): 1280 channels are 320 registers per lane, so TWO waves share a
// pixel subtile - wave w takes the subtile w & 3 and the K half w >> 2 (KB = 40 channel blocks each), a stage holds one 20-KiB part of the
// slab for either half, the upper half's 16 accumulators meet the lower half's in LDS behind the slab's last part (one more barrier per slab;
// exchange area double-buffered by slab parity) and the lower-half waves run the epilogue.  acc = (half 0) + (half 1): fixed order,
// deterministic, not conv3's sequential order (fp32 rounding only).
template <int KB, int KPB, int KS>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void lin_fk_kernel(const K3Args a, const int cpg, const int ngroups, const int nslabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, kh = wave8 >> 2;         // pixel subtile, K half (KS = 1: kh = 0)
    const int l31 = lane & 31, hh = lane >> 5;
    const int grp = blockIdx.x % ngroups, mt = blockIdx.x / ngroups;
    const int s_begin = grp * cpg, s_end = min(nslabs, s_begin + cpg);
    const int P = a.HoA * a.WoA;
    constexpr int SLAB = KB * KS * 1024;                 // bytes of one 32-cout slab
    constexpr int PXW = 1;                               // 32-pixel subtiles per wave (2 was built: 116-164 bytes of scratch per lane, slower)
    constexpr int KP = KB / KPB;                         // a slab goes through LDS in KP parts of KPB channel blocks (K = 640: two 20-KiB parts;
    constexpr int PART = KPB * 1024;                     // its 80-KiB stage pairs left ONE block per CU and nothing for the scale / shift image)
    constexpr int STAGE = KS * PART;                     // a stage: part h of the slab for every K half
    constexpr int SS_OFF = 2 * STAGE;                    // [2][scale 32 | shift 32] fp32 behind the two weight stages
    constexpr int XCH_OFF = SS_OFF + 512;                // KS = 2: [2 slab parities][4 subtiles][4][64 lanes][16 B] accumulators of the upper K half
    static_assert(KB % KPB == 0 && KPB % 4 == 0 && (KS == 1 || KS == 2), "whole parts, whole KiB pieces per wave");
    bool ok[PXW];
    int n[PXW], pix[PXW];
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        const long long m = (long long)mt * (128 * PXW) + (wave * PXW + j) * 32 + l31;
        ok[j] = m < a.Mtot;
        const int mm = (int)(ok[j] ? m : a.Mtot - 1);
        n[j] = mm / P; pix[j] = mm - n[j] * P;
    }

    // ---- weights: slab s -> stage buf (KB pieces of 1 KiB, piece k*4 + wave by this wave), its scale / shift behind them
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.w);
    auto stage = [&](int s, int h, int buf) {
        // (K half kh's part h sits (kh * KP + h) parts into the slab; every wave copies KPB / 4 KiB pieces of its own half's part)
        const unsigned char* src = wbase + (size_t)s * SLAB + (kh * KP + h) * PART + lane * 16;
        unsigned char* dst = smem + buf * STAGE + kh * PART;
#pragma unroll
        for (int k = 0; k < KPB / 4; ++k) GLDS16(src + (k * 4 + wave) * 1024, dst + (k * 4 + wave) * 1024);
        if (h == 0 && wave8 == 0 && lane < 16)
            GLDS16((lane < 8 ? a.scale : a.shift - 32) + s * 32 + lane * 4, smem + SS_OFF + ((s - s_begin) & 1) * 256);
    };
    stage(s_begin, 0, 0);
    int parts_done = 0;

    // ---- this lane's pixels: 8 channels (half hh) of every channel block
    f16x8 af[PXW][KB];
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        const f16* xb = a.x + ((size_t)(n[j] * a.x_cbt + a.x_cb0 + kh * KB) * P + pix[j]) * 16 + hh * 8;
#pragma unroll
        for (int q = 0; q < KB; ++q) af[j][q] = *reinterpret_cast<const f16x8*>(xb + (size_t)q * P * 16);
    }

    // LayerNorm fold, consumer side: mean / rstd of this lane's tokens (once per block)
    int tok[PXW];
    float lmean[PXW], lrstd[PXW];
    const bool ln_cons = a.ln_in != nullptr, ln_prod = a.ln_out != nullptr;
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        tok[j] = n[j] * P + pix[j];
        lmean[j] = 0.f; lrstd[j] = 1.f;
        if (ln_cons && kh == 0) {
            const float2* pp = reinterpret_cast<const float2*>(a.ln_in) + (size_t)tok[j] * a.ln_in_tiles;
            float su = 0.f, sq = 0.f;
            for (int t = hh; t < a.ln_in_tiles; t += 2) { const float2 v = pp[t]; su += v.x; sq += v.y; }
            su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
            const float invC = 1.f / (float)(a.ln_in_tiles * 32);
            lmean[j] = su * invC;
            lrstd[j] = rsqrtf(fmaxf(sq * invC - lmean[j] * lmean[j], 0.f) + a.ln_eps);
        }
    }
    const bool has_res = a.res != nullptr;

    for (int s = s_begin; s < s_end; ++s) {
        const int cout0 = s * 32;
        const int ncb_valid = min(2, (a.Cout - cout0) >> 4);
        // residual values of this slab's outputs: requested in front of the MFMAs, used behind them
        // (32 pixels per wave only: the 64-pixel variant has no registers to spare and serves the wide projections, which have no residual)
        constexpr bool RES_AHEAD = PXW == 1;
        f16x4 rr[PXW][2][2];
        if (RES_AHEAD && has_res && kh == 0) {
#pragma unroll
            for (int j = 0; j < PXW; ++j)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int eo = 0; eo < 2; ++eo)
                        if (pr < ncb_valid)
                            rr[j][pr][eo] = *reinterpret_cast<const f16x4*>(a.res + ((size_t)(n[j] * a.res_cbt + a.res_cb0 + (cout0 >> 4) + pr) * P + pix[j]) * 16 + hh * 4 + eo * 8);
        }
        f32x16 acc[PXW];
#pragma unroll
        for (int j = 0; j < PXW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int h = 0; h < KP; ++h) {
            const int cur = parts_done & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies of the part landed (explicit: not left to hipcc's barrier lowering)
            __syncthreads();               // ... every wave's did; every wave left stage cur^1
            if (h + 1 < KP) stage(s, h + 1, cur ^ 1);
            else if (s + 1 < s_end) stage(s + 1, 0, cur ^ 1);
            ++parts_done;
            const unsigned char* Bs = smem + cur * STAGE + kh * PART + (hh * 32 + l31) * 16;
            // weight fragments WD channel blocks ahead of their MFMAs
            constexpr int WD = PXW == 1 ? 4 : 2;
            f16x8 wf[WD];
#pragma unroll
            for (int q = 0; q < WD; ++q) wf[q] = *reinterpret_cast<const f16x8*>(Bs + q * 1024);
#pragma unroll
            for (int q = 0; q < KPB; ++q) {
#pragma unroll
                for (int j = 0; j < PXW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[q % WD], af[j][h * KPB + q], acc[j], 0, 0, 0);
                if (q + WD < KPB) wf[q % WD] = *reinterpret_cast<const f16x8*>(Bs + (q + WD) * 1024);
            }
        }

        if constexpr (KS == 2) {           // the two K halves of a subtile meet in LDS; the lower half's wave goes on to the epilogue
            f32x4* const xw = reinterpret_cast<f32x4*>(smem + XCH_OFF + (((s - s_begin) & 1) * 4 + wave) * 4096) + lane;
            if (kh == 1) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) xw[g4 * 64] = (f32x4){acc[0][4 * g4], acc[0][4 * g4 + 1], acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]};
            }
            __syncthreads();
            if (kh == 1) continue;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 o = xw[g4 * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[0][4 * g4 + r] += o[r];
            }
        }
        // ---- epilogue of the 32 output channels of slab s (conv3_item's 1x1 epilogue with NBT = 1)
        const float* const ssb = reinterpret_cast<const float*>(smem + SS_OFF + ((s - s_begin) & 1) * 256);      // [scale 32 | shift 32]
        if (a.relu == 4) {                 // GEGLU: the slab is [16 value | 16 gate] channels -> one 16-channel block of the output
            if (ncb_valid <= 0) continue;
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                unsigned pk[2][2];
#pragma unroll
                for (int eo = 0; eo < 2; ++eo) {
                    // (scale / shift re-read from LDS per use: four ds_read_b128 instead of 32 registers held across the pixel subtiles)
                    const int cl = 8 * eo + 4 * hh;
                    const f32x4 scv = *reinterpret_cast<const f32x4*>(ssb + cl), sfv = *reinterpret_cast<const f32x4*>(ssb + 32 + cl);
                    const f32x4 scg = *reinterpret_cast<const f32x4*>(ssb + cl + 16), sfg = *reinterpret_cast<const f32x4*>(ssb + 32 + cl + 16);
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v, gt;
                        if (ln_cons) {
                            v = lrstd[j] * (acc[j][4 * eo + r] - lmean[j] * scv[r]) + sfv[r];
                            gt = lrstd[j] * (acc[j][4 * (2 + eo) + r] - lmean[j] * scg[r]) + sfg[r];
                        } else {
                            v = acc[j][4 * eo + r] * scv[r] + sfv[r];
                            gt = acc[j][4 * (2 + eo) + r] * scg[r] + sfg[r];
                        }
                        const float t = v * gelu_as(gt);
                        o[r] = (f16)__builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
                    }
                    const uint2 u = *reinterpret_cast<const uint2*>(&o);
                    pk[eo][0] = u.x; pk[eo][1] = u.y;
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                if (ok[j]) *reinterpret_cast<uint4*>(a.y + ((size_t)(n[j] * a.y_cbt + a.y_cb0 + (cout0 >> 5)) * P + pix[j]) * 16 + hh * 8) = out;
            }
            continue;
        }
        auto epilogue = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            const int cbo = cout0 >> 4;
            float ls[PXW], lq[PXW];
#pragma unroll
            for (int j = 0; j < PXW; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (pr >= ncb_valid) continue;           // wave-uniform
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int eo = 0; eo < 2; ++eo) {
                        const int q4 = 2 * pr + eo;
                        const int cl = 8 * q4 + 4 * hh;
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ssb + cl), sf = *reinterpret_cast<const f32x4*>(ssb + 32 + cl);
                        float v[4];
                        if (ln_cons) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = lrstd[j] * (acc[j][4 * q4 + r] - lmean[j] * sc[r]) + sf[r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = acc[j][4 * q4 + r] * sc[r] + sf[r];
                        }
                        if (has_res) {
                            if constexpr (!RES_AHEAD)
                                rr[j][pr][eo] = *reinterpret_cast<const f16x4*>(a.res + ((size_t)(n[j] * a.res_cbt + a.res_cb0 + cbo + pr) * P + pix[j]) * 16 + hh * 4 + eo * 8);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)rr[j][pr][eo][r];
                        }
                        f16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t = v[r];
                            if constexpr (ACT == 1) t = __builtin_amdgcn_fmed3f(t, 0.f, 65504.f);
                            else t = __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
                            o[r] = (f16)t;
                        }
                        if (ln_prod) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; ls[j] += f; lq[j] += f * f; }
                        }
                        const uint2 u = *reinterpret_cast<const uint2*>(&o);
                        pk[eo][0] = u.x; pk[eo][1] = u.y;
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    if (ok[j]) *reinterpret_cast<uint4*>(a.y + ((size_t)(n[j] * a.y_cbt + a.y_cb0 + cbo + pr) * P + pix[j]) * 16 + hh * 8) = out;
                }
            }
            if (ln_prod && ncb_valid > 0) {
#pragma unroll
                for (int j = 0; j < PXW; ++j) {
                    const float su = ls[j] + __shfl_xor(ls[j], 32), sq = lq[j] + __shfl_xor(lq[j], 32);
                    if (ok[j] && hh == 0) reinterpret_cast<float2*>(a.ln_out)[(size_t)tok[j] * a.ln_out_tiles + s] = make_float2(su, sq);
                }
            }
        };
        if (a.relu == 1) epilogue(std::integral_constant<int, 1>{});       // (GELU / SiLU epilogues stay on conv3: no linear layer of the path has one)
        else epilogue(std::integral_constant<int, 0>{});
    }
}


// One slab's epilogue for one wave's 32 pixels (lin_fk_kernel's, as a function for lin_mp_kernel): 32 x 32 accumulators -> y
__device__ __forceinline__ void lin_slab_epilogue(const K3Args& a, const f32x16& acc, const int s, const float* const ssb, const int n, const int pix,
                                                   const bool ok, const int tok, const float lmean, const float lrstd, const int hh, const int P) {
    const int cout0 = s * 32;
    const int ncb_valid = min(2, (a.Cout - cout0) >> 4);
    const bool ln_cons = a.ln_in != nullptr, ln_prod = a.ln_out != nullptr, has_res = a.res != nullptr;
    if (ncb_valid <= 0) return;
    if (a.relu == 4) {
        unsigned pk[2][2];
#pragma unroll
        for (int eo = 0; eo < 2; ++eo) {
            const int cl = 8 * eo + 4 * hh;
            const f32x4 scv = *reinterpret_cast<const f32x4*>(ssb + cl), sfv = *reinterpret_cast<const f32x4*>(ssb + 32 + cl);
            const f32x4 scg = *reinterpret_cast<const f32x4*>(ssb + cl + 16), sfg = *reinterpret_cast<const f32x4*>(ssb + 32 + cl + 16);
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v, gt;
                if (ln_cons) {
                    v = lrstd * (acc[4 * eo + r] - lmean * scv[r]) + sfv[r];
                    gt = lrstd * (acc[4 * (2 + eo) + r] - lmean * scg[r]) + sfg[r];
                } else {
                    v = acc[4 * eo + r] * scv[r] + sfv[r];
                    gt = acc[4 * (2 + eo) + r] * scg[r] + sfg[r];
                }
                const float t = v * gelu_as(gt);
                o[r] = (f16)__builtin_amdgcn_fmed3f(t, -65504.f, 65504.f);
            }
            const uint2 u = *reinterpret_cast<const uint2*>(&o);
            pk[eo][0] = u.x; pk[eo][1] = u.y;
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if (ok) *reinterpret_cast<uint4*>(a.y + ((size_t)(n * a.y_cbt + a.y_cb0 + (cout0 >> 5)) * P + pix) * 16 + hh * 8) = out;
        return;
    }
    const int cbo = cout0 >> 4;
    const float lo = a.relu == 1 ? 0.f : -65504.f;
    float ls = 0.f, lq = 0.f;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        if (pr >= ncb_valid) continue;           // wave-uniform
        unsigned pk[2][2];
#pragma unroll
        for (int eo = 0; eo < 2; ++eo) {
            const int q4 = 2 * pr + eo;
            const int cl = 8 * q4 + 4 * hh;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ssb + cl), sf = *reinterpret_cast<const f32x4*>(ssb + 32 + cl);
            float v[4];
            if (ln_cons) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = lrstd * (acc[4 * q4 + r] - lmean * sc[r]) + sf[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[4 * q4 + r] * sc[r] + sf[r];
            }
            if (has_res) {
                const f16x4 rr = *reinterpret_cast<const f16x4*>(a.res + ((size_t)(n * a.res_cbt + a.res_cb0 + cbo + pr) * P + pix) * 16 + hh * 4 + eo * 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            }
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)__builtin_amdgcn_fmed3f(v[r], lo, 65504.f);
            if (ln_prod) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; ls += f; lq += f * f; }
            }
            const uint2 u = *reinterpret_cast<const uint2*>(&o);
            pk[eo][0] = u.x; pk[eo][1] = u.y;
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if (ok) *reinterpret_cast<uint4*>(a.y + ((size_t)(n * a.y_cbt + a.y_cb0 + cbo + pr) * P + pix) * 16 + hh * 8) = out;
    }
    if (ln_prod) {
        const float su = ls + __shfl_xor(ls, 32), sq = lq + __shfl_xor(lq, 32);
        if (ok && hh == 0) reinterpret_cast<float2*>(a.ln_out)[(size_t)tok * a.ln_out_tiles + s] = make_float2(su, sq);
    }
}

// lin_mp_kernel<NSL>: lin_fk's K = 1280 arrangement (8 waves: pixel subtile w & 3, K half w >> 2, 640 channels = 160 registers per wave) for
// K = 2560 / 5120 (ff.net.2 of the 16^2 / 8^2 levels, the 2560-channel shortcuts): the K loop runs in `npass` passes of 1280 channels - a
// wave reloads its A fragments per pass - and the accumulators of the block's NSL slabs stay in registers across the passes (NSL x 16), so
// a block is 128 pixels x NSL x 32 output channels over the whole K.  Weight stream: (pass, slab, part) in that order, 20-KiB parts per K
// half, two stages; after the last pass the upper half's accumulators meet the lower half's in LDS and the lower-half waves run the NSL
// epilogues.  A group's missing slabs (nslabs not a multiple of NSL) are clamped to the last slab: computed, not stored.
template <int NSL>
__global__ __launch_bounds__(512, 1) void lin_mp_kernel(const K3Args a, const int ngroups, const int nslabs, const int npass) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KB = 40, KPB = 20, KP = 2, PART = KPB * 1024, STAGE = 2 * PART;
    constexpr int SS_OFF = 2 * STAGE, XCH_OFF = SS_OFF + 1024;        // [NSL][scale 32 | shift 32] (NSL <= 4); [NSL][4 subtiles][4][64 lanes][16 B]
    static_assert(NSL >= 1 && NSL <= 3, "accumulator budget (4 slabs spill)");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, kh = wave8 >> 2;
    const int l31 = lane & 31, hh = lane >> 5;
    const int grp = blockIdx.x % ngroups, mt = blockIdx.x / ngroups;
    const int s0 = grp * NSL;
    const int P = a.HoA * a.WoA;
    const long long m = (long long)mt * 128 + wave * 32 + l31;
    const bool ok = m < a.Mtot;
    const int mm = (int)(ok ? m : a.Mtot - 1);
    const int n = mm / P, pix = mm - n * P;
    const size_t slab_bytes = (size_t)npass * 2 * KB * 1024;
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.w);
    auto slab_of = [&](int sl) { return min(s0 + sl, nslabs - 1); };
    auto stage = [&](int p, int sl, int h, int buf) {
        const unsigned char* src = wbase + (size_t)slab_of(sl) * slab_bytes + (size_t)((p * 2 + kh) * KP + h) * PART + lane * 16;
        unsigned char* dst = smem + buf * STAGE + kh * PART;
#pragma unroll
        for (int k = 0; k < KPB / 4; ++k) GLDS16(src + (k * 4 + wave) * 1024, dst + (k * 4 + wave) * 1024);
    };
    stage(0, 0, 0, 0);
    if (wave8 < NSL && lane < 16) GLDS16((lane < 8 ? a.scale : a.shift - 32) + slab_of(wave8) * 32 + lane * 4, smem + SS_OFF + wave8 * 256);

    const int tok = n * P + pix;
    float lmean = 0.f, lrstd = 1.f;
    if (a.ln_in != nullptr && kh == 0) {
        const float2* pp = reinterpret_cast<const float2*>(a.ln_in) + (size_t)tok * a.ln_in_tiles;
        float su = 0.f, sq = 0.f;
        for (int t = hh; t < a.ln_in_tiles; t += 2) { const float2 v = pp[t]; su += v.x; sq += v.y; }
        su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
        const float invC = 1.f / (float)(a.ln_in_tiles * 32);
        lmean = su * invC;
        lrstd = rsqrtf(fmaxf(sq * invC - lmean * lmean, 0.f) + a.ln_eps);
    }
    f32x16 acc[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[sl][r] = 0.f;

    int parts_done = 0;
    for (int p = 0; p < npass; ++p) {
        // this wave's 640 channels of the pass
        const f16* xb = a.x + ((size_t)(n * a.x_cbt + a.x_cb0 + (p * 2 + kh) * KB) * P + pix) * 16 + hh * 8;
        f16x8 af[KB];
#pragma unroll
        for (int q = 0; q < KB; ++q) af[q] = *reinterpret_cast<const f16x8*>(xb + (size_t)q * P * 16);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int h = 0; h < KP; ++h) {
                const int cur = parts_done & 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's copies of the part (and, at a pass's first part, its A rows) landed
                __syncthreads();
                if (h + 1 < KP) stage(p, sl, h + 1, cur ^ 1);
                else if (sl + 1 < NSL) stage(p, sl + 1, 0, cur ^ 1);
                else if (p + 1 < npass) stage(p + 1, 0, 0, cur ^ 1);
                ++parts_done;
                const unsigned char* Bs = smem + cur * STAGE + kh * PART + (hh * 32 + l31) * 16;
                f16x8 wf[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wf[q] = *reinterpret_cast<const f16x8*>(Bs + q * 1024);
#pragma unroll
                for (int q = 0; q < KPB; ++q) {
                    acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[q & 3], af[h * KPB + q], acc[sl], 0, 0, 0);
                    if (q + 4 < KPB) wf[q & 3] = *reinterpret_cast<const f16x8*>(Bs + (q + 4) * 1024);
                }
            }
    }
    // the K halves of a subtile meet in LDS; the lower half's wave runs the epilogues
    if (kh == 1) {
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            f32x4* const xw = reinterpret_cast<f32x4*>(smem + XCH_OFF + (sl * 4 + wave) * 4096) + lane;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) xw[g4 * 64] = (f32x4){acc[sl][4 * g4], acc[sl][4 * g4 + 1], acc[sl][4 * g4 + 2], acc[sl][4 * g4 + 3]};
        }
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        if (s0 + sl >= nslabs) break;              // block-uniform
        const f32x4* const xw = reinterpret_cast<const f32x4*>(smem + XCH_OFF + (sl * 4 + wave) * 4096) + lane;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 o = xw[g4 * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[sl][4 * g4 + r] += o[r];
        }
        lin_slab_epilogue(a, acc[sl], s0 + sl, reinterpret_cast<const float*>(smem + SS_OFF + sl * 256), n, pix, ok, tok, lmean, lrstd, hh, P);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*k3_kernel_t)(const K3Args);

static k3_kernel_t k3_pick(int G, int NBT, int PXW, int NC8, int T) {
#define K3CASE(g, n, p, c, t) \
    if (G == g && NBT == n && PXW == p && NC8 == c && T == t) return (k3_kernel_t)conv3_kernel<g, n, p, c, t>
    K3CASE(1, 2, 4, 2, 9); K3CASE(1, 2, 2, 2, 9); K3CASE(1, 1, 4, 2, 9); K3CASE(1, 1, 2, 2, 9);
    K3CASE(1, 2, 2, 4, 9); K3CASE(1, 1, 2, 4, 9);
    K3CASE(1, 2, 1, 2, 9); K3CASE(1, 1, 1, 2, 9); K3CASE(1, 2, 1, 4, 9); K3CASE(1, 1, 1, 4, 9);      // 128-pixel tiles (small maps)
    K3CASE(1, 2, 2, 8, 1); K3CASE(1, 1, 2, 8, 1); K3CASE(1, 2, 2, 2, 1); K3CASE(1, 1, 2, 2, 1);
    K3CASE(1, 4, 2, 4, 1); K3CASE(1, 2, 2, 4, 1); K3CASE(1, 1, 2, 4, 1);
    K3CASE(4, 1, 2, 2, 9); K3CASE(4, 1, 2, 4, 9);
    K3CASE(4, 1, 2, 2, 16);                             // upsample + conv as four phases (ConvPlan::ups4)
#undef K3CASE
    return nullptr;
}

static k3_kernel_t k3_pick_q8(int NBT, int PXW, int NC8) {     // fp8 operands: 3x3 stride 1
#define K3Q(n, p, c) if (NBT == n && PXW == p && NC8 == c) return (k3_kernel_t)conv3_kernel<1, n, p, c, 9, 1, 1>
    K3Q(2, 4, 2); K3Q(2, 2, 2); K3Q(1, 4, 2); K3Q(1, 2, 2); K3Q(2, 2, 4); K3Q(1, 2, 4); K3Q(2, 1, 2); K3Q(1, 1, 2);
#undef K3Q
    return nullptr;
}

static k3_kernel_t k3_pick_mx(int NBT, int PXW) {     // MX-scaled fp8 operands: 3x3 stride 1, 64-channel chunks
#define K3M(n, p) if (NBT == n && PXW == p) return (k3_kernel_t)conv3_kernel<1, n, p, 4, 9, 1, 2>
    K3M(2, 2); K3M(1, 2); K3M(2, 1); K3M(1, 1);
#undef K3M
    return nullptr;
}

static k3_kernel_t k3_pick_s2(int NBT, int NC8) {     // 3x3 stride 2 pad 1 (face-encoder / U-Net downsamples): PXW = 2
    if (NC8 == 2) return NBT == 2 ? (k3_kernel_t)conv3_kernel<1, 2, 2, 2, 9, 2> : (k3_kernel_t)conv3_kernel<1, 1, 2, 2, 9, 2>;
    return nullptr;
}

static int ceil_log2_(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static unsigned magic_u16_(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }

#define HIPCHK3(expr)                                                                 \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(_e);        \
            return -2;                                                                \
        }                                                                             \
    } while (0)

constexpr int kMaxKSplit = 32;

bool conv3_lin_fk_k(int Cin) { return Cin == 320 || Cin == 640 || Cin == 512 || Cin == 384 || Cin == 1280; }

// lin_mp_kernel's slabs per block for a K = 2560 / 5120 linear layer on `rows` tokens, 0: not its case.  Knob LIN_MP = 1: the kernel is used where its
// grid is ONE round of blocks on the 256 CUs (measured, profiles/r06_lin_mp_ab.txt: 16^2 ff.net.2 of a 16-frame pass 43 -> 36 us as 224 blocks of 3
// slabs, 8^2 52 -> 32 us as 160 blocks of 2; in several rounds - 64 frames - it loses to conv3: 113 -> 131 us), with the fewest slabs per block that fit
// one round; 2 / 3: forced.
int conv3_lin_mp_nsl(int Cin, long long rows, int Cout) {
    const int kn = knob(K_LIN_MP);
    if (!kn || !knob(K_LIN_FK) || !(Cin == 2560 || Cin == 5120) || rows < knob(K_LIN_FK_MIN_ROWS) || Cout % 16) return 0;
    if (kn == 2 || kn == 3) return kn;
    const long long mt = (rows + 127) / 128;
    const int nslabs = (Cout + 31) / 32;
    for (int nsl = 2; nsl <= 3; ++nsl)
        if (mt * ((nslabs + nsl - 1) / nsl) <= 256) return nsl;
    return 0;
}

// Split-K factor of a launch with `base` items and `nchunks` channel chunks: only under-filled grids are split, every
// split keeps >= 2 chunks.  Depends on the batch size through `base`: outputs of launches with different splits
// differ by fp32 summation order only.  `cap` > 0 (3x3 / transposed / strided layers): split only below 128 items,
// towards 256 items, at most `cap` ways -- measured (profiles/r02_conv_sweep.txt): a second split of a 192-item
// transposed conv costs 55.7 us against 41.9 unsplit (the fp32 slabs + finish launch outweigh the fill), while 16..64
// item launches on 4^2 / 8^2 maps are 15-25 % faster at 4 (conv) / 4-8 (transposed) splits.  cap == 0 (1x1 / linear layers,
// K up to 8192): fill the chip.
static int k3_ksplit(long long base, int nchunks, int cap) {
    if (nchunks < 4) return 1;
    int s;
    if (cap > 0) {
        if (base >= 128) {
            // ... except a very deep channel loop on a small map: the 3x3 convs on the 960..2560-channel concatenated inputs of
            // MuseTalk's U-Net up blocks (60-160 chunks, 320-640 items of 128 px x 32 ch) are 10-20 % faster four ways
            // (profiles/r02_mt_linear_split_ab.txt); plain convs only (a split 192-item transposed conv loses, see above)
            if (!(cap == 4 && nchunks >= 60 && base <= 1024)) return 1;
            s = 4;
        } else {
            s = std::min(cap, (int)((256 + base - 1) / base));
        }
    } else {
        // 1x1 / linear layers.  Between 128 and 200 items only a deep channel loop repays the fp32 slabs + finish launch:
        // MuseTalk's 160-item projections with K = 384..640 (12-20 chunks) are 6-10 us faster unsplit, its K = 1920..5120
        // layers (ff.net.2, the concatenated-input shortcuts) 12-33 us slower (profiles/r02_mt_linear_split_ab.txt)
        if (base >= 200 || (base >= 128 && nchunks < 48)) return 1;
        s = (int)((256 + base - 1) / base);
    }
    s = std::min(s, std::min(nchunks / 2, kMaxKSplit));
    return std::max(s, 1);
}

size_t conv3_partial_bytes(const ConvPlan& p, int N, int H, int W) {
    if (!p.v3) return 0;
    int Ho = H, Wo = W;
    if (p.v3_G == 4) { Ho = 2 * H; Wo = 2 * W; }
    // generous bound: the largest split any batch <= N may choose
    return (size_t)kMaxKSplit * N * Ho * Wo * p.CoutPad * sizeof(float);
}

int conv3_launch(const ConvPlan& p, const ConvIO& io_in, hipStream_t stream, std::string* err) {
    // four-phase upsample-conv: the caller describes the conv on the UPSAMPLED map (io.H x io.W, ups = 1); the kernel works on
    // the source map and writes the four phases of every source pixel, exactly like the merged transposed conv
    ConvIO io = io_in;
    if (p.ups4) {
        if (!io.ups || ((io.H | io.W) & 1)) { if (err) *err = "conv3: a four-phase upsample-conv plan needs ups = 1 and even H, W"; return -1; }
        io.H /= 2; io.W /= 2; io.ups = 0;
    }
    K3Args a;
    memset(&a, 0, sizeof(a));
    const int G = p.v3_G, T = p.v3_T, NC8 = p.NC8;
    a.x = io.x; a.w = p.d_w; a.scale = p.d_scale; a.shift = p.d_shift; a.res = io.res; a.y = io.y;
    a.N = io.N; a.H = io.H; a.W = io.W; a.x_cbt = io.x_ld >> 4; a.x_cb0 = io.x_coff >> 4;
    a.res_cbt = io.res_ld >> 4; a.res_cb0 = io.res_coff >> 4;
    int y_ld = io.y_ld;
    a.relu = io.relu ? 1 : io.act;
    a.ups = io.ups ? 1 : 0;
    if (io.ups && ((io.H | io.W) & 1)) { if (err) *err = "conv3: upsampled input needs even H, W"; return -1; }
    a.Cout = p.lCout; a.CoutP = p.CoutPad;
    int ext;
    const int S = p.v3_S;
    a.Ho = io.H; a.Wo = io.W;
    if (G == 4) { a.HoA = 2 * io.H; a.WoA = 2 * io.W; a.pad = p.ups4 ? 1 : 0; ext = p.ups4 ? 2 : 1; }
    else if (S == 2) {   // pad 1, or pad 0 + one zero row/column at the bottom/right (p.out_pad)
        a.Ho = (io.H + 2 * p.ph + p.out_pad - 3) / 2 + 1; a.Wo = (io.W + 2 * p.pw + p.out_pad - 3) / 2 + 1;
        a.HoA = a.Ho; a.WoA = a.Wo; a.pad = p.ph; ext = 2;
    }
    else { a.HoA = io.H; a.WoA = io.W; a.pad = (T == 9) ? 1 : 0; ext = (T == 9) ? 2 : 0; }
    if (p.gemm_1x1_expand) {
        if (io.H != 1 || io.W != 1) { if (err) *err = "k x k transposed conv only supported on 1x1 maps"; return -1; }
        if (io.y_ld != p.Cout || io.y_coff != 0) { if (err) *err = "1x1-expand output must be contiguous"; return -1; }
        y_ld = p.lCout;     // [N][k*k*Cout/16][1][1][16] == [N][k][k][Cout] == a 1-pixel NHWC map: the caller re-views it
    }
    a.y_cbt = y_ld >> 4; a.y_cb0 = io.y_coff >> 4;
    if ((io.x_ld | io.x_coff | y_ld | io.y_coff | p.lCout) & 15) { if (err) *err = "conv3: channel counts/offsets must be multiples of 16"; return -1; }
    if (io.res && ((io.res_ld | io.res_coff) & 15)) { if (err) *err = "conv3: residual channel count/offset must be a multiple of 16"; return -1; }
    if ((double)io.N * io.H * io.W * io.x_ld >= 2147483647.0 || (double)io.N * a.HoA * a.WoA >= 2147483647.0) {
        if (err) *err = "tensor too large for 32-bit offsets"; return -1;
    }
    a.nchunks = (p.Cin / 8) / NC8;
    a.Mtot = (long long)io.N * a.HoA * a.WoA;

    // tile selection (does not change any output element's summation order)
    int NBT = (G == 4) ? 1 : ((p.lCout >= 64) ? 2 : 1);
    int PXW = (G == 1 && T == 9 && NC8 == 2 && S == 1) ? 4 : 2;
    // a caller's per-layer choice wins over the sweep knobs; 0 = the rule below
    const int kn_pxw = io.force_pxw ? io.force_pxw : knob(K_CONV_PXW), kn_nbt = io.force_nbt ? io.force_nbt : knob(K_CONV3_NBT);
    if (kn_pxw == 2 || (kn_pxw == 1 && G == 1 && T == 9 && S == 1) || (kn_pxw == 4 && PXW == 4)) PXW = kn_pxw;
    if (kn_nbt == 1) NBT = 1;
    const bool forced_tile = io.force_pxw != 0 && io.force_nbt != 0;
    int l2w = 0, l2h = 0, NB = 1, PH = 1, PW = 1, npix = 0, SLOTS = 0, swz_x = 1, swz_row = 0;
    long long blocks = 0;
    auto geom = [&](int pxw) -> bool {
        const int M = 128 * pxw;
        const int lm = ceil_log2_(M);
        l2w = std::min(5, ceil_log2_(a.Wo));
        l2h = std::min(lm - l2w, ceil_log2_(a.Ho));
        NB = M >> (l2w + l2h);
        NB = std::max(1, std::min(NB, io.N));
        PH = ((1 << l2h) - 1) * S + 1 + ext; PW = ((1 << l2w) - 1) * S + 1 + ext;
        if (S == 2) PW = (PW + 1) & ~1;                      // the stride-2 LDS image permutes units inside aligned pixel pairs of a row
        // stride-1 images: column key on 32-pixel tile rows, row key below (header); 8-pixel rows with a halo need a row pitch of 12 pixels for it
        swz_x = 1; swz_row = 0;
        if (S == 1 && l2w <= 4 && knob(K_LDS_SWZ)) {
            swz_x = 0; swz_row = 1;
            if (l2w == 3 && ext > 0) PW = 12;
        }
        // tiny maps: the halo makes NB patches larger than the staging budget -> fewer images per tile
        while (NB > 1 && (NC8 / 2) * ((2 * NB * PH * PW + 63) / 64 * 64) > k3_maxa(pxw, NC8, S, T) * 256) --NB;
        npix = NB * PH * PW;
        SLOTS = (2 * npix + 63) / 64 * 64;
        const int tiles_x = (a.Wo + (1 << l2w) - 1) >> l2w, tiles_y = (a.Ho + (1 << l2h) - 1) >> l2h;
        const int tiles_n = (io.N + NB - 1) / NB;
        blocks = (long long)tiles_x * tiles_y * tiles_n;
        a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.tiles_n = tiles_n;
        return (NC8 / 2) * SLOTS <= k3_maxa(pxw, NC8, S, T) * 256 && npix < 32768;
    };
    bool fit;
    if (G == 1 && T == 9 && S == 1 && kn_pxw == 0 && kn_nbt == 0 && knob(K_TILE_RULE)) {
        // 3x3 stride 1: the largest tile that still gives the chip ~1.5 items per CU.  Measured per layer and frame count
        // (scripts/conv_sweep2.py, profiles/r02_conv_sweep.txt): with fewer items a launch runs one wave per SIMD and
        // cannot hide its own DMA latency -- at 16 frames 128 ch @32^2 takes 22.8 us as 128 items of 256 px x 64 ch and
        // 10.7 us as 512 items of 128 px x 32 ch -- while smaller tiles than needed re-read the weight slab for nothing.
        static const int cand[6][2] = {{4, 2}, {4, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};      // {PXW, NBT}
        fit = false;
        for (int ci = 0; ci < 6; ++ci) {
            const int pxw = cand[ci][0], nbt = cand[ci][1];
            // 512 px x 32 ch only for the layers that have no more than 32 output channels
            if ((pxw == 4 && NC8 != 2) || (nbt == 2 && p.lCout < 64) || (pxw == 4 && nbt == 1 && p.lCout >= 64)) continue;
            // the fused output head exists for the 512- and 256-pixel tiles only: a one-frame launch (--batch_size 1, or the
            // one-frame remainder of a micro-batched call) would otherwise settle on 128-pixel tiles and be rejected below
            if (io.head_w != nullptr && io.head_outs != nullptr && pxw == 1) continue;
            if (!geom(pxw)) continue;
            fit = true; PXW = pxw; NBT = nbt;
            // (640 input channels and more: ~1 item per CU is enough - a 40-chunk item hides its own start-up, and the larger tile
            // re-reads less: MuseTalk's 640-channel 16^2 convs 67 -> 48 us as 320 items of 128 px x 64 ch instead of 640 of 128 x 32)
            if (blocks * ((p.lCout + 32 * nbt - 1) / (32 * nbt)) >= (a.nchunks >= 40 ? 256 : 384)) break;
        }
    } else {
        fit = geom(PXW);
        if (PXW == 4) {
            const long long nt = (p.lCout + 32 * NBT - 1) / (32 * NBT);
            if (!fit || (!forced_tile && blocks * nt < knob(K_CONV_PXW4_MIN))) { PXW = 2; fit = geom(PXW); }
        }
    }
    if (!fit) { if (err) *err = "conv3: patch does not fit the staging budget"; return -1; }
    // 1x1 convs (plain GEMMs): a 128-cout block halves the A traffic per MAC (the A tile has no tap reuse to amortise it)
    if (T == 1 && NC8 == 4 && G == 1 && p.lCout % 128 == 0 && ((blocks * (p.lCout / 128) >= 384 && kn_nbt == 0) || kn_nbt == 4)) NBT = 4;
    // (1x1 / linear layers on >= 1024 pixels: below one item per CU; MuseTalk's 640-channel projections on 4096 tokens are 10 %
    // faster as 320 items of 256 px x 32 ch than as 160 of 256 x 64)
    if (NBT == 2 && blocks * ((p.lCout + 63) / 64) < ((T == 1 && blocks >= 4) ? 256 : 128) && !(G == 1 && T == 9 && S == 1 && knob(K_TILE_RULE)) && !forced_tile) NBT = 1;
    const int BN = NBT * 32;
    a.n_ntiles = (p.lCout + BN - 1) / BN;
    a.ablate = LTK_ABLATE_BUILD ? knob(K_ABLATE) : 0;
    // LTK_SPLITK=0: never split (batch-size independent summation order); LTK_KSPLIT=n forces a factor (sweeps)
    int ksplit = knob(K_SPLITK) ? k3_ksplit(blocks * a.n_ntiles, a.nchunks, (T == 1 || !knob(K_TILE_RULE)) ? 0 : (G == 4 ? 8 : 4)) : 1;
    const int fks = io.force_ksplit ? io.force_ksplit : knob(K_KSPLIT);
    if (fks > 0 && knob(K_SPLITK)) ksplit = std::max(1, std::min(std::min(fks, kMaxKSplit), a.nchunks));
    if (io.head_w != nullptr && io.head_outs != nullptr) ksplit = 1;      // the fused head finishes in the epilogue: no partial slabs
    a.ln_out = io.ln_out; a.ln_in = io.ln_in; a.ln_out_tiles = io.ln_out_tiles; a.ln_in_tiles = io.ln_in_tiles; a.ln_eps = io.ln_eps;
    if (io.ln_out || io.ln_in) {                                          // LayerNorm fold: 1x1 layers, the epilogue sees whole sums
        if (!(G == 1 && T == 1 && !p.q8) || (io.ln_out && (p.lCout % 32 || io.ln_out_tiles != p.lCout / 32)) || (io.ln_in && io.ln_in_tiles <= 0)) {
            if (err) *err = "conv3: the LayerNorm fold is a 1x1-layer feature (32-channel tiles)"; return -1;
        }
        ksplit = 1;
    }
    if (a.relu == 4) {                                                    // GEGLU epilogue: value and gate meet in the accumulators
        if (!(G == 1 && T == 1 && !p.q8 && p.lCout % 32 == 0 && !io.res)) { if (err) *err = "conv3: the GEGLU epilogue is a 1x1-layer feature"; return -1; }
        ksplit = 1;
    }
    // short-K linear layers on many tokens: lin_fk_kernel (A rows in registers, full-K weight slabs through LDS)
    const int mp_nsl = (G == 1 && T == 1 && S == 1) ? conv3_lin_mp_nsl(p.Cin, a.Mtot, p.lCout) : 0;
    if (G == 1 && T == 1 && S == 1 && !p.q8 && !p.mx && !a.ups && !p.gemm_1x1_expand && knob(K_LIN_FK) && (conv3_lin_fk_k(p.Cin) || mp_nsl) &&
        a.nchunks * NC8 * 8 == p.Cin && a.Mtot >= knob(K_LIN_FK_MIN_ROWS) && a.ablate == 0 && p.lCout % 16 == 0 &&
        (a.relu == 0 || a.relu == 1 || a.relu == 4)) {
        const int nslabs = (p.lCout + 31) / 32;
        if (mp_nsl) {          // lin_mp_kernel: passes of 1280 channels, NSL slabs' accumulators per block
            const int NSL = mp_nsl;
            const long long mt = (a.Mtot + 127) / 128;
            const int ng = (nslabs + NSL - 1) / NSL;
            typedef void (*mp_t)(const K3Args, int, int, int);
            const mp_t mk = NSL == 2 ? (mp_t)lin_mp_kernel<2> : (mp_t)lin_mp_kernel<3>;
            const size_t mb = (size_t)2 * 2 * 20 * 1024 + 1024 + (size_t)NSL * 4 * 4096;
            HIPCHK3((hipError_t)ensure_dyn_lds((const void*)mk, (int)mb));
            a.ksplit = 1; a.partial = nullptr;
            hipLaunchKernelGGL(mk, dim3((unsigned)(mt * ng)), dim3(512), mb, stream, a, ng, nslabs, p.Cin / 1280);
            HIPCHK3(hipGetLastError());
            return 0;
        }
        const long long mtiles = (a.Mtot + 127) / 128;
        // (K = 1280: one 8-wave block per CU)
        const int want = p.Cin == 1280 ? knob(K_LIN_FK_BLOCKS) / 2 : knob(K_LIN_FK_BLOCKS);
        int ngroups = (int)std::max(1ll, std::min((long long)nslabs, (want + mtiles - 1) / mtiles));
        const int cpg = (nslabs + ngroups - 1) / ngroups;
        ngroups = (nslabs + cpg - 1) / cpg;
        const long long grid = mtiles * ngroups;
        if (grid > 0 && grid <= 0x7fffffffll) {
            typedef void (*lin_t)(const K3Args, int, int, int);
            // K = 320 / 640: MuseTalk's 32^2 / 16^2 transformer levels; 512: the VAE mid-block attention; 384: the stacked k | v projection of the audio context
            // 1280: the 8^2 level's projections and ff.net.2 of the 32^2 level (two waves per pixel subtile, K halves)
            const lin_t lk = p.Cin == 320 ? (lin_t)lin_fk_kernel<20, 20, 1> : p.Cin == 640 ? (lin_t)lin_fk_kernel<40, 20, 1> : p.Cin == 512 ? (lin_t)lin_fk_kernel<32, 16, 1>
                           : p.Cin == 384 ? (lin_t)lin_fk_kernel<24, 24, 1> : (lin_t)lin_fk_kernel<40, 20, 2>;
            const int ks = p.Cin == 1280 ? 2 : 1;
            const size_t lbytes = (size_t)2 * ks * (p.Cin == 512 ? 16 : p.Cin == 384 ? 24 : 20) * 1024 + 512 + (ks == 2 ? 2 * 4 * 4096 : 0);
            HIPCHK3((hipError_t)ensure_dyn_lds((const void*)lk, (int)lbytes));
            a.ksplit = 1; a.partial = nullptr;
            hipLaunchKernelGGL(lk, dim3((unsigned)grid), dim3(256 * ks), lbytes, stream, a, cpg, ngroups, nslabs);
            HIPCHK3(hipGetLastError());
            return 0;
        }
    }
    if (ksplit > 1) {   // fall back to fewer splits when the caller's scratch is smaller
        while (ksplit > 1 && (!io.partial || io.partial_cap < (size_t)ksplit * a.Mtot * p.CoutPad * sizeof(float))) ksplit /= 2;
    }
    a.ksplit = ksplit;
    a.chunks_per_split = (a.nchunks + ksplit - 1) / ksplit;
    if ((long long)a.chunks_per_split * (ksplit - 1) >= a.nchunks) {   // no empty split
        ksplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
        a.ksplit = ksplit;
    }
    a.log2TW = l2w; a.log2TH = l2h; a.NB = NB; a.PH = PH; a.PW = PW; a.npix = npix; a.SLOTS = SLOTS;
    a.swz_x = swz_x; a.swz_row = swz_row;
    a.magicPW = magic_u16_(PW); a.magicPHW = magic_u16_(PH * PW);

    if (ksplit > 1) a.partial = io.partial;

    const size_t a_bytes = (size_t)(NC8 / 2) * a.SLOTS * 16, b_bytes = (size_t)T * NC8 * BN * 16;
    size_t lds = 2 * (a_bytes + b_bytes);
    lds = (lds + 255) / 256 * 256;
    a.lds_scale_off = (int)lds;
    if (T != 1) lds += 2 * BN * sizeof(float);
    if (lds > 160 * 1024) { if (err) *err = "conv3: LDS budget exceeded"; return -1; }
    if (p.q8 && !(G == 1 && T == 9 && S == 1)) { if (err) *err = "conv3: fp8 operands are implemented for 3x3 stride-1 convs"; return -1; }
    const bool head = io.head_w != nullptr && io.head_outs != nullptr;
    if (head && !(G == 1 && T == 9 && S == 1 && !p.q8 && NC8 == 2 && NBT == 1 && p.lCout == 32 && (PXW == 4 || PXW == 2) && ksplit == 1)) {
        if (err) *err = "conv3: fused head needs the 3x3 stride-1 32-cout configuration";
        return -1;
    }
    k3_kernel_t k = p.mx ? k3_pick_mx(NBT, PXW) : p.q8 ? k3_pick_q8(NBT, PXW, NC8) : (S == 2) ? k3_pick_s2(NBT, NC8) : k3_pick(G, NBT, PXW, NC8, T);
    if (!k) { if (err) *err = "conv3: no kernel instantiation"; return -1; }
    const long long nblk = blocks * a.n_ntiles * ksplit;
    if (nblk <= 0 || nblk > 0x7fffffffll) { if (err) *err = "bad grid"; return -1; }
    HIPCHK3((hipError_t)ensure_dyn_lds((const void*)k, 160 * 1024));
    a.nitems = (int)nblk;
    const int persist_blocks = knob(K_CONV_PERSIST);      // 0: one block per item
    const long long grid = (persist_blocks > 0 && nblk > persist_blocks) ? persist_blocks : nblk;
    if (head) {
        typedef void (*k3_head_t)(const K3Args, const HeadArgs);
        const k3_head_t kh = PXW == 4 ? (k3_head_t)conv3_head_kernel<4> : (k3_head_t)conv3_head_kernel<2>;
        HIPCHK3((hipError_t)ensure_dyn_lds((const void*)kh, 160 * 1024));
        HeadArgs h;
        h.w = io.head_w;
        h.outs = reinterpret_cast<const OutPtrs*>(io.head_outs);
        hipLaunchKernelGGL(kh, dim3((unsigned)grid), dim3(256), lds, stream, a, h);
        HIPCHK3(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, stream, a);
    HIPCHK3(hipGetLastError());
    if (ksplit > 1) {
        const long long items = a.Mtot * (p.lCout >> 3);
        hipLaunchKernelGGL(conv3_finish_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream,
                           (const float*)io.partial, ksplit, a.Mtot, a.HoA * a.WoA, p.CoutPad, p.lCout, (const float*)p.d_scale,
                           (const float*)p.d_shift, io.res, a.res_cbt, a.res_cb0, io.y, a.y_cbt, a.y_cb0, a.relu);
        HIPCHK3(hipGetLastError());
    }
    return 0;
}

}  // namespace ltk
