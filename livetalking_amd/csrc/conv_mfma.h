// Implicit-GEMM fp16 convolution on gfx950 MFMA (v_mfma_f32_32x32x16_f16) over channel-blocked activations
// (CB16: [N][C/16][H][W][16]; network inputs of <= 8 channels are [N][H][W][8]).  Host-side plan + launch interface
// shared by conv_mfma.hip (register-staged first-generation kernel) and conv3_mfma.hip (LDS-DMA kernel); DESIGN.md §2-3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace ltk {

typedef _Float16 f16;

// GELU (exact / erf form, torch.nn.functional.gelu's default, which diffusers' GEGLU uses) for the GEGLU feed-forward:
// x * Phi(x) with erfc(|x| / sqrt 2) from Abramowitz-Stegun 7.1.26 (absolute error <= 1.5e-7: 3 000x below the fp16 resolution of the
// result), evaluated without cancellation on either side of 0: one v_rcp, one v_exp, 9 fma / mul - erff() is ~3x that and was
// most of the GEGLU epilogue's time (21 M gates per 16-frame launch on the 32^2 level).
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));      // v_rcp_f32 (1 ulp); __frcp_rn compiles to the 12-instruction correctly rounded division
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float pe = p * t * __expf(-z * z);          // erfc(z)
    return 0.5f * x * (x >= 0.f ? 2.f - pe : pe);
}

constexpr int kConvBM = 256;       // output pixels per workgroup (4 waves x 64)
constexpr int kMaxTaps = 52;       // 7x7 = 49 (+ pairing pad)
constexpr int kMaxPhases = 4;      // sub-pixel phases of a stride-2 transposed conv

// One sub-pixel phase of a (transposed) convolution: a tap list over the input
// patch, its own packed weight slab and output pixel offset.
struct ConvPhase {
    int T;                  // taps
    int ooy, oox;           // output pixel = (oy*osy + ooy, ox*osx + oox)
    int w_off16;            // offset of this phase's packed weights, in 16-byte units
    short tapoff[kMaxTaps]; // dy*PW + dx per tap (patch-pixel units); unused entries 0
};

struct ConvArgs {
    const f16* x;           // input CB16 buffer of x_ld channels; this tensor = channels [x_coff, x_coff+Cin)
    const f16* w;           // packed weights (see pack_weights)
    const float* scale;     // [Cout] folded BN scale
    const float* shift;     // [Cout] folded BN shift (+conv bias)
    const f16* res;         // residual (same pixel mapping as y) or nullptr
    f16* y;                 // output CB16 buffer of y_ld channels, written at [y_coff, y_coff+Cout)
    int N, H, W;
    int x_ld, x_coff;
    int Ho, Wo;             // logical output grid one phase covers
    int y_ld, y_coff, HoA, WoA, osy, osx;
    int res_ld, res_coff;
    int Cin8;               // Cin / 8 (channel planes of 8 halfs)
    int Cout;
    int sh, sw, pad_y, pad_x;
    int PH, PW;             // input patch per image tile
    int NPIXP;              // padded pixels per LDS channel plane
    int log2TW, log2TH, NB; // tile = NB images x TH x TW output pixels (<= 256)
    int tiles_x, tiles_y, tiles_n, n_ntiles, nphase;
    unsigned magicPW, magicPHW;
    int nchunks;            // ceil(Cin8 / NC8)
    int Tp;                 // taps per packed chunk slab (T, or T rounded up to even when NC8==1)
    int relu;
    ConvPhase ph[kMaxPhases];
};

// Static description of a conv layer (weights already packed on device).
struct ConvPlan {
    // geometry
    int Cin = 0, Cout = 0, CoutPad = 0;   // CoutPad: multiple of 128
    int lCout = 0;                        // output channels of the executed conv (k*k*Cout for the 1x1-expand)
    int kh = 1, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0;
    bool transposed = false;              // ConvTranspose2d
    int out_pad = 0;
    bool gemm_1x1_expand = false;         // convT k x k on a 1x1 map == 1x1 conv to k*k*Cout channels
    // kernel config
    int NC8 = 4, NBT = 2;                 // channel planes per chunk, 32-cout subtiles per block
    bool tt9 = false;                     // 3x3 taps compiled in
    int mode = 1;                         // 0 double-buffered LDS, 1 single-buffered + register prefetch
    int nphase = 1;
    int Tp = 0;
    // conv3 (LDS-DMA staged, merged-phase convT, split-K): conv3_mfma.hip
    bool v3 = false;
    int v3_G = 1;                         // 1 = conv, 4 = merged transposed conv (4 sub-pixel phases per block)
    int v3_T = 9;                         // taps: 9 (3x3 / merged convT) or 1 (1x1)
    int v3_S = 1;                         // stride (3x3 pad 1 only): 1 or 2
    // nearest-2x upsample + 3x3 conv (diffusers Upsample2D) as FOUR sub-pixel phases of 2x2 taps each on the source map:
    // v3_G = 4, v3_T = 16 (operand, phase) pairs with the 3x3 weights pre-summed per phase -- 16 MACs per source pixel and
    // channel pair instead of 36; `io.H, io.W` stay the upsampled size
    bool ups4 = false;
    // fp8 operands (conv3, 3x3 stride 1): the input is [N][CinReal/32][H][W][32] e4m3 bytes holding x * act_scale; Cin
    // above then counts 16-bit units (= CinReal / 2) so that every byte offset of the fp16 path carries over
    bool q8 = false;
    int CinReal = 0;
    // q8 on the MX-scaled instruction (v_mfma_scale_f32_32x32x64_f8f6f4, unit E8M0 scales: 2x the MAC rate of every other fp8 /
    // fp16 MFMA on gfx950): 64 e4m3 channels per chunk (NC8 = 4) = ONE MFMA per tap and tile pair; lane half hh contracts the
    // 32-byte cell 2*chunk + hh.  Same tensors, weight bytes and scales as q8.
    bool mx = false;
    ConvPhase phase[kMaxPhases];
    // device data
    f16* d_w = nullptr;
    float* d_scale = nullptr;
    float* d_shift = nullptr;
    size_t w_bytes = 0;
    double macs_per_image(int H, int W) const;
    void out_dims(int H, int W, int* Ho, int* Wo) const;
};

// Build a plan: choose kernel configuration, pack weights (fp32 torch layout ->
// fp16 [ntile][chunk][tap][plane][cout][8]) and upload.  `weight` is
// [Cout][Cin][kh][kw] (conv) or [Cin][Cout][kh][kw] (transposed).
// Cin is padded up to a multiple of 8 with zero weights (CinPad = plan.Cin).
// `hint_hw` = pixels per image of the map the layer runs on (0 = unknown): it only steers the
// channel-chunk width, which is baked into the weight pack order.
int conv_plan_create(ConvPlan* p, const float* weight, int Cin, int Cout, int kh, int kw,
                     int sh, int sw, int ph, int pw, bool transposed, int out_pad,
                     const float* scale, const float* shift, std::string* err, int hint_hw = 0,
                     int quant = 0, float act_scale = 1.f, int ups4 = 0);
// `ups4` = 1: the conv always runs on a nearest-2x upsampled input (ConvIO::ups): build the four-phase form (ConvPlan::ups4).
// `quant` = 2: the same on the MX-scaled fp8 MFMA (ConvPlan::mx; needs Cin % 64 == 0).
// `quant` = 1: e4m3 weights with one scale per output channel (224 / max|w|), folded together with `act_scale`
// (what the producer of the fp8 input multiplied by) into the epilogue scale.  3x3 stride-1 pad-1 convs, Cin % 32 == 0.

// fp32 -> OCP e4m3fn byte, round to nearest even, saturating to +-448 (host side of the weight packer; the device side
// uses v_cvt_pk_fp8_f32)
unsigned char f32_to_e4m3(float v);
void conv_plan_destroy(ConvPlan* p);

struct ConvIO {
    const f16* x; int N, H, W; int x_ld, x_coff;
    f16* y; int y_ld, y_coff;
    const f16* res; int res_ld, res_coff;
    int relu;                      // 1: ReLU epilogue (shorthand for act == 1)
    int act = 0;                   // epilogue activation when relu == 0: 0 none, 2 GELU (erf), 3 SiLU
    int ups = 0;                   // conv3 only: the input map is H/2 x W/2 and is read through a nearest-neighbour
                                   // 2x upsample (diffusers Upsample2D: F.interpolate(scale_factor=2) then conv)
    int force_pxw = 0, force_nbt = 0, force_ksplit = 0;   // conv3: per-layer tile / split choice of the caller's table (0 = rule)
    const float* head_w = nullptr;     // conv3, 3x3 stride-1 layers of 32 output channels only: fuse the Wav2Lip output head
    const void* head_outs = nullptr;   // (1x1 conv 32->3 + sigmoid + uint8 truncation; wav2lip_v2.py:90-91): device [3][32]+[3]
                                       // weights and a DEVICE table (OutPtrs, misc_kernels.h) of per-frame uint8 [256][256][3] outputs; `y` unused
    // conv3 1x1 / rowconv: LayerNorm folded into the linear layers around it (conv3_mfma.hip K3Args): ln_out = per (token, 32-channel
    // tile) sum / sum of squares of the stored output, [tokens][ln_out_tiles] float2; ln_in = the same of the INPUT tensor, with
    // ln_in_tiles = its channels / 32: the layer then computes W LN(x) + b from the raw x (weights / scale / shift folded by the caller)
    float* ln_out = nullptr; const float* ln_in = nullptr;
    int ln_out_tiles = 0, ln_in_tiles = 0;
    float ln_eps = 1e-5f;
    float* partial = nullptr;      // split-K scratch (fp32 slabs) and its capacity in bytes; conv3 splits the
    size_t partial_cap = 0;        // channel loop of under-filled launches only when this is large enough
};

// Enqueue the layer on `stream`.  Returns 0 or a negative error (message in *err).
int conv_launch(const ConvPlan& p, const ConvIO& io, hipStream_t stream, std::string* err);

// conv3_mfma.hip
int conv3_launch(const ConvPlan& p, const ConvIO& io, hipStream_t stream, std::string* err);
// upper bound of the split-K scratch a launch of `N` images of H x W may use
size_t conv3_partial_bytes(const ConvPlan& p, int N, int H, int W);

// rowgemm.hip: the layers whose maps are one pixel per frame (plain GEMMs with as many rows as frames), for launches of up to
// kRowGemmMaxFrames frames: Y[frame][y_coff + j] = act(scale[j] * sum_k X[frame][x_coff + k] * W[j][k] + shift[j]), j < J, k < K
constexpr int kRowGemmMaxFrames = 32;
struct RowGemmPlan {
    f16* d_w = nullptr;            // packed MFMA fragments [J/16][K/32][64][8]
    float* d_scale = nullptr;
    float* d_shift = nullptr;
    int J = 0, K = 0;
};
int rowgemm_plan_create(RowGemmPlan* p, const float* w_eff /*[J][K]*/, int J, int K, const float* scale /*[J]*/, const float* shift /*[J]*/,
                        std::string* err);
void rowgemm_plan_destroy(RowGemmPlan* p);
int rowgemm_launch(const RowGemmPlan& p, const f16* x, int x_ld, int x_coff, f16* y, int y_ld, int y_coff, int M, int relu,
                   hipStream_t stream, std::string* err);

// rowconv (rowgemm.hip): k x k convolutions on maps of <= 8 x 8 output pixels as the same weight-streaming GEMM, the im2col rows
// gathered on the fly; the plan is a RowGemmPlan over W_eff[j][tap * C + c] (tap = ky * k + kx).  Launches of up to kRowConvMaxRows
// output pixels (frames x Ho x Wo).
constexpr int kRowConvMaxRows = 2048;
struct RowConvIO {
    const f16* x = nullptr; int x_ld = 0, x_coff = 0, H = 0, W = 0;      // input map [N][x_ld/16][H][W][16]
    f16* y = nullptr; int y_ld = 0, y_coff = 0, Ho = 0, Wo = 0;
    const f16* res = nullptr; int res_ld = 0, res_coff = 0;              // residual with the output's geometry, or nullptr
    int N = 0, KW = 3, stride = 1, pad = 1, relu = 1;
    int stride_w = 0;                                                    // column stride when it differs from `stride` (0 = the same)
    float* ln_out = nullptr; const float* ln_in = nullptr;               // LayerNorm fold, as ConvIO (1x1 plans only)
    int ln_out_tiles = 0, ln_in_tiles = 0;
    float ln_eps = 1e-5f;
};
int rowconv_launch(const RowGemmPlan& p, const RowConvIO& io, hipStream_t stream, std::string* err);
// ConvTranspose2d(k3, s2, p1, op1) on a source map of <= 8 x 8 pixels: four per-phase plans p[py * 2 + px] over
// W_eff[j][(dy * (1 + px) + dx) * C + c] = w[c][j][py + 1 - 2 dy][px + 1 - 2 dx], one launch; io.H x io.W source, io.Ho x io.Wo = 2H x 2W output
int rowconvT_launch(const RowGemmPlan* p, const RowConvIO& io, hipStream_t stream, std::string* err);

// conv3_mfma.hip: input-channel counts of 1x1 / linear layers that lin_fk_kernel serves (knob LIN_FK; >= LIN_FK_MIN_ROWS pixels or tokens)
bool conv3_lin_fk_k(int Cin);
int conv3_lin_mp_nsl(int Cin, long long rows, int Cout);     // lin_mp_kernel's slabs per block for this layer on `rows` tokens, 0: not its case

// conv7_mfma.hip: the generator's first layer (Conv2d(6,16,7,1,3) + BN + ReLU on 256x256) fused with the input pack
struct Conv7Plan;
struct FacePtrs;
int conv7_plan_create(Conv7Plan** out, const float* weight /*[16][6][7][7]*/, const float* scale, const float* shift, std::string* err);
void conv7_plan_destroy(Conv7Plan* p);
// faces != nullptr: read the uint8 bank crops (pack fused); else read the packed fp16 input x0 [N][256][256][8]
int conv7_launch(const Conv7Plan* p, const FacePtrs* faces, const f16* x0, int N, f16* y, int y_ld, int y_coff, hipStream_t stream,
                 std::string* err);


// conv7_mfma.hip: the audio encoder's first layer (Conv2d(1,32,3,1,1) + BN + ReLU on the 80 x 16 mel window) as a VALU kernel with the
// mel pack fused (knob AUDIO0)
struct Audio0Plan;
struct MelPtrs;
int audio0_plan_create(Audio0Plan** out, const float* weight /*[32][1][3][3]*/, const float* scale, const float* shift, std::string* err);
void audio0_plan_destroy(Audio0Plan* p);
// `mels`: DEVICE table of per-frame float32 [80][16] windows
int audio0_launch(const Audio0Plan* p, const MelPtrs* mels, int N, f16* y, int y_ld, int y_coff, hipStream_t stream, std::string* err);
// ... and its stride-(3, 1) layer audio_encoder.3 (Conv2d(32,64,3,(3,1),1) + BN + ReLU, 80 x 16 -> 27 x 16): one-wave blocks, MFMA operands loaded straight from global memory (knob AUDIO0 bit 1)
struct Audio3Plan;
int audio3_plan_create(Audio3Plan** out, const float* weight /*[64][32][3][3]*/, const float* scale, const float* shift, std::string* err);
void audio3_plan_destroy(Audio3Plan* p);
int audio3_launch(const Audio3Plan* p, const f16* x, int x_ld, int x_coff, int N, f16* y, int y_ld, int y_coff, hipStream_t stream, std::string* err);
// ... and the face encoder's shallow stride-2 layers (Conv2d(16 / 32 -> Cout, 3, 2, 1) + BN + ReLU) the same way: a wave = one output row x 32 output
// channels, weights in registers, pixel operands straight from global memory (knob CONV_S2D)
struct ConvS2dPlan;
int convs2d_plan_create(ConvS2dPlan** out, const float* weight /*[Cout][Cin][3][3]*/, int Cin, int Cout, const float* scale, const float* shift, std::string* err);
void convs2d_plan_destroy(ConvS2dPlan* p);
int convs2d_launch(const ConvS2dPlan* p, const f16* x, int x_ld, int x_coff, int N, int H, int W, f16* y, int y_ld, int y_coff, hipStream_t stream, std::string* err);

}  // namespace ltk
