// Non-convolution kernels of the MuseTalk path (gfx950): GroupNorm(+SiLU), LayerNorm, multi-head attention on
// MFMA, GEGLU, elementwise helpers, VAE output post-process.  All activations are channel-blocked CB16 fp16
// ([N][C/16][P][16], P = pixels or tokens); see conv3_mfma.hip for the layout rationale.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_mfma.h"

namespace ltk {

// ---- GroupNorm (diffusers ResnetBlock2D.norm1/2, Transformer2DModel.norm, AutoencoderKL norms)
// stats: partial [N][C/16][segs][16][2] fp32 (sum, sum of squares) over `segs` pixel segments
int gn_segments(int N, int C, int P);
void launch_gn_stats(const f16* x, int N, int cbt, int cb0, int C, int P, int segs, float* partial, hipStream_t s);
// y = (x - mean_g) * rstd_g * gamma + beta, optional SiLU; groups of C/groups consecutive channels
void launch_gn_apply(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* partial,
                     int segs, const float* gamma, const float* beta, int silu, f16* y, int y_cbt, int y_cb0, hipStream_t s);

// same, writing e4m3 bytes min(max(y * out_scale, -448), 448) into a [N][C/32][P][32] tensor (y_cbt / y_cb0 in 32-channel
// blocks): the operand format of the fp8 conv path (conv3_mfma.hip, Q = 1)
void launch_gn_apply_fp8(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* partial,
                         int segs, const float* gamma, const float* beta, int silu, float out_scale, unsigned char* y, int y_cbt,
                         int y_cb0, hipStream_t s);

// GroupNorm in ONE launch, block = (image, group): for maps whose (image, group) fits one block's registers (gn_group_fits: even
// channels per group, pixels x channels-per-group <= 30 720).  fp8 != 0: e4m3 output as launch_gn_apply_fp8 (y_cbt / y_cb0 in
// 32-channel blocks)
bool gn_group_fits(int C, int P, int groups);
void launch_gn_group(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, const float* gamma, const float* beta,
                     int silu, f16* y, int y_cbt, int y_cb0, int fp8, float out_scale, hipStream_t s);

// GroupNorm of the large maps in ONE tensor pass (gn_coop_kernel): a block keeps a 2048-pixel slice of an (image, 16-channel block) in registers,
// the gn_coop_members() = P / 2048 blocks of that (image, channel block) exchange partial sums through `slots` (8 words per block, this launch's
// N * C/16 * members * 8 words; filled with 0xFFFFFFFF by launch_gn_coop_reset BEFORE the launch, once per pass for all its GroupNorms).  0: not its
// case (channels per group other than 4 / 8 / 16, P not a multiple of 2048, fewer than 2 or more than 32 slices).  *err (host-visible) is set to 1
// if a block's wait for its set ran out.
int gn_coop_members(int C, int P, int groups);
void launch_gn_coop_reset(unsigned* slots, size_t words, hipStream_t s);
void launch_gn_coop(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, int groups, float eps, unsigned* slots, unsigned* err,
                    const float* gamma, const float* beta, int silu, f16* y, int y_cbt, int y_cb0, int fp8, float out_scale, hipStream_t s);

// ---- LayerNorm over channels per token (BasicTransformerBlock.norm1/2/3, Whisper layer norms)
void launch_layernorm(const f16* x, int N, int cbt, int cb0, int C, int P, float eps, const float* gamma, const float* beta,
                      f16* y, int y_cbt, int y_cb0, hipStream_t s);

// ---- attention: O = softmax(Q K^T) V per (image, head); the 1/sqrt(d) scale is folded into the q projection.
// q/k/o: CB16 with head h at channel blocks [cb0 + h*d16/16, +d16/16); vt: [N][heads][dv32][Tkp] fp16, keys of every
// 16-group stored in MFMA slot order (launch_v_transpose writes it).  d16 in {48,64,80,160,512}.
int attn_dv32(int d16);
int attn_tkp(int Tk);
void launch_v_transpose(const f16* v, int N, int cbt, int cb0, int heads, int d16, int Tk, f16* vt, hipStream_t s);
// several value tensors over the same Tk keys in ONE launch (h0 / dv32 / Tkp are filled in by the launcher)
struct VtMulti {
    struct Item { const f16* v; f16* vt; int cbt, cb0, heads, d16, dv32, h0; } it[16];
    int n, Tk, Tkp;
};
void launch_v_transpose_multi(VtMulti m, int N, hipStream_t s);
int launch_attention(const f16* q, int q_cbt, int q_cb0, int Tq, const f16* k, int k_cbt, int k_cb0, int Tk, const f16* vt,
                     f16* o, int o_cbt, int o_cb0, int N, int heads, int d16, hipStream_t s);

// ---- GEGLU: y[c] = a[c] * gelu(gate[c]), a = channels [0,C), gate = [C,2C) of x (diffusers GEGLU, exact erf gelu)
void launch_geglu(const f16* x, int N, int x_cbt, int x_cb0, int C, int P, f16* y, int y_cbt, int y_cb0, hipStream_t s);
// y = act(x) elementwise over a channel range (act: 2 = gelu(erf), 3 = silu)
void launch_act(const f16* x, int N, int cbt, int cb0, int C, int P, int act, f16* y, int y_cbt, int y_cb0, hipStream_t s);
// y += pos[token][c] (Whisper embed_positions), pos fp32 [P][C]
void launch_add_pos(f16* x, int N, int cbt, int cb0, int C, int P, const float* pos, hipStream_t s);

// ---- layout bridges
// fp32 NCHW / token-major host-style tensors on the device -> CB16 fp16 (zero padded to whole channel blocks)
void launch_nchw_to_cb16(const float* x, int N, int C, int P, f16* y, int y_cbt, int y_cb0, hipStream_t s);
// tokens fp32 [N][P][C] (+ optional additive table [P][C], e.g. the MuseTalk positional encoding) -> CB16
void launch_tokens_to_cb16(const float* x, int N, int P, int C, const float* add, f16* y, int y_cbt, int y_cb0, hipStream_t s);
struct PtrList64 { const void* p[64]; };
// same, with one fp32 [P][C] block per image given by pointer
void launch_tokens_gather_to_cb16(const PtrList64& src, int N, int P, int C, const float* add, f16* y, int y_cbt, hipStream_t s);
// latents: per-frame fp32 [C][P] planes gathered by pointer -> CB16 (musetalk_avatar.py:134-141)
void launch_gather_latents(const PtrList64& src, int nframes, int C, int P, f16* y, int y_cbt, hipStream_t s);
struct OutList64 { uint8_t* p[64]; };
// vae.py:104-107: (x/2+0.5).clamp(0,1) -> *255 -> round-half-even -> uint8, RGB -> BGR, NHWC [256][256][3]
void launch_vae_post(const f16* x, int x_cbt, int nframes, int P, const OutList64& out, float* out_f32_nchw, hipStream_t s);

// ---- VAE encode bridges (avatar preparation): uint8 BGR faces [n][256][256][3] -> normalised RGB CB16 images
// (2 per face: lower-half masked, full); moments -> fp32 latents [n][8][32][32]
void launch_vae_pre(const uint8_t* d_bgr, int nfaces, f16* y, hipStream_t s);
void launch_vae_latents(const f16* moments, int nfaces, const float* d_noise, float scaling, float* out, hipStream_t s);

// ---- Whisper front end / feature slicing (MuseTalk audio features)
// pcm device fp32 [n_samples] (<= 30 s); basis fp32 [80][201]; logspec scratch fp32 [80][3000]; gmax scratch int;
// y: CB16 [5][3000][16] log-mel features
void launch_whisper_logmel(const float* d_pcm, int n_samples, const float* d_basis, float* d_logspec, int* d_gmax, f16* y, hipStream_t s);
struct WhisperStates { const f16* p[5]; int cb0[5]; };     // each state: CB16 [cbt >= 24][T][16], first block cb0 (buffers are 24 blocks wide)
void launch_whisper_chunks(const WhisperStates& st, int T, int batch, int first_row, int row_step, int rows, float* out, hipStream_t s);

}  // namespace ltk
