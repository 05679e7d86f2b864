// MuseTalk device program (musetalk.hip): graph construction from diffusers-named state dicts and execution.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ltk.h"
#include "conv_mfma.h"

namespace ltk {

struct MtGraph;
struct MtTensor;

MtGraph* mt_graph_new();
void mt_graph_delete(MtGraph* g);
const char* mt_graph_error(const MtGraph* g);
// after the pass's stream has been synchronised: 1 (and mt_graph_error says so) if a cooperative GroupNorm block's wait ran out during it
int mt_gn_error(MtGraph* g);
// before mt_build: run the ResnetBlock2D 3x3 convs on fp8 (e4m3) operands; act_scale = what GroupNorm+SiLU outputs are
// multiplied by before the saturating conversion (<= 0 keeps the default 8)
void mt_set_fp8(MtGraph* g, int on, float act_scale);
double mt_macs_fp8_per_frame(const MtGraph* g);
// debug (knob SAT_CHECK): device counters [2] (halfs at the fp16 / e4m3 limit, non-finite ones) that every op's output is scanned into
void mt_set_sat_counter(MtGraph* g, unsigned long long* d_ctr);
// builds U-Net then VAE decoder; returns 0 or a negative code
int mt_build(MtGraph* g, const ltk_named_tensor* unet_sd, int n_unet, const ltk_named_tensor* vae_sd, int n_vae, int frames);
// tensors the engine feeds / reads
f16* mt_latent_in(MtGraph* g, int* cbt);       // [N][1][1024][16] (8 real channels)
f16* mt_ctx_in(MtGraph* g, int* cbt);          // [N][24][50][16]
f16* mt_unet_out(MtGraph* g, int* cbt);        // [N][1][1024][16] (4 real channels)
f16* mt_vae_out(MtGraph* g, int* cbt);         // [N][1][65536][16] (3 real channels, RGB)
int mt_run(MtGraph* g, int nf, float* partial, size_t partial_cap, hipStream_t s);
// per-op view (profiling): ops in execution order; type 0 conv/linear, 1 GroupNorm, 2 LayerNorm, 3 attention, 4 GEGLU, 5 add-pos, 6 value transpose (hoisted cross-attention values)
int mt_op_count(MtGraph* g);
const char* mt_op_name(MtGraph* g, int i, int* type);
int mt_run_timed(MtGraph* g, int nf, float* partial, size_t partial_cap, hipStream_t s, std::vector<hipEvent_t>* evs);
// named intermediate (debug / parity): returns device pointer + geometry, or null
f16* mt_named(MtGraph* g, const char* name, int* C, int* ld, int* coff, int* H, int* W);
double mt_macs_per_frame(const MtGraph* g);
// VAE encoder graph (avatar preparation): image input = mt_latent_in() ([N][1][65536][16], RGB in [-1,1]),
// moments (mean | logvar, 8 channels at 32x32) = mt_unet_out()
int mt_build_vae_encoder_graph(MtGraph* g, const ltk_named_tensor* vae_sd, int n, int frames);
// Whisper encoder graph (one 30-s window per run): input log-mel tensor = mt_latent_in(), 5 hidden states
int mt_build_whisper_graph(MtGraph* g, const ltk_named_tensor* encoder_sd, int n);
f16* mt_whisper_state(MtGraph* g, int i, int* cbt, int* cb0);

}  // namespace ltk
