// audio_mid (audio_fused.hip): audio_encoder.4 .. .8 of the Wav2Lip generator in one launch, one workgroup per frame.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace ltk {

typedef _Float16 f16;

struct AudioMidArgs {
    const f16* x; int x_stride;        // input: audio_encoder.3's output, CB16 [frame][4][27*16][16]; halfs between two frames
    f16* y; int y_stride;              // output: audio_encoder.8's output, CB16 [frame][8][9*6][16]
    const f16* w[6];                   // per layer [cout/32][cin/16][9 taps][64 lanes][8 halfs] (MFMA row-operand fragments)
    const float* scale[6];             // folded BatchNorm scale / shift per output channel
    const float* shift[6];
    int N;
};

struct AudioMidPlan {
    f16* d_w[6] = {nullptr};
    float* d_scale[6] = {nullptr};
    float* d_shift[6] = {nullptr};     // = d_scale + cout (one allocation)
    bool ready = false;
};

// weight[l]: fp32 [cout][cin][3][3] of audio_encoder.(3 + l) (l = 0 is packed but not used: layer .3 stays its own launch), the residual layers' identity already folded into the centre tap
// (w[c][c][1][1] += 1 / scale[c]); scale / shift: folded BatchNorm.  Returns 0 or a negative code (message in *err).
int audio_mid_pack(AudioMidPlan* p, const float* const weight[6], const float* const scale[6], const float* const shift[6], std::string* err);
void audio_mid_destroy(AudioMidPlan* p);
int audio_mid_launch(const AudioMidPlan& p, const f16* x, int x_stride, f16* y, int y_stride, int nframes, hipStream_t s, std::string* err);

}  // namespace ltk
