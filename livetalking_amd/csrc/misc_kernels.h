// HBM-bound helper kernels of the render hot path: input gather/pack, output
// head, mel-spectrogram, paste-back composite.  Host launch interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ltk {

typedef _Float16 f16;

constexpr int kPackMaxFrames = 256;

// Per-frame pointer tables of one launch: the bank face crop, the mel window and the output frame of every frame.  They live in
// DEVICE memory (ltk_engine::d_tab) so that the launch sequence of a pass has no per-call kernel arguments and can be replayed as a
// captured hipGraph; launch_upload_tables fills them from the host copies with one small launch (two above 128 frames) in front of
// the pass, on the pass's own stream (no host->device copy engine round trip).
struct FacePtrs {
    const uint8_t* p[kPackMaxFrames];
};

// wav2lip_avatar.py:125-134: face u8 BGR [256][256][3] -> fp16 [256][256][8] (pixel-major, one 16-byte item per pixel)
// = {masked b,g,r (rows >= 128 zero), b,g,r, 0, 0} / 255.
// `faces` is a DEVICE pointer
void launch_pack_faces(const FacePtrs* faces, int nframes, f16* x0, hipStream_t s);

struct MelPtrs {
    const float* p[kPackMaxFrames];   // per frame: float32 [80][16]
};

// mel float32 [80][16] per frame -> fp16 [B][80][16][8] (pixel-major, channel 0 = value).
// `mel` is a DEVICE pointer
void launch_pack_mel(const MelPtrs* mel, int nframes, f16* out, hipStream_t s);

// face6 float32 NCHW [B][6][256][256] -> fp16 [B][256][256][8] (test hook).
void launch_pack_face6_nchw(const float* face6, int nframes, f16* x0, hipStream_t s);

// wav2lip_v2.py:90-91 + wav2lip_avatar.py:138,145: 1x1 conv 32->3 + sigmoid;
// writes uint8 trunc(sigmoid*255) NHWC [B][256][256][3] and/or float32 sigmoid
// NCHW [B][3][256][256] (either may be null).
struct OutPtrs {
    uint8_t* p[kPackMaxFrames];       // per frame: uint8 [256][256][3] (null = skip)
};
// `out_u8` is a DEVICE pointer (or null)
void launch_head(const f16* x32, int x_ld, int nframes, const float* w3x32, const float* b3,
                 const OutPtrs* out_u8, float* out_f32_nchw, hipStream_t s);

// Face-encoder skip cache (knob FACE_CACHE): a bank frame's record holds the eight skip tensors of the face encoder back to back,
// each as its channel-block range [C/16][H][W][16] of fp16 (what the encoder's last layer of a block writes into the decoder's
// concat buffer).  One launch moves the records of `nframes` frames: dir 0 record -> concat buffers (a pass; `recs` = the pass's
// DEVICE table, carried in the FacePtrs slot: the bank crops are not read in this mode), dir 1 concat buffers -> records (build).
struct FeatGeom {
    f16* cat[8];                 // level k: first half of the tensor inside frame 0 of its concat buffer
    unsigned cat_stride[8];      // halfs between two frames of that buffer
    unsigned off[9];             // record offset of level k, in 16-byte items; off[8] = items per record
};
void launch_feat_copy(const FacePtrs* recs, int nframes, const FeatGeom& g, int dir, hipStream_t s);

// The three tables of one pass, contiguous in device memory.
struct DevTables {
    FacePtrs faces;
    MelPtrs mels;
    OutPtrs outs;
};
// host tables (any of them may be null: that table is left alone) -> *d_tab, entries [0, nframes), on stream s
void launch_upload_tables(const FacePtrs* faces, const MelPtrs* mels, const OutPtrs* outs, int nframes, DevTables* d_tab, hipStream_t s);

// fp16 CB16 (or [N][H][W][8] when ld <= 8) channel range (ld, coff, C) -> float32 NCHW (debug capture).
// Saturation scan (debug, knob SAT_CHECK): counts, over the channel-blocked tensor view [N][cbt][P][16] blocks [cb0, cb0 + CB), the
// halfs sitting AT the fp16 limit (|v| == 65504: what an epilogue clamp `fmed3f(t, -65504, 65504)` leaves behind) into ctr[0] and
// the non-finite ones (a kernel without a clamp overflowed) into ctr[1].  q8 != 0: e4m3 bytes of a [N][cbt][P][32] tensor
// (|v| == 448 / NaN).  One atomic per wave that found something.
void launch_sat_scan(const f16* x, int N, int cbt, int cb0, int CB, long long P, int q8, unsigned long long* ctr, hipStream_t s);

void launch_nhwc_to_nchw_f32(const f16* x, int N, int H, int W, int ld, int coff, int C, float* out, hipStream_t s);

// audio.py:45-51 melspectrogram columns + mel.py:56-63 window gather.
// pcm device float32 [n_samples]; win_start device int32 [n_win];
// out float32 [n_win][80][16].  basis float32 [80][401], lo/hi int32 [80].
void launch_mel(const float* pcm, int n_samples, const int32_t* win_start, int n_win, int col_min, int n_cols,
                const float* basis, const int32_t* lohi, float* out, hipStream_t s);

// wav2lip_avatar.py:141-147 paste_back_frame.
void launch_paste(const uint8_t* full, int H, int W, const uint8_t* pred256, int y1, int y2, int x1, int x2,
                  uint8_t* out, hipStream_t s);

// the same for up to kPasteBatch frames in one launch (the B composites of one inference_batch result): frame f pastes prediction
// pred0 + f * 256*256*3 onto full[f] with box (y1,y2,x1,x2)[f] and writes out0 + f * out_stride
constexpr int kPasteBatch = 16;
struct PasteBatch {
    const uint8_t* full[kPasteBatch];
    int y1[kPasteBatch], y2[kPasteBatch], x1[kPasteBatch], x2[kPasteBatch];
};
void launch_paste_batch(const PasteBatch& b, int n, int H, int W, const uint8_t* pred0, uint8_t* out0, size_t out_stride, hipStream_t s);

// musetalk_avatar.py:154-164 paste_back_frame (resize + paste into the crop + cv2.blendLinear under the mask).
void launch_paste_blend(const uint8_t* full, int H, int W, const uint8_t* pred256, int x1, int y1, int x2, int y2, int xs, int ys,
                        int xe, int ye, const uint8_t* mask, uint8_t* out, hipStream_t s);

// Frame egress (base_avatar.py:419-449 transition blend + watermark, BGR24 -> I420): see egress_kernels.hip
// launch_egress for n frames in one launch, no transition blend / cache: frame f reads src0 + f * src_stride, writes out0 + f * out_stride
void launch_egress_batch(const uint8_t* src0, size_t src_stride, int n, const uint8_t* wm, int wm_x, int wm_y, int wm_w, int wm_h, int wm_b,
                         int wm_g, int wm_r, uint8_t* out0, size_t out_stride, int H, int W, int i420, int chroma, hipStream_t s);
void launch_egress(const uint8_t* src, const uint8_t* prev, float w_prev, float w_src, uint8_t* cache, const uint8_t* wm, int wm_x,
                   int wm_y, int wm_w, int wm_h, int wm_b, int wm_g, int wm_r, uint8_t* out, int H, int W, int i420, int chroma,
                   hipStream_t s);

}  // namespace ltk
