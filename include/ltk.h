/*
 * libltk_hip.so - MI355X (gfx950) native lip-sync render hot path.
 *
 * C ABI: plain pointers and sizes, no torch types.  Every function returns 0 on
 * success or a negative LTK_E_* code; ltk_last_error() returns a thread-local
 * message.  All entry points are thread-safe (the reference calls
 * inference_batch / paste_back_frame / run_step from three threads per session,
 * avatars/base_avatar.py:475-481).  Device pointers and streams come from the
 * host runtime (PyTorch-ROCm: tensor.data_ptr(), torch.cuda.current_stream()
 * .cuda_stream) or are owned by the engine; `stream` is a hipStream_t passed as
 * void* (NULL = the engine's own stream).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the upstream LiveTalking checkout).
 */
#ifndef LTK_H
#define LTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTK_OK 0
#define LTK_E_INVALID (-1)   /* bad argument / shape / missing tensor */
#define LTK_E_HIP (-2)       /* a HIP runtime call failed */
#define LTK_E_STATE (-3)     /* call order (model not loaded, avatar unknown, ...) */
#define LTK_E_NOMEM (-4)

typedef struct ltk_engine ltk_engine;

/* One named fp32 host tensor of a PyTorch state_dict (row-major, torch layout). */
typedef struct ltk_named_tensor {
    const char* name;
    const float* data;
    int ndim;
    const int64_t* shape;
} ltk_named_tensor;

/* One session's request inside a cross-session batch: `batch` consecutive
 * frames starting at running frame index `index` (ping-pong `mirror_index`
 * over the avatar's bank, utils/image.py:26-32), mel windows at d_mel. */
typedef struct ltk_w2l_req {
    int avatar;            /* id returned by ltk_avatar_register */
    int index;             /* BaseAvatar.inference's running `index` (base_avatar.py:328,366) */
    int batch;             /* frames in this request (opt.batch_size) */
    const void* d_mel;     /* device, float32 [batch][80][16] (what ltk_mel_step wrote) */
    void* d_pred;          /* device, uint8 [batch][256][256][3] BGR: this request's frames */
} ltk_w2l_req;

const char* ltk_last_error(void);
const char* ltk_version(void);

/* utils/device.py:4-10 initialize_device + per-process model ownership
 * (app.py:62,140-151): one engine per GPU. */
int ltk_engine_create(int device, ltk_engine** out);
void ltk_engine_destroy(ltk_engine* e);
int ltk_engine_sync(ltk_engine* e);

/* avatars/wav2lip_avatar.py:59-70 load_model: takes checkpoint["state_dict"]
 * (380 fp32 tensors, names as in avatars/wav2lip/models/wav2lip_v2.py with any
 * "module." prefix already stripped), folds eval-mode BatchNorm
 * (conv.py:7-10,36-39) into per-channel scale/shift, repacks the conv kernels
 * to fp16 MFMA tiles and sizes the activation arena for `max_frames` frames
 * per launch (the sum of all requests of one ltk_wav2lip_infer call). */
int ltk_wav2lip_load(ltk_engine* e, const ltk_named_tensor* sd, int n, int max_frames);

/* avatars/wav2lip_avatar.py:72-88 load_avatar: uploads one avatar bank.
 * face_bank  uint8 [n][256][256][3] BGR (face_imgs/), full_bank uint8
 * [n][H][W][3] BGR (full_imgs/), coords int32 [n][4] = (y1,y2,x1,x2)
 * (coords.pkl, avatars/wav2lip/genavatar.py:130).  Host pointers. */
int ltk_avatar_register(ltk_engine* e, const uint8_t* face_bank, const uint8_t* full_bank,
                        const int32_t* coords, int n, int H, int W, int* avatar_id);
/* Drops a bank registered by ltk_avatar_register or ltk_musetalk_avatar_register (ids of both kinds come from one counter).
 * Safe while other threads render from it: every call holds a reference to its bank until it returns, the device buffers are
 * freed when the last of them does. */
int ltk_avatar_release(ltk_engine* e, int avatar_id);

/* avatars/audio_features/mel.py:43-63 (MelASR.run_step feature part) +
 * avatars/wav2lip/audio.py:45-51 melspectrogram: `n_samples` PCM samples (the
 * concatenated l + 2B + r 20-ms chunks, host float32) -> `n_win` windows of
 * (80,16) float32 written to d_out [n_win][80][16]; window i starts at STFT
 * column win_start[i] (mel.py:56).  hop 200 / n_fft 800 / 80 mels
 * (avatars/wav2lip/hparams.py:33-73). */
int ltk_mel_step(ltk_engine* e, const float* pcm, int n_samples, const int32_t* win_start,
                 int n_win, void* d_out, void* stream);

/* avatars/wav2lip_avatar.py:116-139 LipReal.inference_batch for `nreq`
 * sessions at once: bank gather + lower-half mask + 6-channel pack, the 55
 * conv/convT layers of Wav2Lip.forward (wav2lip_v2.py:123-163), sigmoid*255 and
 * the uint8 truncation paste_back_frame applies (wav2lip_avatar.py:138,145).
 * Each request's frames land in its own d_pred.  Returns when they are ready;
 * an error return likewise leaves none of the call's launches in flight (the
 * caller may free d_pred / d_mel at once), and a face-encoder prefetch for the
 * session's NEXT call (knob PREFETCH) that cannot be launched is not an error. */
int ltk_wav2lip_infer(ltk_engine* e, const ltk_w2l_req* reqs, int nreq, void* stream);

/* avatars/wav2lip_avatar.py:141-147 LipReal.paste_back_frame: bilinear-resize
 * (cv2.resize INTER_LINEAR semantics) the 256x256 prediction to the frame's box
 * and paste it into a copy of full_bank[idx].  d_pred: device uint8
 * [256][256][3].  out: uint8 [H][W][3]; out_is_device selects a device buffer
 * or a (pinned or pageable) host buffer. */
int ltk_paste_back(ltk_engine* e, int avatar_id, int idx, const void* d_pred, void* out,
                   int out_is_device, void* stream);

/* The same composite for the `n` frames of one inference_batch result at once (the process thread of
 * avatars/base_avatar.py:383-467 calls paste_back_frame once per frame, in order, for the items inference_batch returned):
 * d_pred = n contiguous device uint8 [256][256][3] predictions, idx[i] = bank frame of prediction i, out = HOST uint8
 * [n][H][W][3] - pinned memory moves at the PCIe rate.  n composites on the device, ONE device-to-host copy, one
 * synchronisation (per frame: ltk_paste_back costs a 2.76 MB pageable copy and a stream synchronisation each). */
int ltk_paste_back_batch(ltk_engine* e, int avatar_id, const int32_t* idx, const void* d_pred, int n, void* out, void* stream);

/* =========================== MuseTalk path (avatars/musetalk_avatar.py) =========================== */

/* avatars/musetalk_avatar.py:57-67 load_model + avatars/musetalk/utils/utils.py:16-37 load_all_model: the
 * conditional U-Net (diffusers UNet2DConditionModel state_dict, MuseTalk-1.5 config) and the AutoencoderKL
 * decoder of sd-vae (state_dict keys "post_quant_conv.*", "decoder.*"), both fp32 host tensors under their
 * diffusers names.  The timestep-0 embedding, attention scale, VAE scaling factor and the sinusoidal
 * PositionalEncoding (avatars/musetalk/models/unet.py:12-27) are folded / precomputed here; arenas are sized
 * for `max_frames` frames per launch. */
int ltk_musetalk_load(ltk_engine* e, const ltk_named_tensor* unet_sd, int n_unet, const ltk_named_tensor* vae_sd,
                      int n_vae, int max_frames);

/* avatars/musetalk_avatar.py:69-91 load_avatar.  latents: fp32 [n][8][32][32] (input_latent_list_cycle,
 * latents.pt); full_bank uint8 [n][H][W][3] BGR; face_boxes int32 [n][4] = (x1,y1,x2,y2) (coords.pkl,
 * musetalk_avatar.py:157); crop_boxes int32 [n][4] = (x_s,y_s,x_e,y_e) (mask_coords.pkl); masks: the n blend
 * masks (mask/<i>.png, uint8 [h_i][w_i][3] with h_i = y_e-y_s, w_i = x_e-x_s) concatenated, mask i starting at
 * byte mask_offsets[i] (n+1 offsets).  Boxes must lie inside the frame.  Host pointers. */
int ltk_musetalk_avatar_register(ltk_engine* e, const float* latents, const uint8_t* full_bank, const int32_t* face_boxes,
                                 const int32_t* crop_boxes, const uint8_t* masks, const int64_t* mask_offsets, int n,
                                 int H, int W, int* avatar_id);

/* BASELINE configs[4] "fp8 conv path": before ltk_musetalk_load, ask for the ResnetBlock2D 3x3 convolutions of the U-Net
 * and the VAE decoder (the GroupNorm -> SiLU -> conv pairs) to run on OCP e4m3 operands: the GroupNorm kernel writes
 * saturate(y * act_scale) as fp8, weights are quantised per output channel (224 / max|w|), accumulation stays fp32 and
 * the residual stream fp16.  act_scale <= 0 selects the default 8 (e4m3 then covers |y| <= 56 with 2^-12 resolution
 * near zero).  Everything else (attention, linears, up/down-samplers, conv_in/out) stays fp16. */
int ltk_musetalk_set_fp8(ltk_engine* e, int enable, float act_scale);
/* conv / linear MACs per frame of the loaded U-Net + VAE decoder, and the part of them on fp8 operands */
int ltk_musetalk_info(ltk_engine* e, double* macs_per_frame, double* macs_fp8_per_frame);

typedef struct ltk_mt_req {
    int avatar;            /* id returned by ltk_musetalk_avatar_register */
    int index;             /* running frame index (mirror_index over the latent bank) */
    int batch;
    const void* d_feat;    /* device, float32 [batch][50][384]: the whisper chunks of WhisperASR (before PE) */
    void* d_pred;          /* device, uint8 [batch][256][256][3] BGR */
} ltk_mt_req;

/* avatars/musetalk_avatar.py:130-152 MuseReal.inference_batch for `nreq` sessions at once: latent gather,
 * positional encoding, U-Net (timestep 0), vae.decode_latents incl. the uint8 rounding and RGB->BGR flip. */
int ltk_musetalk_infer(ltk_engine* e, const ltk_mt_req* reqs, int nreq, void* stream);

/* avatars/musetalk_avatar.py:154-164 paste_back_frame + avatars/musetalk/myutil.py:4-25 get_image_blending:
 * resize the 256x256 prediction to the face box, paste into the crop region and cv2.blendLinear it with the
 * cached frame under the avatar's mask.  out: uint8 [H][W][3] (device or host, as ltk_paste_back). */
int ltk_paste_blend(ltk_engine* e, int avatar_id, int idx, const void* d_pred, void* out, int out_is_device, void* stream);

/* ---- frame egress: the steps between paste_back_frame and the encoder ------------
 * Reference: avatars/base_avatar.py:384-453 process_frames (silent path :407-428, transition blend :419-426 and
 * :436-445, watermark :449), server/webrtc.py:190-193 (VideoFrame.from_ndarray(bgr24), converted to yuv420p by the
 * encoder's swscale), streamout/rtmp.py:81-83.
 *
 * One egress session per render session: it owns the two cached frames of the transition effect
 * (_last_silent_frame / _last_speaking_frame) and the watermark. */
typedef struct ltk_egress ltk_egress;
int ltk_egress_open(ltk_engine* e, int H, int W, ltk_egress** out);
int ltk_egress_close(ltk_engine* e, ltk_egress* s);
/* Watermark as a coverage bitmap: cv2.putText(..., thickness 1, LINE_8) sets every covered pixel to the colour, so
 * the host rasterises the text once (cv2.putText on a zero image) and hands the non-zero rectangle over.
 * mask host uint8 [h][w] (non-zero = covered) placed at (x, y); NULL removes the watermark. */
int ltk_egress_watermark(ltk_engine* e, ltk_egress* s, const uint8_t* mask, int x, int y, int w, int h, int b, int g, int r);

#define LTK_FMT_BGR24 0
#define LTK_FMT_I420 1      /* Y [H][W], U [H/2][W/2], V [H/2][W/2]; H and W even; BT.601 limited range, swscale's integer matrix */
#define LTK_SRC_WAV2LIP 0   /* composite = ltk_paste_back(avatar, idx, d_pred); d_pred NULL = the cached full frame (silent) */
#define LTK_SRC_MUSETALK 1  /* composite = ltk_paste_blend(avatar, idx, d_pred); d_pred NULL = the cached full frame */
#define LTK_SRC_HOST 2      /* frame = h_frame, host uint8 [H][W][3] (custom action video, base_avatar.py:411-414) */
typedef struct ltk_egress_req {
    int source;             /* LTK_SRC_* */
    int avatar, idx;        /* bank frame (ignored for LTK_SRC_HOST) */
    const void* d_pred;     /* device uint8 [256][256][3] or NULL */
    const uint8_t* h_frame; /* LTK_SRC_HOST only */
    int speaking;           /* which cache this frame refreshes: 1 = _last_speaking_frame, 0 = _last_silent_frame */
    double alpha;           /* transition weight of THIS frame: out = addWeighted(other-state cache, 1-alpha, frame, alpha);
                             * < 0 or >= 1 (or no cached frame of the other state yet): no blend */
    int keep;               /* non-zero: store the (blended, un-watermarked) frame as this state's cache, as the reference
                             * does while enable_transition is on */
    int format;             /* LTK_FMT_* */
    int chroma;             /* I420 chroma: 1 = 2x2 mean, 0 = top-left pixel of each quad */
} ltk_egress_req;
/* h_out: host uint8, H*W*3 bytes (BGR24) or H*W*3/2 bytes (I420); returns after the copy completed. */
int ltk_egress_frame(ltk_engine* e, ltk_egress* s, const ltk_egress_req* req, uint8_t* h_out, void* stream);

/* The speaking frames of ONE inference_batch result at once, for sessions without the transition effect (the reference's
 * default, base_avatar.py:384 enable_transition = False): n composites (ltk_paste_back / ltk_paste_blend of prediction i onto
 * bank frame idx[i]), watermark and format conversion on the device, ONE device-to-host copy into `h_out`
 * ([n][H][W][3] for LTK_FMT_BGR24, [n][H*3/2][W] for LTK_FMT_I420; pinned memory moves at the PCIe rate) and one
 * synchronisation.  With I420 a 25-fps 720p session costs 35 MB/s of PCIe instead of 69: the link, not the GPU, bounds
 * how many sessions a GPU can serve (bench.py `delivered`).  source: LTK_SRC_WAV2LIP or LTK_SRC_MUSETALK; d_pred: n
 * contiguous device uint8 [256][256][3] predictions.  The session's transition caches are neither read nor written. */
int ltk_egress_batch(ltk_engine* e, ltk_egress* s, int source, int avatar, const int32_t* idx, const void* d_pred, int n, int format,
                     int chroma, uint8_t* h_out, void* stream);

/* avatars/musetalk/whisper/audio2feature.py:15-23 Audio2Feature.__init__: the Whisper-tiny ENCODER
 * (transformers WhisperModel(...).encoder.state_dict(): conv1, conv2, embed_positions, layers.{0..3}.*, layer_norm),
 * fp32 host tensors.  The WhisperFeatureExtractor constants (n_fft 400, hop 160, 80 slaney mels, 30-s padding) are
 * built in. */
int ltk_whisper_load(ltk_engine* e, const ltk_named_tensor* encoder_sd, int n);

/* avatars/audio_features/whisper.py:58-76 WhisperASR.run_step feature part: audio2feat
 * (audio2feature.py:106-117: log-mel of the zero-padded 30-s window -> encoder with all 5 hidden states) and
 * _feature2chunks (whisper.py:35-56, base_asr.py:91-133): frame i takes encoder rows
 * [first_row + i*row_step, first_row + i*row_step + rows), index-clamped, each row = its 5 hidden states.
 * pcm: host float32 [n_samples] (the concatenated l + 2B + r chunks).  d_out: device float32
 * [batch][rows*5][384] (= the (50,384) whisper chunks at rows 10, first_row 2*(l/2), row_step 2). */
int ltk_whisper_step(ltk_engine* e, const float* pcm, int n_samples, int batch, int first_row, int row_step, int rows,
                     void* d_out, void* stream);

/* Avatar preparation: avatars/musetalk/models/vae.py:84-94,110-122 get_latents_for_unet, as
 * avatars/musetalk/genavatar.py:116-128 calls it.  vae_sd: the AutoencoderKL state_dict keys "encoder.*" and
 * "quant_conv.*" (fp32 host).  faces_bgr: host uint8 [nfaces][256][256][3] (the LANCZOS-resized crops);
 * latents_out: host fp32 [nfaces][8][32][32] = cat(masked, reference) latents * scaling_factor.
 * noise: host fp32 [nfaces][2][4][32][32] standard-normal draws for latent_dist.sample() (the reference draws them
 * from torch's global RNG), or NULL for the distribution mean. */
int ltk_vae_encoder_load(ltk_engine* e, const ltk_named_tensor* vae_sd, int n, int max_faces);
int ltk_vae_encode_faces(ltk_engine* e, const uint8_t* faces_bgr, int nfaces, const float* noise, float* latents_out);

/* ---- test / measurement hooks (not on the production call path) ---- */

/* named tensor of the last Whisper step as [C][T] float32: "input_features", "hidden_states.0".."hidden_states.4",
 * or an op name ("conv1", "layers.0.self_attn.out_proj", ...) */
int ltk_whisper_debug_get(ltk_engine* e, const char* name, float* out, size_t n_floats);

/* U-Net + VAE decoder on explicit inputs: latents host fp32 [B][8][32][32], feat host fp32 [B][50][384] (before
 * the positional encoding).  Outputs (any may be NULL): unet_out fp32 [B][4][32][32], image fp32 [B][3][256][256]
 * (AutoencoderKL.decode sample, RGB), frames uint8 [B][256][256][3] BGR. */
int ltk_musetalk_forward_host(ltk_engine* e, const float* latents, const float* feat, int B, float* unet_out, float* image,
                              uint8_t* frames);
/* copy a named intermediate of the last MuseTalk forward as NCHW float32 (first `frames` frames) */
int ltk_musetalk_debug_get(ltk_engine* e, const char* name, int frames, float* out, size_t n_floats);
/* average milliseconds of one U-Net + VAE pass over `frames` frames, and its conv/linear MACs */
int ltk_musetalk_time(ltk_engine* e, int frames, int iters, float* ms_per_pass, double* macs_per_pass);
/* Per-op view of the MuseTalk launch program (profiling): op names in execution order (diffusers module paths), type 0 conv /
 * linear, 1 GroupNorm, 2 LayerNorm, 3 attention, 4 GEGLU, 5 add-pos; time of every op inside a whole pass (HIP events between
 * consecutive ops on the compute stream). */
int ltk_musetalk_op_count(ltk_engine* e);
int ltk_musetalk_op_name(ltk_engine* e, int op, char* buf, int buf_len, int* type);
int ltk_musetalk_time_ops(ltk_engine* e, int frames, int iters, float* ms_per_op, int n_ops);


/* Run Wav2Lip.forward on explicit inputs: mel host float32 [B][80][16], face6
 * host float32 [B][6][256][256] in [0,1] (as wav2lip_avatar.py:133-134 builds
 * them); pred host float32 [B][3][256][256] = sigmoid output (before *255). */
int ltk_wav2lip_forward_host(ltk_engine* e, const float* mel, const float* face6, int B, float* pred);

/* After a forward with capture enabled, copy one layer's activation
 * (state_dict prefix, e.g. "face_encoder_blocks.1.0") as NCHW float32. */
/* Saturation counters (debug; knob SAT_CHECK = 1, environment LTK_SAT_CHECK or ltk_debug_set_knob).  The conv epilogues clamp to
 * the fp16 range (fmed3f(t, -65504, 65504)): a network whose activations reach it is no longer computing the reference's fp32
 * numbers, and the fp16 parity tolerance does not hold from there on.  With the knob on, every layer's / op's output (Wav2Lip
 * layers, MuseTalk ops incl. the e4m3 tensors of the fp8 path) is scanned as it is produced: `n_at_limit` = values exactly at the
 * limit of their type (what a clamp leaves behind), `n_nonfinite` = inf / NaN (a kernel without a clamp overflowed).  Waits for
 * the device; `reset` != 0 zeroes the counters afterwards.  Reference: the reference runs these networks in fp32 on the CPU and
 * fp16 autocast on CUDA (avatars/wav2lip_avatar.py:51-70, avatars/musetalk_avatar.py:57-66); it has no such check. */
int ltk_debug_saturation(ltk_engine* e, int reset, unsigned long long* n_at_limit, unsigned long long* n_nonfinite);
int ltk_debug_capture(ltk_engine* e, int enable);
int ltk_debug_get(ltk_engine* e, const char* layer, float* out, size_t n_floats);

/* Tuning / A-B knobs (livetalking_amd/csrc/tune.h).  Every knob is read from the environment once per process; this call
 * changes one in-process (sweep scripts, tests).  `name` with or without the LTK_ prefix.  Process-wide, not per engine:
 * knobs that shape a plan (weight pack order) only affect plans created afterwards. */
int ltk_debug_set_knob(const char* name, int value);

/* Device-free consistency check of the engine's measured per-layer tile table (csrc/engine.hip kTileTable, knob TILE_TABLE):
 * every entry must name an existing layer, a frame-count bucket and a tile / split conv3 has an instantiation for.  Returns
 * the number of bad entries (0 = consistent) and writes their descriptions into `msg` (NUL-terminated, at most `cap` bytes). */
int ltk_debug_tile_table_check(char* msg, int cap);

/* The device side of one ltk_wav2lip_infer pass, exactly as that call enqueues it (mel pack, conv stack with the bank gather
 * and the output head fused, replayed from the captured hipGraph under knob GRAPH), on dummy inputs, timed with HIP events
 * on the engine's compute stream: used by bench.py for `roofline.achieved`.  Returns average milliseconds per pass over
 * `iters` passes of `frames` frames, and the number of conv/convT MACs one pass executes (27,788,599,296 x frames for
 * wav2lip256). */
int ltk_wav2lip_time_convs(ltk_engine* e, int frames, int iters, float* ms_per_pass, double* macs_per_pass);

/* Number of frame counts whose pass currently runs from a captured hipGraph (knob GRAPH; a frame count is captured the
 * second time ltk_wav2lip_infer sees it).  Tests and bench.py use it to prove that the graph path is the one that ran. */
int ltk_wav2lip_graph_count(ltk_engine* e);

/* Opt-in deployment mode, knob FACE_CACHE (environment LTK_FACE_CACHE=1 or ltk_debug_set_knob): the Wav2Lip face encoder reads
 * the bank frame only (avatars/wav2lip/models/wav2lip_v2.py:132-140: `feats` come from the masked + reference crop, the audio
 * enters at the decoder), so its eight skip tensors are computed once per avatar - on the avatar's first ltk_wav2lip_infer call,
 * 4.15 MB of fp16 per bank frame, resident in HBM - and a pass copies them into the decoder's concat buffers instead of running
 * conv7 + 20 encoder layers.  Frames are byte-identical to the mode off for 16-frame calls (the cache is built by 16-frame
 * launches), within 1 LSB for other call sizes, byte-identical for every size under LTK_SPLITK=0.  bench.py never times this
 * mode on its headline line (cached outputs are skipped work there); it reports it on its own also[] entry.  One avatar's cache
 * is limited to LTK_FACE_CACHE_MAX_MB (default 16384; a call for a longer avatar fails with LTK_E_NOMEM, nothing allocated); the
 * first call of an avatar after the mode was switched off frees its records.  The getter may be polled from any thread.
 * Returns the bytes of skip cache the avatar currently holds (0 = none built). */
int ltk_avatar_face_cache_bytes(ltk_engine* e, int avatar_id, size_t* bytes);

/* Knob PREFETCH (default on; LTK_PREFETCH=0 switches it off): software pipelining ACROSS the calls of one session.  The face
 * encoder reads the bank frame only (wav2lip_v2.py:132-140) and a session walks its bank in order (base_avatar.py:366-376:
 * inference_batch(index, ...), index advancing by one per frame), so when a single-request call of <= 32 frames continues
 * the previous one (same avatar, index = previous index + batch) the engine runs, beside that call's audio encoder + decoder and
 * on a third stream, the face encoder of the frames the NEXT call will ask for, into the other set of concat buffers; the next
 * call then starts at the decoder.  Every layer still runs once per frame and step and the frames are byte-identical to the
 * knob off (same kernels, same launch shapes).  The prefetch is its own hipGraph on the engine's third stream, replayed right behind
 * the call's own graph.  A call that does not continue the sequence (another session, a jump of the
 * index, a multi-request call) runs the whole pass and the unused prefetch is dropped.  Counters since engine creation:
 * single-request calls that found their encoder outputs prefetched / that did not / prefetches issued. */
int ltk_wav2lip_prefetch_stats(ltk_engine* e, unsigned long long* hits, unsigned long long* misses, unsigned long long* issued);

/* Number of (program, frame count) pairs of the MuseTalk side - the U-Net + VAE decoder pass behind ltk_musetalk_infer, the
 * Whisper encoder behind ltk_whisper_step - that currently run from a captured hipGraph (knob GRAPH; captured the second time
 * a frame count is seen). */
int ltk_program_graph_count(ltk_engine* e);

/* Per-layer view of the same pass (tuning / profiling): layer names in execution order (state_dict prefixes), the time
 * of every layer inside a whole pass on one stream (HIP events between consecutive launches, so a layer sees the cache
 * state its predecessor left, not the hot loop of an isolated microbenchmark), and the engine's per-layer tile table:
 * bucket 0..4 = launches of <= 16 / 32 / 64 / 128 / more frames; pxw in {0,1,2,4}, nbt in {0,1,2}, ksplit >= 0; 0 = rule. */
int ltk_wav2lip_layer_count(ltk_engine* e);
int ltk_wav2lip_layer_name(ltk_engine* e, int layer, char* buf, int buf_len);
int ltk_wav2lip_set_layer_tile(ltk_engine* e, int layer, int bucket, int pxw, int nbt, int ksplit);
int ltk_wav2lip_time_layers(ltk_engine* e, int frames, int iters, float* ms_per_layer, int n_layers);

/* Generic standalone fp16 conv used by kernel unit tests and per-layer timing.  Activations are in the engine's
 * channel-blocked layout: x device fp16 [N][Cin/16][H][W][16] ([N][H][W][8] when Cin <= 8), res / y device fp16
 * [N][Cout/16][Ho][Wo][16]; weight host fp32 torch layout ([Cout][Cin][kh][kw], or [Cin][Cout][kh][kw] when
 * transposed), scale/shift host fp32 [Cout] (NULL = 1 / 0).  livetalking_amd/layout.py converts from NCHW. */
int ltk_conv2d_f16(ltk_engine* e, const void* d_x, int N, int H, int W, int Cin,
                   const float* weight, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                   int transposed, int out_pad, const float* scale, const float* shift,
                   const void* d_res, int relu, void* d_y, int iters, float* ms_avg);

/* Standalone GroupNorm (+ SiLU) over a channel-blocked fp16 tensor, for kernel unit tests and timing (diffusers GroupNorm as the MuseTalk U-Net / VAE
 * use it; `avatars/musetalk/models/unet.py:36-46`, `vae.py:96-108` call sites).  x, y: device fp16 [N][C/16][P][16]; gamma / beta host fp32 [C];
 * impl: 0 = what the MuseTalk program would pick for this shape, 1 = gn_stats + gn_apply (two launches, three tensor passes), 2 = gn_group_kernel
 * (one block per (image, group)), 3 = gn_coop_kernel (one tensor pass, blocks exchange partial sums); an impl that does not serve the shape is refused.
 * out_fp8 != 0: y is e4m3 [N][C/32][P][32] = min(max(result * out_scale, -448), 448) (the operand format of the fp8 conv path).
 * iters > 0: additionally timed over `iters` back-to-back runs. */
int ltk_groupnorm_f16(ltk_engine* e, const void* d_x, int N, int C, int P, int groups, float eps, const float* gamma, const float* beta,
                      int silu, int impl, int out_fp8, float out_scale, void* d_y, int iters, float* ms_avg);

/* host-side fp32 -> OCP e4m3fn conversion of the weight packer (round to nearest even, saturating at +-448); no GPU needed */
int ltk_f32_to_e4m3(const float* in, size_t n, uint8_t* out);

/* fp8-operand 3x3 stride-1 pad-1 conv used by kernel unit tests and per-layer timing: x device e4m3 bytes
 * [N][Cin/32][H][W][32] holding round(x * act_scale), weight host fp32 [Cout][Cin][3][3] (quantised per output channel
 * by the engine), y = act((conv(x, w)) * scale + shift + res) as fp16 [N][Cout/16][H][W][16]; act: 0 none, 1 ReLU,
 * 2 GELU, 3 SiLU. */
int ltk_conv2d_fp8(ltk_engine* e, const void* d_x, int N, int H, int W, int Cin, const float* weight, int Cout,
                   const float* scale, const float* shift, float act_scale, const void* d_res, int act, void* d_y, int iters,
                   float* ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* LTK_H */
