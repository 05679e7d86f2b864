// Tuning / A-B knobs of the engine.  Every knob is an integer read from the environment ONCE per process (first use,
// normally ltk_engine_create) and kept in a table; the launch path reads the table, never getenv.  Sweep scripts and
// tests change a knob in-process through ltk_debug_set_knob (include/ltk.h).
#pragma once

namespace ltk {

enum Knob {
    K_CONV_V3 = 0,        // LTK_CONV_V3        1: conv3 (LDS-DMA) kernels where they apply
    K_CONV_V3_S2,         // LTK_CONV_V3_S2     1: conv3 stride-2 for deep layers, 2: for all, 0: never
    K_CONV_NBT,           // LTK_CONV_NBT       conv_mfma: cap of 32-cout subtiles per block (0 = plan's choice)
    K_CONV_NC8,           // LTK_CONV_NC8       conv_mfma: channel planes per chunk (0 = plan's choice)
    K_GEMM_NC8,           // LTK_GEMM_NC8       conv3 1x1: 4 = 32-channel chunks for wide outputs
    K_CONV_MODE,          // LTK_CONV_MODE      conv_mfma: 0 double-buffered LDS, 1 single buffer + register prefetch
    K_CONV_MIN_BLOCKS,    // LTK_CONV_MIN_BLOCKS conv_mfma: narrow the block until the grid has this many blocks
    K_CONV_PXW,           // LTK_CONV_PXW       conv3: force 2 (256-pixel tiles); 0 = heuristic
    K_CONV3_NBT,          // LTK_CONV3_NBT      conv3: force 32-cout subtiles per block (1, 2, 4); 0 = heuristic
    K_CONV_PXW4_MIN,      // LTK_CONV_PXW4_MIN  conv3: 512-pixel tiles only when they still give this many items
    K_SPLITK,             // LTK_SPLITK         0: never split the channel loop (batch-size independent summation order)
    K_KSPLIT,             // LTK_KSPLIT         conv3: force this split factor (sweeps); 0 = heuristic
    K_CONV_PERSIST,       // LTK_CONV_PERSIST   conv3: resident grid size above which a launch walks items persistently
    K_NO_FOLD_RESIDUAL,   // LTK_NO_FOLD_RESIDUAL 1: keep the residual read instead of folding it into the centre tap
    K_NO_FLATTEN,         // LTK_NO_FLATTEN     1: run the 4x4 valid conv as a conv, not as a flattened 1x1
    K_NO_AUX_STREAM,      // LTK_NO_AUX_STREAM  1: audio encoder on the compute stream
    K_MICROBATCH,         // LTK_MICROBATCH     wav2lip frames per arena pass (0 = min(max_frames, 256))
    K_MT_NO_QKV_FUSE,     // LTK_MT_NO_QKV_FUSE 1: separate q / k / v projection launches
    K_HEAD_FUSED,         // LTK_HEAD_FUSED     1: output_block conv 80->32 + 1x1 head + sigmoid in one launch
    K_CONV3_NC8,          // LTK_CONV3_NC8      conv3 3x3: channel planes per chunk (2 = 16 channels, 4 = 32); 0 = by map size
    K_TILE_RULE,          // LTK_TILE_RULE      conv3 3x3 tile selection: 1 = items-per-CU rule (round 2), 0 = round-1 heuristic (A/B)
    K_TILE_TABLE,         // LTK_TILE_TABLE     1: use the engine's per-layer measured tile table where it has an entry
    K_CONV7,              // LTK_CONV7          1: dedicated first-layer kernel (7x7, 6 -> 16) with the input pack fused
    K_ATTN_WIDE,          // LTK_ATTN_WIDE      1: cooperative kernel for the 512-channel single-head attention of the VAE mid block
    K_UPS4,               // LTK_UPS4           1: nearest-2x upsample + 3x3 conv as four 2x2-tap phases (16 instead of 36 MACs per source pixel)
    K_FP8_MX,             // LTK_FP8_MX         fp8 convs on the MX-scaled MFMA (32x32x64, 2x MAC rate): 1 = where it wins (Cin >= 512:
                          //                    its 64-channel chunks leave one block per CU, which only deep K loops repay), 2 = all, 0 = none
    K_ROWGEMM,            // LTK_ROWGEMM        1: the one-pixel-map layers of a <= 32-frame launch as skinny GEMMs (rowgemm.hip) instead of conv3 + split-K finish
    K_ROWCONV,            // LTK_ROWCONV        the 3x3 layers on the 4x4 / 8x8 maps as weight-streaming GEMMs over gathered rows (rowgemm.hip rowconv) when the
                          //                    launch has at most this many output pixels (frames x Ho x Wo); 0 = never
    K_ABLATE,             // LTK_ABLATE         measurement builds only (make ABLATE=1): bit mask, see conv3_mfma.hip
    K_GRAPH,              // LTK_GRAPH          non-zero (default 1): a Wav2Lip pass of a given frame count is captured once as a hipGraph and replayed (the per-call
                          //                    pointer tables live in device memory, filled by one small launch in front of the graph); 0: launch by launch
    K_DF_FRAMES,          // LTK_DF_FRAMES      > 0: the decoder blocks >= LTK_DF_BLOCK and the output conv run depth-first over sub-batches of this
                          //                    many frames (producer -> consumer tensors stay in the 256 MiB Infinity Cache); 0 = layer by layer
    K_DF_BLOCK,           // LTK_DF_BLOCK       first decoder block of the depth-first region (6: the 128^2 and 256^2 levels)
    K_DF_MIN,             // LTK_DF_MIN         depth-first only for launches of at least this many frames
    K_ROWCONVT,           // LTK_ROWCONVT       the stride-2 transposed convs on the 4x4 / 8x8 maps as per-phase weight-streaming GEMMs (rowconvT_launch)
                          //                    when the launch has at most this many SOURCE pixels (frames x H x W), instead of conv3's merged-phase
                          //                    items + split-K finish; 0 = never.  512: measured (profiles/r04_rowconvT_ab.txt) - 256 rows -7.5 us, 1024 rows +-0
    K_MT_ROWCONV,         // LTK_MT_ROWCONV     MuseTalk: the 1x1 / linear layers (<= 2560 outputs) on maps of <= 64 pixels / tokens as weight-streaming GEMMs over gathered rows
                          //                    (rowconv) when the launch has at most this many rows (frames x pixels); 0 = never (and no plans are built)
    K_MT_TILE_TABLE,      // LTK_MT_TILE_TABLE  1: measured per-level conv3 tile width for the U-Net's 3x3 convs in passes of <= 16 frames (musetalk.hip mt_graph_run)
    K_LDS_SWZ,            // LTK_LDS_SWZ        1: conv3's stride-1 LDS image takes the row-parity key on tiles narrower than 32 pixels (conflict-free
                          //                    ds_read_b128 on 16- / 8-pixel-wide maps); 0: column key everywhere (rounds 1-3)
    K_FACE_CACHE,         // LTK_FACE_CACHE     1 (opt-in deployment mode, default 0): the face encoder's eight skip tensors depend on the BANK frame only
                          //                    (wav2lip_v2.py:132-140), so they are computed once per avatar (4.15 MB of fp16 per bank frame, resident in HBM)
                          //                    and a pass copies them into the decoder's concat buffers instead of running conv7 + 20 encoder layers
    K_PREFETCH,           // LTK_PREFETCH       1 (default): a session's consecutive single-request calls (index advancing by its batch size, <= 32 frames) are
                          //                    software-pipelined across calls: while call N runs its audio encoder + decoder, the face encoder of the frames call
                          //                    N+1 will ask for (bank frames index+B ..) runs beside it on a third stream into the other set of concat buffers; call
                          //                    N+1 then starts at the decoder.  Every layer still runs once per frame and step; a call that does not continue the
                          //                    sequence runs the whole pass.  0: every call runs the whole pass (rounds 1-4)
    K_AUDIO_ROWCONV,      // LTK_AUDIO_ROWCONV  audio-encoder 3 x 3 layers whose output map has at most this many pixels per frame run as weight-streaming GEMMs over
                          //                    gathered rows (rowconv, row / column strides) in launches of <= ROWCONV rows: 54 (default) = audio_encoder.6 .. .10,
                          //                    9 = .9 / .10 only, 0 = none (first-generation kernel / conv3 + split-K finish: rounds 1-4)
    K_MT_FUSE,            // LTK_MT_FUSE        MuseTalk program, read when the program is BUILT (weights are packed for it): bit 0 = GEGLU in the epilogue of
                          //                    ff.net.0.proj (no 8C-wide intermediate, no geglu launch); bit 1 = the k | v projections of the 16 cross-attentions
                          //                    (they read the audio context only) as ONE stacked projection + ONE value-transpose launch at the head of the pass;
                          //                    bit 2 = the transformer blocks' LayerNorms folded into the linear layers around them (needs bit 1);
                          //                    0 = the launch list of rounds 2-5
    K_MT_GN1,             // LTK_MT_GN1         1 (default): GroupNorm of the maps whose (image, group) fits one block's registers (U-Net levels, the VAE's 32^2
                          //                    maps) as ONE launch (nn_kernels.hip gn_group_kernel) instead of gn_stats + gn_apply
    K_ATTN_PF,            // LTK_ATTN_PF        1 (default): multi-head attention (head dims 48 / 64 / 80) loads the K / V^T fragments of key tile t+1 while tile t
                          //                    is computed (register double buffer); 0: loaded in front of their MFMAs (rounds 2-5)
    K_SAT_CHECK,          // LTK_SAT_CHECK      debug, default 0: behind every layer / op its output is scanned for values AT the limit of its type (what an epilogue's
                          //                    clamp to +-65504 leaves behind; +-448 for e4m3) and for non-finite values; counters through ltk_debug_saturation.
                          //                    The fused Wav2Lip head (which writes bytes) runs unfused under it.
    K_CONV_S2SPLIT,       // LTK_CONV_S2SPLIT   1 (default): the first-generation kernel's stride-2 3x3 layers (face_encoder_blocks.1.0 / 2.0) stage their patch rows
                          //                    split by column parity, so that a ds_read_b128 lane group reads 256 contiguous bytes (conv_mfma.hip KArgs::s2half)
    K_FACE_CACHE_MAX_MB,  // LTK_FACE_CACHE_MAX_MB  largest face cache ONE avatar may take under knob FACE_CACHE (default 16384 MB = a 3 900-frame bank); a
                          //                    call for a longer avatar fails with LTK_E_NOMEM instead of allocating
    K_LIN_FK,             // LTK_LIN_FK         1 (default): 1x1 / linear layers with K = 320 / 384 / 512 / 640 input channels on >= LIN_FK_MIN_ROWS pixels or tokens run on
                          //                    lin_fk_kernel (conv3_mfma.hip: a wave's A rows in registers, full-K 32-cout weight slabs through LDS); 0: conv3 1x1
    K_LIN_FK_BLOCKS,      // LTK_LIN_FK_BLOCKS  blocks a lin_fk launch aims at (the output channels are split into groups of slabs until the grid has this many)
    K_LIN_FK_MIN_ROWS,    // LTK_LIN_FK_MIN_ROWS  (default 512)
    K_ATTN_LDS,           // LTK_ATTN_LDS       1 (default): self-attention with head dims 40 / 80 over >= 128 keys (whole 64-key tiles) shares its K / V^T tiles between a
                          //                    block's four query tiles through LDS (nn_kernels.hip attn_lds_kernel); 0: attn_kernel (every wave reads them from L2)
    K_LIN_MP,             // LTK_LIN_MP         1 (default): 1x1 / linear layers with K = 2560 / 5120 run on lin_mp_kernel (conv3_mfma.hip: passes of 1280 channels, the
                          //                    accumulators of 2 or 3 weight slabs per block in registers) where its grid is one round of blocks
                          //                    (conv3_lin_mp_nsl); 2 / 3: always, with that many slabs per block; 0: conv3 / rowconv
    K_GN_COOP,            // LTK_GN_COOP        1 (default): GroupNorm of the maps too large for MT_GN1 (the VAE's 64^2 .. 256^2 maps) in ONE tensor pass: blocks keep their
                          //                    slice in registers and exchange partial sums through global memory (nn_kernels.hip gn_coop_kernel); 0: gn_stats + gn_apply
    K_AUDIO0,             // LTK_AUDIO0         bit 0: audio_encoder.0 (1 -> 32 channels on the 80 x 16 mel window) as a VALU kernel that reads the float32 mel
                          //                    windows itself (conv7_mfma.hip audio0_kernel: no pack_mel launch, no 8-channel padded MFMA launch); bit 1: the
                          //                    stride-(3, 1) layer audio_encoder.3 on MFMAs fed straight from global memory (audio3_kernel); default 3; 0: pack_mel + conv_mfma_kernel
    K_CONV_S2D,           // LTK_CONV_S2D       1 (default): the face encoder's shallow stride-2 layers (face_encoder_blocks.1.0 / 2.0: 16 -> 32 @256^2, 32 -> 64 @128^2) on
                          //                    convs2d_kernel (conv7_mfma.hip: a wave = one output row x 32 output channels, weights in registers, pixel operands
                          //                    straight from global memory, no LDS); 0: conv_mfma_kernel (first generation)
    K_PF_LRU,             // LTK_PF_LRU         0 (default): a prefetch goes into the MOST recently used free slot (a lone session alternates between two slots: 7 launch
                          //                    graphs); 1: least recently used (round 6's first rule: a lone session walks all 16 slots, 33 graphs; kept for A/Bs)
    K_COUNT
};

int knob(Knob k);
// bumped by every knob_set: cached launch plans (captured graphs) are keyed by it
unsigned knob_epoch();
// hipFuncAttributeMaxDynamicSharedMemorySize = `bytes` for `func` on the CURRENT device, once per (device, function) and process
// (a launch path may run on any host thread, also inside a stream capture: the attribute is set by the first eager pass).
// Returns the hipError_t of the attribute call (0 = ok).
int ensure_dyn_lds(const void* func, int bytes);

// returns 0, or -1 when `name` (without the LTK_ prefix or with it) is not a knob
int knob_set(const char* name, int value);

}  // namespace ltk
