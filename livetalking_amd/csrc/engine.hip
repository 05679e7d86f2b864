// libltk_hip.so: engine + C ABI (include/ltk.h).
//
// Host-side statement of the Wav2Lip-256 generator graph
// (avatars/wav2lip/models/wav2lip_v2.py:12-91, forward :123-163) as a static
// layer program over a device activation arena: every layer is one launch of
// the MFMA implicit-GEMM kernels (conv3_mfma.hip / conv_mfma.hip); torch.cat skip
// connections are channel-block ranges of shared CB16 buffers; eval-mode BatchNorm
// is folded into the epilogue scale/shift at load time.  The MuseTalk / Whisper /
// VAE-encoder programs (musetalk.hip), frame egress and the test hooks follow.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ltk.h"
#include "conv_mfma.h"
#include "misc_kernels.h"
#include "musetalk.h"
#include "nn_kernels.h"
#include "tune.h"

using namespace ltk;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

// Entry of every call that launches: select the engine's GPU, and drop whatever error an EARLIER runtime call left behind on this
// host thread (ours after a reported failure, or another library's) - hipGetLastError() after a launch must speak about that launch
static hipError_t enter_device(int device) {
    (void)hipGetLastError();
    return hipSetDevice(device);
}

#define CHK(expr)                                                                        \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) return fail(LTK_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace {

// ---------------------------------------------------------------- network description
struct LayerDef {
    const char* prefix;
    bool transposed;
    int cin, cout, k, sh, sw, pad, out_pad;
    bool residual;
};

// wav2lip_v2.py:41-58
const LayerDef kAudio[] = {
    {"audio_encoder.0", false, 1, 32, 3, 1, 1, 1, 0, false},
    {"audio_encoder.1", false, 32, 32, 3, 1, 1, 1, 0, true},
    {"audio_encoder.2", false, 32, 32, 3, 1, 1, 1, 0, true},
    {"audio_encoder.3", false, 32, 64, 3, 3, 1, 1, 0, false},
    {"audio_encoder.4", false, 64, 64, 3, 1, 1, 1, 0, true},
    {"audio_encoder.5", false, 64, 64, 3, 1, 1, 1, 0, true},
    {"audio_encoder.6", false, 64, 128, 3, 3, 3, 1, 0, false},
    {"audio_encoder.7", false, 128, 128, 3, 1, 1, 1, 0, true},
    {"audio_encoder.8", false, 128, 128, 3, 1, 1, 1, 0, true},
    {"audio_encoder.9", false, 128, 256, 3, 3, 2, 1, 0, false},
    {"audio_encoder.10", false, 256, 256, 3, 1, 1, 1, 0, true},
    {"audio_encoder.11", false, 256, 512, 3, 1, 1, 0, 0, false},
    {"audio_encoder.12", false, 512, 512, 1, 1, 1, 0, 0, false},
};
// wav2lip_v2.py:12-39 (blocks separated by block index)
struct BlockLayer { int block; LayerDef d; };
const BlockLayer kFaceEnc[] = {
    {0, {"face_encoder_blocks.0.0", false, 6, 16, 7, 1, 1, 3, 0, false}},
    {1, {"face_encoder_blocks.1.0", false, 16, 32, 3, 2, 2, 1, 0, false}},
    {1, {"face_encoder_blocks.1.1", false, 32, 32, 3, 1, 1, 1, 0, true}},
    {1, {"face_encoder_blocks.1.2", false, 32, 32, 3, 1, 1, 1, 0, true}},
    {2, {"face_encoder_blocks.2.0", false, 32, 64, 3, 2, 2, 1, 0, false}},
    {2, {"face_encoder_blocks.2.1", false, 64, 64, 3, 1, 1, 1, 0, true}},
    {2, {"face_encoder_blocks.2.2", false, 64, 64, 3, 1, 1, 1, 0, true}},
    {2, {"face_encoder_blocks.2.3", false, 64, 64, 3, 1, 1, 1, 0, true}},
    {3, {"face_encoder_blocks.3.0", false, 64, 128, 3, 2, 2, 1, 0, false}},
    {3, {"face_encoder_blocks.3.1", false, 128, 128, 3, 1, 1, 1, 0, true}},
    {3, {"face_encoder_blocks.3.2", false, 128, 128, 3, 1, 1, 1, 0, true}},
    {4, {"face_encoder_blocks.4.0", false, 128, 256, 3, 2, 2, 1, 0, false}},
    {4, {"face_encoder_blocks.4.1", false, 256, 256, 3, 1, 1, 1, 0, true}},
    {4, {"face_encoder_blocks.4.2", false, 256, 256, 3, 1, 1, 1, 0, true}},
    {5, {"face_encoder_blocks.5.0", false, 256, 512, 3, 2, 2, 1, 0, false}},
    {5, {"face_encoder_blocks.5.1", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {6, {"face_encoder_blocks.6.0", false, 512, 512, 3, 2, 2, 1, 0, false}},
    {6, {"face_encoder_blocks.6.1", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {7, {"face_encoder_blocks.7.0", false, 512, 512, 4, 1, 1, 0, 0, false}},
    {7, {"face_encoder_blocks.7.1", false, 512, 512, 1, 1, 1, 0, 0, false}},
};
// wav2lip_v2.py:60-87
const BlockLayer kFaceDec[] = {
    {0, {"face_decoder_blocks.0.0", false, 512, 512, 1, 1, 1, 0, 0, false}},
    {1, {"face_decoder_blocks.1.0", true, 1024, 512, 4, 1, 1, 0, 0, false}},
    {1, {"face_decoder_blocks.1.1", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {2, {"face_decoder_blocks.2.0", true, 1024, 512, 3, 2, 2, 1, 1, false}},
    {2, {"face_decoder_blocks.2.1", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {3, {"face_decoder_blocks.3.0", true, 1024, 512, 3, 2, 2, 1, 1, false}},
    {3, {"face_decoder_blocks.3.1", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {3, {"face_decoder_blocks.3.2", false, 512, 512, 3, 1, 1, 1, 0, true}},
    {4, {"face_decoder_blocks.4.0", true, 768, 384, 3, 2, 2, 1, 1, false}},
    {4, {"face_decoder_blocks.4.1", false, 384, 384, 3, 1, 1, 1, 0, true}},
    {4, {"face_decoder_blocks.4.2", false, 384, 384, 3, 1, 1, 1, 0, true}},
    {5, {"face_decoder_blocks.5.0", true, 512, 256, 3, 2, 2, 1, 1, false}},
    {5, {"face_decoder_blocks.5.1", false, 256, 256, 3, 1, 1, 1, 0, true}},
    {5, {"face_decoder_blocks.5.2", false, 256, 256, 3, 1, 1, 1, 0, true}},
    {6, {"face_decoder_blocks.6.0", true, 320, 128, 3, 2, 2, 1, 1, false}},
    {6, {"face_decoder_blocks.6.1", false, 128, 128, 3, 1, 1, 1, 0, true}},
    {6, {"face_decoder_blocks.6.2", false, 128, 128, 3, 1, 1, 1, 0, true}},
    {7, {"face_decoder_blocks.7.0", true, 160, 64, 3, 2, 2, 1, 1, false}},
    {7, {"face_decoder_blocks.7.1", false, 64, 64, 3, 1, 1, 1, 0, true}},
    {7, {"face_decoder_blocks.7.2", false, 64, 64, 3, 1, 1, 1, 0, true}},
};
const LayerDef kOutConv = {"output_block.0", false, 80, 32, 3, 1, 1, 1, 0, false};  // wav2lip_v2.py:89
const int kDecCh[8] = {512, 512, 512, 512, 384, 256, 128, 64};
const int kFeatCh[8] = {16, 32, 64, 128, 256, 512, 512, 512};
const int kFeatHW[8] = {256, 128, 64, 32, 16, 8, 4, 1};
const float kBnEps = 1e-5f;  // nn.BatchNorm2d default (conv.py:9,38)

constexpr int kPrefetchMaxFrames = 32;     // knob PREFETCH: calls of at most this many frames are pipelined across calls
enum BufId { B_MEL = 0, B_AT0, B_AT1, B_X0, B_T0, B_T1, B_OUT32, B_CAT0, B_COUNT = B_CAT0 + 8 };

struct Layer {
    std::string name;
    ConvPlan plan;
    RowGemmPlan rg;                 // set for the layers whose input and output maps are one pixel per frame (rowgemm.hip)
    int rg_y_ld = 0;               // output row pitch of that GEMM (the 1x1-expand layer writes k*k*Cout contiguous channels)
    bool rowconv = false;          // `rg` is a rowconv plan instead: 3x3 conv on a map of <= 8 x 8 output pixels (rowgemm.hip)
    int rc_stride = 1, rc_stride_w = 0;
    RowGemmPlan rgT[4];            // ConvTranspose2d(k3,s2,p1,op1) on a source map of <= 8 x 8 pixels: one plan per output phase (rowconvT_launch)
    int cin_real = 0;
    int in_buf = 0, in_ld = 0, in_coff = 0, H = 0, W = 0;
    int out_buf = 0, out_ld = 0, out_coff = 0, Ho = 0, Wo = 0;
    bool residual = false;
    bool res_folded = false;   // the identity branch lives in the centre tap of the packed weights
    bool audio = false;   // audio-encoder layer (independent of the face encoder until decoder block 0)
    int special = 0;      // 3: audio_encoder.3, which has a kernel of its own (audio3_kernel, knob AUDIO0 bit 1)
    ConvS2dPlan* s2d = nullptr;   // face_encoder_blocks.1.0 / 2.0: the shallow stride-2 layers on convs2d_kernel (conv7_mfma.hip, knob CONV_S2D)
    bool face_enc = false;   // face-encoder layer: depends on the bank frame only (knob FACE_CACHE)
    double macs = 0;  // per frame
    // measured tile / split choice per frame-count bucket (<= 16, 32, 64, 128, 256+ frames per launch); 0 = conv3's rule
    struct Tile { signed char pxw = 0, nbt = 0, ks = 0; } tile[5];
};

int frame_bucket(int nf) { return nf <= 16 ? 0 : nf <= 32 ? 1 : nf <= 64 ? 2 : nf <= 128 ? 3 : 4; }

// Per-layer tile / split choices that beat conv3's rule inside a whole pass (scripts/tile_tune.py on MI355X,
// profiles/r02_tile_tune.txt: every layer timed between its neighbours, so with the cache state they leave).  Tiles never
// change an output element's summation order; the few split entries replace the split the rule would have chosen.
struct TileEntry { const char* layer; int bucket, pxw, nbt, ks; };
const TileEntry kTileTable[] = {
    // <= 16 frames per launch
    {"face_encoder_blocks.5.1", 0, 2, 1, 0},     // 512 ch @ 8^2: 27.7 -> 22.6 us (half the weight-slab re-reads of 128-px tiles)
    {"face_decoder_blocks.2.1", 0, 2, 1, 0},     // 512 ch @ 8^2: 28.1 -> 22.9 us
    {"face_encoder_blocks.6.0", 0, 0, 0, 8},     // 512 -> 512 stride 2 @ 8^2: 20.7 -> 18.1 us
    {"face_decoder_blocks.1.0", 0, 0, 0, 4},     // convT 4x4 on the 1x1 map: 20.9 -> 18.0 us
    {"audio_encoder.7", 0, 0, 0, 1},             // 128 ch @ 9x6: 14.1 -> 12.0 us unsplit
    {"audio_encoder.8", 0, 0, 0, 1},
    // <= 64 frames per launch
    {"face_decoder_blocks.4.1", 2, 2, 2, 0},     // 384 ch @ 32^2: 182 -> 150 us
    {"face_decoder_blocks.4.2", 2, 2, 2, 0},
    {"face_encoder_blocks.5.0", 2, 2, 1, 0},     // 256 -> 512 stride 2: 33.6 -> 26.5 us
    {"face_encoder_blocks.6.1", 2, 1, 2, 0},     // 512 ch @ 4^2: 29.0 -> 24.3 us
    {"face_decoder_blocks.1.1", 2, 1, 2, 0},     // 512 ch @ 4^2: 29.4 -> 23.2 us
    {"face_decoder_blocks.1.0", 2, 0, 0, 4},
};

// Device-free consistency check of kTileTable (include/ltk.h ltk_debug_tile_table_check): every entry names a layer of the network
// description above, a frame-count bucket, and a tile / split conv3 has an instantiation for on that layer.  The table is keyed by
// strings and was tuned on single boxes: an entry that no longer matches anything would cost speed silently.
int check_tile_table_impl(std::string& msg) {
    int bad = 0;
    auto find = [](const char* name) -> const LayerDef* {
        for (const LayerDef& d : kAudio) if (!strcmp(d.prefix, name)) return &d;
        for (const BlockLayer& b : kFaceEnc) if (!strcmp(b.d.prefix, name)) return &b.d;
        for (const BlockLayer& b : kFaceDec) if (!strcmp(b.d.prefix, name)) return &b.d;
        if (!strcmp(kOutConv.prefix, name)) return &kOutConv;
        return nullptr;
    };
    const size_t n = sizeof(kTileTable) / sizeof(kTileTable[0]);
    for (size_t i = 0; i < n; ++i) {
        const TileEntry& t = kTileTable[i];
        auto complain = [&](const char* what) { ++bad; msg += std::string(t.layer) + " (bucket " + std::to_string(t.bucket) + "): " + what + "; "; };
        const LayerDef* d = find(t.layer);
        if (!d) { complain("no such layer"); continue; }
        if (t.bucket < 0 || t.bucket >= 5) complain("bucket outside 0..4");
        if ((t.pxw == 0) != (t.nbt == 0)) complain("pxw and nbt must be given together");
        if (t.pxw != 0 && t.pxw != 1 && t.pxw != 2 && t.pxw != 4) complain("pxw must be 0, 1, 2 or 4");
        if (t.nbt < 0 || t.nbt > 2) complain("nbt must be 0, 1 or 2");
        if (t.ks < 0 || t.ks > 32) complain("split factor outside 0..32");
        if (t.pxw == 0 && t.nbt == 0 && t.ks == 0) complain("entry changes nothing");
        const bool s1_3x3 = !d->transposed && d->k == 3 && d->sh == 1 && d->sw == 1;
        if ((t.pxw == 1 || t.pxw == 4) && !s1_3x3) complain("128- / 512-pixel tiles exist for 3x3 stride-1 layers only");
        if (t.pxw == 4 && d->cout > 32) complain("512-pixel tiles exist for <= 32 output channels only");
        if (t.nbt == 2 && d->cout < 64) complain("64-cout blocks need >= 64 output channels");
        if (d->k == 7) complain("the first layer runs on conv7, not conv3");
        for (size_t j = 0; j < i; ++j)
            if (!strcmp(kTileTable[j].layer, t.layer) && kTileTable[j].bucket == t.bucket) complain("duplicate entry");
    }
    return bad;
}

void apply_tile_table_impl(std::vector<Layer>& layers) {
    for (const TileEntry& t : kTileTable)
        for (Layer& L : layers)
            if (L.name == t.layer) { L.tile[t.bucket].pxw = (signed char)t.pxw; L.tile[t.bucket].nbt = (signed char)t.nbt; L.tile[t.bucket].ks = (signed char)t.ks; }
}


// Avatar banks are shared_ptr-owned: an entry point keeps its avatar alive for the duration of the call, so a concurrent
// ltk_avatar_release only drops the table's reference and the device buffers go when the last call using them returns
// (every entry point synchronises its stream before it returns).  The destructor also frees a half-built bank when a
// register call fails part-way.
struct Avatar {
    uint8_t* d_face = nullptr;
    uint8_t* d_full = nullptr;
    std::vector<int32_t> coords;
    int n = 0, H = 0, W = 0;
    int device = 0;
    // knob FACE_CACHE: the face encoder's skip tensors of every bank frame (records of feat_rec_bytes, misc_kernels.h FeatGeom),
    // built on first use under the engine's enqueue lock; feat_epoch = knob_epoch() it was built under
    // (d_feat / feat_rec_bytes / feat_epoch are touched under the engine's enqueue lock only; feat_bytes is what the statistics
    // getter reads from other threads)
    uint8_t* d_feat = nullptr;
    size_t feat_rec_bytes = 0;
    unsigned feat_epoch = 0;
    std::atomic<size_t> feat_bytes{0};
    Avatar() = default;
    Avatar(const Avatar&) = delete;
    Avatar& operator=(const Avatar&) = delete;
    ~Avatar() {
        (void)hipSetDevice(device);
        if (d_face) (void)hipFree(d_face);
        if (d_full) (void)hipFree(d_full);
        if (d_feat) (void)hipFree(d_feat);
    }
};

struct Scratch {
    void* d = nullptr;
    size_t cap = 0;
};

// MuseTalk avatar bank (musetalk_avatar.py:69-91)
struct MtAvatar {
    float* d_latents = nullptr;      // [n][8][32][32]
    uint8_t* d_full = nullptr;       // [n][H][W][3]
    uint8_t* d_masks = nullptr;      // concatenated
    std::vector<int64_t> mask_off;
    std::vector<int32_t> face_box, crop_box;
    int n = 0, H = 0, W = 0;
    int device = 0;
    MtAvatar() = default;
    MtAvatar(const MtAvatar&) = delete;
    MtAvatar& operator=(const MtAvatar&) = delete;
    ~MtAvatar() {
        (void)hipSetDevice(device);
        if (d_latents) (void)hipFree(d_latents);
        if (d_full) (void)hipFree(d_full);
        if (d_masks) (void)hipFree(d_masks);
    }
};

// RAII HIP event: error returns between create and destroy do not leak it
struct Ev {
    hipEvent_t e = nullptr;
    hipError_t create() { return hipEventCreateWithFlags(&e, hipEventDisableTiming); }
    ~Ev() { if (e) (void)hipEventDestroy(e); }
};

}  // namespace

struct ltk_engine {
    int device = 0;
    hipStream_t compute = nullptr;
    hipStream_t aux = nullptr;            // audio encoder runs beside the face encoder (wav2lip_v2.py:132 vs :136-140)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* d_partial = nullptr;           // conv3 split-K scratch of the compute stream
    float* d_partial_aux = nullptr;       // ... of the aux stream
    float* d_partial_pf = nullptr;        // ... of the prefetch stream (aux2, knob PREFETCH)
    unsigned long long* d_sat = nullptr;  // [2] saturation counters of knob SAT_CHECK (ltk_debug_saturation): halfs at the fp16 limit, non-finite halfs
    size_t partial_cap = 0, partial_aux_cap = 0, partial_pf_cap = 0;
    std::mutex mu;            // enqueue order on `compute` + arena ownership
    std::mutex pool_mu;       // scratch / stream pools, avatar table
    // wav2lip
    bool loaded = false;
    int max_frames = 0;
    int micro_batch = 0;
    std::vector<Layer> layers;
    f16* buf[B_COUNT] = {nullptr};
    // knob PREFETCH: kPfSlots (16 x 0.13 GB at 32 frames) further instances of the eight concat buffers ("slots" 1..kPfSlots; set 0 = buf, where a call that runs
    // the whole network works), sized for alt_frames frames, each holding the prefetched face-encoder outputs of ONE upcoming call,
    // keyed by (avatar, first bank index, frame count): interleaved solo calls of several paced sessions each find their own slot
    // (round 5 kept one engine-wide slot, which only a lone session's calls ever hit).  pf_tmp: the prefetched encoder's own
    // temporaries (prefetches are serialised on aux2).  A call's decoder works in the set its skip tensors were written to.
    struct PfSlot {
        f16* cat[B_COUNT] = {nullptr};
        bool valid = false;               // holds the outputs for (avatar, first, nf) computed under `epoch`
        int avatar = -1, first = -1, nf = 0;
        unsigned epoch = 0;
        unsigned long stamp = 0;          // LRU clock of the last fill / use
        double filled_at = 0;             // host time of the last fill (seconds): a valid slot nobody came for is reclaimed after kPfStale
        hipEvent_t ev_done = nullptr;     // the prefetch into this slot has finished (recorded on aux2)
        hipEvent_t ev_read = nullptr;     // the last pass that worked in this slot has finished (recorded on compute)
        bool filled = false, read = false;
        std::shared_ptr<Avatar> hold;     // the bank a prefetch into this slot reads
    };
    static constexpr int kPfSlots = 16;
    PfSlot pfs[kPfSlots + 1];             // [0] unused
    f16* pf_tmp[B_COUNT] = {nullptr};
    int alt_frames = 0;
    hipStream_t aux2 = nullptr;
    unsigned long pf_clock = 0;
    DevTables* d_tab_next = nullptr;  // faces table of the prefetched frames
    // recent solo calls, per session position: a call that starts where one of them ended (same avatar, same size) continues a session
    struct SoloSeq { int avatar = -1, next = -1, nf = 0; unsigned long stamp = 0; };
    SoloSeq solo_seq[2 * kPfSlots];
    unsigned long pf_hits = 0, pf_misses = 0, pf_issued = 0;
    std::atomic<bool> pf_fail_logged{false};       // a prefetch that could not be launched is reported once (the call itself succeeds)
    // LTK_INFER_TIMING=1 (measurement): host time of ltk_wav2lip_infer by phase, printed when the engine is destroyed
    double tm_prep = 0, tm_launch = 0, tm_pf = 0, tm_wait = 0;
    unsigned long tm_calls = 0;
    size_t buf_halfs[B_COUNT] = {0};  // per frame
    float* d_head = nullptr;          // 96 weights + 3 bias
    Conv7Plan* c7 = nullptr;          // first layer (7x7, 6 -> 16) with the input pack fused: conv7_mfma.hip
    Audio0Plan* a0 = nullptr;         // audio_encoder.0 (3x3, 1 -> 32) with the mel pack fused (VALU): conv7_mfma.hip, knob AUDIO0 bit 0
    Audio3Plan* a3 = nullptr;         // audio_encoder.3 (3x3 stride (3,1), 32 -> 64), MFMA operands straight from global memory: conv7_mfma.hip, knob AUDIO0 bit 1
    double macs_per_frame = 0;
    DevTables* d_tab = nullptr;       // per-frame pointer tables of the pass being enqueued (misc_kernels.h), filled on the compute stream
    // captured passes (knob GRAPH): one executable graph per frame count of the product configuration (bank crops in, fused head out);
    // a frame count is captured the second time it is seen, the least recently used graph goes when the table is full
    struct PassGraph { hipGraphExec_t exec = nullptr; int seen = 0; unsigned long stamp = 0; };
    std::map<int, PassGraph> graphs;
    unsigned graph_epoch = 0;         // knob_epoch() the graphs were captured under
    unsigned long graph_clock = 0;
    // captured MuseTalk / Whisper programs (run_program): the static launch list of a program over its persistent buffers, one
    // executable graph per (program, frame count); the kernels that carry per-call pointers stay outside the graph
    std::map<std::pair<const void*, int>, PassGraph> prog_graphs;
    unsigned prog_graph_epoch = 0;
    // debug capture
    bool capture = false;
    std::map<std::string, std::vector<float>> taps;
    std::map<std::string, std::vector<int>> tap_shape;
    // avatars
    std::map<int, std::shared_ptr<Avatar>> avatars;
    int next_avatar = 1;
    // mel
    float* d_basis = nullptr;
    int32_t* d_lohi = nullptr;
    // musetalk
    MtGraph* mt = nullptr;
    int mt_max_frames = 0;
    int mt_fp8 = 0;                    // ltk_musetalk_set_fp8
    float mt_fp8_ascale = 8.f;
    MtGraph* vae_enc = nullptr;           // AutoencoderKL encoder graph (avatar preparation), 2 images per face
    int vae_enc_faces = 0;
    MtGraph* whisper = nullptr;           // Whisper-tiny encoder graph (Audio2Feature)
    float* d_wbasis = nullptr;            // slaney mel basis [80][201] (n_fft 400, 0..8000 Hz)
    float* d_wlogspec = nullptr;          // [80][3000]
    float* d_wpcm = nullptr;              // staging, 30 s
    int* d_wgmax = nullptr;
    float* d_pe = nullptr;                // PositionalEncoding table [50][384]
    float* d_mt_feat = nullptr;           // staging: fp32 [max_frames][50][384]
    float* d_mt_lat = nullptr;            // staging for the host-input hook: fp32 [max_frames][8][32][32]
    std::map<int, std::shared_ptr<MtAvatar>> mt_avatars;
    // pools
    std::vector<Scratch> scratch_free;
    std::vector<hipStream_t> stream_free;
};

namespace {

struct ScratchLease {
    ltk_engine* e;
    Scratch s;
    ScratchLease(ltk_engine* e_, size_t bytes) : e(e_) {
        {
            std::lock_guard<std::mutex> g(e->pool_mu);
            for (size_t i = 0; i < e->scratch_free.size(); ++i)
                if (e->scratch_free[i].cap >= bytes) {
                    s = e->scratch_free[i];
                    e->scratch_free.erase(e->scratch_free.begin() + i);
                    break;
                }
        }
        if (!s.d) {
            size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
            if (hipMalloc(&s.d, cap) == hipSuccess) s.cap = cap; else s.d = nullptr;
        }
    }
    ~ScratchLease() {
        if (s.d) {
            std::lock_guard<std::mutex> g(e->pool_mu);
            e->scratch_free.push_back(s);
        }
    }
};

struct StreamLease {
    ltk_engine* e;
    hipStream_t s = nullptr;
    bool owned = false;
    StreamLease(ltk_engine* e_, void* user) : e(e_) {
        if (user) { s = (hipStream_t)user; return; }
        owned = true;
        {
            std::lock_guard<std::mutex> g(e->pool_mu);
            if (!e->stream_free.empty()) { s = e->stream_free.back(); e->stream_free.pop_back(); }
        }
        if (!s) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    ~StreamLease() {
        if (owned && s) {
            std::lock_guard<std::mutex> g(e->pool_mu);
            e->stream_free.push_back(s);
        }
    }
};

const float* find_tensor(const ltk_named_tensor* sd, int n, const std::string& name, size_t expect) {
    for (int i = 0; i < n; ++i) {
        if (name == sd[i].name) {
            size_t cnt = 1;
            for (int d = 0; d < sd[i].ndim; ++d) cnt *= (size_t)sd[i].shape[d];
            if (cnt != expect) return nullptr;
            return sd[i].data;
        }
    }
    return nullptr;
}

// `hint_hw`: pixels per image of the layer's input map.  `flat_ld` > 0: the k x k "valid" conv that collapses a
// k x k map to 1x1 (face_encoder_blocks.7.0) is run as a 1x1 conv over the map viewed as ONE pixel of
// k*k*cin channels (a channel-blocked k x k map is contiguous per channel block).
// `map_w` > 0: the input map is map_w x map_w (face encoder / decoder): 3x3 layers whose OUTPUT map is at most 8 x 8 also get a rowconv plan.
int build_layer_impl(ltk_engine* e, const LayerDef& d, const ltk_named_tensor* sd, int n, Layer* L, int hint_hw, int flat_ld, int map_w) {
    const std::string p = d.prefix;
    const size_t wcount = (size_t)d.cin * d.cout * d.k * d.k;
    const float* w = find_tensor(sd, n, p + ".conv_block.0.weight", wcount);
    const float* b = find_tensor(sd, n, p + ".conv_block.0.bias", d.cout);
    const float* g = find_tensor(sd, n, p + ".conv_block.1.weight", d.cout);
    const float* beta = find_tensor(sd, n, p + ".conv_block.1.bias", d.cout);
    const float* mean = find_tensor(sd, n, p + ".conv_block.1.running_mean", d.cout);
    const float* var = find_tensor(sd, n, p + ".conv_block.1.running_var", d.cout);
    if (!w || !b || !g || !beta || !mean || !var)
        return fail(LTK_E_INVALID, "state_dict is missing (or has a wrong shape for) tensors of layer " + p);
    std::vector<float> sc(d.cout), sf(d.cout);
    for (int c = 0; c < d.cout; ++c) {
        // BatchNorm2d eval: y = (x - mean)/sqrt(var+eps)*gamma + beta, x = conv + bias
        const float s = g[c] / sqrtf(var[c] + kBnEps);
        sc[c] = s;
        sf[c] = (b[c] - mean[c]) * s + beta[c];
    }
    std::string err;
    int rc;
    // Residual blocks (conv.py:16-17: out = relu(bn(conv(x)) + x), x = the block's own input): with
    // y = s*conv(x) + t + x the identity is the centre tap of a k x k kernel, w[co][co][c][c] += 1/s[co].
    // The accumulation is fp32 and the fp16 rounding of (w + 1/s) perturbs the identity term by one fp16 ulp of
    // x - the same error x already carries - while the separate residual read (one extra pass over the
    // activation) disappears.  Not applied when a scale is too small for 1/s to be a sane fp16 weight.
    std::vector<float> wfold;
    L->res_folded = false;
    if (d.residual && !d.transposed && d.cin == d.cout && (d.k & 1) && d.sh == 1 && d.sw == 1 && d.pad == d.k / 2 &&
        !knob(K_NO_FOLD_RESIDUAL)) {
        bool ok = true;
        for (int c = 0; c < d.cout; ++c) ok = ok && fabsf(sc[c]) >= 1e-3f;
        if (ok) {
            wfold.assign(w, w + wcount);
            const int kk = d.k * d.k, ctr = (d.k / 2) * d.k + d.k / 2;
            for (int c = 0; c < d.cout; ++c) wfold[((size_t)c * d.cin + c) * kk + ctr] += 1.0f / sc[c];
            w = wfold.data();
            L->res_folded = true;
        }
    }
    if (flat_ld > 0) {
        // channel-blocked map [n][cb][k*k][16] read as ONE pixel of cin*k*k channels: flat channel = ((cb*kk + t)*16 + c16)
        const int kk = d.k * d.k;
        const int cin_flat = kk * d.cin;
        std::vector<float> wf((size_t)d.cout * cin_flat, 0.f);
        for (int co = 0; co < d.cout; ++co)
            for (int ci = 0; ci < d.cin; ++ci)
                for (int t = 0; t < kk; ++t)
                    wf[(size_t)co * cin_flat + ((size_t)(ci >> 4) * kk + t) * 16 + (ci & 15)] = w[((size_t)co * d.cin + ci) * kk + t];
        rc = conv_plan_create(&L->plan, wf.data(), cin_flat, d.cout, 1, 1, 1, 1, 0, 0, false, 0, sc.data(), sf.data(), &err, 1);
    } else {
        rc = conv_plan_create(&L->plan, w, d.cin, d.cout, d.k, d.k, d.sh, d.sw, d.pad, d.pad, d.transposed, d.out_pad,
                              sc.data(), sf.data(), &err, hint_hw);
    }
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, p + ": " + err);
    // one pixel per frame on both sides: a plain GEMM with as many rows as frames (rowgemm.hip, used for launches of <= 32 frames).
    // Not built (45 MB of duplicated weights) when the process starts with the paths switched off.
    const bool want_rowgemm = knob(K_ROWGEMM) && knob(K_SPLITK), want_rowconv = knob(K_ROWCONV) > 0 && knob(K_SPLITK);
    {
        std::vector<float> we, se, fe;
        int J = 0, K = 0;
        if (flat_ld > 0) {                                            // k x k valid conv collapsing the k x k map: W = the flattened weights above
            const int kk = d.k * d.k;
            J = d.cout; K = kk * d.cin;
            we.assign((size_t)J * K, 0.f);
            for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < d.cin; ++ci)
                    for (int t = 0; t < kk; ++t)
                        we[(size_t)co * K + ((size_t)(ci >> 4) * kk + t) * 16 + (ci & 15)] = w[((size_t)co * d.cin + ci) * kk + t];
            se = sc; fe = sf;
        } else if (!d.transposed && d.k == 1 && hint_hw == 1 && d.cin % 32 == 0 && d.cout % 16 == 0) {
            J = d.cout; K = d.cin;
            we.assign(w, w + (size_t)J * K);
            se = sc; fe = sf;
        } else if (d.transposed && hint_hw == 1 && d.sh == 1 && d.pad == 0 && d.out_pad == 0 && d.cin % 32 == 0 && d.cout % 16 == 0) {
            // ConvTranspose2d(k, 1, 0) on a 1x1 map: output channel-blocked k x k map [cout block][position][16] = k*k*Cout columns
            const int kk = d.k * d.k;
            J = kk * d.cout; K = d.cin;
            we.assign((size_t)J * K, 0.f); se.assign(J, 0.f); fe.assign(J, 0.f);
            for (int j = 0; j < J; ++j) {
                const int c16 = j & 15, tt = j >> 4, pos = tt % kk, co = (tt / kk) * 16 + c16;
                for (int ci = 0; ci < d.cin; ++ci) we[(size_t)j * K + ci] = w[((size_t)ci * d.cout + co) * kk + pos];
                se[j] = sc[co]; fe[j] = sf[co];
            }
        }
        if (J > 0 && !want_rowgemm) {
            // conv3 + split-K finish serves the layer
        } else if (J > 0) {
            rc = rowgemm_plan_create(&L->rg, we.data(), J, K, se.data(), fe.data(), &err);
            if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, p + ": " + err);
            L->rg_y_ld = (d.transposed ? J : 0);
        } else if (want_rowconv && flat_ld == 0 && !d.transposed && d.k == 3 && d.pad == 1 && (d.cout % 256 == 0 || (map_w < 0 && d.cout == 128)) &&
                   ((map_w > 0 && d.sh == d.sw && (d.sh == 1 || d.sh == 2) && map_w % d.sh == 0 && map_w / d.sh <= 8 && (d.cin == 256 || d.cin == 512)) ||
                    // round 5: the audio encoder's last two 3 x 3 layers (audio_encoder.9: 128 -> 256, stride (3, 2), 9 x 6 -> 3 x 3; .10: 256 -> 256 on
                    // 3 x 3): 144 output pixels per 16-frame launch behind 0.6 / 1.2 MB of weights (map_w < 0: the caller vouches for a small map)
                    (map_w < 0 && d.cin % 32 == 0))) {
            // 3x3 conv whose output map is at most 8 x 8: W_eff[j][tap * Cin + c], tap = ky * 3 + kx (`w` carries the folded identity
            // of a residual layer, exactly as the conv3 plan above does)
            J = d.cout; K = 9 * d.cin;
            we.assign((size_t)J * K, 0.f);
            for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < d.cin; ++ci)
                    for (int t = 0; t < 9; ++t) we[(size_t)co * K + (size_t)t * d.cin + ci] = w[((size_t)co * d.cin + ci) * 9 + t];
            rc = rowgemm_plan_create(&L->rg, we.data(), J, K, sc.data(), sf.data(), &err);
            if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, p + ": " + err);
            L->rowconv = true;
            L->rc_stride = d.sh;
            L->rc_stride_w = d.sw;
        } else if (want_rowconv && map_w > 0 && map_w <= 8 && d.transposed && d.k == 3 && d.sh == 2 && d.sw == 2 && d.pad == 1 && d.out_pad == 1 &&
                   (d.cin == 256 || d.cin == 512 || d.cin == 1024) && d.cout % 256 == 0) {
            // stride-2 transposed conv on the 4x4 / 8x8 maps: output pixel (2y + py, 2x + px) = sum over (dy, dx) of x[y + dy][x + dx] * w[:, :, ky, kx]
            // with ky = py + 1 - 2 dy, kx = px + 1 - 2 dx (torch ConvTranspose2d: oy = 2 iy - 1 + ky; weight layout [cin][cout][kh][kw])
            for (int gph = 0; gph < 4; ++gph) {
                const int py = gph >> 1, px = gph & 1, ny = 1 + py, nx = 1 + px;
                J = d.cout; K = ny * nx * d.cin;
                we.assign((size_t)J * K, 0.f);
                for (int dy = 0; dy < ny; ++dy)
                    for (int dx = 0; dx < nx; ++dx) {
                        const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx, t = dy * nx + dx;
                        for (int co = 0; co < d.cout; ++co)
                            for (int ci = 0; ci < d.cin; ++ci)
                                we[(size_t)co * K + (size_t)t * d.cin + ci] = w[((size_t)ci * d.cout + co) * 9 + ky * 3 + kx];
                    }
                rc = rowgemm_plan_create(&L->rgT[gph], we.data(), J, K, sc.data(), sf.data(), &err);
                if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, p + ": " + err);
            }
        }
    }
    if (!d.transposed && d.k == 7 && d.cin == 6 && d.cout == 16 && d.sh == 1 && d.pad == 3 && knob(K_CONV7) && !e->c7) {
        rc = conv7_plan_create(&e->c7, w, sc.data(), sf.data(), &err);
        if (rc) return fail(LTK_E_HIP, p + ": " + err);
    }
    if (!d.transposed && d.k == 3 && d.cin == 1 && d.cout == 32 && d.sh == 1 && d.sw == 1 && d.pad == 1 && !d.residual && (knob(K_AUDIO0) & 1) && !e->a0) {
        rc = audio0_plan_create(&e->a0, w, sc.data(), sf.data(), &err);
        if (rc) return fail(LTK_E_HIP, p + ": " + err);
    }
    if (!d.transposed && d.k == 3 && d.cin == 32 && d.cout == 64 && d.sh == 3 && d.sw == 1 && d.pad == 1 && !d.residual && (knob(K_AUDIO0) & 2) && !e->a3) {
        rc = audio3_plan_create(&e->a3, w, sc.data(), sf.data(), &err);
        if (rc) return fail(LTK_E_HIP, p + ": " + err);
        L->special = 3;
    }
    if (!d.transposed && d.k == 3 && d.sh == 2 && d.sw == 2 && d.pad == 1 && !d.residual && (d.cin == 16 || d.cin == 32) && d.cout % 32 == 0 &&
        knob(K_CONV_S2D) && !L->s2d) {
        rc = convs2d_plan_create(&L->s2d, w, d.cin, d.cout, sc.data(), sf.data(), &err);
        if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, p + ": " + err);
    }
    L->name = p;
    L->cin_real = d.cin;
    L->residual = d.residual;
    return 0;
}


// A Layer is pushed into e->layers only after it is complete: on a failure the device plans built so far go here (a failed load
// that is retried would otherwise leak the packed weights each time)
int build_layer(ltk_engine* e, const LayerDef& d, const ltk_named_tensor* sd, int n, Layer* L, int hint_hw = 0, int flat_ld = 0, int map_w = 0) {
    const int rc = build_layer_impl(e, d, sd, n, L, hint_hw, flat_ld, map_w);
    if (rc) {
        conv_plan_destroy(&L->plan); rowgemm_plan_destroy(&L->rg);
        for (RowGemmPlan& q : L->rgT) rowgemm_plan_destroy(&q);
        convs2d_plan_destroy(L->s2d); L->s2d = nullptr;
    }
    return rc;
}

void bump(size_t* cur, size_t v) { if (v > *cur) *cur = v; }

// Everything ltk_wav2lip_load creates (layer plans, head weights, first-layer plan, activation arena): a failed load leaves
// the engine as it found it, and can be retried.
void drop_graphs(ltk_engine* e) {
    if (e->aux2) (void)hipStreamSynchronize(e->aux2);      // a prefetch graph may still be running on the third stream
    for (auto& kv : e->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    e->graphs.clear();
}

void drop_prog_graphs(ltk_engine* e) {
    for (auto& kv : e->prog_graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    e->prog_graphs.clear();
}

void wav2lip_unload(ltk_engine* e) {
    if (e->aux2) (void)hipStreamSynchronize(e->aux2);        // an outstanding prefetch writes buffers that go away below
    drop_graphs(e);
    if (e->d_tab) { (void)hipFree(e->d_tab); e->d_tab = nullptr; }
    for (Layer& L : e->layers) {
        conv_plan_destroy(&L.plan); rowgemm_plan_destroy(&L.rg);
        for (RowGemmPlan& q : L.rgT) rowgemm_plan_destroy(&q);
        convs2d_plan_destroy(L.s2d); L.s2d = nullptr;
    }
    e->layers.clear();
    for (int i = 0; i < B_COUNT; ++i) {
        if (e->buf[i]) { (void)hipFree(e->buf[i]); e->buf[i] = nullptr; }
        if (e->pf_tmp[i]) { (void)hipFree(e->pf_tmp[i]); e->pf_tmp[i] = nullptr; }
        for (ltk_engine::PfSlot& sl : e->pfs)
            if (sl.cat[i]) { (void)hipFree(sl.cat[i]); sl.cat[i] = nullptr; }
    }
    for (ltk_engine::PfSlot& sl : e->pfs) {
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.ev_read) (void)hipEventDestroy(sl.ev_read);
        sl = ltk_engine::PfSlot();
    }
    e->alt_frames = 0;
    for (ltk_engine::SoloSeq& q : e->solo_seq) q = ltk_engine::SoloSeq();
    if (e->d_tab_next) { (void)hipFree(e->d_tab_next); e->d_tab_next = nullptr; }
    if (e->d_head) { (void)hipFree(e->d_head); e->d_head = nullptr; }
    conv7_plan_destroy(e->c7);
    e->c7 = nullptr;
    audio0_plan_destroy(e->a0);
    e->a0 = nullptr;
    audio3_plan_destroy(e->a3);
    e->a3 = nullptr;
    e->loaded = false;
}

// Wire the layer program: buffers, channel offsets (torch.cat), spatial dims.
int build_program(ltk_engine* e, const ltk_named_tensor* sd, int n) {
    wav2lip_unload(e);
    size_t* bh = e->buf_halfs;
    for (int i = 0; i < B_COUNT; ++i) bh[i] = 0;
    bh[B_MEL] = 80 * 16 * 8;
    bh[B_X0] = 65536 * 8;
    bh[B_OUT32] = 65536 * 32;
    for (int k = 0; k < 8; ++k) {
        const int hw = kFeatHW[7 - k];
        bh[B_CAT0 + k] = (size_t)hw * hw * (kDecCh[k] + kFeatCh[7 - k]);
    }
    int rc;
    // ---- audio encoder (wav2lip_v2.py:132): MEL -> AT0/AT1 ping-pong
    {
        int H = 80, W = 16, in_buf = B_MEL, in_ld = 8, pp = 0;
        for (const LayerDef& d : kAudio) {
            Layer L;
            // audio_encoder.11: the 3x3 "valid" conv on the 3x3 map = a GEMM over the flattened map with one row per frame (K = 2304),
            // like face_encoder_blocks.7.0: rowgemm for launches of <= 32 frames (16 blocks of the first-generation kernel streamed its
            // 2.4 MB of weights in 26 us - the longest launch of the audio branch, which heads the critical path under knob PREFETCH)
            const bool flat = !d.transposed && d.pad == 0 && d.k > 1 && d.k == H && d.k == W && d.cin % 64 == 0 && !knob(K_NO_FLATTEN);
            const int oh = (H + 2 * d.pad - d.k) / d.sh + 1, ow = (W + 2 * d.pad - d.k) / d.sw + 1;
            // audio_encoder.6 .. .10 (output maps 9 x 6 and 3 x 3: <= 864 rows per 16-frame launch): rowconv (build_layer)
            const bool small = !flat && d.k == 3 && d.pad == 1 && oh * ow <= 54 && d.cin % 32 == 0;
            if ((rc = build_layer(e, d, sd, n, &L, H * W, flat ? in_ld : 0, small ? -1 : 0))) return rc;
            L.audio = true;
            L.in_buf = in_buf; L.in_ld = in_ld; L.in_coff = 0; L.H = H; L.W = W;
            if (flat) { L.Ho = 1; L.Wo = 1; L.H = 1; L.W = 1; L.in_ld = d.k * d.k * in_ld; }
            else
            L.plan.out_dims(H, W, &L.Ho, &L.Wo);
            L.out_buf = B_AT0 + pp; L.out_ld = d.cout; L.out_coff = 0;
            bump(&bh[L.out_buf], (size_t)L.Ho * L.Wo * d.cout);
            L.macs = (double)d.cin * d.cout * d.k * d.k * L.Ho * L.Wo;
            e->layers.push_back(L);
            in_buf = L.out_buf; in_ld = d.cout; H = L.Ho; W = L.Wo; pp ^= 1;
        }
    }
    const int audio_emb_buf = e->layers.back().out_buf;
    // ---- face encoder (wav2lip_v2.py:136-140): block i ends in CAT[7-i] at channel offset dec_ch
    {
        int H = 256, W = 256, in_buf = B_X0, in_ld = 8, in_coff = 0, pp = 0;
        const int nl = (int)(sizeof(kFaceEnc) / sizeof(kFaceEnc[0]));
        for (int li = 0; li < nl; ++li) {
            const BlockLayer& bl = kFaceEnc[li];
            const bool last = (li + 1 == nl) || kFaceEnc[li + 1].block != bl.block;
            Layer L;
            // the 4x4 "valid" conv on the 4x4 map: a 1x1 conv over the flattened map (needs in_ld % 64 == 0)
            const bool flat = !bl.d.transposed && bl.d.pad == 0 && bl.d.k > 1 && bl.d.k == H && bl.d.k == W &&
                              bl.d.cin % 64 == 0 && !knob(K_NO_FLATTEN);
            if ((rc = build_layer(e, bl.d, sd, n, &L, H * W, flat ? in_ld : 0, H == W ? W : 0))) return rc;
            L.in_buf = in_buf; L.in_ld = in_ld; L.in_coff = in_coff; L.H = H; L.W = W;
            if (flat) {
                L.Ho = 1; L.Wo = 1;
                L.H = 1; L.W = 1;                                      // one "pixel" per image
                L.in_ld = bl.d.k * bl.d.k * in_ld; L.in_coff = bl.d.k * bl.d.k * in_coff;
            } else {
                L.plan.out_dims(H, W, &L.Ho, &L.Wo);
            }
            if (last) {
                const int k = 7 - bl.block;
                L.out_buf = B_CAT0 + k; L.out_ld = kDecCh[k] + kFeatCh[bl.block]; L.out_coff = kDecCh[k];
                if (L.Ho != kFeatHW[bl.block] || bl.d.cout != kFeatCh[bl.block]) return fail(LTK_E_INVALID, "encoder geometry mismatch");
            } else {
                L.out_buf = B_T0 + pp; L.out_ld = bl.d.cout; L.out_coff = 0; pp ^= 1;
                bump(&bh[L.out_buf], (size_t)L.Ho * L.Wo * bl.d.cout);
            }
            L.macs = (double)bl.d.cin * bl.d.cout * bl.d.k * bl.d.k * L.Ho * L.Wo;
            L.face_enc = true;
            e->layers.push_back(L);
            in_buf = L.out_buf; in_ld = L.out_ld; in_coff = L.out_coff; H = L.Ho; W = L.Wo;
        }
    }
    // ---- decoder (wav2lip_v2.py:142-152): block k reads CAT[k-1] (all channels), ends in CAT[k][0:dec_ch)
    {
        int H = 1, W = 1, in_buf = audio_emb_buf, in_ld = 512, in_coff = 0, pp = 0;
        const int nl = (int)(sizeof(kFaceDec) / sizeof(kFaceDec[0]));
        for (int li = 0; li < nl; ++li) {
            const BlockLayer& bl = kFaceDec[li];
            const bool last = (li + 1 == nl) || kFaceDec[li + 1].block != bl.block;
            Layer L;
            if ((rc = build_layer(e, bl.d, sd, n, &L, H * W, 0, H == W ? W : 0))) return rc;
            L.in_buf = in_buf; L.in_ld = in_ld; L.in_coff = in_coff; L.H = H; L.W = W;
            L.plan.out_dims(H, W, &L.Ho, &L.Wo);
            if (last) {
                const int k = bl.block;
                L.out_buf = B_CAT0 + k; L.out_ld = kDecCh[k] + kFeatCh[7 - k]; L.out_coff = 0;
                if (L.Ho != kFeatHW[7 - k] || bl.d.cout != kDecCh[k]) return fail(LTK_E_INVALID, "decoder geometry mismatch");
            } else {
                L.out_buf = B_T0 + pp; L.out_ld = bl.d.cout; L.out_coff = 0; pp ^= 1;
                bump(&bh[L.out_buf], (size_t)L.Ho * L.Wo * bl.d.cout);
            }
            if (bl.d.transposed) L.macs = (double)bl.d.cin * bl.d.cout * bl.d.k * bl.d.k * H * W;
            else L.macs = (double)bl.d.cin * bl.d.cout * bl.d.k * bl.d.k * L.Ho * L.Wo;
            e->layers.push_back(L);
            in_buf = L.out_buf; in_ld = L.out_ld; in_coff = L.out_coff; H = L.Ho; W = L.Wo;
        }
    }
    // ---- output block conv (wav2lip_v2.py:89,154)
    {
        Layer L;
        if ((rc = build_layer(e, kOutConv, sd, n, &L, 65536))) return rc;
        L.in_buf = B_CAT0 + 7; L.in_ld = 80; L.in_coff = 0; L.H = 256; L.W = 256; L.Ho = 256; L.Wo = 256;
        L.out_buf = B_OUT32; L.out_ld = 32; L.out_coff = 0;
        L.macs = 80.0 * 32 * 9 * 65536;
        e->layers.push_back(L);
    }
    // head: plain nn.Conv2d(32,3,1) (wav2lip_v2.py:90)
    const float* hw = find_tensor(sd, n, "output_block.1.weight", 96);
    const float* hb = find_tensor(sd, n, "output_block.1.bias", 3);
    if (!hw || !hb) return fail(LTK_E_INVALID, "state_dict is missing output_block.1.{weight,bias}");
    std::vector<float> h(99);
    memcpy(h.data(), hw, 96 * sizeof(float));
    memcpy(h.data() + 96, hb, 3 * sizeof(float));
    CHK(hipMalloc((void**)&e->d_head, 99 * sizeof(float)));
    CHK(hipMemcpy(e->d_head, h.data(), 99 * sizeof(float), hipMemcpyHostToDevice));
    e->macs_per_frame = 32.0 * 3 * 65536;
    for (const Layer& L : e->layers) e->macs_per_frame += L.macs;
    apply_tile_table_impl(e->layers);
    return 0;
}

f16* bufp(ltk_engine* e, int id, int frame0) { return e->buf[id] + (size_t)frame0 * e->buf_halfs[id]; }

// Enqueue the 54 conv/convT layers for frames [0, nf) of the arena on `s`.  The audio encoder has no
// dependency on the face encoder until decoder block 0 (wav2lip_v2.py:132-142): its 13 small launches run on
// the aux stream beside the face encoder instead of in front of it.
// `head_outs` != nullptr (a DEVICE table): the last layer (output_block.0) also applies the 1x1 head + sigmoid and writes the
// uint8 frames (one launch and one 4 MB/frame round trip of the 32-channel map less); the caller then skips launch_head.
// `evs` != nullptr (measurement): everything on `s`, one event in front of every layer and one behind the last.
// `faces` != nullptr (a DEVICE table): the first layer reads the uint8 bank crops itself (the caller then skips launch_pack_faces).
// Knob DF_FRAMES > 0: the decoder blocks >= DF_BLOCK and the output conv run depth-first over sub-batches of that many frames
// (all their layers for frames [f0, f0 + df), then the next sub-batch), so that a producer's output is still in the 256 MiB
// Infinity Cache when its consumer reads it; every layer sees the same frames with the same weights, only the launch size changes.
// `part`: 0 the whole network; 1 the face encoder only (builds the skip cache of knob FACE_CACHE: no audio branch, no decoder);
// 2 everything but the face encoder (its skip tensors are already in the concat buffers).
int run_convs(ltk_engine* e, int nf, hipStream_t s, const OutPtrs* head_outs = nullptr, std::vector<hipEvent_t>* evs = nullptr,
              const FacePtrs* faces = nullptr, int part = 0, int par = 0, bool pf_enc = false) {
    // `par`: which set of concat buffers the launched layers use (0 = the arena's own, 1..kPfSlots = a prefetch slot, knob PREFETCH);
    // `pf_enc`: the launched layers are a prefetched face encoder running beside another call's decoder, with temporaries of its own
    auto B = [&](int id) -> f16* {
        if (id >= B_CAT0) return par ? e->pfs[par].cat[id] : e->buf[id];
        if (pf_enc && (id == B_X0 || id == B_T0 || id == B_T1)) return e->pf_tmp[id];
        return e->buf[id];
    };
    std::string err;
    const bool fork = !e->capture && e->aux && !knob(K_NO_AUX_STREAM) && !evs && part != 1;
    size_t evi = 0;
    bool joined = !fork;
    if (fork) {
        CHK(hipEventRecord(e->ev_fork, s));
        CHK(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
    }
    // enqueue order: the first two face-encoder launches go out before the 13 audio launches, so the main
    // stream is busy while the host is still issuing the small audio kernels (each launch costs the host
    // a few microseconds); stream order per stream is unchanged
    std::vector<Layer*> order;
    if (fork) {
        size_t first_face = 0;
        while (first_face < e->layers.size() && e->layers[first_face].audio) ++first_face;
        const size_t head = std::min(e->layers.size(), first_face + 2);
        for (size_t i = first_face; i < head; ++i) order.push_back(&e->layers[i]);
        for (size_t i = 0; i < first_face; ++i) order.push_back(&e->layers[i]);
        for (size_t i = head; i < e->layers.size(); ++i) order.push_back(&e->layers[i]);
    } else {
        for (Layer& L : e->layers) order.push_back(&L);
    }
    if (part != 0) {
        std::vector<Layer*> kept;
        for (Layer* L : order)
            if ((part == 1) == L->face_enc) kept.push_back(L);
        order.swap(kept);
    }
    // one layer on frames [f0, f0 + n) of the arena
    auto launch_layer = [&](Layer& L, int f0, int n, bool on_aux) -> int {
        const int bucket = frame_bucket(n);
        ConvIO io;
        io.x = B(L.in_buf) + (size_t)f0 * L.in_ld * L.H * L.W; io.N = n; io.H = L.H; io.W = L.W; io.x_ld = L.in_ld; io.x_coff = L.in_coff;
        io.y = B(L.out_buf) + (size_t)f0 * L.out_ld * L.Ho * L.Wo; io.y_ld = L.out_ld; io.y_coff = L.out_coff;
        io.res = (L.residual && !L.res_folded) ? io.x : nullptr; io.res_ld = L.in_ld; io.res_coff = L.in_coff;
        io.relu = 1;
        io.partial = pf_enc ? e->d_partial_pf : on_aux ? e->d_partial_aux : e->d_partial;
        io.partial_cap = pf_enc ? e->partial_pf_cap : on_aux ? e->partial_aux_cap : e->partial_cap;
        if (head_outs && &L == &e->layers.back()) { io.head_w = e->d_head; io.head_outs = reinterpret_cast<const uint8_t* const*>(head_outs) + f0; }
        if (knob(K_TILE_TABLE)) { io.force_pxw = L.tile[bucket].pxw; io.force_nbt = L.tile[bucket].nbt; io.force_ksplit = L.tile[bucket].ks; }
        int rc;
        if (e->c7 && knob(K_CONV7) && L.in_buf == B_X0)       // face_encoder_blocks.0.0
            rc = conv7_launch(e->c7, faces ? reinterpret_cast<const FacePtrs*>(reinterpret_cast<const uint8_t* const*>(faces) + f0) : nullptr,
                              B(B_X0) + (size_t)f0 * 65536 * 8, n, io.y, L.out_ld, L.out_coff, s, &err);
        else if (L.s2d && knob(K_CONV_S2D) && !(L.H & 1) && !(L.W & 63))                          // face_encoder_blocks.1.0 / 2.0
            rc = convs2d_launch(L.s2d, io.x, L.in_ld, L.in_coff, n, L.H, L.W, io.y, L.out_ld, L.out_coff, on_aux ? e->aux : s, &err);
        else if (e->a3 && (knob(K_AUDIO0) & 2) && L.special == 3 && L.H == 80 && L.W == 16)      // audio_encoder.3
            rc = audio3_launch(e->a3, io.x, L.in_ld, L.in_coff, n, io.y, L.out_ld, L.out_coff, on_aux ? e->aux : s, &err);
        else if (e->a0 && (knob(K_AUDIO0) & 1) && L.in_buf == B_MEL)  // audio_encoder.0: reads the float32 mel windows of the pass's table itself
            rc = audio0_launch(e->a0, reinterpret_cast<const MelPtrs*>(e->d_tab->mels.p + f0), n, io.y, L.out_ld, L.out_coff, on_aux ? e->aux : s, &err);
        // one-pixel maps: a skinny GEMM, no split-K finish launch.  Not under LTK_SPLITK=0, whose promise is ONE summation order per
        // output element whatever the launch's frame count (larger launches run these layers on conv3)
        else if (L.rowconv && L.rg.d_w && (long long)n * L.Ho * L.Wo <= std::min(knob(K_ROWCONV), kRowConvMaxRows) && knob(K_SPLITK) &&
                 (!L.audio || L.Ho * L.Wo <= knob(K_AUDIO_ROWCONV))) {
            // 3x3 layers on the 4x4 / 8x8 maps: the same weight-streaming GEMM over gathered im2col rows (same LTK_SPLITK=0 rule)
            RowConvIO rio;
            rio.x = io.x; rio.x_ld = L.in_ld; rio.x_coff = L.in_coff; rio.H = L.H; rio.W = L.W;
            rio.y = io.y; rio.y_ld = L.out_ld; rio.y_coff = L.out_coff; rio.Ho = L.Ho; rio.Wo = L.Wo;
            rio.res = io.res; rio.res_ld = io.res_ld; rio.res_coff = io.res_coff;
            rio.N = n; rio.KW = 3; rio.stride = L.rc_stride; rio.stride_w = L.rc_stride_w; rio.pad = 1; rio.relu = 1;
            rc = rowconv_launch(L.rg, rio, on_aux ? e->aux : s, &err);
        } else if (L.rgT[0].d_w && (long long)n * L.H * L.W <= std::min(knob(K_ROWCONVT), kRowConvMaxRows) && knob(K_ROWCONV) > 0 && knob(K_SPLITK)) {
            // stride-2 transposed convs on the 4x4 / 8x8 maps: four per-phase weight-streaming GEMMs in one launch (no split-K finish)
            RowConvIO rio;
            rio.x = io.x; rio.x_ld = L.in_ld; rio.x_coff = L.in_coff; rio.H = L.H; rio.W = L.W;
            rio.y = io.y; rio.y_ld = L.out_ld; rio.y_coff = L.out_coff; rio.Ho = L.Ho; rio.Wo = L.Wo;
            rio.N = n; rio.relu = 1;
            rc = rowconvT_launch(L.rgT, rio, on_aux ? e->aux : s, &err);
        } else if (!L.rowconv && L.rg.d_w && n <= kRowGemmMaxFrames && knob(K_ROWGEMM) && knob(K_SPLITK) &&
                   // the k x k expansion of a one-pixel map writes k*k*Cout contiguous columns per frame: only into a dense output
                   // (a CAT buffer's skip channels would be overwritten)
                   (L.rg_y_ld == 0 || (L.out_coff == 0 && L.out_ld * L.Ho * L.Wo == L.rg_y_ld)))
            rc = rowgemm_launch(L.rg, io.x, L.in_ld, L.in_coff, io.y, L.rg_y_ld ? L.rg_y_ld : L.out_ld, L.out_coff, n, 1,
                                on_aux ? e->aux : s, &err);
        else
            rc = conv_launch(L.plan, io, on_aux ? e->aux : s, &err);
        if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, L.name + ": " + err);
        // debug (knob SAT_CHECK): count what this layer's epilogue clamped to the fp16 limit (a fused head writes bytes: run it unfused)
        if (knob(K_SAT_CHECK) && !(io.head_w && io.head_outs))
            launch_sat_scan(io.y, n, L.out_ld / 16, L.out_coff / 16, (L.plan.Cout + 15) / 16, (long long)L.Ho * L.Wo, 0, e->d_sat, on_aux ? e->aux : s);
        return 0;
    };
    // depth-first region: [df_first, end) of `order`
    const int df = (!e->capture && !evs && nf >= std::max(1, knob(K_DF_MIN))) ? knob(K_DF_FRAMES) : 0;
    size_t df_first = order.size();
    if (df > 0 && df < nf) {
        const std::string first_name = "face_decoder_blocks." + std::to_string(std::max(1, std::min(7, knob(K_DF_BLOCK)))) + ".0";
        for (size_t i = 0; i < order.size(); ++i)
            if (order[i]->name == first_name) { df_first = i; break; }
    }
    for (size_t oi = 0; oi < df_first; ++oi) {
        Layer& L = *order[oi];
        const bool on_aux = fork && L.audio;
        if (!on_aux && !joined && !L.audio && L.in_buf >= B_AT0 && L.in_buf <= B_AT1 && L.name.rfind("face_decoder", 0) == 0) {
            CHK(hipEventRecord(e->ev_join, e->aux));
            CHK(hipStreamWaitEvent(s, e->ev_join, 0));
            joined = true;
        }
        if (evs) CHK(hipEventRecord((*evs)[evi++], s));
        const int rc = launch_layer(L, 0, nf, on_aux);
        if (rc) return rc;
        if (e->capture) {
            const int C = L.plan.Cout;
            std::vector<float>& t = e->taps[L.name];
            t.resize((size_t)nf * C * L.Ho * L.Wo);
            float* d_tmp = nullptr;
            CHK(hipMalloc((void**)&d_tmp, t.size() * sizeof(float)));
            launch_nhwc_to_nchw_f32(e->buf[L.out_buf], nf, L.Ho, L.Wo, L.out_ld, L.out_coff, C, d_tmp, s);
            CHK(hipStreamSynchronize(s));
            CHK(hipMemcpy(t.data(), d_tmp, t.size() * sizeof(float), hipMemcpyDeviceToHost));
            CHK(hipFree(d_tmp));
            e->tap_shape[L.name] = {nf, C, L.Ho, L.Wo};
        }
    }
    if (!joined) {
        CHK(hipEventRecord(e->ev_join, e->aux));
        CHK(hipStreamWaitEvent(s, e->ev_join, 0));
    }
    for (int f0 = 0; df_first < order.size() && f0 < nf; f0 += df)
        for (size_t oi = df_first; oi < order.size(); ++oi) {
            const int rc = launch_layer(*order[oi], f0, std::min(df, nf - f0), false);
            if (rc) return rc;
        }
    if (evs) CHK(hipEventRecord((*evs)[evi++], s));
    return 0;
}

// ---- Slaney mel filterbank (librosa.filters.mel semantics: htk=False, norm='slaney', float32),
// as avatars/wav2lip/audio.py:98-101 requests it (sr 16000, n_fft 800, 80 mels, 55..7600 Hz).
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}
void build_mel_basis(std::vector<float>* basis, std::vector<int32_t>* lohi, int n_bins = 401, double f_lo = 55.0,
                     double f_hi = 7600.0) {
    const int n_mels = 80;
    const double sr = 16000.0;
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(f_lo), m1 = hz_to_mel(f_hi);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
    basis->assign((size_t)n_mels * n_bins, 0.f);
    lohi->assign(2 * n_mels, 0);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int lo = n_bins, hi = 0;
        for (int k = 0; k < n_bins; ++k) {
            const double f = (sr / 2) * k / (n_bins - 1);
            const double lower = -(mel_f[i] - f) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
            const double w = fmax(0.0, lower < upper ? lower : upper) * enorm;
            const float wf = (float)w;
            (*basis)[(size_t)i * n_bins + k] = wf;
            if (wf != 0.f) { if (k < lo) lo = k; if (k + 1 > hi) hi = k + 1; }
        }
        if (lo > hi) lo = hi = 0;
        (*lohi)[2 * i] = lo; (*lohi)[2 * i + 1] = hi;
    }
}

int mirror_index(int size, int index) {  // utils/image.py:26-32
    const int turn = index / size, res = index % size;
    return (turn % 2 == 0) ? res : size - res - 1;
}

}  // namespace

// ================================================================================ C ABI
extern "C" {

const char* ltk_last_error(void) { return g_err.c_str(); }
const char* ltk_version(void) { return "ltk_hip 0.1 (gfx950)"; }

int ltk_engine_create(int device, ltk_engine** out) {
    if (!out) return fail(LTK_E_INVALID, "out is null");
    int ndev = 0;
    CHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(LTK_E_INVALID, "no such HIP device");
    CHK(hipSetDevice(device));
    std::unique_ptr<ltk_engine> e(new ltk_engine());
    e->device = device;
    CHK(hipStreamCreateWithFlags(&e->compute, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&e->aux2, hipStreamNonBlocking));
    CHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    e->partial_cap = (size_t)128 << 20;
    e->partial_aux_cap = (size_t)16 << 20;
    e->partial_pf_cap = (size_t)64 << 20;
    CHK(hipMalloc((void**)&e->d_partial, e->partial_cap));
    CHK(hipMalloc((void**)&e->d_partial_aux, e->partial_aux_cap));
    CHK(hipMalloc((void**)&e->d_partial_pf, e->partial_pf_cap));
    CHK(hipMalloc((void**)&e->d_sat, 2 * sizeof(unsigned long long)));
    CHK(hipMemset(e->d_sat, 0, 2 * sizeof(unsigned long long)));
    std::vector<float> basis;
    std::vector<int32_t> lohi;
    build_mel_basis(&basis, &lohi);
    CHK(hipMalloc((void**)&e->d_basis, basis.size() * sizeof(float)));
    CHK(hipMemcpy(e->d_basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    CHK(hipMalloc((void**)&e->d_lohi, lohi.size() * sizeof(int32_t)));
    CHK(hipMemcpy(e->d_lohi, lohi.data(), lohi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    *out = e.release();
    return LTK_OK;
}

void ltk_engine_destroy(ltk_engine* e) {
    if (!e) return;
    if (e->tm_calls)
        fprintf(stderr, "ltk: ltk_wav2lip_infer host time per call over %lu calls: prepare %.1f us, upload + pass launch %.1f us, prefetch join / serial launch %.1f us, wait for the device %.1f us\n",
                e->tm_calls, e->tm_prep / e->tm_calls, e->tm_launch / e->tm_calls, e->tm_pf / e->tm_calls, e->tm_wait / e->tm_calls);
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    wav2lip_unload(e);
    if (e->d_basis) (void)hipFree(e->d_basis);
    if (e->d_lohi) (void)hipFree(e->d_lohi);
    e->avatars.clear();
    e->mt_avatars.clear();
    drop_prog_graphs(e);
    if (e->mt) mt_graph_delete(e->mt);
    if (e->whisper) mt_graph_delete(e->whisper);
    if (e->vae_enc) mt_graph_delete(e->vae_enc);
    if (e->d_wbasis) (void)hipFree(e->d_wbasis);
    if (e->d_wlogspec) (void)hipFree(e->d_wlogspec);
    if (e->d_wpcm) (void)hipFree(e->d_wpcm);
    if (e->d_wgmax) (void)hipFree(e->d_wgmax);
    if (e->d_pe) (void)hipFree(e->d_pe);
    if (e->d_mt_feat) (void)hipFree(e->d_mt_feat);
    if (e->d_mt_lat) (void)hipFree(e->d_mt_lat);
    for (Scratch& s : e->scratch_free) (void)hipFree(s.d);
    for (hipStream_t s : e->stream_free) (void)hipStreamDestroy(s);
    if (e->d_partial) (void)hipFree(e->d_partial);
    if (e->d_partial_aux) (void)hipFree(e->d_partial_aux);
    if (e->d_partial_pf) (void)hipFree(e->d_partial_pf);
    if (e->d_sat) (void)hipFree(e->d_sat);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->aux2) (void)hipStreamDestroy(e->aux2);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    if (e->compute) (void)hipStreamDestroy(e->compute);
    delete e;
}

int ltk_engine_sync(ltk_engine* e) {
    if (!e) return fail(LTK_E_INVALID, "engine is null");
    CHK(enter_device(e->device));
    CHK(hipDeviceSynchronize());
    return LTK_OK;
}

int ltk_wav2lip_load(ltk_engine* e, const ltk_named_tensor* sd, int n, int max_frames) {
    if (!e || !sd || n <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (max_frames < 1 || max_frames > 4096) return fail(LTK_E_INVALID, "max_frames must be in [1, 4096]");
    std::lock_guard<std::mutex> g(e->mu);
    if (e->loaded) return fail(LTK_E_STATE, "a model is already loaded in this engine");
    CHK(enter_device(e->device));
    const int rc = [&]() -> int {
        const int brc = build_program(e, sd, n);
        if (brc) return brc;
        e->micro_batch = knob(K_MICROBATCH);
        if (e->micro_batch <= 0 || e->micro_batch > max_frames) e->micro_batch = max_frames;
        e->max_frames = max_frames;
        if (hipMalloc((void**)&e->d_tab, sizeof(DevTables)) != hipSuccess) return fail(LTK_E_NOMEM, "pointer table allocation failed");
        CHK(hipMemset(e->d_tab, 0, sizeof(DevTables)));
        const int arena_frames = e->micro_batch;
        for (int i = 0; i < B_COUNT; ++i) {
            if (!e->buf_halfs[i]) continue;
            const size_t bytes = e->buf_halfs[i] * arena_frames * sizeof(f16) + 4096;
            if (hipMalloc((void**)&e->buf[i], bytes) != hipSuccess) return fail(LTK_E_NOMEM, "activation arena allocation failed");
            CHK(hipMemset(e->buf[i], 0, bytes));
        }
        if (knob(K_PREFETCH)) {       // the prefetch slots (8 x 0.13 GB of concat buffers at 32 frames) + the prefetched encoder's temporaries
            e->alt_frames = std::min(arena_frames, kPrefetchMaxFrames);
            for (int i = 0; i < B_COUNT; ++i) {
                if (!e->buf_halfs[i] || !(i >= B_CAT0 || i == B_X0 || i == B_T0 || i == B_T1)) continue;
                const size_t bytes = e->buf_halfs[i] * e->alt_frames * sizeof(f16) + 4096;
                if (i < B_CAT0) {
                    if (hipMalloc((void**)&e->pf_tmp[i], bytes) != hipSuccess) return fail(LTK_E_NOMEM, "prefetch arena allocation failed");
                    CHK(hipMemset(e->pf_tmp[i], 0, bytes));
                    continue;
                }
                for (int k = 1; k <= ltk_engine::kPfSlots; ++k) {
                    if (hipMalloc((void**)&e->pfs[k].cat[i], bytes) != hipSuccess) return fail(LTK_E_NOMEM, "prefetch arena allocation failed");
                    CHK(hipMemset(e->pfs[k].cat[i], 0, bytes));
                }
            }
            for (int k = 1; k <= ltk_engine::kPfSlots; ++k) {
                CHK(hipEventCreateWithFlags(&e->pfs[k].ev_done, hipEventDisableTiming));
                CHK(hipEventCreateWithFlags(&e->pfs[k].ev_read, hipEventDisableTiming));
            }
            if (hipMalloc((void**)&e->d_tab_next, sizeof(DevTables)) != hipSuccess) return fail(LTK_E_NOMEM, "pointer table allocation failed");
            CHK(hipMemset(e->d_tab_next, 0, sizeof(DevTables)));
        }
        return LTK_OK;
    }();
    if (rc) { wav2lip_unload(e); return rc; }       // nothing half-built stays behind (the error text is already set)
    e->loaded = true;
    return LTK_OK;
}

int ltk_avatar_register(ltk_engine* e, const uint8_t* face_bank, const uint8_t* full_bank,
                        const int32_t* coords, int n, int H, int W, int* avatar_id) {
    if (!e || !face_bank || !full_bank || !coords || !avatar_id || n <= 0 || H <= 0 || W <= 0)
        return fail(LTK_E_INVALID, "bad arguments");
    for (int i = 0; i < n; ++i) {
        const int32_t* c = coords + 4 * i;  // (y1,y2,x1,x2), wav2lip_avatar.py:144
        if (c[0] < 0 || c[2] < 0 || c[1] > H || c[3] > W || c[1] <= c[0] || c[3] <= c[2])
            return fail(LTK_E_INVALID, "coords box outside the frame");
    }
    CHK(enter_device(e->device));
    auto ap = std::make_shared<Avatar>();
    Avatar& a = *ap;
    a.device = e->device;
    a.n = n; a.H = H; a.W = W;
    a.coords.assign(coords, coords + 4 * (size_t)n);
    const size_t fb = (size_t)n * 256 * 256 * 3, ub = (size_t)n * H * W * 3;
    CHK(hipMalloc((void**)&a.d_face, fb));
    CHK(hipMalloc((void**)&a.d_full, ub));
    CHK(hipMemcpy(a.d_face, face_bank, fb, hipMemcpyHostToDevice));
    CHK(hipMemcpy(a.d_full, full_bank, ub, hipMemcpyHostToDevice));
    std::lock_guard<std::mutex> g(e->pool_mu);
    const int id = e->next_avatar++;
    e->avatars[id] = ap;
    *avatar_id = id;
    return LTK_OK;
}

int ltk_avatar_release(ltk_engine* e, int avatar_id) {
    if (!e) return fail(LTK_E_INVALID, "engine is null");
    {   // prefetch slots keyed by this avatar: their data will never be asked for again, and their hold on the bank goes once the
        // prefetch that reads it has finished (ids are never reused, so a stale key could not match anyway)
        std::lock_guard<std::mutex> ge(e->mu);
        for (ltk_engine::PfSlot& sl : e->pfs)
            if (sl.avatar == avatar_id && (sl.valid || sl.hold)) {
                if (sl.filled && sl.ev_done) (void)hipEventSynchronize(sl.ev_done);
                sl.valid = false;
                sl.hold.reset();
            }
    }
    std::lock_guard<std::mutex> g(e->pool_mu);
    auto it = e->avatars.find(avatar_id);
    if (it != e->avatars.end()) { e->avatars.erase(it); return LTK_OK; }      // buffers go with the last call that still uses them
    auto mt = e->mt_avatars.find(avatar_id);                                       // ids of both kinds come from one counter
    if (mt != e->mt_avatars.end()) { e->mt_avatars.erase(mt); return LTK_OK; }
    return fail(LTK_E_STATE, "unknown avatar id");
}

int ltk_mel_step(ltk_engine* e, const float* pcm, int n_samples, const int32_t* win_start, int n_win,
                 void* d_out, void* stream) {
    if (!e || !pcm || !win_start || !d_out || n_samples <= 0 || n_win <= 0 || n_win > 1024)
        return fail(LTK_E_INVALID, "bad arguments");
    const int n_cols_total = 1 + n_samples / 200;  // librosa.stft(center=True)
    int cmin = 1 << 30, cmax = -1;
    for (int i = 0; i < n_win; ++i) {
        if (win_start[i] < 0 || win_start[i] + 16 > n_cols_total) return fail(LTK_E_INVALID, "mel window outside the spectrogram");
        if (win_start[i] < cmin) cmin = win_start[i];
        if (win_start[i] + 15 > cmax) cmax = win_start[i] + 15;
    }
    CHK(enter_device(e->device));
    const size_t pcm_bytes = (size_t)n_samples * sizeof(float);
    const size_t ws_off = (pcm_bytes + 255) / 256 * 256;
    ScratchLease sc(e, ws_off + (size_t)n_win * sizeof(int32_t));
    if (!sc.s.d) return fail(LTK_E_NOMEM, "scratch allocation failed");
    StreamLease sl(e, stream);
    CHK(hipMemcpyAsync(sc.s.d, pcm, pcm_bytes, hipMemcpyHostToDevice, sl.s));
    CHK(hipMemcpyAsync((char*)sc.s.d + ws_off, win_start, (size_t)n_win * sizeof(int32_t), hipMemcpyHostToDevice, sl.s));
    launch_mel((const float*)sc.s.d, n_samples, (const int32_t*)((char*)sc.s.d + ws_off), n_win, cmin, cmax - cmin + 1,
               e->d_basis, e->d_lohi, (float*)d_out, sl.s);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(sl.s));
    return LTK_OK;
}

// One pass over frames [0, nf) of the arena on `s`; the per-frame pointer tables are already in e->d_tab.
// bank_faces: the faces table holds uint8 bank crops (else `d_face6`: float32 NCHW test input); have_outs: the outs table holds
// the uint8 frame destinations; d_pred_f32 (test hook): float32 NCHW sigmoid output.
// geometry of a face-cache record against the concat buffers (misc_kernels.h FeatGeom): level k = face_encoder_blocks.k's output
static FeatGeom feat_geom(ltk_engine* e) {
    FeatGeom g;
    unsigned off = 0;
    for (int k = 0; k < 8; ++k) {
        const int cat = B_CAT0 + (7 - k), hw = kFeatHW[k] * kFeatHW[k];
        g.cat[k] = e->buf[cat] + (size_t)(kDecCh[7 - k] / 16) * hw * 16;     // channel blocks [dec_ch/16, +feat_ch/16) of a frame
        g.cat_stride[k] = (unsigned)e->buf_halfs[cat];
        g.off[k] = off;
        off += (unsigned)((size_t)kFeatCh[k] * hw * sizeof(f16) / 16);
    }
    g.off[8] = off;
    return g;
}

// `cached` (knob FACE_CACHE): the faces table holds the frames' skip-cache records instead of their bank crops; the face encoder
// does not run, one copy launch puts its eight outputs where it would have written them.
// Knob PREFETCH (see tune.h): `par` = the concat-buffer set this call's decoder works in; `have_feats` = the face encoder's
// outputs for this call's frames are already there (the previous call prefetched them): the pass starts at the audio encoder /
// decoder.
static int enqueue_pass(ltk_engine* e, int nf, hipStream_t s, bool bank_faces, const float* d_face6, bool fused, bool have_outs,
                        float* d_pred_f32, bool cached = false, int par = 0, bool have_feats = false) {
    const FacePtrs* d_faces = &e->d_tab->faces;
    const OutPtrs* d_outs = &e->d_tab->outs;
    const bool pack_fused = bank_faces && e->c7 && knob(K_CONV7);     // the first layer reads the bank crops itself
    if (cached) launch_feat_copy(d_faces, nf, feat_geom(e), 0, s);
    else if (have_feats) {}
    else if (bank_faces) { if (!pack_fused) launch_pack_faces(d_faces, nf, e->buf[B_X0], s); }
    else launch_pack_face6_nchw(d_face6, nf, e->buf[B_X0], s);
    if (!(e->a0 && (knob(K_AUDIO0) & 1))) launch_pack_mel(&e->d_tab->mels, nf, e->buf[B_MEL], s);     // (audio0_kernel reads the windows itself)
    const int rc = run_convs(e, nf, s, fused ? d_outs : nullptr, nullptr, (pack_fused && !cached && !have_feats) ? d_faces : nullptr,
                             (cached || have_feats) ? 2 : 0, par);
    if (rc) return rc;
    if (!fused) {
        launch_head(e->buf[B_OUT32], 32, nf, e->d_head, e->d_head + 96, have_outs ? d_outs : nullptr, d_pred_f32, s);
        CHK(hipGetLastError());
    }
    return 0;
}

constexpr size_t kMaxPassGraphs = 64;

// enqueue_pass, replayed from a captured hipGraph where the pass has no per-call arguments: the product configuration (bank crops
// in, fused head out) on the engine's own streams.  A frame count runs eagerly the first time it is seen (which also sets every
// kernel's dynamic-LDS attribute) and is captured the second time; a dependent launch costs ~3.1 us on a stream and ~2.0 us inside a
// graph (profiles/r03_ubench_launch_chain.txt), and the host issues one launch instead of ~70.  The audio-encoder branch on the aux
// stream becomes a branch of the graph (its fork / join events are captured as dependencies).
// the table of captured Wav2Lip passes (passes, pipelined variants per slot, prefetch graphs) is full: the least recently used one goes
static void evict_graph_if_full(ltk_engine* e) {
    size_t live = 0;
    for (auto& kv : e->graphs) live += kv.second.exec ? 1 : 0;
    if (live < kMaxPassGraphs) return;
    auto victim = e->graphs.end();
    for (auto it = e->graphs.begin(); it != e->graphs.end(); ++it)
        if (it->second.exec && (victim == e->graphs.end() || it->second.stamp < victim->second.stamp)) victim = it;
    if (victim == e->graphs.end()) return;
    (void)hipStreamSynchronize(e->compute);            // it may still be running for the previous call ...
    if (victim->first & (1 << 22)) (void)hipStreamSynchronize(e->aux2);     // ... a prefetch graph: on the third stream
    (void)hipGraphExecDestroy(victim->second.exec); victim->second.exec = nullptr; victim->second.seen = 1;
}

static int launch_pass(ltk_engine* e, int nf, hipStream_t s, bool bank_faces, const float* d_face6, bool have_outs, float* d_pred_f32,
                       bool cached = false, int par = 0, bool have_feats = false) {
    // the float32 NCHW output (test hook) and layer capture need the 32-channel map in memory: unfused
    const bool fused = have_outs && !d_pred_f32 && !e->capture && knob(K_HEAD_FUSED) && !knob(K_SAT_CHECK);
    // knob GRAPH: 0 never, non-zero (default 1) every eligible pass.  Measured (profiles/r04_vs_r03_same_job.txt, r04_graph_auto_ab.txt): the replay of a
    // 16-frame pass is ~5-15 us (0.5-1 %) slower on the device than the same launches issued one by one (equal from 64 frames on), the host side is
    // one launch instead of ~70: a single session's step is 0..1.8 % faster end to end depending on the box's host (three interleaved pairs on the
    // last box: 1.3836 / 1.3904 / 1.3956 ms eager, 1.3619 / 1.3699 / 1.3596 ms replayed), and a host serving hundreds of sessions sustains 512 instead
    // of 448 of them (profiles/r04_delivered_graph_ab.txt).
    const bool graphable = knob(K_GRAPH) && bank_faces && fused && e->c7 && knob(K_CONV7) && s == e->compute;
    if (!graphable) {
        const bool product = bank_faces && fused && e->c7 && knob(K_CONV7) && s == e->compute;
        if ((par || have_feats) && !product) return fail(LTK_E_STATE, "pipelined pass outside the product configuration");
        return enqueue_pass(e, nf, s, bank_faces, d_face6, fused, have_outs, d_pred_f32, cached && bank_faces, par, have_feats);
    }
    if (e->graph_epoch != knob_epoch()) {           // a knob changed (tests, tuners): the captured launch sequences are stale
        CHK(hipStreamSynchronize(s));
        drop_graphs(e);
        e->graph_epoch = knob_epoch();
    }
    // the cached pass and the pipelined variants of a frame count are different launch sequences
    ltk_engine::PassGraph& g = e->graphs[nf | (cached ? (1 << 20) : 0) | (have_feats ? (1 << 21) : 0) | (par << 24)];
    g.stamp = ++e->graph_clock;
    if (g.exec) { CHK(hipGraphLaunch(g.exec, s)); return 0; }
    if (g.seen < 0 || g.seen++ == 0) return enqueue_pass(e, nf, s, bank_faces, d_face6, fused, have_outs, d_pred_f32, cached, par, have_feats);
    evict_graph_if_full(e);
    hipGraph_t graph = nullptr;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    const int rc = enqueue_pass(e, nf, s, bank_faces, d_face6, fused, have_outs, d_pred_f32, cached, par, have_feats);
    const hipError_t ce = hipStreamEndCapture(s, &graph);       // always: the stream must leave capture mode
    if (rc) {
        // enqueue_pass failed mid-capture (possibly with the aux stream forked and never joined: EndCapture then reports an
        // unjoined capture): clear the sticky HIP error so that the next call's own checks do not report this one
        if (ce != hipSuccess) fprintf(stderr, "ltk: capture of the %d-frame pass aborted (%s)\n", nf, hipGetErrorString(ce));
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        g.seen = -1;
        return rc;
    }
    hipGraphExec_t exec = nullptr;
    hipError_t ie = ce;
    if (ce == hipSuccess && graph) ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (ie != hipSuccess || !exec) {
        // the pass still runs, launch by launch; say so once per frame count instead of failing the call
        (void)hipGetLastError();
        g.seen = -1;
        fprintf(stderr, "ltk: hipGraph capture of the %d-frame pass failed (%s); running it as separate launches\n", nf, hipGetErrorString(ie));
        return enqueue_pass(e, nf, s, bank_faces, d_face6, fused, have_outs, d_pred_f32, cached, par, have_feats);
    }
    g.exec = exec;
    CHK(hipGraphLaunch(exec, s));
    return 0;
}

// Knob PREFETCH.  A whole-pass call works in set 0; the prefetched encoder has temporaries, split-K scratch and a pointer table of its
// own and writes prefetch slots only, so nothing but a slot's own users has to wait for it (PfSlot::ev_done / ev_read).
// The face encoder of `nf` frames (bank crops in e->d_tab_next, uploaded on aux2 by the caller) into slot `slot`, on the third stream,
// its own graph per (frame count, slot).  Ordering: behind the previous prefetch (stream order) and behind the pass that last worked in
// the slot (ev_read); whoever then works in the slot waits for ev_done.
static int launch_prefetch(ltk_engine* e, int nf, int slot) {
    hipStream_t s = e->aux2;
    ltk_engine::PfSlot& sl = e->pfs[slot];
    if (sl.read) CHK(hipStreamWaitEvent(s, sl.ev_read, 0));
    auto enq = [&]() -> int { return run_convs(e, nf, s, nullptr, nullptr, &e->d_tab_next->faces, 1, slot, true); };
    int rc = 0;
    bool launched = false;
    if (!knob(K_GRAPH)) { rc = enq(); launched = true; }
    else {
        ltk_engine::PassGraph& g = e->graphs[nf | (1 << 22) | (slot << 24)];
        g.stamp = ++e->graph_clock;
        if (g.exec) { CHK(hipGraphLaunch(g.exec, s)); launched = true; }
        else if (g.seen < 0 || g.seen++ == 0) { rc = enq(); launched = true; }
        else {
            evict_graph_if_full(e);
            hipGraph_t graph = nullptr;
            CHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            rc = enq();
            const hipError_t ce = hipStreamEndCapture(s, &graph);
            if (rc) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); g.seen = -1; return rc; }      // nothing was launched
            hipGraphExec_t exec = nullptr;
            hipError_t ie = ce;
            if (ce == hipSuccess && graph) ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (ie != hipSuccess || !exec) {
                (void)hipGetLastError();
                g.seen = -1;
                fprintf(stderr, "ltk: hipGraph capture of the %d-frame prefetch failed (%s); running it as separate launches\n", nf, hipGetErrorString(ie));
                rc = enq();
            } else { g.exec = exec; CHK(hipGraphLaunch(exec, s)); }
            launched = true;
        }
    }
    // whatever reached the stream (also part of a failed eager sequence) is ordered in front of the slot's next user
    if (launched) { CHK(hipEventRecord(sl.ev_done, s)); sl.filled = true; }
    return rc;
}

int ltk_debug_tile_table_check(char* msg, int cap) {
    std::string m;
    const int bad = check_tile_table_impl(m);
    if (msg && cap > 0) { strncpy(msg, m.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; }
    return bad;
}

// Knob FACE_CACHE: the face encoder's outputs of every bank frame of `a`, computed by the SAME kernels a 16-frame pass runs (the
// bank is walked in chunks of 16 frames - the last chunk overlaps its predecessor so that every launch has 16 frames - hence a
// 16-frame call renders byte for byte what it renders with the knob off; other call sizes may pick other split factors for the
// small-map encoder layers, as two different call sizes do among themselves: <= 1 LSB, exact under LTK_SPLITK=0).  Under e->mu, on
// the compute stream (stream order keeps the arena and the pointer table consistent with the calls around it).
static int build_face_cache(ltk_engine* e, Avatar& a) {
    const FeatGeom g = feat_geom(e);
    const size_t rec = (size_t)g.off[8] * 16;
    if (!a.d_feat) {
        // 4.15 MB per bank frame: a long avatar is gigabytes; the budget (knob FACE_CACHE_MAX_MB, per avatar) refuses instead of
        // taking the HBM from under the arenas of later loads
        const size_t budget = (size_t)std::max(0, knob(K_FACE_CACHE_MAX_MB)) << 20;
        if (rec * a.n > budget)
            return fail(LTK_E_NOMEM, "face cache of this avatar needs " + std::to_string((rec * a.n) >> 20) + " MB, over LTK_FACE_CACHE_MAX_MB = " +
                                         std::to_string(knob(K_FACE_CACHE_MAX_MB)));
        if (hipMalloc((void**)&a.d_feat, rec * a.n) != hipSuccess) { (void)hipGetLastError(); return fail(LTK_E_NOMEM, "face-cache allocation failed"); }
        a.feat_rec_bytes = rec;
        a.feat_bytes.store(rec * a.n, std::memory_order_release);
    }
    const int chunk = std::min(std::min(16, a.n), std::min(e->micro_batch, kPackMaxFrames));
    const bool pack_fused = e->c7 && knob(K_CONV7);
    for (int f0 = 0; f0 < a.n; f0 += chunk) {
        const int first = std::min(f0, a.n - chunk);
        FacePtrs fp;
        for (int i = 0; i < chunk; ++i) fp.p[i] = a.d_face + (size_t)(first + i) * 256 * 256 * 3;
        launch_upload_tables(&fp, nullptr, nullptr, chunk, e->d_tab, e->compute);
        if (!pack_fused) launch_pack_faces(&e->d_tab->faces, chunk, e->buf[B_X0], e->compute);
        const int rc = run_convs(e, chunk, e->compute, nullptr, nullptr, pack_fused ? &e->d_tab->faces : nullptr, 1);
        if (rc) return rc;
        for (int i = 0; i < chunk; ++i) fp.p[i] = a.d_feat + (size_t)(first + i) * rec;
        launch_upload_tables(&fp, nullptr, nullptr, chunk, e->d_tab, e->compute);
        launch_feat_copy(&e->d_tab->faces, chunk, g, 1, e->compute);
        CHK(hipGetLastError());
    }
    a.feat_epoch = knob_epoch();
    return 0;
}

int ltk_avatar_face_cache_bytes(ltk_engine* e, int avatar_id, size_t* bytes) {
    if (!e || !bytes) return fail(LTK_E_INVALID, "bad arguments");
    std::lock_guard<std::mutex> g(e->pool_mu);
    auto it = e->avatars.find(avatar_id);
    if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
    *bytes = it->second->feat_bytes.load(std::memory_order_acquire);
    return LTK_OK;
}

int ltk_wav2lip_infer(ltk_engine* e, const ltk_w2l_req* reqs, int nreq, void* stream) {
    if (!e || !reqs || nreq <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->loaded) return fail(LTK_E_STATE, "ltk_wav2lip_load has not been called");
    static const bool timing = getenv("LTK_INFER_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto tp1 = tp0, tp2 = tp0, tp3 = tp0;
    CHK(enter_device(e->device));
    // resolve every frame's bank crop and mel window up front
    const bool want_cache = knob(K_FACE_CACHE) != 0;
    std::vector<int> fidx;                            // knob FACE_CACHE: (request, bank frame) of every frame
    std::vector<const uint8_t*> fptr;
    std::vector<const float*> mptr;
    std::vector<uint8_t*> optr;
    std::vector<std::shared_ptr<Avatar>> hold;        // the banks stay alive until this call has synchronised
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        for (int r = 0; r < nreq; ++r) {
            auto it = e->avatars.find(reqs[r].avatar);
            if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
            if (reqs[r].batch <= 0 || reqs[r].index < 0 || !reqs[r].d_mel || !reqs[r].d_pred) return fail(LTK_E_INVALID, "bad request");
            hold.push_back(it->second);
            const Avatar& a = *it->second;
            for (int i = 0; i < reqs[r].batch; ++i) {
                const int idx = mirror_index(a.n, reqs[r].index + i);  // wav2lip_avatar.py:121-124
                fptr.push_back(a.d_face + (size_t)idx * 256 * 256 * 3);
                if (want_cache) { fidx.push_back(r); fidx.push_back(idx); }
                mptr.push_back((const float*)reqs[r].d_mel + (size_t)i * 80 * 16);
                optr.push_back((uint8_t*)reqs[r].d_pred + (size_t)i * 65536 * 3);
            }
        }
    }
    const int total = (int)fptr.size();
    if (total > e->max_frames) return fail(LTK_E_INVALID, "more frames than max_frames given to ltk_wav2lip_load");
    Ev done_ev;
    CHK(done_ev.create());
    const hipEvent_t done = done_ev.e;
    int rc = 0;
    {
        // Calls of several host threads are serialised HERE only for the enqueue (stream order then keeps them apart on the GPU: one
        // arena, one table); the lock is released before the wait, so the next call's launches queue up behind this one's kernels
        // instead of behind this thread's wake-up (scheduler.py keeps two calls in flight).
        std::lock_guard<std::mutex> g(e->mu);
        if (stream) {  // inputs were produced on the caller's stream
            Ev ready;
            CHK(ready.create());
            CHK(hipEventRecord(ready.e, (hipStream_t)stream));
            CHK(hipStreamWaitEvent(e->compute, ready.e, 0));
        }
        const int mbs = std::min(e->micro_batch, kPackMaxFrames);
        // knob FACE_CACHE: every avatar of the call gets its skip cache on first use (and again after a knob change); the call then
        // runs without the face encoder.  The mode needs the product configuration (fused head, no layer capture).
        const bool cached = want_cache && !e->capture && knob(K_HEAD_FUSED);
        if (cached) {
            for (auto& ap : hold)
                if (!rc && (!ap->d_feat || ap->feat_epoch != knob_epoch())) rc = build_face_cache(e, *ap);
            if (!rc)
                for (int i = 0; i < total; ++i) fptr[i] = hold[fidx[2 * i]]->d_feat + (size_t)fidx[2 * i + 1] * hold[fidx[2 * i]]->feat_rec_bytes;
        } else if (!want_cache) {
            // the mode was switched off: give the records back (earlier cached calls may still read them on the compute stream)
            for (auto& ap : hold)
                if (ap->d_feat) {
                    CHK(hipStreamSynchronize(e->compute));
                    (void)hipFree(ap->d_feat);
                    ap->d_feat = nullptr;
                    ap->feat_bytes.store(0, std::memory_order_release);
                }
        }
        // knob PREFETCH: a single-request call of <= 32 frames finds the face-encoder outputs of its frames in the slot a previous call of
        // its session prefetched them into (key: avatar, first bank index, frame count), and - when it continues a session's sequence
        // (it was a hit, or it starts where a recent solo call of the same avatar and size ended) - prefetches the next call's in turn
        const bool solo = nreq == 1 && !cached && knob(K_PREFETCH) && e->alt_frames > 0 && total <= std::min(e->alt_frames, mbs) &&
                          !e->capture && knob(K_HEAD_FUSED) && e->c7 && knob(K_CONV7);
        const int first = reqs[0].index;
        int slot = 0;
        if (solo)
            for (int k = 1; k <= ltk_engine::kPfSlots && !slot; ++k) {
                const ltk_engine::PfSlot& sl = e->pfs[k];
                if (sl.valid && sl.avatar == reqs[0].avatar && sl.first == first && sl.nf == total && sl.epoch == knob_epoch()) slot = k;
            }
        const bool hit = slot > 0;
        ltk_engine::SoloSeq* seq_rec = nullptr;
        if (solo)
            for (ltk_engine::SoloSeq& q : e->solo_seq)
                if (q.avatar == reqs[0].avatar && q.next == first && q.nf == total) { seq_rec = &q; break; }
        const bool prefetch = solo && (hit || seq_rec != nullptr);
        const int par = slot;
        if (solo) { if (hit) ++e->pf_hits; else ++e->pf_misses; }
        if (hit) {                              // the slot's data must have landed; the slot is consumed by this call
            CHK(hipStreamWaitEvent(e->compute, e->pfs[slot].ev_done, 0));
            e->pfs[slot].valid = false;
            e->pfs[slot].stamp = ++e->pf_clock;
        }
        if (timing) tp1 = std::chrono::steady_clock::now();
        for (int f0 = 0; f0 < total && !rc; f0 += mbs) {
            const int nf = std::min(mbs, total - f0);
            FacePtrs fp; MelPtrs mp; OutPtrs op;
            for (int i = 0; i < nf; ++i) { fp.p[i] = fptr[f0 + i]; mp.p[i] = mptr[f0 + i]; op.p[i] = optr[f0 + i]; }
            launch_upload_tables(hit ? nullptr : &fp, &mp, &op, nf, e->d_tab, e->compute);       // a hit does not read its own bank crops
            if (hipGetLastError() != hipSuccess) rc = fail(LTK_E_HIP, "pointer table upload failed");
            else rc = launch_pass(e, nf, e->compute, true, nullptr, true, nullptr, cached, par, hit);
        }
        if (hit) {                              // the next prefetch into this slot starts behind this pass
            if (hipEventRecord(e->pfs[slot].ev_read, e->compute) != hipSuccess) { if (!rc) rc = fail(LTK_E_HIP, "hipEventRecord failed"); }
            else e->pfs[slot].read = true;
        }
        if (timing) tp2 = std::chrono::steady_clock::now();
        if (!rc && prefetch) {
            // behind the pass (its launch costs the host ~40 us, this one ~15 us: the branch reaches the GPU ~55 us into the pass, beside the
            // audio encoder): the next call's face encoder, on the third stream, into a free slot other than this call's.
            // Victim: a slot nobody is waiting for - consumed, never used, or filled for a call that did not come within kPfStale seconds
            // (a session that jumped or left).  A slot another session still waits for is NOT taken: round-robin sessions are the worst
            // case of plain LRU (the oldest slot belongs to the session that calls next), so with more interleaved sessions than free
            // slots the surplus sessions simply run whole passes instead of evicting each other.
            constexpr double kPfStale = 1.5;
            const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            int victim = 0, busy = 0;
            for (int k = 1; k <= ltk_engine::kPfSlots; ++k) {
                if (k == slot) continue;
                const ltk_engine::PfSlot& sl = e->pfs[k];
                if (sl.valid && now - sl.filled_at < kPfStale) continue;
                // ... and among the free ones the MOST recently used: a lone session then alternates between two slots (five graphs: captured
                // within its first six calls) instead of walking all sixteen (33 graphs, each launch variant run eagerly once and captured
                // once: the first ~35 calls of a session - all of a 20-step benchmark run - paid for captures, 4.5 % on its timed line)
                // (... whose last reader is done: with calls of several sessions in flight the most recently consumed slot may still be read by
                // another session's pass, and a prefetch into it would wait for that pass instead of running beside it)
                if (knob(K_PF_LRU)) { if (!victim || sl.stamp < e->pfs[victim].stamp) victim = k; continue; }      // (A/B: the rule this replaced)
                if (sl.read && hipEventQuery(sl.ev_read) != hipSuccess) { if (!busy || sl.stamp > e->pfs[busy].stamp) busy = k; continue; }
                if (!victim || sl.stamp > e->pfs[victim].stamp) victim = k;
            }
            (void)hipGetLastError();          // hipEventQuery's hipErrorNotReady is not an error
            if (!victim) victim = busy;
            if (victim) {
            ltk_engine::PfSlot& sl = e->pfs[victim];
            sl.valid = false;
            FacePtrs nx;
            const Avatar& a = *hold[0];
            for (int i = 0; i < total; ++i) nx.p[i] = a.d_face + (size_t)mirror_index(a.n, first + total + i) * 256 * 256 * 3;
            sl.hold = hold[0];                  // (a previous prefetch into this slot is behind us on aux2: its bank may go now)
            launch_upload_tables(&nx, nullptr, nullptr, total, e->d_tab_next, e->aux2);
            // a prefetch that cannot be launched does not fail the call: this call's pass is already enqueued and complete without it (returning
            // an error here would hand the caller an error while the pass still writes its frames); the session's next call misses and runs whole
            if (launch_prefetch(e, total, victim) == 0) {
                ++e->pf_issued;
                sl.valid = true; sl.avatar = reqs[0].avatar; sl.first = first + total; sl.nf = total; sl.epoch = knob_epoch();
                sl.stamp = ++e->pf_clock;
                sl.filled_at = now;
            } else {
                (void)hipGetLastError();
                if (!e->pf_fail_logged.exchange(true))
                    fprintf(stderr, "ltk: prefetch of %d frames could not be launched (%s); such calls run whole passes\n", total, g_err.c_str());
            }
            }
        }
        if (!rc && solo) {                      // where this session's next call will start
            if (!seq_rec) {
                seq_rec = &e->solo_seq[0];
                for (ltk_engine::SoloSeq& q : e->solo_seq) if (q.stamp < seq_rec->stamp) seq_rec = &q;
            }
            seq_rec->avatar = reqs[0].avatar; seq_rec->next = first + total; seq_rec->nf = total; seq_rec->stamp = ++e->pf_clock;
        }
        if (!rc) {
            if (hipEventRecord(done, e->compute) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventRecord failed");
        }
        if (timing) tp3 = std::chrono::steady_clock::now();
    }
    if (!rc) {
        if (stream) { if (hipStreamWaitEvent((hipStream_t)stream, done, 0) != hipSuccess) rc = fail(LTK_E_HIP, "hipStreamWaitEvent failed"); }
        if (hipEventSynchronize(done) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventSynchronize failed");
    } else {
        (void)hipStreamSynchronize(e->compute);         // an error behind launches: nothing of this call may still be writing the caller's buffers when it returns
    }
    if (timing) {
        const auto tp4 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::lock_guard<std::mutex> g(e->pool_mu);
        e->tm_prep += us(tp0, tp1); e->tm_launch += us(tp1, tp2); e->tm_pf += us(tp2, tp3); e->tm_wait += us(tp3, tp4); ++e->tm_calls;
    }
    return rc;
}

int ltk_paste_back(ltk_engine* e, int avatar_id, int idx, const void* d_pred, void* out, int out_is_device, void* stream) {
    if (!e || !d_pred || !out) return fail(LTK_E_INVALID, "bad arguments");
    std::shared_ptr<Avatar> ap;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        auto it = e->avatars.find(avatar_id);
        if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
        ap = it->second;
    }
    const Avatar& a = *ap;
    if (idx < 0 || idx >= a.n) return fail(LTK_E_INVALID, "frame index outside the bank");
    CHK(enter_device(e->device));
    const int32_t* c = a.coords.data() + 4 * (size_t)idx;
    const size_t bytes = (size_t)a.H * a.W * 3;
    StreamLease sl(e, stream);
    const uint8_t* full = a.d_full + (size_t)idx * bytes;
    if (out_is_device) {
        launch_paste(full, a.H, a.W, (const uint8_t*)d_pred, c[0], c[1], c[2], c[3], (uint8_t*)out, sl.s);
        CHK(hipGetLastError());
        CHK(hipStreamSynchronize(sl.s));
        return LTK_OK;
    }
    ScratchLease sc(e, bytes);
    if (!sc.s.d) return fail(LTK_E_NOMEM, "scratch allocation failed");
    launch_paste(full, a.H, a.W, (const uint8_t*)d_pred, c[0], c[1], c[2], c[3], (uint8_t*)sc.s.d, sl.s);
    CHK(hipGetLastError());
    CHK(hipMemcpyAsync(out, sc.s.d, bytes, hipMemcpyDeviceToHost, sl.s));
    CHK(hipStreamSynchronize(sl.s));
    return LTK_OK;
}

int ltk_paste_back_batch(ltk_engine* e, int avatar_id, const int32_t* idx, const void* d_pred, int n, void* out, void* stream) {
    if (!e || !idx || !d_pred || !out || n <= 0) return fail(LTK_E_INVALID, "bad arguments");
    std::shared_ptr<Avatar> ap;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        auto it = e->avatars.find(avatar_id);
        if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
        ap = it->second;
    }
    const Avatar& a = *ap;
    for (int i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= a.n) return fail(LTK_E_INVALID, "frame index outside the bank");
    CHK(enter_device(e->device));
    const size_t bytes = (size_t)a.H * a.W * 3;
    StreamLease sl(e, stream);
    ScratchLease sc(e, bytes * n);
    if (!sc.s.d) return fail(LTK_E_NOMEM, "scratch allocation failed");
    for (int i0 = 0; i0 < n; i0 += kPasteBatch) {          // one launch per 16 frames
        const int m = std::min(kPasteBatch, n - i0);
        PasteBatch pb;
        for (int i = 0; i < m; ++i) {
            const int32_t* c = a.coords.data() + 4 * (size_t)idx[i0 + i];
            pb.full[i] = a.d_full + (size_t)idx[i0 + i] * bytes;
            pb.y1[i] = c[0]; pb.y2[i] = c[1]; pb.x1[i] = c[2]; pb.x2[i] = c[3];
        }
        launch_paste_batch(pb, m, a.H, a.W, (const uint8_t*)d_pred + (size_t)i0 * 256 * 256 * 3, (uint8_t*)sc.s.d + (size_t)i0 * bytes, bytes, sl.s);
    }
    // an error past this point must not hand the scratch back to the pool while earlier launches may still be writing it
    hipError_t pe = hipGetLastError();
    if (pe == hipSuccess) pe = hipMemcpyAsync(out, sc.s.d, bytes * n, hipMemcpyDeviceToHost, sl.s);
    const hipError_t se = hipStreamSynchronize(sl.s);
    if (pe != hipSuccess || se != hipSuccess) return fail(LTK_E_HIP, std::string("paste_back_batch: ") + hipGetErrorString(pe != hipSuccess ? pe : se));
    return LTK_OK;
}

// ------------------------------------------------------------------ test / measurement hooks
namespace {
struct DevBuf {                     // device scratch of a host-side hook: freed on every return path
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

// B frames from host tensors (warm_up, tests).  Runs as passes of at most one arena (micro-batch) each, so a start-up
// warm_up(batch_size) also works when LTK_MICROBATCH is smaller than the session batch.
int ltk_wav2lip_forward_host(ltk_engine* e, const float* mel, const float* face6, int B, float* pred) {
    if (!e || !mel || !face6 || !pred || B <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->loaded) return fail(LTK_E_STATE, "ltk_wav2lip_load has not been called");
    if (B > e->max_frames) return fail(LTK_E_INVALID, "B exceeds max_frames");
    CHK(enter_device(e->device));
    const int mb = std::min(e->micro_batch, kPackMaxFrames);
    const int cap = std::min(B, mb);
    DevBuf d_mel, d_face, d_pred;
    CHK(hipMalloc(&d_mel.p, (size_t)cap * 80 * 16 * sizeof(float)));
    CHK(hipMalloc(&d_face.p, (size_t)cap * 6 * 65536 * sizeof(float)));
    CHK(hipMalloc(&d_pred.p, (size_t)cap * 3 * 65536 * sizeof(float)));
    for (int f0 = 0; f0 < B; f0 += mb) {
        const int nf = std::min(mb, B - f0);
        CHK(hipMemcpy(d_mel.p, mel + (size_t)f0 * 1280, (size_t)nf * 1280 * sizeof(float), hipMemcpyHostToDevice));
        CHK(hipMemcpy(d_face.p, face6 + (size_t)f0 * 6 * 65536, (size_t)nf * 6 * 65536 * sizeof(float), hipMemcpyHostToDevice));
        MelPtrs mp;
        for (int i = 0; i < nf; ++i) mp.p[i] = (float*)d_mel.p + (size_t)i * 1280;
        {
            std::lock_guard<std::mutex> g(e->mu);
            launch_upload_tables(nullptr, &mp, nullptr, nf, e->d_tab, e->compute);
            CHK(hipGetLastError());
            const int rc = launch_pass(e, nf, e->compute, false, (const float*)d_face.p, false, (float*)d_pred.p);
            if (rc) return rc;
            CHK(hipStreamSynchronize(e->compute));
        }
        CHK(hipMemcpy(pred + (size_t)f0 * 3 * 65536, d_pred.p, (size_t)nf * 3 * 65536 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return LTK_OK;
}

int ltk_debug_saturation(ltk_engine* e, int reset, unsigned long long* n_at_limit, unsigned long long* n_nonfinite) {
    if (!e) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    CHK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    CHK(hipMemcpy(h, e->d_sat, sizeof(h), hipMemcpyDeviceToHost));
    if (n_at_limit) *n_at_limit = h[0];
    if (n_nonfinite) *n_nonfinite = h[1];
    if (reset) CHK(hipMemset(e->d_sat, 0, sizeof(h)));
    return LTK_OK;
}

int ltk_debug_capture(ltk_engine* e, int enable) {
    if (!e) return fail(LTK_E_INVALID, "engine is null");
    std::lock_guard<std::mutex> g(e->mu);
    e->capture = enable != 0;
    if (!enable) { e->taps.clear(); e->tap_shape.clear(); }
    return LTK_OK;
}

int ltk_debug_set_knob(const char* name, int value) {
    if (knob_set(name, value)) return fail(LTK_E_INVALID, std::string("unknown knob ") + (name ? name : "(null)"));
    return LTK_OK;
}

int ltk_debug_get(ltk_engine* e, const char* layer, float* out, size_t n_floats) {
    if (!e || !layer || !out) return fail(LTK_E_INVALID, "bad arguments");
    std::lock_guard<std::mutex> g(e->mu);
    auto it = e->taps.find(layer);
    if (it == e->taps.end()) return fail(LTK_E_STATE, std::string("no capture for layer ") + layer);
    if (it->second.size() != n_floats) return fail(LTK_E_INVALID, "size mismatch: captured " + std::to_string(it->second.size()));
    memcpy(out, it->second.data(), n_floats * sizeof(float));
    return LTK_OK;
}

// Dummy inputs of the timing hooks: every frame reads one zero bank crop and one zero mel window and writes its own scratch frame, so
// that the hooks run the pass exactly as ltk_wav2lip_infer does (bank crops in, fused head out, captured graph included).
namespace {
struct TimingIO {
    DevBuf face, mel, frames;
    int setup(ltk_engine* e, int nf) {
        CHK(hipMalloc(&face.p, 65536 * 3));
        CHK(hipMemset(face.p, 0, 65536 * 3));
        CHK(hipMalloc(&mel.p, 1280 * sizeof(float)));
        CHK(hipMemset(mel.p, 0, 1280 * sizeof(float)));
        CHK(hipMalloc(&frames.p, (size_t)nf * 65536 * 3));
        FacePtrs fp; MelPtrs mp; OutPtrs op;
        for (int i = 0; i < nf; ++i) { fp.p[i] = (const uint8_t*)face.p; mp.p[i] = (const float*)mel.p; op.p[i] = (uint8_t*)frames.p + (size_t)i * 65536 * 3; }
        launch_upload_tables(&fp, &mp, &op, nf, e->d_tab, e->compute);
        CHK(hipGetLastError());
        return 0;
    }
};
}  // namespace

int ltk_wav2lip_time_convs(ltk_engine* e, int frames, int iters, float* ms_per_pass, double* macs_per_pass) {
    if (!e || frames <= 0 || iters <= 0 || !ms_per_pass) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->loaded) return fail(LTK_E_STATE, "ltk_wav2lip_load has not been called");
    if (frames > e->max_frames) return fail(LTK_E_INVALID, "frames exceeds max_frames");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    if (e->capture) return fail(LTK_E_STATE, "disable capture before timing");
    hipEvent_t t0, t1;
    CHK(hipEventCreate(&t0));
    CHK(hipEventCreate(&t1));
    // the pass as ltk_wav2lip_infer runs it (pack_mel + conv stack with the bank gather and the output head fused, knobs CONV7 /
    // HEAD_FUSED; replayed from the captured graph under knob GRAPH), same micro-batch schedule, frames going to a scratch buffer
    const int mbs = std::min(e->micro_batch, kPackMaxFrames);
    TimingIO tio;
    int rc = tio.setup(e, std::min(mbs, frames));
    if (rc) return rc;
    // knob PREFETCH: a session's consecutive <= 32-frame calls are pipelined across calls (tune.h); what is timed is then that steady
    // state - every pass finds its face-encoder outputs prefetched and prefetches the next pass's (dummy bank crops here) - which
    // is what the session's calls enqueue from the third call on
    for (ltk_engine::PfSlot& sl : e->pfs) sl.valid = false;        // the timing passes fill slots 1 and 2 with dummy crops
    const bool pipe = knob(K_PREFETCH) && e->alt_frames > 0 && frames <= std::min(e->alt_frames, mbs) && knob(K_HEAD_FUSED) && e->c7 && knob(K_CONV7);
    if (pipe) {
        FacePtrs nx;
        for (int i = 0; i < frames; ++i) nx.p[i] = (const uint8_t*)tio.face.p;
        launch_upload_tables(&nx, nullptr, nullptr, frames, e->d_tab_next, e->aux2);
    }
    int cur = 0;              // slot this pass works in (0: the priming pass runs the whole network in the arena's own set)
    auto pass = [&]() -> int {
        int prc = 0;
        if (pipe) {
            if (cur) CHK(hipStreamWaitEvent(e->compute, e->pfs[cur].ev_done, 0));
            prc = launch_pass(e, frames, e->compute, true, nullptr, true, nullptr, false, cur, cur != 0);
            if (cur) {
                if (hipEventRecord(e->pfs[cur].ev_read, e->compute) != hipSuccess) { if (!prc) prc = fail(LTK_E_HIP, "hipEventRecord failed"); }
                else e->pfs[cur].read = true;
            }
            const int nxt = cur == 1 ? 2 : 1;
            if (!prc) prc = launch_prefetch(e, frames, nxt);
            cur = nxt;
            return prc;
        }
        for (int f0 = 0; f0 < frames && !prc; f0 += mbs) prc = launch_pass(e, std::min(mbs, frames - f0), e->compute, true, nullptr, true, nullptr);
        return prc;
    };
    if (pipe) { rc = pass(); if (!rc) rc = pass(); if (!rc) rc = pass(); if (rc) return rc; }     // prime, then both slots seen once (eager)
    rc = pass();              // warm (eager)
    if (!rc) rc = pass();     // warm (captures the graph under knob GRAPH)
    if (rc) return rc;
    CHK(hipEventRecord(t0, e->compute));
    for (int i = 0; i < iters && !rc; ++i) rc = pass();
    if (rc) return rc;
    CHK(hipEventRecord(t1, e->compute));
    CHK(hipEventSynchronize(t1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, t0, t1));
    *ms_per_pass = ms / iters;
    if (macs_per_pass) *macs_per_pass = (e->macs_per_frame - (knob(K_HEAD_FUSED) ? 0.0 : 32.0 * 3 * 65536)) * frames;
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    return LTK_OK;
}

int ltk_wav2lip_prefetch_stats(ltk_engine* e, unsigned long long* hits, unsigned long long* misses, unsigned long long* issued) {
    if (!e) return fail(LTK_E_INVALID, "engine is null");
    std::lock_guard<std::mutex> g(e->mu);
    if (hits) *hits = e->pf_hits;
    if (misses) *misses = e->pf_misses;
    if (issued) *issued = e->pf_issued;
    return LTK_OK;
}

int ltk_program_graph_count(ltk_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> g(e->mu);
    int n = 0;
    for (auto& kv : e->prog_graphs) n += kv.second.exec ? 1 : 0;
    return n;
}

int ltk_wav2lip_graph_count(ltk_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> g(e->mu);
    int n = 0;
    for (auto& kv : e->graphs) n += kv.second.exec ? 1 : 0;
    return n;
}

int ltk_wav2lip_layer_count(ltk_engine* e) {
    if (!e || !e->loaded) return 0;
    return (int)e->layers.size();
}

int ltk_wav2lip_layer_name(ltk_engine* e, int layer, char* buf, int buf_len) {
    if (!e || !e->loaded || layer < 0 || layer >= (int)e->layers.size() || !buf || buf_len <= 0) return fail(LTK_E_INVALID, "bad arguments");
    snprintf(buf, (size_t)buf_len, "%s", e->layers[layer].name.c_str());
    return LTK_OK;
}

int ltk_wav2lip_set_layer_tile(ltk_engine* e, int layer, int bucket, int pxw, int nbt, int ksplit) {
    if (!e || !e->loaded || layer < 0 || layer >= (int)e->layers.size() || bucket < 0 || bucket > 4) return fail(LTK_E_INVALID, "bad arguments");
    std::lock_guard<std::mutex> g(e->mu);
    Layer::Tile& t = e->layers[layer].tile[bucket];
    t.pxw = (signed char)pxw; t.nbt = (signed char)nbt; t.ks = (signed char)ksplit;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->compute);
    drop_graphs(e);                    // captured passes carry the old tile choice
    return LTK_OK;
}

int ltk_wav2lip_time_layers(ltk_engine* e, int frames, int iters, float* ms_per_layer, int n_layers) {
    if (!e || frames <= 0 || iters <= 0 || !ms_per_layer) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->loaded) return fail(LTK_E_STATE, "ltk_wav2lip_load has not been called");
    if (frames > e->micro_batch || frames > kPackMaxFrames) return fail(LTK_E_INVALID, "frames exceeds one arena pass");
    if (n_layers != (int)e->layers.size()) return fail(LTK_E_INVALID, "n_layers != ltk_wav2lip_layer_count");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    if (e->capture) return fail(LTK_E_STATE, "disable capture before timing");
    std::vector<hipEvent_t> evs(e->layers.size() + 1);
    for (auto& ev : evs) CHK(hipEventCreate(&ev));
    const bool fused = knob(K_HEAD_FUSED) != 0;
    TimingIO tio;
    int rc = tio.setup(e, frames);
    if (rc) return rc;
    const OutPtrs* d_outs = fused ? &e->d_tab->outs : nullptr;
    const FacePtrs* d_faces = (e->c7 && knob(K_CONV7)) ? &e->d_tab->faces : nullptr;
    std::vector<double> acc(e->layers.size(), 0.0);
    rc = run_convs(e, frames, e->compute, d_outs, nullptr, d_faces);     // warm
    for (int it = 0; it < iters && !rc; ++it) {
        rc = run_convs(e, frames, e->compute, d_outs, &evs, d_faces);
        if (rc) break;
        CHK(hipEventSynchronize(evs.back()));
        for (size_t i = 0; i < e->layers.size(); ++i) {
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
            acc[i] += ms;
        }
    }
    for (auto& ev : evs) (void)hipEventDestroy(ev);
    if (rc) return rc;
    for (size_t i = 0; i < e->layers.size(); ++i) ms_per_layer[i] = (float)(acc[i] / iters);
    return LTK_OK;
}

int ltk_conv2d_f16(ltk_engine* e, const void* d_x, int N, int H, int W, int Cin,
                   const float* weight, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                   int transposed, int out_pad, const float* scale, const float* shift,
                   const void* d_res, int relu, void* d_y, int iters, float* ms_avg) {
    if (!e || !d_x || !weight || !d_y) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    ConvPlan plan;
    std::string err;
    int rc = conv_plan_create(&plan, weight, Cin, Cout, kh, kw, sh, sw, ph, pw, transposed != 0, out_pad, scale, shift, &err, H * W);
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, err);
    ConvIO io;
    io.partial = e->d_partial; io.partial_cap = e->partial_cap;
    io.x = (const f16*)d_x; io.N = N; io.H = H; io.W = W; io.x_ld = plan.Cin; io.x_coff = 0;
    io.y = (f16*)d_y; io.y_ld = Cout; io.y_coff = 0;
    io.res = (const f16*)d_res; io.res_ld = Cout; io.res_coff = 0;
    io.relu = relu;
    hipStream_t s = e->compute;
    std::lock_guard<std::mutex> g(e->mu);
    rc = conv_launch(plan, io, s, &err);
    if (!rc && iters > 0 && ms_avg) {
        hipEvent_t t0, t1;
        (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
        (void)hipEventRecord(t0, s);
        for (int i = 0; i < iters && !rc; ++i) rc = conv_launch(plan, io, s, &err);
        (void)hipEventRecord(t1, s);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        *ms_avg = ms / iters;
        (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    }
    hipError_t he = hipStreamSynchronize(s);
    conv_plan_destroy(&plan);
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, err);
    if (he != hipSuccess) return fail(LTK_E_HIP, std::string("conv kernel: ") + hipGetErrorString(he));
    return LTK_OK;
}

int ltk_groupnorm_f16(ltk_engine* e, const void* d_x, int N, int C, int P, int groups, float eps, const float* gamma, const float* beta,
                      int silu, int impl, int out_fp8, float out_scale, void* d_y, int iters, float* ms_avg) {
    if (!e || !d_x || !d_y || !gamma || !beta || N <= 0 || C <= 0 || P <= 0 || groups <= 0 || C % groups || C % 16 || (out_fp8 && C % 32))
        return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    const bool fits_group = gn_group_fits(C, P, groups);
    const int members = gn_coop_members(C, P, groups);
    if (impl == 0) impl = (knob(K_MT_GN1) && fits_group) ? 2 : (knob(K_GN_COOP) && members) ? 3 : 1;
    if ((impl == 2 && !fits_group) || (impl == 3 && !members) || impl < 1 || impl > 3) return fail(LTK_E_INVALID, "this GroupNorm kernel does not serve the shape");
    float *d_gamma = nullptr, *d_beta = nullptr, *d_partial = nullptr;
    unsigned *d_slots = nullptr, *err_host = nullptr, *err_dev = nullptr;
    const int segs = gn_segments(N, C, P);
    const size_t slot_words = (size_t)N * (C / 16) * std::max(members, 1) * 8;
    hipStream_t s = e->compute;
    std::lock_guard<std::mutex> g(e->mu);
    int rc = LTK_OK;
    auto cleanup = [&]() {
        if (d_gamma) (void)hipFree(d_gamma);
        if (d_beta) (void)hipFree(d_beta);
        if (d_partial) (void)hipFree(d_partial);
        if (d_slots) (void)hipFree(d_slots);
        if (err_host) (void)hipHostFree(err_host);
    };
    if (hipMalloc((void**)&d_gamma, C * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_beta, C * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&d_partial, (size_t)N * (C / 16) * segs * 32 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&d_slots, slot_words * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&err_host, sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&err_dev, err_host, 0) != hipSuccess) { cleanup(); return fail(LTK_E_HIP, "allocation failed"); }
    *err_host = 0u;
    (void)hipMemcpyAsync(d_gamma, gamma, C * sizeof(float), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(d_beta, beta, C * sizeof(float), hipMemcpyHostToDevice, s);
    const f16* x = (const f16*)d_x;
    const int ycb = out_fp8 ? C / 32 : C / 16;
    auto run = [&]() {
        if (impl == 3) {
            launch_gn_coop_reset(d_slots, slot_words, s);
            launch_gn_coop(x, N, C / 16, 0, C, P, groups, eps, d_slots, err_dev, d_gamma, d_beta, silu, (f16*)d_y, ycb, 0, out_fp8 ? 1 : 0, out_scale, s);
        } else if (impl == 2) {
            launch_gn_group(x, N, C / 16, 0, C, P, groups, eps, d_gamma, d_beta, silu, (f16*)d_y, ycb, 0, out_fp8 ? 1 : 0, out_scale, s);
        } else {
            launch_gn_stats(x, N, C / 16, 0, C, P, segs, d_partial, s);
            if (out_fp8) launch_gn_apply_fp8(x, N, C / 16, 0, C, P, groups, eps, d_partial, segs, d_gamma, d_beta, silu, out_scale, (unsigned char*)d_y, ycb, 0, s);
            else launch_gn_apply(x, N, C / 16, 0, C, P, groups, eps, d_partial, segs, d_gamma, d_beta, silu, (f16*)d_y, ycb, 0, s);
        }
    };
    run();
    if (iters > 0 && ms_avg) {
        hipEvent_t t0, t1;
        (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
        (void)hipEventRecord(t0, s);
        for (int i = 0; i < iters; ++i) run();
        (void)hipEventRecord(t1, s);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        *ms_avg = ms / iters;
        (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    }
    const hipError_t he = hipStreamSynchronize(s);
    if (he != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(LTK_E_HIP, std::string("GroupNorm kernel: ") + hipGetErrorString(he));
    else if (*reinterpret_cast<volatile unsigned*>(err_host)) rc = fail(LTK_E_HIP, "a cooperative GroupNorm block gave up waiting for its set (gn_coop_kernel)");
    cleanup();
    return rc;
}

int ltk_f32_to_e4m3(const float* in, size_t n, uint8_t* out) {
    if (!in || !out) return fail(LTK_E_INVALID, "bad arguments");
    for (size_t i = 0; i < n; ++i) out[i] = f32_to_e4m3(in[i]);
    return LTK_OK;
}

int ltk_conv2d_fp8(ltk_engine* e, const void* d_x, int N, int H, int W, int Cin, const float* weight, int Cout,
                   const float* scale, const float* shift, float act_scale, const void* d_res, int act, void* d_y, int iters,
                   float* ms_avg) {
    if (!e || !d_x || !weight || !d_y) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    ConvPlan plan;
    std::string err;
    int rc = conv_plan_create(&plan, weight, Cin, Cout, 3, 3, 1, 1, 1, 1, false, 0, scale, shift, &err, H * W,
                              (Cin % 64 == 0 && (knob(K_FP8_MX) == 2 || (knob(K_FP8_MX) == 1 && Cin >= 512))) ? 2 : 1, act_scale);
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, err);
    ConvIO io;
    io.partial = e->d_partial; io.partial_cap = e->partial_cap;
    io.x = (const f16*)d_x; io.N = N; io.H = H; io.W = W; io.x_ld = plan.Cin; io.x_coff = 0;   // 16-bit units
    io.y = (f16*)d_y; io.y_ld = Cout; io.y_coff = 0;
    io.res = (const f16*)d_res; io.res_ld = Cout; io.res_coff = 0;
    io.relu = 0; io.act = act;
    hipStream_t s = e->compute;
    std::lock_guard<std::mutex> g(e->mu);
    rc = conv_launch(plan, io, s, &err);
    if (!rc && iters > 0 && ms_avg) {
        hipEvent_t t0, t1;
        (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
        (void)hipEventRecord(t0, s);
        for (int i = 0; i < iters && !rc; ++i) rc = conv_launch(plan, io, s, &err);
        (void)hipEventRecord(t1, s);
        (void)hipEventSynchronize(t1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t0, t1);
        *ms_avg = ms / iters;
        (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    }
    hipError_t he = hipStreamSynchronize(s);
    conv_plan_destroy(&plan);
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, err);
    if (he != hipSuccess) return fail(LTK_E_HIP, std::string("conv kernel: ") + hipGetErrorString(he));
    return LTK_OK;
}

// ================================================================================ MuseTalk
int ltk_musetalk_set_fp8(ltk_engine* e, int enable, float act_scale) {
    if (!e) return fail(LTK_E_INVALID, "bad arguments");
    std::lock_guard<std::mutex> g(e->mu);
    if (e->mt) return fail(LTK_E_STATE, "ltk_musetalk_set_fp8 must precede ltk_musetalk_load");
    e->mt_fp8 = enable ? 1 : 0;
    e->mt_fp8_ascale = act_scale > 0.f ? act_scale : 8.f;
    return LTK_OK;
}

int ltk_musetalk_info(ltk_engine* e, double* macs_per_frame, double* macs_fp8_per_frame) {
    if (!e) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    if (macs_per_frame) *macs_per_frame = mt_macs_per_frame(e->mt);
    if (macs_fp8_per_frame) *macs_fp8_per_frame = mt_macs_fp8_per_frame(e->mt);
    return LTK_OK;
}

int ltk_musetalk_load(ltk_engine* e, const ltk_named_tensor* unet_sd, int n_unet, const ltk_named_tensor* vae_sd, int n_vae,
                      int max_frames) {
    if (!e || !unet_sd || !vae_sd || n_unet <= 0 || n_vae <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (max_frames < 1 || max_frames > 64) return fail(LTK_E_INVALID, "max_frames must be in [1, 64] for MuseTalk");
    std::lock_guard<std::mutex> g(e->mu);
    if (e->mt) return fail(LTK_E_STATE, "a MuseTalk model is already loaded in this engine");
    CHK(enter_device(e->device));
    MtGraph* mg = mt_graph_new();
    mt_set_sat_counter(mg, e->d_sat);
    mt_set_fp8(mg, e->mt_fp8, e->mt_fp8_ascale);
    const int rc = mt_build(mg, unet_sd, n_unet, vae_sd, n_vae, max_frames);
    if (rc) {
        const std::string msg = mt_graph_error(mg);
        mt_graph_delete(mg);
        return fail(rc == -4 ? LTK_E_NOMEM : LTK_E_INVALID, "musetalk: " + msg);
    }
    // avatars/musetalk/models/unet.py:12-27 PositionalEncoding(d_model=384), first 50 positions
    std::vector<float> pe(50 * 384);
    for (int pos = 0; pos < 50; ++pos)
        for (int i = 0; i < 384; i += 2) {
            const float div = expf((float)i * (-logf(10000.0f) / 384.0f));
            pe[pos * 384 + i] = sinf((float)pos * div);
            pe[pos * 384 + i + 1] = cosf((float)pos * div);
        }
    const int arc = [&]() -> int {
        CHK(hipMalloc((void**)&e->d_pe, pe.size() * sizeof(float)));
        CHK(hipMemcpy(e->d_pe, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
        CHK(hipMalloc((void**)&e->d_mt_feat, (size_t)max_frames * 50 * 384 * sizeof(float)));
        CHK(hipMalloc((void**)&e->d_mt_lat, (size_t)max_frames * 8 * 1024 * sizeof(float)));
        return LTK_OK;
    }();
    if (arc) {                                       // a failed load leaves nothing behind and can be retried
        if (e->d_pe) { (void)hipFree(e->d_pe); e->d_pe = nullptr; }
        if (e->d_mt_feat) { (void)hipFree(e->d_mt_feat); e->d_mt_feat = nullptr; }
        if (e->d_mt_lat) { (void)hipFree(e->d_mt_lat); e->d_mt_lat = nullptr; }
        mt_graph_delete(mg);
        return arc;
    }
    e->mt = mg;
    e->mt_max_frames = max_frames;
    return LTK_OK;
}

int ltk_musetalk_avatar_register(ltk_engine* e, const float* latents, const uint8_t* full_bank, const int32_t* face_boxes,
                                 const int32_t* crop_boxes, const uint8_t* masks, const int64_t* mask_offsets, int n, int H,
                                 int W, int* avatar_id) {
    if (!e || !latents || !full_bank || !face_boxes || !crop_boxes || !masks || !mask_offsets || !avatar_id || n <= 0 || H <= 0 || W <= 0)
        return fail(LTK_E_INVALID, "bad arguments");
    for (int i = 0; i < n; ++i) {
        const int32_t* f = face_boxes + 4 * i;   // (x1,y1,x2,y2), musetalk_avatar.py:157
        const int32_t* c = crop_boxes + 4 * i;   // (x_s,y_s,x_e,y_e), myutil.py:7
        if (c[0] < 0 || c[1] < 0 || c[2] > W || c[3] > H || c[2] <= c[0] || c[3] <= c[1])
            return fail(LTK_E_INVALID, "crop box outside the frame (the reference's slicing is undefined there)");
        if (f[0] < c[0] || f[1] < c[1] || f[2] > c[2] || f[3] > c[3] || f[2] <= f[0] || f[3] <= f[1])
            return fail(LTK_E_INVALID, "face box must lie inside its crop box");
        if (mask_offsets[i + 1] - mask_offsets[i] != (int64_t)(c[3] - c[1]) * (c[2] - c[0]) * 3)
            return fail(LTK_E_INVALID, "mask size does not match its crop box");
    }
    CHK(enter_device(e->device));
    auto ap = std::make_shared<MtAvatar>();
    MtAvatar& a = *ap;
    a.device = e->device;
    a.n = n; a.H = H; a.W = W;
    a.face_box.assign(face_boxes, face_boxes + 4 * (size_t)n);
    a.crop_box.assign(crop_boxes, crop_boxes + 4 * (size_t)n);
    a.mask_off.assign(mask_offsets, mask_offsets + n + 1);
    const size_t lb = (size_t)n * 8 * 1024 * sizeof(float), ub = (size_t)n * H * W * 3, mb = (size_t)mask_offsets[n];
    CHK(hipMalloc((void**)&a.d_latents, lb));
    CHK(hipMalloc((void**)&a.d_full, ub));
    CHK(hipMalloc((void**)&a.d_masks, mb));
    CHK(hipMemcpy(a.d_latents, latents, lb, hipMemcpyHostToDevice));
    CHK(hipMemcpy(a.d_full, full_bank, ub, hipMemcpyHostToDevice));
    CHK(hipMemcpy(a.d_masks, masks, mb, hipMemcpyHostToDevice));
    std::lock_guard<std::mutex> g(e->pool_mu);
    const int id = e->next_avatar++;
    e->mt_avatars[id] = ap;
    *avatar_id = id;
    return LTK_OK;
}

// One run of a device program (the U-Net + VAE decoder pass, the Whisper encoder) on the compute stream, under e->mu.  Every op
// of a program reads and writes the program's own persistent buffers with launch arguments that depend on the frame count only,
// so the whole launch list (436 launches for a MuseTalk pass, ~60 for a Whisper step) is captured as ONE hipGraph the second time a
// (program, frame count) is seen and replayed from then on (knob GRAPH, as for the Wav2Lip pass: the first, eager run also sets
// every kernel's dynamic-LDS attribute, which a capture must not do).  The kernels that carry per-call pointers - latent / token
// gather in front, uint8 frame writer behind - stay outside the graph.  What this buys is the host side: one launch per pass
// instead of hundreds, on a host that also runs the sessions' Python.
static int run_program(ltk_engine* e, MtGraph* prog, int nf) {
    hipStream_t s = e->compute;
    if (!knob(K_GRAPH)) return mt_run(prog, nf, e->d_partial, e->partial_cap, s);
    if (e->prog_graph_epoch != knob_epoch()) {
        if (hipStreamSynchronize(s) != hipSuccess) return -2;
        drop_prog_graphs(e);
        e->prog_graph_epoch = knob_epoch();
    }
    ltk_engine::PassGraph& g = e->prog_graphs[{(const void*)prog, nf}];
    g.stamp = ++e->graph_clock;
    if (g.exec) return hipGraphLaunch(g.exec, s) == hipSuccess ? 0 : -2;
    if (g.seen < 0 || g.seen++ == 0) return mt_run(prog, nf, e->d_partial, e->partial_cap, s);
    size_t live = 0;
    for (auto& kv : e->prog_graphs) live += kv.second.exec ? 1 : 0;
    if (live >= kMaxPassGraphs) {
        auto victim = e->prog_graphs.end();
        for (auto it = e->prog_graphs.begin(); it != e->prog_graphs.end(); ++it)
            if (it->second.exec && (victim == e->prog_graphs.end() || it->second.stamp < victim->second.stamp)) victim = it;
        if (victim != e->prog_graphs.end()) {
            if (hipStreamSynchronize(s) != hipSuccess) return -2;
            (void)hipGraphExecDestroy(victim->second.exec); victim->second.exec = nullptr; victim->second.seen = 1;
        }
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); g.seen = -1; return mt_run(prog, nf, e->d_partial, e->partial_cap, s); }
    const int rc = mt_run(prog, nf, e->d_partial, e->partial_cap, s);
    const hipError_t ce = hipStreamEndCapture(s, &graph);       // always: the stream must leave capture mode
    if (rc) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); g.seen = -1; return rc; }
    hipGraphExec_t exec = nullptr;
    hipError_t ie = ce;
    if (ce == hipSuccess && graph) ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (ie != hipSuccess || !exec) {
        (void)hipGetLastError();
        g.seen = -1;
        fprintf(stderr, "ltk: hipGraph capture of a %d-frame program failed (%s); running it as separate launches\n", nf, hipGetErrorString(ie));
        return mt_run(prog, nf, e->d_partial, e->partial_cap, s);
    }
    g.exec = exec;
    return hipGraphLaunch(exec, s) == hipSuccess ? 0 : -2;
}

// latents already gathered into the graph's latent tensor; d_feat = fp32 [nf][50][384] on the device
static int mt_run_locked(ltk_engine* e, const float* d_feat, const PtrList64* feat_ptrs, int nf, const OutList64* outs,
                         float* d_image_f32) {
    hipStream_t s = e->compute;
    int cbt;
    f16* ctx = mt_ctx_in(e->mt, &cbt);
    if (feat_ptrs) launch_tokens_gather_to_cb16(*feat_ptrs, nf, 50, 384, e->d_pe, ctx, cbt, s);
    else launch_tokens_to_cb16(d_feat, nf, 50, 384, e->d_pe, ctx, cbt, 0, s);
    const int rc = run_program(e, e->mt, nf);
    if (rc) return fail(rc == -2 ? LTK_E_HIP : LTK_E_INVALID, std::string("musetalk: ") + mt_graph_error(e->mt));
    if (outs || d_image_f32) {
        OutList64 none;
        for (int i = 0; i < 64; ++i) none.p[i] = nullptr;
        f16* img = mt_vae_out(e->mt, &cbt);
        launch_vae_post(img, cbt, nf, 65536, outs ? *outs : none, d_image_f32, s);
    }
    CHK(hipGetLastError());
    return 0;
}

int ltk_musetalk_infer(ltk_engine* e, const ltk_mt_req* reqs, int nreq, void* stream) {
    if (!e || !reqs || nreq <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    CHK(enter_device(e->device));
    std::vector<const float*> lptr, fptr;
    std::vector<uint8_t*> optr;
    std::vector<std::shared_ptr<MtAvatar>> hold;      // the banks stay alive until this call has synchronised
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        for (int r = 0; r < nreq; ++r) {
            auto it = e->mt_avatars.find(reqs[r].avatar);
            if (it == e->mt_avatars.end()) return fail(LTK_E_STATE, "unknown MuseTalk avatar id");
            if (reqs[r].batch <= 0 || reqs[r].index < 0 || !reqs[r].d_feat || !reqs[r].d_pred) return fail(LTK_E_INVALID, "bad request");
            hold.push_back(it->second);
            const MtAvatar& a = *it->second;
            for (int i = 0; i < reqs[r].batch; ++i) {
                const int idx = mirror_index(a.n, reqs[r].index + i);   // musetalk_avatar.py:137-139
                lptr.push_back(a.d_latents + (size_t)idx * 8 * 1024);
                fptr.push_back((const float*)reqs[r].d_feat + (size_t)i * 50 * 384);
                optr.push_back((uint8_t*)reqs[r].d_pred + (size_t)i * 65536 * 3);
            }
        }
    }
    const int total = (int)lptr.size();
    Ev done_ev;
    CHK(done_ev.create());
    const hipEvent_t done = done_ev.e;
    int rc = 0;
    {
        std::lock_guard<std::mutex> g(e->mu);
        if (stream) {
            Ev ready;
            CHK(ready.create());
            CHK(hipEventRecord(ready.e, (hipStream_t)stream));
            CHK(hipStreamWaitEvent(e->compute, ready.e, 0));
        }
        for (int f0 = 0; f0 < total && !rc; f0 += e->mt_max_frames) {
            const int nf = std::min(e->mt_max_frames, total - f0);
            PtrList64 lp, fp;
            OutList64 op;
            for (int i = 0; i < 64; ++i) { lp.p[i] = nullptr; fp.p[i] = nullptr; op.p[i] = nullptr; }
            for (int i = 0; i < nf; ++i) { lp.p[i] = lptr[f0 + i]; fp.p[i] = fptr[f0 + i]; op.p[i] = optr[f0 + i]; }
            int cbt;
            f16* lat = mt_latent_in(e->mt, &cbt);
            launch_gather_latents(lp, nf, 8, 1024, lat, cbt, e->compute);
            rc = mt_run_locked(e, nullptr, &fp, nf, &op, nullptr);
        }
        if (!rc && hipEventRecord(done, e->compute) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventRecord failed");
    }
    if (!rc) {
        if (stream) { if (hipStreamWaitEvent((hipStream_t)stream, done, 0) != hipSuccess) rc = fail(LTK_E_HIP, "hipStreamWaitEvent failed"); }
        if (hipEventSynchronize(done) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventSynchronize failed");
        if (!rc && mt_gn_error(e->mt)) rc = fail(LTK_E_HIP, std::string("musetalk: ") + mt_graph_error(e->mt));
    } else {
        (void)hipStreamSynchronize(e->compute);         // (as in ltk_wav2lip_infer: no error return with this call's launches still in flight)
    }
    return rc;
}

int ltk_paste_blend(ltk_engine* e, int avatar_id, int idx, const void* d_pred, void* out, int out_is_device, void* stream) {
    if (!e || !d_pred || !out) return fail(LTK_E_INVALID, "bad arguments");
    const uint8_t *full, *mask;
    int H, W;
    int32_t fb[4], cb[4];
    std::shared_ptr<MtAvatar> hold;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        auto it = e->mt_avatars.find(avatar_id);
        if (it == e->mt_avatars.end()) return fail(LTK_E_STATE, "unknown MuseTalk avatar id");
        hold = it->second;
        const MtAvatar& a = *hold;
        if (idx < 0 || idx >= a.n) return fail(LTK_E_INVALID, "frame index outside the bank");
        H = a.H; W = a.W;
        full = a.d_full + (size_t)idx * H * W * 3;
        mask = a.d_masks + a.mask_off[idx];
        for (int k = 0; k < 4; ++k) { fb[k] = a.face_box[4 * idx + k]; cb[k] = a.crop_box[4 * idx + k]; }
    }
    CHK(enter_device(e->device));
    const size_t bytes = (size_t)H * W * 3;
    StreamLease sl(e, stream);
    if (out_is_device) {
        launch_paste_blend(full, H, W, (const uint8_t*)d_pred, fb[0], fb[1], fb[2], fb[3], cb[0], cb[1], cb[2], cb[3], mask, (uint8_t*)out, sl.s);
        CHK(hipGetLastError());
        CHK(hipStreamSynchronize(sl.s));
        return LTK_OK;
    }
    ScratchLease sc(e, bytes);
    if (!sc.s.d) return fail(LTK_E_NOMEM, "scratch allocation failed");
    launch_paste_blend(full, H, W, (const uint8_t*)d_pred, fb[0], fb[1], fb[2], fb[3], cb[0], cb[1], cb[2], cb[3], mask, (uint8_t*)sc.s.d, sl.s);
    CHK(hipGetLastError());
    CHK(hipMemcpyAsync(out, sc.s.d, bytes, hipMemcpyDeviceToHost, sl.s));
    CHK(hipStreamSynchronize(sl.s));
    return LTK_OK;
}

// ------------------------------------------------------------------ frame egress (base_avatar.py:384-453)
struct ltk_egress {
    int H = 0, W = 0;
    std::mutex mu;                     // one frame at a time per session (the reference's process thread is serial)
    uint8_t* d_cache[2] = {nullptr, nullptr};   // [0] _last_silent_frame, [1] _last_speaking_frame
    bool have[2] = {false, false};
    uint8_t* d_frame = nullptr;        // composite / uploaded frame
    uint8_t* d_out = nullptr;          // converted frame before the D2H copy
    uint8_t* d_wm = nullptr;
    int wm_x = 0, wm_y = 0, wm_w = 0, wm_h = 0, wm_b = 0, wm_g = 0, wm_r = 0;
};

int ltk_egress_open(ltk_engine* e, int H, int W, ltk_egress** out) {
    if (!e || !out || H <= 0 || W <= 0) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    ltk_egress* s = new ltk_egress();
    s->H = H; s->W = W;
    const size_t bytes = (size_t)H * W * 3;
    if (hipMalloc((void**)&s->d_cache[0], bytes) != hipSuccess || hipMalloc((void**)&s->d_cache[1], bytes) != hipSuccess ||
        hipMalloc((void**)&s->d_frame, bytes) != hipSuccess || hipMalloc((void**)&s->d_out, bytes) != hipSuccess) {
        (void)hipFree(s->d_cache[0]); (void)hipFree(s->d_cache[1]); (void)hipFree(s->d_frame); (void)hipFree(s->d_out);
        delete s;
        return fail(LTK_E_NOMEM, "egress session buffers");
    }
    *out = s;
    return LTK_OK;
}

int ltk_egress_close(ltk_engine* e, ltk_egress* s) {
    if (!e || !s) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    {
        std::lock_guard<std::mutex> g(s->mu);
        (void)hipFree(s->d_cache[0]); (void)hipFree(s->d_cache[1]); (void)hipFree(s->d_frame); (void)hipFree(s->d_out); (void)hipFree(s->d_wm);
    }
    delete s;
    return LTK_OK;
}

int ltk_egress_watermark(ltk_engine* e, ltk_egress* s, const uint8_t* mask, int x, int y, int w, int h, int b, int g, int r) {
    if (!e || !s) return fail(LTK_E_INVALID, "bad arguments");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> gd(s->mu);
    (void)hipFree(s->d_wm);
    s->d_wm = nullptr;
    s->wm_w = s->wm_h = 0;
    if (!mask) return LTK_OK;
    if (w <= 0 || h <= 0) return fail(LTK_E_INVALID, "empty watermark rectangle");
    CHK(hipMalloc((void**)&s->d_wm, (size_t)w * h));
    CHK(hipMemcpy(s->d_wm, mask, (size_t)w * h, hipMemcpyHostToDevice));
    s->wm_x = x; s->wm_y = y; s->wm_w = w; s->wm_h = h; s->wm_b = b; s->wm_g = g; s->wm_r = r;
    return LTK_OK;
}

int ltk_egress_frame(ltk_engine* e, ltk_egress* s, const ltk_egress_req* q, uint8_t* h_out, void* stream) {
    if (!e || !s || !q || !h_out) return fail(LTK_E_INVALID, "bad arguments");
    const int H = s->H, W = s->W;
    const size_t bytes = (size_t)H * W * 3;
    if (q->format != LTK_FMT_BGR24 && q->format != LTK_FMT_I420) return fail(LTK_E_INVALID, "unknown output format");
    if (q->format == LTK_FMT_I420 && ((H | W) & 1)) return fail(LTK_E_INVALID, "I420 needs even frame dimensions");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> gs(s->mu);
    StreamLease sl(e, stream);
    const uint8_t* src = nullptr;
    std::shared_ptr<Avatar> hold_w;           // keep the bank alive until the stream has been synchronised below
    std::shared_ptr<MtAvatar> hold_m;
    if (q->source == LTK_SRC_HOST) {
        if (!q->h_frame) return fail(LTK_E_INVALID, "LTK_SRC_HOST without h_frame");
        CHK(hipMemcpyAsync(s->d_frame, q->h_frame, bytes, hipMemcpyHostToDevice, sl.s));
        src = s->d_frame;
    } else if (q->source == LTK_SRC_WAV2LIP) {
        {
            std::lock_guard<std::mutex> g(e->pool_mu);
            auto it = e->avatars.find(q->avatar);
            if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
            hold_w = it->second;
        }
        const Avatar& a = *hold_w;
        if (q->idx < 0 || q->idx >= a.n) return fail(LTK_E_INVALID, "frame index outside the bank");
        if (a.H != H || a.W != W) return fail(LTK_E_INVALID, "avatar frame size differs from the egress session");
        const uint8_t* full = a.d_full + (size_t)q->idx * bytes;
        if (q->d_pred) {
            const int32_t* c = a.coords.data() + 4 * (size_t)q->idx;
            launch_paste(full, H, W, (const uint8_t*)q->d_pred, c[0], c[1], c[2], c[3], s->d_frame, sl.s);
            src = s->d_frame;
        } else {
            src = full;                               // base_avatar.py:417: the cached frame itself
        }
    } else if (q->source == LTK_SRC_MUSETALK) {
        const uint8_t *full, *mask;
        int32_t fb[4], cb[4];
        {
            std::lock_guard<std::mutex> g(e->pool_mu);
            auto it = e->mt_avatars.find(q->avatar);
            if (it == e->mt_avatars.end()) return fail(LTK_E_STATE, "unknown MuseTalk avatar id");
            hold_m = it->second;
            const MtAvatar& a = *hold_m;
            if (q->idx < 0 || q->idx >= a.n) return fail(LTK_E_INVALID, "frame index outside the bank");
            if (a.H != H || a.W != W) return fail(LTK_E_INVALID, "avatar frame size differs from the egress session");
            full = a.d_full + (size_t)q->idx * bytes;
            mask = a.d_masks + a.mask_off[q->idx];
            for (int k = 0; k < 4; ++k) { fb[k] = a.face_box[4 * q->idx + k]; cb[k] = a.crop_box[4 * q->idx + k]; }
        }
        if (q->d_pred) {
            launch_paste_blend(full, H, W, (const uint8_t*)q->d_pred, fb[0], fb[1], fb[2], fb[3], cb[0], cb[1], cb[2], cb[3], mask,
                               s->d_frame, sl.s);
            src = s->d_frame;
        } else {
            src = full;
        }
    } else {
        return fail(LTK_E_INVALID, "unknown frame source");
    }
    const int me = q->speaking ? 1 : 0, other = me ^ 1;
    const bool blend = q->alpha >= 0.0 && q->alpha < 1.0 && s->have[other];
    // cv2.addWeighted(other, 1 - alpha, frame, alpha, 0): the weights are Python doubles there, OpenCV's 8-bit kernel
    // computes in float32
    const float w_src = (float)q->alpha, w_prev = (float)(1.0 - q->alpha);
    launch_egress(src, blend ? s->d_cache[other] : nullptr, w_prev, w_src, q->keep ? s->d_cache[me] : nullptr, s->d_wm, s->wm_x,
                  s->wm_y, s->wm_w, s->wm_h, s->wm_b, s->wm_g, s->wm_r, s->d_out, H, W, q->format == LTK_FMT_I420, q->chroma, sl.s);
    CHK(hipGetLastError());
    if (q->keep) s->have[me] = true;
    const size_t out_bytes = q->format == LTK_FMT_I420 ? bytes / 2 : bytes;
    CHK(hipMemcpyAsync(h_out, s->d_out, out_bytes, hipMemcpyDeviceToHost, sl.s));
    CHK(hipStreamSynchronize(sl.s));
    return LTK_OK;
}

int ltk_egress_batch(ltk_engine* e, ltk_egress* s, int source, int avatar, const int32_t* idx, const void* d_pred, int n, int format,
                     int chroma, uint8_t* h_out, void* stream) {
    if (!e || !s || !idx || !d_pred || !h_out || n <= 0) return fail(LTK_E_INVALID, "bad arguments");
    const int H = s->H, W = s->W;
    const size_t bytes = (size_t)H * W * 3;
    if (format != LTK_FMT_BGR24 && format != LTK_FMT_I420) return fail(LTK_E_INVALID, "unknown output format");
    if (format == LTK_FMT_I420 && ((H | W) & 1)) return fail(LTK_E_INVALID, "I420 needs even frame dimensions");
    if (source != LTK_SRC_WAV2LIP && source != LTK_SRC_MUSETALK) return fail(LTK_E_INVALID, "batch egress: Wav2Lip or MuseTalk frames only");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> gs(s->mu);
    std::shared_ptr<Avatar> hold_w;           // keep the bank alive until the stream has been synchronised below
    std::shared_ptr<MtAvatar> hold_m;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        if (source == LTK_SRC_WAV2LIP) {
            auto it = e->avatars.find(avatar);
            if (it == e->avatars.end()) return fail(LTK_E_STATE, "unknown avatar id");
            hold_w = it->second;
        } else {
            auto it = e->mt_avatars.find(avatar);
            if (it == e->mt_avatars.end()) return fail(LTK_E_STATE, "unknown MuseTalk avatar id");
            hold_m = it->second;
        }
    }
    const int bank_n = hold_w ? hold_w->n : hold_m->n, bank_h = hold_w ? hold_w->H : hold_m->H, bank_w = hold_w ? hold_w->W : hold_m->W;
    if (bank_h != H || bank_w != W) return fail(LTK_E_INVALID, "avatar frame size differs from the egress session");
    for (int i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= bank_n) return fail(LTK_E_INVALID, "frame index outside the bank");
    const size_t out_bytes = format == LTK_FMT_I420 ? bytes / 2 : bytes;
    StreamLease sl(e, stream);
    ScratchLease sc(e, (bytes + out_bytes) * n);           // [n composites][n converted frames]
    if (!sc.s.d) return fail(LTK_E_NOMEM, "scratch allocation failed");
    uint8_t* const comp = (uint8_t*)sc.s.d;
    uint8_t* const conv = comp + bytes * n;
    if (hold_w) {                                          // composites: one launch per 16 frames
        const Avatar& a = *hold_w;
        for (int i0 = 0; i0 < n; i0 += kPasteBatch) {
            const int m = std::min(kPasteBatch, n - i0);
            PasteBatch pb;
            for (int i = 0; i < m; ++i) {
                const int32_t* c = a.coords.data() + 4 * (size_t)idx[i0 + i];
                pb.full[i] = a.d_full + (size_t)idx[i0 + i] * bytes;
                pb.y1[i] = c[0]; pb.y2[i] = c[1]; pb.x1[i] = c[2]; pb.x2[i] = c[3];
            }
            launch_paste_batch(pb, m, H, W, (const uint8_t*)d_pred + (size_t)i0 * 256 * 256 * 3, comp + bytes * i0, bytes, sl.s);
        }
    } else {
        const MtAvatar& a = *hold_m;
        for (int i = 0; i < n; ++i) {
            const int32_t* fb = a.face_box.data() + 4 * (size_t)idx[i];
            const int32_t* cb = a.crop_box.data() + 4 * (size_t)idx[i];
            launch_paste_blend(a.d_full + (size_t)idx[i] * bytes, H, W, (const uint8_t*)d_pred + (size_t)i * 256 * 256 * 3, fb[0], fb[1], fb[2], fb[3],
                               cb[0], cb[1], cb[2], cb[3], a.d_masks + a.mask_off[idx[i]], comp + bytes * i, sl.s);
        }
    }
    // watermark + format conversion of all n composites in one launch
    launch_egress_batch(comp, bytes, n, s->d_wm, s->wm_x, s->wm_y, s->wm_w, s->wm_h, s->wm_b, s->wm_g, s->wm_r, conv, out_bytes, H, W,
                        format == LTK_FMT_I420, chroma, sl.s);
    // an error past this point must not hand the scratch back to the pool while earlier launches may still be writing it
    hipError_t pe = hipGetLastError();
    if (pe == hipSuccess) pe = hipMemcpyAsync(h_out, conv, out_bytes * n, hipMemcpyDeviceToHost, sl.s);
    const hipError_t se = hipStreamSynchronize(sl.s);
    if (pe != hipSuccess || se != hipSuccess) return fail(LTK_E_HIP, std::string("egress_batch: ") + hipGetErrorString(pe != hipSuccess ? pe : se));
    return LTK_OK;
}

int ltk_musetalk_forward_host(ltk_engine* e, const float* latents, const float* feat, int B, float* unet_out, float* image,
                              uint8_t* frames) {
    if (!e || !latents || !feat || B <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    if (B > e->mt_max_frames) return fail(LTK_E_INVALID, "B exceeds max_frames");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = e->compute;
    CHK(hipMemcpyAsync(e->d_mt_lat, latents, (size_t)B * 8 * 1024 * sizeof(float), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(e->d_mt_feat, feat, (size_t)B * 50 * 384 * sizeof(float), hipMemcpyHostToDevice, s));
    int cbt;
    f16* lat = mt_latent_in(e->mt, &cbt);
    launch_nchw_to_cb16(e->d_mt_lat, B, 8, 1024, lat, cbt, 0, s);
    float* d_img = nullptr;
    uint8_t* d_frames = nullptr;
    if (image) CHK(hipMalloc((void**)&d_img, (size_t)B * 3 * 65536 * sizeof(float)));
    if (frames) CHK(hipMalloc((void**)&d_frames, (size_t)B * 65536 * 3));
    OutList64 op;
    for (int i = 0; i < 64; ++i) op.p[i] = (frames && i < B) ? d_frames + (size_t)i * 65536 * 3 : nullptr;
    int rc = mt_run_locked(e, e->d_mt_feat, nullptr, B, &op, d_img);
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(LTK_E_HIP, "stream sync failed");
    if (!rc && unet_out) {
        int C, ld, coff, H, W;
        f16* t = mt_named(e->mt, "conv_out", &C, &ld, &coff, &H, &W);
        float* d_tmp = nullptr;
        CHK(hipMalloc((void**)&d_tmp, (size_t)B * 4 * 1024 * sizeof(float)));
        launch_nhwc_to_nchw_f32(t, B, H, W, ld, coff, 4, d_tmp, s);
        CHK(hipStreamSynchronize(s));
        CHK(hipMemcpy(unet_out, d_tmp, (size_t)B * 4 * 1024 * sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(d_tmp);
    }
    if (!rc && image) CHK(hipMemcpy(image, d_img, (size_t)B * 3 * 65536 * sizeof(float), hipMemcpyDeviceToHost));
    if (!rc && frames) CHK(hipMemcpy(frames, d_frames, (size_t)B * 65536 * 3, hipMemcpyDeviceToHost));
    if (d_img) (void)hipFree(d_img);
    if (d_frames) (void)hipFree(d_frames);
    return rc;
}

int ltk_musetalk_debug_get(ltk_engine* e, const char* name, int frames, float* out, size_t n_floats) {
    if (!e || !name || !out || frames <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    int C, ld, coff, H, W;
    f16* t = mt_named(e->mt, name, &C, &ld, &coff, &H, &W);
    if (!t) return fail(LTK_E_STATE, std::string("no MuseTalk tensor named ") + name);
    const size_t cnt = (size_t)frames * C * H * W;
    if (cnt != n_floats) return fail(LTK_E_INVALID, "size mismatch: tensor has " + std::to_string(cnt) + " floats for these frames");
    float* d_tmp = nullptr;
    CHK(hipMalloc((void**)&d_tmp, cnt * sizeof(float)));
    launch_nhwc_to_nchw_f32(t, frames, H, W, ld, coff, C, d_tmp, e->compute);
    CHK(hipStreamSynchronize(e->compute));
    CHK(hipMemcpy(out, d_tmp, cnt * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_tmp);
    return LTK_OK;
}

int ltk_musetalk_op_count(ltk_engine* e) { return (e && e->mt) ? mt_op_count(e->mt) : 0; }

int ltk_musetalk_op_name(ltk_engine* e, int op, char* buf, int buf_len, int* type) {
    if (!e || !e->mt || !buf || buf_len <= 0) return fail(LTK_E_INVALID, "bad arguments");
    const char* n = mt_op_name(e->mt, op, type);
    if (!n) return fail(LTK_E_INVALID, "no such op");
    snprintf(buf, (size_t)buf_len, "%s", n);
    return LTK_OK;
}

int ltk_musetalk_time_ops(ltk_engine* e, int frames, int iters, float* ms_per_op, int n_ops) {
    if (!e || frames <= 0 || iters <= 0 || !ms_per_op) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    if (frames > e->mt_max_frames) return fail(LTK_E_INVALID, "frames exceeds max_frames");
    if (n_ops != mt_op_count(e->mt)) return fail(LTK_E_INVALID, "n_ops != ltk_musetalk_op_count");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    std::vector<hipEvent_t> evs((size_t)n_ops + 1);
    for (auto& ev : evs) CHK(hipEventCreate(&ev));
    std::vector<double> acc((size_t)n_ops, 0.0);
    int rc = mt_run(e->mt, frames, e->d_partial, e->partial_cap, e->compute);
    for (int it = 0; it < iters && !rc; ++it) {
        rc = mt_run_timed(e->mt, frames, e->d_partial, e->partial_cap, e->compute, &evs);
        if (rc) break;
        CHK(hipEventSynchronize(evs.back()));
        for (int i = 0; i < n_ops; ++i) {
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, evs[i], evs[i + 1]));
            acc[i] += ms;
        }
    }
    for (auto& ev : evs) (void)hipEventDestroy(ev);
    if (rc) return fail(LTK_E_INVALID, std::string("musetalk: ") + mt_graph_error(e->mt));
    if (mt_gn_error(e->mt)) return fail(LTK_E_HIP, std::string("musetalk: ") + mt_graph_error(e->mt));
    for (int i = 0; i < n_ops; ++i) ms_per_op[i] = (float)(acc[i] / iters);
    return LTK_OK;
}

int ltk_musetalk_time(ltk_engine* e, int frames, int iters, float* ms_per_pass, double* macs_per_pass) {
    if (!e || frames <= 0 || iters <= 0 || !ms_per_pass) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->mt) return fail(LTK_E_STATE, "ltk_musetalk_load has not been called");
    if (frames > e->mt_max_frames) return fail(LTK_E_INVALID, "frames exceeds max_frames");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    hipEvent_t t0, t1;
    CHK(hipEventCreate(&t0));
    CHK(hipEventCreate(&t1));
    // as ltk_musetalk_infer enqueues the program: eagerly the first time a frame count is seen, then captured, then replayed
    int rc = run_program(e, e->mt, frames);
    if (!rc) rc = run_program(e, e->mt, frames);
    if (rc) return fail(LTK_E_INVALID, std::string("musetalk: ") + mt_graph_error(e->mt));
    CHK(hipEventRecord(t0, e->compute));
    for (int i = 0; i < iters && !rc; ++i) rc = run_program(e, e->mt, frames);
    if (rc) return fail(LTK_E_INVALID, std::string("musetalk: ") + mt_graph_error(e->mt));
    CHK(hipEventRecord(t1, e->compute));
    CHK(hipEventSynchronize(t1));
    if (mt_gn_error(e->mt)) return fail(LTK_E_HIP, std::string("musetalk: ") + mt_graph_error(e->mt));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, t0, t1));
    *ms_per_pass = ms / iters;
    if (macs_per_pass) *macs_per_pass = mt_macs_per_frame(e->mt) * frames;
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    return LTK_OK;
}

// ================================================================================ Whisper audio features
int ltk_whisper_load(ltk_engine* e, const ltk_named_tensor* encoder_sd, int n) {
    if (!e || !encoder_sd || n <= 0) return fail(LTK_E_INVALID, "bad arguments");
    std::lock_guard<std::mutex> g(e->mu);
    if (e->whisper) return fail(LTK_E_STATE, "a Whisper encoder is already loaded in this engine");
    CHK(enter_device(e->device));
    MtGraph* wg = mt_graph_new();
    if (mt_build_whisper_graph(wg, encoder_sd, n)) {
        const std::string msg = mt_graph_error(wg);
        mt_graph_delete(wg);
        return fail(LTK_E_INVALID, "whisper: " + msg);
    }
    std::vector<float> basis;
    std::vector<int32_t> lohi;
    build_mel_basis(&basis, &lohi, 201, 0.0, 8000.0);     // WhisperFeatureExtractor.mel_filters (slaney, 80 x 201)
    CHK(hipMalloc((void**)&e->d_wbasis, basis.size() * sizeof(float)));
    CHK(hipMemcpy(e->d_wbasis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    CHK(hipMalloc((void**)&e->d_wlogspec, (size_t)80 * 3000 * sizeof(float)));
    CHK(hipMalloc((void**)&e->d_wpcm, (size_t)480000 * sizeof(float)));
    CHK(hipMalloc((void**)&e->d_wgmax, 16));
    e->whisper = wg;
    return LTK_OK;
}

int ltk_whisper_step(ltk_engine* e, const float* pcm, int n_samples, int batch, int first_row, int row_step, int rows, void* d_out,
                     void* stream) {
    if (!e || !pcm || !d_out || n_samples <= 0 || n_samples > 479000 || batch <= 0 || rows <= 0 || rows > 64)
        return fail(LTK_E_INVALID, "bad arguments");
    if (!e->whisper) return fail(LTK_E_STATE, "ltk_whisper_load has not been called");
    CHK(enter_device(e->device));
    Ev done_ev;
    CHK(done_ev.create());
    const hipEvent_t done = done_ev.e;
    int rc = 0;
    {
        std::lock_guard<std::mutex> g(e->mu);
        hipStream_t s = e->compute;
        CHK(hipMemcpyAsync(e->d_wpcm, pcm, (size_t)n_samples * sizeof(float), hipMemcpyHostToDevice, s));
        int cbt, cb0;
        f16* mel = mt_latent_in(e->whisper, &cbt);
        launch_whisper_logmel(e->d_wpcm, n_samples, e->d_wbasis, e->d_wlogspec, e->d_wgmax, mel, s);
        rc = run_program(e, e->whisper, 1);
        if (rc) rc = fail(LTK_E_INVALID, std::string("whisper: ") + mt_graph_error(e->whisper));
        if (!rc) {
            WhisperStates st;
            for (int i = 0; i < 5; ++i) { st.p[i] = mt_whisper_state(e->whisper, i, &cbt, &cb0); st.cb0[i] = cb0; }
            launch_whisper_chunks(st, 1500, batch, first_row, row_step, rows, (float*)d_out, s);
            if (hipGetLastError() != hipSuccess) rc = fail(LTK_E_HIP, "whisper kernels failed to launch");
        }
        if (!rc && hipEventRecord(done, s) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventRecord failed");
    }
    if (!rc) {
        if (stream && hipStreamWaitEvent((hipStream_t)stream, done, 0) != hipSuccess) rc = fail(LTK_E_HIP, "hipStreamWaitEvent failed");
        if (hipEventSynchronize(done) != hipSuccess) rc = fail(LTK_E_HIP, "hipEventSynchronize failed");
    }
    return rc;
}

int ltk_whisper_debug_get(ltk_engine* e, const char* name, float* out, size_t n_floats) {
    if (!e || !name || !out) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->whisper) return fail(LTK_E_STATE, "ltk_whisper_load has not been called");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    int C, ld, coff, H, W;
    f16* t = (std::string(name) == "input_features") ? mt_named(e->whisper, "input_features", &C, &ld, &coff, &H, &W) : mt_named(e->whisper, name, &C, &ld, &coff, &H, &W);
    if (!t) return fail(LTK_E_STATE, std::string("no Whisper tensor named ") + name);
    const size_t cnt = (size_t)C * H * W;
    if (cnt != n_floats) return fail(LTK_E_INVALID, "size mismatch: tensor has " + std::to_string(cnt) + " floats");
    float* d_tmp = nullptr;
    CHK(hipMalloc((void**)&d_tmp, cnt * sizeof(float)));
    launch_nhwc_to_nchw_f32(t, 1, H, W, ld, coff, C, d_tmp, e->compute);
    CHK(hipStreamSynchronize(e->compute));
    CHK(hipMemcpy(out, d_tmp, cnt * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_tmp);
    return LTK_OK;
}

// ================================================================================ VAE encoder (avatar preparation)
int ltk_vae_encoder_load(ltk_engine* e, const ltk_named_tensor* vae_sd, int n, int max_faces) {
    if (!e || !vae_sd || n <= 0 || max_faces < 1 || max_faces > 32) return fail(LTK_E_INVALID, "bad arguments (max_faces in [1,32])");
    std::lock_guard<std::mutex> g(e->mu);
    if (e->vae_enc) return fail(LTK_E_STATE, "a VAE encoder is already loaded in this engine");
    CHK(enter_device(e->device));
    MtGraph* vg = mt_graph_new();
    if (mt_build_vae_encoder_graph(vg, vae_sd, n, 2 * max_faces)) {
        const std::string msg = mt_graph_error(vg);
        mt_graph_delete(vg);
        return fail(LTK_E_INVALID, "vae encoder: " + msg);
    }
    e->vae_enc = vg;
    e->vae_enc_faces = max_faces;
    return LTK_OK;
}

int ltk_vae_encode_faces(ltk_engine* e, const uint8_t* faces_bgr, int nfaces, const float* noise, float* latents_out) {
    if (!e || !faces_bgr || !latents_out || nfaces <= 0) return fail(LTK_E_INVALID, "bad arguments");
    if (!e->vae_enc) return fail(LTK_E_STATE, "ltk_vae_encoder_load has not been called");
    CHK(enter_device(e->device));
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = e->compute;
    uint8_t* d_faces = nullptr;
    float *d_noise = nullptr, *d_out = nullptr;
    const int cap = e->vae_enc_faces;
    CHK(hipMalloc((void**)&d_faces, (size_t)cap * 65536 * 3));
    CHK(hipMalloc((void**)&d_out, (size_t)cap * 8 * 1024 * sizeof(float)));
    if (noise) CHK(hipMalloc((void**)&d_noise, (size_t)cap * 2 * 4 * 1024 * sizeof(float)));
    int rc = 0;
    for (int f0 = 0; f0 < nfaces && !rc; f0 += cap) {
        const int nf = std::min(cap, nfaces - f0);
        CHK(hipMemcpyAsync(d_faces, faces_bgr + (size_t)f0 * 65536 * 3, (size_t)nf * 65536 * 3, hipMemcpyHostToDevice, s));
        if (noise) CHK(hipMemcpyAsync(d_noise, noise + (size_t)f0 * 2 * 4 * 1024, (size_t)nf * 2 * 4 * 1024 * sizeof(float), hipMemcpyHostToDevice, s));
        int cbt;
        launch_vae_pre(d_faces, nf, mt_latent_in(e->vae_enc, &cbt), s);
        rc = mt_run(e->vae_enc, 2 * nf, e->d_partial, e->partial_cap, s);
        if (rc) { rc = fail(LTK_E_INVALID, std::string("vae encoder: ") + mt_graph_error(e->vae_enc)); break; }
        launch_vae_latents(mt_unet_out(e->vae_enc, &cbt), nf, noise ? d_noise : nullptr, 0.18215f, d_out, s);
        CHK(hipMemcpyAsync(latents_out + (size_t)f0 * 8 * 1024, d_out, (size_t)nf * 8 * 1024 * sizeof(float), hipMemcpyDeviceToHost, s));
        CHK(hipStreamSynchronize(s));
        if (mt_gn_error(e->vae_enc)) { rc = fail(LTK_E_HIP, std::string("vae encoder: ") + mt_graph_error(e->vae_enc)); break; }
    }
    (void)hipFree(d_faces); (void)hipFree(d_out);
    if (d_noise) (void)hipFree(d_noise);
    return rc;
}

}  // extern "C"
