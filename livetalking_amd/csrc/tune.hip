// Knob table: environment read once, then plain array reads (tune.h).
#include "tune.h"

#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <map>
#include <set>
#include <vector>
#include <utility>

namespace ltk {
namespace {

struct KnobDef { const char* name; int dflt; };
const KnobDef kDefs[K_COUNT] = {
    {"LTK_CONV_V3", 1},        {"LTK_CONV_V3_S2", 1},      {"LTK_CONV_NBT", 0},          {"LTK_CONV_NC8", 0},
    {"LTK_GEMM_NC8", 4},       {"LTK_CONV_MODE", 1},       {"LTK_CONV_MIN_BLOCKS", 512}, {"LTK_CONV_PXW", 0},
    {"LTK_CONV3_NBT", 0},      {"LTK_CONV_PXW4_MIN", 448}, {"LTK_SPLITK", 1},            {"LTK_KSPLIT", 0},
    {"LTK_CONV_PERSIST", 512}, {"LTK_NO_FOLD_RESIDUAL", 0}, {"LTK_NO_FLATTEN", 0},       {"LTK_NO_AUX_STREAM", 0},
    {"LTK_MICROBATCH", 0},     {"LTK_MT_NO_QKV_FUSE", 0},  {"LTK_HEAD_FUSED", 1},
    {"LTK_CONV3_NC8", 0},      {"LTK_TILE_RULE", 1},       {"LTK_TILE_TABLE", 1},        {"LTK_CONV7", 1},           {"LTK_ATTN_WIDE", 1},       {"LTK_UPS4", 1},            {"LTK_FP8_MX", 1},
    {"LTK_ROWGEMM", 1},
    {"LTK_ROWCONV", 1024},
    {"LTK_ABLATE", 0},
    {"LTK_GRAPH", 1},
    {"LTK_DF_FRAMES", 0},
    {"LTK_DF_BLOCK", 6},
    {"LTK_DF_MIN", 32},
    {"LTK_ROWCONVT", 512},
    {"LTK_MT_ROWCONV", 1024},
    {"LTK_MT_TILE_TABLE", 1},
    {"LTK_LDS_SWZ", 1},
    {"LTK_FACE_CACHE", 0},
    {"LTK_PREFETCH", 1},
    {"LTK_AUDIO_ROWCONV", 54},
    {"LTK_MT_FUSE", 7},
    {"LTK_MT_GN1", 1},
    {"LTK_ATTN_PF", 1},
    {"LTK_SAT_CHECK", 0},
    {"LTK_CONV_S2SPLIT", 1},
    {"LTK_FACE_CACHE_MAX_MB", 16384},
    {"LTK_LIN_FK", 1},
    {"LTK_LIN_FK_BLOCKS", 512},
    {"LTK_LIN_FK_MIN_ROWS", 512},
    {"LTK_ATTN_LDS", 1},
    {"LTK_LIN_MP", 1},
    {"LTK_GN_COOP", 1},
    {"LTK_AUDIO0", 3},
    {"LTK_CONV_S2D", 1},
    {"LTK_PF_LRU", 0},
};

std::atomic<int> g_val[K_COUNT];     // knob_set (tests, tuners) may run beside launch threads reading the table
std::atomic<unsigned> g_epoch{0};
std::once_flag g_once;

void init() {
    for (int i = 0; i < K_COUNT; ++i) {
        const char* e = getenv(kDefs[i].name);
        // presence-style switches (LTK_NO_*): any value means 1 - the empty string, "true", "yes" - unless the WHOLE value
        // parses as a number (LTK_NO_AUX_STREAM=0 switches it off again)
        int v = kDefs[i].dflt;
        if (e) {
            char* end = nullptr;
            const long n = strtol(e, &end, 10);
            v = (*e != 0 && end && *end == 0) ? (int)n : 1;
        }
        g_val[i].store(v, std::memory_order_relaxed);
    }
}

}  // namespace

int knob(Knob k) {
    std::call_once(g_once, init);
    return g_val[k].load(std::memory_order_relaxed);
}

int ensure_dyn_lds(const void* func, int bytes) {
    // Runs in front of every conv launch (~70 per eager pass, from any engine / session thread): the common case is answered from
    // a per-thread cache without a lock; the process-wide table behind it records the byte count configured per (device,
    // function), so a later, larger request re-sets the attribute instead of being skipped.
    struct Seen { int dev; const void* func; int bytes; };
    thread_local std::vector<Seen> mine;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (const Seen& s : mine)
        if (s.func == func && s.dev == dev && s.bytes >= bytes) return 0;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;
    int have = 0;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = done.find({dev, func});
        if (it != done.end()) have = it->second;
        if (have < bytes) {
            const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return (int)e;
            done[{dev, func}] = have = bytes;
        }
    }
    for (Seen& s : mine)
        if (s.func == func && s.dev == dev) { s.bytes = have; return 0; }
    mine.push_back({dev, func, have});
    return 0;
}

unsigned knob_epoch() { return g_epoch.load(std::memory_order_relaxed); }

int knob_set(const char* name, int value) {
    std::call_once(g_once, init);
    if (!name) return -1;
    for (int i = 0; i < K_COUNT; ++i)
        if (!strcmp(name, kDefs[i].name) || !strcmp(name, kDefs[i].name + 4)) {
            g_val[i].store(value, std::memory_order_relaxed);
            g_epoch.fetch_add(1, std::memory_order_relaxed);
            return 0;
        }
    return -1;
}

}  // namespace ltk
