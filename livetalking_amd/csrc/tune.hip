// Knob table: environment read once, then plain array reads (tune.h).
#include "tune.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

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
    {"LTK_ABLATE", 0},
};

int g_val[K_COUNT];
std::once_flag g_once;

void init() {
    for (int i = 0; i < K_COUNT; ++i) {
        const char* e = getenv(kDefs[i].name);
        // presence-style switches (LTK_NO_*): any value, also the empty string, means 1 unless it parses as a number
        if (!e) g_val[i] = kDefs[i].dflt;
        else if (*e == 0) g_val[i] = 1;
        else g_val[i] = atoi(e);
    }
}

}  // namespace

int knob(Knob k) {
    std::call_once(g_once, init);
    return g_val[k];
}

int knob_set(const char* name, int value) {
    std::call_once(g_once, init);
    if (!name) return -1;
    for (int i = 0; i < K_COUNT; ++i)
        if (!strcmp(name, kDefs[i].name) || !strcmp(name, kDefs[i].name + 4)) { g_val[i] = value; return 0; }
    return -1;
}

}  // namespace ltk
