// tuning.cpp -- the process-wide tuning knobs of kernels.h: defaults, environment overrides (read once,
// before the first use) and run-time changes through j2k_hip_debug_tune() (include/j2k_hip.h).
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kernels.h"

namespace j2k_hip {
namespace {

struct Knob { const char *key; const char *env; int Tuning::*field; };
const Knob kKnobs[] = {
    {"overlap", nullptr, &Tuning::overlap},          // env: J2K_NO_OVERLAP (inverted, below)
    {"no_fuse", "J2K_NO_FUSE", &Tuning::no_fuse},
    {"level_events", "J2K_DWT_LEVEL_EVENTS", &Tuning::level_events},
    {"mq_prio", "J2K_MQ_PRIO", &Tuning::mq_prio},
    {"level1_dispatch_events", "J2K_LEVEL1_DISPATCH_EVENTS", &Tuning::level1_dispatch_events},
    {"groups", "J2K_GROUPS", &Tuning::groups},
    {"heavy_min", "J2K_MQ_HEAVY", &Tuning::heavy_min},
    {"mq_wait_us", "J2K_MQ_WAIT_US", &Tuning::mq_wait_us},
    {"mq_single", "J2K_MQ_SINGLE", &Tuning::mq_single},
    {"coder_cus", "J2K_CODER_CUS", &Tuning::coder_cus},
    {"dwt_pairs", "J2K_DWT_PAIRS", &Tuning::dwt_pairs},
    {"mq_yield", "J2K_MQ_YIELD", &Tuning::mq_yield},
    {"dwt_ahead", "J2K_DWT_AHEAD", &Tuning::dwt_ahead},
    {"dense_chain", "J2K_DENSE_CHAIN", &Tuning::dense_chain},
    {"alloc_threads", "J2K_ALLOC_THREADS", &Tuning::alloc_threads},
    {"rate_dev", "J2K_RATE_DEV", &Tuning::rate_dev},
    {"rate_dev_scan", "J2K_RATE_DEV_SCAN", &Tuning::rate_dev_scan},
    {"dwt_depth", "J2K_DWT_DEPTH", &Tuning::dwt_depth},
    {"dwt_ppc", "J2K_DWT_PPC", &Tuning::dwt_ppc},
    {"dwt_min_waves", "J2K_DWT_MIN_WAVES", &Tuning::dwt_min_waves},
    {"fused_wpb", "J2K_DWT_FUSED_WPB", &Tuning::fused_wpb},
    {"fused_ppc", "J2K_DWT_FUSED_PPC", &Tuning::fused_ppc},
    {"fused_generic", "J2K_DWT_FUSED_GENERIC", &Tuning::fused_generic},
    {"dwt_xcd", "J2K_DWT_XCD", &Tuning::dwt_xcd},
    {"dwt_nt", "J2K_DWT_NT", &Tuning::dwt_nt},
    {"dwt_ntl", "J2K_DWT_NTL", &Tuning::dwt_ntl},
    {"t1dec_lanes", "J2K_T1DEC_LANES", &Tuning::t1dec_lanes},
    {"t1dec_tail", "J2K_T1DEC_TAIL", &Tuning::t1dec_tail},
    {"bands", "J2K_BANDS", &Tuning::bands},
    {"staging", "J2K_STAGING", &Tuning::staging},
    {"stage_kb", "J2K_STAGE_KB", &Tuning::stage_kb},
};

Tuning g_tuning;
std::once_flag g_once;

void from_env()
{
    for (const Knob &k : kKnobs) {
        if (!k.env) continue;
        const char *v = std::getenv(k.env);
        if (v && *v) g_tuning.*(k.field) = std::atoi(v);
        else if (v && (k.field == &Tuning::no_fuse || k.field == &Tuning::level_events || k.field == &Tuning::mq_single))
            g_tuning.*(k.field) = 1; // historic use: the bare presence of the variable switches it on
    }
    if (std::getenv("J2K_NO_OVERLAP")) g_tuning.overlap = 0;
}

} // namespace

Tuning &tuning()
{
    std::call_once(g_once, from_env);
    return g_tuning;
}

int tune(const char *key, int value)
{
    if (!key) return 1;
    Tuning &t = tuning();
    for (const Knob &k : kKnobs)
        if (std::strcmp(k.key, key) == 0) { t.*(k.field) = value; return 0; }
    return 1;
}

int get_tune(const char *key, int *value)
{
    if (!key || !value) return 1;
    Tuning &t = tuning();
    for (const Knob &k : kKnobs)
        if (std::strcmp(k.key, key) == 0) { *value = t.*(k.field); return 0; }
    return 1;
}

} // namespace j2k_hip
