// ugp_knobs.hpp -- the tuning switches of a handle, read from the environment ONCE (ugp_mat_create; the tuning hook
// ugp_mat_reload_knobs re-reads them) and never during a placement call: a call is free of getenv, so handles driven
// from different host threads do not race with a caller that changes its environment.  Results never depend on a knob
// (the GPU tests run every one against the oracle).  Experiments and diagnostics exist only in the build with
// -DUGP_EXPERIMENTS (libusher_amd_exp.so); the release library ignores their variables.
#pragma once
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <string>

namespace ugp {

struct Knobs {
    // switches
    bool force_v1 = false, no_sort = false, no_prune = false, coarse_phase2 = false, no_seed = false, no_descent = false, no_pad_fix = false,
         no_lpt = false, radix_sort = false, refill_all = false, scores_dfs = false, ties_dfs = false, debug_sharing = false, no_graph = false, ex_slow = false, no_bound3 = false, no_uniq = false, no_fork = true;   // (no_fork: the side stream of a lone call is opt-in, UGP_FORK=1 -- see ugp_capi.cpp)
    // -1 = the library's own choice
    int bound3 = -2,   // third pruning bound: -2 = decided from the tree and the batch (b3_static_choice), -1 = UGP_BOUND3=auto: the run-time tuner, 0 / 1 pinned
        tile_build = -1, nmask = -1, lds_bits = -1, light_order = -1, unit_grow = -1, split_cycles = -1, split_heavy = -1, split_dense = -1,
        descent_slack = 2;
    // 0 = the library's own choice
    uint32_t target_waves = 0, groups = 0, unit_chunks = 0, heavy_chunks = 0, unit_max = 0, shared_waves = 0, waves_per_cu = 0, ub_every = 0,
             heavy_prio = 0, descent_max = 0, scores_block = 0;
    uint32_t split_many_heavy = 2;   // the same minimum for the units of a tile's own region: dense and short (UGP_SPLIT_MANY_HEAVY; 0: as split_many)
    uint32_t split_many = 4;   // a unit that is cut hands out as many pieces as waves wait, of at least this many chunks each (UGP_SPLIT_MANY; 0: one half, as until round 4)
    uint32_t lbest_gib = 0;   // cap of the per-(chunk, sample) minima of one sub-batch in GiB (UGP_LBEST_GIB; 0: 24 -- 262,144 samples per launch sequence at 10M nodes)
    uint32_t depth = 3;   // calls of ugp_place_device_overlapped on the device at a time (2..4 workspace sets; UGP_PIPELINE_DEPTH)
    // experiments / diagnostics (always off in the release build)
    bool seed_prev = false, seed_check = false, phase2_packed = false, kbest_exclusive = false, stats = false;
    std::string trace;

    static bool flag(const char *name) { return getenv(name) != nullptr; }
    static int num(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
    static uint32_t pos(const char *name, int lo = 1) { const char *e = getenv(name); return e ? (uint32_t)(atoi(e) < lo ? lo : atoi(e)) : 0u; }

    static Knobs from_env() {
        Knobs k;
        k.force_v1 = flag("UGP_FORCE_V1"); k.no_sort = flag("UGP_NO_SORT"); k.no_prune = flag("UGP_NO_PRUNE");
        k.coarse_phase2 = flag("UGP_COARSE_PHASE2"); k.no_seed = flag("UGP_NO_SEED"); k.no_descent = flag("UGP_NO_DESCENT");
        k.no_pad_fix = flag("UGP_NO_PAD_FIX"); k.no_lpt = flag("UGP_NO_LPT"); k.refill_all = flag("UGP_REFILL_ALL"); k.radix_sort = flag("UGP_RADIX_SORT");
        k.scores_dfs = flag("UGP_SCORES_DFS"); k.ties_dfs = flag("UGP_TIES_DFS"); k.debug_sharing = flag("UGP_DEBUG_SHARING");
        k.no_graph = flag("UGP_NO_GRAPH"); k.ex_slow = flag("UGP_EX_SLOW"); k.no_bound3 = flag("UGP_NO_BOUND3"); k.no_uniq = flag("UGP_NO_UNIQ"); k.no_fork = !flag("UGP_FORK") || flag("UGP_NO_FORK");
        if (const char *e = getenv("UGP_BOUND3")) k.bound3 = (e[0] == 'a' || e[0] == 'A') ? -1 : (atoi(e) != 0 ? 1 : 0);
        k.tile_build = num("UGP_TILE_BUILD", -1); k.nmask = num("UGP_NMASK", -1); k.lds_bits = num("UGP_LDS_BITS", -1);
        k.light_order = num("UGP_LIGHT_ORDER", -1);
        if (getenv("UGP_UNIT_GROW")) k.unit_grow = num("UGP_UNIT_GROW", 0) < 0 ? 0 : num("UGP_UNIT_GROW", 0);
        if (getenv("UGP_SPLIT_CYCLES")) k.split_cycles = k.split_heavy = num("UGP_SPLIT_CYCLES", 0) < 0 ? 0 : num("UGP_SPLIT_CYCLES", 0);
        if (getenv("UGP_SPLIT_HEAVY")) k.split_heavy = num("UGP_SPLIT_HEAVY", 0) < 0 ? 0 : num("UGP_SPLIT_HEAVY", 0);
        if (getenv("UGP_SPLIT_DENSE")) k.split_dense = num("UGP_SPLIT_DENSE", 0) < 0 ? 0 : num("UGP_SPLIT_DENSE", 0);
        k.descent_slack = num("UGP_DESCENT_SLACK", 2);
        k.target_waves = pos("UGP_TARGET_WAVES"); k.groups = pos("UGP_GROUPS"); k.unit_chunks = pos("UGP_UNIT_CHUNKS");
        k.heavy_chunks = pos("UGP_HEAVY_CHUNKS"); k.unit_max = pos("UGP_UNIT_MAX"); k.shared_waves = pos("UGP_SHARED_WAVES");
        k.waves_per_cu = pos("UGP_WAVES_PER_CU"); k.ub_every = pos("UGP_UB_EVERY"); k.heavy_prio = pos("UGP_HEAVY_PRIO", 0);
        k.descent_max = pos("UGP_DESCENT_MAX");
        k.lbest_gib = pos("UGP_LBEST_GIB");
        if (getenv("UGP_SPLIT_MANY")) k.split_many = pos("UGP_SPLIT_MANY", 0);
        if (getenv("UGP_SPLIT_MANY_HEAVY")) k.split_many_heavy = pos("UGP_SPLIT_MANY_HEAVY", 0);
        if (getenv("UGP_PIPELINE_DEPTH")) k.depth = pos("UGP_PIPELINE_DEPTH", 2);
        if (const char *e = getenv("UGP_SCORES_BLOCK")) { int v = atoi(e) / 64 * 64; k.scores_block = (uint32_t)(v < 64 ? 64 : v > 1024 ? 1024 : v); }
#ifdef UGP_EXPERIMENTS
        k.seed_prev = flag("UGP_SEED_PREV"); k.seed_check = flag("UGP_SEED_CHECK"); k.phase2_packed = flag("UGP_PHASE2_PACKED");
        k.kbest_exclusive = flag("UGP_KBEST_EXCLUSIVE"); k.stats = flag("UGP_STATS");
        if (const char *e = getenv("UGP_TRACE")) k.trace = e;
#endif
        return k;
    }
};

}  // namespace ugp
