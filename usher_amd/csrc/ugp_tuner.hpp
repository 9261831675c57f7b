// ugp_tuner.hpp -- the per-handle choice "third pruning bound: build its tables for this batch or not" (ugp_capi.cpp uses it; no HIP
// in here: tests/test_tuner_cpu.py drives the decision logic on the CPU).
#pragma once
#include <stdint.h>

#include <algorithm>

namespace ugp {

// The third pruning bound is exact either way; whether its per-batch tables (ugp_bound3.hip: ~8 us per 512-sample tile on a
// 10M-node tree) cost less than they save the walk depends on the tree and on the queries (measured, round 5: +5 % on the
// headline batches, +67 % on config 5's ambiguous ones, -12 % on a SARS-CoV-2-shaped tree).  So the handle measures what counts --
// throughput: sub-batches of one class (by rows per sample) run in blocks of six in the same mode; the time between the
// completions (HIP events every call records anyway, read without waiting: tuner_poll, ugp_capi.cpp) of a block's last four sub-batches -- by then the device
// holds only that block's work, however many calls overlap -- gives the block's milliseconds per tile.  The first four blocks
// alternate with / without; after that the faster mode runs, and every 16th block (64th when the two differ by a quarter) is one of
// the other mode to keep its figure fresh.
// UGP_BOUND3=1 / 0 (or UGP_NO_BOUND3) pins the choice.  (First version: each sub-batch's own first-to-last-kernel time -- with
// three calls in flight that mostly measures the neighbours, and the choice flipped at random.)
// The default (round 6): no A/B inside the caller's steps.  Whether the tables pay is decided from two things known
// before the batch runs -- the size and shape of the tree (fixed at ugp_mat_create: `polytomy_tree` = more than 5 % of the nodes hang off a node
// with more than 16 children, the SARS-CoV-2 shape: short branches, hsub is small already, the walk's cost is sibling runs that no
// subtree test removes) and the batch's class by rows per sample (B3Tuner::class_of).  The measured cases (tools/b3_static_probe.sh,
// profiles/r06_b3_static_probe.txt) separate on exactly these; B3Tuner stays available as UGP_BOUND3=auto.
inline bool b3_static_choice(bool polytomy_tree, int row_class, uint64_t n_nodes) {
    // measured on one MI355X, 16 384 samples per call, M placements/s without / with (profiles/r06_b3_static_probe.txt):
    //   random-attachment 10 M nodes, plain queries     12.06 / 12.97        the same tree, ambiguous queries   3.69 / 6.94
    //   random-attachment  1 M nodes, plain queries     23.65 / 22.45        (the tables cost per block of the stream, the walk of a small tree is short)
    //   SARS-CoV-2 shape  10 M nodes, plain queries     11.53 / 10.59        the same tree, ambiguous queries   4.87 / 6.38
    //   SARS-CoV-2 shape  15 M x 10 000                  6.65 /  6.59        SARS-CoV-2 shape 1 M nodes        22.60 / 21.84
    if (row_class >= 2) return true;          // hundreds of rows per sample (runs of N, IUPAC cells): large bounds, weak first test -- the third decides
    if (polytomy_tree) return false;          // short branches: hsub is small already; what the walk costs there is sibling runs
    return n_nodes >= 3000000ull;             // plain batches: from a few million nodes on the walk saves more than the tables cost
}

struct B3Tuner {
    static constexpr int kClasses = 3;
    static constexpr uint32_t kBlock = 6, kSkip = 2;
    double ema[kClasses][2] = {};
    uint32_t n[kClasses][2] = {};
    uint32_t blocks[kClasses] = {};
    int cls = -1;              // the open block: class, mode, sub-batches issued
    bool mode = true, first = false;
    uint32_t issued = 0;
    uint64_t seq = 0;          // sub-batches issued in all
    uint64_t next_seq = 0;     // completion side: the next one to account for, the completion event of the one before it
    void *prev_done = nullptr;   // (a hipEvent_t)
    double acc = 0;
    uint32_t acc_n = 0;
    static int class_of(uint64_t rows, uint64_t samples) { const uint64_t r = samples ? rows / samples : 0; return r < 32 ? 0 : r < 256 ? 1 : 2; }
    // issue side: mode and position of the next sub-batch of class c
    bool next(int c, uint32_t *pos, uint64_t *sq) {
        if (c != cls || issued == kBlock) {
            cls = c; issued = 0;
            const uint32_t k = blocks[c]++;
            first = k == 0;
            if (k < 4) mode = (k & 1u) == 0;                // (with, without, with, without; the very first block -- allocations, cold caches -- is not counted)
            else if (!n[c][0] || !n[c][1]) mode = (k & 1u) == 0;   // (no figure for one of the modes yet -- still on the way, or the blocks of this class keep
                                                                   //  being interrupted by another class and never complete: keep alternating, never pin a mode unmeasured)
            else {
                const double a = ema[c][1], b = ema[c][0];
                const bool best = a <= b;
                const uint32_t probe = std::max(a, b) > 1.25 * std::min(a, b) ? 64u : 16u;   // (a clear case is looked at again less often)
                mode = (k % probe) == probe - 1 ? !best : best;
            }
        }
        *pos = issued++; *sq = seq++;
        return mode;
    }
    void record(int c, bool used, double ms_per_tile) {
        double &e = ema[c][used ? 1 : 0];
        uint32_t &k = n[c][used ? 1 : 0];
        e = k ? e + (ms_per_tile - e) * (k < 4 ? 1.0 / (k + 1) : 0.25) : ms_per_tile;   // (plain mean of the first four blocks, then a moving average)
        k++;
    }
    void resync(uint64_t past) { next_seq = std::max(next_seq, past); prev_done = nullptr; acc = 0; acc_n = 0; }
};

}  // namespace ugp
