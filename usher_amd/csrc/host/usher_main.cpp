// usher_main.cpp -- `usher-amd`: the usher-compatible CLI bound to the GPU
// placement library (libusher_amd.so) through its C ABI.
//
// Multi-GPU (`--devices 0-7`): the reference places samples one after another on one host
// (usher_common.cpp:310); on a static tree they are independent, so the backend shards every batch of
// samples into contiguous blocks, one per device, each placed by its own host thread on that device's
// replica of the flattened tree (ugp_mat_create_multi: flattened once, uploaded n times); results land in
// the caller's buffers at the shard's offset -- the gather is a host-memory write, no collective needed
// inside one process.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "driver.hpp"
#include "usher_amd.h"

namespace uh { int usher_main(int argc, char **argv, const Backend &be); extern bool g_leak_tree_at_exit; }

namespace {

struct GpuCtx {
    std::vector<int> devices{0};
    std::vector<ugp_mat *> mats;
    ugp_fitch *fitch = nullptr;
    bool fitch_pool = false;   // ugp_fitch_sankoff has run: its pooled device buffers are handed back before the first tree goes to the device
    uint64_t version = 0;
    std::string err;
    std::thread warm_th;   // Backend::warm: the device runtime comes up while the inputs are read
};

void drop_mats(GpuCtx *c) {
    for (ugp_mat *m : c->mats) ugp_mat_destroy(m);
    c->mats.clear();
}

int ensure(GpuCtx *c, const ugp_tree_desc *t, uint64_t version) {
    if (c->warm_th.joinable()) c->warm_th.join();
    if (!c->mats.empty() && c->version == version) return UGP_OK;
    drop_mats(c);
    if (c->fitch_pool) { ugp_fitch_release(c->devices[0]); c->fitch_pool = false; }   // (the MAT is built: up to 4 GiB of row storage back to the placement handles)
    c->mats.assign(c->devices.size(), nullptr);
    int rc = ugp_mat_create_multi(t, c->devices.data(), (int)c->devices.size(), c->mats.data());
    if (rc != UGP_OK) { c->err = ugp_last_error(); c->mats.clear(); return rc; }
    c->version = version;
    return UGP_OK;
}

// A contiguous block [lo, hi) of a query batch as a batch of its own.
struct Shard {
    std::vector<uint64_t> ent_off;
    ugp_queries q{};
    uint64_t lo = 0, hi = 0;
    Shard(const ugp_queries *all, uint64_t lo_, uint64_t hi_) : lo(lo_), hi(hi_) {
        const uint64_t e0 = all->ent_off[lo];
        ent_off.resize(hi - lo + 1);
        for (uint64_t i = lo; i <= hi; i++) ent_off[i - lo] = all->ent_off[i] - e0;
        q.n_queries = hi - lo; q.ent_off = ent_off.data();
        q.pos = all->pos + e0; q.ref = all->ref + e0; q.nuc = all->nuc + e0; q.is_missing = all->is_missing + e0;
    }
};

// Run fn(device slot, shard) on one host thread per device over contiguous sample blocks whose sizes differ
// by at most one.  Small batches use fewer devices (a 512-sample tile is the unit of work of the kernels).
template <typename F>
int sharded(GpuCtx *c, const ugp_queries *q, F fn) {
    const uint64_t Q = q->n_queries;
    uint64_t min_block = 512;
    if (const char *e = getenv("USHER_AMD_SHARD_MIN")) min_block = (uint64_t)std::max(1, atoi(e));   // tests lower it
    const size_t n = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(c->mats.size(), (Q + min_block - 1) / min_block));
    if (n == 1) {
        int rc = fn(0, *q, 0);
        if (rc != UGP_OK) c->err = ugp_last_error();
        return rc;
    }
    std::vector<int> rcs(n, UGP_OK);
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (size_t d = 0; d < n; d++)
        th.emplace_back([&, d]() {
            const uint64_t base = Q / n, extra = Q % n;
            const uint64_t lo = d * base + std::min<uint64_t>(d, extra), hi = lo + base + (d < extra ? 1 : 0);
            Shard sh(q, lo, hi);
            rcs[d] = fn(d, sh.q, lo);
            if (rcs[d] != UGP_OK) errs[d] = ugp_last_error();   // (the message is per thread)
        });
    for (auto &t : th) t.join();
    for (size_t d = 0; d < n; d++)
        if (rcs[d] != UGP_OK) { c->err = "device " + std::to_string(c->devices[d]) + ": " + errs[d]; return rcs[d]; }
    return UGP_OK;
}

int gpu_place(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, ugp_result *out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return sharded(c, q, [&](size_t d, const ugp_queries &part, uint64_t lo) { return ugp_place_batch(c->mats[d], &part, out + lo); });
}
int gpu_scores(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, int32_t *out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return sharded(c, q, [&](size_t d, const ugp_queries &part, uint64_t lo) { return ugp_scores_per_node(c->mats[d], &part, out + lo * t->n_nodes); });
}
int gpu_ties(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, uint32_t cap, uint32_t *tj, uint8_t *th, uint32_t *tc) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return sharded(c, q, [&](size_t d, const ugp_queries &part, uint64_t lo) {
        return ugp_tied_nodes(c->mats[d], &part, cap, tj + lo * cap, th + lo * cap, tc + lo);
    });
}
// Add mode: the edits go to every replica (each keeps excluding the rewritten nodes from its searches); the records are scored on
// the first device.
int gpu_update(void *ctx, const ugp_touched *recs, const uint32_t *retired, uint64_t n_retired, uint32_t *first_id) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (c->mats.empty()) { c->err = "no tree on the device"; return UGP_ERR_INVALID; }
    for (size_t d = 0; d < c->mats.size(); d++) {
        uint32_t f = 0;
        if (int rc = ugp_mat_update(c->mats[d], recs, retired, n_retired, &f)) { c->err = ugp_last_error(); return rc; }
        if (d == 0 && first_id) *first_id = f;
    }
    return UGP_OK;
}
#define GPU_FWD(call) GpuCtx *c = (GpuCtx *)ctx; if (c->mats.empty()) { c->err = "no tree on the device"; return UGP_ERR_INVALID; } \
                      const int rc = call; if (rc != UGP_OK) c->err = ugp_last_error(); return rc;
int gpu_touched_open(void *ctx, const ugp_queries *q) { GPU_FWD(ugp_touched_open(c->mats[0], q)) }
int gpu_touched_score(void *ctx, uint32_t first_id, uint64_t first_sample) { GPU_FWD(ugp_touched_score(c->mats[0], first_id, first_sample)) }
int gpu_touched_rescore(void *ctx, uint64_t sample) { GPU_FWD(ugp_touched_rescore(c->mats[0], sample)) }
int gpu_touched_fetch(void *ctx, uint64_t first_sample, uint64_t n, uint32_t cap, int32_t *best, uint32_t *count, uint32_t *ids, uint8_t *hu) {
    GPU_FWD(ugp_touched_fetch(c->mats[0], first_sample, n, cap, best, count, ids, hu))
}
const char *gpu_err(void *ctx) { return ((GpuCtx *)ctx)->err.c_str(); }
void gpu_warm(void *ctx) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (c->warm_th.joinable()) return;
    const std::vector<int> devs = c->devices;
    c->warm_th = std::thread([devs]() { for (int d : devs) (void)ugp_device_warmup(d); });
}
int gpu_prepare(void *ctx, const ugp_tree_desc *t, uint64_t v) { return ensure((GpuCtx *)ctx, t, v); }
int gpu_fitch(void *ctx, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, uint64_t *n_out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (c->fitch) { ugp_fitch_destroy(c->fitch); c->fitch = nullptr; }
    c->fitch_pool = true;
    if (int rc = ugp_fitch_sankoff(c->devices[0], n_nodes, parent, sites, &c->fitch)) { c->err = ugp_last_error(); return rc; }
    *n_out = ugp_fitch_count(c->fitch);
    return UGP_OK;
}
int gpu_fitch_get(void *ctx, uint32_t *site, uint32_t *node, uint8_t *par_nuc, uint8_t *mut_nuc) {
    GpuCtx *c = (GpuCtx *)ctx;
    int rc = ugp_fitch_get(c->fitch, site, node, par_nuc, mut_nuc);
    if (rc != UGP_OK) c->err = ugp_last_error();
    ugp_fitch_destroy(c->fitch);
    c->fitch = nullptr;
    return rc;
}

// "0-7", "0,2,3", "1": device ordinals (a device named twice gets two replicas).  Empty result = malformed.
std::vector<int> parse_devices(const std::string &s) {
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string part = s.substr(i, j - i);
        const size_t dash = part.find('-');
        char *end = nullptr;
        if (part.empty()) return {};
        long a = strtol(part.c_str(), &end, 10), b = a;
        if (dash != std::string::npos) {
            if (end != part.c_str() + dash) return {};
            b = strtol(part.c_str() + dash + 1, &end, 10);
        }
        if (*end != 0 || a < 0 || b < a || b > 1023) return {};
        for (long d = a; d <= b; d++) out.push_back((int)d);
        i = j + 1;
    }
    return out;
}

}  // namespace

int main(int argc, char **argv) {
    GpuCtx ctx;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        std::string val;
        bool is_list = false, hit = false;
        if (a == "--device" || a == "--devices") { if (i + 1 < argc) { val = argv[i + 1]; hit = true; is_list = a == "--devices"; } }
        else if (a.rfind("--devices=", 0) == 0) { val = a.substr(10); hit = true; is_list = true; }
        else if (a.rfind("--device=", 0) == 0) { val = a.substr(9); hit = true; }
        if (!hit) continue;
        ctx.devices = is_list ? parse_devices(val) : std::vector<int>{atoi(val.c_str())};
        if (ctx.devices.empty()) { fprintf(stderr, "ERROR: cannot parse the device list '%s' (expected e.g. 0-7 or 0,2,3)\n", val.c_str()); return 1; }
    }
    uh::Backend be;
    be.ctx = &ctx; be.place = gpu_place; be.scores = gpu_scores; be.ties = gpu_ties; be.last_error = gpu_err;
    be.fitch = gpu_fitch; be.fitch_get = gpu_fitch_get;
    be.update = gpu_update; be.touched_open = gpu_touched_open; be.touched_score = gpu_touched_score; be.touched_rescore = gpu_touched_rescore;
    be.touched_fetch = gpu_touched_fetch;
    be.warm = gpu_warm; be.prepare = gpu_prepare;
    uh::g_leak_tree_at_exit = true;   // (this process ends with the run: the tree is not taken apart node by node first)
    int rc = uh::usher_main(argc, argv, be);
    if (ctx.warm_th.joinable()) ctx.warm_th.join();
    // (the handles' device memory goes with the process: freeing 3 GB of tables buffer by buffer is 0.1-0.2 s of nothing; the
    // tests and tools that embed the library destroy their handles)
    if (getenv("USHER_AMD_TIDY_EXIT")) { drop_mats(&ctx); if (ctx.fitch) ugp_fitch_destroy(ctx.fitch); }
    fflush(nullptr);
    return rc;
}
