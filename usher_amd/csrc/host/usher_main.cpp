// usher_main.cpp -- `usher-amd`: the usher-compatible CLI bound to the GPU
// placement library (libusher_amd.so) through its C ABI.
#include <cstdio>
#include <string>

#include "driver.hpp"
#include "usher_amd.h"

namespace uh { int usher_main(int argc, char **argv, const Backend &be); }

namespace {

struct GpuCtx {
    ugp_mat *mat = nullptr;
    ugp_fitch *fitch = nullptr;
    uint64_t version = 0;
    int device = 0;
};

int ensure(GpuCtx *c, const ugp_tree_desc *t, uint64_t version) {
    if (c->mat && c->version == version) return UGP_OK;
    if (c->mat) { ugp_mat_destroy(c->mat); c->mat = nullptr; }
    int rc = ugp_mat_create(t, c->device, &c->mat);
    if (rc == UGP_OK) c->version = version;
    return rc;
}

int gpu_place(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, ugp_result *out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return ugp_place_batch(c->mat, q, out);
}
int gpu_scores(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, int32_t *out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return ugp_scores_per_node(c->mat, q, out);
}
int gpu_ties(void *ctx, const ugp_tree_desc *t, uint64_t v, const ugp_queries *q, uint32_t cap, uint32_t *tj, uint8_t *th, uint32_t *tc) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (int rc = ensure(c, t, v)) return rc;
    return ugp_tied_nodes(c->mat, q, cap, tj, th, tc);
}
const char *gpu_err(void *) { return ugp_last_error(); }
int gpu_fitch(void *ctx, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, uint64_t *n_out) {
    GpuCtx *c = (GpuCtx *)ctx;
    if (c->fitch) { ugp_fitch_destroy(c->fitch); c->fitch = nullptr; }
    if (int rc = ugp_fitch_sankoff(c->device, n_nodes, parent, sites, &c->fitch)) return rc;
    *n_out = ugp_fitch_count(c->fitch);
    return UGP_OK;
}
int gpu_fitch_get(void *ctx, uint32_t *site, uint32_t *node, uint8_t *par_nuc, uint8_t *mut_nuc) {
    GpuCtx *c = (GpuCtx *)ctx;
    int rc = ugp_fitch_get(c->fitch, site, node, par_nuc, mut_nuc);
    ugp_fitch_destroy(c->fitch);
    c->fitch = nullptr;
    return rc;
}

}  // namespace

int main(int argc, char **argv) {
    GpuCtx ctx;
    for (int i = 1; i + 1 < argc; i++) if (std::string(argv[i]) == "--device") ctx.device = atoi(argv[i + 1]);
    uh::Backend be;
    be.ctx = &ctx; be.place = gpu_place; be.scores = gpu_scores; be.ties = gpu_ties; be.last_error = gpu_err;
    be.fitch = gpu_fitch; be.fitch_get = gpu_fitch_get;
    int rc = uh::usher_main(argc, argv, be);
    if (ctx.mat) ugp_mat_destroy(ctx.mat);
    if (ctx.fitch) ugp_fitch_destroy(ctx.fitch);
    return rc;
}
