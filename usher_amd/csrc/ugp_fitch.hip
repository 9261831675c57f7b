// ugp_fitch.hip -- Fitch-Sankoff small-parsimony assignment of a batch of VCF sites
// onto a fixed tree, for gfx950.  Replaces mapper_body::operator()
// (src/usher_mapper.cpp:6-161) as driven by the VCF reader
// (src/mutation_annotated_tree.cpp:2108-2179).
//
// Formulation.  With unit substitution costs the Sankoff recurrence of :86-111
//     sc[p][b] += min_k( sc[c][k] + [k != b] )  =  m_c + [b not in F_c],
// m_c = min_k sc[c][k], F_c = argmin set of child c.  So the forward pass only needs the
// 4-bit set F per (node, site):  F_p = argmin over the bases allowed at p (all four; the
// genotype mask when the VCF has a column for p; {REF} for a leaf without a variant -- the
// "large value" num_nodes of :35-62 always exceeds the number of children) of
// cnt_p[b] = #{children c : b not in F_c}.  The backward pass (:114-157) keeps the parent's
// state when it is in F_n and otherwise takes the lowest base of F_n (the first strictly
// smaller score in the j = 0..3 scan).
//
// Layout.  F[node][W] u32, eight sites per word (one nibble each), nodes in breadth-first
// order so the children of a node are W-word rows that lie back to back.  One wave owns
// one node x 64 words (512 sites); lanes read consecutive words, so every child row is one
// 256 B coalesced load.  Child counts are kept bit-sliced (plane k = bit k of all 32
// (site, base) counters of the lane), so a node with c children costs ~3*log2(c) VALU ops
// per child and the argmin is a bit-sliced tournament.  Levels are processed bottom-up then
// top-down, one launch per level; states overwrite F in place and a final pass lists the
// (site, node) pairs whose state differs from the parent's.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "usher_amd.h"

namespace ugp {
int set_error(int code, const std::string &msg);   // ugp_capi.cpp
}

namespace {

constexpr int FS_PLANES = 32;   // counters of up to 2^32 - 1 children

__device__ __forceinline__ uint32_t nib_any(uint32_t x) {   // 0xF in every nibble of x that is non-zero
    uint32_t t = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
    return t * 15u;
}

__device__ __forceinline__ uint32_t nib_lowbit(uint32_t f) {   // lowest set bit of every nibble
    const uint32_t up = ((f << 1) & 0xEEEEEEEEu) | ((f << 2) & 0xCCCCCCCCu) | ((f << 3) & 0x88888888u);
    return f & ~up;
}

// leaves start as {REF}, internal nodes as "any base"
__global__ void k_fs_init(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw, const uint32_t *__restrict__ n_children,
                          uint64_t n_nodes, uint32_t W) {
    const uint64_t total = n_nodes * W;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t n = i / W;
        const uint32_t w = (uint32_t)(i - n * W);
        F[i] = n_children[n] ? 0xFFFFFFFFu : refw[w];
    }
}

// genotype cells of tree nodes: replace the initial nibble by the allele mask (:47-62)
__global__ void k_fs_scatter(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw, const uint32_t *__restrict__ n_children,
                             const uint32_t *__restrict__ v_site, const uint32_t *__restrict__ v_node,
                             const uint8_t *__restrict__ v_nuc, uint64_t n_var, uint32_t W) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_var) return;
    const uint32_t s = v_site[i], n = v_node[i];
    const uint32_t w = s >> 3, sh = (s & 7) * 4;
    const uint32_t old = n_children[n] ? 0xFu : ((refw[w] >> sh) & 0xFu);
    const uint32_t x = (old ^ (v_nuc[i] & 0xFu)) << sh;
    if (x) atomicXor(&F[(uint64_t)n * W + w], x);
}

// forward pass for the internal nodes of one level (:86-111)
__global__ __launch_bounds__(256) void k_fs_forward(uint32_t *__restrict__ F, const uint32_t *__restrict__ nodes, uint32_t n_level,
                                                    const uint32_t *__restrict__ first_child,
                                                    const uint32_t *__restrict__ n_children, uint32_t W) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t idx = blockIdx.x * 4 + wave;
    const uint32_t w = blockIdx.y * 64 + lane;
    if (idx >= n_level || w >= W) return;
    const uint32_t p = __builtin_amdgcn_readfirstlane(nodes[idx]);
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(first_child[p]);
    const uint32_t nc = __builtin_amdgcn_readfirstlane(n_children[p]);
    const int K = 32 - __builtin_clz(nc);   // planes needed to count to nc
    uint32_t plane[FS_PLANES];
#pragma unroll
    for (int k = 0; k < FS_PLANES; k++) plane[k] = 0;
    const uint32_t *row = F + (uint64_t)c0 * W + w;
    uint32_t nxt = *row;
    for (uint32_t c = 0; c < nc; c++) {
        const uint32_t x = nxt;
        if (c + 1 < nc) nxt = row[(uint64_t)(c + 1) * W];
        uint32_t carry = ~x;   // +1 for every (site, base) with base not in F_c
#pragma unroll
        for (int k = 0; k < FS_PLANES; k++) {
            if (k >= K) break;
            const uint32_t t = plane[k] & carry;
            plane[k] ^= carry;
            carry = t;
        }
    }
    uint32_t cand = F[(uint64_t)p * W + w];   // allowed bases (all four, or the node's own genotype mask)
#pragma unroll
    for (int k = FS_PLANES - 1; k >= 0; k--) {
        if (k >= K) continue;
        const uint32_t z = cand & ~plane[k];   // candidates whose counter has a 0 here
        const uint32_t m = nib_any(z);
        cand = (z & m) | (cand & ~m);
    }
    F[(uint64_t)p * W + w] = cand;
}

// backward pass for all nodes of one level (:114-141); states replace F in place.
__global__ __launch_bounds__(256) void k_fs_backward(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw,
                                                     const uint32_t *__restrict__ parent, uint32_t lvl_begin, uint32_t lvl_end,
                                                     uint32_t W, unsigned long long *__restrict__ n_mut) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n = lvl_begin + blockIdx.x * 4 + wave;
    const uint32_t w = blockIdx.y * 64 + lane;
    uint32_t changed = 0;
    if (n < lvl_end && w < W) {
        const uint32_t par = __builtin_amdgcn_readfirstlane(parent[n]);
        const uint32_t sp = par == 0xFFFFFFFFu ? refw[w] : F[(uint64_t)par * W + w];
        const uint32_t f = F[(uint64_t)n * W + w];
        const uint32_t keep = nib_any(f & sp);
        const uint32_t s = (sp & keep) | (nib_lowbit(f) & ~keep);
        F[(uint64_t)n * W + w] = s;
        changed = __builtin_popcount(nib_any(s ^ sp) & 0x11111111u);
    }
    // one atomic per wave
    for (int o = 32; o; o >>= 1) changed += __shfl_down(changed, o, 64);
    if (lane == 0 && changed) atomicAdd(n_mut, (unsigned long long)changed);
}

// list the (site, node) pairs whose state differs from the parent's (:143-156)
__global__ __launch_bounds__(256) void k_fs_emit(const uint32_t *__restrict__ F, const uint32_t *__restrict__ refw,
                                                 const uint32_t *__restrict__ parent, uint32_t n_nodes, uint32_t W, uint32_t site_base,
                                                 unsigned long long *__restrict__ cursor, uint64_t *__restrict__ out_key,
                                                 uint8_t *__restrict__ out_val) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + wave;
    const uint32_t w = blockIdx.y * 64 + lane;
    if (n >= n_nodes || w >= W) return;
    const uint32_t par = __builtin_amdgcn_readfirstlane(parent[n]);
    const uint32_t sp = par == 0xFFFFFFFFu ? refw[w] : F[(uint64_t)par * W + w];
    const uint32_t s = F[(uint64_t)n * W + w];
    uint32_t d = nib_any(s ^ sp) & 0x11111111u;
    if (!d) return;
    const unsigned long long at = atomicAdd(cursor, (unsigned long long)__builtin_popcount(d));
    uint32_t k = 0;
    while (d) {
        const uint32_t sh = __builtin_ctz(d);
        d &= d - 1;
        const uint32_t site = site_base + w * 8 + (sh >> 2);
        out_key[at + k] = ((uint64_t)site << 32) | n;
        out_val[at + k] = (uint8_t)((((sp >> sh) & 0xFu) << 4) | ((s >> sh) & 0xFu));
        k++;
    }
}

template <typename T>
struct Dev {
    T *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
    }
    hipError_t upload(const T *src, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice);
    }
};

}  // namespace

struct ugp_fitch {
    std::vector<uint32_t> site, node;
    std::vector<uint8_t> par, nuc;
};

#define FS_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return ugp::set_error(UGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int ugp_fitch_sankoff(int device, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, ugp_fitch **out) {
    if (!out) return ugp::set_error(UGP_ERR_INVALID, "null output handle");
    *out = nullptr;
    if (!parent || !sites || n_nodes == 0) return ugp::set_error(UGP_ERR_INVALID, "null or empty tree / site arrays");
    if (n_nodes >= 0xFFFFFFFFull) return ugp::set_error(UGP_ERR_UNSUPPORTED, "more than 2^32-2 nodes");
    const uint64_t S = sites->n_sites;
    if (S >= (1ull << 31)) return ugp::set_error(UGP_ERR_UNSUPPORTED, "more than 2^31 sites");
    if (S && (!sites->ref || !sites->var_off)) return ugp::set_error(UGP_ERR_INVALID, "null site arrays");
    const uint64_t n_var = S ? sites->var_off[S] : 0;
    if (n_var && (!sites->var_node || !sites->var_nuc)) return ugp::set_error(UGP_ERR_INVALID, "null variant arrays");
    // topology: breadth-first order means parent[] is non-decreasing and children are contiguous
    if (parent[0] != 0xFFFFFFFFu) return ugp::set_error(UGP_ERR_INVALID, "parent[0] must be the root (UINT32_MAX)");
    const uint32_t N = (uint32_t)n_nodes;
    std::vector<uint32_t> first_child(N, 0), n_children(N, 0), level(N, 0);
    for (uint32_t j = 1; j < N; j++) {
        const uint32_t p = parent[j];
        if (p >= j || (j > 1 && p < parent[j - 1])) return ugp::set_error(UGP_ERR_INVALID, "tree arrays are not in breadth-first order");
        if (n_children[p]++ == 0) first_child[p] = j;
        level[j] = level[p] + 1;
    }
    std::vector<uint32_t> lvl_off;   // nodes of level L are [lvl_off[L], lvl_off[L+1])
    for (uint32_t j = 0; j < N; j++)
        if (j == 0 || level[j] != level[j - 1]) lvl_off.push_back(j);
    lvl_off.push_back(N);
    const uint32_t n_levels = (uint32_t)lvl_off.size() - 1;
    std::vector<uint32_t> inodes, ilvl_off(n_levels + 1, 0);   // internal nodes, grouped by level
    for (uint32_t L = 0; L < n_levels; L++) {
        ilvl_off[L] = (uint32_t)inodes.size();
        for (uint32_t j = lvl_off[L]; j < lvl_off[L + 1]; j++)
            if (n_children[j]) inodes.push_back(j);
    }
    ilvl_off[n_levels] = (uint32_t)inodes.size();
    for (uint64_t s = 0; s < S; s++) {
        const uint8_t r = sites->ref[s];
        if (r != 1 && r != 2 && r != 4 && r != 8) return ugp::set_error(UGP_ERR_INVALID, "site reference allele is not one of A,C,G,T");
        if (sites->var_off[s + 1] < sites->var_off[s]) return ugp::set_error(UGP_ERR_INVALID, "var_off is not monotone");
    }
    for (uint64_t v = 0; v < n_var; v++) {
        if (sites->var_node[v] >= N) return ugp::set_error(UGP_ERR_INVALID, "variant node index out of range");
        if ((sites->var_nuc[v] & 0xF) == 0 || sites->var_nuc[v] > 15) return ugp::set_error(UGP_ERR_INVALID, "variant allele mask must be 1..15");
    }

    ugp_fitch *res = new (std::nothrow) ugp_fitch();
    if (!res) return ugp::set_error(UGP_ERR_NOMEM, "out of host memory");
    struct Guard { ugp_fitch *r; ~Guard() { delete r; } } guard{res};
    if (S == 0) { guard.r = nullptr; *out = res; return UGP_OK; }

    FS_TRY(hipSetDevice(device));
    hipStream_t stream = nullptr;
    Dev<uint32_t> d_parent, d_first, d_nchild, d_inodes;
    FS_TRY(d_parent.upload(parent, N));
    FS_TRY(d_first.upload(first_child.data(), N));
    FS_TRY(d_nchild.upload(n_children.data(), N));
    FS_TRY(d_inodes.upload(inodes.data(), inodes.size()));

    // sites per pass: F takes N * W * 4 bytes; use up to half of the free HBM (UGP_FITCH_BYTES overrides)
    size_t free_b = 0, total_b = 0;
    FS_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = free_b / 2;
    if (const char *e = getenv("UGP_FITCH_BYTES")) budget = strtoull(e, nullptr, 10);
    uint64_t W_max = std::max<uint64_t>(budget / ((uint64_t)N * 4), 1);
    if (W_max >= 64) W_max &= ~63ull;   // whole 512-site wave rows
    const uint64_t W_all = (S + 7) / 8;
    const uint32_t W_pass = (uint32_t)std::min<uint64_t>(W_max, W_all);
    Dev<uint32_t> d_F, d_refw, d_vsite, d_vnode;
    Dev<uint8_t> d_vnuc, d_oval, d_oval2;
    Dev<uint64_t> d_okey, d_okey2;
    Dev<unsigned long long> d_cnt;
    Dev<uint8_t> d_tmp;
    FS_TRY(d_F.alloc((size_t)N * W_pass));
    FS_TRY(d_refw.alloc(W_pass));
    FS_TRY(d_cnt.alloc(2));
    std::vector<uint32_t> refw(W_pass), vsite, vnode;
    std::vector<uint8_t> vnuc;
    std::vector<uint64_t> h_key;
    std::vector<uint8_t> h_val;

    for (uint64_t w0 = 0; w0 < W_all; w0 += W_pass) {
        const uint32_t W = (uint32_t)std::min<uint64_t>(W_pass, W_all - w0);
        const uint64_t s0 = w0 * 8, s1 = std::min<uint64_t>(S, s0 + (uint64_t)W * 8);
        std::fill(refw.begin(), refw.end(), 0x11111111u);
        for (uint64_t s = s0; s < s1; s++) {
            const uint32_t sh = (uint32_t)((s - s0) & 7) * 4;
            uint32_t &x = refw[(s - s0) >> 3];
            x = (x & ~(0xFu << sh)) | ((uint32_t)sites->ref[s] << sh);
        }
        // genotype cells of this pass; a node named twice at one site keeps the last cell (:47-62 runs in order)
        vsite.clear(); vnode.clear(); vnuc.clear();
        std::unordered_map<uint32_t, size_t> seen;
        for (uint64_t s = s0; s < s1; s++) {
            seen.clear();
            for (uint64_t v = sites->var_off[s]; v < sites->var_off[s + 1]; v++) {
                auto it = seen.find(sites->var_node[v]);
                if (it != seen.end()) { vnuc[it->second] = sites->var_nuc[v]; continue; }
                seen.emplace(sites->var_node[v], vsite.size());
                vsite.push_back((uint32_t)(s - s0)); vnode.push_back(sites->var_node[v]); vnuc.push_back(sites->var_nuc[v]);
            }
        }
        FS_TRY(hipMemcpy(d_refw.p, refw.data(), (size_t)W * 4, hipMemcpyHostToDevice));
        FS_TRY(d_vsite.upload(vsite.data(), vsite.size()));
        FS_TRY(d_vnode.upload(vnode.data(), vnode.size()));
        FS_TRY(d_vnuc.upload(vnuc.data(), vnuc.size()));
        FS_TRY(hipMemsetAsync(d_cnt.p, 0, 2 * sizeof(unsigned long long), stream));

        const uint64_t cells = (uint64_t)N * W;
        hipLaunchKernelGGL(k_fs_init, dim3((unsigned)std::min<uint64_t>((cells + 255) / 256, 1u << 20)), dim3(256), 0, stream, d_F.p,
                           d_refw.p, d_nchild.p, (uint64_t)N, W);
        if (!vsite.empty())
            hipLaunchKernelGGL(k_fs_scatter, dim3((unsigned)((vsite.size() + 255) / 256)), dim3(256), 0, stream, d_F.p, d_refw.p,
                               d_nchild.p, d_vsite.p, d_vnode.p, d_vnuc.p, (uint64_t)vsite.size(), W);
        const unsigned gy = (W + 63) / 64;
        for (uint32_t L = n_levels; L-- > 0;) {
            const uint32_t cnt = ilvl_off[L + 1] - ilvl_off[L];
            if (cnt)
                hipLaunchKernelGGL(k_fs_forward, dim3((cnt + 3) / 4, gy), dim3(256), 0, stream, d_F.p, d_inodes.p + ilvl_off[L], cnt,
                                   d_first.p, d_nchild.p, W);
        }
        for (uint32_t L = 0; L < n_levels; L++) {
            const uint32_t cnt = lvl_off[L + 1] - lvl_off[L];
            hipLaunchKernelGGL(k_fs_backward, dim3((cnt + 3) / 4, gy), dim3(256), 0, stream, d_F.p, d_refw.p, d_parent.p, lvl_off[L],
                               lvl_off[L + 1], W, d_cnt.p);
        }
        FS_TRY(hipGetLastError());
        unsigned long long n_mut = 0;
        FS_TRY(hipMemcpyAsync(&n_mut, d_cnt.p, sizeof(n_mut), hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        if (n_mut == 0) continue;
        FS_TRY(d_okey.alloc(n_mut));
        FS_TRY(d_okey2.alloc(n_mut));
        FS_TRY(d_oval.alloc(n_mut));
        FS_TRY(d_oval2.alloc(n_mut));
        hipLaunchKernelGGL(k_fs_emit, dim3((N + 3) / 4, gy), dim3(256), 0, stream, d_F.p, d_refw.p, d_parent.p, N, W, (uint32_t)s0,
                           d_cnt.p + 1, d_okey.p, d_oval.p);
        FS_TRY(hipGetLastError());
        // deterministic order: by site, then breadth-first node index
        size_t tmp_bytes = 0;
        FS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_okey.p, d_okey2.p, d_oval.p, d_oval2.p, (int64_t)n_mut, 0, 64, stream));
        FS_TRY(d_tmp.alloc(tmp_bytes));
        FS_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_okey.p, d_okey2.p, d_oval.p, d_oval2.p, (int64_t)n_mut, 0, 64, stream));
        h_key.resize(n_mut);
        h_val.resize(n_mut);
        FS_TRY(hipMemcpyAsync(h_key.data(), d_okey2.p, n_mut * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(h_val.data(), d_oval2.p, n_mut, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        for (unsigned long long i = 0; i < n_mut; i++) {
            res->site.push_back((uint32_t)(h_key[i] >> 32));
            res->node.push_back((uint32_t)h_key[i]);
            res->par.push_back(h_val[i] >> 4);
            res->nuc.push_back(h_val[i] & 0xF);
        }
    }
    guard.r = nullptr;
    *out = res;
    return UGP_OK;
}

extern "C" uint64_t ugp_fitch_count(const ugp_fitch *f) { return f ? f->site.size() : 0; }

extern "C" int ugp_fitch_get(const ugp_fitch *f, uint32_t *site, uint32_t *node, uint8_t *par_nuc, uint8_t *mut_nuc) {
    if (!f) return ugp::set_error(UGP_ERR_INVALID, "null handle");
    const size_t n = f->site.size();
    if (n && (!site || !node || !par_nuc || !mut_nuc)) return ugp::set_error(UGP_ERR_INVALID, "null output arrays");
    if (n) {
        memcpy(site, f->site.data(), n * 4);
        memcpy(node, f->node.data(), n * 4);
        memcpy(par_nuc, f->par.data(), n);
        memcpy(mut_nuc, f->nuc.data(), n);
    }
    return UGP_OK;
}

extern "C" void ugp_fitch_destroy(ugp_fitch *f) { delete f; }
