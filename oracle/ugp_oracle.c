/*
 * oracle/ugp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-file CPU restatement of the reference's placement hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (libusher_amd.so) never links, loads or
 * calls it.
 *
 * What is restated (every function cites the reference file:line it follows;
 * all paths are relative to /root/reference):
 *   - mapper2_body            src/usher_mapper.cpp:167-504   -> orc_mapper2()
 *   - the per-sample driver   src/usher_common.cpp:342-449   -> orc_place_sample()
 *   - Tree::get_num_leaves    src/mutation_annotated_tree.cpp:866-879
 *   - mapper_body (Fitch-Sankoff per VCF site)
 *                             src/usher_mapper.cpp:6-161     -> orc_fitch_site()
 *   - Node::add_mutation      src/mutation_annotated_tree.cpp:720-752 (used by
 *                             the Fitch-Sankoff restatement only)
 *
 * The restatement is LITERAL: it keeps the reference's control flow (the
 * start_index scans, the missing-base no-break, the O(A^2) ancestor walk, the
 * sort, the early returns) so that it reproduces the reference on odd inputs
 * too (unsorted / duplicated sample rows, masked mutations, ambiguous tree
 * alleles).  It is deliberately not "optimised into" the closed form the GPU
 * kernel uses; oracle/closed_form.py is the second, independent restatement.
 *
 * Pinning status.  The reference's own sources cannot be rebuilt inside this
 * repository: they need oneTBB and Boost headers and protoc-generated
 * parsimony.pb.h, none of which exist in the image, and the build rules forbid
 * stand-in headers (see DESIGN.md "Oracle").  The oracle is therefore pinned
 * against (a) the reference's only in-tree known-answer test,
 * scripts/testBranchLen2.{nwk,vcf,sh} (pins orc_fitch_site), and (b) the
 * reference outputs recorded during the survey stage (data files
 * under tests/golden/survey_ref/, provenance in its README) for placement:
 * 2,370 + 67,950 per-node scores, (score, num_best) for 5 + 50 + 64 samples in
 * -n mode, the tie sets / starred winners of a DEBUG=1 run, and the
 * default-mode (sequential add) results on the in-tree fixture.  No reference
 * binary is built or run by this repository: beyond those recorded outputs,
 * parity is unpinned.
 *
 * Nodes are addressed by their index in the reference's breadth-first
 * expansion (Tree::breadth_first_expansion, mutation_annotated_tree.cpp:
 * 1225-1251), i.e. exactly the "j" of usher_common.cpp:391-403.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef struct {
    int32_t position;      /* < 0  => masked (mutation_annotated_tree.hpp:76-78) */
    int8_t ref_nuc;        /* one-hot A=1 C=2 G=4 T=8 (mutation_annotated_tree.cpp:19-74) */
    int8_t par_nuc;
    int8_t mut_nuc;
    int8_t is_missing;
} orc_mut;

typedef struct orc_tree {
    int64_t n;
    int64_t *parent;       /* -1 for the root (j == 0)              */
    int64_t *mut_off;      /* CSR into muts, n + 1 entries          */
    orc_mut *muts;
    int64_t *child_off;    /* CSR into children                     */
    int64_t *children;
    int64_t *num_leaves;   /* Tree::get_num_leaves per node         */
} orc_tree;

/* ------------------------------------------------------------------ tree */

void orc_tree_destroy(orc_tree *t) {
    if (!t) return;
    free(t->parent); free(t->mut_off); free(t->muts);
    free(t->child_off); free(t->children); free(t->num_leaves);
    free(t);
}

/* parent[] is in BFS order: parent[0] == -1, parent[j] < j. */
orc_tree *orc_tree_create(int64_t n, const int64_t *parent, const int64_t *mut_off,
                          const int32_t *mut_pos, const int8_t *mut_ref,
                          const int8_t *mut_par, const int8_t *mut_nuc) {
    if (n <= 0 || parent[0] != -1) return NULL;
    orc_tree *t = (orc_tree *)calloc(1, sizeof(orc_tree));
    t->n = n;
    t->parent = (int64_t *)malloc(sizeof(int64_t) * n);
    t->mut_off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    memcpy(t->parent, parent, sizeof(int64_t) * n);
    memcpy(t->mut_off, mut_off, sizeof(int64_t) * (n + 1));
    int64_t m = mut_off[n];
    t->muts = (orc_mut *)malloc(sizeof(orc_mut) * (m > 0 ? m : 1));
    for (int64_t i = 0; i < m; i++) {
        t->muts[i].position = mut_pos[i];
        t->muts[i].ref_nuc = mut_ref[i];
        t->muts[i].par_nuc = mut_par[i];
        t->muts[i].mut_nuc = mut_nuc[i];
        t->muts[i].is_missing = 0;
    }
    t->child_off = (int64_t *)calloc(n + 1, sizeof(int64_t));
    t->children = (int64_t *)malloc(sizeof(int64_t) * n);
    for (int64_t j = 1; j < n; j++) {
        if (parent[j] < 0 || parent[j] >= j) { orc_tree_destroy(t); return NULL; }
        t->child_off[parent[j] + 1]++;
    }
    for (int64_t j = 0; j < n; j++) t->child_off[j + 1] += t->child_off[j];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * n);
    memcpy(fill, t->child_off, sizeof(int64_t) * n);
    for (int64_t j = 1; j < n; j++) t->children[fill[parent[j]]++] = j;  /* keeps child order */
    free(fill);
    /* Tree::get_num_leaves (mutation_annotated_tree.cpp:866-879): leaf -> 1,
     * else the sum over children.  parent[j] < j lets one reverse sweep do it. */
    t->num_leaves = (int64_t *)calloc(n, sizeof(int64_t));
    for (int64_t j = n - 1; j >= 0; j--) {
        if (t->child_off[j + 1] == t->child_off[j]) t->num_leaves[j] = 1;
        if (j > 0) t->num_leaves[parent[j]] += t->num_leaves[j];
    }
    return t;
}

int64_t orc_tree_num_leaves(const orc_tree *t, int64_t j) { return t->num_leaves[j]; }

static inline int is_leaf(const orc_tree *t, int64_t j) { return t->child_off[j + 1] == t->child_off[j]; }
static inline int is_root(const orc_tree *t, int64_t j) { return t->parent[j] < 0; }
static inline int is_masked(const orc_mut *m) { return m->position < 0; }

/* ------------------------------------------------------------ mapper2 */

typedef struct { orc_mut *v; int64_t n, cap; } mvec;
static void mv_push(mvec *a, orc_mut m) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 32; a->v = (orc_mut *)realloc(a->v, sizeof(orc_mut) * a->cap); }
    a->v[a->n++] = m;
}
static int cmp_mut_pos(const void *a, const void *b) {
    int32_t x = ((const orc_mut *)a)->position, y = ((const orc_mut *)b)->position;
    return (x > y) - (x < y);
}

/* Shared per-sample search state: the pointer bundle of mapper2_input,
 * usher_graph.hpp:73-101. */
typedef struct {
    int best_set_difference;
    int64_t best_node_num_leaves;
    int64_t best_j;
    int64_t num_best;
    int best_has_unique;          /* *input.has_unique          */
    int8_t *node_has_unique;      /* (*input.node_has_unique)[] */
    int64_t *best_j_vec;          /* *input.best_j_vec          */
    int64_t best_j_vec_n, best_j_vec_cap;
    /* the other callers of mapper2_body (matUtils uncertainty / annotate / merge, ripples) pass their own
     * index j -- position in THEIR node vector, DFS or a sub-BFS -- and ripples a per-node distance
     * (mapper2_input::j, ::distance, usher_graph.hpp:82-85).  jidx < 0: j is the BFS index itself, distance 0. */
    int64_t jidx;                 /* input.j of the node being scored      */
    int64_t distance;             /* input.distance                        */
    int64_t best_distance;        /* *input.best_distance                  */
} orc_shared;

static void bjv_push(orc_shared *s, int64_t j) {
    if (s->best_j_vec_n == s->best_j_vec_cap) {
        s->best_j_vec_cap = s->best_j_vec_cap ? s->best_j_vec_cap * 2 : 16;
        s->best_j_vec = (int64_t *)realloc(s->best_j_vec, sizeof(int64_t) * s->best_j_vec_cap);
    }
    s->best_j_vec[s->best_j_vec_n++] = j;
}

/*
 * mapper2_body, usher_mapper.cpp:167-504, for node j and one sample.
 * `sm`/`n_sm` = *input.missing_sample_mutations.  set_difference_out may be
 * NULL unless compute_parsimony_scores.  excess/imputed may be NULL unless
 * compute_vecs.  scratch vectors are passed in to avoid per-call malloc.
 */
static void orc_mapper2(const orc_tree *t, int64_t j, const orc_mut *sm, int64_t n_sm,
                        orc_shared *sh, int compute_parsimony_scores, int compute_vecs,
                        int *set_difference_out, mvec *excess, mvec *imputed,
                        mvec *anc /* scratch: ancestral_mutations */,
                        int *has_unique_out /* may be NULL */) {
    int set_difference = 0;                                   /* :172 */
    int best_set_difference = sh->best_set_difference;        /* :176 */
    anc->n = 0;                                               /* :178-179 (anc_positions == positions of anc) */
    int has_unique = 0;                                       /* :183 */
    int node_num_mut = 0, num_common_mut = 0;                 /* :184-185 */
    const orc_mut *nm = t->muts + t->mut_off[j];
    int64_t n_nm = t->mut_off[j + 1] - t->mut_off[j];

    if (!is_root(t, j)) {                                     /* :190 */
        int64_t start_index = 0;                              /* :191 */
        for (int64_t a = 0; a < n_nm; a++) {                  /* :192 */
            const orc_mut m1 = nm[a];
            node_num_mut++;                                   /* :193 */
            int8_t anc_nuc = m1.mut_nuc;                      /* :194 */
            if (is_masked(&m1)) { has_unique = 1; break; }    /* :197-200 */
            int found = 0, found_pos = 0;                     /* :202-203 */
            for (int64_t k = start_index; k < n_sm; k++) {    /* :204 */
                const orc_mut m2 = sm[k];
                start_index = k;                              /* :206 */
                if (m1.position == m2.position) {             /* :207 */
                    found_pos = 1;
                    if (m2.is_missing) {                      /* :209-211 (no break) */
                        found = 1;
                        num_common_mut++;
                    } else {
                        int8_t nuc = m2.mut_nuc;              /* :213 */
                        if ((nuc & anc_nuc) != 0) {           /* :214 */
                            orc_mut m;
                            m.position = m1.position; m.ref_nuc = m1.ref_nuc;
                            m.par_nuc = m1.par_nuc; m.mut_nuc = anc_nuc; m.is_missing = 0;
                            mv_push(anc, m);                  /* :222-223 */
                            if (compute_vecs) mv_push(excess, m);   /* :225-227 */
                            found = 1;
                            num_common_mut++;
                            break;                            /* :233-235 */
                        }
                    }
                }
                if (m1.position < m2.position) break;         /* :239-241 */
            }
            if (!found) {                                     /* :243 */
                if (!found_pos && (anc_nuc == m1.ref_nuc)) {  /* :244 */
                    orc_mut m;
                    m.position = m1.position; m.ref_nuc = m1.ref_nuc;
                    m.par_nuc = m1.par_nuc; m.mut_nuc = anc_nuc; m.is_missing = 0;
                    mv_push(anc, m);                          /* :252-253 */
                    if (compute_vecs) mv_push(excess, m);     /* :255-257 */
                    num_common_mut++;                         /* :259 */
                } else {
                    has_unique = 1;                           /* :261 */
                }
            }
        }
    } else {
        for (int64_t a = 0; a < n_nm; a++) mv_push(anc, nm[a]);   /* :266-269 */
    }

    /* :275-286 ancestor walk, most recent mutation per position wins */
    {
        int64_t n = j;
        while (t->parent[n] >= 0) {
            n = t->parent[n];
            const orc_mut *pm = t->muts + t->mut_off[n];
            int64_t n_pm = t->mut_off[n + 1] - t->mut_off[n];
            for (int64_t a = 0; a < n_pm; a++) {
                if (is_masked(&pm[a])) continue;
                int seen = 0;
                for (int64_t q = 0; q < anc->n; q++)
                    if (anc->v[q].position == pm[a].position) { seen = 1; break; }
                if (!seen) mv_push(anc, pm[a]);
            }
        }
    }

    /* :289 sort by position.  (std::sort is unstable; ties only occur between
     * masked root mutations, which never influence the result.) */
    qsort(anc->v, (size_t)anc->n, sizeof(orc_mut), cmp_mut_pos);

    /* :292-388 iterate over the sample's mutations */
    for (int64_t a = 0; a < n_sm; a++) {
        const orc_mut m1 = sm[a];
        if (m1.is_missing) continue;                          /* :294-296 */
        int found_pos = 0, found = 0, has_ref = 0;            /* :298-300 */
        int8_t anc_nuc = m1.ref_nuc;                          /* :301 */
        if ((m1.mut_nuc & m1.ref_nuc) != 0) has_ref = 1;      /* :302-304 */
        for (int64_t k = 0; k < anc->n; k++) {                /* :306 */
            const orc_mut m2 = anc->v[k];
            if (is_masked(&m2)) continue;                     /* :309-311 */
            if (m1.position == m2.position) {                 /* :313 */
                found_pos = 1;
                anc_nuc = m2.mut_nuc;
                if ((m1.mut_nuc & anc_nuc) != 0) found = 1;
                break;
            }
        }
        int ambiguous = (m1.mut_nuc & (m1.mut_nuc - 1)) != 0;
        if (found) {                                          /* :322-335 */
            if (compute_vecs && ambiguous) {
                orc_mut m; m.position = m1.position; m.ref_nuc = m1.ref_nuc;
                m.par_nuc = anc_nuc; m.mut_nuc = anc_nuc; m.is_missing = 0;
                mv_push(imputed, m);
            }
        } else if (!found_pos && has_ref) {                   /* :341-351 */
            if (compute_vecs && ambiguous) {
                orc_mut m; m.position = m1.position; m.ref_nuc = m1.ref_nuc;
                m.par_nuc = anc_nuc; m.mut_nuc = m1.ref_nuc; m.is_missing = 0;
                mv_push(imputed, m);
            }
        } else {                                              /* :356-387 */
            orc_mut m; m.position = m1.position; m.ref_nuc = m1.ref_nuc;
            m.par_nuc = anc_nuc; m.is_missing = 0;
            m.mut_nuc = 0;   /* the reference leaves this uninitialised when no bit is set */
            if (has_ref) {
                m.mut_nuc = m1.ref_nuc;
            } else {
                for (int b = 0; b < 4; b++)
                    if (((1 << b) & m1.mut_nuc) != 0) { m.mut_nuc = (int8_t)(1 << b); break; }
            }
            if (compute_vecs && ambiguous) mv_push(imputed, m);   /* :375-377 */
            if (m.mut_nuc != m.par_nuc) {                     /* :378 */
                if (compute_vecs) mv_push(excess, m);
                set_difference += 1;
                if (!compute_parsimony_scores && (set_difference > best_set_difference)) return;  /* :383-385 */
            }
        }
    }

    /* :393-445 back-mutations */
    for (int64_t a = 0; a < anc->n; a++) {
        const orc_mut m1 = anc->v[a];
        int found = 0, found_pos = 0;
        int8_t anc_nuc = m1.mut_nuc;
        for (int64_t k = 0; k < n_sm; k++) {                  /* :398 */
            if (is_masked(&m1)) break;                        /* :401-403 */
            const orc_mut m2 = sm[k];
            if (m1.position == m2.position) {
                found_pos = 1;
                if (m2.is_missing) { found = 1; break; }      /* :409-412 */
                if ((m2.mut_nuc & anc_nuc) != 0) found = 1;   /* :413-415 */
            }
        }
        if (found) {
        } else if (!found_pos && !is_masked(&m1) && (anc_nuc == m1.ref_nuc)) {
        } else if (found_pos && !found) {
        } else {                                              /* :427-444 */
            orc_mut m; m.position = m1.position; m.ref_nuc = m1.ref_nuc;
            m.par_nuc = anc_nuc; m.mut_nuc = m1.ref_nuc; m.is_missing = 0;
            if (m.mut_nuc != m.par_nuc) {
                set_difference += 1;
                if (!compute_parsimony_scores && (set_difference > best_set_difference)) return;
                if (compute_vecs) mv_push(excess, m);
            }
        }
    }

    if (compute_parsimony_scores) *set_difference_out = set_difference;   /* :448-450 */
    if (has_unique_out) *has_unique_out = has_unique;

    int leaf = is_leaf(t, j);
    if (is_root(t, j) ||
        ((has_unique && !leaf && (num_common_mut > 0) && (node_num_mut != num_common_mut)) ||
         (leaf && (num_common_mut > 0)) ||
         (!has_unique && !leaf && (node_num_mut == num_common_mut)))) {          /* :454-455 */
        if (set_difference > sh->best_set_difference) return;                   /* :457-461 */
        const int64_t jj = sh->jidx >= 0 ? sh->jidx : j;                        /* input.j */
        int64_t num_leaves = t->num_leaves[j];                                  /* :464 */
        if (set_difference < sh->best_set_difference) {                         /* :465-475 */
            sh->best_set_difference = set_difference;
            sh->best_node_num_leaves = num_leaves;
            sh->best_j = jj;
            sh->num_best = 1;
            sh->best_has_unique = has_unique;
            sh->best_distance = sh->distance;
            sh->node_has_unique[jj] = (int8_t)has_unique;
            sh->best_j_vec_n = 0;
            bjv_push(sh, jj);
        } else if (set_difference == sh->best_set_difference) {                 /* :476-497 */
            if (((sh->distance == sh->best_distance) &&
                 ((num_leaves > sh->best_node_num_leaves) ||
                  ((num_leaves == sh->best_node_num_leaves) && (sh->best_j < jj)))) ||
                (sh->distance < sh->best_distance)) {
                sh->best_set_difference = set_difference;
                sh->best_node_num_leaves = num_leaves;
                sh->best_j = jj;
                sh->best_has_unique = has_unique;
                sh->best_distance = sh->distance;
            }
            sh->num_best += 1;
            sh->node_has_unique[jj] = (int8_t)has_unique;
            bjv_push(sh, jj);
        }
    } else if (compute_parsimony_scores) {
        *set_difference_out = set_difference + 1;                               /* :498-503 */
    }
}

static orc_mut *make_sample(int64_t n_ent, const int32_t *pos, const int8_t *ref,
                            const int8_t *nuc, const int8_t *is_missing) {
    orc_mut *sm = (orc_mut *)malloc(sizeof(orc_mut) * (n_ent > 0 ? n_ent : 1));
    for (int64_t i = 0; i < n_ent; i++) {
        sm[i].position = pos[i]; sm[i].ref_nuc = ref[i]; sm[i].par_nuc = ref[i];   /* read_vcf :2244 */
        sm[i].mut_nuc = nuc[i]; sm[i].is_missing = is_missing[i];
    }
    return sm;
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/*
 * One iteration of the per-sample block, usher_common.cpp:342-449, on a static
 * tree.  compute_scores != 0 is the -p path (:412 with print_parsimony_scores):
 * one pass with (true,true), scores[] filled.  Otherwise pass 1 is
 * mapper2_body(inp,false,false) over all nodes (:389-414) and pass 2 re-runs
 * the tied nodes with best+1 (:416-449).
 *
 * Outputs: *out_best, *out_num_best, *out_best_j, *out_has_unique; scores[n]
 * (only with compute_scores); tied node indices sorted ascending into
 * best_j_vec[0..min(num_best,cap)) with their node_has_unique flags in
 * tied_has_unique[] (either may be NULL).  Returns 0.
 */
int orc_place_sample(const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                     const int8_t *nuc, const int8_t *is_missing, int compute_scores,
                     int32_t *out_best, int64_t *out_num_best, int64_t *out_best_j,
                     int8_t *out_has_unique, int32_t *scores,
                     int64_t *best_j_vec, int64_t cap, int8_t *tied_has_unique) {
    orc_mut *sm = make_sample(n_ent, pos, ref, nuc, is_missing);
    orc_shared sh; memset(&sh, 0, sizeof(sh));
    sh.jidx = -1;
    sh.node_has_unique = (int8_t *)calloc((size_t)t->n, 1);                   /* :379 */
    int64_t root_muts = t->mut_off[1] - t->mut_off[0];
    sh.best_set_difference = (int)(n_ent + root_muts + 1);                    /* :374 */
    sh.best_j = 0; sh.num_best = 1; sh.best_has_unique = 0;                   /* :376-384 */
    sh.best_node_num_leaves = 0;                                              /* :367 */
    bjv_push(&sh, 0);                                                         /* :381 */
    mvec anc = {0, 0, 0}, ex = {0, 0, 0}, im = {0, 0, 0};

    for (int64_t k = 0; k < t->n; k++) {                                      /* :389-414 */
        int sd = 0;
        ex.n = 0; im.n = 0;
        orc_mapper2(t, k, sm, n_ent, &sh, compute_scores, compute_scores, &sd, &ex, &im, &anc, NULL);
        if (compute_scores) scores[k] = sd;
    }
    if (!compute_scores) {                                                    /* :416-449 */
        sh.best_set_difference += 1;
        int64_t ntmp = sh.best_j_vec_n;
        int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (ntmp > 0 ? ntmp : 1));
        memcpy(tmp, sh.best_j_vec, sizeof(int64_t) * ntmp);
        sh.num_best = 0;
        sh.best_j_vec_n = 0;
        for (int64_t l = 0; l < ntmp; l++) {
            ex.n = 0; im.n = 0;
            orc_mapper2(t, tmp[l], sm, n_ent, &sh, 0, 1, NULL, &ex, &im, &anc, NULL);
        }
        free(tmp);
    }
    *out_best = sh.best_set_difference;
    *out_num_best = sh.num_best;
    *out_best_j = sh.best_j;
    *out_has_unique = (int8_t)sh.best_has_unique;
    if (best_j_vec) {
        qsort(sh.best_j_vec, (size_t)sh.best_j_vec_n, sizeof(int64_t), cmp_i64);  /* :588 */
        for (int64_t i = 0; i < sh.best_j_vec_n && i < cap; i++) {
            best_j_vec[i] = sh.best_j_vec[i];
            if (tied_has_unique) tied_has_unique[i] = sh.node_has_unique[sh.best_j_vec[i]];
        }
    }
    free(sh.node_has_unique); free(sh.best_j_vec);
    free(anc.v); free(ex.v); free(im.v); free(sm);
    return 0;
}

/*
 * (*input.node_has_unique)[0 .. k) after the two passes of usher_common.cpp:389-449 run in breadth-first order on one
 * thread: what --multiple-placements reads when it indexes the per-node flags with its loop counter (:647).  An entry
 * is the node's has_unique if the node matched or beat the running optimum when it was visited, or is optimal; else 0.
 */
int orc_node_has_unique_prefix(const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                               const int8_t *nuc, const int8_t *is_missing, int64_t k, int8_t *out) {
    orc_mut *sm = make_sample(n_ent, pos, ref, nuc, is_missing);
    orc_shared sh; memset(&sh, 0, sizeof(sh));
    sh.jidx = -1;
    sh.node_has_unique = (int8_t *)calloc((size_t)t->n, 1);
    sh.best_set_difference = (int)(n_ent + (t->mut_off[1] - t->mut_off[0]) + 1);
    sh.best_j = 0; sh.num_best = 1;
    bjv_push(&sh, 0);
    mvec anc = {0, 0, 0}, ex = {0, 0, 0}, im = {0, 0, 0};
    for (int64_t j = 0; j < t->n; j++) orc_mapper2(t, j, sm, n_ent, &sh, 0, 0, NULL, &ex, &im, &anc, NULL);
    sh.best_set_difference += 1;
    int64_t ntmp = sh.best_j_vec_n;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (ntmp > 0 ? ntmp : 1));
    memcpy(tmp, sh.best_j_vec, sizeof(int64_t) * ntmp);
    sh.num_best = 0; sh.best_j_vec_n = 0;
    for (int64_t l = 0; l < ntmp; l++) { ex.n = 0; im.n = 0; orc_mapper2(t, tmp[l], sm, n_ent, &sh, 0, 1, NULL, &ex, &im, &anc, NULL); }
    for (int64_t j = 0; j < k && j < t->n; j++) out[j] = sh.node_has_unique[j];
    free(tmp); free(sh.node_has_unique); free(sh.best_j_vec); free(anc.v); free(ex.v); free(im.v); free(sm);
    return 0;
}

/*
 * The search as the OTHER callers of mapper2_body run it: over a caller-supplied node vector `nodes[0..n_list)`
 * (BFS indices of this tree) whose POSITION k is the index j handed to mapper2_body -- a depth-first expansion
 * with one node left out (matUtils uncertainty.cpp:212-235, the node itself; annotate.cpp:615-638), a
 * breadth-first expansion of a subtree cut at max_levels (merge.cpp:253-280), all nodes with enough
 * descendants and a per-node distance (ripples/main.cpp:343-377).  compute_scores: mapper2_body(inp, true)
 * as ripples calls it (one pass, scores[k] filled, no second pass); otherwise pass 1 only, as uncertainty /
 * annotate / merge call it (mapper2_body(inp, false); the early return of :383-385 only ever drops nodes that
 * are strictly worse than the best so far, so num_best / best_j_vec are those of a full scan).
 * init_best: the caller's initial *best_set_difference (1e9 in annotate / ripples, |S| + |root muts| + 1 in
 * uncertainty / merge).  The initial best_j_vec = {0}, num_best = 1 of every caller is reproduced.
 * Outputs: best, num_best, best_j (the entry's j), has_unique, the tied j ascending.  jidx[] must stay below t->n + n_list.
 */
int orc_place_sample_list(const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                          const int8_t *nuc, const int8_t *is_missing,
                          int64_t n_list, const int64_t *nodes, const int64_t *jidx /* index j per entry; NULL: its position */,
                          const int64_t *distance /* may be NULL */, int compute_scores, int32_t init_best, int64_t init_best_distance,
                          int32_t *out_best, int64_t *out_num_best, int64_t *out_best_j, int8_t *out_has_unique,
                          int32_t *scores /* [n_list] or NULL */, int64_t *best_j_vec, int64_t cap, int8_t *tied_has_unique) {
    orc_mut *sm = make_sample(n_ent, pos, ref, nuc, is_missing);
    orc_shared sh; memset(&sh, 0, sizeof(sh));
    sh.node_has_unique = (int8_t *)calloc((size_t)t->n + (size_t)n_list + 1, 1);
    sh.best_set_difference = init_best;
    sh.best_j = 0; sh.num_best = 1; sh.best_has_unique = 0; sh.best_node_num_leaves = 0;
    sh.best_distance = init_best_distance;
    bjv_push(&sh, 0);
    mvec anc = {0, 0, 0}, ex = {0, 0, 0}, im = {0, 0, 0};
    for (int64_t k = 0; k < n_list; k++) {
        int sd = 0;
        ex.n = 0; im.n = 0;
        sh.jidx = jidx ? jidx[k] : k;
        sh.distance = distance ? distance[k] : 0;
        orc_mapper2(t, nodes[k], sm, n_ent, &sh, compute_scores, 1, &sd, &ex, &im, &anc, NULL);
        if (compute_scores && scores) scores[k] = sd;
    }
    *out_best = sh.best_set_difference;
    *out_num_best = sh.num_best;
    *out_best_j = sh.best_j;
    *out_has_unique = (int8_t)sh.best_has_unique;
    if (best_j_vec) {
        qsort(sh.best_j_vec, (size_t)sh.best_j_vec_n, sizeof(int64_t), cmp_i64);
        for (int64_t i = 0; i < sh.best_j_vec_n && i < cap; i++) {
            best_j_vec[i] = sh.best_j_vec[i];
            if (tied_has_unique) tied_has_unique[i] = sh.node_has_unique[sh.best_j_vec[i]];
        }
    }
    free(sh.node_has_unique); free(sh.best_j_vec);
    free(anc.v); free(ex.v); free(im.v); free(sm);
    return 0;
}

/*
 * Excess / imputed mutation vectors for placing the sample at node j:
 * mapper2_body(inp, false, true) as run by pass 2 (usher_common.cpp:426-449)
 * with best_set_difference large enough never to return early.  Arrays are
 * caller-allocated with capacity cap; counts are returned through n_excess /
 * n_imputed (the true counts, even when > cap).
 */
int orc_node_vecs(const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                  const int8_t *nuc, const int8_t *is_missing, int64_t j, int64_t cap,
                  int32_t *ex_pos, int8_t *ex_ref, int8_t *ex_par, int8_t *ex_mut, int64_t *n_excess,
                  int32_t *im_pos, int8_t *im_ref, int8_t *im_par, int8_t *im_mut, int64_t *n_imputed,
                  int32_t *set_difference, int8_t *has_unique) {
    orc_mut *sm = make_sample(n_ent, pos, ref, nuc, is_missing);
    orc_shared sh; memset(&sh, 0, sizeof(sh));
    sh.jidx = -1;
    sh.node_has_unique = (int8_t *)calloc((size_t)t->n, 1);
    sh.best_set_difference = 0x3fffffff;
    mvec anc = {0, 0, 0}, ex = {0, 0, 0}, im = {0, 0, 0};
    int sd = 0, hu = 0;
    orc_mapper2(t, j, sm, n_ent, &sh, 1, 1, &sd, &ex, &im, &anc, &hu);
    for (int64_t i = 0; i < ex.n && i < cap; i++) {
        ex_pos[i] = ex.v[i].position; ex_ref[i] = ex.v[i].ref_nuc; ex_par[i] = ex.v[i].par_nuc; ex_mut[i] = ex.v[i].mut_nuc;
    }
    for (int64_t i = 0; i < im.n && i < cap; i++) {
        im_pos[i] = im.v[i].position; im_ref[i] = im.v[i].ref_nuc; im_par[i] = im.v[i].par_nuc; im_mut[i] = im.v[i].mut_nuc;
    }
    *n_excess = ex.n; *n_imputed = im.n; *set_difference = sd;
    *has_unique = (int8_t)hu;   /* the branch loop's has_unique (:183-264), eligible or not */
    free(sh.node_has_unique); free(sh.best_j_vec);
    free(anc.v); free(ex.v); free(im.v); free(sm);
    return 0;
}

/* ------------------------------------------------- threaded CPU baseline */

/*
 * Persistent worker pool (the reference keeps its TBB workers alive across
 * samples, usher.cpp:117; tbb::parallel_for hands out node ranges dynamically,
 * usher_common.cpp:389).  Workers sleep on a condition variable between
 * samples; within a sample they pull fixed-size node ranges from an atomic
 * counter, so deep (expensive) regions of the BFS order do not end up on one
 * thread.  Test infrastructure: only the timed CPU baseline uses it.
 */
#define ORC_GRAIN 2048

typedef struct {
    const orc_tree *t; const orc_mut *sm; int64_t n_sm;
    int init_best;
    orc_shared sh;
} orc_job;

typedef struct orc_pool {
    int nthreads;                 /* workers, not counting the caller */
    pthread_t *th;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    uint64_t epoch;               /* bumped per sample */
    int running, stop;
    orc_job *jobs;                /* [nthreads + 1], slot nthreads = the caller */
    int64_t next;                 /* next unclaimed node (atomic) */
    int64_t n;
    int shared_best;              /* smallest best_set_difference any worker has found (atomic): the reference's
                                     workers all read ONE *input.best_set_difference (usher_mapper.cpp:457-461) */
} orc_pool;

static void orc_run_ranges(orc_pool *pl, orc_job *jb) {
    mvec anc = {0, 0, 0};
    for (;;) {
        int64_t lo = __atomic_fetch_add(&pl->next, ORC_GRAIN, __ATOMIC_RELAXED);
        if (lo >= pl->n) break;
        int64_t hi = lo + ORC_GRAIN < pl->n ? lo + ORC_GRAIN : pl->n;
        /* adopt a strictly better score another worker has found: this worker's own ties are then obsolete
         * (the early return of usher_mapper.cpp:383-385 only ever drops nodes that are strictly worse) */
        int g = __atomic_load_n(&pl->shared_best, __ATOMIC_RELAXED);
        if (g < jb->sh.best_set_difference) {
            jb->sh.best_set_difference = g; jb->sh.num_best = 0; jb->sh.best_j_vec_n = 0;
            jb->sh.best_node_num_leaves = -1; jb->sh.best_j = -1;   /* no winner of its own at this score yet */
        }
        for (int64_t k = lo; k < hi; k++)
            orc_mapper2(jb->t, k, jb->sm, jb->n_sm, &jb->sh, 0, 0, NULL, NULL, NULL, &anc, NULL);
        g = __atomic_load_n(&pl->shared_best, __ATOMIC_RELAXED);
        while (jb->sh.best_set_difference < g &&
               !__atomic_compare_exchange_n(&pl->shared_best, &g, jb->sh.best_set_difference, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
    free(anc.v);
}

typedef struct { orc_pool *pl; int idx; } orc_warg;

static void *orc_pool_worker(void *p) {
    orc_warg *wa = (orc_warg *)p;
    orc_pool *pl = wa->pl; int idx = wa->idx;
    free(wa);
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&pl->mu);
        while (!pl->stop && pl->epoch == seen) pthread_cond_wait(&pl->cv_go, &pl->mu);
        if (pl->stop) { pthread_mutex_unlock(&pl->mu); return NULL; }
        seen = pl->epoch;
        pthread_mutex_unlock(&pl->mu);
        orc_run_ranges(pl, &pl->jobs[idx]);
        pthread_mutex_lock(&pl->mu);
        if (--pl->running == 0) pthread_cond_signal(&pl->cv_done);
        pthread_mutex_unlock(&pl->mu);
    }
}

orc_pool *orc_pool_create(int nthreads) {
    if (nthreads < 1) nthreads = 1;
    orc_pool *pl = (orc_pool *)calloc(1, sizeof(orc_pool));
    pl->nthreads = nthreads - 1;
    pl->jobs = (orc_job *)calloc((size_t)nthreads, sizeof(orc_job));
    pl->th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    pthread_mutex_init(&pl->mu, NULL);
    pthread_cond_init(&pl->cv_go, NULL);
    pthread_cond_init(&pl->cv_done, NULL);
    for (int i = 0; i < pl->nthreads; i++) {
        orc_warg *wa = (orc_warg *)malloc(sizeof(orc_warg));
        wa->pl = pl; wa->idx = i;
        pthread_create(&pl->th[i], NULL, orc_pool_worker, wa);
    }
    return pl;
}

void orc_pool_destroy(orc_pool *pl) {
    if (!pl) return;
    pthread_mutex_lock(&pl->mu);
    pl->stop = 1;
    pthread_cond_broadcast(&pl->cv_go);
    pthread_mutex_unlock(&pl->mu);
    for (int i = 0; i < pl->nthreads; i++) pthread_join(pl->th[i], NULL);
    pthread_mutex_destroy(&pl->mu); pthread_cond_destroy(&pl->cv_go); pthread_cond_destroy(&pl->cv_done);
    free(pl->th); free(pl->jobs); free(pl);
}

/*
 * Node-parallel pass 1 (usher_common.cpp:389-414) on the pool, one sample at a
 * time as the reference's file-scope locks force (usher_mapper.cpp:3-4).  Each
 * worker keeps a private copy of the shared state; the copies are merged with
 * the same (score, num_leaves, j) rule, which is order-independent.  Used only
 * as the timed CPU baseline; returns the same (best, num_best, best_j).
 */
int orc_place_sample_pool(orc_pool *pl, const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                          const int8_t *nuc, const int8_t *is_missing,
                          int32_t *out_best, int64_t *out_num_best, int64_t *out_best_j) {
    orc_mut *sm = make_sample(n_ent, pos, ref, nuc, is_missing);
    int64_t root_muts = t->mut_off[1] - t->mut_off[0];
    int init_best = (int)(n_ent + root_muts + 1);
    int8_t *nhu = (int8_t *)calloc((size_t)t->n, 1);
    const int nj = pl->nthreads + 1;
    for (int i = 0; i < nj; i++) {
        memset(&pl->jobs[i], 0, sizeof(orc_job));
        pl->jobs[i].sh.jidx = -1;
        pl->jobs[i].t = t; pl->jobs[i].sm = sm; pl->jobs[i].n_sm = n_ent;
        pl->jobs[i].sh.best_set_difference = init_best;
        pl->jobs[i].sh.node_has_unique = nhu;   /* disjoint indices per worker */
        pl->jobs[i].sh.num_best = 0;
    }
    pthread_mutex_lock(&pl->mu);
    pl->n = t->n; pl->next = 0; pl->running = pl->nthreads; pl->shared_best = init_best; pl->epoch++;
    pthread_cond_broadcast(&pl->cv_go);
    pthread_mutex_unlock(&pl->mu);
    orc_run_ranges(pl, &pl->jobs[pl->nthreads]);   /* the caller works too */
    pthread_mutex_lock(&pl->mu);
    while (pl->running > 0) pthread_cond_wait(&pl->cv_done, &pl->mu);
    pthread_mutex_unlock(&pl->mu);
    int best = init_best; int64_t nb = 0, bj = 0, bl = -1;
    for (int i = 0; i < nj; i++) {
        orc_shared *s = &pl->jobs[i].sh;
        if (s->num_best == 0) { free(s->best_j_vec); continue; }
        if (s->best_set_difference < best) {
            best = s->best_set_difference; nb = s->num_best; bj = s->best_j; bl = s->best_node_num_leaves;
        } else if (s->best_set_difference == best) {
            nb += s->num_best;
            if (s->best_node_num_leaves > bl || (s->best_node_num_leaves == bl && s->best_j > bj)) {
                bj = s->best_j; bl = s->best_node_num_leaves;
            }
        }
        free(s->best_j_vec);
    }
    *out_best = best; *out_num_best = nb; *out_best_j = bj;
    free(nhu); free(sm);
    return 0;
}

/* Convenience form kept for the tests: a pool per (process, thread count), created on first use. */
int orc_place_sample_mt(const orc_tree *t, int64_t n_ent, const int32_t *pos, const int8_t *ref,
                        const int8_t *nuc, const int8_t *is_missing, int nthreads,
                        int32_t *out_best, int64_t *out_num_best, int64_t *out_best_j) {
    static orc_pool *pool = NULL;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    if (nthreads < 1) nthreads = 1;
    pthread_mutex_lock(&mu);   /* one sample at a time (usher_mapper.cpp:3-4) */
    if (!pool || pool->nthreads + 1 != nthreads) { orc_pool_destroy(pool); pool = orc_pool_create(nthreads); }
    int rc = orc_place_sample_pool(pool, t, n_ent, pos, ref, nuc, is_missing, out_best, out_num_best, out_best_j);
    pthread_mutex_unlock(&mu);
    return rc;
}

/* -------------------------------------------- closed form, at full size */

/*
 * The closed form of mapper2_body (oracle/closed_form.py is the
 * readable statement, checked against the literal restatement above on every
 * small test tree) as one O(N + M) sweep per sample in BFS order, so that the
 * parity tests can check EVERY sample of a 10M-node batch and not only the
 * handful the literal O(N * depth) routine finishes in seconds:
 *     D(n)    = D(parent) + sum_m ([prev(m) in S] - [mut(m) in S])      (non-masked m)
 *     cost(n) = D(parent) + sum_{m before the first masked one} min(delta, 0);  cost(root) = D(root)
 *     common  = #{m before the first masked one : mut(m) in S}
 *     eligible = root | common > 0 | (internal & no mutations)          usher_mapper.cpp:454-455
 *     has_unique = masked | common != num_mut                           usher_mapper.cpp:197-264
 *     winner = argmax (num_leaves, j) among the eligible minima         usher_mapper.cpp:476-497
 * prev(m) = the true parent state at m's position (usher_mapper.cpp:275-286
 * ignores the stored par_nuc as well), derived once per tree by a DFS.
 * Preconditions as in closed_form.py: sample rows sorted, no duplicate
 * positions; tree alleles one-hot.  Returns NULL when the tree violates them.
 */
typedef struct orc_cf {
    const orc_tree *t;
    int8_t *prev;        /* [M] true parent-state allele of every mutation (0 for masked) */
    int32_t max_pos;
    int8_t *ref_at;      /* [max_pos + 1] reference base at positions the tree mutates, else 0 */
} orc_cf;

void orc_cf_destroy(orc_cf *c) { if (!c) return; free(c->prev); free(c->ref_at); free(c); }

orc_cf *orc_cf_create(const orc_tree *t) {
    orc_cf *c = (orc_cf *)calloc(1, sizeof(orc_cf));
    c->t = t;
    const int64_t M = t->mut_off[t->n];
    c->prev = (int8_t *)calloc((size_t)(M > 0 ? M : 1), 1);
    int32_t mp = 0;
    for (int64_t i = 0; i < M; i++) if (t->muts[i].position > mp) mp = t->muts[i].position;
    c->max_pos = mp;
    c->ref_at = (int8_t *)calloc((size_t)mp + 1, 1);
    int8_t *state = (int8_t *)calloc((size_t)mp + 1, 1);   /* 0 = reference */
    for (int64_t i = 0; i < M; i++) {
        const orc_mut *m = &t->muts[i];
        if (m->position < 0) continue;
        if (m->mut_nuc <= 0 || (m->mut_nuc & (m->mut_nuc - 1)) || m->mut_nuc > 8) { free(state); orc_cf_destroy(c); return NULL; }
        if (c->ref_at[m->position] == 0) c->ref_at[m->position] = m->ref_nuc;
    }
    /* iterative DFS with an undo log */
    typedef struct { int32_t pos; int8_t old; } undo_t;
    undo_t *undo = (undo_t *)malloc(sizeof(undo_t) * (size_t)(M > 0 ? M : 1));
    int64_t n_undo = 0;
    int64_t *stk_node = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->n);
    int64_t *stk_next = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->n);
    int64_t *stk_mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->n);
    int64_t sp = 0;
    stk_node[0] = 0; stk_next[0] = t->child_off[0]; stk_mark[0] = 0; sp = 1;
    for (int64_t i = t->mut_off[0]; i < t->mut_off[1]; i++) {
        const orc_mut *m = &t->muts[i];
        if (m->position < 0) continue;
        c->prev[i] = state[m->position] ? state[m->position] : c->ref_at[m->position];
        undo[n_undo].pos = m->position; undo[n_undo].old = state[m->position]; n_undo++;
        state[m->position] = m->mut_nuc;
    }
    while (sp > 0) {
        const int64_t j = stk_node[sp - 1];
        if (stk_next[sp - 1] < t->child_off[j + 1]) {
            const int64_t ch = t->children[stk_next[sp - 1]++];
            stk_node[sp] = ch; stk_next[sp] = t->child_off[ch]; stk_mark[sp] = n_undo; sp++;
            for (int64_t i = t->mut_off[ch]; i < t->mut_off[ch + 1]; i++) {
                const orc_mut *m = &t->muts[i];
                if (m->position < 0) continue;
                c->prev[i] = state[m->position] ? state[m->position] : c->ref_at[m->position];
                undo[n_undo].pos = m->position; undo[n_undo].old = state[m->position]; n_undo++;
                state[m->position] = m->mut_nuc;
            }
        } else {
            while (n_undo > stk_mark[sp - 1]) { n_undo--; state[undo[n_undo].pos] = undo[n_undo].old; }
            sp--;
        }
    }
    free(undo); free(stk_node); free(stk_next); free(stk_mark); free(state);
    return c;
}

/* One sample.  S = scratch [max_pos + 1], all zero on entry and on return (0 = "no row": reference base).
 * D = scratch int32 [n].  ties (optional): ascending BFS indices of the optimal nodes, at most cap. */
static int orc_cf_place_one(const orc_cf *c, int64_t n_ent, const int32_t *pos, const int8_t *ref, const int8_t *nuc,
                            const int8_t *is_missing, uint8_t *S, int32_t *D, int32_t *scores,
                            int32_t *out_best, int64_t *out_num_best, int64_t *out_best_j, int8_t *out_hu,
                            int64_t *ties, int8_t *ties_hu, int64_t cap) {
    const orc_tree *t = c->t;
    int32_t dbot = 0;
    for (int64_t i = 0; i < n_ent; i++) {
        if (i > 0 && pos[i] <= pos[i - 1]) return -1;
        const uint8_t a = is_missing[i] ? 15 : (uint8_t)nuc[i];
        if (!is_missing[i] && (a & (uint8_t)ref[i]) == 0) dbot++;
        if (pos[i] >= 0 && pos[i] <= c->max_pos) S[pos[i]] = a;
    }
    int32_t best = 0x7fffffff; int64_t nb = 0, bj = 0, bl = -1; int8_t bhu = 0;
    int64_t n_t = 0;
    for (int64_t j = 0; j < t->n; j++) {
        const int32_t dpar = j ? D[t->parent[j]] : dbot;
        int32_t tsum = 0, neg = 0, common = 0, num_mut = 0; int masked = 0;
        for (int64_t i = t->mut_off[j]; i < t->mut_off[j + 1]; i++) {
            const orc_mut *m = &t->muts[i];
            if (m->position < 0) { if (!masked) num_mut++; masked = 1; continue; }
            const uint8_t s = S[m->position] ? S[m->position] : (uint8_t)c->ref_at[m->position];
            const int cc = (s & (uint8_t)m->mut_nuc) != 0, pp = (s & (uint8_t)c->prev[i]) != 0;
            const int d = pp - cc;
            tsum += d;
            if (!masked) { num_mut++; common += cc; if (d < 0) neg += d; }
        }
        D[j] = dpar + tsum;
        int32_t cost; int elig, hu;
        if (j == 0) { cost = D[0]; elig = 1; hu = 0; }
        else {
            const int leaf = is_leaf(t, j);
            cost = dpar + neg;
            elig = common > 0 || (!leaf && num_mut == 0);
            hu = masked || common != num_mut;
        }
        if (scores) scores[j] = cost + (elig ? 0 : 1);
        if (!elig) continue;
        if (cost < best) {
            best = cost; nb = 1; bj = j; bl = t->num_leaves[j]; bhu = (int8_t)hu;
            if (ties && cap > 0) { ties[0] = j; if (ties_hu) ties_hu[0] = (int8_t)hu; }
            n_t = 1;
        }
        else if (cost == best) {
            if (ties && n_t < cap) { ties[n_t] = j; if (ties_hu) ties_hu[n_t] = (int8_t)hu; }
            n_t++; nb++;
            if (t->num_leaves[j] > bl || (t->num_leaves[j] == bl && j > bj)) { bj = j; bl = t->num_leaves[j]; bhu = (int8_t)hu; }
        }
    }
    for (int64_t i = 0; i < n_ent; i++) if (pos[i] >= 0 && pos[i] <= c->max_pos) S[pos[i]] = 0;
    *out_best = best; *out_num_best = nb; *out_best_j = bj; *out_hu = bhu;
    return 0;
}

typedef struct {
    const orc_cf *c; int64_t n_q; const int64_t *ent_off; const int32_t *pos; const int8_t *ref, *nuc, *mis;
    int32_t *best; int64_t *num_best, *best_j; int8_t *hu;
    int64_t *ties; int8_t *ties_hu; int64_t cap;
    int64_t *next; int rc;
} orc_cf_job;

static void *orc_cf_worker(void *p) {
    orc_cf_job *jb = (orc_cf_job *)p;
    uint8_t *S = (uint8_t *)calloc((size_t)jb->c->max_pos + 1, 1);
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * (size_t)jb->c->t->n);
    for (;;) {
        const int64_t q = __atomic_fetch_add(jb->next, 1, __ATOMIC_RELAXED);
        if (q >= jb->n_q) break;
        const int64_t b = jb->ent_off[q], e = jb->ent_off[q + 1];
        if (orc_cf_place_one(jb->c, e - b, jb->pos + b, jb->ref + b, jb->nuc + b, jb->mis + b, S, D, NULL,
                             &jb->best[q], &jb->num_best[q], &jb->best_j[q], &jb->hu[q],
                             jb->ties ? jb->ties + q * jb->cap : NULL, jb->ties_hu ? jb->ties_hu + q * jb->cap : NULL, jb->cap) != 0) {
            jb->rc = -1;
            memset(S, 0, (size_t)jb->c->max_pos + 1);
        }
    }
    free(S); free(D);
    return NULL;
}

/* A CSR batch of samples, sample-parallel over nthreads (each thread owns an S and a D array). */
int orc_cf_place_batch(const orc_cf *c, int64_t n_q, const int64_t *ent_off, const int32_t *pos, const int8_t *ref,
                       const int8_t *nuc, const int8_t *is_missing, int nthreads,
                       int32_t *best, int64_t *num_best, int64_t *best_j, int8_t *hu,
                       int64_t *ties, int8_t *ties_hu, int64_t cap) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n_q) nthreads = (int)(n_q > 0 ? n_q : 1);
    int64_t next = 0;
    orc_cf_job *jobs = (orc_cf_job *)calloc((size_t)nthreads, sizeof(orc_cf_job));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) {
        orc_cf_job jb = {c, n_q, ent_off, pos, ref, nuc, is_missing, best, num_best, best_j, hu, ties, ties_hu, cap, &next, 0};
        jobs[i] = jb;
        if (i + 1 < nthreads) pthread_create(&th[i], NULL, orc_cf_worker, &jobs[i]);
    }
    orc_cf_worker(&jobs[nthreads - 1]);
    int rc = jobs[nthreads - 1].rc;
    for (int i = 0; i + 1 < nthreads; i++) { pthread_join(th[i], NULL); if (jobs[i].rc) rc = jobs[i].rc; }
    free(th); free(jobs);
    return rc;
}

/* D(n, s) of every node for one sample (tuning / analysis aid for the pruning bounds). */
int orc_cf_D(const orc_cf *c, int64_t n_ent, const int32_t *pos, const int8_t *ref, const int8_t *nuc,
             const int8_t *is_missing, int32_t *D) {
    uint8_t *S = (uint8_t *)calloc((size_t)c->max_pos + 1, 1);
    int32_t b; int64_t nb, bj; int8_t hu;
    int rc = orc_cf_place_one(c, n_ent, pos, ref, nuc, is_missing, S, D, NULL, &b, &nb, &bj, &hu, NULL, NULL, 0);
    free(S);
    return rc;
}

/* Per-node scores of one sample (the -p column: cost, +1 when not eligible). */
int orc_cf_scores(const orc_cf *c, int64_t n_ent, const int32_t *pos, const int8_t *ref, const int8_t *nuc,
                  const int8_t *is_missing, int32_t *scores) {
    uint8_t *S = (uint8_t *)calloc((size_t)c->max_pos + 1, 1);
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->t->n);
    int32_t b; int64_t nb, bj; int8_t hu;
    int rc = orc_cf_place_one(c, n_ent, pos, ref, nuc, is_missing, S, D, scores, &b, &nb, &bj, &hu, NULL, NULL, 0);
    free(S); free(D);
    return rc;
}

/* --------------------------------------------------------- Fitch-Sankoff */

/*
 * Node::add_mutation, mutation_annotated_tree.cpp:720-752, on a small
 * per-node vector kept sorted by position.  Returns -1 on the reference's
 * "called out of order" error.
 */
static int add_mutation(mvec *v, orc_mut mut) {
    int64_t it = 0;
    while (it < v->n && v->v[it].position < mut.position) it++;      /* lower_bound */
    if (it < v->n && v->v[it].position == mut.position) {
        if (v->v[it].par_nuc != mut.mut_nuc) {
            v->v[it].mut_nuc = mut.mut_nuc;
        } else {
            if (v->v[it].mut_nuc != mut.par_nuc) return -1;
            int32_t p = v->v[it].position; int64_t w = 0;
            for (int64_t i = 0; i < v->n; i++) if (v->v[i].position != p) v->v[w++] = v->v[i];
            v->n = w;
        }
    } else {
        mv_push(v, mut);
        for (int64_t i = v->n - 1; i > it; i--) v->v[i] = v->v[i - 1];
        v->v[it] = mut;
    }
    return 0;
}

static int8_t get_nt(int8_t nuc_id) {   /* mutation_annotated_tree.cpp:142-162 */
    switch (nuc_id) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: return -1; }
}

/*
 * mapper_body::operator(), usher_mapper.cpp:6-161, for ONE VCF site on a tree
 * given by BFS arrays (no mutations needed).  variants: n_var pairs
 * (bfs index of a tree sample, allele mask); samples absent from the tree are
 * not passed (the caller keeps them as missing samples).  Emits, for every
 * node whose parsimony state differs from its parent's, par/mut one-hot
 * alleles: out_state[j] in 0..3, and out_mut_par[j]/out_mut_nuc[j] (0 when the
 * node carries no mutation at this site).
 */
int orc_fitch_site(int64_t n, const int64_t *parent, int8_t ref_nuc,
                   int64_t n_var, const int64_t *var_node, const int8_t *var_nuc,
                   int8_t *out_state, int8_t *out_mut_par, int8_t *out_mut_nuc) {
    int *scores = (int *)calloc((size_t)n * 4, sizeof(int));       /* :21-31 */
    int8_t *states = (int8_t *)calloc((size_t)n, 1);
    int64_t *nchild = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    for (int64_t j = 1; j < n; j++) nchild[parent[j]]++;
    int8_t ref_id = get_nt(ref_nuc);
    if (ref_id < 0) { free(scores); free(states); free(nchild); return -1; }
    for (int64_t j = 0; j < n; j++)                                 /* :35-44 leaves */
        if (nchild[j] == 0)
            for (int b = 0; b < 4; b++) if (b != ref_id) scores[j * 4 + b] = (int)n;
    for (int64_t v = 0; v < n_var; v++) {                           /* :47-62 */
        int64_t idx = var_node[v];
        for (int b = 0; b < 4; b++) {
            scores[idx * 4 + b] = (int)n;
            if (((1 << b) & var_nuc[v]) != 0) scores[idx * 4 + b] = 0;
        }
    }
    for (int64_t j = n - 1; j >= 0; j--) {                          /* :86-111 forward pass */
        /* the reference adds each child's contribution while visiting the
         * parent in reverse BFS order; adding it while visiting the child in
         * reverse BFS order performs the same integer sums. */
        if (j == 0) break;
        int64_t p = parent[j];
        for (int b = 0; b < 4; b++) {
            int min_s = (int)n + 1;
            for (int k = 0; k < 4; k++) {
                int c_s = (k == b) ? scores[j * 4 + k] : scores[j * 4 + k] + 1;
                if (c_s < min_s) min_s = c_s;
            }
            scores[p * 4 + b] += min_s;
        }
    }
    for (int64_t j = 0; j < n; j++) {                               /* :114-156 backward pass */
        int8_t par_state = (j == 0) ? ref_id : states[parent[j]];
        int8_t state = par_state;
        int min_s = scores[j * 4 + par_state];
        for (int b = 0; b < 4; b++)
            if (scores[j * 4 + b] < min_s) { min_s = scores[j * 4 + b]; state = (int8_t)b; }
        if (state != par_state && scores[j * 4 + par_state] == min_s) state = par_state;
        states[j] = state;
        out_state[j] = state;
        if (state != par_state) { out_mut_par[j] = (int8_t)(1 << par_state); out_mut_nuc[j] = (int8_t)(1 << state); }
        else { out_mut_par[j] = 0; out_mut_nuc[j] = 0; }
    }
    free(scores); free(states); free(nchild);
    return 0;
}

/*
 * Thin wrapper so python tests can exercise add_mutation on a list: applies
 * `k` mutations in order to an initially empty node and writes the result.
 */
int orc_add_mutations(int64_t k, const int32_t *pos, const int8_t *ref, const int8_t *par,
                      const int8_t *mut, int64_t cap, int32_t *o_pos, int8_t *o_ref,
                      int8_t *o_par, int8_t *o_mut, int64_t *o_n) {
    mvec v = {0, 0, 0};
    for (int64_t i = 0; i < k; i++) {
        orc_mut m; m.position = pos[i]; m.ref_nuc = ref[i]; m.par_nuc = par[i]; m.mut_nuc = mut[i]; m.is_missing = 0;
        if (add_mutation(&v, m) != 0) { free(v.v); return -1; }
    }
    for (int64_t i = 0; i < v.n && i < cap; i++) {
        o_pos[i] = v.v[i].position; o_ref[i] = v.v[i].ref_nuc; o_par[i] = v.v[i].par_nuc; o_mut[i] = v.v[i].mut_nuc;
    }
    *o_n = v.n;
    free(v.v);
    return 0;
}
