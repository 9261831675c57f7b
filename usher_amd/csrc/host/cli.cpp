// cli.cpp -- command line of the usher-compatible front end: the 22 flags of the
// reference's src/usher.cpp:47-86 (+ --version / --help), tree / MAT / VCF
// loading (usher.cpp:132-173) and the call into the driver (usher.cpp:175-178).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "driver.hpp"

#define UH_VERSION "0.7.0-amd"

namespace uh {

bool g_leak_tree_at_exit = false;   // set by bin/usher-amd's main()

static const char *kHelp =
    "Options:\n"
    "  -v [ --vcf ] arg                          Input VCF file (in uncompressed or gzip-compressed .gz format) [REQUIRED]\n"
    "  -t [ --tree ] arg                         Input tree file\n"
    "  -d [ --outdir ] arg (=.)                  Output directory to dump output and log files [DEFAULT uses current directory]\n"
    "  -i [ --load-mutation-annotated-tree ] arg Load mutation-annotated tree object\n"
    "  -o [ --save-mutation-annotated-tree ] arg Save output mutation-annotated tree object to the specified filename\n"
    "  -s [ --sort-before-placement-1 ]          Sort new samples based on computed parsimony score and then number of optimal placements before the actual placement [EXPERIMENTAL].\n"
    "  -S [ --sort-before-placement-2 ]          Sort new samples based on the number of optimal placements and then the parsimony score before the actual placement [EXPERIMENTAL].\n"
    "  -A [ --sort-before-placement-3 ]          Sort new samples based on the number of ambiguous bases [EXPERIMENTAL].\n"
    "  -r [ --reverse-sort ]                     Reverse the sorting order of sorting options [EXPERIMENTAL]\n"
    "  -c [ --collapse-tree ]                    Collapse internal nodes of the input tree with no mutations and condense identical sequences in polytomies into a single node and the save the tree to file condensed-tree.nh in outdir\n"
    "  -C [ --collapse-output-tree ]             Collapse internal nodes of the output tree with no mutations before the saving the tree to file final-tree.nh in outdir\n"
    "  -e [ --max-uncertainty-per-sample ] arg (=1000000) Maximum number of equally parsimonious placements allowed per sample beyond which the sample is ignored\n"
    "  -E [ --max-parsimony-per-sample ] arg (=1000000)   Maximum parsimony score of the most parsimonious placement(s) allowed per sample beyond which the sample is ignored\n"
    "  -u [ --write-uncondensed-final-tree ]     Write the final tree in uncondensed format and save to file uncondensed-final-tree.nh in outdir\n"
    "  -k [ --write-subtrees-size ] arg (=0)     Write minimum set of subtrees covering the newly added samples of size equal to this value\n"
    "  -K [ --write-single-subtree ] arg (=0)    Similar to write-subtrees-size but produces a single subtree with all newly added samples along with random samples up to the value specified by this argument\n"
    "  -p [ --write-parsimony-scores-per-node ]  Write the parsimony scores for adding new samples at each existing node in the tree without modifying the tree in a file names parsimony-scores.tsv in outdir\n"
    "  -M [ --multiple-placements ] arg (=1)     Create a new tree up to this limit for each possibility of parsimony-optimal placement\n"
    "  -l [ --retain-input-branch-lengths ]      Retain the branch lengths from the input tree in out newick files instead of using number of mutations for the branch lengths.\n"
    "  -n [ --no-add ]                           Do not add new samples to the tree\n"
    "  -D [ --detailed-clades ]                  In clades.txt, write a histogram of annotated clades and counts across all equally parsimonious placements\n"
    "  -T [ --threads ] arg                      Number of host threads (the node x sample search runs on the GPU)\n"
    "  --device arg (=0)                         HIP device ordinal\n"
    "  --devices arg                             HIP devices to shard the samples across, e.g. 0-7 or 0,2,3 (the tree is replicated)\n"
    "  --version                                 Print version number\n"
    "  -h [ --help ]                             Print help messages\n";

// Returns 0 to continue, 1 = exit with error, 2 = exit(0) (help / version).
static int parse(int argc, char **argv, Options &o) {
    struct Spec { char s; const char *l; int kind; };   // kind 0 flag, 1 string/number value
    static const Spec specs[] = {
        {'v', "vcf", 1}, {'t', "tree", 1}, {'d', "outdir", 1}, {'i', "load-mutation-annotated-tree", 1},
        {'o', "save-mutation-annotated-tree", 1}, {'s', "sort-before-placement-1", 0}, {'S', "sort-before-placement-2", 0},
        {'A', "sort-before-placement-3", 0}, {'r', "reverse-sort", 0}, {'c', "collapse-tree", 0}, {'C', "collapse-output-tree", 0},
        {'e', "max-uncertainty-per-sample", 1}, {'E', "max-parsimony-per-sample", 1}, {'u', "write-uncondensed-final-tree", 0},
        {'k', "write-subtrees-size", 1}, {'K', "write-single-subtree", 1}, {'p', "write-parsimony-scores-per-node", 0},
        {'M', "multiple-placements", 1}, {'l', "retain-input-branch-lengths", 0}, {'n', "no-add", 0}, {'D', "detailed-clades", 0},
        {'T', "threads", 1}, {0, "device", 1}, {0, "devices", 1}, {0, "version", 0}, {'h', "help", 0}};
    bool version = false, help = false, bad = false;
    auto apply = [&](const Spec &sp, const char *val) {
        const std::string l = sp.l;
        auto num = [&](const char *v) { return strtoull(v, nullptr, 10); };
        if (l == "vcf") o.vcf = val; else if (l == "tree") o.tree = val; else if (l == "outdir") o.outdir = val;
        else if (l == "load-mutation-annotated-tree") o.load_mat = val; else if (l == "save-mutation-annotated-tree") o.save_mat = val;
        else if (l == "sort-before-placement-1") o.sort1 = true; else if (l == "sort-before-placement-2") o.sort2 = true;
        else if (l == "sort-before-placement-3") o.sort3 = true; else if (l == "reverse-sort") o.reverse_sort = true;
        else if (l == "collapse-tree") o.collapse_tree = true; else if (l == "collapse-output-tree") o.collapse_output_tree = true;
        else if (l == "max-uncertainty-per-sample") o.max_uncertainty = (uint32_t)num(val);
        else if (l == "max-parsimony-per-sample") o.max_parsimony = (uint32_t)num(val);
        else if (l == "write-uncondensed-final-tree") o.write_uncondensed = true;
        else if (l == "write-subtrees-size") o.subtrees_size = num(val); else if (l == "write-single-subtree") o.subtrees_single = num(val);
        else if (l == "write-parsimony-scores-per-node") o.print_scores = true; else if (l == "multiple-placements") o.max_trees = (uint32_t)num(val);
        else if (l == "retain-input-branch-lengths") o.retain_branch_len = true; else if (l == "no-add") o.no_add = true;
        else if (l == "detailed-clades") o.detailed_clades = true; else if (l == "threads") o.threads = (uint32_t)num(val);
        else if (l == "device") o.device = (int)num(val); else if (l == "devices") o.devices = val; else if (l == "version") version = true; else if (l == "help") help = true;
    };
    for (int i = 1; i < argc && !bad; i++) {
        const std::string a = argv[i];
        const Spec *sp = nullptr;
        const char *val = nullptr;
        std::string inline_val;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2);
            size_t eq = name.find('=');
            if (eq != std::string::npos) { inline_val = name.substr(eq + 1); name = name.substr(0, eq); val = inline_val.c_str(); }
            for (const Spec &s : specs) if (name == s.l) sp = &s;
        } else if (a.size() >= 2 && a[0] == '-') {
            for (const Spec &s : specs) if (s.s && a[1] == s.s) sp = &s;
            if (sp && a.size() > 2) {
                if (sp->kind == 1) { inline_val = a.substr(2); val = inline_val.c_str(); }
                else {   // stuck switches, e.g. -un
                    for (size_t k = 1; k < a.size(); k++) {
                        const Spec *f = nullptr;
                        for (const Spec &s : specs) if (s.s && a[k] == s.s && s.kind == 0) f = &s;
                        if (!f) { bad = true; break; }
                        apply(*f, nullptr);
                    }
                    continue;
                }
            }
        }
        if (!sp) { bad = true; break; }
        if (sp->kind == 1 && !val) {
            if (i + 1 >= argc) { bad = true; break; }
            val = argv[++i];
        }
        apply(*sp, val);
    }
    if (version) { printf("UShER (v%s)\n", UH_VERSION); if (o.vcf.empty() || bad || help) return 2; }
    if (help || bad || o.vcf.empty()) {   // usher.cpp:92-107
        if (!version) { fprintf(stderr, "UShER (v%s)\n", UH_VERSION); fprintf(stderr, "%s\n", kHelp); }
        return (help || version) ? 2 : 1;
    }
    return 0;
}

static bool assign_via_backend(void *ctx, const SiteBatch &in, SiteMutations &out, std::string &err) {
    const Backend &be = *(const Backend *)ctx;
    if (!be.fitch || !be.fitch_get) { err = "ERROR: the placement backend has no Fitch-Sankoff entry point."; return false; }
    ugp_sites sites{};
    sites.n_sites = in.ref.size(); sites.ref = in.ref.data(); sites.var_off = in.var_off.data();
    sites.var_node = in.var_node.data(); sites.var_nuc = in.var_nuc.data();
    uint64_t n = 0;
    auto why = [&]() { return std::string(be.last_error ? be.last_error(be.ctx) : "?"); };
    if (be.fitch(be.ctx, in.parent.size(), in.parent.data(), &sites, &n) != 0) { err = "ERROR: Fitch-Sankoff backend failed: " + why(); return false; }
    out.site.resize(n); out.node.resize(n); out.par_nuc.resize(n); out.mut_nuc.resize(n);
    if (be.fitch_get(be.ctx, out.site.data(), out.node.data(), out.par_nuc.data(), out.mut_nuc.data()) != 0) { err = "ERROR: Fitch-Sankoff backend failed: " + why(); return false; }
    return true;
}

int usher_main(int argc, char **argv, const Backend &be) {
    Options opt;
    int pr = parse(argc, argv, opt);
    if (pr == 2) return 0;
    if (pr == 1) return 1;
    if (be.warm) be.warm(be.ctx);   // (the device runtime starts while the inputs are read)
    // (in the CLI process the tree of a run lives until the process ends: tearing down 10M nodes one by one before exit buys nothing;
    // callers that run many trees in one process -- the tests' C entry -- leave the flag off)
    Tree *Tp = new Tree();
    struct Drop { Tree *t; ~Drop() { if (!g_leak_tree_at_exit) delete t; } } drop{Tp};
    Tree &T = *Tp;
    std::vector<MissingSample> missing;
    std::string err;
    Prebuilt *pre = nullptr;
    if (!opt.tree.empty()) {                                                    // usher.cpp:132-149
        fprintf(stderr, "Loading input tree.\n");
        FILE *f = fopen(opt.tree.c_str(), "r");
        if (!f) { fprintf(stderr, "ERROR: Could not open the tree file: %s!\n", opt.tree.c_str()); return 1; }
        std::string nwk;
        int c;
        while ((c = fgetc(f)) != EOF && c != '\n') nwk += (char)c;
        fclose(f);
        if (!tree_from_newick(nwk, T, err)) { fprintf(stderr, "ERROR: %s!\n", err.c_str()); return 1; }
        if (!T.root) { fprintf(stderr, "ERROR: Empty tree.\n"); return 1; }
        fprintf(stderr, "Loading VCF file.\nComputing parsimonious assignments for input variants.\n");
        if (!read_vcf_build(T, opt.vcf, missing, err, assign_via_backend, (void *)&be)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else if (!opt.load_mat.empty()) {                                         // usher.cpp:151-170
        fprintf(stderr, "Loading existing mutation-annotated tree object from file %s\n", opt.load_mat.c_str());
        const auto t0 = std::chrono::steady_clock::now();
        if (!load_mat(opt.load_mat, T, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!T.root) { fprintf(stderr, "ERROR: Empty tree.\n"); return 1; }
        const auto t1 = std::chrono::steady_clock::now();
        pre = prebuild_start(opt, T, be);   // tree -> arrays -> device, under the VCF read
        fprintf(stderr, "Loading VCF file\n");
        if (!read_vcf_missing(T, opt.vcf, missing, err)) { prebuild_drop(pre); fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (getenv("USHER_AMD_PROFILE"))
            fprintf(stderr, "[usher-amd profile] load MAT %.3f s, read VCF %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    } else {
        fprintf(stderr, "Error! No input tree or assignment file provided!\n");
        return 1;
    }
    const auto t2 = std::chrono::steady_clock::now();
    const int rc = run_usher(opt, T, missing, be, pre);
    if (getenv("USHER_AMD_PROFILE")) fprintf(stderr, "[usher-amd profile] run_usher (placement loop + output files) %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count());
    return rc;
}

}  // namespace uh

// C entry for tests: the same front end with a caller-supplied placement backend.
extern "C" int uh_usher_main(int argc, char **argv, const uh::Backend *be) {
    if (!be || !be->place) { fprintf(stderr, "ERROR: no placement backend\n"); return 1; }
    return uh::usher_main(argc, argv, *be);
}

// Bench / test utilities (no GPU): write the synthetic workload as the files the front end reads, and time the two loaders.
extern "C" int uh_write_pb_arrays(uint64_t n, const uint32_t *parent, const uint64_t *mut_off, const int32_t *pos, const uint8_t *ref, const uint8_t *par,
                                  const uint8_t *nuc, const char *path) {
    std::string err;
    if (!uh::write_pb_from_arrays(n, parent, mut_off, pos, ref, par, nuc, path, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
}
extern "C" int uh_write_vcf_csr(uint64_t n_samples, const uint64_t *ent_off, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                                const char *prefix, const char *path) {
    std::string err;
    if (!uh::write_vcf_from_csr(n_samples, ent_off, pos, ref, nuc, is_missing, prefix, path, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
}
// out[0] = seconds in load_mat, out[1] = seconds in read_vcf_missing, out[2] = nodes, out[3] = samples, out[4] = sample rows, out[5] = tree mutations
extern "C" int uh_time_load(const char *pb, const char *vcf, double *out) {
    uh::Tree T;
    std::vector<uh::MissingSample> missing;
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    if (!uh::load_mat(pb, T, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const auto t1 = std::chrono::steady_clock::now();
    if (vcf && *vcf && !uh::read_vcf_missing(T, vcf, missing, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const auto t2 = std::chrono::steady_clock::now();
    out[0] = std::chrono::duration<double>(t1 - t0).count();
    out[1] = std::chrono::duration<double>(t2 - t1).count();
    out[2] = (double)T.all_nodes.size();
    out[3] = (double)missing.size();
    size_t rows = 0;
    for (auto &m : missing) rows += m.mutations.size();
    out[4] = (double)rows;
    out[5] = (double)T.parsimony_score();
    return 0;
}

// Test utility: a parsimony.proto file as the breadth-first arrays of ugp_tree_desc (what the oracle takes).  Call with null arrays
// for the sizes (counts[0] = nodes, counts[1] = mutations), then with arrays of those sizes; name_of_leaf (optional, n x 24 bytes)
// receives the first 23 characters of every node's name.
extern "C" int uh_pb_to_arrays(const char *pb, uint64_t *counts, int64_t *parent, int64_t *mut_off, int32_t *pos, int8_t *ref, int8_t *par, int8_t *nuc,
                               char *names24) {
    uh::Tree T;
    std::string err;
    if (!uh::load_mat(pb, T, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    if (!T.condensed_nodes.empty()) T.uncondense_leaves();   // (a MAT is saved with identical sequences condensed; the search runs on the full tree)
    const std::vector<uh::Node *> bfs = T.bfs();
    uint64_t m = 0;
    for (const uh::Node *n : bfs) m += n->mutations.size();
    counts[0] = bfs.size(); counts[1] = m;
    if (!parent) return 0;
    for (size_t j = 0; j < bfs.size(); j++) { bfs[j]->flat_index = (uint32_t)j; }
    uint64_t k = 0;
    for (size_t j = 0; j < bfs.size(); j++) {
        const uh::Node *n = bfs[j];
        parent[j] = n->parent ? (int64_t)n->parent->flat_index : -1;
        mut_off[j] = (int64_t)k;
        for (const uh::Mutation &mu : n->mutations) { pos[k] = mu.position; ref[k] = mu.ref_nuc; par[k] = mu.par_nuc; nuc[k] = mu.mut_nuc; k++; }
        if (names24) { snprintf(names24 + j * 24, 24, "%s", n->id.c_str()); }
    }
    mut_off[bfs.size()] = (int64_t)k;
    return 0;
}

// Test hook: parse a newick string with the general routine (bulk = 0) or the bulk one (bulk = 1) and describe the tree --
// one line per node in depth-first order: id, parent id, level, branch length, number of children.  Returns the length
// needed (the text is truncated to cap - 1), or -1 with the error message in `out`.
extern "C" long uh_newick_digest(const char *nwk, int bulk, char *out, size_t cap) {
    uh::Tree T;
    std::string err, text;
    const std::string s(nwk);
    const bool ok = bulk ? uh::tree_from_newick_bulk(s.data(), s.size(), T, err) : uh::tree_from_newick(s, T, err);
    if (!ok) { snprintf(out, cap, "%s", err.c_str()); return -1; }
    char line[256];
    for (uh::Node *n : T.dfs()) {
        snprintf(line, sizeof line, "|%zu|%g|%zu\n", n->level, (double)n->branch_length, n->children.size());
        text += n->id; text += '|'; text += n->parent ? n->parent->id : std::string("-"); text += line;
    }
    text += "internal=" + std::to_string(T.curr_internal_node) + " nodes=" + std::to_string(T.all_nodes.size()) + "\n";
    for (uh::Node *n : T.dfs()) if (T.get_node(n->id) != n) text += "INDEX MISMATCH " + n->id + "\n";
    snprintf(out, cap, "%s", text.c_str());
    return (long)text.size();
}

// Test utility (round 6, tools/pin_pb_with_reference.py): what load_mat() decoded from a parsimony.proto file, as JSON in the
// shape of the message itself (parsimony.proto:1-31) -- newick as save_mat() would write it again, per node in depth-first
// preorder (the order of data.node_mutations, mutation_annotated_tree.cpp:553-557) the mutations
// [position, ref_nuc, par_nuc, [mut_nuc...], chromosome] with the nucleotides as the file's 0..3 indices (-1, -1, [] for a masked
// mutation, :632-634) and the clade annotations, then the condensed nodes in file order.  The leaves are NOT uncondensed.  The
// pinned dumps under tests/golden/pb_pinned/ come from the reference's own generated module (parsimony_pb2.py) reading the same bytes.
static void json_str(std::string &o, const std::string &s) {
    o += '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    o += '"';
}
extern "C" int uh_pb_dump(const char *pb, const char *out_path) {
    uh::Tree T;
    std::string err;
    if (!uh::load_mat(pb, T, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::string o = "{\"newick\": ";
    json_str(o, uh::newick(T, T.root, false, true));
    o += ", \"node_mutations\": [";
    const std::vector<uh::Node *> dfs = T.dfs();
    auto idx = [](int8_t onehot) { return onehot == 1 ? 0 : onehot == 2 ? 1 : onehot == 4 ? 2 : onehot == 8 ? 3 : -1; };
    for (size_t i = 0; i < dfs.size(); i++) {
        o += i ? ", [" : "[";
        bool first = true;
        for (const uh::Mutation &m : dfs[i]->mutations) {
            if (!first) o += ", ";
            first = false;
            o += "[" + std::to_string(m.position) + ", ";
            if (m.masked()) o += "-1, -1, []";
            else {
                o += std::to_string(idx(m.ref_nuc)) + ", " + std::to_string(idx(m.par_nuc)) + ", [";
                bool f2 = true;
                for (int b = 0; b < 4; b++) if (m.mut_nuc & (1 << b)) { if (!f2) o += ", "; f2 = false; o += std::to_string(b); }
                o += "]";
            }
            o += ", ";
            json_str(o, T.chroms[m.chrom]);
            o += "]";
        }
        o += "]";
    }
    o += "], \"metadata\": [";
    for (size_t i = 0; i < dfs.size(); i++) {
        o += i ? ", [" : "[";
        for (size_t k = 0; k < dfs[i]->clade_annotations.size(); k++) { if (k) o += ", "; json_str(o, dfs[i]->clade_annotations[k]); }
        o += "]";
    }
    o += "], \"condensed_nodes\": [";
    bool first = true;
    for (const std::string &name : T.condensed_order) {
        if (!first) o += ", ";
        first = false;
        o += "[";
        json_str(o, name);
        o += ", [";
        const auto &ids = T.condensed_nodes.at(name);
        for (size_t k = 0; k < ids.size(); k++) { if (k) o += ", "; json_str(o, ids[k]); }
        o += "]]";
    }
    o += "]}\n";
    FILE *f = fopen(out_path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    return (fclose(f) == 0 && ok) ? 0 : 1;
}
// load_mat() + save_mat(): a file written again from what was read (no uncondense / condense in between)
extern "C" int uh_pb_resave(const char *pb, const char *out_path) {
    uh::Tree T;
    std::string err;
    if (!uh::load_mat(pb, T, err) || !uh::save_mat(T, out_path, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
}
