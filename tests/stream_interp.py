"""Test helper: a python interpreter of the flattened DFS record stream.

It executes exactly the per-record algorithm of k_place (usher_amd/csrc/
ugp_kernels.hip) -- read slot / register / D_bottom, per-mutation delta from the
sample's allele nibble, write slot, score, (min, count, max key) reduction,
preamble replay per group -- so the host flattening (slots, preambles, chunk
cuts, true parent alleles, tie ranks) can be checked on CPU against the oracle
before any GPU run.  Test infrastructure only; never imported by usher_amd.
"""
from __future__ import annotations

import numpy as np

RS_REG, RS_BOTTOM, WS_NONE = 63, 62, 63
F_LEAF, F_NOSCORE, F_ROOT, F_MASKED = 1 << 28, 1 << 29, 1 << 30, 1 << 31
M_AFTER_MASK = 1 << 31
IDX2HOT = [1, 2, 4, 8]


def sample_site_alleles(flat, sample):
    """nibble per site (reference base unless the sample has a row) and D_bottom."""
    nib = flat.site_ref.astype(np.int64).copy()
    dbot = 0
    for p, r, a, mis in zip(sample["pos"], sample["ref"], sample["nuc"], sample["is_missing"]):
        a = 15 if mis else int(a)
        if not mis and (a & int(r)) == 0:
            dbot += 1
        if 0 <= p < len(flat.pos2site) and flat.pos2site[p] >= 0:
            nib[flat.pos2site[p]] = a
    return nib, dbot


def variant_rows(sample):
    """V of the second pruning bound: rows whose allele set is neither missing nor just the reference base."""
    return sum(1 for r, a, mis in zip(sample["ref"], sample["nuc"], sample["is_missing"]) if not mis and int(a) != int(r))


def run_group(flat, nib, dbot, c0, c1, want_scores=None):
    """One wave's work: preamble of chunk c0, bodies of chunks [c0, c1)."""
    slots = {}
    best, cnt, key_best = 0x7FFFFFFF, 0, 0
    dcur = 0
    node_idx = int(flat.chunk_node_off[c0])
    for words, lo, hi in ((flat.pre_stream, int(flat.chunk_pre_off[c0]), int(flat.chunk_pre_off[c0 + 1])),
                          (flat.stream, int(flat.chunk_body_off[c0]), int(flat.chunk_body_off[c1]))):
        i = lo
        while i < hi:
            w0 = int(words[i]); key = int(words[i + 1]); i += 2
            nmut = w0 & 0xFFFF
            rslot, wslot = (w0 >> 16) & 63, (w0 >> 22) & 63
            if rslot == RS_REG:
                dpar = dcur
            elif rslot == RS_BOTTOM:
                dpar = dbot
            else:
                dpar = slots[rslot]
            tsum = neg = common = n_before = 0
            for _ in range(nmut):
                w = int(words[i]); i += 1
                site, mi, pi = w & 0x3FFFFF, (w >> 22) & 3, (w >> 24) & 3
                x = int(nib[site])
                c, p = (x >> mi) & 1, (x >> pi) & 1
                d = p - c
                tsum += d
                if not (w & M_AFTER_MASK):
                    n_before += 1
                    common += c
                    neg += min(d, 0)
            dn = dpar + tsum
            if wslot != WS_NONE:
                slots[wslot] = dn
            dcur = dn
            if not (w0 & F_NOSCORE):
                if w0 & F_ROOT:
                    cost, elig, hu = dn, True, 0
                else:
                    cost = dpar + neg
                    masked = bool(w0 & F_MASKED)
                    elig = common > 0 or (not (w0 & F_LEAF) and not masked and nmut == 0)
                    hu = 1 if (masked or common != n_before) else 0
                if want_scores is not None:
                    want_scores[int(flat.dfs2bfs[node_idx])] = cost + (0 if elig else 1)
                node_idx += 1
                if elig:
                    k = key | hu
                    if cost < best:
                        best, cnt, key_best = cost, 1, k
                    elif cost == best:
                        cnt += 1
                        key_best = max(key_best, k)
    return best, cnt, key_best


def place(flat, sample, n_groups=1, want_scores=False):
    nib, dbot = sample_site_alleles(flat, sample)
    n_chunks = len(flat.chunk_body_off) - 1
    n_groups = max(1, min(n_groups, n_chunks))
    scores = np.zeros(len(flat.dfs2bfs), dtype=np.int64) if want_scores else None
    best, cnt, key = 0x7FFFFFFF, 0, 0
    for g in range(n_groups):
        c0, c1 = g * n_chunks // n_groups, (g + 1) * n_chunks // n_groups
        if c0 >= c1:
            continue
        b, c, k = run_group(flat, nib, dbot, c0, c1, scores)
        if c == 0:
            continue
        if b < best:
            best, cnt, key = b, c, k
        elif b == best:
            cnt += c
            key = max(key, k)
    out = {"best": best, "num_best": cnt, "best_j": int(flat.rank2bfs[key >> 1]), "has_unique": bool(key & 1)}
    if want_scores:
        out["scores"] = scores
    return out


# --------------------------------------------------------------------------
# packed stream ("stream8"): interpreter of k_best8 + phase 2, one sample at a
# time but with the kernel's 16-bit wrap-around arithmetic and 4-bit counters.
# --------------------------------------------------------------------------
# layout of ugp_flatten.hpp (final encoding)
H_TAG, H_INFO, H_RARE, H_SIB = 1 << 31, 1 << 30, 1 << 29, 1 << 21
H_REG, H_STORE, H_NOSCORE8, H_END, H_FREE, H_SKIPD, _H_UNUSED6, H_SLOW, H_CHUNK_END, H_NOP = (1 << k for k in range(10))
H_RSLOT_SHIFT, H_WSLOT_SHIFT, INFO_HS_SHIFT = 10, 20, 22
CE_LEN_SHIFT, CE_LEN_MASK = 10, (1 << 19) - 1
INFO_HR_SHIFT, INFO_HR_NONE, INFO_JUMP_MASK, PRE_HS_NONE = 18, 7, (1 << 18) - 1, 127


def far(d, b, rec, ub, pre=False):
    """The kernel's all_far for one sample: D - hs > ub, or B - hr > ub (second bound, when the record has hr; B = the part
    of D at sites where the sample's set holds the reference base, one byte in the kernel).
    pre: a preamble record, whose hs field may say "not available"."""
    hs, hr = (rec >> INFO_HS_SHIFT) & 0x7F, (rec >> INFO_HR_SHIFT) & 7
    if not (pre and hs == PRE_HS_NONE) and d >= ub + 1 + hs:
        return True
    if hr == INFO_HR_NONE:
        return False
    assert 0 <= b <= 255
    return b >= ub + 1 + hr
M_FLUSH, M_END = 1 << 28, 1 << 30
U16 = 0xFFFF


B3_BLOCK_SHIFT = 4
B3_GROUP_SHIFT = 8


def b3_tables(flat, nibs):
    """The per-tile tables of the third pruning bound as ugp_bound3.hip builds them, for a "tile" = the samples whose site nibbles are
    `nibs`: a (site, allele) pair is useful when some sample's set holds the allele and not the reference base; the events of the
    useful pairs (the tree's posting lists) raise, per block of 16 packed-stream words,
        over[b]  = events whose block range contains b,      under[b] = events whose range contains b strictly inside."""
    n_sites = len(flat.site_ref)
    useful = np.zeros(n_sites, np.int64)
    for nib in nibs:
        excl = (nib & flat.site_ref.astype(np.int64)) == 0
        useful[excl] |= nib[excl]
    ng = (len(flat.b3_group_off) // 4) - 1
    nb = ng << B3_GROUP_SHIFT
    assert nb >= ((len(flat.stream8) + 15) >> B3_BLOCK_SHIFT) + 1
    off = flat.b3_group_off.astype(np.int64).reshape(4, ng + 1)
    ev = flat.b3_events.astype(np.int64)
    cnt = np.zeros((3, nb), np.int64)       # 0: inside one block, 1: range starts, 2: range ends
    for k in range(3):
        e = ev[off[k, 0]:off[k, ng]]
        grp = np.repeat(np.arange(ng), np.diff(off[k]))
        pair = e & 0xFFFFFF
        use = ((useful[pair >> 2] >> (pair & 3)) & 1) != 0
        np.add.at(cnt[k], (grp[use] << B3_GROUP_SHIFT) + (e[use] >> 24), 1)
    same, start, end = cnt[0], cnt[1], cnt[2]
    S, E = np.cumsum(start), np.cumsum(end)
    Sprev = np.concatenate([[0], S[:-1]]); Eprev = np.concatenate([[0], E[:-1]])
    # list 3: the events open at each group's first block -- what the device uses in place of a scan across the groups
    e3 = ev[off[3, 0]:off[3, ng]]
    use3 = ((useful[(e3 & 0xFFFFFF) >> 2] >> (e3 & 3)) & 1) != 0
    open0 = np.bincount(np.repeat(np.arange(ng), np.diff(off[3]))[use3], minlength=ng)
    first = np.arange(ng) << B3_GROUP_SHIFT
    assert (open0 == (Sprev - Eprev)[first]).all()
    over = S - Eprev + same
    under = Sprev - E
    assert (under >= 0).all()
    return {"over": over, "under": under, "useful": useful}


def b3_hu(b3, P, J):
    """Upper bound of the useful events on any path below the node whose last word sits at stream position P, its descendants in
    the J words behind it -- the exact maximum over the blocks (the kernel reads it through its 64-ary levels)."""
    q0, q1 = (P + 1) >> B3_BLOCK_SHIFT, (P + J) >> B3_BLOCK_SHIFT
    m = int(b3["over"][q0:q1 + 1].max())
    u = int(b3["under"][P >> B3_BLOCK_SHIFT])
    if m >= 255:              # (the device keeps one byte per block since round 6: 255 = "255 or more" = no bound)
        return 1 << 20
    return max(0, m - u)


def best8_group(flat, nib, dbot, c0, c1, ub=None, stats=None, use_pre_records=True, b3=None):
    """Chunk-local minima of one sample for chunks [c0, c1), as k_best8 computes them.
    ub: None = no pruning; otherwise a one-element list holding an upper bound of the sample's best
    score, used (and tightened at chunk ends) exactly like the kernel's shared bound."""
    slots = {}
    lbest = {}
    best = U16
    dcur, dpar = dbot, 0        # (the root reads "the previous node's D": every unit starts from D(bottom))
    bcur, bpar = 0, 0           # B(bottom) = 0: every state is the reference base there
    accP = accC = accN = accPB = accCB = 0
    carryD = carryN = carryC = carryB = 0
    flushed = False
    hdr = 0
    chunk = c0

    def finish():
        nonlocal best, dcur, bcur, accP, accC, accN, accPB, accCB, flushed, carryD, carryN, carryC, carryB
        assert accP <= 15 and accC <= 15 and accN <= 15
        # (the kernel computes D(node) for H_SKIPD nodes too; nothing reads it)
        dn = (dpar + accP - accC + (carryD if flushed else 0)) & U16
        dcur = dn
        bcur = bpar + accPB - accCB + (carryB if flushed else 0)   # (a byte in the kernel; only read where a record offers the second bound)
        assert 0 <= bcur <= dn
        if hdr & H_STORE:
            slots[(hdr >> H_WSLOT_SHIFT) & 63] = (dn, bcur)
        if not (hdr & H_NOSCORE8):
            cost = (dpar - accN - (carryN if flushed else 0)) & U16
            common = (accC + (carryC if flushed else 0) + (1 if hdr & H_FREE else 0)) & U16
            inelig = 0x8000 if common == 0 else 0   # bit 15 = ineligible flag; valid costs stay below 0x8000
            best = min(best, cost | inelig)
        accP = accC = accN = accPB = accCB = 0
        flushed = False

    info = None
    sinfo = None
    body_start = 0
    n_chunks = len(flat.chunk8_body_off) - 1
    body0 = int(flat.chunk8_body_off[c0])
    segments = [(0, flat.pre8_stream, int(flat.chunk8_pre_off[c0]), int(flat.chunk8_pre_off[c0 + 1]))]
    for phase, words, lo, hi in segments:   # (the body segments are appended once the preamble has been replayed)
        i = lo - 1
        if phase == 1:
            # close the chunks that lie wholly in front of this range (the kernel's next_range)
            while chunk < c1 and lo > int(flat.chunk8_body_off[chunk + 1]) - 1:
                lbest[chunk] = best
                if ub is not None:
                    ub[0] = min(ub[0], best)
                best = U16
                chunk += 1
        while i + 1 < hi:
            i += 1
            w = int(words[i])
            ended = False
            if w & H_TAG:
                if w & (H_INFO | H_CHUNK_END | H_NOP) or (w & H_SLOW):
                    assert w & H_RARE   # everything off the fast path carries the one bit the kernel tests first
                else:
                    assert not (w & H_RARE)
                if w & H_INFO:          # first: the jump length of a pruning record overlaps the other flag bits
                    if ub is not None and phase == 1:
                        if w & H_SIB:
                            sinfo = w
                        else:
                            info = w
                    elif ub is not None and phase == 0 and use_pre_records:
                        assert not (w & H_SIB)
                        info = w
                    continue
                if w & H_NOP:
                    continue
                if w & H_CHUNK_END:
                    # the end word names the length of the chunk it opens (the kernel's cend, no table lookup)
                    ln = (w >> CE_LEN_SHIFT) & CE_LEN_MASK
                    assert phase == 1 and i == int(flat.chunk8_body_off[chunk + 1]) - 1
                    want = int(flat.chunk8_body_off[chunk + 2]) - int(flat.chunk8_body_off[chunk + 1]) if chunk + 1 < n_chunks else 0
                    assert ln == (want if want <= CE_LEN_MASK else 0)
                    lbest[chunk] = best
                    if ub is not None:
                        ub[0] = min(ub[0], best)
                    best = U16
                    chunk += 1
                    continue
                hdr = w
                # words of this node: a node with more than 15 of them overflows the kernel's 4-bit counters and takes the general step
                n_mut = 0
                if not (w & H_END):
                    k = i + 1
                    while not (int(words[k]) & M_END):
                        k += 1
                    n_mut = k - i
                hdr_slow = bool(w & H_SLOW)
                if w & H_SLOW:          # cold slot (beyond the LDS-resident ones), or a long node
                    rs, ws = (w >> H_RSLOT_SHIFT) & 63, (w >> H_WSLOT_SHIFT) & 63
                    assert (not (w & H_REG) and rs >= flat.lds_slots) or ((w & H_STORE) and ws >= flat.lds_slots) or n_mut > 15
                else:
                    assert (w & H_REG) or ((w >> H_RSLOT_SHIFT) & 63) < flat.lds_slots
                    assert not (w & H_STORE) or ((w >> H_WSLOT_SHIFT) & 63) < flat.lds_slots
                    assert n_mut <= 15
                dpar, bpar = (dcur, bcur) if w & H_REG else slots[(w >> H_RSLOT_SHIFT) & 63]
                if sinfo is not None:   # sibling record: skip this child and the non-last siblings after it?
                    rec, jump = sinfo, sinfo & INFO_JUMP_MASK
                    sinfo = None
                    if far(dpar, bpar, rec, ub[0]):
                        target = i + jump
                        info = None
                        if stats is not None:
                            stats["skipped"] = stats.get("skipped", 0) + min(target, hi) - i
                            stats["sibling_jumps"] = stats.get("sibling_jumps", 0) + 1
                        while chunk < c1 and target > int(flat.chunk8_body_off[chunk + 1]) - 1:
                            lbest[chunk] = best
                            ub[0] = min(ub[0], best)
                            best = U16
                            chunk += 1
                        i = target - 1
                        continue
                if w & H_END:
                    finish()
                    ended = True
            else:
                site, mi, pi, ri = w & 0x3FFFFF, (w >> 22) & 3, (w >> 24) & 3, (w >> 26) & 3
                x = int(nib[site])
                c, p, r = (x >> mi) & 1, (x >> pi) & 1, (x >> ri) & 1
                accP += p
                accC += c
                accN += c & (1 - p)
                accPB += p & r
                accCB += c & r
                if w & M_FLUSH:
                    assert hdr_slow     # (the pipelined loop of k_best8 has no spill code: only the general step handles M_FLUSH)
                    carryD = ((carryD if flushed else 0) + accP - accC) & U16
                    carryN = ((carryN if flushed else 0) + accN) & U16
                    carryC = ((carryC if flushed else 0) + accC) & U16
                    carryB = (carryB if flushed else 0) + accPB - accCB
                    accP = accC = accN = accPB = accCB = 0
                    flushed = True
                if w & M_END:
                    finish()
                    ended = True
            if ended and info is not None and phase == 0:
                # preamble record of a path node: nothing of its subtree is needed -> stop the replay, start the body behind it
                rec, info = info, None
                if far(dcur, bcur, rec, ub[0], pre=True):
                    body_start = rec & INFO_JUMP_MASK
                    if stats is not None:
                        stats["pre_skipped"] = stats.get("pre_skipped", 0) + min(body_start, int(flat.chunk8_body_off[c1]) - body0)
                    break
            elif ended and info is not None:
                rec, jump = info, info & INFO_JUMP_MASK
                info = None
                is_far = far(dcur, bcur, rec, ub[0])     # D(node) - bound > upper bound: no descendant can tie or win
                if not is_far and b3 is not None:        # third bound (the kernel: at the restart this record asks for)
                    hs3, hr3 = (rec >> INFO_HS_SHIFT) & 0x7F, (rec >> INFO_HR_SHIFT) & 7
                    if hr3 != INFO_HR_NONE and hs3 > hr3 and jump >= B3_MIN_JUMP and \
                            far(dcur, bcur, (rec & ~(0x7F << INFO_HS_SHIFT)) | (hr3 << INFO_HS_SHIFT), ub[0]):   # (asked only if hU = 0 would decide)
                        hu = b3_hu(b3, i, jump)
                        if stats is not None:
                            stats["b3_asked"] = stats.get("b3_asked", 0) + 1
                        if hu + hr3 < hs3:
                            is_far = far(dcur, bcur, (rec & ~(0x7F << INFO_HS_SHIFT)) | ((hu + hr3) << INFO_HS_SHIFT), ub[0])
                            if is_far and stats is not None:
                                stats["b3_jumps"] = stats.get("b3_jumps", 0) + 1
                if is_far:
                    target = i + 1 + jump
                    if stats is not None:
                        stats["skipped"] = stats.get("skipped", 0) + min(target, hi) - (i + 1)
                    while chunk < c1 and target > int(flat.chunk8_body_off[chunk + 1]) - 1:
                        lbest[chunk] = best
                        ub[0] = min(ub[0], best)
                        best = U16
                        chunk += 1
                    i = target - 1
        if phase == 0:
            body_end = int(flat.chunk8_body_off[c1])
            info = None
            segments.append((1, flat.stream8, min(body0 + body_start, body_end), body_end))
    assert chunk == c1
    return lbest


B3_MIN_JUMP = 12


def place8(flat, sample, n_groups=1, prune_ub=None, stats=None, b3=None):
    """Phase 1 (k_best8) + phase 2 (k_gbest, k_select, k_ties with the 32-bit walk, k_final).
    prune_ub: None = no pruning; an int = initial upper bound shared by the groups (0x7F7F = the
    kernel's start value; the true best = the tightest legal bound)."""
    nib, dbot = sample_site_alleles(flat, sample)
    n_chunks = len(flat.chunk8_body_off) - 1
    n_groups = max(1, min(n_groups, n_chunks))
    lbest = {}
    ub = None if prune_ub is None else [int(prune_ub)]
    for g in range(n_groups):
        c0, c1 = g * n_chunks // n_groups, (g + 1) * n_chunks // n_groups
        if c0 < c1:
            lbest.update(best8_group(flat, nib, dbot, c0, c1, ub, stats, b3=b3))
    assert len(lbest) == n_chunks
    gbest = min(lbest.values())
    cnt, key = 0, 0
    for c in range(n_chunks):
        if lbest[c] != gbest:
            continue
        b, k_cnt, k_key = _ties_chunk(flat, nib, dbot, c, gbest)
        cnt += k_cnt
        key = max(key, k_key)
    return {"best": gbest, "num_best": cnt, "best_j": int(flat.rank2bfs[key >> 1]), "has_unique": bool(key & 1),
            "candidate_chunks": sum(1 for c in range(n_chunks) if lbest[c] == gbest)}


def _ties_chunk(flat, nib, dbot, c, want):
    """walk<3>: count / key of the eligible nodes of chunk c whose cost equals `want`."""
    scores = {}
    slots = {}
    cnt, key_best = 0, 0
    dcur = bcur = 0
    info = None
    info_hr = 255
    for words, lo, hi in ((flat.pre_stream, int(flat.chunk_pre_off[c]), int(flat.chunk_pre_off[c + 1])),
                          (flat.stream_t, int(flat.chunk_t_off[c]), int(flat.chunk_t_off[c + 1]))):
        i = lo
        while i < hi:
            w0 = int(words[i]); key = int(words[i + 1]); i += 2
            nmut = w0 & 0xFFFF
            if nmut == 0xFFFF:          # pruning pseudo-record of the tie stream (walk_ties)
                info = key
                info_hr = (w0 >> 16) & 0xFF
                continue
            rslot, wslot = (w0 >> 16) & 63, (w0 >> 22) & 63
            dpar, bpar = (dcur, bcur) if rslot == RS_REG else ((dbot, 0) if rslot == RS_BOTTOM else slots[rslot])
            tsum = tsum_b = neg = common = n_before = 0
            for _ in range(nmut):
                w = int(words[i]); i += 1
                site, mi, pi, ri = w & 0x3FFFFF, (w >> 22) & 3, (w >> 24) & 3, (w >> 26) & 3
                x = int(nib[site])
                cc, pp = (x >> mi) & 1, (x >> pi) & 1
                d = pp - cc
                tsum += d
                tsum_b += d if (x >> ri) & 1 else 0
                if not (w & M_AFTER_MASK):
                    n_before += 1
                    common += cc
                    neg += min(d, 0)
            dn = dpar + tsum
            bn = bpar + tsum_b          # (the kernel keeps D | B << 16 in one word)
            assert 0 <= bn <= dn < 0x8000
            if wslot != WS_NONE:
                slots[wslot] = (dn, bn)
            dcur, bcur = dn, bn
            if not (w0 & F_NOSCORE):
                if w0 & F_ROOT:
                    cost, elig, hu = dn, True, 0
                else:
                    cost = dpar + neg
                    masked = bool(w0 & F_MASKED)
                    elig = common > 0 or (not (w0 & F_LEAF) and not masked and nmut == 0)
                    hu = 1 if (masked or common != n_before) else 0
                if elig and cost == want:
                    cnt += 1
                    key_best = max(key_best, key | hu)
            if info is not None:
                hs, jump = info >> 24, info & 0xFFFFFF
                info = None
                if not (dn <= want + hs and (info_hr == 255 or bn <= want + info_hr)):     # D - hsub > want or B - second hits > want: no descendant can tie
                    i += jump
                    assert i <= hi
    return want, cnt, key_best
