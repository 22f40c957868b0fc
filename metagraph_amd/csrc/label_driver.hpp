// label_driver.hpp — the per-read driver of label-aware alignment: LabeledAligner::filter_seeds, the per-label
// AlignmentAggregator and DBGAligner::align_both_directions / align_core with labeled seeds (A/aligner_labeled.cpp:612-721,
// A/aligner_aggregator.hpp:24-206, A/dbg_aligner.cpp:105-149,360-384,531-758).  Included by align_core.hpp inside namespace mgx
// in builds with MGX_WITH_LABELS; BASIC-, PRIMARY- and CANONICAL-mode graphs, annotation without coordinates (the
// reference's ColumnCompressed case).
//
// Alignment buffers (DevLimits::lab_ext = E, lab_pool): [0, E) the extensions of the current seed, [E, 2E) their reversals
// (seeds of the backward pass), [2E, 3E) backward extensions, [3E, 3E + pool) the aggregator's alignments.  The reference's
// queues hold shared_ptrs; here a queue holds pool indices and Wave::agg counts the references.

enum { AG_NQ = 0, AG_UNL_N = 1, AG_UNL = 2, AG_Q0 = 16, AG_QW = 8 };
MGX_DEV uint32_t ag_ref0(const DevLimits &lim) { return AG_Q0 + lim.lab_queues * AG_QW; }
MGX_DEV uint32_t ag_list0(const DevLimits &lim) { return ag_ref0(lim) + lim.lab_pool + 8; }
MGX_DEV int lab_e(const Wave &w) { return (int)MGX_PARAMS_OF(w).lim.lab_ext; }
MGX_DEV DevAln &lab_pool_aln(Wave &w, uint32_t p) { return w.aln[3 * lab_e(w) + (int)p]; }

MGX_DEV void lab_agg_reset(Wave &w) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    w.agg[AG_NQ] = 0; w.agg[AG_UNL_N] = 0;
    for (uint32_t p = 0; p < lim.lab_pool; ++p) w.agg[ag_ref0(MGX_PARAMS_OF(w).lim) + p] = 0;
}

// ---- LabeledAligner::filter_seeds (aligner_labeled.cpp:612-721, no coordinates) for the seeds of strand s ----
// A label counts the query positions covered by the first k-mers of the seeds whose first node carries it; labels below
// min_exact_match x |query| are dropped, every seed keeps the labels of its first node that are left, seeds without any go.
// With LabeledAligner's seed lengths (<= k) a 150-bp read carries ~120 one-k-mer seeds per strand, nearly all with the same
// label(s) and first k-mers that overlap their neighbour's: the rows of the seeds' first nodes are fetched one seed per lane,
// and the position bitmaps are written one coalesced range per run of seeds (same label, touching ranges) — exact for any
// order of the seeds, a handful of writes for the usual one.
MGX_DEV LabRow lab_row_of_head(const AlignParams &P, uint64_t head, uint64_t row) {
    LabRow r;
    uint32_t c = (uint32_t)(head & 0xFFFF);
    if (c == 0xFFFF) c = P.anno_count[row];
    r.n = c; r.one = (uint32_t)(head >> 16); r.more = c >= 2 ? P.anno_more + (head >> 16) : nullptr;
    return r;
}
MGX_NI_G4 void lab_filter_seeds(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t n = w.n_seeds[s];
    if (!n) return;
    const int32_t k = (int32_t)P.g.k, L = w.L;
    const uint32_t W = (uint32_t)(L + 31) / 32;
    // scratch: the backtracking's start-cell list (dead between extensions)
    uint32_t *scr = (uint32_t *)w.indices;
    const uint64_t cap_words = (uint64_t)P.lim.max_columns * 2 * (sizeof(BtIndex) / 4);
    const uint32_t nrw = 2 * (((uint32_t)n + 63) / 64);
    const uint32_t ML = (P.lim.lab_queues + 1u) & ~1u;       // labels the read's seeds may carry (as many as the aggregator has queues)
    const uint64_t fixed_words = ML + 2ull * n + n + (n & 1) + nrw + n + n;
    if (cap_words < fixed_words + (uint64_t)W) { w.status = ST_CAPACITY; return; }
    const uint32_t max_l = (uint32_t)imin<uint64_t>(ML, (cap_words - fixed_words) / W);
    uint32_t *mlab = scr;                                    // labels seen on the seeds' first nodes (VectorMap order)
    uint64_t *heads = (uint64_t *)(scr + ML);                // per seed: the head word of its first node's row (0: no labels)
    uint32_t *span = scr + ML + 2 * n;                       // per seed: first k-mer's query range, lo | hi << 16
    uint64_t *rs = (uint64_t *)(span + n + (n & 1));         // bit i: seed i starts a run (below)
    uint32_t *sh = (uint32_t *)rs + nrw;                     // per run start: the run's label set
    uint32_t *ends = sh + n;                                 // per kept seed: its query end (num_matching)
    uint32_t *bits = ends + n;                               // per label: covered query positions
    // Pass 1, one seed per lane: the row of its first node ("skip dummy nodes": W == 0 has no labels) and whether it CONTINUES
    // the run of the seed before — both rows are the same single label and its range touches and extends the other's — so that
    // the sequential parts below run per run of seeds, not per seed (a read's ~120 one-k-mer seeds are a handful of runs)
    for (uint32_t x = 0; x < nrw / 2; ++x) rs[x] = 0;
    LV<int32_t> lines;
    FOR_LANES(l) { lines[l] = 0; }
    int32_t c_lbl = -1, c_lo = 0, c_hi = 0, c_simple = 0;       // the last seed of the chunk before
    for (int32_t base = 0; base < n; base += WAVE) {
        LV<int32_t> lbl, lo, hi, simple;
        FOR_LANES(l) {
            const int32_t i = base + l;
            lbl[l] = -1; lo[l] = 0; hi[l] = 0; simple[l] = 0;
            if (i < n) {
                const DevSeed sd = w.seeds[s][i];
                const uint32_t node0 = lab_base_node(P, sd.offset == 0 ? w.nodes[s][sd.clipping] : sd.node);
                uint64_t h = 0;
                if (node0 && node0 <= P.g.n && (uint64_t)node0 - 1 < P.anno_rows) {
                    bool real = true;
                    if (!(P.labeled & 2u)) {
                        LineCtr lc = { 0, 0, 0 };
                        real = get_W(P.g, node0, lc) != 0;
                        lines[l] += (int32_t)(lc.rank_lines + lc.select_lines + lc.bit_lines);
                    }
                    if (real) h = gld(P.anno_head + ((uint64_t)node0 - 1));
                }
                lo[l] = (int32_t)sd.clipping; hi[l] = imin((int32_t)sd.clipping + k - (int32_t)sd.offset, L);
                simple[l] = (h & 0xFFFF) == 1 ? 1 : 0;
                lbl[l] = simple[l] ? (int32_t)(uint32_t)(h >> 16) : -1;
                gst(heads + i, h);
                gst(span + i, (uint32_t)lo[l] | ((uint32_t)hi[l] << 16));
            }
        }
        const LV<int32_t> p_lbl = wave_shift_up1(lbl, c_lbl), p_lo = wave_shift_up1(lo, c_lo), p_hi = wave_shift_up1(hi, c_hi),
                          p_simple = wave_shift_up1(simple, c_simple);
        LV<bool> st;
        FOR_LANES(l) {
            const int32_t i = base + l;
            const bool cont = i > 0 && simple[l] && p_simple[l] && lbl[l] == p_lbl[l] && lo[l] >= p_lo[l] && lo[l] <= p_hi[l] && hi[l] >= p_hi[l];
            st[l] = i < n && !cont;
        }
        const uint64_t sb = wave_ballot(st);
        FOR_LANES(l) { if (l == 0 && sb) rs[base >> 6] |= sb << (base & 63); }
        c_lbl = wave_bcast(lbl, WAVE - 1); c_lo = wave_bcast(lo, WAVE - 1); c_hi = wave_bcast(hi, WAVE - 1); c_simple = wave_bcast(simple, WAVE - 1);
        wave_sync();
    }
    w.ctr.rank_lines += (uint32_t)wave_sum(lines);
    wave_sync();
    // Pass 2, per run: its labels' position bitmaps (label_mapper / indicator, :620-640)
    uint32_t nl = 0;
    for (int32_t i = bits_next(rs, n, 0, true); i < n; ) {
        const int32_t j = bits_next(rs, n, i + 1, true);             // the next run's start (n: none)
        const uint64_t h = heads[i];
        if (h & 0xFFFF) {
            const DevSeed sd = w.seeds[s][i];
            const uint32_t node0 = lab_base_node(P, sd.offset == 0 ? w.nodes[s][sd.clipping] : sd.node);
            const LabRow r = lab_row_of_head(P, h, (uint64_t)node0 - 1);
            const int32_t lo = (int32_t)(span[i] & 0xFFFF), hi = (int32_t)(span[j - 1] >> 16);     // (a run of several seeds: single-label rows, ranges nested in order)
            for (uint32_t x = 0; x < r.n; ++x) {
                const uint32_t lbl = row_at(r, x);
                uint32_t t = 0;
                while (t < nl && mlab[t] != lbl) ++t;
                if (t == nl) {
                    if (nl == max_l) { w.status = ST_CAPACITY; return; }
                    mlab[nl] = lbl;
                    for (uint32_t y = 0; y < W; ++y) bits[nl * W + y] = 0;
                    ++nl;
                }
                uint32_t *bw = bits + t * W;
                for (int32_t wd = lo >> 5; hi > lo && wd <= (hi - 1) >> 5; ++wd) {
                    const int32_t a0 = imax(lo, wd << 5) & 31, b0 = imin(hi, (wd + 1) << 5) - (wd << 5);      // bits [a0, b0) of word wd
                    bw[wd] |= (b0 >= 32 ? 0xFFFFFFFFu : ((1u << b0) - 1u)) & ~((1u << a0) - 1u);
                }
            }
        }
        i = j;
    }
    if (!nl) { w.n_seeds[s] = 0; w.num_matching[s] = 0; return; }
    // labels at or above the cut-off, ascending (only the SET is used afterwards)
    const double cutoff = P.cfg.min_exact_match * (double)L;
    uint32_t nk = 0;
    for (uint32_t t = 0; t < nl; ++t) {
        uint32_t cnt = 0;
        for (uint32_t y = 0; y < W; ++y) cnt += (uint32_t)popc64((uint64_t)bits[t * W + y]);
        if ((double)cnt < cutoff) continue;
        const uint32_t lbl = mlab[t];
        uint32_t pos = nk;                                   // (nk <= t: the slots below t are free to reuse)
        while (pos > 0 && mlab[pos - 1] > lbl) { mlab[pos] = mlab[pos - 1]; --pos; }
        mlab[pos] = lbl;
        ++nk;
    }
    if (!nk) { w.n_seeds[s] = 0; w.num_matching[s] = 0; return; }
    uint32_t cntk = 0;
    for (uint32_t t = 0; t < nk; ++t) lab_push(w, cntk, mlab[t]);
    const uint32_t hk = lab_end(w, cntk);
    if (w.status != ST_OK) return;
    // Pass 3, per run: the label set its seeds keep (first node's labels & the labels left, :700-712)
    uint32_t hk_kept = 0;                                    // the whole set, kept for the read (made once)
    uint32_t one_lbl = 0xFFFFFFFFu, one_h = 0;               // (runs with the same single label share their set)
    for (int32_t i = bits_next(rs, n, 0, true); i < n; i = bits_next(rs, n, i + 1, true)) {
        const uint64_t hd = heads[i];
        uint32_t h = 0;
        if (hd & 0xFFFF) {
            if ((hd & 0xFFFF) == 1 && (uint32_t)(hd >> 16) == one_lbl) {
                h = one_h;
            } else {
                const DevSeed sd = w.seeds[s][i];
                const uint32_t node0 = lab_base_node(P, sd.offset == 0 ? w.nodes[s][sd.clipping] : sd.node);
                const LabRow r = lab_row_of_head(P, hd, (uint64_t)node0 - 1);
                h = lab_isect_row(w, hk, r);
                if (w.status != ST_OK) return;
                if (h) {
                    if (h == hk) { if (!hk_kept) hk_kept = lab_persist(w, hk); h = hk_kept; }
                    else h = lab_persist(w, h);
                    if (w.status != ST_OK) return;
                }
                if ((hd & 0xFFFF) == 1) { one_lbl = (uint32_t)(hd >> 16); one_h = h; }
            }
        }
        sh[i] = h;
    }
    wave_sync();
    // Pass 4, one seed per lane: seeds without labels go (:714-716), the others move up; num_matching of what is left
    // (get_num_char_matches_in_seeds, alignment.hpp:100-127, incl. its quirk: nothing after the first sub-k seed is counted)
    int32_t m = 0, first_off = INT32_MAX;
    uint32_t num_matching = 0;
    for (int32_t base = 0; base < n; base += WAVE) {
        LV<int32_t> keep, qb, qe, offp;
        LV<uint32_t> hset;
        LV<DevSeed> sdv;
        FOR_LANES(l) {
            const int32_t i = base + l;
            keep[l] = 0; qb[l] = 0; qe[l] = 0; offp[l] = INT32_MAX; hset[l] = 0;
            if (i < n) {
                // the run seed i belongs to: the last start at or before i
                int32_t wd = i >> 6;
                uint64_t bw = rs[wd] & (~0ull >> (63 - (i & 63)));
                while (!bw) bw = rs[--wd];
                const int32_t st = (wd << 6) + 63 - clz64(bw);
                const uint32_t h = sh[st];
                sdv[l] = w.seeds[s][i];
                if (h) { keep[l] = 1; hset[l] = h; qb[l] = (int32_t)sdv[l].clipping; qe[l] = qb[l] + (int32_t)sdv[l].length; }
            }
        }
        wave_sync();
        const LV<int32_t> po = wave_prefix_sum_excl(keep);
        const int32_t tot = wave_sum(keep);
        FOR_LANES(l) {
            if (keep[l]) {
                const int32_t pos = m + po[l];
                w.seeds[s][pos] = sdv[l]; w.seed_lab[s][pos] = hset[l]; w.alive[s][pos] = 1;
                ends[pos] = (uint32_t)qe[l];
                if (sdv[l].offset) offp[l] = pos;
            }
        }
        wave_sync();
        const int32_t lim_pos = imin(first_off, wave_min(offp));      // seeds up to the first sub-k seed count
        LV<int32_t> contrib;
        FOR_LANES(l) {
            int32_t c = 0;
            const int32_t pos = m + po[l];
            if (keep[l] && pos <= lim_pos) {
                const int32_t last_q_end = pos ? (int32_t)ends[pos - 1] : 0;
                if (qe[l] > last_q_end) c = (qe[l] - qb[l]) - (qb[l] < last_q_end ? last_q_end - qb[l] : 0);
            }
            contrib[l] = c;
        }
        num_matching += (uint32_t)wave_sum(contrib);
        first_off = lim_pos;
        m += tot;
    }
    w.n_seeds[s] = m;
    w.num_matching[s] = num_matching;
    wave_sync();
}

// ---- AlignmentAggregator with labels (aligner_aggregator.hpp:24-206) ----
MGX_DEV int lab_pool_alloc(Wave &w) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    for (uint32_t p = 0; p < lim.lab_pool; ++p) if (!w.agg[ag_ref0(MGX_PARAMS_OF(w).lim) + p]) return (int)p;
    w.status = ST_CAPACITY;
    return -1;
}
// (queues are short: items[] of a queue record or of the unlabeled queue, *size its length)
MGX_DEV int lab_q_max(Wave &w, const uint32_t *items, uint32_t size) {       // std::max_element: the first of equal maxima
    uint32_t mx = 0;
    for (uint32_t t = 1; t < size; ++t) if (aln_less(lab_pool_aln(w, items[mx]), lab_pool_aln(w, items[t]))) mx = t;
    return (int)mx;
}
MGX_DEV int lab_q_min(Wave &w, const uint32_t *items, uint32_t size) {       // std::min_element: the first of equal minima
    uint32_t mn = 0;
    for (uint32_t t = 1; t < size; ++t) if (aln_less(lab_pool_aln(w, items[t]), lab_pool_aln(w, items[mn]))) mn = t;
    return (int)mn;
}
MGX_DEV int32_t lab_global_cutoff(Wave &w) {                // get_global_cutoff (:141-149)
    const uint32_t un = w.agg[AG_UNL_N];
    if (!un) return NINF;
    const int32_t cur_max = lab_pool_aln(w, w.agg[AG_UNL + lab_q_max(w, w.agg + AG_UNL, un)]).score;
    return cur_max > 0 ? (int32_t)((double)cur_max * MGX_PARAMS_OF(w).cfg.rel_score_cutoff) : cur_max;
}
MGX_DEV int lab_find_queue(const Wave &w, uint32_t label) {
    const uint32_t nq = w.agg[AG_NQ];
    for (uint32_t q = 0; q < nq; ++q) if (w.agg[AG_Q0 + q * AG_QW] == label) return (int)q;
    return -1;
}
MGX_DEV int32_t lab_label_cutoff(Wave &w, uint32_t label) {  // get_label_cutoff (:168-177)
    const int q = lab_find_queue(w, label);
    if (q < 0) return NINF;
    const uint32_t *rec = w.agg + AG_Q0 + (uint32_t)q * AG_QW;
    if (rec[1] < (uint32_t)n_alt_of(w)) return NINF;
    return lab_pool_aln(w, rec[2 + lab_q_min(w, rec + 2, rec[1])]).score;
}
MGX_DEV int32_t lab_score_cutoff(Wave &w, uint32_t lab) {    // get_score_cutoff (:152-166)
    const int32_t global_min = lab_global_cutoff(w);
    int32_t min_score = INT32_MAX;
    const uint32_t n = lab_size(w, lab);
    for (uint32_t x = 0; x < n; ++x) {
        min_score = imin(min_score, lab_label_cutoff(w, lab_at(w, lab, x)));
        if (min_score < global_min) return global_min;
    }
    return min_score;
}
// get_min_path_score of align_batch (dbg_aligner.cpp:277-282 with labels)
MGX_DEV int32_t lab_min_path_score(Wave &w, uint32_t lab) {
    return imax(MGX_PARAMS_OF(w).cfg.min_path_score, lab ? lab_score_cutoff(w, lab) : lab_global_cutoff(w));
}

// add_alignment (:68-138); true if the alignment was added
MGX_NI_G4 bool lab_add_alignment(Wave &w, const DevAln &a) {
    MGX_ASSUME_LDS(&w);
    uint32_t *g = w.agg;
    const uint32_t n_alt = (uint32_t)n_alt_of(w);
    int pa = -1;                                            // a's slot in the pool once it is needed
    auto own = [&]() -> int {
        if (pa < 0) { pa = lab_pool_alloc(w); if (pa >= 0) copy_aln(lab_pool_aln(w, (uint32_t)pa), a); }
        return pa;
    };
    auto queue_of = [&](uint32_t label) -> int {
        int q = lab_find_queue(w, label);
        if (q >= 0) return q;
        if (g[AG_NQ] >= MGX_PARAMS_OF(w).lim.lab_queues) { w.status = ST_CAPACITY; return -1; }
        q = (int)g[AG_NQ]++;
        g[AG_Q0 + (uint32_t)q * AG_QW] = label; g[AG_Q0 + (uint32_t)q * AG_QW + 1] = 0;
        return q;
    };
    const uint32_t nl = a.lab ? lab_size(w, a.lab) : 0;
    if (!g[AG_UNL_N]) {
        if (own() < 0) return false;
        g[AG_UNL] = (uint32_t)pa; g[AG_UNL_N] = 1; ++g[ag_ref0(MGX_PARAMS_OF(w).lim) + (uint32_t)pa];
        for (uint32_t x = 0; x < nl; ++x) {
            const int q = queue_of(lab_at(w, a.lab, x));
            if (q < 0) return false;
            uint32_t *rec = g + AG_Q0 + (uint32_t)q * AG_QW;
            rec[2 + rec[1]++] = (uint32_t)pa; ++g[ag_ref0(MGX_PARAMS_OF(w).lim) + (uint32_t)pa];
        }
        return true;
    }
    if (a.score < lab_global_cutoff(w)) return false;
    auto push_to_queue = [&](uint32_t *items, uint32_t *size) -> bool {
        for (uint32_t t = 0; t < *size; ++t) if (aln_equal(w, a, lab_pool_aln(w, items[t]))) return false;
        if (*size < n_alt) {
            if (own() < 0) return false;
            items[(*size)++] = (uint32_t)pa; ++g[ag_ref0(MGX_PARAMS_OF(w).lim) + (uint32_t)pa];
            return true;
        }
        const int mn = lab_q_min(w, items, *size);
        if (aln_less(a, lab_pool_aln(w, items[mn]))) return false;
        if (own() < 0) return false;
        ++g[ag_ref0(MGX_PARAMS_OF(w).lim) + (uint32_t)pa];                       // (before the release: the slot `a` was copied to stays taken)
        --g[ag_ref0(MGX_PARAMS_OF(w).lim) + items[mn]];
        items[mn] = (uint32_t)pa;                            // queue.update(minimum, a)
        return true;
    };
    if (!nl) return push_to_queue(g + AG_UNL, g + AG_UNL_N);
    if (!g[AG_NQ] && g[AG_UNL_N] > 1) {
        // the first labeled alignment: the global queue only serves the global cut-off from now on (:110-117)
        const uint32_t keep = g[AG_UNL + lab_q_max(w, g + AG_UNL, g[AG_UNL_N])];
        for (uint32_t t = 0; t < g[AG_UNL_N]; ++t) if (g[AG_UNL + t] != keep) --g[ag_ref0(MGX_PARAMS_OF(w).lim) + g[AG_UNL + t]];
        g[AG_UNL] = keep; g[AG_UNL_N] = 1;
    }
    bool added = false;
    for (uint32_t x = 0; x < nl; ++x) {
        const int q = queue_of(lab_at(w, a.lab, x));
        if (q < 0) return false;
        uint32_t *rec = g + AG_Q0 + (uint32_t)q * AG_QW;
        added |= push_to_queue(rec + 2, rec + 1);
        if (w.status != ST_OK) return false;
    }
    if (!added) return false;
    if (!aln_less(a, lab_pool_aln(w, g[AG_UNL + lab_q_max(w, g + AG_UNL, g[AG_UNL_N])]))) {
        const int mn = lab_q_min(w, g + AG_UNL, g[AG_UNL_N]);
        ++g[ag_ref0(MGX_PARAMS_OF(w).lim) + (uint32_t)pa];
        --g[ag_ref0(MGX_PARAMS_OF(w).lim) + g[AG_UNL + mn]];
        g[AG_UNL + mn] = (uint32_t)pa;
    }
    return true;
}

// filter_seed (dbg_aligner.cpp:105-149, no coordinates): a seed (or reversed alignment) that check_seed rejected keeps only the
// labels the extended one did not have; returns the labels left (0: the seed is dropped)
MGX_DEV uint32_t lab_filter_seed(Wave &w, uint32_t prev_lab, uint32_t lab) {
    if (!prev_lab) return 0;
    const uint32_t d = lab_diff(w, lab, prev_lab);
    return lab_persist(w, d);
}

// filter_seed for every later seed of the list after seed i's extensions (dbg_aligner.cpp:379-382 / :731-734): the look-ups into
// the extender's convergence table are independent — one seed per lane — and the label bookkeeping of the rejected ones
// follows, in order
MGX_DEV void lab_check_later(Wave &w, const ExtenderState &F, int s, int32_t i, int32_t n) {
    const uint32_t prev = w.seed_lab[s][i];
    for (int32_t base = i + 1; base < n; base += WAVE) {
        FOR_LANES(l) {
            const int32_t j = base + l;
            if (j < n && w.alive[s][j]) {
                const DevSeed sj = w.seeds[s][j];
                const uint32_t last_node = sj.offset == 0 ? w.nodes[s][sj.clipping + sj.n_nodes - 1] : sj.node;
                const SeedRef rj = seedref_from_seed(w, s, j, nullptr);
                // rejected: it keeps the labels seed i did not have — none if it holds the very set of seed i (the usual case:
                // the read's seeds share one set); other sets are worked out below, in order
                if (!check_seed(w, F, last_node, rj.qlen, rj.clipping, rj.score))
                    w.alive[s][j] = (prev && w.seed_lab[s][j] == prev) ? 0 : 2;
            }
        }
    }
    wave_sync();
    uint32_t memo_in = 0xFFFFFFFFu, memo_out = 0;            // (runs of seeds share one label set)
    for (int32_t base = i + 1; base < n; base += WAVE) {
        LV<bool> todo;
        FOR_LANES(l) { const int32_t j = base + l; todo[l] = j < n && w.alive[s][j] == 2; }
        uint64_t tb = wave_ballot(todo);
        while (tb) {
            const int32_t j = base + ctz64(tb);
            tb &= tb - 1;
            const uint32_t cur = w.seed_lab[s][j];
            if (cur != memo_in) { memo_out = lab_filter_seed(w, prev, cur); memo_in = cur; if (w.status != ST_OK) return; }
            w.seed_lab[s][j] = memo_out;
            w.alive[s][j] = memo_out ? 1 : 0;
        }
    }
}

// ---- align_both_directions (dbg_aligner.cpp:531-758, the branch without chaining) with labeled seeds (every graph mode) ----
MGX_NI_G4 void lab_aln_both(Wave &w, int s) {
    MGX_ASSUME_LDS(&w);
    const AlignParams &P = MGX_PARAMS_OF(w);
    ExtenderState &F = w.ext[s];
    ExtenderState &B = w.ext[1 - s];
    // a PRIMARY graph behind the CanonicalDBG wrapper holds both strands (:644-655): the backward pass runs on the same graph,
    // an alignment on the reverse strand is reported as the forward alignment it mirrors (is_reversible: orientation && !offset)
    const bool canon = kWithPrimary && P.cfg.canonical != 0;
    F.rc_view = 0;
    B.rc_view = canon ? 0 : 1;                                // use_rcdbg
    const int E = lab_e(w);
    const int32_t n = w.n_seeds[s];
    for (int32_t i = 0; i < n; ++i) {
        if (!w.alive[s][i]) continue;
        SeedRef seed = seedref_from_seed(w, s, i, nullptr);
        conv_clear(w, F.conv);
        uint64_t t0 = cycle_clock();
        extend(w, s, seed, false);
        uint64_t t1 = cycle_clock();
        w.cyc[2] += t1 - t0;
        const ExtendResult er = w.er;
        if (w.status != ST_OK) return;
        const int n_fwd = backtrack(w, s, seed, nullptr, er, imax(0, P.cfg.min_cell_score), &w.aln[0], E);
        w.cyc[3] += cycle_clock() - t1;
        if (w.status != ST_OK) return;
        int n_rev = 0;
        uint64_t rev_alive = 0;           // bit r: reversal r is still to be extended (lab_ext <= 64: derive_limits)
        for (int e = 0; e < n_fwd; ++e) {
            DevAln &path = w.aln[e];
            DevAln &rev = w.aln[E + n_rev];
            if (canon) {
                // the mirror image serves both purposes: what is reported for a reverse-strand alignment (:683-689) and the seed
                // of the backward pass (:695) — the same reverse_complement(graph_, query_rc)
                const bool reversible = path.orientation && !path.offset;
                const bool to_left = aln_clipping(path) && !path.offset;
                const bool good = path.score >= lab_min_path_score(w, path.lab);
                bool have_rev = false;
                if ((good && reversible) || to_left) { copy_aln(rev, path); have_rev = reverse_complement_aln_stored(w, rev); }
                if (good) {
                    if (reversible) { if (have_rev) lab_add_alignment(w, rev); } else lab_add_alignment(w, path);
                    if (w.status != ST_OK) return;
                }
                if (!to_left || !have_rev) continue;
                rev_alive |= 1ull << n_rev++;
                continue;
            }
            if (path.score >= lab_min_path_score(w, path.lab)) { lab_add_alignment(w, path); if (w.status != ST_OK) return; }
            if (!aln_clipping(path) || path.offset) continue;
            copy_aln(rev, path);
            if (!reverse_complement_aln(w, rev)) continue;
            rev_alive |= 1ull << n_rev++;
        }
        // align_core (:360-384) on the backward extender over the reversed extensions
        for (int r = 0; r < n_rev; ++r) {
            if (!((rev_alive >> r) & 1)) continue;
            DevAln &rev = w.aln[E + r];
            SeedRef rseed = seedref_from_aln(rev);
            const int32_t mps2 = imax(0, lab_min_path_score(w, rev.lab));
            conv_clear(w, B.conv);
            const uint64_t t2 = cycle_clock();
            extend(w, 1 - s, rseed, true);
            const uint64_t t3 = cycle_clock();
            w.cyc[2] += t3 - t2;
            const ExtendResult er2 = w.er;
            if (w.status != ST_OK) return;
            const int n_bwd = backtrack(w, 1 - s, rseed, &rev, er2, mps2, &w.aln[2 * E], E);
            w.cyc[3] += cycle_clock() - t3;
            if (w.status != ST_OK) return;
            for (int b = 0; b < n_bwd; ++b) {
                DevAln &p2 = w.aln[2 * E + b];
                if (canon && !(p2.orientation && !p2.offset)) {       // not reversible: as it is (:711)
                    lab_add_alignment(w, p2);
                    if (w.status != ST_OK) return;
                    continue;
                }
                if (!(canon ? reverse_complement_aln_stored(w, p2) : reverse_complement_aln(w, p2))) continue;
                const int32_t clip = aln_clipping(p2), eclip = aln_end_clipping(p2);
                for (int32_t x = 0; x < p2.n_nodes; ++x) filter_nodes(w, F, p2.nodes[x], clip, w.L - eclip);
                if (w.status != ST_OK) return;
                lab_add_alignment(w, p2);
                if (w.status != ST_OK) return;
            }
            for (int r2 = r + 1; r2 < n_rev; ++r2) {
                if (!((rev_alive >> r2) & 1)) continue;
                DevAln &o = w.aln[E + r2];
                if (!check_seed(w, B, o.nodes[o.n_nodes - 1], o.qlen, aln_clipping(o), o.score)) {
                    o.lab = lab_filter_seed(w, rev.lab, o.lab);
                    if (w.status != ST_OK) return;
                    if (!o.lab) rev_alive &= ~(1ull << r2);
                }
            }
        }
        lab_check_later(w, F, s, i, n);
        if (w.status != ST_OK) return;
    }
}

// align_core (:360-384) with the labeled seeds of strand 0, forward only
MGX_NI_G4 void lab_align_core_fwd(Wave &w) {
    MGX_ASSUME_LDS(&w);
    ExtenderState &F = w.ext[0];
    F.rc_view = 0;
    const int E = lab_e(w);
    const int32_t n = w.n_seeds[0];
    for (int32_t i = 0; i < n; ++i) {
        if (!w.alive[0][i]) continue;
        SeedRef seed = seedref_from_seed(w, 0, i, nullptr);
        const int32_t mps = imax(0, lab_min_path_score(w, seed.lab));
        conv_clear(w, F.conv);
        extend(w, 0, seed, false);
        const ExtendResult er = w.er;
        if (w.status != ST_OK) return;
        const int n_fwd = backtrack(w, 0, seed, nullptr, er, mps, &w.aln[0], E);
        if (w.status != ST_OK) return;
        for (int e = 0; e < n_fwd; ++e) { lab_add_alignment(w, w.aln[e]); if (w.status != ST_OK) return; }
        lab_check_later(w, F, 0, i, n);
        if (w.status != ST_OK) return;
    }
}

// get_alignments (:180-202): every queue's alignments in the queues' insertion order, then the global queue's; stable sort
// ascending by LocalAlignmentLess; emitted from the back, an alignment that sits in several queues once (where it comes
// first from the back).  Writes the emission order (pool indices) to order[] and returns its length.
MGX_DEV int lab_get_alignments(Wave &w, uint32_t *order) {
    const DevLimits &lim = MGX_PARAMS_OF(w).lim;
    uint32_t *g = w.agg;
    uint32_t *list = g + ag_list0(lim);
    int n = 0;
    auto insert = [&](uint32_t p) {
        int pos = n;
        while (pos > 0 && aln_less(lab_pool_aln(w, p), lab_pool_aln(w, list[pos - 1]))) { list[pos] = list[pos - 1]; --pos; }
        list[pos] = p;
        ++n;
    };
    for (uint32_t q = 0; q < g[AG_NQ]; ++q) {
        const uint32_t *rec = g + AG_Q0 + q * AG_QW;
        for (uint32_t t = 0; t < rec[1]; ++t) insert(rec[2 + t]);
    }
    for (uint32_t t = 0; t < g[AG_UNL_N]; ++t) insert(g[AG_UNL + t]);
    int m = 0;
    for (int t = n - 1; t >= 0; --t) {
        const uint32_t p = list[t];
        bool seen = false;
        for (int x = 0; x < m; ++x) seen |= order[x] == p;
        if (seen || !lab_pool_aln(w, p).n_nodes) continue;
        order[m++] = p;
    }
    return m;
}
