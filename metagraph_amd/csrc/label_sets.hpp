// label_sets.hpp — label sets of label-aware alignment and the LabeledExtender's bookkeeping on the DP table
// (A/aligner_labeled.{hpp,cpp}, annotation without coordinates).  Included by align_core.hpp inside namespace mgx, in
// builds with MGX_WITH_LABELS only; AlignParams::labeled switches the hooks on at run time.
//
// What the reference keeps and what stands in for it here:
//   * AnnotationBuffer (annotation_buffer.{hpp,cpp}) caches node -> label set for the nodes it was asked to fetch and interns
//     the sets (cache_column_set).  Here a node's labels ARE its row of the row-major label matrix (mgx_annot.hip): lab_row()
//     reads them where they are needed, so there is nothing to queue or fetch; dummy nodes (W == 0) have no labels
//     (annotation_buffer.cpp: "skip dummy nodes").
//   * A label set (Alignment::Columns, a sorted vector) is a run of the read's label arena (Wave::lab): word h = its size,
//     words h + 1 .. its labels in ascending order; the handle h = 0 is the empty set, which is what the reference's set
//     index 0 is.  Handles are never compared for anything but emptiness, so sets are not interned; an intersection that
//     keeps every label of its first operand returns that operand's handle (the common case along a path), so straight
//     stretches of the graph allocate nothing.
//   * Sets made during an extension (columns, the labels still to be reported) live from extend_begin to the end of its
//     backtracking: they grow up from lab_lo, which extend_begin resets.  Sets that outlive it (seeds, alignments) grow down
//     from lab_hi for the whole read.  The two meeting is ST_CAPACITY.
//   * LabeledExtender::node_labels_ is Wave::col_lab (one handle per table column), last_flushed_table_i_ /
//     remaining_labels_i_ are Wave::last_flushed / remaining_lab.
//
// Everything here is uniform (scalar) code of the wave program: label sets of short reads hold a handful of labels.

MGX_DEV uint32_t lab_size(const Wave &w, uint32_t h) { return w.lab[h]; }
MGX_DEV uint32_t lab_at(const Wave &w, uint32_t h, uint32_t i) { return w.lab[h + 1 + i]; }

// a set under construction at lab_lo: lab_push() appends (ascending), lab_end() closes it; nothing else may allocate in between
MGX_DEV void lab_push(Wave &w, uint32_t &n, uint32_t label) {
    if ((uint64_t)w.lab_lo + 2 + n >= w.lab_hi) { w.status = ST_CAPACITY; return; }
    w.lab[w.lab_lo + 1 + n] = label;
    ++n;
}
MGX_DEV uint32_t lab_end(Wave &w, uint32_t n) {
    if (!n || w.status != ST_OK) return 0;
    const uint32_t h = w.lab_lo;
    w.lab[h] = n;
    w.lab_lo = h + 1 + n;
    return h;
}
// a copy that lives as long as the read (no copy for a set that already does)
MGX_DEV uint32_t lab_persist(Wave &w, uint32_t h) {
    if (!h || h >= w.lab_hi) return h;
    const uint32_t n = w.lab[h];
    if ((uint64_t)w.lab_lo + n + 2 >= w.lab_hi) { w.status = ST_CAPACITY; return 0; }
    const uint32_t d = w.lab_hi - (n + 1);
    for (uint32_t x = 0; x <= n; ++x) w.lab[d + x] = w.lab[h + x];
    w.lab_hi = d;
    return d;
}

// A PRIMARY graph is annotated and looked up by BASE node: the wrapper's ids above n are the reverse complements of the base
// nodes v - n (CanonicalDBG::get_base_node, canonical_dbg.hpp; annotation_buffer.cpp:41-63, :195-217)
// A CANONICAL-mode graph is annotated and looked up by the k-mer's representative — the smaller BOSS index of the k-mer and its
// reverse complement (annotation_buffer.cpp:56-62: spell_path + map_to_nodes; here a table built once per aligner)
MGX_DEV uint32_t lab_base_node(const AlignParams &P, uint32_t node) {
    if (P.anno_base) return (node && node <= P.g.n) ? P.anno_base[node] : 0u;
    return (kWithPrimary && P.cfg.canonical >= 2 && node > P.g.n) ? node - (uint32_t)P.g.n : node;
}

// the labels of a node: its row of the label matrix (row = node - 1: AnnotatedDBG::graph_to_anno_index, annotated_dbg.hpp:50-52)
struct LabRow { uint32_t n, one; const uint32_t *more; };
MGX_DEV uint32_t row_at(const LabRow &r, uint32_t i) { return r.n == 1 ? r.one : r.more[i]; }
MGX_DEV LabRow lab_row(Wave &w, uint32_t node) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    LabRow r;
    r.n = 0; r.one = 0; r.more = nullptr;
    node = lab_base_node(P, node);
    if (!node || node > P.g.n) return r;                     // npos (annotation_buffer.cpp:59-62)
    // "skip dummy nodes" (annotation_buffer.cpp:64-68): !boss.get_W(node) — unless the annotation was checked to hold no
    // label on any dummy node's row (AlignParams::labeled bit 1: mgx_labeled_aligner_create looks once), when the row alone says it
    if (!(P.labeled & 2u)) {
        ++w.ctr.rank_lines;
        const Block b = load_block_uniform(P.g, uni(node >> 6));
        if (block_W(b, (int)(node & 63)) == 0) return r;
    }
    const uint64_t row = (uint64_t)node - 1;
    if (row >= P.anno_rows) return r;
    const uint64_t h = P.anno_head[row];
    uint32_t c = (uint32_t)(h & 0xFFFF);
    if (c == 0xFFFF) c = P.anno_count[row];
    r.n = c;
    if (c == 1) r.one = (uint32_t)(h >> 16);
    else if (c >= 2) r.more = P.anno_more + (h >> 16);
    return r;
}

// set h & the labels of a row; h itself if nothing is lost
MGX_DEV uint32_t lab_isect_row(Wave &w, uint32_t h, const LabRow &r) {
    if (!h || !r.n) return 0;
    const uint32_t n = lab_size(w, h);
    uint32_t cnt = 0, i = 0, j = 0;
    while (i < n && j < r.n) {
        const uint32_t a = lab_at(w, h, i), b = row_at(r, j);
        if (a < b) ++i;
        else if (b < a) ++j;
        else { lab_push(w, cnt, a); ++i; ++j; }
        if (w.status != ST_OK) return 0;
    }
    if (cnt == n) return h;
    return lab_end(w, cnt);
}
// a & b; a itself if nothing is lost
MGX_DEV uint32_t lab_isect(Wave &w, uint32_t a, uint32_t b) {
    if (!a || !b) return 0;
    const uint32_t na = lab_size(w, a), nb = lab_size(w, b);
    uint32_t cnt = 0, i = 0, j = 0;
    while (i < na && j < nb) {
        const uint32_t x = lab_at(w, a, i), y = lab_at(w, b, j);
        if (x < y) ++i;
        else if (y < x) ++j;
        else { lab_push(w, cnt, x); ++i; ++j; }
        if (w.status != ST_OK) return 0;
    }
    if (cnt == na) return a;
    return lab_end(w, cnt);
}
// a - b; a itself if nothing is lost
MGX_DEV uint32_t lab_diff(Wave &w, uint32_t a, uint32_t b) {
    if (!a) return 0;
    if (!b) return a;
    const uint32_t na = lab_size(w, a), nb = lab_size(w, b);
    uint32_t cnt = 0, j = 0;
    for (uint32_t i = 0; i < na; ++i) {
        const uint32_t x = lab_at(w, a, i);
        while (j < nb && lab_at(w, b, j) < x) ++j;
        if (j < nb && lab_at(w, b, j) == x) continue;
        lab_push(w, cnt, x);
        if (w.status != ST_OK) return 0;
    }
    if (cnt == na) return a;
    return lab_end(w, cnt);
}
// utils::set_intersection_difference (common/algorithms.hpp:160-180): a & b and a - b
MGX_DEV void lab_isect_diff(Wave &w, uint32_t a, uint32_t b, uint32_t *isect, uint32_t *diff) {
    *isect = lab_isect(w, a, b);
    *diff = (w.status == ST_OK) ? lab_diff(w, a, b) : 0;
}

// LabeledExtender::flush's clear(): S, E, F of the column become ninf (aligner_labeled.cpp:92-100).  On the device a column
// is what it left in HBM (ColSlot): S (and F where something may reload it) plus one flag byte per cell that relates it to its
// parent; nothing reads a parent's E.  Every form is cleared in place: S / F read ninf, every flag byte 0.  A cleared column's
// children are cleared by the same flush (their parent has no labels), so the flags of columns that stay never refer to
// cleared values; the chain window in registers is never a cleared column (a flush runs at a fork — the window has been
// spilled and is the fork's parent, which keeps its labels if it gets children — or before backtracking).
MGX_DEV void lab_clear_column(Wave &w, int32_t idx) {
    const ColMeta c = uni_col(col_load(w, idx));
    w.col_lab[idx] = 0;
    if (col_compact(c)) {
        // one line: words 4 + 2 l = four flag bytes, 5 + 2 l = four 8-bit S (l < 6)
        uint32_t *m = (uint32_t *)(w.cols + idx);
        FOR_LANES(l) { if (l < 6) { gst(m + 4 + 2 * l, 0u); gst(m + 5 + 2 * l, 0x80808080u); } }
    } else if (col_chain(c)) {
        uint8_t *fl = w.cols[idx].flags;
        int16_t *srow = w.cols_s16 + (int64_t)idx * FWS;
        for (int32_t base = 0; base < FWS; base += WAVE) {
            FOR_LANES(l) { const int32_t j = base + l; if (j < FWS) { gst(fl + j, (uint8_t)0); gst(srow + j, S16_NINF); } }
        }
        if (c.cells != NO_CELLS) {                           // stayed behind in the frontier: its window as an S / F record
            int32_t *rec = w.cells + c.cells;
            for (int32_t base = 0; base < 2 * CHW; base += WAVE) {
                FOR_LANES(l) { const int32_t j = base + l; if (j < 2 * CHW) gst(rec + j, NINF); }
            }
        }
    } else {
        if (c.cells == NO_CELLS) { w.status = ST_CAPACITY; return; }       // cannot happen: a general column owns a record
        const int32_t wc = col_wc(c);
        int32_t *rec = w.cells + c.cells;
        uint8_t *fb = (uint8_t *)(rec + 2 * wc);
        for (int32_t base = 0; base < wc; base += WAVE) {
            FOR_LANES(l) {
                const int32_t j = base + l;
                if (j < wc) { gst(rec + j, NINF); gst(rec + wc + j, NINF); gst(fb + j, (uint8_t)0); }
            }
        }
    }
    for (int b = 0; b < 2; ++b) if (w.st[b].col == idx) w.st[b].col = -1;          // a staged copy is stale now
    wave_sync();
}

// LabeledExtender::flush (aligner_labeled.cpp:81-137): the columns added since the last flush inherited their parent's labels
// unseen ("annotations are preserved in unitigs"); now each is intersected with its node's labels, and a column left without
// labels is cleared
// Two passes: the columns' (node, parent) and the head words of their nodes' rows are gathered one column per lane into
// the backtracking's start-cell list (unused during an extension and before bt_begin fills it) — three dependent loads per
// column that would otherwise run one column after the other; the intersections then follow in table order.
MGX_DEV void lab_flush(Wave &w, int32_t tsize) {
    const AlignParams &P = MGX_PARAMS_OF(w);
    const int32_t first = w.last_flushed;
    if (first >= tsize) return;
    uint64_t *scr = (uint64_t *)w.indices;                   // per column: head word, node | parent << 32
    // The usual outcome is "nothing changes": the columns since the last flush lie on one path whose nodes all carry the labels
    // the path set out with.  Checked while gathering: every column still holds the set ph0 the first one inherited, so does its
    // parent, and its row equals the first column's row, which loses no label of ph0 — then every intersection below would
    // return ph0 again and the table order pass is skipped.
    const ColMeta c0 = uni_col(col_load(w, first));
    const uint32_t ph0 = w.col_lab[c0.parent];
    LV<int32_t> lines;
    LV<bool> differs;
    FOR_LANES(l) { lines[l] = 0; differs[l] = false; }
    uint64_t hd0 = 0;
    for (int32_t base = first; base < tsize; base += WAVE) {
        LV<uint64_t> hv;
        FOR_LANES(l) {
            const int32_t i = base + l;
            hv[l] = 0;
            if (i < tsize) {
                const ColMeta c = col_load(w, i);
                const uint32_t bn = lab_base_node(P, c.node);
                uint64_t h = 0;
                if (bn && bn <= P.g.n && (uint64_t)bn - 1 < P.anno_rows) {
                    bool real = true;
                    if (!(P.labeled & 2u)) { LineCtr lc = { 0, 0, 0 }; real = get_W(P.g, bn, lc) != 0; lines[l] += (int32_t)(lc.rank_lines + lc.select_lines + lc.bit_lines); }
                    if (real) h = gld(P.anno_head + ((uint64_t)bn - 1));
                }
                gst(scr + 2 * (i - first), h);
                gst(scr + 2 * (i - first) + 1, (uint64_t)bn | ((uint64_t)(uint32_t)c.parent << 32));
                hv[l] = h;
                if (!bn || gld(w.col_lab + i) != ph0 || gld(w.col_lab + c.parent) != ph0) differs[l] = true;
            }
        }
        if (base == first) hd0 = wave_bcast(hv, 0);
        FOR_LANES(l) { if (base + l < tsize && hv[l] != hd0) differs[l] = true; }
    }
    w.ctr.rank_lines += (uint32_t)wave_sum(lines);
    wave_sync();
    if (ph0 && (hd0 & 0xFFFF) && !wave_ballot(differs)) {
        LabRow r0;
        uint32_t cn0 = (uint32_t)(hd0 & 0xFFFF);
        if (cn0 == 0xFFFF) cn0 = P.anno_count[(uint64_t)lab_base_node(P, c0.node) - 1];
        r0.n = cn0; r0.one = (uint32_t)(hd0 >> 16); r0.more = cn0 >= 2 ? P.anno_more + (hd0 >> 16) : nullptr;
        const uint32_t lo_before = w.lab_lo;
        const uint32_t nh0 = lab_isect_row(w, ph0, r0);
        if (w.status != ST_OK) return;
        if (nh0 == ph0) { w.last_flushed = tsize; return; }
        w.lab_lo = lo_before;                                // (a strict subset: the pass below makes it again, once)
    }
    // (along a path the same (parent's set, row) pair repeats column after column: its intersection is computed once)
    uint32_t memo_ph = 0, memo_nh = 0;
    uint64_t memo_hd = ~0ull;
    for (; w.last_flushed < tsize; ++w.last_flushed) {
        const int32_t i = w.last_flushed;
        const uint64_t hd = scr[2 * (i - first)], np = scr[2 * (i - first) + 1];
        const uint32_t node = (uint32_t)np;
        const int32_t parent = (int32_t)(uint32_t)(np >> 32);
        const uint32_t ph = w.col_lab[parent];
        if (!ph) { lab_clear_column(w, i); if (w.status != ST_OK) return; continue; }
        if (!node) continue;
        if (ph == memo_ph && hd == memo_hd && memo_nh) {
            w.col_lab[i] = memo_nh;
            continue;
        }
        LabRow r;
        uint32_t cn = (uint32_t)(hd & 0xFFFF);
        if (cn == 0xFFFF) cn = P.anno_count[(uint64_t)node - 1];
        r.n = cn; r.one = (uint32_t)(hd >> 16); r.more = cn >= 2 ? P.anno_more + (hd >> 16) : nullptr;
        const uint32_t nh = lab_isect_row(w, ph, r);
        if (w.status != ST_OK) return;
        if (!nh) { lab_clear_column(w, i); if (w.status != ST_OK) return; }
        else w.col_lab[i] = nh;
        memo_ph = ph; memo_hd = hd; memo_nh = nh;
    }
}

// LabeledExtender::call_outgoing (aligner_labeled.cpp:176-302, the branch without coordinates) on the children of column i in
// Wave::out_*: an only child inherits the column's labels unseen; at a fork the table is flushed and every child keeps the
// labels it shares with the column — a child that shares none is dropped.  Returns the number of children left; their label
// sets are in Wave::out_lab.
MGX_DEV int lab_filter_children(Wave &w, int32_t i, int n_out) {
    if (n_out <= 0) return n_out;
    if (n_out == 1) { w.out_lab[0] = w.col_lab[i]; return 1; }
    lab_flush(w, w.x.tsize);
    if (w.status != ST_OK) return 0;
    const uint32_t ph = w.col_lab[i];
    if (!ph) return 0;
    int m = 0;
    for (int oi = 0; oi < n_out; ++oi) {
        const uint32_t next = w.out_nodes[oi];
        const LabRow r = lab_row(w, next);
        const uint32_t h = lab_isect_row(w, ph, r);
        if (w.status != ST_OK) return 0;
        if (!h) continue;
        w.out_nodes[m] = next; w.out_chars[m] = w.out_chars[oi]; w.out_scores[m] = w.out_scores[oi];
        w.out_lab[m] = h;
        ++m;
    }
    return m;
}
