// C-ABI over boss_files.hpp: the reference's `.dbg` and `.column.annodbg` files as inputs of mgx_graph_create /
// mgx_annotation_create_sparse (include/mgx.h, "files").  Host code; the two *_read entry points need no GPU.
#include <cstdarg>
#include <cstdio>
#include <new>
#include <set>

#include "../../include/mgx.h"
#include "boss_files.hpp"

extern "C" void mgx_set_last_error(const char *msg);     // mgx.hip

struct mgx_column_file {
    mgx::files::ColumnFile f;
};

namespace {
int ffail(int code, const char *fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mgx_set_last_error(buf);
    return code;
}
template <class F>
int guarded(const char *path, F &&f) {
    try {
        return f();
    } catch (const mgx::files::Unsupported &e) {
        return ffail(MGX_ERR_UNSUPPORTED, "%s: %s", path, e.what());
    } catch (const mgx::files::ParseError &e) {
        return ffail(MGX_ERR_INVALID, "%s: %s", path, e.what());
    } catch (const std::bad_alloc &) {
        return ffail(MGX_ERR_OOM, "%s: out of host memory", path);
    }
}
}  // namespace

extern "C" {

int mgx_boss_file_read(const char *path, mgx_boss_file *out) {
    if (!path || !out) return ffail(MGX_ERR_INVALID, "mgx_boss_file_read: bad arguments");
    memset(out, 0, sizeof(*out));
    return guarded(path, [&]() {
        auto *g = new mgx::files::BossFile(mgx::files::read_dbg(path));
        out->k = g->k; out->sigma = g->sigma; out->mode = g->mode; out->state = g->state; out->n_edges = g->n_edges;
        out->F = g->F.data(); out->W = g->W.data(); out->last = g->last.data();
        out->owner = g;
        return (int)MGX_OK;
    });
}

void mgx_boss_file_free(mgx_boss_file *f) {
    if (!f) return;
    delete (mgx::files::BossFile *)f->owner;
    memset(f, 0, sizeof(*f));
}

int mgx_edgemask_read(const char *path, uint32_t state, uint64_t n_edges, uint8_t *valid_out) {
    if (!path || !valid_out) return ffail(MGX_ERR_INVALID, "mgx_edgemask_read: bad arguments");
    return guarded(path, [&]() {
        const std::vector<uint8_t> buf = mgx::files::read_whole_file(path);
        const std::vector<uint8_t> v = mgx::files::parse_edgemask(buf.data(), buf.size(), state, n_edges);
        memcpy(valid_out, v.data(), v.size());
        return (int)MGX_OK;
    });
}

int mgx_graph_load_dbg(const char *path, int device, mgx_graph **out) {
    if (!path || !out) return ffail(MGX_ERR_INVALID, "mgx_graph_load_dbg: bad arguments");
    mgx_boss_file f;
    int rc = mgx_boss_file_read(path, &f);
    if (rc != MGX_OK) return rc;
    if (f.sigma != 5) {
        const uint32_t sigma = f.sigma;
        mgx_boss_file_free(&f);
        return ffail(MGX_ERR_UNSUPPORTED, "%s: alphabet of %u characters; the device index holds DNA ($ACGT) graphs only", path, sigma);
    }
    mgx_boss_view view;
    memset(&view, 0, sizeof(view));
    view.k = f.k; view.sigma = 5; view.n_edges = f.n_edges; view.W = f.W; view.last = f.last; view.F = f.F; view.mode = f.mode;
    rc = mgx_graph_create(&view, device, out);       // its message stays in mgx_last_error
    mgx_boss_file_free(&f);
    return rc;
}

int mgx_column_file_read(const char *const *paths, uint32_t n_paths, mgx_column_file **out) {
    if (!paths || !n_paths || !out) return ffail(MGX_ERR_INVALID, "mgx_column_file_read: bad arguments");
    *out = nullptr;
    auto *m = new (std::nothrow) mgx_column_file();
    if (!m) return ffail(MGX_ERR_OOM, "mgx_column_file_read: out of host memory");
    std::set<std::string> seen;
    for (uint32_t i = 0; i < n_paths; ++i) {
        if (!paths[i]) { delete m; return ffail(MGX_ERR_INVALID, "mgx_column_file_read: path %u is null", i); }
        const int rc = guarded(paths[i], [&]() {
            mgx::files::ColumnFile f = mgx::files::read_columns(paths[i]);
            // ColumnCompressed::merge_load (annotate_column_compressed.cpp:493-640): the files' columns side by side; every file
            // describes the same rows.  A label present in two files would have its columns OR-ed there; not read here.
            if (i && f.n_rows != m->f.n_rows) throw mgx::files::ParseError("annotates " + std::to_string(f.n_rows) + " rows, the files before it " + std::to_string(m->f.n_rows));
            m->f.n_rows = f.n_rows;
            for (size_t j = 0; j < f.labels.size(); ++j) {
                if (!seen.insert(f.labels[j]).second) throw mgx::files::Unsupported("label '" + f.labels[j] + "' occurs in two files");
                m->f.labels.push_back(f.labels[j]);
                m->f.rows.insert(m->f.rows.end(), f.rows.begin() + (std::ptrdiff_t)f.col_begin[j], f.rows.begin() + (std::ptrdiff_t)f.col_begin[j + 1]);
                m->f.col_begin.push_back(m->f.rows.size());
            }
            return (int)MGX_OK;
        });
        if (rc != MGX_OK) { delete m; return rc; }
    }
    *out = m;
    return MGX_OK;
}

void mgx_column_file_free(mgx_column_file *f) { delete f; }
uint64_t mgx_column_file_num_rows(const mgx_column_file *f) { return f ? f->f.n_rows : 0; }
uint32_t mgx_column_file_num_labels(const mgx_column_file *f) { return f ? (uint32_t)f->f.labels.size() : 0; }
const char *mgx_column_file_label(const mgx_column_file *f, uint32_t j) { return f && j < f->f.labels.size() ? f->f.labels[j].c_str() : nullptr; }
const uint64_t *mgx_column_file_col_begin(const mgx_column_file *f) { return f ? f->f.col_begin.data() : nullptr; }
const uint64_t *mgx_column_file_rows(const mgx_column_file *f) { return f ? f->f.rows.data() : nullptr; }

int mgx_annotation_create_from_file(const mgx_column_file *f, int device, mgx_annotation **out) {
    if (!f || !out) return ffail(MGX_ERR_INVALID, "mgx_annotation_create_from_file: bad arguments");
    return mgx_annotation_create_sparse(f->f.n_rows, (uint32_t)f->f.labels.size(), f->f.col_begin.data(), f->f.rows.data(), 0, device, out);
}

}  // extern "C"
