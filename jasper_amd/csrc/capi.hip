// capi.hip -- extern "C" surface of libjasper_hip.so (declared in include/jasper_hip.h)
#include "../../include/jasper_hip.h"
#include "ingest.hpp"
#include "polish.hpp"
#include "table.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace jk;

static thread_local std::string g_err;

struct jasper_table {
    Table t;
};

struct jasper_result {
    std::vector<std::string> seqs;
    std::vector<jasper_fixrec> recs;
    std::vector<std::string> aux;
    std::vector<int> status;
    int64_t qv[4] = {0, 0, 0, 0};
    uint64_t lookups = 0;
    double seconds = 0;
};

static_assert(sizeof(jasper_fixrec) == sizeof(FixRec), "public and device record layouts must match");

#define CHK(x)                                                        \
    do {                                                              \
        hipError_t e_ = (x);                                          \
        if (e_ != hipSuccess) {                                       \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);   \
            return JASPER_ERR;                                        \
        }                                                             \
    } while (0)

extern "C" {

const char *jasper_last_error(void) { return g_err.c_str(); }

int jasper_device_count(int *n) {
    CHK(hipGetDeviceCount(n));
    return JASPER_OK;
}

int jasper_table_create(int k, uint64_t min_slots, int device, jasper_table **out) {
    if (!out) { g_err = "null out"; return JASPER_ERR; }
    jasper_table *h = new jasper_table();
    int rc = h->t.init(k, min_slots, device, g_err);
    if (rc) { h->t.destroy(); delete h; return rc; }
    *out = h;
    return JASPER_OK;
}

void jasper_table_destroy(jasper_table *t) {
    if (!t) return;
    t->t.destroy();
    delete t;
}

int jasper_table_info(jasper_table *t, int *k, uint64_t *slots, uint64_t *distinct, uint64_t *occurrences) {
    if (t->t.read_stats(g_err)) return JASPER_ERR;
    if (k) *k = t->t.k;
    if (slots) *slots = t->t.nslots;
    if (distinct) *distinct = t->t.h_stats[ST_DISTINCT];
    if (occurrences) *occurrences = t->t.h_stats[ST_OCCURRENCES];
    return JASPER_OK;
}

int jasper_table_sync(jasper_table *t) {
    CHK(hipSetDevice(t->t.device));
    CHK(hipStreamSynchronize(t->t.stream));
    return JASPER_OK;
}

int jasper_table_clear(jasper_table *t) {
    CHK(hipSetDevice(t->t.device));
    CHK(hipMemsetAsync(t->t.d.slots, 0, t->t.nslots * 16, t->t.stream));
    CHK(hipMemsetAsync(t->t.d.stats, 0, ST_WORDS * sizeof(unsigned long long), t->t.stream));
    return JASPER_OK;
}

int jasper_count_bases(jasper_table *t, const char *bases, uint64_t n) {
    t->t.reset_timing();
    return t->t.count_host(bases, n, g_err);
}

int jasper_count_bases_device(jasper_table *t, const void *d_bases, uint64_t n) {
    t->t.reset_timing();
    return t->t.count_device((const uint8_t *)d_bases, n, g_err);
}

int jasper_count_reads_text(jasper_table *t, const char *text, uint64_t n) {
    t->t.reset_timing();
    Table *T = &t->t;
    FastxParser p([T](const char *b, size_t m) { return T->count_host(b, m, g_err); });
    int rc = p.feed(text, n);
    if (!rc) rc = p.finish();
    if (rc && !p.error().empty()) g_err = p.error();
    return rc;
}

int jasper_count_reads_files(jasper_table *t, const char *const *paths, int n_paths) {
    t->t.reset_timing();
    Table *T = &t->t;
    FastxParser p([T](const char *b, size_t m) { return T->count_host(b, m, g_err); });
    std::string e;
    int rc = parse_files(paths, n_paths, p, e);
    if (rc && !e.empty()) g_err = e;
    return rc;
}

int jasper_last_count_timing(jasper_table *t, double *kernel_ms, uint64_t *launches) {
    if (kernel_ms) *kernel_ms = t->t.count_kernel_ms;
    if (launches) *launches = t->t.count_launches;
    return JASPER_OK;
}

int jasper_histogram(jasper_table *t, uint64_t *out10002) { return t->t.histogram(out10002, g_err); }

int jasper_lookup(jasper_table *t, const char *chars, const int64_t *offsets, uint64_t n, uint32_t *out) {
    return t->t.lookup_strings(chars, offsets, n, out, g_err);
}

int jasper_table_export_device(jasper_table *t, uint64_t *n_entries, void **d_entries) {
    unsigned long long *p = nullptr;
    int rc = t->t.export_entries(n_entries, &p, g_err);
    if (rc) return rc;
    *d_entries = p;
    return JASPER_OK;
}

int jasper_table_export_to(jasper_table *t, void *d_dst, uint64_t cap_entries, uint64_t *n_entries) {
    unsigned long long *p = nullptr;
    uint64_t n = 0;
    int rc = t->t.export_entries(&n, &p, g_err);
    if (rc) return rc;
    if (n > cap_entries) { (void)hipFree(p); g_err = "export buffer too small"; return JASPER_ERR_CAPACITY; }
    if (n) CHK(hipMemcpy(d_dst, p, n * 24, hipMemcpyDeviceToDevice));
    CHK(hipFree(p));
    *n_entries = n;
    return JASPER_OK;
}

int jasper_table_import_device(jasper_table *t, const void *d_entries, uint64_t n_entries) {
    return t->t.import_entries((const unsigned long long *)d_entries, n_entries, g_err);
}

int jasper_device_free(jasper_table *t, void *d_ptr) {
    CHK(hipSetDevice(t->t.device));
    CHK(hipFree(d_ptr));
    return JASPER_OK;
}

int jasper_table_export(jasper_table *t, uint64_t *n_entries, uint64_t *host_entries) {
    if (!host_entries) {
        if (t->t.read_stats(g_err)) return JASPER_ERR;
        *n_entries = t->t.h_stats[ST_DISTINCT];
        return JASPER_OK;
    }
    void *d = nullptr;
    uint64_t n = 0;
    int rc = jasper_table_export_device(t, &n, &d);
    if (rc) return rc;
    if (n > *n_entries) { (void)hipFree(d); g_err = "export buffer too small"; return JASPER_ERR_CAPACITY; }
    CHK(hipMemcpy(host_entries, d, n * 24, hipMemcpyDeviceToHost));
    CHK(hipFree(d));
    *n_entries = n;
    return JASPER_OK;
}

int jasper_table_import(jasper_table *t, const uint64_t *host_entries, uint64_t n_entries) {
    if (!n_entries) return JASPER_OK;
    CHK(hipSetDevice(t->t.device));
    void *d = nullptr;
    CHK(hipMalloc(&d, n_entries * 24));
    CHK(hipMemcpy(d, host_entries, n_entries * 24, hipMemcpyHostToDevice));
    int rc = t->t.import_entries((const unsigned long long *)d, n_entries, g_err);
    (void)hipFree(d);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// polishing: layout of one batch in HBM
//   per chunk: gap buffer (len + slack), fix records, aux bytes, extension-search scratch
// ---------------------------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

int jasper_polish_batch(jasper_table *t, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre,
                        int passes, int fix, jasper_result **out) {
    Table &T = t->t;
    if (T.k < 6) { g_err = "polishing needs k >= 6"; return JASPER_ERR; }
    if (passes < 0 || passes > 200) { g_err = "bad number of passes"; return JASPER_ERR; }
    if (solid_thre < 0) { g_err = "bad threshold"; return JASPER_ERR; }
    CHK(hipSetDevice(T.device));
    jasper_result *R = new jasper_result();
    R->seqs.resize(n_chunks);
    R->aux.resize(n_chunks);
    R->status.assign(n_chunks, 0);
    *out = R;
    if (n_chunks == 0) return JASPER_OK;

    const int k = T.k;
    const uint32_t node_cap = 1u << 20, front_cap = 20480, patch_cap = 1u << 16;
    std::vector<ChunkDev> hc(n_chunks);
    std::vector<size_t> off_buf(n_chunks), off_rec(n_chunks), off_aux(n_chunks), off_nodes(n_chunks), off_front(n_chunks),
        off_patch(n_chunks);
    size_t total = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int64_t len = lens[c];
        ChunkDev &C = hc[c];
        memset(&C, 0, sizeof C);
        C.len = len;
        C.cap = len + std::max<int64_t>(4096, len / 8);
        C.gs = 0;
        C.glen = C.cap - len;
        C.rec_cap = (uint32_t)std::min<int64_t>(0x7fffffff, 2 * len / k + 64);
        C.aux_cap = (uint32_t)std::min<int64_t>(0x7fffffff, std::max<int64_t>(1 << 16, len / 4));
        C.node_cap = node_cap; C.front_cap = front_cap; C.patch_cap = patch_cap;
        off_buf[c] = total;   total += al256((size_t)C.cap + 16);
        off_rec[c] = total;   total += al256((size_t)C.rec_cap * sizeof(FixRec));
        off_aux[c] = total;   total += al256(C.aux_cap);
        off_nodes[c] = total; total += al256((size_t)node_cap * 4);
        off_front[c] = total; total += al256((size_t)front_cap * 80);
        off_patch[c] = total; total += al256(patch_cap);
    }
    uint8_t *arena = nullptr;
    ChunkDev *d_chunks = nullptr;
    hipError_t e = hipMalloc((void **)&arena, total);
    if (e != hipSuccess) { g_err = std::string("polish arena: ") + hipGetErrorString(e); return JASPER_ERR_CAPACITY; }
    CHK(hipMalloc((void **)&d_chunks, sizeof(ChunkDev) * n_chunks));
    for (int c = 0; c < n_chunks; ++c) {
        ChunkDev &C = hc[c];
        C.buf = arena + off_buf[c];
        C.recs = (FixRec *)(arena + off_rec[c]);
        C.aux = arena + off_aux[c];
        C.nodes = (uint32_t *)(arena + off_nodes[c]);
        C.front = arena + off_front[c];
        C.patch = arena + off_patch[c];
        // text sits right of the gap: buf[cap-len, cap)
        if (C.len) CHK(hipMemcpyAsync(C.buf + C.glen, seqs[c], (size_t)C.len, hipMemcpyHostToDevice, T.stream));
    }
    CHK(hipMemcpyAsync(d_chunks, hc.data(), sizeof(ChunkDev) * n_chunks, hipMemcpyHostToDevice, T.stream));

    PolishParams pp;
    pp.k = k;
    pp.step = std::max(2, (int)std::nearbyint((double)k / 8.0));   // src/jasper.py:20 (python round = half-to-even)
    pp.solid = (uint32_t)solid_thre;
    pp.passes = passes;
    pp.fix = fix ? 1 : 0;
    hipEvent_t ev0, ev1;
    CHK(hipEventCreate(&ev0));
    CHK(hipEventCreate(&ev1));
    CHK(hipEventRecord(ev0, T.stream));
    for (int pass = 0; pass <= passes; ++pass) {                    // src/jasper.py:25
        launch_polish_pass(T.d, d_chunks, n_chunks, pp, pass, T.stream);
        CHK(hipGetLastError());
    }
    CHK(hipEventRecord(ev1, T.stream));
    CHK(hipMemcpyAsync(hc.data(), d_chunks, sizeof(ChunkDev) * n_chunks, hipMemcpyDeviceToHost, T.stream));
    CHK(hipStreamSynchronize(T.stream));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, ev0, ev1));
    R->seconds = ms * 1e-3;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);

    // gather texts
    std::vector<int64_t> out_off(n_chunks);
    int64_t out_total = 0;
    for (int c = 0; c < n_chunks; ++c) { out_off[c] = out_total; out_total += hc[c].len; }
    uint8_t *d_out = nullptr;
    int64_t *d_off = nullptr;
    std::vector<char> h_out((size_t)out_total + 1);
    CHK(hipMalloc((void **)&d_out, (size_t)out_total + 16));
    CHK(hipMalloc((void **)&d_off, sizeof(int64_t) * n_chunks));
    CHK(hipMemcpyAsync(d_off, out_off.data(), sizeof(int64_t) * n_chunks, hipMemcpyHostToDevice, T.stream));
    launch_pack(d_chunks, n_chunks, d_out, d_off, T.stream);
    CHK(hipGetLastError());
    if (out_total) CHK(hipMemcpyAsync(h_out.data(), d_out, (size_t)out_total, hipMemcpyDeviceToHost, T.stream));
    int rc = JASPER_OK;
    size_t nrec_total = 0;
    for (int c = 0; c < n_chunks; ++c) nrec_total += std::min(hc[c].nrec, hc[c].rec_cap);
    R->recs.resize(nrec_total);
    size_t rpos = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const uint32_t nr = std::min(hc[c].nrec, hc[c].rec_cap);
        if (nr) CHK(hipMemcpyAsync(&R->recs[rpos], hc[c].recs, nr * sizeof(FixRec), hipMemcpyDeviceToHost, T.stream));
        rpos += nr;
        const uint32_t na = std::min(hc[c].naux, hc[c].aux_cap);
        R->aux[c].resize(na);
        if (na) CHK(hipMemcpyAsync(&R->aux[c][0], hc[c].aux, na, hipMemcpyDeviceToHost, T.stream));
    }
    CHK(hipStreamSynchronize(T.stream));
    for (int c = 0; c < n_chunks; ++c) {
        R->seqs[c].assign(h_out.data() + out_off[c], (size_t)hc[c].len);
        R->status[c] = hc[c].status;
        R->qv[0] += hc[c].wrong[0]; R->qv[1] += hc[c].total[0];
        R->qv[2] += hc[c].wrong[1]; R->qv[3] += hc[c].total[1];
        R->lookups += hc[c].lookups;
        if (hc[c].status != PS_OK && rc == JASPER_OK) {
            static const char *names[] = {"ok", "chunk grew beyond its slack", "fix-record buffer overflow", "aux buffer overflow",
                                          "path-extension scratch exhausted", "reference IndexError (src/jasper.py:221)",
                                          "trial string too long"};
            g_err = std::string("polish: chunk ") + std::to_string(c) + ": " + names[hc[c].status];
            rc = hc[c].status == PS_REF_INDEXERROR ? JASPER_ERR_REFERENCE_EXIT : JASPER_ERR_CAPACITY;
        }
    }
    (void)hipFree(d_out);
    (void)hipFree(d_off);
    (void)hipFree(d_chunks);
    (void)hipFree(arena);
    return rc;
}

int jasper_result_num_chunks(const jasper_result *r) { return (int)r->seqs.size(); }
int jasper_result_seq(const jasper_result *r, int chunk, const char **seq, int64_t *len) {
    if (chunk < 0 || chunk >= (int)r->seqs.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    *seq = r->seqs[chunk].data();
    *len = (int64_t)r->seqs[chunk].size();
    return JASPER_OK;
}
int jasper_result_records(const jasper_result *r, const jasper_fixrec **recs, uint64_t *n) {
    *recs = r->recs.data();
    *n = r->recs.size();
    return JASPER_OK;
}
int jasper_result_aux(const jasper_result *r, int chunk, const char **aux, uint64_t *n) {
    if (chunk < 0 || chunk >= (int)r->aux.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    *aux = r->aux[chunk].data();
    *n = r->aux[chunk].size();
    return JASPER_OK;
}
int jasper_result_qv(const jasper_result *r, int64_t out4[4]) {
    for (int i = 0; i < 4; ++i) out4[i] = r->qv[i];
    return JASPER_OK;
}
int jasper_result_lookups(const jasper_result *r, uint64_t *n) { *n = r->lookups; return JASPER_OK; }
double jasper_result_seconds(const jasper_result *r) { return r->seconds; }
void jasper_result_free(jasper_result *r) { delete r; }

}  // extern "C"
