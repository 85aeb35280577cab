// capi.hip -- extern "C" surface of libjasper_hip.so (declared in include/jasper_hip.h)
#include "../../include/jasper_hip.h"
#include "ingest.hpp"
#include "polish_host.hpp"
#include "table.hpp"
#include "pgunzip.hpp"
#include "asmio.hpp"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace jk;

static thread_local std::string g_err;
std::string &jasper_err_ref() { return g_err; }

struct jasper_result;
struct jasper_table {
    Table t;
    jasper_result *pending = nullptr;   // result whose polished text still lies in this table's workspace
};

struct jasper_result {
    jasper_table *owner = nullptr;       // non-null while the text is on the device
    std::vector<const uint8_t *> d_seqs;
    std::vector<int64_t> d_lens;
    std::vector<std::string> seqs;
    std::vector<jasper_fixrec> recs;
    std::vector<std::string> aux;
    int64_t qv[4] = {0, 0, 0, 0};
    std::vector<int64_t> qv_chunk;
    uint64_t lookups = 0;
    double seconds = 0;
    uint64_t n_segments = 0, n_respeculated = 0;
    int retried = 0;                     // the batch was repeated with 8x the room because a bound was exceeded
};

static_assert(sizeof(jasper_fixrec) == sizeof(FixRec), "public and device record layouts must match");

// copy a device-resident result's text to the host (before its workspace is reused, or when the caller asks for it)
static int result_fetch(jasper_result *r) {
    if (!r->owner) return JASPER_OK;
    jasper_table *t = r->owner;
    Table &T = t->t;
    hipError_t e = hipSetDevice(T.device);
    for (size_t c = 0; c < r->d_seqs.size() && e == hipSuccess; ++c) {
        r->seqs[c].resize((size_t)r->d_lens[c]);
        if (r->d_lens[c]) e = hipMemcpyAsync(&r->seqs[c][0], r->d_seqs[c], (size_t)r->d_lens[c], hipMemcpyDeviceToHost, T.stream);
    }
    if (e == hipSuccess) e = jk_stream_wait(T.stream);
    if (t->pending == r) t->pending = nullptr;
    r->owner = nullptr;
    if (e != hipSuccess) { g_err = std::string("fetching polished text: ") + hipGetErrorString(e); return JASPER_ERR; }
    return JASPER_OK;
}

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

int jasper_request_cancel(int on) { jk::g_cancel.store(on ? 1 : 0); return JASPER_OK; }

int jasper_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMemGetInfo(&f, &t);
    if (e != hipSuccess) { g_err = std::string("hipMemGetInfo: ") + hipGetErrorString(e); return JASPER_ERR; }
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
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

int jasper_table_load_jf(const char *path, int device, jasper_table **out) {
    if (!out || !path) { g_err = "null argument"; return JASPER_ERR; }
    JfHeader h;
    int rc = jf_read_header(path, h, g_err);
    if (rc) return rc == -3 ? JASPER_ERR_FORMAT : JASPER_ERR;
    jasper_table *t = nullptr;
    rc = jasper_table_create(h.key_len / 2, std::max<uint64_t>(1u << 16, 2 * h.n_records), device, &t);
    if (rc) return rc;
    Table *T = &t->t;
    rc = T->load_jf_records(path, h.data_offset, h.n_records, h.key_len, h.counter_len, g_err);
    if (rc) { jasper_table_destroy(t); return rc < -1 ? rc : JASPER_ERR; }
    *out = t;
    return JASPER_OK;
}

int jasper_table_load_jf_part(const char *path, int device, uint32_t part, uint32_t nparts, jasper_table **out) {
    if (!out || !path || nparts == 0 || part >= nparts) { g_err = "bad argument"; return JASPER_ERR; }
    JfHeader h;
    int rc = jf_read_header(path, h, g_err);
    if (rc) return rc == -3 ? JASPER_ERR_FORMAT : JASPER_ERR;
    const uint64_t lo = h.n_records / nparts * part + std::min<uint64_t>(part, h.n_records % nparts);
    const uint64_t n = h.n_records / nparts + (part < h.n_records % nparts ? 1 : 0);
    const uint64_t rec = (uint64_t)((h.key_len + 7) / 8) + (uint64_t)h.counter_len;
    jasper_table *t = nullptr;
    rc = jasper_table_create(h.key_len / 2, std::max<uint64_t>(1u << 16, 2 * n), device, &t);
    if (rc) return rc;
    rc = t->t.load_jf_records(path, h.data_offset + lo * rec, n, h.key_len, h.counter_len, g_err);
    if (rc) { jasper_table_destroy(t); return rc < -1 ? rc : JASPER_ERR; }
    *out = t;
    return JASPER_OK;
}

int jasper_table_write_jf(jasper_table *t, const char *path, const char *const *cmdline, int n_cmdline) {
    if (!t || !path || n_cmdline < 0 || (n_cmdline && !cmdline)) { g_err = "null argument"; return JASPER_ERR; }
    return t->t.write_jf(path, cmdline, n_cmdline, g_err) ? JASPER_ERR : JASPER_OK;
}

int jasper_debug_mix(int k, int inverse, uint64_t hi, uint64_t lo, uint64_t out2[2]) {
    if (k < 1 || k > 64 || !out2) { g_err = "bad arguments"; return JASPER_ERR; }
    const u128 r = inverse ? unmix(mk(hi, lo), 2 * k) : mix(mk(hi, lo), 2 * k);
    out2[0] = r.hi;
    out2[1] = r.lo;
    return JASPER_OK;
}

void jasper_table_destroy(jasper_table *t) {
    if (!t) return;
    if (t->pending) (void)result_fetch(t->pending);   // a live result must not lose its text with the table
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
    CHK(jk_stream_wait(t->t.stream));
    return JASPER_OK;
}

int jasper_table_clear(jasper_table *t) { return t->t.clear(g_err) ? JASPER_ERR : JASPER_OK; }

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

int jasper_count_reads_file_ranges(jasper_table *t, const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths) {
    if (!begins || !ends) { g_err = "null range arrays"; return JASPER_ERR; }
    t->t.reset_timing();
    Table *T = &t->t;
    T->ingest_gpu_bytes = T->ingest_host_bytes = 0;
    T->ingest_begin = begins;
    T->ingest_end = ends;
    const int rc = T->count_files_gpu(paths, n_paths, &T->ingest_gpu_bytes, &T->ingest_host_bytes, g_err);
    T->ingest_begin = T->ingest_end = nullptr;
    return rc < -1 ? rc : (rc ? JASPER_ERR : JASPER_OK);
}

int jasper_count_reads_files(jasper_table *t, const char *const *paths, int n_paths) {
    t->t.reset_timing();
    Table *T = &t->t;
    T->ingest_gpu_bytes = T->ingest_host_bytes = 0;
    if (!getenv("JASPER_INGEST_HOST")) {       // text parsed on the GPU; the host state machine takes over whatever is not plain 4-line FASTQ / FASTA
        const int rc = T->count_files_gpu(paths, n_paths, &T->ingest_gpu_bytes, &T->ingest_host_bytes, g_err);
        return rc < -1 ? rc : (rc ? JASPER_ERR : JASPER_OK);
    }
    FastxParser p([T](const char *b, size_t m) { return T->count_host(b, m, g_err); });
    std::string e;
    int rc = parse_files(paths, n_paths, p, e);
    if (rc && !e.empty()) g_err = e;
    return rc;
}

int jasper_last_ingest(jasper_table *t, uint64_t *gpu_bytes, uint64_t *host_bytes) {
    if (gpu_bytes) *gpu_bytes = t->t.ingest_gpu_bytes;
    if (host_bytes) *host_bytes = t->t.ingest_host_bytes;
    return JASPER_OK;
}

int jasper_last_count_timing(jasper_table *t, double *kernel_ms, uint64_t *launches) {
    if (kernel_ms) *kernel_ms = t->t.count_kernel_ms;
    if (launches) *launches = t->t.count_launches;
    return JASPER_OK;
}

int jasper_last_count_stages(jasper_table *t, double stage_ms[8], uint64_t *partitioned_launches, int *path) {
    for (int i = 0; i < 8; ++i) stage_ms[i] = t->t.part_stage_ms[i];
    if (partitioned_launches) *partitioned_launches = t->t.count_partitioned_launches;
    if (path) *path = t->t.count_path;
    return JASPER_OK;
}

int jasper_histogram_part(jasper_table *t, uint32_t part, uint32_t nparts, uint64_t *out10002) {
    if (nparts == 0 || part >= nparts || !out10002) { g_err = "bad partition"; return JASPER_ERR; }
    return t->t.histogram_part(part, nparts, out10002, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_histogram_is_fused(jasper_table *t) { return t && t->t.histo_cached ? 1 : 0; }

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

int jasper_table_export_packed(jasper_table *t, void *d_dst, uint64_t cap_entries, uint64_t *n_entries, uint32_t part, uint32_t nparts) {
    if (nparts == 0 || part >= nparts) { g_err = "bad partition"; return JASPER_ERR; }
    return t->t.export_packed(d_dst, cap_entries, n_entries, part, nparts, g_err);
}
int jasper_table_import_packed(jasper_table *t, const void *d_src, uint64_t n_entries, int mode) {
    return t->t.import_packed(d_src, n_entries, mode, g_err);
}
int jasper_table_import_packed_multi(jasper_table *t, const void *const *d_srcs, const uint64_t *counts, uint32_t n_src) {
    if (!d_srcs || !counts) { g_err = "null argument"; return JASPER_ERR; }
    return t->t.import_packed_multi(d_srcs, counts, n_src, g_err);
}
int jasper_table_reserve(jasper_table *t, uint64_t min_slots) { return t->t.reserve(min_slots, g_err); }

int jasper_table_fit(jasper_table *t, double max_load) { return t->t.fit(max_load, g_err); }
int jasper_table_export_owner(jasper_table *t, void *d_dst, uint64_t cap_entries, uint32_t n_owners, uint64_t *counts) {
    if (!counts) { g_err = "counts is null"; return JASPER_ERR; }
    return t->t.export_owner(d_dst, cap_entries, n_owners, 0, counts, g_err);
}
int jasper_read_feed_start(jasper_table *t, const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths) {
    if (!t || n_paths < 0 || (n_paths && !paths) || ((begins == nullptr) != (ends == nullptr))) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.feed_start(paths, begins, ends, n_paths, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_read_feed_next(jasper_table *t, const void **d_bases, uint64_t *n) {
    if (!t || !d_bases || !n) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.feed_next(d_bases, n, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_read_feed_release(jasper_table *t) {
    if (!t) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.feed_release(g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_count_exchange_plan(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, uint64_t *out8) {
    if (!t || !out8) { g_err = "bad argument"; return JASPER_ERR; }
    const int rc = t->t.xchg_plan(piece_max, records_max, n_owners, out8, g_err);
    return rc == 0 ? JASPER_OK : rc == 1 ? 1 : JASPER_ERR;
}
int jasper_count_exchange_scan(jasper_table *t, const void *d_bases, uint64_t n, uint64_t pos, uint64_t end, uint64_t piece_max, uint32_t n_owners, void *d_deferred,
                               uint64_t deferred_cap, uint64_t *records) {
    if (!t || !d_deferred || !records || (n && !d_bases)) { g_err = "bad argument"; return JASPER_ERR; }
    if (pos == 0) t->t.reset_timing();                       // (the first round of a call: stage times are per call, like the other count entry points)
    return t->t.xchg_scan((const uint8_t *)d_bases, n, pos, end, piece_max, n_owners, d_deferred, deferred_cap, records, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_count_exchange_partition(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, void *d_send, void *d_send_counts, void *d_deferred,
                                    uint64_t deferred_cap) {
    if (!t || !d_send || !d_send_counts || !d_deferred) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.xchg_partition(piece_max, records_max, n_owners, d_send, d_send_counts, d_deferred, deferred_cap, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_count_exchange_dedupe(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, void *d_send, void *d_send_counts, uint32_t *max_fill,
                                 int *count_bits) {
    if (!t || !d_send || !d_send_counts || !max_fill || !count_bits) { g_err = "bad argument"; return JASPER_ERR; }
    const int rc = t->t.xchg_dedupe(piece_max, records_max, n_owners, d_send, d_send_counts, max_fill, count_bits, g_err);
    return rc == 0 ? JASPER_OK : rc == 1 ? 1 : JASPER_ERR;
}
int jasper_count_exchange_insert(jasper_table *t, const void *d_recv, const void *d_recv_counts, uint64_t piece_max, uint64_t records_max, uint32_t n_owners,
                                 uint32_t self, const void *d_deferred_all, uint64_t n_deferred_all, int whole_input, uint32_t slice_cap, int count_bits) {
    if (!t || !d_recv || !d_recv_counts || self >= n_owners || (n_deferred_all && !d_deferred_all) || count_bits < 0) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.xchg_insert(d_recv, d_recv_counts, piece_max, records_max, n_owners, self, d_deferred_all, n_deferred_all, whole_input, slice_cap, count_bits, g_err)
               ? JASPER_ERR : JASPER_OK;
}
int jasper_table_export_file_ranges(jasper_table *t, void *d_dst, uint64_t cap_entries, uint32_t n_ranges, int size_log2, uint64_t *counts) {
    if (!counts || size_log2 < 1) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.export_owner(d_dst, cap_entries, n_ranges, size_log2, counts, g_err);
}
int jasper_table_write_jf_piece(jasper_table *t, const char *path, const char *const *cmdline, int n_cmdline, int size_log2, int what) {
    if (!t || !path || n_cmdline < 0 || (n_cmdline && !cmdline) || what < 0 || what > 2 || size_log2 < 1) { g_err = "bad argument"; return JASPER_ERR; }
    return t->t.write_jf(path, cmdline, n_cmdline, g_err, size_log2, what) ? JASPER_ERR : JASPER_OK;
}
int jasper_table_ipc_handle(jasper_table *t, void *out64) { return t->t.ipc_handle(out64, g_err) ? JASPER_ERR : JASPER_OK; }
int jasper_table_attach_ipc(jasper_table *t, const void *handles, uint32_t n, uint32_t self) {
    return t->t.attach_ipc(handles, n, self, g_err) ? JASPER_ERR : JASPER_OK;
}
int jasper_table_attach_tables(jasper_table *t, jasper_table *const *shards, uint32_t n, uint32_t self) {
    if (n < 1 || n > MAX_SHARDS || !shards) { g_err = "attach: 1..8 shards"; return JASPER_ERR; }
    Table *p[MAX_SHARDS] = {};
    for (uint32_t i = 0; i < n; ++i) p[i] = shards[i] ? &shards[i]->t : nullptr;
    return t->t.attach_tables(p, n, self, g_err) ? JASPER_ERR : JASPER_OK;
}
// Can this process map these slot arrays at all?  Meant to be called from a THROW-AWAY process with a time limit: a mapping
// call that never returns (seen on this stack for one allocation size) then costs a killed helper, not a hung rank.
int jasper_table_release_retired(jasper_table *t) {
    if (!t) { g_err = "bad argument"; return JASPER_ERR; }
    t->t.release_retired();
    return JASPER_OK;
}

int jasper_inflate_file(const char *path, int threads, uint64_t chunk_bytes, const char *out_path, uint64_t *n_out, int *parallel) {
    if (!path) { g_err = "bad arguments"; return JASPER_ERR; }
    FILE *f = nullptr;
    if (out_path) { f = fopen(out_path, "wb"); if (!f) { g_err = std::string("cannot write ") + out_path; return JASPER_ERR; } }
    uint64_t total = 0;
    int par = 0, rc = JASPER_OK;
    size_t fsz = 0;
    { struct stat st; if (stat(path, &st) == 0) fsz = (size_t)st.st_size; }
    jk::ParallelGunzip pg(path, threads, chunk_bytes ? (size_t)chunk_bytes : jk::ParallelGunzip::chunk_for(fsz, threads));
    if (threads >= 2 && pg.open()) {
        par = 1;
        std::vector<std::vector<uint8_t>> pieces;
        while (pg.next(pieces))
            for (auto &pc : pieces) { total += pc.size(); if (f && !pc.empty() && fwrite(pc.data(), 1, pc.size(), f) != pc.size()) { g_err = "write error"; rc = JASPER_ERR; } }
        if (!pg.error().empty()) { g_err = pg.error(); rc = JASPER_ERR; }
    } else {
        gzFile g = gzopen(path, "rb");
        if (!g) { g_err = std::string("cannot open ") + path; rc = JASPER_ERR; }
        else {
            std::vector<char> buf(4u << 20);
            for (;;) {
                const int r = gzread(g, buf.data(), (unsigned)buf.size());
                if (r < 0) { g_err = std::string("read error in ") + path; rc = JASPER_ERR; break; }
                if (r == 0) break;
                total += (uint64_t)r;
                if (f && fwrite(buf.data(), 1, (size_t)r, f) != (size_t)r) { g_err = "write error"; rc = JASPER_ERR; break; }
            }
            gzclose(g);
        }
    }
    if (f) fclose(f);
    if (n_out) *n_out = total;
    if (parallel) *parallel = par;
    return rc;
}

int jasper_ipc_probe(int device, const void *handles, uint32_t n, uint32_t self) {
    if (!handles || n < 1 || n > MAX_SHARDS || self >= n) { g_err = "probe: 1..8 handles, self among them"; return JASPER_ERR; }
    CHK(hipSetDevice(device));
    for (uint32_t i = 0; i < n; ++i) {
        if (i == self) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * (size_t)i, 64);
        void *p = nullptr;
        CHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device != device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device, at.device) != hipSuccess || !can) { (void)hipIpcCloseMemHandle(p); g_err = "no peer access"; return JASPER_ERR; }
        }
        CHK(hipIpcCloseMemHandle(p));
    }
    return JASPER_OK;
}

int jasper_table_detach(jasper_table *t) {
    if (hipSetDevice(t->t.device) != hipSuccess) { g_err = "hipSetDevice"; return JASPER_ERR; }
    (void)jk_stream_wait(t->t.stream);
    t->t.detach_shards();
    return JASPER_OK;
}
uint32_t jasper_owner_of(uint64_t hash_lo, uint64_t hash_hi, uint32_t n) { return owner_of(mk(hash_hi, hash_lo), n); }

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
// polishing (host orchestration in polish_host.hip)
// ---------------------------------------------------------------------------------------------------
static int polish_common(jasper_table *t, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre, int passes, int fix,
                         bool device_io, jasper_result **out, int keep = -1) {
    const bool device_in = device_io, keep_on_device = keep < 0 ? device_io : keep != 0;
    Table &T = t->t;
    if (T.k < 6) { g_err = "polishing needs k >= 6"; return JASPER_ERR; }
    if (passes < 0 || passes > 200) { g_err = "bad number of passes"; return JASPER_ERR; }
    if (solid_thre < 0) { g_err = "bad threshold"; return JASPER_ERR; }
    if (n_chunks < 0 || !out) { g_err = "bad arguments"; return JASPER_ERR; }
    if (t->pending && result_fetch(t->pending)) return JASPER_ERR;     // its text is about to be overwritten
    jasper_result *R = new jasper_result();
    *out = R;
    PolishOut po;
    int rc = run_polish(T, n_chunks, seqs, lens, solid_thre, passes, fix, po, g_err, device_in, keep_on_device, getenv("JASPER_POLISH_ROOMY") ? 1 : 0);      // (tests: the roomy sizes at once)
    if (rc == -2) {                      // a slack / record / scratch bound was too small for this input: once more with 8x the room
        R->retried = 1;
        po = PolishOut();
        rc = run_polish(T, n_chunks, seqs, lens, solid_thre, passes, fix, po, g_err, device_in, keep_on_device, 1);
    }
    R->seqs.swap(po.seqs);
    R->aux.swap(po.aux);
    R->recs.resize(po.recs.size());
    if (!po.recs.empty()) memcpy(R->recs.data(), po.recs.data(), po.recs.size() * sizeof(FixRec));
    for (int i = 0; i < 4; ++i) R->qv[i] = po.qv[i];
    R->qv_chunk.swap(po.qv_chunk);
    R->lookups = po.lookups;
    R->seconds = po.seconds;
    R->n_segments = po.n_segments;
    R->n_respeculated = po.n_respeculated;
    if (rc == 0 && keep_on_device) {
        R->d_seqs.swap(po.d_seqs);
        R->d_lens.swap(po.d_lens);
        R->owner = t;
        t->pending = R;
    }
    if (rc == -4) return JASPER_ERR_REFERENCE_EXIT;
    if (rc == -2) return JASPER_ERR_CAPACITY;
    return rc ? JASPER_ERR : JASPER_OK;
}

int jasper_polish_batch(jasper_table *t, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre,
                        int passes, int fix, jasper_result **out) {
    return polish_common(t, n_chunks, seqs, lens, solid_thre, passes, fix, false, out);
}

int jasper_polish_batch_device(jasper_table *t, int n_chunks, const void *d_text, const int64_t *offsets, int solid_thre,
                               int passes, int fix, jasper_result **out) {
    if (n_chunks < 0 || (n_chunks && (!d_text || !offsets))) { g_err = "bad arguments"; return JASPER_ERR; }
    std::vector<const char *> ptrs((size_t)n_chunks);
    std::vector<int64_t> lens((size_t)n_chunks);
    for (int c = 0; c < n_chunks; ++c) {
        if (offsets[c + 1] < offsets[c]) { g_err = "offsets must not decrease"; return JASPER_ERR; }
        ptrs[c] = (const char *)d_text + offsets[c];
        lens[c] = offsets[c + 1] - offsets[c];
    }
    return polish_common(t, n_chunks, ptrs.data(), lens.data(), solid_thre, passes, fix, true, out);
}

// the chunk records of the listed batch files of a split assembly (asmio.cpp), straight from the job's arena; result chunk i is
// the i-th record of the files in list order
static int asm_records(const jasper_asm *a, const uint32_t *files, uint32_t n_files, std::vector<size_t> &recs) {
    if (!a || (n_files && !files)) { g_err = "bad arguments"; return JASPER_ERR; }
    for (uint32_t i = 0; i < n_files; ++i) {
        if ((size_t)files[i] + 1 >= a->file_first.size()) { g_err = "batch file out of range"; return JASPER_ERR; }
        for (size_t c = a->file_first[files[i]]; c < a->file_first[files[i] + 1]; ++c) recs.push_back(c);
    }
    return JASPER_OK;
}

int jasper_asm_polish(jasper_table *t, jasper_asm *a, const uint32_t *files, uint32_t n_files, int solid_thre, int passes, int fix, jasper_result **out) {
    std::vector<size_t> recs;
    if (int rc = asm_records(a, files, n_files, recs)) return rc;
    if (recs.size() > (size_t)INT32_MAX) { g_err = "too many chunk records for one call"; return JASPER_ERR; }
    std::vector<const char *> ptrs(recs.size());
    std::vector<int64_t> lens(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        const AsmChunk &ch = a->chunks[recs[i]];
        ptrs[i] = (const char *)a->arena + a->contigs[ch.contig].seq_off + ch.ci;
        lens[i] = (int64_t)ch.len;
    }
    // host text in; the polished text stays in HBM until jasper_asm_take copies it -- into the job's pinned buffer when there is one
    return polish_common(t, (int)recs.size(), ptrs.data(), lens.data(), solid_thre, passes, fix, false, out, 1);
}

static void asm_gpu_release(jasper_asm *a) {
    if (a->arena_registered) { (void)hipHostUnregister(a->arena); a->arena_registered = false; }
    if (a->out_pinned) { (void)hipHostFree(a->out_pinned); a->out_pinned = nullptr; a->out_cap = 0; }
}

// (any thread, once: blocks while the GPU runtime starts -- the caller runs it beside the counting)
int jasper_asm_pin(jasper_asm *a, int device) {
    if (!a) { g_err = "bad arguments"; return JASPER_ERR; }
    CHK(hipSetDevice(device));
    a->gpu_release = asm_gpu_release;
    if (!a->arena_registered && a->arena && a->arena_cap)
        a->arena_registered = hipHostRegister(a->arena, a->arena_cap, hipHostRegisterDefault) == hipSuccess;      // (refused: the copies are staged, as before)
    (void)hipGetLastError();
    if (!a->out_pinned) {
        const size_t own = a->own_bases ? (size_t)a->own_bases : a->arena_len;      // (one GPU of several: its own batch files' records only)
        const size_t want = own + own / 64 + (1u << 20);
        if (hipHostMalloc((void **)&a->out_pinned, want, hipHostMallocDefault) == hipSuccess) a->out_cap = want;
        else { a->out_pinned = nullptr; (void)hipGetLastError(); }
    }
    return JASPER_OK;
}

int jasper_asm_take(jasper_asm *a, jasper_result *r, const uint32_t *files, uint32_t n_files) {
    std::vector<size_t> recs;
    if (int rc = asm_records(a, files, n_files, recs)) return rc;
    if (!r || r->seqs.size() != recs.size()) { g_err = "the result is not the one of these batch files"; return JASPER_ERR; }
    if (r->owner) {
        // still on the device: one copy per record into the job's pinned buffer (after what earlier calls put there); a buffer
        // that is missing or too small -> through the result's own host strings
        size_t total = 0;
        for (size_t i = 0; i < recs.size(); ++i) total += (size_t)r->d_lens[i];
        if (a->out_pinned && a->out_used + total <= a->out_cap) {
            Table &T = r->owner->t;
            CHK(hipSetDevice(T.device));
            size_t at = a->out_used;
            for (size_t i = 0; i < recs.size(); ++i) {
                const size_t n = (size_t)r->d_lens[i];
                if (n) CHK(hipMemcpyAsync(a->out_pinned + at, r->d_seqs[i], n, hipMemcpyDeviceToHost, T.stream));
                jasper_asm::Polished &P = a->polished[recs[i]];
                P.p = a->out_pinned + at;
                P.n = n;
                P.own.clear();
                at += n;
            }
            CHK(jk_stream_wait(T.stream));
            a->out_used = at;
            for (size_t i = 0; i < recs.size(); ++i) a->have[recs[i]] = 1;
            if (r->owner->pending == r) r->owner->pending = nullptr;
            r->owner = nullptr;              // (the text has left the device with the job: the result keeps records, counters, aux)
            return JASPER_OK;
        }
        if (result_fetch(r)) return JASPER_ERR;
    }
    for (size_t i = 0; i < recs.size(); ++i) {
        a->polished[recs[i]].p = nullptr;
        a->polished[recs[i]].own.swap(r->seqs[i]);
        a->have[recs[i]] = 1;
    }
    return JASPER_OK;
}

int jasper_result_num_chunks(const jasper_result *r) { return (int)r->seqs.size(); }
int jasper_result_seq(const jasper_result *r, int chunk, const char **seq, int64_t *len) {
    if (chunk < 0 || chunk >= (int)r->seqs.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    if (r->owner && result_fetch(const_cast<jasper_result *>(r))) return JASPER_ERR;
    *seq = r->seqs[chunk].data();
    *len = (int64_t)r->seqs[chunk].size();
    return JASPER_OK;
}
int jasper_result_seq_len(const jasper_result *r, int chunk, int64_t *len) {
    if (chunk < 0 || chunk >= (int)r->seqs.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    *len = r->owner ? r->d_lens[chunk] : (int64_t)r->seqs[chunk].size();
    return JASPER_OK;
}
int jasper_result_seq_device(const jasper_result *r, int chunk, const void **d_seq, int64_t *len) {
    if (chunk < 0 || chunk >= (int)r->seqs.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    if (!r->owner) { g_err = "the polished text of this result is no longer on the device"; return JASPER_ERR; }
    *d_seq = r->d_seqs[chunk];
    *len = r->d_lens[chunk];
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
int jasper_result_segments(const jasper_result *r, uint64_t *n_segments, uint64_t *n_respeculated) {
    if (n_segments) *n_segments = r->n_segments;
    if (n_respeculated) *n_respeculated = r->n_respeculated;
    return JASPER_OK;
}
int jasper_result_qv_chunk(const jasper_result *r, int chunk, int64_t out4[4]) {
    if (chunk < 0 || (size_t)(4 * chunk + 3) >= r->qv_chunk.size()) { g_err = "chunk out of range"; return JASPER_ERR; }
    for (int i = 0; i < 4; ++i) out4[i] = r->qv_chunk[4 * (size_t)chunk + i];
    return JASPER_OK;
}
int jasper_result_retried(const jasper_result *r) { return r ? r->retried : 0; }
void jasper_result_free(jasper_result *r) {
    if (r && r->owner && r->owner->pending == r) r->owner->pending = nullptr;
    delete r;
}

}  // extern "C"
