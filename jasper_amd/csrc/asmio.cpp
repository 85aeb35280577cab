// asmio.cpp -- see asmio.hpp.  Host only: no GPU call in this file (jasper_asm_polish, which hands chunk records to the polisher,
// is in capi.hip).
#include "asmio.hpp"
#include "../../include/jasper_hip.h"
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <unordered_set>

namespace {

int clamp_threads(int threads) {
    if (threads <= 0) {
        unsigned hw = std::thread::hardware_concurrency();
        threads = (int)std::max(4u, std::min(16u, hw ? hw / 4u : 8u));
    }
    return std::min(threads, 64);
}

// fn(task) for task in [0, n) on up to `threads` threads (the caller's included); tasks are handed out one by one
void parallel_for(size_t n, int threads, const std::function<void(size_t)> &fn) {
    if (n == 0) return;
    threads = (int)std::min<size_t>((size_t)std::max(1, threads), n);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            fn(i);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < threads; ++i) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

struct Mapped {
    int fd = -1;
    const uint8_t *p = nullptr;
    size_t n = 0;
    ~Mapped() {
        if (p && n) munmap(const_cast<uint8_t *>(p), n);
        if (fd >= 0) close(fd);
    }
};

// bytes of [p, p + n) that are NOT printable ASCII without blanks (0x21 .. 0x7e) and not '\n', and the number of '\n'
// (plain loops the compiler turns into vector code)
inline void body_stats(const uint8_t *p, size_t n, size_t &newlines, size_t &odd) {
    size_t nl = 0, od = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = p[i];
        nl += c == '\n';
        od += (uint8_t)(c - 0x21) > 0x5d;
    }
    newlines += nl;
    odd += od - nl;
}

bool write_all(int fd, const void *buf, size_t n, off_t at) {
    const char *p = (const char *)buf;
    while (n) {
        const ssize_t k = pwrite(fd, p, n, at);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        at += k;
        n -= (size_t)k;
    }
    return true;
}

bool writev_all(int fd, struct iovec *iov, int cnt) {
    while (cnt) {
        ssize_t k = writev(fd, iov, std::min(cnt, 1024));
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        while (cnt && (size_t)k >= iov->iov_len) {
            k -= (ssize_t)iov->iov_len;
            ++iov;
            --cnt;
        }
        if (cnt && k) {
            iov->iov_base = (char *)iov->iov_base + k;
            iov->iov_len -= (size_t)k;
        }
    }
    return true;
}

const size_t UNIT = 4u << 20;      // bytes of file per unit of parallel work

}  // namespace

jasper_asm::~jasper_asm() {
    if (writer_running) writer.join();
    if (gpu_release) gpu_release(this);
    if (arena && arena_cap) munmap(arena, arena_cap);
}

extern "C" {

int jasper_asm_open(const char *path, int threads, jasper_asm **out) {
    std::string &err = jasper_err_ref();
    if (!path || !out) { err = "bad arguments"; return JASPER_ERR; }
    *out = nullptr;
    threads = clamp_threads(threads);
    Mapped m;
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) { err = std::string("cannot open ") + path + ": " + strerror(errno); return JASPER_ERR; }
    struct stat st;
    if (fstat(m.fd, &st) != 0) { err = std::string("cannot stat ") + path; return JASPER_ERR; }
    if (!S_ISREG(st.st_mode) || st.st_size == 0) return 1;          // (a pipe, an empty file: the caller's own reader decides)
    m.n = (size_t)st.st_size;
    void *mp = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (mp == MAP_FAILED) { m.n = 0; err = std::string("cannot map ") + path + ": " + strerror(errno); return JASPER_ERR; }
    m.p = (const uint8_t *)mp;
    madvise(mp, m.n, MADV_WILLNEED);
    const uint8_t *d = m.p;
    const size_t n = m.n;
    if (d[0] != '>') return 1;

    // 1. header lines: a '>' at the start of a line.  Per unit of the file, in parallel.
    const size_t n_units = (n + UNIT - 1) / UNIT;
    std::vector<std::vector<size_t>> found(n_units);
    std::atomic<int> has_cr{0};
    parallel_for(n_units, threads, [&](size_t u) {
        const size_t lo = u * UNIT, hi = std::min(n, lo + UNIT);
        if (memchr(d + lo, '\r', hi - lo)) has_cr.store(1);
        const uint8_t *p = d + lo;
        while (p < d + hi) {
            p = (const uint8_t *)memchr(p, '>', (size_t)(d + hi - p));
            if (!p) break;
            const size_t at = (size_t)(p - d);
            if (at == 0 || d[at - 1] == '\n') found[u].push_back(at);
            ++p;
        }
    });
    if (has_cr.load()) return 1;
    std::vector<size_t> hs;
    for (auto &v : found) hs.insert(hs.end(), v.begin(), v.end());
    // 2. header tokens, body ranges
    struct Raw { size_t hs, he, body_end; };
    std::vector<Raw> raw(hs.size());
    for (size_t i = 0; i < hs.size(); ++i) {
        const uint8_t *e = (const uint8_t *)memchr(d + hs[i], '\n', n - hs[i]);
        raw[i].hs = hs[i];
        raw[i].he = e ? (size_t)(e - d) + 1 : n;
        raw[i].body_end = i + 1 < hs.size() ? hs[i + 1] : n;
    }
    std::vector<std::string> names(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) {
        size_t e = raw[i].he;
        if (e > raw[i].hs && d[e - 1] == '\n') --e;
        size_t tok = raw[i].hs;
        for (size_t q = raw[i].hs; q < e; ++q) {
            const uint8_t c = d[q];
            if (c != '\t' && (c < 0x20 || c > 0x7e)) return 1;       // (what str.split() / perl -a would make of it is the caller's business)
        }
        while (tok < e && d[tok] != ' ' && d[tok] != '\t') ++tok;
        names[i].assign((const char *)d + raw[i].hs, tok - raw[i].hs);
    }
    // 3. body units: newline counts -> where every unit's bases go in the arena
    struct BodyUnit { size_t lo, hi, contig, dst; size_t newlines; };
    std::vector<BodyUnit> units;
    for (size_t i = 0; i < raw.size(); ++i)
        for (size_t lo = raw[i].he; lo < raw[i].body_end; lo += UNIT) units.push_back(BodyUnit{lo, std::min(raw[i].body_end, lo + UNIT), i, 0, 0});
    std::atomic<size_t> odd_total{0};
    parallel_for(units.size(), threads, [&](size_t u) {
        size_t nl = 0, odd = 0;
        body_stats(d + units[u].lo, units[u].hi - units[u].lo, nl, odd);
        units[u].newlines = nl;
        if (odd) odd_total.fetch_add(odd);
    });
    if (odd_total.load()) return 1;
    std::vector<size_t> seq_len(raw.size(), 0);
    size_t total = 0;
    {
        std::vector<size_t> at(raw.size(), 0);
        for (auto &u : units) seq_len[u.contig] += (u.hi - u.lo) - u.newlines;
        size_t off = 0;
        for (size_t i = 0; i < raw.size(); ++i) { at[i] = off; off += seq_len[i]; }
        total = off;
        std::vector<size_t> cur = at;
        for (auto &u : units) { u.dst = cur[u.contig]; cur[u.contig] += (u.hi - u.lo) - u.newlines; }
    }
    jasper_asm *a = new jasper_asm();
    a->path = path;
    a->sequence_bytes = total;
    a->arena_len = total;
    a->arena_cap = ((total + 64 + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
    void *ar = mmap(nullptr, a->arena_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (ar == MAP_FAILED) { a->arena_cap = 0; delete a; err = "not enough host memory for the assembly"; return JASPER_ERR; }
    madvise(ar, a->arena_cap, MADV_HUGEPAGE);
    a->arena = (uint8_t *)ar;
    // 4. the bases, line ends taken out
    parallel_for(units.size(), threads, [&](size_t u) {
        const uint8_t *p = d + units[u].lo, *e = d + units[u].hi;
        uint8_t *o = a->arena + units[u].dst;
        while (p < e) {
            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            const size_t len = (size_t)((q ? q : e) - p);
            memcpy(o, p, len);
            o += len;
            p += len + 1;
        }
    });
    // 5. the contigs perl #1 emits records for: those with a sequence (src/jasper.sh:155 `if(not($seq eq ""))`)
    std::unordered_set<std::string> seen;
    size_t off = 0;
    for (size_t i = 0; i < raw.size(); ++i) {
        if (seq_len[i]) {
            if (!seen.insert(names[i]).second) { delete a; return 1; }      // (the join's hash keeps the LAST record of a name: not this path)
            AsmContig c;
            c.name.swap(names[i]);
            c.seq_off = off;
            c.seq_len = seq_len[i];
            a->contigs.push_back(std::move(c));
        }
        off += seq_len[i];
    }
    *out = a;
    return JASPER_OK;
}

void jasper_asm_close(jasper_asm *a) { delete a; }

int jasper_asm_info(const jasper_asm *a, uint64_t *sequence_bytes, uint64_t *n_contigs, uint64_t *n_bases) {
    if (!a) { jasper_err_ref() = "bad arguments"; return JASPER_ERR; }
    if (sequence_bytes) *sequence_bytes = a->sequence_bytes;
    if (n_contigs) *n_contigs = a->contigs.size();
    if (n_bases) *n_bases = a->arena_len;
    return JASPER_OK;
}

int jasper_asm_contig(const jasper_asm *a, uint64_t i, const char **name, uint64_t *name_len, uint64_t *n_bases) {
    if (!a || i >= a->contigs.size()) { jasper_err_ref() = "contig out of range"; return JASPER_ERR; }
    if (name) *name = a->contigs[i].name.data();
    if (name_len) *name_len = a->contigs[i].name.size();
    if (n_bases) *n_bases = a->contigs[i].seq_len;
    return JASPER_OK;
}

static std::string batch_file_name(const jasper_asm *a, size_t f) { return a->prefix + ".batch." + std::to_string(f) + ".fa"; }

static bool write_batch_file(const jasper_asm *a, size_t f, std::string &why) {
    const std::string fn = batch_file_name(a, f);
    const int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { why = "cannot create " + fn + ": " + strerror(errno); return false; }
    std::vector<std::string> hdr;
    std::vector<struct iovec> iov;
    const size_t c0 = a->file_first[f], c1 = a->file_first[f + 1];
    hdr.reserve(c1 - c0);
    static const char nl = '\n';
    for (size_t c = c0; c < c1; ++c) {
        const AsmChunk &ch = a->chunks[c];
        const AsmContig &ct = a->contigs[ch.contig];
        hdr.push_back(ct.name + ":" + std::to_string(ch.ci) + "\n");
    }
    for (size_t c = c0; c < c1; ++c) {
        const AsmChunk &ch = a->chunks[c];
        const AsmContig &ct = a->contigs[ch.contig];
        iov.push_back({(void *)hdr[c - c0].data(), hdr[c - c0].size()});
        // (pieces of at most 64 MiB: one writev call per piece keeps a cancel or an error close)
        for (uint64_t at = 0; at < ch.len; at += (64u << 20))
            iov.push_back({(void *)(a->arena + ct.seq_off + ch.ci + at), (size_t)std::min<uint64_t>(64u << 20, ch.len - at)});
        iov.push_back({(void *)&nl, 1});
    }
    const bool ok = writev_all(fd, iov.data(), (int)iov.size());
    if (!ok) why = "writing " + fn + ": " + strerror(errno);
    if (close(fd) != 0 && ok) { why = "closing " + fn + ": " + strerror(errno); return false; }
    return ok;
}

int jasper_asm_split(jasper_asm *a, uint64_t batch_size, const char *prefix, const uint32_t *only_files, uint32_t n_only, int write_files, int threads,
                     uint64_t *n_chunks, uint64_t *n_files) {
    std::string &err = jasper_err_ref();
    if (!a || !prefix || batch_size == 0) { err = "bad arguments"; return JASPER_ERR; }
    if (a->writer_running) { err = "the batch files of an earlier split are still being written"; return JASPER_ERR; }
    threads = clamp_threads(threads);
    a->batch_size = batch_size;
    a->prefix = prefix;
    a->chunks.clear();
    a->file_first.clear();
    a->file_bytes.clear();
    // perl #1 (src/jasper.sh:155): records of <= batch_size bases at offsets 0, bs, 2 bs ...;  perl #2 (:156): a new file starts at a
    // record once MORE than batch_size bases have gone into the current one
    uint64_t output = 0;
    uint32_t file = 0;
    a->file_first.push_back(0);
    a->file_bytes.push_back(0);
    for (size_t i = 0; i < a->contigs.size(); ++i) {
        AsmContig &ct = a->contigs[i];
        ct.first_chunk = a->chunks.size();
        for (uint64_t ci = 0; ci < ct.seq_len; ci += batch_size) {
            if (output > batch_size) {
                ++file;
                a->file_first.push_back(a->chunks.size());
                a->file_bytes.push_back(0);
                output = 0;
            }
            AsmChunk ch;
            ch.contig = (uint32_t)i;
            ch.file = file;
            ch.ci = ci;
            ch.len = std::min<uint64_t>(batch_size, ct.seq_len - ci);
            output += ch.len;
            a->file_bytes.back() += ct.name.size() + 1 + std::to_string(ci).size() + 1 + ch.len + 1;
            a->chunks.push_back(ch);
        }
        ct.n_chunks = a->chunks.size() - ct.first_chunk;
    }
    a->file_first.push_back(a->chunks.size());
    a->polished.assign(a->chunks.size(), jasper_asm::Polished());
    a->out_used = 0;
    a->have.assign(a->chunks.size(), 0);
    if (n_chunks) *n_chunks = a->chunks.size();
    if (n_files) *n_files = a->file_bytes.size();
    a->own_bases = a->arena_len;
    if (only_files) {
        a->own_bases = 0;
        for (uint32_t i = 0; i < n_only; ++i)
            if (only_files[i] < a->file_bytes.size())
                for (size_t c = a->file_first[only_files[i]]; c < a->file_first[only_files[i] + 1]; ++c) a->own_bases += a->chunks[c].len;
    }
    if (write_files) {
        std::vector<uint32_t> todo;
        if (only_files) {
            for (uint32_t i = 0; i < n_only; ++i) {
                if (only_files[i] >= a->file_bytes.size()) { err = "batch file out of range"; return JASPER_ERR; }
                todo.push_back(only_files[i]);
            }
        } else {
            for (uint32_t f = 0; f < a->file_bytes.size(); ++f) todo.push_back(f);
        }
        a->writer_failed.store(0);
        a->writer_err.clear();
        a->writer_running = true;
        a->writer = std::thread([a, todo, threads]() {
            std::vector<std::string> why(todo.size());
            parallel_for(todo.size(), threads, [&](size_t i) {
                if (a->writer_failed.load()) return;
                if (!write_batch_file(a, todo[i], why[i])) a->writer_failed.store(1);
            });
            for (auto &w : why)
                if (!w.empty()) { a->writer_err = w; break; }
        });
    }
    return JASPER_OK;
}

int jasper_asm_split_wait(jasper_asm *a) {
    if (!a) { jasper_err_ref() = "bad arguments"; return JASPER_ERR; }
    if (a->writer_running) {
        a->writer.join();
        a->writer_running = false;
    }
    if (a->writer_failed.load()) { jasper_err_ref() = a->writer_err.empty() ? std::string("writing the batch files failed") : a->writer_err; return JASPER_ERR; }
    return JASPER_OK;
}

int jasper_asm_chunks(const jasper_asm *a, uint32_t *contig, uint64_t *ci, uint64_t *len, uint32_t *file) {
    if (!a) { jasper_err_ref() = "bad arguments"; return JASPER_ERR; }
    for (size_t c = 0; c < a->chunks.size(); ++c) {
        if (contig) contig[c] = a->chunks[c].contig;
        if (ci) ci[c] = a->chunks[c].ci;
        if (len) len[c] = a->chunks[c].len;
        if (file) file[c] = a->chunks[c].file;
    }
    return JASPER_OK;
}

int jasper_asm_file_bytes(const jasper_asm *a, uint64_t *bytes) {
    if (!a || !bytes) { jasper_err_ref() = "bad arguments"; return JASPER_ERR; }
    for (size_t f = 0; f < a->file_bytes.size(); ++f) bytes[f] = a->file_bytes[f];
    return JASPER_OK;
}

int jasper_asm_chunk_text(const jasper_asm *a, uint64_t chunk, int polished, const char **text, uint64_t *len) {
    if (!a || chunk >= a->chunks.size() || !text || !len) { jasper_err_ref() = "chunk out of range"; return JASPER_ERR; }
    if (polished) {
        if (!a->have[chunk]) { jasper_err_ref() = "no polished text is held for this chunk record"; return JASPER_ERR; }
        *text = a->polished[chunk].data();
        *len = a->polished[chunk].size();
    } else {
        const AsmChunk &ch = a->chunks[chunk];
        *text = (const char *)a->arena + a->contigs[ch.contig].seq_off + ch.ci;
        *len = ch.len;
    }
    return JASPER_OK;
}

int jasper_asm_put(jasper_asm *a, uint64_t chunk, const char *text, uint64_t len) {
    if (!a || chunk >= a->chunks.size() || (len && !text)) { jasper_err_ref() = "chunk out of range"; return JASPER_ERR; }
    a->polished[chunk].p = nullptr;
    a->polished[chunk].own.assign(text ? text : "", (size_t)len);
    a->have[chunk] = 1;
    return JASPER_OK;
}

// src/jasper.py:120-128,142-147: ">name\n" + lines of 60 (the reference writes them one by one; an empty record has no line)
int jasper_asm_write_fixed(jasper_asm *a, const uint32_t *files, const char *const *out_paths, uint32_t n_files, int threads) {
    std::string &err = jasper_err_ref();
    if (!a || (n_files && (!files || !out_paths))) { err = "bad arguments"; return JASPER_ERR; }
    threads = clamp_threads(threads);
    for (uint32_t i = 0; i < n_files; ++i) {
        if (files[i] >= a->file_bytes.size()) { err = "batch file out of range"; return JASPER_ERR; }
        for (size_t c = a->file_first[files[i]]; c < a->file_first[files[i] + 1]; ++c)
            if (!a->have[c]) { err = "no polished text is held for a chunk record of this batch file"; return JASPER_ERR; }
    }
    std::vector<std::string> why(n_files);
    std::atomic<int> failed{0};
    parallel_for(n_files, threads, [&](size_t i) {
        const int fd = open(out_paths[i], O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) { why[i] = std::string("cannot create ") + out_paths[i] + ": " + strerror(errno); failed.store(1); return; }
        std::string buf;
        bool ok = true;
        for (size_t c = a->file_first[files[i]]; c < a->file_first[files[i] + 1] && ok; ++c) {
            const AsmChunk &ch = a->chunks[c];
            const jasper_asm::Polished &s = a->polished[c];
            buf.clear();
            buf.reserve(s.size() + s.size() / 60 + 64 + a->contigs[ch.contig].name.size());
            buf += a->contigs[ch.contig].name;
            buf += ':';
            buf += std::to_string(ch.ci);
            buf += '\n';
            for (size_t at = 0; at < s.size(); at += 60) {
                buf.append(s.data() + at, std::min<size_t>(60, s.size() - at));
                buf += '\n';
            }
            struct iovec v = {(void *)buf.data(), buf.size()};
            ok = writev_all(fd, &v, 1);
        }
        if (!ok) why[i] = std::string("writing ") + out_paths[i] + ": " + strerror(errno);
        if (close(fd) != 0 && ok) { why[i] = std::string("closing ") + out_paths[i]; ok = false; }
        if (!ok) failed.store(1);
    });
    if (failed.load()) {
        for (auto &w : why)
            if (!w.empty()) { err = w; break; }
        return JASPER_ERR;
    }
    return JASPER_OK;
}

int jasper_asm_polished_lens(const jasper_asm *a, uint64_t *lens, uint8_t *have) {
    if (!a) { jasper_err_ref() = "bad arguments"; return JASPER_ERR; }
    for (size_t c = 0; c < a->chunks.size(); ++c) {
        if (lens) lens[c] = a->have[c] ? a->polished[c].size() : 0;
        if (have) have[c] = a->have[c];
    }
    return JASPER_OK;
}

// src/jasper.sh:220: per contig ">name\n", its records in offset order, "\n".  The place of every record in the file follows
// from the records' lengths alone, so several processes (one per GPU) can each write the records they polished into ONE file:
// mode 1 creates the file at its final size, mode 2 writes what this job holds -- and the ">name\n" of a contig whose first
// record it holds, the closing "\n" of one whose last record it holds --, mode 3 does both (one process).
int jasper_asm_join(jasper_asm *a, const char *out_path, const uint64_t *all_lens, int mode, int threads) {
    std::string &err = jasper_err_ref();
    if (!a || !out_path || !(mode & 3)) { err = "bad arguments"; return JASPER_ERR; }
    threads = clamp_threads(threads);
    const size_t nc = a->chunks.size();
    std::vector<uint64_t> own;
    if (!all_lens) {
        own.resize(nc);
        for (size_t c = 0; c < nc; ++c) {
            if (!a->have[c]) { err = "the join needs the polished text of every chunk record"; return JASPER_ERR; }
            own[c] = a->polished[c].size();
        }
        all_lens = own.data();
    }
    std::vector<uint64_t> at(nc), hdr_at(a->contigs.size()), end_at(a->contigs.size());
    uint64_t off = 0;
    for (size_t i = 0; i < a->contigs.size(); ++i) {
        const AsmContig &ct = a->contigs[i];
        hdr_at[i] = off;
        off += ct.name.size() + 1;
        for (size_t c = ct.first_chunk; c < ct.first_chunk + ct.n_chunks; ++c) { at[c] = off; off += all_lens[c]; }
        end_at[i] = off;
        off += 1;
    }
    const int fd = open(out_path, (mode & 1) ? (O_WRONLY | O_CREAT | O_TRUNC) : O_WRONLY, 0666);
    if (fd < 0) { err = std::string("cannot open ") + out_path + ": " + strerror(errno); return JASPER_ERR; }
    if ((mode & 1) && ftruncate(fd, (off_t)off) != 0) { err = std::string("cannot size ") + out_path + ": " + strerror(errno); close(fd); return JASPER_ERR; }
    bool ok = true;
    if (mode & 2) {
        struct Piece { const char *p; size_t n; uint64_t at; };
        std::vector<Piece> pieces;
        static const char nl = '\n';
        std::vector<std::string> hdrs;
        hdrs.reserve(a->contigs.size());
        for (size_t i = 0; i < a->contigs.size(); ++i) {
            const AsmContig &ct = a->contigs[i];
            if (!ct.n_chunks) continue;
            if (a->have[ct.first_chunk]) {
                hdrs.push_back(ct.name + "\n");
                pieces.push_back({hdrs.back().data(), hdrs.back().size(), hdr_at[i]});
            }
            if (a->have[ct.first_chunk + ct.n_chunks - 1]) pieces.push_back({&nl, 1, end_at[i]});
            for (size_t c = ct.first_chunk; c < ct.first_chunk + ct.n_chunks; ++c) {
                if (!a->have[c]) continue;
                const jasper_asm::Polished &s = a->polished[c];
                if (s.size() != all_lens[c]) { err = "a chunk record's polished length is not the one the file was laid out for"; close(fd); return JASPER_ERR; }
                for (size_t q = 0; q < s.size(); q += UNIT) pieces.push_back({s.data() + q, std::min(UNIT, s.size() - q), at[c] + q});
            }
        }
        std::atomic<int> failed{0};
        parallel_for(pieces.size(), threads, [&](size_t i) {
            if (failed.load()) return;
            if (!write_all(fd, pieces[i].p, pieces[i].n, (off_t)pieces[i].at)) failed.store(errno ? errno : EIO);
        });
        if (failed.load()) { err = std::string("writing ") + out_path + ": " + strerror(failed.load()); ok = false; }
    }
    if (close(fd) != 0 && ok) { err = std::string("closing ") + out_path + ": " + strerror(errno); ok = false; }
    return ok ? JASPER_OK : JASPER_ERR;
}

// src/jasper.sh:222-226:  awk 'NR==1 || FNR>1' FILES | awk -F ':' '{print $1" "$2}' | sort -k1,1 -k2,2n -k3,3n | awk '{print $1":"$2" "$3" "$4" "$5}'
// on the per-batch fix CSVs (space-delimited, CRLF: the '\r' is no separator for awk and stays on the last field).  Byte order
// for the name key and for sort's last-resort comparison of whole lines.  Returns 1 (nothing written) when a line holds a byte
// that is not printable ASCII, blank, tab or '\r', or a number of more than 18 digits: the caller's own restatement of the rules
// (jasper_amd/cli.py: merge_fix_csvs) decides then.
int jasper_merge_fix_csvs(const char *const *paths, uint32_t n_paths, const char *out_path) {
    std::string &err = jasper_err_ref();
    if ((n_paths && !paths) || !out_path) { err = "bad arguments"; return JASPER_ERR; }
    struct Row { std::string line; uint32_t f[5][2]; uint32_t nf; int64_t k1, k2; };
    std::vector<Row> rows;
    std::string content;
    for (uint32_t i = 0; i < n_paths; ++i) {
        FILE *fp = fopen(paths[i], "rb");
        if (!fp) { err = std::string("cannot open ") + paths[i] + ": " + strerror(errno); return JASPER_ERR; }
        content.clear();
        char buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, fp)) > 0) content.append(buf, k);
        const bool bad = ferror(fp) != 0;
        fclose(fp);
        if (bad) { err = std::string("read error in ") + paths[i]; return JASPER_ERR; }
        size_t pos = 0, fnr = 0;
        while (pos < content.size()) {                           // (a last line without '\n' counts; nothing after a final '\n' does)
            size_t e = content.find('\n', pos);
            if (e == std::string::npos) e = content.size();
            ++fnr;
            if ((i == 0 && fnr == 1) || fnr > 1) {
                // awk -F ':' '{print $1" "$2}'
                const char *l = content.data() + pos;
                const size_t ln = e - pos;
                for (size_t q = 0; q < ln; ++q) {
                    const unsigned char c = (unsigned char)l[q];
                    if (!((c >= 0x20 && c <= 0x7e) || c == '\t' || c == '\r')) return 1;
                }
                const char *c1 = (const char *)memchr(l, ':', ln);
                Row r;
                if (!c1) { r.line.assign(l, ln); r.line += ' '; }
                else {
                    const char *c2 = (const char *)memchr(c1 + 1, ':', (size_t)(l + ln - c1 - 1));
                    r.line.assign(l, (size_t)(c1 - l));
                    r.line += ' ';
                    r.line.append(c1 + 1, (size_t)((c2 ? c2 : l + ln) - c1 - 1));
                }
                // awk's default fields: runs of blanks / tabs separate
                r.nf = 0;
                const std::string &s = r.line;
                size_t q = 0;
                uint32_t n_all = 0;
                while (q < s.size()) {
                    while (q < s.size() && (s[q] == ' ' || s[q] == '\t')) ++q;
                    if (q >= s.size()) break;
                    size_t b = q;
                    while (q < s.size() && s[q] != ' ' && s[q] != '\t') ++q;
                    if (n_all < 5) { r.f[n_all][0] = (uint32_t)b; r.f[n_all][1] = (uint32_t)(q - b); r.nf = n_all + 1; }
                    ++n_all;
                }
                auto num = [&](uint32_t fi, int64_t &out) -> bool {      // sort -n: a leading integer, 0 if none
                    out = 0;
                    if (fi >= r.nf) return true;
                    const char *p = s.data() + r.f[fi][0];
                    uint32_t m = r.f[fi][1], at = 0;
                    bool neg = false;
                    if (at < m && p[at] == '-') { neg = true; ++at; }
                    uint32_t digits = 0;
                    int64_t v = 0;
                    while (at < m && p[at] >= '0' && p[at] <= '9') { if (++digits > 18) return false; v = v * 10 + (p[at] - '0'); ++at; }
                    out = digits ? (neg ? -v : v) : 0;
                    return true;
                };
                if (!num(1, r.k1) || !num(2, r.k2)) return 1;
                rows.push_back(std::move(r));
            }
            pos = e + 1;
        }
    }
    auto cmp_bytes = [](const char *a, size_t na, const char *b, size_t nb) {
        const int c = memcmp(a, b, std::min(na, nb));
        return c ? c : (na < nb ? -1 : na > nb ? 1 : 0);
    };
    std::vector<uint32_t> order(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const Row &a = rows[x], &b = rows[y];
        const int c = cmp_bytes(a.nf ? a.line.data() + a.f[0][0] : "", a.nf ? a.f[0][1] : 0, b.nf ? b.line.data() + b.f[0][0] : "", b.nf ? b.f[0][1] : 0);
        if (c) return c < 0;
        if (a.k1 != b.k1) return a.k1 < b.k1;
        if (a.k2 != b.k2) return a.k2 < b.k2;
        const int w = cmp_bytes(a.line.data(), a.line.size(), b.line.data(), b.line.size());
        return w ? w < 0 : x < y;
    });
    std::string out;
    out.reserve(rows.size() * 40);
    for (uint32_t idx : order) {                                  // awk '{print $1":"$2" "$3" "$4" "$5}'
        const Row &r = rows[idx];
        for (uint32_t fi = 0; fi < 5; ++fi) {
            if (fi < r.nf) out.append(r.line.data() + r.f[fi][0], r.f[fi][1]);
            if (fi < 4) out += fi == 0 ? ':' : ' ';
        }
        out += '\n';
    }
    const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { err = std::string("cannot create ") + out_path + ": " + strerror(errno); return JASPER_ERR; }
    const bool ok = write_all(fd, out.data(), out.size(), 0);
    if (close(fd) != 0 || !ok) { err = std::string("writing ") + out_path + ": " + strerror(errno); return JASPER_ERR; }
    return JASPER_OK;
}

}  // extern "C"
