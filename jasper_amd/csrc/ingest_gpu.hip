// ingest_gpu.hip -- FASTA / FASTQ text -> base stream ON THE GPU (the step before the hot path).
//
// The host only moves bytes: file (plain or gzip) -> pinned buffer -> HBM.  The text of a chunk is then parsed by kernels
// with the record rules of `jellyfish count` (JF::include/jellyfish/mer_overlap_sequence_parser.hpp:120-307, restated in
// ingest.hpp), and the bases are appended to a device buffer that goes to Table::count_device in large pieces (so file
// input takes the partitioned counting path too).
//
// What makes this exact rather than heuristic: lines are numbered from the START of the stream (newline prefix sums), so
// in a FASTQ whose records are the usual four lines "line number mod 4" decides what a line is -- no guessing whether an
// '@' starts a header or a quality string.  Everything else is VERIFIED, not assumed: every line 0 mod 4 starts with '@',
// every line 2 mod 4 with '+', no sequence line starts with '+', sequence and quality lines have equal lengths, there
// is no '\r' anywhere.  A chunk that fails any check is handed, from its first byte (a record boundary), to the host
// state machine (FastxParser, resumed in the right mode), which implements the reference's rules for multi-line records,
// DOS line ends and malformed input, and which then keeps the rest of the stream.  FASTA needs no such assumption: a line
// is a header iff it starts with '>'.
#include "ingest.hpp"
#include "pgunzip.hpp"
#include "table.hpp"
#include <cstdio>
#include <ctime>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <zlib.h>
#include <rocprim/device/device_scan.hpp>

namespace jk {

namespace {

constexpr int IG_THREADS = 256;
constexpr int IG_BYTES = IG_THREADS * 16;       // text bytes per block
enum { IGF_CR = 1, IGF_BAD = 2 };

struct Raw { uint32_t w[4]; };
__device__ __forceinline__ Raw load_text16(const uint8_t *__restrict__ text, uint64_t pos, uint64_t n) {
    Raw r;
    if (pos + 16 <= n) {
        // (one 16-byte load; pos % 16 == 0, but a chunk's text begins wherever the record carried over from the chunk before puts it:
        //  the type says so, the hardware reads global memory at any address)
        struct __attribute__((packed, aligned(1))) V16 { uint32_t w[4]; };
        const V16 v = *reinterpret_cast<const V16 *>(text + pos);
        r.w[0] = v.w[0]; r.w[1] = v.w[1]; r.w[2] = v.w[2]; r.w[3] = v.w[3];
    } else {
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
        for (int j = 0; j < 16; ++j)
            if (pos + j < n) r.w[j >> 2] |= (uint32_t)text[pos + j] << (8 * (j & 3));
    }
    return r;
}
__device__ __forceinline__ uint32_t byte_of(const Raw &r, int j) { return (r.w[j >> 2] >> (8 * (j & 3))) & 0xFFu; }

// exclusive prefix over the block of a per-thread count; returns the block total through s_tot
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t *s_w, uint32_t &total) {
    const int t = threadIdx.x;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
    if ((t & 63) == 63) s_w[t >> 6] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    total = 0;
    for (int w = 0; w < IG_THREADS / 64; ++w) { if (w < (t >> 6)) wbase += s_w[w]; total += s_w[w]; }
    __syncthreads();
    return wbase + inc - v;
}

// pass 1: newlines per block, and "is there a '\r' at all"
__global__ __launch_bounds__(IG_THREADS) void ig_count_nl_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ blk_nl,
                                                                 unsigned int *__restrict__ flags) {
    __shared__ uint32_t s_w[IG_THREADS / 64];
    const uint64_t pos = (uint64_t)blockIdx.x * IG_BYTES + (uint64_t)threadIdx.x * 16;
    uint32_t c = 0;
    bool cr = false;
    if (pos < n) {
        const Raw r = load_text16(text, pos, n);
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = byte_of(r, j);
            if (pos + j < n) { c += b == '\n'; cr = cr || b == '\r'; }
        }
    }
    uint32_t total;
    (void)block_exclusive(c, s_w, total);
    if (threadIdx.x == 0) blk_nl[blockIdx.x] = total;
    if (__ballot(cr) && (threadIdx.x & 63) == 0) atomicOr(flags, (unsigned int)IGF_CR);
}

// pass 2: position of every newline: nlpos[line] = byte offset of the '\n' that ends line `line` (chunk-relative numbering)
__global__ __launch_bounds__(IG_THREADS) void ig_nlpos_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ blk_base,
                                                              uint32_t *__restrict__ nlpos) {
    __shared__ uint32_t s_w[IG_THREADS / 64];
    const uint64_t pos = (uint64_t)blockIdx.x * IG_BYTES + (uint64_t)threadIdx.x * 16;
    uint32_t c = 0;
    Raw r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    if (pos < n) {
        r = load_text16(text, pos, n);
        for (int j = 0; j < 16; ++j) c += (pos + j < n && byte_of(r, j) == '\n');
    }
    uint32_t total;
    uint32_t line = blk_base[blockIdx.x] + block_exclusive(c, s_w, total);
    if (pos < n)
        for (int j = 0; j < 16; ++j)
            if (pos + j < n && byte_of(r, j) == '\n') nlpos[line++] = (uint32_t)(pos + j);
}

// FASTQ: the four-line shape of every record, verified
__global__ __launch_bounds__(256) void ig_check_fastq_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ nlpos, uint64_t nrec,
                                                             unsigned int *__restrict__ flags) {
    bool bad = false;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < nrec; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e0 = nlpos[4 * r], e1 = nlpos[4 * r + 1], e2 = nlpos[4 * r + 2], e3 = nlpos[4 * r + 3];
        const uint32_t s0 = r ? nlpos[4 * r - 1] + 1 : 0u;
        const uint32_t len1 = e1 - e0 - 1, len3 = e3 - e2 - 1;
        bad = bad || e0 == s0 || text[s0] != '@';                        // header line: not empty, starts with '@'
        bad = bad || e2 == e1 + 1 || text[e1 + 1] != '+';                // separator line starts with '+'
        bad = bad || (len1 && text[e0 + 1] == '+');                      // a sequence line starting with '+' would be taken as the separator
        bad = bad || len1 != len3;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, (unsigned int)IGF_BAD);
}

// what a text byte contributes to the base stream: 0 nothing, else the byte to emit
__device__ __forceinline__ uint32_t emit_of(const uint8_t *__restrict__ text, const uint32_t *__restrict__ nlpos, int fastq, uint32_t line, uint32_t b,
                                            uint64_t p) {
    if (fastq) {
        if ((line & 3u) != 1u) return 0u;
        return b == '\n' ? (uint32_t)'N' : b;                            // the sequence line; its newline becomes the record separator
    }
    const uint32_t start = line ? nlpos[line - 1] + 1 : 0u;
    if (text[start] == '>') return p == start ? (uint32_t)'N' : 0u;      // header line: one separator
    return b == '\n' ? 0u : b;                                           // sequence lines are concatenated
}

// pass 3 (write == 0): emitted bytes per block; pass 4 (write == 1): emit them at out + blk_out[block]
template <int WRITE>
__global__ __launch_bounds__(IG_THREADS) void ig_emit_kernel(const uint8_t *__restrict__ text, uint64_t n_use, const uint32_t *__restrict__ blk_base,
                                                             const uint32_t *__restrict__ nlpos, int fastq, uint32_t *__restrict__ blk_emit,
                                                             const uint32_t *__restrict__ blk_out, uint8_t *__restrict__ out) {
    __shared__ uint32_t s_w[IG_THREADS / 64];
    const uint64_t pos = (uint64_t)blockIdx.x * IG_BYTES + (uint64_t)threadIdx.x * 16;
    uint32_t c = 0;
    Raw r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    if (pos < n_use) {
        r = load_text16(text, pos, n_use);
        for (int j = 0; j < 16; ++j) c += (pos + j < n_use && byte_of(r, j) == '\n');
    }
    uint32_t total;
    uint32_t line = blk_base[blockIdx.x] + block_exclusive(c, s_w, total);
    uint8_t em[16];
    uint32_t ne = 0;
    if (pos < n_use)
        for (int j = 0; j < 16; ++j) {
            if (pos + j >= n_use) break;
            const uint32_t b = byte_of(r, j);
            const uint32_t e = emit_of(text, nlpos, fastq, line, b, pos + j);
            if (e) em[ne++] = (uint8_t)e;
            if (b == '\n') ++line;
        }
    uint32_t tot2;
    const uint32_t off = block_exclusive(ne, s_w, tot2);
    if (!WRITE) {
        if (threadIdx.x == 0) blk_emit[blockIdx.x] = tot2;
    } else {
        uint8_t *dst = out + blk_out[blockIdx.x] + off;
        for (uint32_t q = 0; q < ne; ++q) dst[q] = em[q];
    }
}

// One gzip file inflated AHEAD of the consumer by its own thread, at most `cap` bytes ahead.  `zcat -f` inflates on one core
// (~0.3 GB/s of text) while the GPU parses text at 10 GB/s, so gzip input is inflate-bound; the files of one call are
// independent gzip streams, so all of them are inflated at the same time and the stream the parser sees -- their
// concatenation, in order -- is unchanged.  Also overlaps inflating with the GPU work on the previous chunk.
struct GzAhead {
    static constexpr size_t BLK = 4u << 20;
    struct Block {
        std::unique_ptr<char[]> p; std::vector<uint8_t> v; size_t n = 0, used = 0;      // storage: p (zlib path) or v (parallel path)
        const char *data() const { return p ? p.get() : reinterpret_cast<const char *>(v.data()); }
    };
    std::string path, err;
    size_t cap;
    // threads for THIS file's inflation (1 = the plain zlib reader): see Reader::start_ahead
    int gz_threads = 1;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Block> q;
    size_t queued = 0;
    bool done = false, failed = false, stop = false;
    std::thread th;
    GzAhead(const char *path_, size_t cap_, int gz_threads_ = 1) : path(path_), cap(cap_ < 2 * BLK ? 2 * BLK : cap_), gz_threads(gz_threads_) { th = std::thread([this] { run(); }); }
    ~GzAhead() {
        { std::lock_guard<std::mutex> l(mu); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    // a large regular gzip file: many threads (pgunzip.hpp); true when it was read to its end or failed, false when the
    // file is not for that reader (small, not a regular file) and zlib's reader takes over from the first byte
    bool run_parallel() {
        if (gz_threads < 2) return false;
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
        ParallelGunzip pg(path.c_str(), gz_threads, ParallelGunzip::chunk_for((size_t)st.st_size, gz_threads));
        if (!pg.open()) return false;
        std::vector<std::vector<uint8_t>> pieces;
        while (pg.next(pieces)) {
            for (auto &pc : pieces) {
                if (pc.empty()) continue;
                Block b;
                b.n = pc.size();
                b.v = std::move(pc);
                std::unique_lock<std::mutex> l(mu);
                queued += b.n;
                q.push_back(std::move(b));
                cv.notify_all();
                cv.wait(l, [this] { return stop || queued + BLK <= cap || q.size() <= 1; });
                if (stop) { done = true; cv.notify_all(); return true; }
            }
        }
        std::lock_guard<std::mutex> l(mu);
        if (!pg.error().empty()) { failed = true; err = pg.error(); }
        done = true;
        cv.notify_all();
        return true;
    }
    void run() {
        if (run_parallel()) return;
        gzFile g = gzopen(path.c_str(), "rb");
        if (!g) { std::lock_guard<std::mutex> l(mu); failed = done = true; err = "cannot open " + path; cv.notify_all(); return; }
        gzbuffer(g, 1u << 20);
        for (;;) {
            Block b;
            b.p.reset(new char[BLK]);
            const int r = gzread(g, b.p.get(), (unsigned)BLK);
            std::unique_lock<std::mutex> l(mu);
            if (r < 0) { failed = done = true; err = "read error in " + path; break; }
            if (r == 0) { done = true; break; }
            b.n = (size_t)r;
            queued += b.n;
            q.push_back(std::move(b));
            cv.notify_all();
            cv.wait(l, [this] { return stop || queued + BLK <= cap; });
            if (stop) { done = true; break; }
        }
        cv.notify_all();
        gzclose(g);
    }
    // up to `want` bytes; 0 at the end of the file, -1 on error
    long read(char *dst, size_t want) {
        size_t got = 0;
        std::unique_lock<std::mutex> l(mu);
        while (got < want) {
            cv.wait(l, [this] { return !q.empty() || done; });
            if (q.empty()) { if (failed) return -1; break; }
            Block &b = q.front();
            const size_t m = std::min(want - got, b.n - b.used);
            l.unlock();
            memcpy(dst + got, b.data() + b.used, m);       // (only the consumer touches the front block)
            l.lock();
            b.used += m;
            got += m;
            queued -= m;
            if (b.used == b.n) q.pop_front();
            cv.notify_all();
        }
        return (long)got;
    }
};

struct Reader {           // the concatenation of all input files as one byte stream (src/jasper.sh:177 `zcat -f $READS`)
    const char *const *paths;
    int n_paths, cur = 0;
    // optional byte range [begins[i], ends[i]) of plain file i (ends[i] < 0: to the end); a gzip member or a pipe cannot be
    // cut and is read whole when its range starts at 0 and skipped otherwise (multi-GPU read shards)
    const int64_t *begins = nullptr, *ends = nullptr;
    std::vector<std::unique_ptr<GzAhead>> ahead;   // per path: the inflating thread of a gzip file (started for all of them at once)
    bool ahead_started = false;
    GzAhead *ga = nullptr;                         // the current file's, when it is a gzip file
    gzFile g = nullptr;   // pipes and other non-regular files are read through zlib's pass-through ...
    int fd = -1;          // ... anything else is read as it is (what `zcat -f` does): pread by a few threads, because one
    off_t off = 0, size = 0;   // thread copies out of the page cache at only ~7 GB/s
    // A plain file is MAPPED and the threads copy out of the mapping: the same bytes leave the page cache at 107-114 GB/s by memcpy
    // against 65-82 GB/s by pread (8 GB of distinct bytes into pinned memory, 8-32 threads: tools/probes/read_probe.hip) -- and this
    // copy is what bounds files -> table.  Only for a file that IS in the page cache (sampled with mincore: 32 windows of
    // 64 pages over the part to be read, nine in ten resident): what has to come from a disk or over a network is better asked for by
    // pread in pieces of megabytes than page fault by page fault.  JASPER_INGEST_MMAP=0: always pread, =1: always the mapping (a file
    // that is TRUNCATED while it is being read ends the process with SIGBUS through the mapping; pread would report a short read).
    const char *map = nullptr;
    size_t map_len = 0;
    const int mmap_mode = []() { const char *e = getenv("JASPER_INGEST_MMAP"); return !e ? 2 : atoi(e) != 0 ? 1 : 0; }();     // 2 = when cached
    static bool mostly_cached(const char *m, size_t lo, size_t hi) {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE), win = 64 * page;
        if (hi <= lo) return true;
        size_t seen = 0, in = 0;
        unsigned char vec[64];
        const size_t span = hi - lo, step = std::max(win, (span / 32 + page - 1) / page * page);      // (whole pages: mincore wants its address on one)
        for (size_t at = lo / page * page; at < hi; at += step) {
            const size_t n = std::min(win, (hi - at + page - 1) / page * page);
            if (mincore(const_cast<char *>(m) + at, n, vec) != 0) return false;
            for (size_t i = 0; i < (n + page - 1) / page; ++i) { ++seen; in += vec[i] & 1u; }
        }
        return seen == 0 || in * 10 >= seen * 9;
    }
    // (taking a mapping of GBs down costs tens of ms -- one page-table entry per 4 KB read: done by a thread of its own, off the reader's path)
    void unmap() {
        if (!map) return;
        char *m_ = const_cast<char *>(map);
        const size_t l_ = map_len;
        map = nullptr; map_len = 0;
        if (l_ >= (64u << 20)) std::thread([m_, l_]() { munmap(m_, l_); }).detach();
        else munmap(m_, l_);
    }
    std::string err;
    static constexpr int MAX_READ_THREADS = 32;
    // (measured on the GPU box, 2.9 GB of FASTQ in the page cache: 4 threads 0.25 s, 8 threads 0.16 s for files -> table)
    const int READ_THREADS = []() {
        if (const char *e = getenv("JASPER_INGEST_READ_THREADS")) return std::max(1, std::min(atoi(e), MAX_READ_THREADS));
        return (int)std::max(8u, std::min(16u, std::thread::hardware_concurrency() / 8u));      // (8 is the floor: hardware_concurrency() may be 0)
    }();
    // fills buf with up to want bytes; returns bytes read, 0 at the end of the last file, -1 on error
    long read(char *buf, size_t want) {
        size_t got = 0;
        if (!ahead_started) start_ahead();
        while (got < want) {
            if (!g && fd < 0 && !ga) {
                if (cur >= n_paths) break;
                fd = open(paths[cur], O_RDONLY);
                if (fd < 0) { err = std::string("cannot open ") + paths[cur]; return -1; }
                unsigned char magic[2] = {0, 0};
                const ssize_t m = pread(fd, magic, 2, 0);
                struct stat st;
                const int64_t rb = begins ? begins[cur] : 0, re = ends ? ends[cur] : -1;
                const bool whole = m == 2 && magic[0] == 0x1f && magic[1] == 0x8b ? true : (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode));
                if ((whole && rb != 0) || (re >= 0 && re <= rb)) {       // not this reader's part of the input
                    close(fd);
                    fd = -1;
                    ++cur;
                    continue;
                }
                if (m == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
                    close(fd);
                    fd = -1;
                    if (!ahead[cur]) ahead[cur].reset(new GzAhead(paths[cur], ahead_cap, gz_threads));
                    ga = ahead[cur].get();
                } else if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {       // pipes etc.: sequential reads through zlib's pass-through
                    // (on the descriptor already open: closing a FIFO's only reader would break the writer's pipe)
                    g = gzdopen(fd, "rb");
                    if (!g) { close(fd); fd = -1; err = std::string("cannot open ") + paths[cur]; return -1; }
                    fd = -1;
                    gzbuffer(g, 1u << 20);
                } else {
                    off = (off_t)std::min<int64_t>(rb, (int64_t)st.st_size);
                    size = re >= 0 ? (off_t)std::min<int64_t>(re, (int64_t)st.st_size) : st.st_size;
                    if (mmap_mode && st.st_size > 0 && size > off) {
                        void *m_ = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
                        if (m_ != MAP_FAILED) {
                            const bool cached = mmap_mode != 2 || mostly_cached(static_cast<const char *>(m_), (size_t)off, (size_t)size);
                            if (getenv("JASPER_COUNT_DEBUG")) fprintf(stderr, "[ingest] %s: %s\n", paths[cur], cached ? "copied out of a mapping" : "not in the page cache: pread");
                            if (!cached) munmap(m_, (size_t)st.st_size);
                            else {
                                map = static_cast<const char *>(m_);
                                map_len = (size_t)st.st_size;
                                (void)madvise(m_, map_len, MADV_SEQUENTIAL);
                            }
                        }
                    }
                }
            }
            long r = 0;
            if (ga) {
                r = ga->read(buf + got, want - got);
                if (r < 0) { err = ga->err; return -1; }
            } else if (g) {
                r = gzread(g, buf + got, (unsigned)std::min<size_t>(want - got, 1u << 30));
                if (r < 0) { err = std::string("read error in ") + paths[cur]; return -1; }
            } else {
                const size_t todo = (size_t)std::min<off_t>((off_t)(want - got), size - off);
                if (todo) {
                    const size_t part = (todo + READ_THREADS - 1) / READ_THREADS;
                    bool ok[MAX_READ_THREADS];
                    std::thread th[MAX_READ_THREADS];
                    auto work = [&](int i) {
                        size_t lo = (size_t)i * part, hi = std::min(todo, lo + part);
                        ok[i] = true;
                        if (map) { if (lo < hi) memcpy(buf + got + lo, map + off + (off_t)lo, hi - lo); return; }
                        while (lo < hi) {
                            const ssize_t k = pread(fd, buf + got + lo, hi - lo, off + (off_t)lo);
                            if (k <= 0) { ok[i] = false; return; }
                            lo += (size_t)k;
                        }
                    };
                    for (int i = 1; i < READ_THREADS; ++i) th[i] = std::thread(work, i);
                    work(0);
                    for (int i = 1; i < READ_THREADS; ++i) th[i].join();
                    for (int i = 0; i < READ_THREADS; ++i)
                        if (!ok[i]) { err = std::string("read error in ") + paths[cur]; return -1; }
                    off += (off_t)todo;
                    r = (long)todo;
                }
            }
            if (r == 0) {
                if (ga) { ga = nullptr; ahead[cur].reset(); }
                else if (g) { gzclose(g); g = nullptr; }
                else { unmap(); close(fd); fd = -1; }
                ++cur;
                continue;
            }
            got += (size_t)r;
        }
        return (long)got;
    }
    // every gzip file this reader will read gets its inflating thread now; together they may run ahead by a budget of
    // JASPER_INGEST_AHEAD_MB (default: a quarter of the machine's memory, at most 16 GiB), shared evenly
    size_t ahead_cap = 0;
    int gz_threads = 1;
    void start_ahead() {
        ahead_started = true;
        ahead.resize((size_t)n_paths);
        std::vector<int> gz;
        for (int i = 0; i < n_paths; ++i) {
            if (begins && begins[i] != 0) continue;
            if (ends && ends[i] >= 0 && ends[i] <= (begins ? begins[i] : 0)) continue;
            unsigned char magic[2] = {0, 0};
            struct stat st;
            if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) continue;     // (never open a pipe just to look at it)
            const int f = open(paths[i], O_RDONLY);
            if (f < 0) continue;
            const ssize_t m = pread(f, magic, 2, 0);
            close(f);
            if (m == 2 && magic[0] == 0x1f && magic[1] == 0x8b) gz.push_back(i);
        }
        if (gz.empty()) return;
        size_t budget = 16ull << 30;
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
        if (pages > 0 && psz > 0) budget = std::min<size_t>(budget, (size_t)pages * (size_t)psz / 4);
        // one process per GPU on the same host: the ranks share the machine's memory
        if (const char *w = getenv("WORLD_SIZE")) { const long nw = atol(w); if (nw > 1) budget /= (size_t)nw; }
        if (const char *e = getenv("JASPER_INGEST_AHEAD_MB")) budget = (size_t)strtoull(e, nullptr, 10) << 20;
        ahead_cap = budget / gz.size();
        const size_t max_threads = 16;                       // (more files than that: the later ones start when they are reached)
        // threads that inflate ONE file (pgunzip.hpp): JASPER_INGEST_GZ_THREADS, else the host's threads divided by the ranks of
        // the job and by the gzip files inflated at the same time (at most 48; below 2 the plain zlib reader is used)
        {
            long hw = (long)std::thread::hardware_concurrency();
            if (const char *w = getenv("WORLD_SIZE")) { const long nw = atol(w); if (nw > 1) hw /= nw; }
            hw /= (long)std::min(gz.size(), max_threads);
            gz_threads = (int)std::max<long>(1, std::min<long>(48, hw));
            if (const char *e = getenv("JASPER_INGEST_GZ_THREADS")) gz_threads = std::max(1, atoi(e));
        }
        // a wave of the parallel reader holds ~40 MB of text per thread: the read-ahead budget must have room for two of them
        ahead_cap = std::max<size_t>(ahead_cap, (size_t)gz_threads * (96u << 20));
        for (size_t j = 0; j < gz.size() && j < max_threads; ++j) ahead[(size_t)gz[j]].reset(new GzAhead(paths[gz[j]], ahead_cap, gz_threads));
    }
    ~Reader() { unmap(); if (g) gzclose(g); if (fd >= 0) close(fd); }
};

}  // namespace

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                     \
            return -1;                                                                \
        }                                                                             \
    } while (0)

// returns 0 ok; <0 error (message in err).  `gpu_bytes` / `host_bytes`: how much of the stream each parser handled.
int Table::count_files_gpu(const char *const *paths, int n_paths, uint64_t *gpu_bytes, uint64_t *host_bytes, std::string &err) {
    const bool dbg_t = getenv("JASPER_COUNT_DEBUG") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_in = now_ms();
    // tools/probes/ingest_stages.py: stop after a stage of the chunk loop so that the stages can be timed one on top of the other --
    // 1: file -> pinned buffer only, 2: + the copy to the device, 3: + the parsing kernels (bases not counted); 0 / unset: everything
    const int probe_stage = getenv("JASPER_INGEST_STAGE") ? atoi(getenv("JASPER_INGEST_STAGE")) : 0;
    HIPCHK(hipSetDevice(device));
    // the reader's copies and kernels have a stream of their own: a full buffer of bases is counted on the table's stream (by a thread
    // of its own, below) while the next buffer is being read and parsed here
    if (!ingest_stream) HIPCHK(hipStreamCreateWithFlags(&ingest_stream, hipStreamNonBlocking));
    hipStream_t const table_stream = this->stream;
    (void)table_stream;
    hipStream_t const stream = ingest_stream;         // (every use of `stream` below is the reader's)
    // sizes follow the input (pinning and device allocation cost ~0.1 ms per MB): text chunk <= 128 MiB (+ carry), device
    // base buffer <= 3 GiB (counted and emptied whenever it is that full)
    uint64_t est = 0;
    for (int i = 0; i < n_paths; ++i) {
        struct stat st;
        if (stat(paths[i], &st) == 0 && S_ISREG(st.st_mode)) {
            const size_t L = strlen(paths[i]);
            est += (uint64_t)st.st_size * ((L > 3 && !strcmp(paths[i] + L - 3, ".gz")) ? 5 : 1);
        } else est += 1ull << 32;
    }
    size_t CHUNK = 64u << 20;
    while (CHUNK > (4u << 20) && CHUNK / 2 >= est) CHUNK /= 2;
    if (const char *e = getenv("JASPER_INGEST_CHUNK")) CHUNK = std::max<size_t>(4096, strtoull(e, nullptr, 10));   // tests: many small chunks
    else if (ingest_chunk > CHUNK) CHUNK = ingest_chunk;  // the pinned buffer of an earlier call is kept
    // (the bases of a FASTQ file are less than half of its bytes; the buffer is counted and emptied whenever it is full anyway)
    uint64_t est_bases = est;
    {
        struct stat st0;
        if (n_paths > 0 && stat(paths[0], &st0) == 0 && S_ISREG(st0.st_mode)) {      // (a pipe can be read once: not looked at here)
            FILE *f0 = fopen(paths[0], "rb");
            if (f0) { const int c0 = fgetc(f0); fclose(f0); if (c0 == '@') est_bases = est / 2 + (16u << 20); }
        }
    }
    // Bases go to the counter in pieces of up to 1.5 G (round 3: 3 G).  The counting passes keep 20 bytes of list buffers per base
    // of a piece, and on a GPU whose memory has not been handed out before the first hipMalloc costs ~28 ms per GB (the driver
    // clears what it gives): 64 GB of buffers for a 3-G piece were 1.8 s of a 140-Mb run that takes 1.1 s otherwise, while a
    // further piece costs one more read-modify-write of the table by region_insert_kernel (2 x 16 bytes per slot at ~5 TB/s:
    // 14 ms for 2^31 slots).  The configs[1] input (1.42 G bases) is still one piece.
    const size_t BASES_CAP = (size_t)std::min<uint64_t>(3ull << 29, std::max<uint64_t>(est_bases, 16u << 20));
    uint64_t n_gpu = 0, n_host = 0;
    Reader rd;
    rd.paths = paths;
    rd.n_paths = n_paths;
    rd.begins = ingest_begin;
    rd.ends = ingest_end;
    if (!h_ingest || ingest_chunk < CHUNK) {              // kept with the table between calls
        if (h_ingest) (void)hipHostFree(h_ingest);
        h_ingest = nullptr;
        HIPCHK(hipHostMalloc((void **)&h_ingest, 3 * (2 * CHUNK + 64), hipHostMallocDefault));
        ingest_chunk = CHUNK;
    }
    // Text buffers in pinned memory: while the GPU parses one, a thread reads the next CHUNK of the stream into another (and a third
    // is on its way to the device: OVERLAP below).
    // A chunk's bytes go to offset CHUNK of its buffer; what the chunk before left over (an incomplete record, <= CHUNK) is put
    // right in front of them.
    // OVERLAP (late round 5): the GPU side of a chunk was its copy to the device (1.3 ms per 64 MiB at ~50 GB/s) and then the parsing
    // kernels with their four host waits (0.55 ms), taking turns on ONE device buffer.  Now the text of chunk i + 1 -- read by a
    // thread while chunk i - 1 was parsed -- goes to a SECOND device buffer on a copy stream of its own while chunk i is parsed, and
    // the thread reads chunk i + 2 meanwhile (three pinned buffers).  What chunk i leaves over (an incomplete record) is known only
    // after its parsing: those few bytes follow, in front of the text that is already there.  JASPER_INGEST_OVERLAP=0: as before.
    char *h_two[3] = {h_ingest, h_ingest + ingest_chunk * 2 + 64, h_ingest + 2 * (ingest_chunk * 2 + 64)};
    int cur_buf = 0;
    char *h_buf = h_two[0] + CHUNK;
    std::thread pf;                       // the read ahead
    long pf_got = 0;
    bool pf_active = false;
    int pf_buf = 1;                       // ... into this buffer
    const bool overlap = probe_stage != 1 && probe_stage != 2 && !(getenv("JASPER_INGEST_OVERLAP") && atoi(getenv("JASPER_INGEST_OVERLAP")) == 0);
    bool have_nxt = false;                // overlap: the chunk after this one has been read (nxt_got bytes in h_two[nxt_buf]) and, if it
    long nxt_got = 0;                     //          holds anything, is on its way to the other device buffer (ev_pre)
    int nxt_buf = 0;
    int dev = 0;                          // the device text buffer of the chunk in hand
    if (overlap && !ingest_copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&ingest_copy_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ingest_copy_ev[0], hipEventDisableTiming));      // (one per device buffer: a wait that is still
        HIPCHK(hipEventCreateWithFlags(&ingest_copy_ev[1], hipEventDisableTiming));      //  queued never meets a later recording)
    }
    if (ingest_copy_stream) HIPCHK(jk_stream_wait(ingest_copy_stream));      // (a call that ended early may have left a copy behind)
    uint8_t *d_text0 = reinterpret_cast<uint8_t *>(workspace(WS_INGEST + 0, 2 * (2 * CHUNK + 64), err));
    uint8_t *d_text_two[2] = {d_text0, d_text0 ? d_text0 + 2 * CHUNK + 64 : nullptr};
    uint8_t *d_text = d_text0;
    uint8_t *d_bases = reinterpret_cast<uint8_t *>(workspace(WS_INGEST + 1, BASES_CAP + 2 * CHUNK + 64, err));
    // An input of more than one buffer of bases: TWO buffers, and the counting of a full one is a thread's business (count_device
    // waits for its kernels, reads their statistics, grows the table) while this thread goes on reading, copying and parsing into
    // the other -- files -> bases and bases -> table are two pipeline stages instead of taking turns (tools/probes/ingest_stages.py:
    // 33 GB/s of text for the first alone, 28 with the counting of a 2^29-slot table in between, less with a larger table).
    uint8_t *d_bases_two[2] = {d_bases, nullptr};
    int bases_cur = 0;
    const bool two_buffers = !bases_sink && !probe_stage && est_bases > BASES_CAP && !getenv("JASPER_INGEST_ONE_BUFFER");
    if (two_buffers) {
        d_bases_two[1] = reinterpret_cast<uint8_t *>(workspace(WS_INGEST + 6, BASES_CAP + 2 * CHUNK + 64, err));
        if (!d_bases_two[1]) return -1;
    }
    std::thread counter;
    bool counter_active = false;
    int counter_rc = 0;
    std::string counter_err;
    auto drain = [&]() -> int {              // the buffer handed to the counter has been counted
        if (counter_active) {
            counter.join();
            counter_active = false;
            if (counter_rc) { err = counter_err; return counter_rc; }
        }
        return 0;
    };
    struct DrainOnExit { std::thread &t; bool &active; ~DrainOnExit() { if (active && t.joinable()) t.join(); } } drain_on_exit{counter, counter_active};
    const size_t max_blocks = (2 * CHUNK + IG_BYTES - 1) / IG_BYTES + 1;
    uint32_t *d_blk = reinterpret_cast<uint32_t *>(workspace(WS_INGEST + 2, (3 * max_blocks + 16) * 4, err));
    if (!d_text || !d_bases || !d_blk) return -1;
    uint32_t *d_blk_nl = d_blk, *d_blk_emit = d_blk + max_blocks, *d_blk_out = d_blk_emit + max_blocks;
    unsigned int *d_flags = reinterpret_cast<unsigned int *>(d_blk_out + max_blocks);
    size_t scan_tmp_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, scan_tmp_bytes, d_blk_nl, d_blk_nl, 0u, max_blocks, rocprim::plus<uint32_t>(), stream));
    void *d_scan_tmp = workspace(WS_INGEST + 3, scan_tmp_bytes + 256, err);
    if (!d_scan_tmp) return -1;

    struct Report { uint64_t *g, *h; uint64_t &ng, &nh; ~Report() { if (g) *g = ng; if (h) *h = nh; } } report{gpu_bytes, host_bytes, n_gpu, n_host};
    uint64_t bases_len = 0;
    if (dbg_t) fprintf(stderr, "[ingest] buffers ready %.1f ms after the call (pinned 3 x %zu MiB, device text 2 x %zu MiB + bases %.2f GB)\n", now_ms() - t_in, (2 * CHUNK) >> 20,
                       (2 * CHUNK) >> 20, (double)(BASES_CAP + 2 * CHUNK) / 1e9);
    auto flush_bases = [&]() -> int {
        if (!bases_len) return 0;
        if (two_buffers) {
            if (int rc = drain()) return rc;                 // (the other buffer is free again)
            if (dbg_t) fprintf(stderr, "[ingest] %.1f ms after the call: %llu bases to the counter's thread\n", now_ms() - t_in, (unsigned long long)bases_len);
            const uint8_t *full = d_bases;
            const uint64_t full_n = bases_len;
            counter_active = true;
            counter = std::thread([this, full, full_n, &counter_rc, &counter_err, dbg_t, t_in, now_ms] {
                counter_rc = count_device(full, full_n, counter_err);
                if (dbg_t) fprintf(stderr, "[ingest] %.1f ms after the call: counted\n", now_ms() - t_in);
            });
            bases_cur ^= 1;
            d_bases = d_bases_two[bases_cur];
            bases_len = 0;
            return 0;
        }
        if (dbg_t) fprintf(stderr, "[ingest] %.1f ms after the call: %llu bases to the counter\n", now_ms() - t_in, (unsigned long long)bases_len);
        const int rc = probe_stage ? 0 : bases_sink ? bases_sink(d_bases, bases_len) : count_device(d_bases, bases_len, err);
        if (dbg_t) fprintf(stderr, "[ingest] %.1f ms after the call: counted\n", now_ms() - t_in);
        bases_len = 0;
        return rc;
    };
    // what the host parser emits (whole records, a few MB at a time): counted from host memory, or (feed) collected in a
    // device buffer and handed over when that is full / at the end of the stream
    const size_t HOST_ACC = 256u << 20;
    size_t host_acc = 0;
    uint8_t *d_host = nullptr;
    auto flush_host = [&]() -> int {
        if (!host_acc) return 0;
        HIPCHK(jk_stream_wait(stream));
        const int rc = bases_sink(d_host, host_acc);
        host_acc = 0;
        return rc;
    };
    auto host_bases = [&](const char *b, size_t m) -> int {
        if (!bases_sink) return this->count_host(b, m, err);
        if (!d_host) {
            d_host = reinterpret_cast<uint8_t *>(workspace(WS_INGEST + 5, HOST_ACC + 64, err));
            if (!d_host) return -1;
        }
        for (size_t off = 0; off < m;) {
            if (host_acc == HOST_ACC) { if (int rc = flush_host()) return rc; }
            size_t n = std::min(HOST_ACC - host_acc, m - off);
            if (off + n < m) {                             // cut after a separator, so that no k-mer spans two batches
                size_t cut = n;
                while (cut > 0 && b[off + cut - 1] != 'N') --cut;
                if (cut > 0) n = cut;
                else if (host_acc) { if (int rc = flush_host()) return rc; continue; }
                else { err = "a run of bases without a separator longer than the feed buffer"; return -1; }
            }
            HIPCHK(hipMemcpyAsync(d_host + host_acc, b + off, n, hipMemcpyHostToDevice, stream));
            HIPCHK(jk_stream_wait(stream));                // (b is the parser's buffer: reused as soon as this returns)
            host_acc += n;
            off += n;
            if (off < m) { if (int rc = flush_host()) return rc; }
        }
        return 0;
    };
    // the host state machine takes over from here (and keeps the rest of the stream)
    FastxParser hp(host_bases);
    auto host_rest = [&](int mode, const char *first, size_t first_n) -> int {
        if (ingest_copy_stream) HIPCHK(jk_stream_wait(ingest_copy_stream));      // (a copy under way reads a pinned buffer)
        if (int rc = flush_bases()) return rc;
        if (int rc = drain()) return rc;                     // (the host parser's bases are counted by this thread, on the table's stream)
        hp.resume(mode);
        int rc = hp.feed(first, first_n);
        n_host += first_n;
        if (have_nxt) {                                        // the chunk that has been read already comes next in the stream
            have_nxt = false;
            if (nxt_got < 0) { err = rd.err; return -1; }
            if (!rc && nxt_got > 0) { rc = hp.feed(h_two[nxt_buf] + CHUNK, (size_t)nxt_got); n_host += (uint64_t)nxt_got; }
        }
        if (pf_active) {                                       // then the chunk a thread has been reading ahead
            pf.join();
            pf_active = false;
            if (pf_got < 0) { err = rd.err; return -1; }
            if (!rc && pf_got > 0) { rc = hp.feed(h_two[pf_buf] + CHUNK, (size_t)pf_got); n_host += (uint64_t)pf_got; }
        }
        char *rb = h_two[cur_buf];                             // (`first` and the read-ahead have been consumed: the buffer is free)
        while (!rc) {
            const long got = rd.read(rb, CHUNK);
            if (got < 0) { err = rd.err; return -1; }
            if (got == 0) break;
            rc = hp.feed(rb, (size_t)got);
            n_host += (uint64_t)got;
        }
        if (!rc) rc = hp.finish();
        if (rc && !hp.error().empty()) err = hp.error();
        if (!rc && bases_sink) rc = flush_host();
        return rc;
    };

    int mode = 0;                 // 0 unknown, 1 FASTA, 2 FASTQ (FastxParser's numbering)
    size_t carry = 0;             // bytes at the start of h_buf left over from the previous chunk (an incomplete record / line)
    struct JoinOnExit { std::thread &t; bool &active; ~JoinOnExit() { if (active && t.joinable()) t.join(); } } join_on_exit{pf, pf_active};
    for (;;) {
        if (g_cancel.load(std::memory_order_relaxed)) { err = "cancelled"; return -1; }      // (jasper_request_cancel: the caller is on its way out)
        if (carry > CHUNK) return host_rest(mode, h_buf, carry);                  // a line / record longer than a chunk
        long got;
        bool pre = false;                                                         // this chunk's text (not the carry) is on its way to d_text_two[dev] already
        if (have_nxt) {                                                           // the chunk read two chunks ago
            have_nxt = false;
            got = nxt_got;
            pre = got > 0;
            char *nb = h_two[nxt_buf] + CHUNK - carry;
            if (carry) memcpy(nb, h_buf, carry);
            h_buf = nb;
            cur_buf = nxt_buf;
            if (pre) dev ^= 1;
        } else if (pf_active) {                                                   // the chunk read while the last one was parsed
            pf.join();
            pf_active = false;
            got = pf_got;
            char *nb = h_two[pf_buf] + CHUNK - carry;
            if (carry) memcpy(nb, h_buf, carry);                                  // (h_buf: where the loop below left the incomplete record)
            h_buf = nb;
            cur_buf = pf_buf;
        } else {
            char *nb = h_two[cur_buf] + CHUNK - carry;
            if (carry && nb != h_buf) memmove(nb, h_buf, carry);
            h_buf = nb;
            got = rd.read(h_buf + carry, CHUNK);
        }
        if (got < 0) { err = rd.err; return -1; }
        size_t n = carry + (size_t)got;
        const bool eof = got == 0;
        if (n == 0) break;
        if (mode == 0) {
            // format from the first byte of the stream (:134-148); leading empty lines make the reference's peek() see '\n'
            if (h_buf[0] == '>') mode = 1;
            else if (h_buf[0] == '@') mode = 2;
            else return host_rest(0, h_buf, n);                                   // (the host parser words the error)
        }
        if (eof) return host_rest(mode, h_buf, n);                                // the tail: at most one incomplete record / line
        // a thread reads the next chunk of the stream into a free buffer
        auto read_ahead = [&]() {
            int b = (cur_buf + 1) % 3;
            if (have_nxt && b == nxt_buf) b = (b + 1) % 3;
            pf_buf = b;
            char *dst = h_two[b] + CHUNK;
            pf_active = true;
            pf = std::thread([&rd, &pf_got, dst, CHUNK] { pf_got = rd.read(dst, CHUNK); });
        };
        // overlap: the chunk the thread has read becomes "the next one" -- its text goes to the other device buffer on the copy stream
        // (nobody reads that buffer: the parsing of the chunk before this one was waited for) -- and the thread reads the one after it
        auto look_ahead = [&]() -> int {
            if (have_nxt || !pf_active) return 0;
            pf.join();
            pf_active = false;
            have_nxt = true;
            nxt_got = pf_got;
            nxt_buf = pf_buf;
            if (nxt_got > 0) {
                HIPCHK(hipMemcpyAsync(d_text_two[dev ^ 1] + CHUNK, h_two[nxt_buf] + CHUNK, (size_t)nxt_got, hipMemcpyHostToDevice, ingest_copy_stream));
                HIPCHK(hipEventRecord(ingest_copy_ev[dev ^ 1], ingest_copy_stream));
                read_ahead();
            }
            return 0;
        };
        if (!overlap) read_ahead();
        if (probe_stage == 1) { n_gpu += n; carry = 0; continue; }
        d_text = d_text_two[dev] + CHUNK - carry;
        if (pre) {                                                                // the text is there (or on its way): the carried bytes go in front of it
            HIPCHK(hipStreamWaitEvent(stream, ingest_copy_ev[dev], 0));
            if (carry) HIPCHK(hipMemcpyAsync(d_text, h_buf, carry, hipMemcpyHostToDevice, stream));
        } else HIPCHK(hipMemcpyAsync(d_text, h_buf, n, hipMemcpyHostToDevice, stream));
        if (probe_stage == 2) { HIPCHK(jk_stream_wait(stream)); n_gpu += n; carry = 0; continue; }
        HIPCHK(hipMemsetAsync(d_flags, 0, 8, stream));
        const uint32_t nblk = (uint32_t)((n + IG_BYTES - 1) / IG_BYTES);
        hipLaunchKernelGGL(ig_count_nl_kernel, dim3(nblk), dim3(IG_THREADS), 0, stream, d_text, (uint64_t)n, d_blk_nl, d_flags);
        HIPCHK(rocprim::exclusive_scan(d_scan_tmp, scan_tmp_bytes, d_blk_nl, d_blk_nl, 0u, (size_t)nblk + 1, rocprim::plus<uint32_t>(), stream));
        uint32_t n_lines = 0;
        unsigned int flags = 0;
        HIPCHK(hipMemcpyAsync(&n_lines, d_blk_nl + nblk, 4, hipMemcpyDeviceToHost, stream));      // (entry nblk of the exclusive scan = total)
        HIPCHK(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, stream));
        if (overlap) {                                                            // (while the first parsing kernels run)
            if (pf_active) { if (int rc = look_ahead()) return rc; }
            else if (!have_nxt) read_ahead();                                     // (the stream's first chunk: nothing has been read ahead yet)
        }
        HIPCHK(jk_stream_wait(stream));
        if (flags & IGF_CR) return host_rest(mode, h_buf, n);
        const uint64_t use_lines = mode == 2 ? (uint64_t)(n_lines / 4) * 4 : n_lines;
        if (use_lines == 0) {                                                     // not even one whole record in the chunk
            carry = n;
            continue;
        }
        uint32_t *d_nlpos = reinterpret_cast<uint32_t *>(workspace(WS_INGEST + 4, ((size_t)n_lines + 16) * 4, err));
        if (!d_nlpos) return -1;
        hipLaunchKernelGGL(ig_nlpos_kernel, dim3(nblk), dim3(IG_THREADS), 0, stream, d_text, (uint64_t)n, d_blk_nl, d_nlpos);
        uint32_t end_nl = 0;
        HIPCHK(hipMemcpyAsync(&end_nl, d_nlpos + (use_lines - 1), 4, hipMemcpyDeviceToHost, stream));
        if (mode == 2) hipLaunchKernelGGL(ig_check_fastq_kernel, dim3(1024), dim3(256), 0, stream, d_text, d_nlpos, use_lines / 4, d_flags);
        HIPCHK(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(jk_stream_wait(stream));
        if (flags & IGF_BAD) return host_rest(mode, h_buf, n);
        const uint64_t n_use = (uint64_t)end_nl + 1;                              // whole records / lines only
        const uint32_t ublk = (uint32_t)((n_use + IG_BYTES - 1) / IG_BYTES);
        hipLaunchKernelGGL(ig_emit_kernel<0>, dim3(ublk), dim3(IG_THREADS), 0, stream, d_text, n_use, d_blk_nl, d_nlpos, mode == 2 ? 1 : 0, d_blk_emit,
                           (const uint32_t *)nullptr, (uint8_t *)nullptr);
        HIPCHK(rocprim::exclusive_scan(d_scan_tmp, scan_tmp_bytes, d_blk_emit, d_blk_out, 0u, (size_t)ublk + 1, rocprim::plus<uint32_t>(), stream));
        uint32_t n_emit = 0;
        HIPCHK(hipMemcpyAsync(&n_emit, d_blk_out + ublk, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(jk_stream_wait(stream));
        if (bases_len + n_emit > BASES_CAP + 2 * CHUNK) { if (int rc = flush_bases()) return rc; }
        hipLaunchKernelGGL(ig_emit_kernel<1>, dim3(ublk), dim3(IG_THREADS), 0, stream, d_text, n_use, d_blk_nl, d_nlpos, mode == 2 ? 1 : 0, d_blk_emit,
                           d_blk_out, d_bases + bases_len);
        HIPCHK(hipGetLastError());
        HIPCHK(jk_stream_wait(stream));           // h_buf is rewritten next
        bases_len += n_emit;
        n_gpu += n_use;
        if (bases_len >= BASES_CAP) { if (int rc = flush_bases()) return rc; }
        carry = n - (size_t)n_use;
        h_buf += n_use;                                                           // (moved in front of the next chunk at the top of the loop)
        if (overlap) { if (int rc = look_ahead()) return rc; }                    // (the stream's second chunk: read while this one was parsed)
    }
    if (int rc = flush_bases()) return rc;
    return drain();
}

// ---- feed: the same reader and parsers, the bases handed to the caller instead of counted -----------------------------
struct Table::Feed {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    enum { IDLE, READY, TAKEN, DONE } state = IDLE;
    const uint8_t *ptr = nullptr;
    uint64_t len = 0;
    int rc = 0;
    std::string err;
    bool abort = false;
    std::vector<std::string> paths;
    std::vector<const char *> cpaths;
    std::vector<int64_t> begins, ends;
};

int Table::feed_start(const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths, std::string &err) {
    if (feed) { err = "a feed is already running on this table"; return -1; }
    if (n_paths < 0 || (n_paths && !paths)) { err = "bad argument"; return -1; }
    feed = new Feed();
    Feed *F = feed;
    for (int i = 0; i < n_paths; ++i) F->paths.emplace_back(paths[i]);
    for (const std::string &p : F->paths) F->cpaths.push_back(p.c_str());
    if (begins && ends) { F->begins.assign(begins, begins + n_paths); F->ends.assign(ends, ends + n_paths); }
    bases_sink = [F](const uint8_t *d_bases, uint64_t n) -> int {
        std::unique_lock<std::mutex> lk(F->m);
        F->ptr = d_bases;
        F->len = n;
        F->state = Feed::READY;
        F->cv.notify_all();
        F->cv.wait(lk, [F] { return F->state == Feed::IDLE || F->abort; });
        if (F->abort) { F->err = "feed stopped by the caller"; return -1; }
        return 0;
    };
    F->th = std::thread([this, F] {
        ingest_gpu_bytes = ingest_host_bytes = 0;
        ingest_begin = F->begins.empty() ? nullptr : F->begins.data();
        ingest_end = F->ends.empty() ? nullptr : F->ends.data();
        std::string e;
        const int rc = count_files_gpu(F->cpaths.data(), (int)F->cpaths.size(), &ingest_gpu_bytes, &ingest_host_bytes, e);
        ingest_begin = ingest_end = nullptr;
        std::unique_lock<std::mutex> lk(F->m);
        F->rc = rc;
        if (rc && F->err.empty()) F->err = e;
        F->state = Feed::DONE;
        F->cv.notify_all();
    });
    return 0;
}

int Table::feed_next(const void **d_bases, uint64_t *n, std::string &err) {
    if (!feed) { err = "no feed on this table"; return -1; }
    Feed *F = feed;
    std::unique_lock<std::mutex> lk(F->m);
    if (F->state == Feed::TAKEN) { err = "feed_next before feed_release of the last batch"; return -1; }
    F->cv.wait(lk, [F] { return F->state == Feed::READY || F->state == Feed::DONE; });
    if (F->state == Feed::DONE) {
        lk.unlock();
        if (F->th.joinable()) F->th.join();
        const int rc = F->rc;
        if (rc) err = F->err;
        bases_sink = nullptr;
        delete F;
        feed = nullptr;
        *d_bases = nullptr;
        *n = 0;
        return rc ? -1 : 0;
    }
    F->state = Feed::TAKEN;
    *d_bases = F->ptr;
    *n = F->len;
    return 0;
}

int Table::feed_release(std::string &err) {
    if (!feed) { err = "no feed on this table"; return -1; }
    Feed *F = feed;
    std::unique_lock<std::mutex> lk(F->m);
    if (F->state != Feed::TAKEN) { err = "feed_release without a batch taken"; return -1; }
    F->state = Feed::IDLE;
    F->cv.notify_all();
    return 0;
}

void Table::feed_stop() {
    if (!feed) return;
    Feed *F = feed;
    {
        std::unique_lock<std::mutex> lk(F->m);
        F->abort = true;
        F->cv.notify_all();
    }
    if (F->th.joinable()) F->th.join();
    bases_sink = nullptr;
    delete F;
    feed = nullptr;
}

}  // namespace jk
