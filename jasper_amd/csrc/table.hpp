// table.hpp -- the k-mer count table resident in HBM and its host-side owner.
//
// Replaces (function, not layout) Jellyfish's lock-free hash + sorted DB + mmap query:
//   JF::include/jellyfish/large_hash_array.hpp:291,509-597,674-752  (add / claim_key / add_val)
//   JF::include/jellyfish/hash_counter.hpp:91-115,200-238             (add, double_size)
//   JF::include/jellyfish/binary_dumper.hpp:148-199                   (query: exact count or 0)
//   JF::sub_commands/histo_main.cc:34-44                              (histogram)
#pragma once
#include "kmer.hpp"
#include <atomic>
#include <functional>
#include <string>
#include <vector>

namespace jk {

// set by jasper_request_cancel (a driver that is about to exit on an error elsewhere): the long-running host loops -- read files ->
// table, the .jf writer -- stop between two chunks / blocks with an error instead of finishing their work first
extern std::atomic<int> g_cancel;


// what kernels receive (by value)
constexpr uint32_t MAX_SHARDS = 8;   // the GPUs of one node
struct TableDev {
    unsigned long long *slots;  // 2 words per slot: tag, count
    uint64_t mask;              // nslots - 1
    int s;                      // log2(nslots)
    int B;                      // 2k
    int k;
    unsigned long long *stats;  // [0] distinct keys  [1] spilled insertions  [2] k-mer occurrences added  [3] fatal
    unsigned long long *spill;  // 3 words per spilled insertion: hash.hi, hash.lo, increment
    uint64_t spill_cap;
    // Owner-sharded table (multi-GPU): READS of a key go to the slot array of the GPU that owns the key; shard[i] is the
    // i-th owner's slot array (this GPU's own, or a peer's HBM mapped over xGMI), all of one geometry.  nshard <= 1: the
    // table is whole and `slots` is read.  Insertions always go to `slots`.
    const unsigned long long *shard[MAX_SHARDS];
    uint32_t nshard;
    // wide remainders (kmer.hpp: wide_rem): the low 64 remainder bits of slot i are ext[i]; null when the tag holds the whole
    // remainder.  A slot's ext word is valid once its count is non-zero: the claimant writes tag (CAS), ext, then adds its count.
    unsigned long long *ext;
};

// the hash of the key stored in slot i (tag != 0)
__device__ __forceinline__ u128 slot_hash(const TableDev &T, uint64_t i, unsigned long long tag) {
    const uint32_t off = (uint32_t)(tag & (MAXPROBE - 1));
    const uint64_t rem = (tag & ~OCC) >> OFFBITS;
    const uint64_t home = (i - off) & T.mask;
    if (T.ext) return hash_from_wide(home, rem, T.ext[i], T.B, T.s);
    return hash_from(home, rem, T.B, T.s);
}
// Insert-or-add / insert-or-assign for a wide table.  A claimant writes tag (compare-and-swap), ext, then its count (release);
// a lane that meets a matching tag whose count is still zero cannot tell yet whether that is its own key.  It must not wait in
// place: the claimant may be a lane of the SAME wave whose side of the branch has not run yet.  So the loop below leaves only
// when every lane of the wave is done -- each trip is then complete for all lanes before the next one starts -- and the
// undecided lane simply looks at the same slot again on the next trip (a bounded number of times; then it reports 0 and the
// caller spills the insertion, to be re-inserted later).  assign: the count is set to `val` instead of being added to.
__device__ __forceinline__ int table_put_wide(const TableDev &T, u128 h, unsigned long long val, bool assign) {
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = tag_rem_of(h, T.B, T.s);
    const unsigned long long ext = ext_of(h, T.B, T.s);
    int result = -1;
    uint32_t off = 0, waited = 0;
    for (;;) {
        if (result < 0) {
            const uint64_t slot = (home + off) & T.mask;
            const unsigned long long want = tag_of(rem, off);
            unsigned long long *p = T.slots + 2 * slot;
            unsigned long long cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                cur = atomicCAS(p, 0ull, want);
                if (cur == 0ull) {
                    __hip_atomic_store(T.ext + slot, ext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p + 1, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // (the claimant is the only writer until the count is non-zero)
                    result = 2;
                }
            }
            if (result < 0) {
                bool next = true;
                if (cur == want) {
                    const unsigned long long c = __hip_atomic_load(p + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (c == 0ull) {                       // claimed, key not complete yet: same slot again on the next trip
                        next = false;
                        if (++waited > (1u << 16)) result = 0;
                    } else if (__hip_atomic_load(T.ext + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ext) {
                        if (assign) __hip_atomic_store(p + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else __hip_atomic_fetch_add(p + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        result = 1;
                    }
                }
                if (result < 0 && next && ++off >= MAXPROBE) result = 0;
            }
        }
        if (__ballot(result < 0) == 0ull) break;
    }
    return result;
}

// The owner of a key among n shards: a second mix of the hash, so that the keys of one owner are spread evenly over the
// slots of its table whatever the table size is.
JK_HD uint32_t owner_of(u128 h, uint32_t n) {
    uint32_t x = (uint32_t)h.lo ^ (uint32_t)(h.lo >> 32) ^ (uint32_t)h.hi;
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return (uint32_t)(((uint64_t)x * n) >> 32);
}
// slot array a lookup of hash h has to read
__device__ __forceinline__ const unsigned long long *read_slots(const TableDev &T, u128 h) {
    if (T.nshard <= 1) return T.slots;
    const uint32_t o = owner_of(h, T.nshard);
    const unsigned long long *b = T.shard[0];
#pragma unroll
    for (uint32_t i = 1; i < MAX_SHARDS; ++i) b = (o == i) ? T.shard[i] : b;   // selects on constant indices: no scratch
    return b;
}

enum { ST_DISTINCT = 0, ST_SPILL = 1, ST_OCCURRENCES = 2, ST_FATAL = 3, ST_WORDS = 8 };

// insert-or-add `inc` for the key whose mixed hash is h. Returns 0 if no slot within MAXPROBE, 1 if the key
// existed, 2 if this call claimed a new slot (callers batch the distinct-key counter: one same-address atomic
// per new key would serialise the whole chip on a single L2 channel).
__device__ __forceinline__ int table_add(const TableDev &T, u128 h, unsigned long long inc) {
    if (T.ext) return table_put_wide(T, h, inc, false);
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = rem_of(h, T.B, T.s);
    for (uint32_t off = 0; off < MAXPROBE; ++off) {
        const uint64_t slot = (home + off) & T.mask;
        const unsigned long long want = tag_of(rem, off);
        unsigned long long *p = T.slots + 2 * slot;
        // a stale (L1) view can only show "empty" where a tag has since been written; the CAS below
        // then returns the real occupant, so plain loads are safe here.
        unsigned long long cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int fresh = 1;
        if (cur == 0ull) {
            cur = atomicCAS(p, 0ull, want);
            if (cur == 0ull) {
                fresh = 2;
                cur = want;
            }
        }
        if (cur == want) {
            __hip_atomic_fetch_add(p + 1, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return fresh;
        }
    }
    return 0;
}

// same as table_add, but the tag of the home slot has already been loaded (`cur0`): lets a thread keep several
// first probes in flight before resolving them
__device__ __forceinline__ int table_add_prefetched(const TableDev &T, u128 h, unsigned long long inc, unsigned long long cur0) {
    if (T.ext) return table_add(T, h, inc);      // (wide tables: the prefetched tag alone does not decide anything)
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = rem_of(h, T.B, T.s);
    for (uint32_t off = 0; off < MAXPROBE; ++off) {
        const uint64_t slot = (home + off) & T.mask;
        const unsigned long long want = tag_of(rem, off);
        unsigned long long *p = T.slots + 2 * slot;
        unsigned long long cur = off == 0 ? cur0 : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int fresh = 1;
        if (cur == 0ull) {
            cur = atomicCAS(p, 0ull, want);
            if (cur == 0ull) { fresh = 2; cur = want; }
        }
        if (cur == want) {
            __hip_atomic_fetch_add(p + 1, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return fresh;
        }
    }
    return 0;
}
__device__ __forceinline__ void table_spill(const TableDev &T, u128 h, unsigned long long inc) {
    unsigned long long idx = atomicAdd(&T.stats[ST_SPILL], 1ull);
    if (idx < T.spill_cap) {
        T.spill[3 * idx + 0] = h.hi;
        T.spill[3 * idx + 1] = h.lo;
        T.spill[3 * idx + 2] = inc;
    } else {
        atomicExch(&T.stats[ST_FATAL], 1ull);
    }
}

// returns 1 when a new distinct key was created
__device__ __forceinline__ unsigned table_add_or_spill(const TableDev &T, u128 h, unsigned long long inc) {
    const int r = table_add(T, h, inc);
    if (r == 2) return 1u;
    if (r == 0) {
        unsigned long long idx = atomicAdd(&T.stats[ST_SPILL], 1ull);
        if (idx < T.spill_cap) {
            T.spill[3 * idx + 0] = h.hi;
            T.spill[3 * idx + 1] = h.lo;
            T.spill[3 * idx + 2] = inc;
        } else {
            atomicExch(&T.stats[ST_FATAL], 1ull);
        }
    }
    return 0u;
}

// exact 64-bit count of the key whose mixed hash is h, or 0
__device__ __forceinline__ unsigned long long table_get(const TableDev &T, u128 h) {
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = tag_rem_of(h, T.B, T.s);
    const unsigned long long *S = read_slots(T, h);
    const bool wide = T.ext != nullptr;          // (a wide table is never sharded: attach refuses it)
    const unsigned long long ext = wide ? ext_of(h, T.B, T.s) : 0ull;
    for (uint32_t off = 0; off < MAXPROBE; ++off) {
        const uint64_t slot = (home + off) & T.mask;
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(S + 2 * slot);  // tag + count, one 16-B load
        if (e.x == tag_of(rem, off) && (!wide || T.ext[slot] == ext)) return e.y;
        if (e.x == 0ull) return 0ull;
    }
    return 0ull;
}

// the same with the home slot already loaded (callers put several home-slot loads in flight before resolving them)
__device__ __forceinline__ unsigned long long table_get_prefetched(const TableDev &T, u128 h, ulonglong2 e0) {
    if (T.ext) return table_get(T, h);
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = rem_of(h, T.B, T.s);
    if (e0.x == tag_of(rem, 0)) return e0.y;
    if (e0.x == 0ull) return 0ull;
    const unsigned long long *S = read_slots(T, h);
    for (uint32_t off = 1; off < MAXPROBE; ++off) {
        const uint64_t slot = (home + off) & T.mask;
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(S + 2 * slot);
        if (e.x == tag_of(rem, off)) return e.y;
        if (e.x == 0ull) return 0ull;
    }
    return 0ull;
}

// count as the reference's DB file reports it: min(count, 2^32-1)  (JF::include/jellyfish/binary_dumper.hpp:36-40)
__device__ __forceinline__ uint32_t clamp32(unsigned long long c) { return c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c; }

// 16 text bytes (4 little-endian words, first base in the low byte of w[0]) -> 2-bit codes (first base in the top bit
// pair) and an "is no base" bit per byte (first base in bit 15).  Four bases per step: code = x ^ (x >> 1) with x = bits
// 1..2 of the letter (A,C,G,T -> 0,1,2,3; case-insensitive); a byte is a base iff the letter that code stands for equals
// the byte with its case bit cleared (v_perm_b32 used as a 4-entry table).  ~4 VALU ops per base instead of ~14.
__device__ __forceinline__ void encode16(const uint32_t w[4], uint32_t &codes, uint32_t &inv) {
    codes = 0;
    inv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x0 = (w[i] >> 1) & 0x03030303u;
        const uint32_t x = x0 ^ ((x0 >> 1) & 0x01010101u);
        const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u, x);
        const uint32_t d = expect ^ (w[i] & 0xDFDFDFDFu);
        const uint32_t nz = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;
        codes = (codes << 8) | ((x * 0x40100401u) >> 24);
        inv = (inv << 4) | ((((nz >> 7) * 0x08040201u) >> 24) & 0xFu);
    }
}
// the 16 bytes text[pos .. pos+16) of a text of n bytes (positions outside it read as 'N')
struct Raw16 { uint32_t w[4]; };
__device__ __forceinline__ Raw16 load16(const uint8_t *__restrict__ text, int64_t pos, int64_t n) {
    struct __attribute__((packed, aligned(1))) V16 { uint32_t w[4]; };
    Raw16 r;
    if (pos >= 0 && pos + 16 <= n) {
        const V16 v = *reinterpret_cast<const V16 *>(text + pos);
        r.w[0] = v.w[0]; r.w[1] = v.w[1]; r.w[2] = v.w[2]; r.w[3] = v.w[3];
    } else {
        r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0x4E4E4E4Eu;   // "NNNN"
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t p = pos + j;
            if (p >= 0 && p < n) r.w[j >> 2] = (r.w[j >> 2] & ~(0xFFu << (8 * (j & 3)))) | ((uint32_t)text[p] << (8 * (j & 3)));
        }
    }
    return r;
}
__device__ __forceinline__ void stage16(const uint8_t *__restrict__ text, int64_t pos, int64_t n, uint32_t &codes, uint32_t &inv) {
    const Raw16 r = load16(text, pos, n);
    encode16(r.w, codes, inv);
}

// H2 / Appendix A.3: k-mer of a string cut at the first non-ACGTacgt byte (or at k) and right-filled with 'A'
// (JF::include/jellyfish/mer_dna.hpp:525-542).  `get(i)` returns byte i of the string, n = its length.
template <typename Get>
__device__ __forceinline__ u128 encode_padded(int k, long n, Get get) {
    u128 m = mk(0, 0);
    int t = 0;
    const int lim = n < k ? (int)n : k;
    for (; t < lim; ++t) {
        int c = code(get(t));
        if (c < 0) break;
        m = bor(shl(m, 2), mk(0, (uint64_t)c));
    }
    if (t < k) m = shl(m, 2 * (k - t));
    return m;
}

// ---- host side ----------------------------------------------------------------------------------
struct Table {
    int device = 0;
    int k = 0;
    hipStream_t stream = nullptr;
    TableDev d{};
    uint64_t nslots = 0;
    // staging for host -> device streaming
    uint8_t *d_stage[2] = {nullptr, nullptr};
    uint8_t *h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    unsigned long long *h_stats = nullptr;  // pinned mirror of stats
    double grow_at = 0.5;
    // HIP-event timing of the counting kernels since reset_timing() (for bench.py's roofline leg)
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
    double count_kernel_ms = 0;
    uint64_t count_launches = 0;
    void reset_timing() { count_kernel_ms = 0; count_launches = 0; count_partitioned_launches = 0; for (double &m : part_stage_ms) m = 0; }
    int launch_count(const uint8_t *d_piece, uint64_t len, uint64_t emit_from, std::string &err);
    // partitioned (atomic-free) path, count_part.hip; `geom` is an opaque PartGeom
    bool partition_geometry(uint64_t piece_bases, void *geom_out) const;
    int launch_count_partitioned(const uint8_t *d_piece, uint64_t len, uint64_t emit_from, const void *geom, std::string &err);
    uint64_t count_partitioned_launches = 0;
    static constexpr int N_STAGES = 8;
    hipEvent_t ev_stage_t[N_STAGES + 1] = {};   // stage boundaries of the last partitioned piece
    // count_part.hip: part1, part2, region_insert, deferred; with the lists exchanged: part1, part2 by owner (+ dedupe), region_insert, -, deferred
    double part_stage_ms[N_STAGES] = {};
    int part_stage_n = 5;          // stages the last piece recorded
    int count_path = 0;            // path of the last piece: 0 direct kernel, 1 count_part.hip, 3 count_part.hip with the lists exchanged between GPUs
    bool part_stage_pending = false;
    // a partitioned piece whose lists overflowed beyond the deferred list abandons itself before it touches the table
    // (count_part.hip: part_decide_kernel); count_device then counts the piece through the direct kernel and stays with it
    unsigned long long *part_defer_header = nullptr;
    bool part_slots_dirty_before = false, part_off = false;
    bool xchg_partitioned = false;  // xchg_partition has recorded its stage events since the last xchg_insert
    bool xchg_deduped = false;      // ... and xchg_dedupe its own
    // Multiplicity histogram taken for free while region_insert_kernel writes the final region images back: valid when one
    // partitioned piece counted the whole input into an empty table and nothing had to take the deferred (direct) path.
    // d_histo: [0, HISTO_WORDS) fused histogram + "deferred records existed" flag, [HISTO_WORDS, 2*HISTO_WORDS) scratch of
    // histo_kernel.  Every other mutation of the table drops the cached histogram.
    static constexpr int HISTO_WORDS = 10002 + 6;
    unsigned long long *d_histo = nullptr;
    bool histo_cached = false, histo_request = false;
    double dup_ratio = 1.0;   // new distinct keys per k-mer of the last piece (sizes the next piece)
    uint64_t size_hint = 0;   // caller's expected number of distinct k-mers (`jellyfish count -s`); 0 = none given
    // grow-only device workspace reused by the polisher across calls (hipMalloc of GBs costs far more than the kernels)
    struct WsBuf { void *p = nullptr; size_t bytes = 0; };
    // The polisher may run a batch as several LANES at once (groups of chunk records, each with a stream, a host thread and a
    // workspace of its own: polish_host.hip): lane 0 uses the table's stream and the first WS_POLISH_MAX slots, lane l > 0
    // polish_stream[l] and the slots from WS_LANE0 + (l - 1) * WS_POLISH_MAX on.
    static constexpr int POLISH_LANES_MAX = 4;
    static constexpr int WS_POLISH_MAX = 40, WS_COUNT = 40, WS_INGEST = 44, WS_XCHG = 51, WS_HOSTBASES = 55, WS_LANE0 = 56,
                         WS_SLOTS = WS_LANE0 + (POLISH_LANES_MAX - 1) * WS_POLISH_MAX;
    WsBuf ws[WS_SLOTS];   // 0..WS_POLISH_MAX-1: polisher (polish_host.hip, in allocation order); WS_COUNT..+3: partitioned counting
    hipStream_t polish_stream[POLISH_LANES_MAX] = {nullptr, nullptr, nullptr, nullptr};      // [0] unused (= stream); created on first use
    hipEvent_t polish_ev = nullptr;
    uint64_t polish_calls = 0;           // polishing calls this table has served (run_polish: a table that is polished again and again gets lanes)
    hipStream_t jf_stream = nullptr;        // write_jf's own (jfwrite.hip)
    hipStream_t ingest_stream = nullptr;    // the file reader's copies and parsing kernels (ingest_gpu.hip): they run while the table's stream counts the bases before
    hipStream_t ingest_copy_stream = nullptr;   // ... the text of the next chunk on its way to the second device buffer while this one is parsed
    hipEvent_t ingest_copy_ev[2] = {nullptr, nullptr};
    void *workspace(int id, size_t bytes, std::string &err);
    void wait_streams();                    // every stream of this table (its own, the polishing lanes', the .jf writer's): before a buffer they may use is freed
    // grow-only PINNED host buffers kept with the table (the polisher's segment tables, candidate lists and records travel through
    // them: a copy to or from pageable memory is staged by the runtime and makes the caller wait)
    static constexpr int PIN_PER_LANE = 8, PIN_SLOTS = PIN_PER_LANE * POLISH_LANES_MAX;
    WsBuf pin[PIN_SLOTS];
    void *pinned(int id, size_t bytes, std::string &err);

    static int min_log2_slots(int k);
    int init(int k, uint64_t min_slots, int device, std::string &err);
    void destroy();
    int read_stats(std::string &err);                 // stream sync + copy stats to h_stats
    int zero_slots(unsigned long long *slots, uint64_t n, std::string &err);
    // Lazy clear: a table that is known to be logically empty does not have to be zeroed in HBM if the next thing that
    // happens to it is a partitioned counting piece -- lds_insert_kernel then starts every region image from zeros
    // instead of reading it, and writes every region back (one write pass instead of write + read + write).
    // Everything else calls materialize() first.
    bool slots_dirty = false;      // slot memory holds garbage; the table is logically empty
    int materialize(std::string &err) { if (!slots_dirty) return 0; slots_dirty = false; return zero_slots(d.slots, nslots, err); }
    int clear(std::string &err);
    int ensure_capacity(uint64_t upcoming_kmers, std::string &err);
    int grow(int new_s, std::string &err);            // rehash into 2^new_s slots if that is more than now
    int resize(int new_s, std::string &err);
    int ensure_narrow(std::string &err);             // whole remainders in the tags (what shards and region-wise imports need)
    int fit(double max_load, std::string &err);
    int after_batch(std::string &err);                // spill / fatal / growth handling
    int count_device(const uint8_t *d_bases, uint64_t n, std::string &err, uint64_t first_new = 0);   // first_new: bases before it are context only (k-mers ending there were counted before)
    int count_host(const char *bases, uint64_t n, std::string &err);
    // FASTA/FASTQ files (plain or gzip, one concatenated stream) parsed on the GPU, host state machine as the fallback
    // (ingest_gpu.hip); reports how many text bytes each parser handled
    int count_files_gpu(const char *const *paths, int n_paths, uint64_t *gpu_bytes, uint64_t *host_bytes, std::string &err);
    uint64_t ingest_gpu_bytes = 0, ingest_host_bytes = 0;
    const int64_t *ingest_begin = nullptr, *ingest_end = nullptr;   // per-file byte ranges of the next count_files_gpu call (or null)
    // The read files as a FEED of base batches in HBM instead of counts in this table (ingest_gpu.hip, "feed"): a thread runs
    // count_files_gpu with `bases_sink` set, every batch it would have counted is handed to the caller of feed_next and the
    // thread waits until feed_release.  For counting paths that are driven from outside (dist.count_sharded).
    std::function<int(const uint8_t *d_bases, uint64_t n)> bases_sink;
    struct Feed;
    Feed *feed = nullptr;
    int feed_start(const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths, std::string &err);
    int feed_next(const void **d_bases, uint64_t *n, std::string &err);      // *n == 0: the stream has ended (the thread is gone)
    int feed_release(std::string &err);
    void feed_stop();
    char *h_ingest = nullptr;      // pinned text staging of count_files_gpu
    size_t ingest_chunk = 0;
    int histogram(uint64_t *out10002, std::string &err);
    int histogram_part(uint32_t part, uint32_t nparts, uint64_t *out10002, std::string &err);
    void part_span(uint32_t part, uint32_t nparts, uint64_t &first, uint64_t &span) const;
    int lookup_strings(const char *chars, const int64_t *offsets, uint64_t n, uint32_t *out, std::string &err);
    int export_entries(uint64_t *n_out, unsigned long long **d_entries_out, std::string &err);  // 3 words each
    int import_entries(const unsigned long long *d_entries, uint64_t n, std::string &err);
    // 16-byte entries for the multi-GPU exchange: { hash.lo, hash.hi | count << (B-64) }; only keys whose home slot
    // falls into partition `part` of `nparts` equal slot ranges; mode 0 = add counts, 1 = set counts
    int export_packed(void *d_dst, uint64_t cap, uint64_t *n_out, uint32_t part, uint32_t nparts, std::string &err);
    int import_packed(const void *d_src, uint64_t n, int mode, std::string &err);
    int import_packed_multi(const void *const *d_srcs, const uint64_t *counts, uint32_t n_src, std::string &err);   // add, all lists in one sweep
    int reserve(uint64_t min_slots, std::string &err);
    // Owner-sharded table (table.hip, "shards"): all entries of this table in ONE pass, grouped by owner_of(hash, nown)
    // -- segment o starts at d_dst + o * cap entries, counts_out[o] entries long (may exceed cap: nothing is written past
    // cap, the caller retries with a larger buffer); attach = lookups through this table read shard o's slot array.
    int export_owner(void *d_dst, uint64_t cap, uint32_t nown, int sort_r, uint64_t *counts_out, std::string &err);   // sort_r > 0: by file range, see table.hip
    // counting as a multi-GPU exchange of region lists (count_part.hip, "count_exchange")
    int xchg_plan(uint64_t piece_max, uint64_t records_max, uint32_t nown, uint64_t out[8], std::string &err);     // 0: available, 1: not for this table / size
    int xchg_scan(const uint8_t *d_bases, uint64_t n, uint64_t pos, uint64_t end, uint64_t piece_max, uint32_t nown, void *d_defer, uint64_t defer_cap, uint64_t *records,
                  std::string &err);
    int xchg_partition(uint64_t piece_max, uint64_t records_max, uint32_t nown, void *d_send, void *d_send_cnt, void *d_defer, uint64_t defer_cap, std::string &err);
    int xchg_dedupe(uint64_t piece_max, uint64_t records_max, uint32_t nown, void *d_send, void *d_send_cnt, uint32_t *max_fill, int *cbits_out, std::string &err);
    int xchg_insert(const void *d_recv, const void *d_recv_cnt, uint64_t piece_max, uint64_t records_max, uint32_t nown, uint32_t self, const void *d_defer_all,
                    uint64_t n_defer_all, int whole_input, uint32_t slice_cap, int cbits, std::string &err);
    int ipc_handle(void *out64, std::string &err);
    int attach_ipc(const void *handles64, uint32_t n, uint32_t self, std::string &err);
    int attach_tables(Table *const *peers, uint32_t n, uint32_t self, std::string &err);
    void detach_shards();
    // A slot array whose IPC handle has been given out may still be mapped by peers when this table outgrows it: it is not
    // freed then but retired, until the ranks have attached to the new one and said so (release_retired), or the table goes
    bool exported = false;
    std::vector<void *> retired;
    void release_retired();
    void *ipc_mapped[MAX_SHARDS] = {};   // peers' slot arrays opened with hipIpcOpenMemHandle (closed by detach_shards)
    // the table as a Jellyfish binary/sorted database (jfwrite.hip); cmdline goes into the header like jellyfish's own
    // r_bits = log2 of the file's `size` (-1: this table's slot count); what: 0 = header + records, 1 = records only (one
    // sorted piece of a file written by several GPUs), 2 = header only
    int write_jf(const char *path, const char *const *cmdline, int n_cmd, std::string &err, int r_bits = -1, int what = 0);
    int load_jf_records(const char *path, uint64_t data_offset, uint64_t n_records, int key_len_bits, int counter_len, std::string &err);
};

}  // namespace jk
