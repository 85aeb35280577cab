// table.hip -- kernels and host logic of the HBM k-mer count table (see table.hpp for what it replaces).
//
// Kernels (all integer / byte work, HBM-bound, no MFMA):
//   count_kernel     K1+K2  bases -> rolling canonical k-mers -> insert-or-increment
//   lookup_kernel    K4     strings -> padded canonical k-mer -> exact count (clamped to 2^32-1)
//   histo_kernel     K3     slots -> 10002-bin multiplicity histogram (LDS-privatised bins)
//   export / import  C1     table <-> list of (hash, count) for growth and for the multi-GPU merge
#include "table.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>
#include <chrono>
#include <vector>

namespace jk {

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                     \
            return -1;                                                                \
        }                                                                             \
    } while (0)

// --------------------------------------------------------------------------------------------------
// K1+K2: counting.  One block stages a tile of TILE bases (+64 bases of left halo) into LDS as 2-bit
// codes and an "invalid" bit per base; thread t then owns the 16 windows that END at tile bases
// 16t..16t+15 and rolls the forward and reverse-complement k-mers across them, exactly the recurrence of
// JF::include/jellyfish/mer_iterator.hpp:66-77 (valid code -> shift both mers, anything else -> reset).
// Global reads are 16 B per lane, fully coalesced; table traffic is one random 16-B slot per k-mer.
// --------------------------------------------------------------------------------------------------
constexpr int CT_THREADS = 256;
constexpr int CT_GROUP = 16;                       // bases per thread
constexpr int CT_TILE = CT_THREADS * CT_GROUP;     // 4096 bases per block iteration
constexpr int CT_HALO = 4;                         // 4 groups = 64 bases >= k-1
constexpr int CT_BATCH = 4;                        // first probes kept in flight per thread

__device__ __forceinline__ void pack16(const uint8_t *b16, uint32_t &codes, uint32_t &inv) {
    codes = 0;
    inv = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int c = code(b16[j]);
        codes = (codes << 2) | (uint32_t)(c & 3);
        inv = (inv << 1) | (uint32_t)(c < 0);
    }
}

// load group g (16 bases starting at absolute position pos, may be negative / beyond n) and pack it
__device__ __forceinline__ void stage_group(const uint8_t *__restrict__ bases, int64_t pos, uint64_t n, bool aligned,
                                            uint32_t &codes, uint32_t &inv) {
    uint8_t b[16];
    if (pos >= 0 && (uint64_t)pos + 16 <= n && aligned) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + pos);
        *reinterpret_cast<uint4 *>(b) = v;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int64_t p = pos + j;
            b[j] = (p >= 0 && (uint64_t)p < n) ? bases[p] : (uint8_t)'N';
        }
    }
    pack16(b, codes, inv);
}

// `emit_from`: only windows ENDING at piece position >= emit_from are counted (pieces overlap by k-1 bases
// plus alignment padding; the overlap belongs to the previous piece).
// MODE 0 = product; 1 = hashing only, 2 = hashing + home-slot load only (tuning experiments, tools/bench_resident.py)
template <int MODE>
__global__ __launch_bounds__(CT_THREADS) void count_kernel(const uint8_t *__restrict__ bases, uint64_t n, uint64_t ntiles,
                                                            uint64_t emit_from, TableDev T) {
    __shared__ uint32_t s_code[CT_THREADS + CT_HALO];
    __shared__ uint32_t s_inv[CT_THREADS + CT_HALO];
    const int t = threadIdx.x;
    const int k = T.k;
    const bool aligned = ((reinterpret_cast<uintptr_t>(bases) & 15) == 0);
    const u128 kmask = maskbits(2 * k);
    unsigned long long added = 0, fresh = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base0 = (int64_t)(tile * CT_TILE);
        uint32_t c, iv;
        stage_group(bases, base0 + (int64_t)t * CT_GROUP, n, aligned, c, iv);
        s_code[t + CT_HALO] = c;
        s_inv[t + CT_HALO] = iv;
        if (t < CT_HALO) {
            uint32_t hc, hiv;
            stage_group(bases, base0 - (int64_t)(CT_HALO - t) * CT_GROUP, n, aligned, hc, hiv);
            s_code[t] = hc;
            s_inv[t] = hiv;
        }
        __syncthreads();
        // the 64 bases before my group, oldest in the high bits
        const uint32_t w4 = s_code[t], w3 = s_code[t + 1], w2 = s_code[t + 2], w1 = s_code[t + 3];
        const uint64_t ivprev = ((uint64_t)s_inv[t] << 48) | ((uint64_t)s_inv[t + 1] << 32) | ((uint64_t)s_inv[t + 2] << 16) |
                                (uint64_t)s_inv[t + 3];
        const uint32_t own = c, owninv = iv;
        __syncthreads();  // LDS is free for the next tile from here on
        u128 fwd = band(mk(((uint64_t)w4 << 32) | w3, ((uint64_t)w2 << 32) | w1), kmask);
        u128 rc = revcomp(fwd, k);
        int run = ivprev ? (int)__builtin_ctzll(ivprev) : 64;  // valid bases in a row ending just before my group
        if (owninv == 0xFFFFu) continue;                       // nothing but separators / padding here
        // four windows at a time: hash all four, put their four home-slot loads in flight together, then resolve
        for (int jb = 0; jb < CT_GROUP; jb += CT_BATCH) {
            u128 h[CT_BATCH];
            bool val[CT_BATCH];
            unsigned long long cur0[CT_BATCH];
#pragma unroll
            for (int u = 0; u < CT_BATCH; ++u) {
                const int j = jb + u;
                const uint32_t cj = (own >> (30 - 2 * j)) & 3u;
                const bool bad = (owninv >> (15 - j)) & 1u;
                fwd = band(bor(shl(fwd, 2), mk(0, cj)), kmask);
                rc = bor(shr(rc, 2), shl(mk(0, 3u - cj), 2 * (k - 1)));
                run = bad ? 0 : run + 1;
                val[u] = run >= k && (uint64_t)(base0 + t * CT_GROUP + j) >= emit_from;
                h[u] = mix(lt(rc, fwd) ? rc : fwd, T.B);
            }
            if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < CT_BATCH; ++u) if (val[u]) { fresh ^= h[u].lo ^ h[u].hi; ++added; }
                continue;
            }
#pragma unroll
            for (int u = 0; u < CT_BATCH; ++u) {
                cur0[u] = 0;
                if (val[u]) cur0[u] = __hip_atomic_load(T.slots + 2 * (home_of(h[u], T.B, T.s) & T.mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < CT_BATCH; ++u) if (val[u]) { fresh ^= cur0[u]; ++added; }
                continue;
            }
#pragma unroll
            for (int u = 0; u < CT_BATCH; ++u) {
                if (val[u]) {
                    const int r = table_add_prefetched(T, h[u], 1ull, cur0[u]);
                    if (r == 2) ++fresh;
                    else if (r == 0) table_spill(T, h[u], 1ull);
                    ++added;
                }
            }
        }
    }
    // one pair of counter updates per wave
    for (int o = 32; o > 0; o >>= 1) { added += __shfl_xor(added, o); fresh += __shfl_xor(fresh, o); }
    if ((threadIdx.x & 63) == 0) {
        if (added) atomicAdd(&T.stats[ST_OCCURRENCES], added);
        if (fresh) atomicAdd(&T.stats[MODE == 0 ? ST_DISTINCT : 7], fresh);
    }
}

// --------------------------------------------------------------------------------------------------
// K4: lookups of arbitrary strings with the reference's truncate-and-pad semantics (qf[MerDNA(s).get_canonical()],
// JF::swig/mer_file.i:41 + JF::swig/mer_dna.i:15).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lookup_kernel(const char *__restrict__ chars, const int64_t *__restrict__ offs, uint64_t n,
                                                     uint32_t *__restrict__ out, TableDev T) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const char *s = chars + offs[i];
        const long len = (long)(offs[i + 1] - offs[i]);
        const u128 m = encode_padded(T.k, len, [&](int q) { return (unsigned char)s[q]; });
        out[i] = clamp32(table_get(T, mix(canonical(m, T.k), T.B)));
    }
}

// --------------------------------------------------------------------------------------------------
// K3: histogram (JF::sub_commands/histo_main.cc:34-44): bucket = min(count, 10001) on the clamped count.
// One coalesced 16-B read per slot; bins are privatised in LDS (40 KB) and flushed once per block.
// --------------------------------------------------------------------------------------------------
constexpr int HISTO_BINS = 10002;
__global__ __launch_bounds__(256) void histo_kernel(TableDev T, unsigned long long *__restrict__ out) {
    __shared__ unsigned int bins[HISTO_BINS];
    for (int i = threadIdx.x; i < HISTO_BINS; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    const uint64_t nslots = T.mask + 1;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
        if (e.x != 0ull && e.y != 0ull) {
            const uint32_t c = clamp32(e.y);
            atomicAdd(&bins[c > 10001u ? 10001u : c], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HISTO_BINS; i += blockDim.x)
        if (bins[i]) atomicAdd(&out[i], (unsigned long long)bins[i]);
}

// --------------------------------------------------------------------------------------------------
// C1 building blocks: table -> (hash.hi, hash.lo, count) entries and back.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void export_kernel(TableDev T, unsigned long long *__restrict__ entries,
                                                     unsigned long long *__restrict__ counter, uint64_t cap) {
    const uint64_t nslots = T.mask + 1;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
        if (e.x == 0ull) continue;
        const u128 h = slot_hash(T, i, e.x);
        const unsigned long long idx = atomicAdd(counter, 1ull);
        if (idx < cap) {
            entries[3 * idx + 0] = h.hi;
            entries[3 * idx + 1] = h.lo;
            entries[3 * idx + 2] = e.y;
        }
    }
}

__global__ __launch_bounds__(256) void import_kernel(const unsigned long long *__restrict__ entries, uint64_t n, TableDev T) {
    unsigned long long fresh = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const u128 h = mk(entries[3 * i + 0], entries[3 * i + 1]);
        const unsigned long long c = entries[3 * i + 2];
        if (c) fresh += table_add_or_spill(T, h, c);
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// ---- 16-byte packed entries (multi-GPU exchange) ------------------------------------------------------------
__device__ __forceinline__ int packed_count_shift(int B) { return B > 64 ? B - 64 : 0; }
// owner partition of a key: by the top 32 bits of its mixed hash, so it does not depend on the table size
// (ranks may hold tables of different sizes while they exchange)
__host__ __device__ __forceinline__ uint32_t part_of(u128 h, int B, uint32_t nparts) {
    const uint64_t top32 = B >= 32 ? shr(h, B - 32).lo : (h.lo << (32 - B));
    return (uint32_t)((top32 * (uint64_t)nparts) >> 32);
}

// scans slots [first, first+span) (mod table size): a key homed in partition `part` sits at most MAXPROBE-1 slots
// behind its home, so one partition costs 1/nparts of a table pass.  Two passes over that range, no global atomic: block b
// owns the contiguous slice [b*chunk, (b+1)*chunk) of it; pass 1 counts its matching entries, a one-block scan turns the
// counts into output offsets, pass 2 writes -- compact and in slot order.  (One shared cursor, even with one returning
// atomic per wave, serialises at ~80 M atomics/s on a single address: 100 ms for half of a 2^30-slot table.)
__device__ __forceinline__ bool packed_entry_of(const TableDev &T, uint64_t i, uint32_t part, uint32_t nparts, int sh, ulonglong2 &o) {
    const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
    if (e.x == 0ull) return false;
    const u128 h = slot_hash(T, i, e.x);
    if (nparts > 1 && part_of(h, T.B, nparts) != part) return false;
    if (sh && (e.y >> (64 - sh)) != 0ull) { atomicExch(&T.stats[ST_FATAL], 2ull); return false; }   // count does not fit the packing
    o = make_ulonglong2(h.lo, sh ? (h.hi | (e.y << sh)) : e.y);   // (B <= 64: the whole second word is the count)
    return true;
}
constexpr int EXP_BLOCKS = 2048;
constexpr int EXP_STRIDE = EXP_BLOCKS + 8;   // words per count array (EXP_BLOCKS block counts, then the total)
__global__ __launch_bounds__(256) void export_packed_count_kernel(TableDev T, uint32_t part, uint32_t nparts, uint64_t first, uint64_t span,
                                                                  uint64_t chunk, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int s_w[4];
    const int sh = packed_count_shift(T.B);
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < span ? lo + chunk : span;
    unsigned int n = 0;
    for (uint64_t q = lo + threadIdx.x; q < hi; q += blockDim.x) {
        ulonglong2 o;
        n += packed_entry_of(T, (first + q) & T.mask, part, nparts, sh, o) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (unsigned long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// exclusive scan of EXP_BLOCKS counts in place; counts[EXP_BLOCKS] = total
__global__ __launch_bounds__(1024) void export_packed_scan_kernel(unsigned long long *__restrict__ counts) {
    __shared__ unsigned long long s_w[16];
    const int t = threadIdx.x;
    counts += (size_t)blockIdx.x * EXP_STRIDE;   // one array per block (export_owner: one per owner)
    const unsigned long long a = counts[2 * t], b = counts[2 * t + 1];
    unsigned long long inc = a + b;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
    if ((t & 63) == 63) s_w[t >> 6] = inc;
    __syncthreads();
    unsigned long long wbase = 0;
    for (int w = 0; w < (t >> 6); ++w) wbase += s_w[w];
    const unsigned long long ex = wbase + inc - (a + b);
    counts[2 * t] = ex;
    counts[2 * t + 1] = ex + a;
    if (t == 1023) counts[EXP_BLOCKS] = wbase + inc;
}
__global__ __launch_bounds__(256) void export_packed_write_kernel(TableDev T, ulonglong2 *__restrict__ out, uint64_t cap, uint32_t part, uint32_t nparts,
                                                                  uint64_t first, uint64_t span, uint64_t chunk,
                                                                  const unsigned long long *__restrict__ bases) {
    __shared__ unsigned int s_w[4];
    const int sh = packed_count_shift(T.B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < span ? lo + chunk : span;
    unsigned long long cursor = bases[blockIdx.x];
    for (uint64_t q0 = lo; q0 < hi; q0 += blockDim.x) {          // block-uniform trip count
        const uint64_t q = q0 + threadIdx.x;
        ulonglong2 o = make_ulonglong2(0ull, 0ull);
        const bool have = q < hi && packed_entry_of(T, (first + q) & T.mask, part, nparts, sh, o);
        const uint64_t m = __ballot(have);
        if (lane == 0) s_w[wave] = (unsigned int)__popcll(m);
        __syncthreads();
        unsigned int before = 0, total = 0;
        for (int w = 0; w < 4; ++w) { const unsigned int c = s_w[w]; total += c; if (w < wave) before += c; }
        const unsigned long long idx = cursor + before + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if (have && idx < cap) out[idx] = o;
        cursor += total;
        __syncthreads();
    }
}

// ---- owner-sharded table: every entry of the table, grouped by owner_of(hash, nown), in ONE pass over the slots ----
// Same two atomic-free passes as above with one count array per owner: counts[o * EXP_STRIDE + block].
// sort_r > 0: group by the RANGE of the key in the order of a binary/sorted database with `size` 2^sort_r instead (that
// order is the numeric order of the key rotated right by sort_r bits, jfwrite.hip: range = its top bits = the key's low
// sort_r bits), so that the groups, each sorted, are consecutive pieces of the file
__device__ __forceinline__ bool owner_entry_of(const TableDev &T, uint64_t i, uint32_t nown, int sort_r, int sh, ulonglong2 &o, uint32_t &owner) {
    const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
    if (e.x == 0ull) return false;
    const u128 h = slot_hash(T, i, e.x);
    if (sh && (e.y >> (64 - sh)) != 0ull) { atomicExch(&T.stats[ST_FATAL], 2ull); return false; }
    if (sort_r > 0) {
        const u128 key = unmix(h, T.B);
        const uint64_t low = sort_r >= 64 ? key.lo : (key.lo & ((1ull << sort_r) - 1ull));
        owner = (uint32_t)(((low >> (sort_r > 32 ? sort_r - 32 : 0)) * (uint64_t)nown) >> (sort_r > 32 ? 32 : sort_r));
    } else
    owner = owner_of(h, nown);
    o = make_ulonglong2(h.lo, sh ? (h.hi | (e.y << sh)) : e.y);   // (B <= 64: the whole second word is the count)
    return true;
}
__global__ __launch_bounds__(256) void export_owner_count_kernel(TableDev T, uint32_t nown, int sort_r, uint64_t chunk, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int s_w[4][MAX_SHARDS];
    const int sh = packed_count_shift(T.B);
    const uint64_t span = T.mask + 1;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < span ? lo + chunk : span;
    unsigned int n[MAX_SHARDS];
#pragma unroll
    for (uint32_t w = 0; w < MAX_SHARDS; ++w) n[w] = 0;
    for (uint64_t q = lo + threadIdx.x; q < hi; q += blockDim.x) {
        ulonglong2 o;
        uint32_t owner = 0;
        if (owner_entry_of(T, q, nown, sort_r, sh, o, owner)) {
#pragma unroll
            for (uint32_t w = 0; w < MAX_SHARDS; ++w) n[w] += owner == w ? 1u : 0u;
        }
    }
#pragma unroll
    for (uint32_t w = 0; w < MAX_SHARDS; ++w) {
        unsigned int v = n[w];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < nown)
        counts[(size_t)threadIdx.x * EXP_STRIDE + blockIdx.x] =
            (unsigned long long)s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void export_owner_write_kernel(TableDev T, ulonglong2 *__restrict__ out, uint64_t cap, uint32_t nown, int sort_r, uint64_t chunk,
                                                                 const unsigned long long *__restrict__ bases) {
    __shared__ unsigned int s_w[4][MAX_SHARDS];
    const int sh = packed_count_shift(T.B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t span = T.mask + 1;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < span ? lo + chunk : span;
    unsigned long long cursor[MAX_SHARDS];
#pragma unroll
    for (uint32_t w = 0; w < MAX_SHARDS; ++w) cursor[w] = w < nown ? bases[(size_t)w * EXP_STRIDE + blockIdx.x] : 0ull;
    for (uint64_t q0 = lo; q0 < hi; q0 += blockDim.x) {          // block-uniform trip count
        const uint64_t q = q0 + threadIdx.x;
        ulonglong2 o = make_ulonglong2(0ull, 0ull);
        uint32_t owner = 0;
        const bool have = q < hi && owner_entry_of(T, q, nown, sort_r, sh, o, owner);
        uint64_t mine = 0;     // ballot of my owner's lanes in this wave
#pragma unroll
        for (uint32_t w = 0; w < MAX_SHARDS; ++w) {
            const uint64_t m = __ballot(have && owner == w);
            if (lane == 0) s_w[wave][w] = (unsigned int)__popcll(m);
            if (owner == w) mine = m;
        }
        __syncthreads();
        unsigned int before = 0;
        unsigned long long cur = 0;
#pragma unroll
        for (uint32_t w = 0; w < MAX_SHARDS; ++w) {
            unsigned int tot = 0, bef = 0;
            for (int v = 0; v < 4; ++v) { const unsigned int c = s_w[v][w]; tot += c; if (v < wave) bef += c; }
            if (owner == w) { before = bef; cur = cursor[w]; }
            cursor[w] += tot;
        }
        const unsigned long long idx = cur + before + (unsigned long long)__popcll(mine & ((1ull << lane) - 1ull));
        if (have && idx < cap) out[(size_t)owner * cap + idx] = o;
        __syncthreads();
    }
}

// insert-or-assign: the key's count becomes `val` (used when an owner's final counts replace a rank's partial ones)
__device__ __forceinline__ int table_set(const TableDev &T, u128 h, unsigned long long val) {
    if (T.ext) return table_put_wide(T, h, val, true);
    const uint64_t home = home_of(h, T.B, T.s);
    const uint64_t rem = rem_of(h, T.B, T.s);
    for (uint32_t off = 0; off < MAXPROBE; ++off) {
        const uint64_t slot = (home + off) & T.mask;
        const unsigned long long want = tag_of(rem, off);
        unsigned long long *p = T.slots + 2 * slot;
        unsigned long long cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int fresh = 1;
        if (cur == 0ull) {
            cur = atomicCAS(p, 0ull, want);
            if (cur == 0ull) { fresh = 2; cur = want; }
        }
        if (cur == want) {
            __hip_atomic_store(p + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return fresh;
        }
    }
    return 0;
}

__global__ __launch_bounds__(256) void import_packed_kernel(const ulonglong2 *__restrict__ in, uint64_t n, TableDev T, int mode) {
    unsigned long long fresh = 0;
    const int sh = packed_count_shift(T.B);
    const unsigned long long himask = sh ? ((1ull << sh) - 1ull) : 0ull;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 e = in[i];
        const u128 h = mk(e.y & himask, e.x);
        const unsigned long long c = sh ? (e.y >> sh) : e.y;
        // (B <= 64 keeps the count in the whole second word)
        if (!c) continue;
        if (mode == 0) fresh += table_add_or_spill(T, h, c);
        else {
            const int r = table_set(T, h, c);
            if (r == 2) ++fresh;
            else if (r == 0) table_spill(T, h, c);   // re-inserted additively after growth: only ever reached for NEW keys
        }
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// The same for up to MAX_SHARDS entry lists at once (an owner adding what every rank sent it).  Each list is in slot order
// of a table with the same hash, so position c/nchunks of every list falls into the same part of THIS table: block group c
// takes chunk c of every list, and because blocks are dispatched in index order the slots being updated at any time are a
// narrow band that stays in L2 / Infinity Cache -- one sweep over the table instead of one per list.
struct MultiSrc { const ulonglong2 *p[MAX_SHARDS]; unsigned long long n[MAX_SHARDS]; uint32_t n_src; };
constexpr uint32_t IMP_BPC = 64;      // blocks per chunk
__global__ __launch_bounds__(256) void import_packed_multi_kernel(MultiSrc S, uint32_t nchunks, TableDev T) {
    unsigned long long fresh = 0;
    const int sh = packed_count_shift(T.B);
    const unsigned long long himask = sh ? ((1ull << sh) - 1ull) : 0ull;
    const uint32_t c = blockIdx.x / IMP_BPC, j = blockIdx.x % IMP_BPC;
#pragma unroll 1
    for (uint32_t s = 0; s < S.n_src; ++s) {
        const ulonglong2 *__restrict__ in = S.p[s];
        // (128-bit products: n * c does not fit 64 bits for n ~ 2^32 and c ~ 2^8 only far beyond any real list; plain math is fine)
        const unsigned long long lo = S.n[s] / nchunks * c + (S.n[s] % nchunks) * c / nchunks;
        const unsigned long long hi = c + 1 == nchunks ? S.n[s] : S.n[s] / nchunks * (c + 1) + (S.n[s] % nchunks) * (c + 1) / nchunks;
        for (unsigned long long i = lo + (unsigned long long)j * 256 + threadIdx.x; i < hi; i += (unsigned long long)IMP_BPC * 256) {
            const ulonglong2 e = in[i];
            const u128 h = mk(e.y & himask, e.x);
            const unsigned long long cnt = sh ? (e.y >> sh) : e.y;
            if (cnt) fresh += table_add_or_spill(T, h, cnt);
        }
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// ---- the same into an EMPTY table, without a global atomic: region images in LDS --------------------------------------
// Global atomics top out at ~17 G entries/s (two or three L2 transactions per entry).  When the table is empty there is
// nothing to merge with in HBM, so the table can be built the way lds_insert_kernel builds it: one block owns a region of
// IMPR slots, collects the region's entries in an LDS image and writes the image out once.  Which entries are a region's?
// The lists are in slot order of same-hash tables, i.e. nearly sorted by destination slot: a histogram of the entries'
// regions (pass 1) and its prefix sums cut every list into consecutive pieces, one per region, that hold exactly as many
// entries as the region gets and -- up to the few inversions next to a boundary -- exactly its entries.  An entry that is
// not the block's (an inversion), or whose probe sequence leaves the region, goes to a deferred list and is added with
// the global kernel afterwards.  Nothing depends on the lists really being sorted: unsorted input just defers everything.
constexpr uint32_t IMPR_LOG = 12, IMPR = 1u << IMPR_LOG;                 // 4096 slots = 64 KB of LDS
__global__ __launch_bounds__(256) void imp_region_hist_kernel(MultiSrc S, TableDev T, uint32_t nreg, unsigned int *__restrict__ hist) {
    const int sh = packed_count_shift(T.B);
    const unsigned long long himask = sh ? ((1ull << sh) - 1ull) : 0ull;
    for (uint32_t s = 0; s < S.n_src; ++s) {
        const ulonglong2 *__restrict__ in = S.p[s];
        unsigned int *__restrict__ hs = hist + (size_t)s * (nreg + 1);
        const unsigned long long n = S.n[s], step = (unsigned long long)gridDim.x * blockDim.x;
        for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x; i0 < n; i0 += step) {
            const unsigned long long i = i0 + threadIdx.x;
            uint32_t reg = 0xFFFFFFFFu;
            if (i < n) {
                const ulonglong2 e = in[i];
                reg = (uint32_t)(home_of(mk(e.y & himask, e.x), T.B, T.s) >> IMPR_LOG);
            }
            // neighbours in the list fall into the same one or two regions: one atomic per distinct region of the wave
            uint64_t todo = __ballot(reg != 0xFFFFFFFFu);
            while (todo) {
                const int leader = (int)__builtin_ctzll(todo);
                const uint32_t r = __shfl(reg, leader);
                const uint64_t same = __ballot(reg == r);
                if ((threadIdx.x & 63) == leader) atomicAdd(&hs[r], (unsigned int)__popcll(same));
                todo &= ~same;
            }
        }
    }
}
// in-place exclusive scan of each list's nreg counts (block s = list s); entry nreg = the total
__global__ __launch_bounds__(1024) void imp_region_scan_kernel(unsigned int *__restrict__ hist, uint32_t nreg) {
    __shared__ unsigned int s_w[16];
    __shared__ unsigned int s_carry;
    unsigned int *h = hist + (size_t)blockIdx.x * (nreg + 1);
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nreg; base += 1024) {
        const unsigned int v = base + t < nreg ? h[base + t] : 0u;
        unsigned int inc = v;
        for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
        if ((t & 63) == 63) s_w[t >> 6] = inc;
        __syncthreads();
        unsigned int wbase = s_carry;
        for (int w = 0; w < (t >> 6); ++w) wbase += s_w[w];
        if (base + t < nreg) h[base + t] = wbase + inc - v;
        __syncthreads();
        if (t == 1023) s_carry = wbase + inc;
        __syncthreads();
    }
    if (t == 0) h[nreg] = s_carry;
}
__global__ __launch_bounds__(256) void lds_import_kernel(MultiSrc S, TableDev T, uint32_t nreg, const unsigned int *__restrict__ bounds,
                                                         ulonglong2 *__restrict__ deferred, unsigned long long *__restrict__ deferred_n, unsigned long long deferred_cap) {
    __shared__ unsigned long long s_tag[IMPR];
    __shared__ unsigned long long s_cnt[IMPR];
    __shared__ unsigned int s_fresh[4];
    const uint32_t r = blockIdx.x;
    const int sh = packed_count_shift(T.B);
    const unsigned long long himask = sh ? ((1ull << sh) - 1ull) : 0ull;
    for (uint32_t j = threadIdx.x; j < IMPR; j += 256) { s_tag[j] = 0ull; s_cnt[j] = 0ull; }
    __syncthreads();
    unsigned int fresh = 0;
    for (uint32_t s = 0; s < S.n_src; ++s) {
        const ulonglong2 *__restrict__ in = S.p[s];
        const unsigned int *__restrict__ b = bounds + (size_t)s * (nreg + 1);
        const unsigned int lo = b[r], hi = b[r + 1];
        for (unsigned int i = lo + threadIdx.x; i < hi; i += 256) {
            const ulonglong2 e = in[i];
            const u128 h = mk(e.y & himask, e.x);
            const unsigned long long c = sh ? (e.y >> sh) : e.y;
            if (!c) continue;
            const uint64_t home = home_of(h, T.B, T.s);
            bool placed = false;
            if ((uint32_t)(home >> IMPR_LOG) == r) {
                const uint64_t rem = rem_of(h, T.B, T.s);
                const uint32_t local = (uint32_t)(home & (IMPR - 1));
                for (uint32_t off = 0; off < MAXPROBE && local + off < IMPR; ++off) {
                    const unsigned long long want = tag_of(rem, off);
                    unsigned long long cur = s_tag[local + off];
                    if (cur == 0ull) {
                        cur = atomicCAS(&s_tag[local + off], 0ull, want);
                        if (cur == 0ull) { ++fresh; cur = want; }
                    }
                    if (cur == want) { atomicAdd(&s_cnt[local + off], c); placed = true; break; }
                }
            }
            if (!placed) {        // an inversion of the list order, or a probe sequence that leaves the region
                const unsigned long long di = atomicAdd(deferred_n, 1ull);
                if (di < deferred_cap) deferred[di] = e;
            }
        }
    }
    __syncthreads();
    ulonglong2 *out = reinterpret_cast<ulonglong2 *>(T.slots) + (size_t)r * IMPR;
    for (uint32_t j = threadIdx.x; j < IMPR; j += 256) out[j] = make_ulonglong2(s_tag[j], s_cnt[j]);
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0) s_fresh[threadIdx.x >> 6] = fresh;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int f = s_fresh[0] + s_fresh[1] + s_fresh[2] + s_fresh[3];
        if (f) atomicAdd(&T.stats[ST_DISTINCT], (unsigned long long)f);
    }
}

// hipMemsetAsync runs at ~1 TB/s on this stack; 16-B streaming stores reach the HBM write rate
__global__ __launch_bounds__(256) void zero_kernel(ulonglong2 *__restrict__ p, uint64_t n16) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = make_ulonglong2(0ull, 0ull);
}

// rehash straight from an old slot array into a (larger) table
__global__ __launch_bounds__(256) void rehash_kernel(TableDev oldT, TableDev newT) {
    unsigned long long fresh = 0;
    const uint64_t nslots = oldT.mask + 1;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(oldT.slots + 2 * i);
        if (e.x == 0ull) continue;
        fresh += table_add_or_spill(newT, slot_hash(oldT, i, e.x), e.y);
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&newT.stats[ST_DISTINCT], fresh);
}

// ==================================================================================================
// host side
// ==================================================================================================
// bytes to allocate for a slot array.  An allocation of exactly 2 GiB (2^27 slots) cannot be mapped by another process on
// this ROCm stack: hipIpcOpenMemHandle never returns (seen with 2 and with 4 processes; 1 GiB, 4 GiB and 8 GiB are fine) --
// bit 31 of the size is the suspect, so such an array is allocated as 4 GiB.
static size_t slot_alloc_bytes(uint64_t nslots) {
    size_t b = (size_t)nslots * 16;
    if ((b & (1ull << 31)) && !getenv("JASPER_NO_IPC_PAD")) b += 1ull << 31;     // (the switch exists for the test of the probe below)
    return b;
}

static int grid_for(uint64_t work_items, int per_block) {
    uint64_t b = (work_items + per_block - 1) / per_block;
    const uint64_t cap = 256ull * 8;  // 256 CUs x 8 blocks: fills the chip, rest is grid-stride
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int Table::min_log2_slots(int k) {
    // the tag holds 63 - OFFBITS remainder bits, the ext word of a wide table (kmer.hpp: wide_rem) 64 more: B - s <= 117
    int need = 2 * k - (63 - OFFBITS) - 64;
    return std::max(need, 10);
}

int Table::init(int k_, uint64_t min_slots, int device_, std::string &err) {
    if (k_ < 1 || k_ > 64) { err = "k must be in [1,64]"; return -1; }
    k = k_;
    device = device_;
    HIPCHK(hipSetDevice(device));
    if (const char *e = getenv("JASPER_EXPERIMENT_CU_MASK")) {
        // experiments only (docs/experiments.md, round 5): the table's stream restricted to a part of the chip -- "alt2" every second
        // CU bit (128 of 256), "alt4x3" three of every four (192), "lo128" / "lo192" the first 128 / 192 bits
        uint32_t m[8];
        for (int w = 0; w < 8; ++w)
            m[w] = !strcmp(e, "alt2") ? 0x55555555u : !strcmp(e, "alt4x3") ? 0x77777777u : !strcmp(e, "lo128") ? (w < 4 ? ~0u : 0u) : !strcmp(e, "lo192") ? (w < 6 ? ~0u : 0u) : ~0u;
        HIPCHK(hipExtStreamCreateWithCUMask(&stream, 8, m));
    } else
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    int s = min_log2_slots(k);
    size_hint = min_slots;
    while ((1ull << s) < min_slots) ++s;
    if (s > 2 * k) s = 2 * k;                 // never more slots than possible keys ...
    if (s < min_log2_slots(k)) {              // ... unless the tag format needs them
        err = "internal: slot count below tag-format minimum"; return -1;
    }
    nslots = 1ull << s;
    d = TableDev{};
    d.s = s; d.B = 2 * k; d.k = k; d.mask = nslots - 1;
    d.spill_cap = 1u << 16;
    HIPCHK(hipMalloc((void **)&d.slots, slot_alloc_bytes(nslots)));
    if (wide_rem(d.B, d.s)) HIPCHK(hipMalloc((void **)&d.ext, (size_t)nslots * 8));      // (never read before the slot's count is non-zero: not zeroed)
    HIPCHK(hipMalloc((void **)&d.stats, ST_WORDS * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&d.spill, d.spill_cap * 3 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&d_histo, 2 * HISTO_WORDS * sizeof(unsigned long long)));
    histo_cached = histo_request = false;
    slots_dirty = true;   // zeroed on first use (or never, if that use is a partitioned counting piece)
    HIPCHK(hipMemsetAsync(d.stats, 0, ST_WORDS * sizeof(unsigned long long), stream));
    HIPCHK(hipHostMalloc((void **)&h_stats, ST_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    HIPCHK(hipEventCreate(&ev_k0));
    HIPCHK(hipEventCreate(&ev_k1));
    memset(h_stats, 0, ST_WORDS * sizeof(unsigned long long));
    HIPCHK(jk_stream_wait(stream));
    return 0;
}

std::atomic<int> g_cancel{0};

void Table::wait_streams() {
    if (stream) (void)jk_stream_wait(stream);
    for (hipStream_t ps : polish_stream) if (ps) (void)jk_stream_wait(ps);
    if (jf_stream) (void)jk_stream_wait(jf_stream);
    if (ingest_stream) (void)jk_stream_wait(ingest_stream);
    if (ingest_copy_stream) (void)jk_stream_wait(ingest_copy_stream);
}

void *Table::workspace(int id, size_t bytes, std::string &err) {
    if (bytes == 0) bytes = 256;
    WsBuf &b = ws[id];
    if (b.bytes >= bytes) return b.p;
    if (b.p) { wait_streams(); (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }      // (a lane's kernels run on its own stream)
    const size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)256 << 20);   // a little headroom so that slightly larger batches do not reallocate
    const char *dbg_e = getenv("JASPER_COUNT_DEBUG");
    const bool dbg = dbg_e != nullptr && (want >= (256u << 20) || atoi(dbg_e) >= 2);      // (2: every allocation)
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&b.p, want);
    if (dbg) fprintf(stderr, "[workspace] slot %d: hipMalloc of %.3f GB took %.1f ms\n", id, (double)want / 1e9,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) {
        e = hipMalloc(&b.p, bytes);
        if (e != hipSuccess) { err = std::string("device workspace allocation failed: ") + hipGetErrorString(e); b.p = nullptr; return nullptr; }
        b.bytes = bytes;
        return b.p;
    }
    b.bytes = want;
    // debugging aid (round 5: a result that depended on what a fresh workspace happened to hold): every new workspace buffer filled with
    // one byte value, so that a read of memory nobody wrote gives the same wrong answer every time -- and another one for another value
    if (const char *pz = getenv("JASPER_DEBUG_POISON")) {
        const char *only = getenv("JASPER_DEBUG_POISON_SLOT");
        if (!only || atoi(only) == id) { (void)hipMemset(b.p, atoi(pz) & 0xFF, want); (void)hipDeviceSynchronize(); }
    }
    return b.p;
}

void *Table::pinned(int id, size_t bytes, std::string &err) {
    if (bytes == 0) bytes = 256;
    WsBuf &b = pin[id];
    if (b.bytes >= bytes) return b.p;
    if (b.p) { wait_streams(); (void)hipHostFree(b.p); b.p = nullptr; b.bytes = 0; }
    const size_t want = bytes + bytes / 4;
    const char *dbg_e = getenv("JASPER_COUNT_DEBUG");
    const auto t0 = std::chrono::steady_clock::now();
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { err = "pinned host buffer allocation failed"; b.p = nullptr; return nullptr; }
    if (dbg_e && atoi(dbg_e) >= 2) fprintf(stderr, "[pinned] slot %d: hipHostMalloc of %.1f MB took %.1f ms\n", id, (double)want / 1e6,
                                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    b.bytes = want;
    return b.p;
}

void Table::destroy() {
    feed_stop();
    (void)hipSetDevice(device);
    if (stream) (void)jk_stream_wait(stream);
    detach_shards();
    for (hipStream_t &ps : polish_stream) { if (ps) { (void)jk_stream_wait(ps); (void)hipStreamDestroy(ps); ps = nullptr; } }
    if (polish_ev) { (void)hipEventDestroy(polish_ev); polish_ev = nullptr; }
    if (jf_stream) { (void)jk_stream_wait(jf_stream); (void)hipStreamDestroy(jf_stream); jf_stream = nullptr; }
    if (ingest_stream) { (void)jk_stream_wait(ingest_stream); (void)hipStreamDestroy(ingest_stream); ingest_stream = nullptr; }
    if (ingest_copy_stream) { (void)jk_stream_wait(ingest_copy_stream); (void)hipStreamDestroy(ingest_copy_stream); ingest_copy_stream = nullptr; }
    for (hipEvent_t &e : ingest_copy_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (WsBuf &b : ws) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    for (WsBuf &b : pin) { if (b.p) (void)hipHostFree(b.p); b.p = nullptr; b.bytes = 0; }
    for (int i = 0; i < 2; ++i) {
        if (d_stage[i]) (void)hipFree(d_stage[i]);
        if (h_stage[i]) (void)hipHostFree(h_stage[i]);
        if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]);
    }
    release_retired();
    if (d.slots) (void)hipFree(d.slots);
    if (d.ext) (void)hipFree(d.ext);
    if (d.stats) (void)hipFree(d.stats);
    if (d.spill) (void)hipFree(d.spill);
    if (d_histo) (void)hipFree(d_histo);
    d_histo = nullptr;
    if (h_ingest) (void)hipHostFree(h_ingest);
    h_ingest = nullptr;
    if (h_stats) (void)hipHostFree(h_stats);
    if (ev_k0) (void)hipEventDestroy(ev_k0);
    if (ev_k1) (void)hipEventDestroy(ev_k1);
    if (stream) (void)hipStreamDestroy(stream);
    d = TableDev{};
    stream = nullptr;
}

int Table::zero_slots(unsigned long long *slots, uint64_t n, std::string &err) {
    hipLaunchKernelGGL(zero_kernel, dim3(256 * 16), dim3(256), 0, stream, reinterpret_cast<ulonglong2 *>(slots), n);
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void zero_stats_kernel(unsigned long long *stats) { if (threadIdx.x < ST_WORDS) stats[threadIdx.x] = 0ull; }

int Table::clear(std::string &err) {
    HIPCHK(hipSetDevice(device));
    slots_dirty = true;
    histo_cached = false;
    timespec a, b, c;
    const bool dbg = getenv("JASPER_COUNT_DEBUG") != nullptr;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &a);
    hipLaunchKernelGGL(zero_stats_kernel, dim3(1), dim3(64), 0, stream, d.stats);
    HIPCHK(hipGetLastError());
    if (dbg) {
        clock_gettime(CLOCK_MONOTONIC, &b);
        (void)jk_stream_wait(stream);
        clock_gettime(CLOCK_MONOTONIC, &c);
        fprintf(stderr, "[count] clear: launch %.3f ms, wait %.3f ms\n", (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6,
                (c.tv_sec - b.tv_sec) * 1e3 + (c.tv_nsec - b.tv_nsec) * 1e-6);
    }
    return 0;
}

int Table::read_stats(std::string &err) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpyAsync(h_stats, d.stats, ST_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    return 0;
}

int Table::grow(int new_s, std::string &err) {
    if (new_s > d.B) new_s = d.B;
    if (new_s <= d.s) return 0;
    return resize(new_s, err);
}

// smallest table that holds the present keys at a load of at most max_load (never below the tag format's minimum): an
// owner's shard is sized from a guess before its keys arrive and cut to size once, after the first exchange
int Table::fit(double max_load, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (read_stats(err)) return -1;
    if (!(max_load > 0.05 && max_load <= 0.9)) { err = "fit: load must be in (0.05, 0.9]"; return -1; }
    int ns = min_log2_slots(k);
    if (!d.ext) ns = std::max(ns, std::min(d.s, d.B - (63 - OFFBITS)));      // a table with whole remainders in its tags keeps them (shards)
    while ((double)h_stats[ST_DISTINCT] > max_load * (double)(1ull << ns) && ns < d.B) ++ns;
    if (ns == d.s) return 0;
    histo_cached = false;
    if (resize(ns, err)) return -1;
    return after_batch(err);      // (a rehash into fewer slots may spill; that path grows again)
}

// rehash into 2^new_s slots (more or fewer than now)
int Table::resize(int new_s, std::string &err) {
    if (new_s == d.s) return 0;
    detach_shards();      // the slot array moves and its geometry changes: the owners have to agree and attach again
    // what this call allocates is freed again on every error path (the old arrays stay the table's until the very end)
    unsigned long long *new_ext = nullptr, *ns = nullptr;      // the new geometry may or may not need the second remainder word
    auto fail = [&](hipError_t e, const char *what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        if (new_ext) (void)hipFree(new_ext);
        if (ns) (void)hipFree(ns);
        return -1;
    };
    hipError_t e;
    if (wide_rem(d.B, new_s) && (e = hipMalloc((void **)&new_ext, (size_t)(1ull << new_s) * 8)) != hipSuccess) return fail(e, "resize: hipMalloc(ext)");
    if ((e = hipMalloc((void **)&ns, slot_alloc_bytes(1ull << new_s))) != hipSuccess) return fail(e, "resize: hipMalloc(slots)");
    // an exported slot array may still be mapped by peers: not freed but retired until they have said so (release_retired)
    auto drop_old = [&]() {
        if (exported) { retired.push_back(d.slots); exported = false; }
        else (void)hipFree(d.slots);
        if (d.ext) (void)hipFree(d.ext);
    };
    if (slots_dirty) {   // logically empty: nothing to rehash, the new slot array stays lazily cleared as well
        if ((e = jk_stream_wait(stream)) != hipSuccess) return fail(e, "resize: stream");
        drop_old();
        d.slots = ns; d.ext = new_ext; d.s = new_s; d.mask = (1ull << new_s) - 1; nslots = 1ull << new_s;
        return 0;
    }
    TableDev nt = d;
    nt.s = new_s;
    nt.mask = (1ull << new_s) - 1;
    nt.ext = new_ext;
    nt.slots = ns;
    if (zero_slots(nt.slots, 1ull << new_s, err)) { (void)fail(hipSuccess, "resize"); return -1; }
    // distinct is recounted by the re-insertion
    if ((e = hipMemsetAsync(d.stats + ST_DISTINCT, 0, sizeof(unsigned long long), stream)) != hipSuccess) return fail(e, "resize: memset");
    hipLaunchKernelGGL(rehash_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, stream, d, nt);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "resize: rehash launch");
    if ((e = jk_stream_wait(stream)) != hipSuccess) return fail(e, "resize: rehash");
    drop_old();
    d = nt;
    nslots = 1ull << new_s;
    return 0;
}

// after a batch of insertions: re-insert spilled k-mers into a bigger table, grow when past the load limit
int Table::after_batch(std::string &err) {
    for (int round = 0; round < 8; ++round) {
        if (read_stats(err)) return -1;
        if (h_stats[ST_FATAL]) { err = "k-mer table overflow (spill buffer exhausted): pass a larger size hint"; return -2; }
        const uint64_t spilled = h_stats[ST_SPILL];
        if (getenv("JASPER_COUNT_DEBUG") && spilled) fprintf(stderr, "[count] after_batch: %llu spilled insertions\n", (unsigned long long)spilled);
        const bool too_full = (double)h_stats[ST_DISTINCT] > grow_at * (double)nslots && d.s < d.B;
        if (!spilled && !too_full) return 0;
        std::vector<unsigned long long> sp;
        if (spilled) {
            sp.resize(3 * spilled);
            HIPCHK(hipMemcpy(sp.data(), d.spill, sp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            HIPCHK(hipMemsetAsync(d.stats + ST_SPILL, 0, sizeof(unsigned long long), stream));
        }
        int ns = d.s + 1;
        while ((double)h_stats[ST_DISTINCT] > 0.5 * grow_at * (double)(1ull << ns) && ns < d.B) ++ns;
        if (d.s >= d.B && spilled) { err = "k-mer table cannot grow further"; return -2; }
        if (grow(ns, err)) return -1;
        if (spilled) {
            unsigned long long *tmp = nullptr;
            HIPCHK(hipMalloc((void **)&tmp, sp.size() * sizeof(unsigned long long)));
            HIPCHK(hipMemcpy(tmp, sp.data(), sp.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(import_kernel, dim3(grid_for(spilled, 256)), dim3(256), 0, stream, tmp, spilled, d);
            HIPCHK(hipGetLastError());
            HIPCHK(jk_stream_wait(stream));
            HIPCHK(hipFree(tmp));
        }
    }
    err = "k-mer table did not settle after growth";
    return -2;
}

int Table::ensure_capacity(uint64_t upcoming_kmers, std::string &err) {
    // keep the worst case (every upcoming k-mer new) under 3/4 full so probe runs stay far below MAXPROBE
    if (read_stats(err)) return -1;
    int ns = d.s;
    while ((double)(h_stats[ST_DISTINCT] + upcoming_kmers) > 0.75 * (double)(1ull << ns) && ns < d.B) ++ns;
    if (ns != d.s) return grow(ns, err);
    return 0;
}

int Table::launch_count(const uint8_t *d_piece, uint64_t len, uint64_t emit_from, std::string &err) {
    alignas(16) char geom[64];
    // the partitioned path streams the whole table once per piece: worth it only for pieces that are large relative
    // to the table (host-staged 64 MiB pieces stay on the direct kernel and are PCIe-bound anyway)
    // break-even measured on MI355X: direct ~18 Gk-mers/s; partitioned ~55 Gk-mers/s for the two list passes plus one
    // streaming pass over the table (32 B/slot, 16 B/slot when the table is still lazily cleared)
    if (!part_off && len >= (slots_dirty ? nslots / 6 : nslots / 4)) {
        if (partition_geometry(len, geom)) {
            ++count_partitioned_launches;
            return launch_count_partitioned(d_piece, len, emit_from, geom, err);
        }
    }
    histo_request = false;
    count_path = 0;
    if (materialize(err)) return -1;
    const uint64_t ntiles = (len + CT_TILE - 1) / CT_TILE;
    HIPCHK(hipEventRecord(ev_k0, stream));
    static const int mode = getenv("JASPER_EXPERIMENT_MODE") ? atoi(getenv("JASPER_EXPERIMENT_MODE")) : 0;
    if (mode == 1) hipLaunchKernelGGL(count_kernel<1>, dim3(grid_for(ntiles, 1)), dim3(CT_THREADS), 0, stream, d_piece, len, ntiles, emit_from, d);
    else if (mode == 2) hipLaunchKernelGGL(count_kernel<2>, dim3(grid_for(ntiles, 1)), dim3(CT_THREADS), 0, stream, d_piece, len, ntiles, emit_from, d);
    else hipLaunchKernelGGL(count_kernel<0>, dim3(grid_for(ntiles, 1)), dim3(CT_THREADS), 0, stream, d_piece, len, ntiles, emit_from, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_k1, stream));
    return 0;
}

// bases already in HBM. Processed in launches of at most nslots/4 bases so that the load factor is re-checked
// (and the table grown) between launches. A piece starts k-1 bases early, rounded down to 16 bytes so the
// kernel keeps its 16-byte vector loads; emit_from keeps every window counted exactly once.
int Table::count_device(const uint8_t *d_bases, uint64_t n, std::string &err, uint64_t first_new) {
    HIPCHK(hipSetDevice(device));
    if (read_stats(err)) return -1;
    histo_cached = false;
    const uint64_t halo = (uint64_t)(k - 1);
    const uint64_t misalign = reinterpret_cast<uintptr_t>(d_bases) & 15;
    uint64_t pos = std::min(first_new, n);
    // Before anything has been measured, the caller's size hint plays the role of `jellyfish count -s`: the expected
    // number of distinct k-mers of this input.  hint / bases is then the expected share of new keys per k-mer; it is
    // only trusted for sizing the pieces (x1.5 below) -- an input that is less repetitive than promised still ends up
    // in the spill / deferred lists and a grown table, or in a clean error, never in wrong counts.
    bool have_ratio = false;
    if (size_hint > 0 && n > (1u << 24) && h_stats[ST_DISTINCT] == 0 && h_stats[ST_OCCURRENCES] == 0) {
        dup_ratio = std::min(1.0, (double)size_hint / (double)n);
        have_ratio = dup_ratio < 0.6;
    }
    const bool started_empty = h_stats[ST_DISTINCT] == 0 && h_stats[ST_OCCURRENCES] == 0;
    size_t mem_have = 0;
    {   // partition lists need ~20 bytes of workspace per base
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            mem_have = free_b;
            for (int w = WS_COUNT; w < WS_COUNT + 4; ++w) mem_have += ws[w].bytes;
        }
    }
    const bool dbg = getenv("JASPER_COUNT_DEBUG") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_call = now_ms();
    if (dbg) fprintf(stderr, "[count] call start: %llu bases\n", (unsigned long long)n);
    while (pos < n) {
        // a launch may add at most as many new keys as keep the table under 3/4 full in the worst case (every
        // base a new k-mer); h_stats holds the distinct count of the last check
        const uint64_t room = (uint64_t)(0.75 * (double)nslots) > h_stats[ST_DISTINCT] ? (uint64_t)(0.75 * (double)nslots) - h_stats[ST_DISTINCT] : 0;
        uint64_t piece = std::max<uint64_t>(room, 1u << 20);
        // after the first piece the measured share of NEW keys per k-mer (x1.5 safety) replaces the worst case "every base
        // a new key"; the spill list / deferred list / growth still catch a piece that turns out less repetitive
        if ((pos > 0 || have_ratio) && dup_ratio < 0.6) piece = std::max<uint64_t>(piece, (uint64_t)((double)room / std::max(0.05, 1.5 * dup_ratio)));
        if (mem_have) piece = std::min<uint64_t>(piece, std::max<uint64_t>(mem_have / 28, 1u << 20));
        piece = std::min<uint64_t>(piece, 1ull << 33);   // (positions are 64-bit throughout; slice counters are 32-bit but per slice)
        // do not leave a small tail for a separate launch (the 1.5x safety factor covers a quarter more)
        if ((pos > 0 || have_ratio) && dup_ratio < 0.6 && n - pos <= piece + piece / 4 && n - pos <= (1ull << 33)) piece = n - pos;
        // the caller's own promise (size hint = expected distinct k-mers of this input) fits the free room: one piece
        if (have_ratio && pos == 0 && (double)n * dup_ratio <= (double)room && n <= (1ull << 33) && (!mem_have || n <= mem_have / 28)) piece = n;
        if (const char *e = getenv("JASPER_EXPERIMENT_PIECE")) piece = strtoull(e, nullptr, 10);   // tuning experiments only
        {   // make room up front for the new keys this piece is expected to bring (worst case for the first piece)
            const uint64_t todo = std::min<uint64_t>(piece, n - pos);
            // (the share of new keys only falls as coverage accumulates, so the last piece's ratio is already an upper
            // estimate; spill list, deferred list and growth after the piece remain the safety net)
            const uint64_t expect = (pos > 0 || have_ratio) && dup_ratio < 0.6 ? (uint64_t)((double)todo * std::min(1.0, dup_ratio)) : todo;
            if (expect > room && !getenv("JASPER_EXPERIMENT_PIECE")) {
                if (ensure_capacity(expect, err)) return -1;
            }
        }
        uint64_t start = pos >= halo ? pos - halo : 0;
        // round the start down so that (d_bases + start) is 16-byte aligned
        const uint64_t a = (start + misalign) & 15;
        start = start >= a ? start - a : 0;
        const uint64_t end = std::min<uint64_t>(n, pos + piece);
        const uint64_t distinct_before = h_stats[ST_DISTINCT], occ_before = h_stats[ST_OCCURRENCES];
        const double t_l0 = dbg ? now_ms() : 0;
        if (dbg) fprintf(stderr, "[count] host: %.2f ms since call start (sizing, capacity)\n", t_l0 - t_call);
        histo_request = started_empty && pos == 0 && first_new == 0 && end == n && h_stats[ST_DISTINCT] == 0;   // one piece, whole input, empty table
        histo_cached = false;
        const int lrc = launch_count(d_bases + start, end - start, pos - start, err);
        const bool fused_histo = histo_request;      // still set only if the partitioned path took the request
        histo_request = false;
        if (lrc) return -1;
        const double t_l1 = dbg ? now_ms() : 0;
        const uint64_t pos_piece = pos;
        pos = end;
        int rc = after_batch(err);
        histo_cached = rc == 0 && fused_histo;
        if (rc == -2 && count_path == 1 && part_defer_header) {
            unsigned long long abandoned = 0;
            HIPCHK(hipMemcpy(&abandoned, part_defer_header + 1, 8, hipMemcpyDeviceToHost));
            if (abandoned) {
                // One k-mer (or a few) so frequent that its region list overflowed beyond the deferred list: the piece stopped before
                // it wrote to the table.  Counted again, like everything after it, by the direct kernel -- the k-mer's atomics
                // queue up on one address there, but they all arrive.
                if (dbg) {
                    float m1 = 0, m2 = 0;
                    (void)hipEventElapsedTime(&m1, ev_stage_t[0], ev_stage_t[1]);
                    (void)hipEventElapsedTime(&m2, ev_stage_t[1], ev_stage_t[2]);
                    fprintf(stderr, "[count] partitioned piece abandoned (lists overflowed the deferred list; its partition passes took %.2f + %.2f ms): direct kernel from here on\n", m1, m2);
                }
                err.clear();
                part_off = true;
                part_stage_pending = false;
                slots_dirty = part_slots_dirty_before;
                const unsigned long long zero = 0, occ = occ_before;
                HIPCHK(hipMemcpy(d.stats + ST_FATAL, &zero, 8, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(d.stats + ST_OCCURRENCES, &occ, 8, hipMemcpyHostToDevice));
                if (read_stats(err)) return -1;
                pos = pos_piece;
                continue;
            }
        }
        if (dbg) fprintf(stderr, "[count] host: launch calls %.2f ms, wait + after_batch %.2f ms\n", t_l1 - t_l0, now_ms() - t_l1);
        if (rc == -2 && have_ratio && started_empty) {
            // The size hint promised a more repetitive input than this one: the piece sized from it overflowed the table.
            // Nothing else was in the table when this call started, so start over with worst-case piece sizes.
            if (dbg) fprintf(stderr, "[count] size hint too small for this input: restarting with worst-case sizing\n");
            err.clear();
            if (clear(err)) return -1;
            if (read_stats(err)) return -1;
            if (grow(std::min(d.B, d.s + 2), err)) return -1;
            have_ratio = false;
            dup_ratio = 1.0;
            pos = std::min(first_new, n);
            continue;
        }
        if (rc) return rc;
        if (h_stats[ST_OCCURRENCES] > occ_before)
            dup_ratio = (double)(h_stats[ST_DISTINCT] - distinct_before) / (double)(h_stats[ST_OCCURRENCES] - occ_before);
        if (part_stage_pending) {
            for (int i = 0; i < part_stage_n; ++i) { float m = 0; if (hipEventElapsedTime(&m, ev_stage_t[i], ev_stage_t[i + 1]) == hipSuccess) part_stage_ms[i] += m; }
            part_stage_pending = false;
        }
        if (getenv("JASPER_COUNT_DEBUG"))
            fprintf(stderr, "[count] piece done: pos %llu / %llu, distinct %llu, slots 2^%d, dup_ratio %.3f, stage ms so far %.2f %.2f %.2f %.2f %.2f\n",
                    (unsigned long long)pos, (unsigned long long)n, (unsigned long long)h_stats[ST_DISTINCT], d.s, dup_ratio,
                    part_stage_ms[0], part_stage_ms[1], part_stage_ms[2], part_stage_ms[3], part_stage_ms[4]);
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev_k0, ev_k1));
        count_kernel_ms += ms;
        count_launches += 1;
    }
    return 0;
}

// bases in host memory: double-buffered pinned staging; the copy of piece i+1 overlaps the kernel of piece i
int Table::count_host(const char *bases, uint64_t n, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (read_stats(err)) return -1;
    histo_cached = histo_request = false;
    if (!stage_bytes) {
        stage_bytes = 64u << 20;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc((void **)&d_stage[i], stage_bytes + 64));
            HIPCHK(hipHostMalloc((void **)&h_stage[i], stage_bytes + 64, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&ev_stage[i], hipEventDisableTiming));
        }
    }
    const uint64_t halo = (uint64_t)(k - 1);
    // A large input is first brought to the device in super-pieces of up to 1 GiB (several threads copy into the pinned staging
    // buffers -- one thread's memcpy, ~9 GB/s, is otherwise what bounds this entry point -- while the DMA of the last chunk runs)
    // and then counted like device-resident bases: the atomic-free paths instead of 64-MiB launches of the direct kernel.
    if (n >= (96u << 20) && !getenv("JASPER_COUNT_HOST_STREAM")) {
        const uint64_t SP = 1ull << 30;
        uint8_t *d_all = reinterpret_cast<uint8_t *>(workspace(WS_HOSTBASES, std::min<uint64_t>(n, SP) + halo + 64 + 16, err));
        if (!d_all) return -1;
        unsigned nth = std::thread::hardware_concurrency();
        if (const char *w = getenv("WORLD_SIZE")) { const long nw = atol(w); if (nw > 1) nth /= (unsigned)nw; }
        nth = std::max(1u, std::min(8u, nth / 2));
        for (uint64_t sp = 0; sp < n; sp += SP) {
            const uint64_t start = sp >= halo ? sp - halo : 0, end = std::min(n, sp + SP), len = end - start;
            int buf = 0;
            for (uint64_t off = 0; off < len; off += stage_bytes) {
                const uint64_t m = std::min<uint64_t>(stage_bytes, len - off);
                HIPCHK(hipEventSynchronize(ev_stage[buf]));      // staging buffer free again?
                {
                    std::vector<std::thread> th;
                    const uint64_t per = (m + nth - 1) / nth;
                    for (unsigned i = 1; i < nth; ++i) {
                        const uint64_t a = std::min(m, i * per), b = std::min(m, (i + 1) * per);
                        if (b > a) th.emplace_back([=] { memcpy(h_stage[buf] + a, bases + start + off + a, b - a); });
                    }
                    memcpy(h_stage[buf], bases + start + off, std::min(m, per));
                    for (std::thread &t : th) t.join();
                }
                HIPCHK(hipMemcpyAsync(d_all + off, h_stage[buf], m, hipMemcpyHostToDevice, stream));
                HIPCHK(hipEventRecord(ev_stage[buf], stream));
                buf ^= 1;
            }
            const int rc = count_device(d_all, len, err, sp - start);
            if (rc) return rc;
        }
        return 0;
    }
    uint64_t pos = 0;
    int buf = 0;
    uint64_t pending = 0;   // bases launched since the table's load was last checked (worst case: all of them new keys)
    while (pos < n) {
        const uint64_t piece = std::min<uint64_t>(stage_bytes - halo, n - pos);
        // the load factor is only re-checked (a stream sync) when the worst case could pass 3/4: between checks the
        // host copy of piece i+1 into pinned memory overlaps the PCIe copy and the kernel of piece i
        if ((double)(h_stats[ST_DISTINCT] + pending + piece) > 0.75 * (double)nslots) {
            int rc = after_batch(err);
            if (rc) return rc;
            pending = 0;
            if (ensure_capacity(piece, err)) return -1;
        }
        const uint64_t start = pos >= halo ? pos - halo : 0;
        const uint64_t end = pos + piece;
        const uint64_t len = end - start;
        HIPCHK(hipEventSynchronize(ev_stage[buf]));  // staging buffer free again?
        memcpy(h_stage[buf], bases + start, len);
        HIPCHK(hipMemcpyAsync(d_stage[buf], h_stage[buf], len, hipMemcpyHostToDevice, stream));
        if (launch_count(d_stage[buf], len, pos - start, err)) return -1;
        HIPCHK(hipEventRecord(ev_stage[buf], stream));
        pos = end;
        pending += piece;
        buf ^= 1;
        count_launches += 1;
    }
    const int rc = after_batch(err);   // spill / growth handling and fresh statistics before returning
    if (rc) return rc;
    {   // kernel time of this call is not tracked per launch on the streaming path (events are reused): report the last one
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_k0, ev_k1) == hipSuccess) count_kernel_ms += ms;
    }
    return 0;
}

int Table::histogram(uint64_t *out, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (histo_cached) {   // taken while the counting pass wrote the table (count_part.hip); word HISTO_BINS = "not complete"
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "histogram words");
        std::vector<unsigned long long> h(HISTO_BINS + 1);
        HIPCHK(hipMemcpyAsync(h.data(), d_histo, (HISTO_BINS + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIPCHK(jk_stream_wait(stream));
        if (h[HISTO_BINS] == 0) {
            memcpy(out, h.data(), HISTO_BINS * sizeof(unsigned long long));
            return 0;
        }
        histo_cached = false;
    }
    unsigned long long *d_out = d_histo + HISTO_WORDS;
    HIPCHK(hipMemsetAsync(d_out, 0, HISTO_BINS * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(histo_kernel, dim3(grid_for(nslots, 256 * 16)), dim3(256), 0, stream, d, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, HISTO_BINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    return 0;
}

// histogram of the keys of ONE partition (multi-GPU: every rank bins the range it owns, the bins are summed over ranks)
__global__ __launch_bounds__(256) void histo_part_kernel(TableDev T, uint32_t part, uint32_t nparts, uint64_t first, uint64_t span,
                                                         unsigned long long *__restrict__ out) {
    __shared__ unsigned int bins[HISTO_BINS];
    for (int i = threadIdx.x; i < HISTO_BINS; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < span; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (first + q) & T.mask;
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
        if (e.x == 0ull || e.y == 0ull) continue;
        if (nparts > 1 && part_of(slot_hash(T, i, e.x), T.B, nparts) != part) continue;
        const uint32_t c = clamp32(e.y);
        atomicAdd(&bins[c > 10001u ? 10001u : c], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HISTO_BINS; i += blockDim.x)
        if (bins[i]) atomicAdd(&out[i], (unsigned long long)bins[i]);
}

int Table::histogram_part(uint32_t part, uint32_t nparts, uint64_t *out, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    uint64_t first, span;
    part_span(part, nparts, first, span);
    unsigned long long *d_out = d_histo + HISTO_WORDS;
    HIPCHK(hipMemsetAsync(d_out, 0, HISTO_BINS * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(histo_part_kernel, dim3(grid_for(span, 256 * 16)), dim3(256), 0, stream, d, part, nparts, first, span, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, HISTO_BINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    return 0;
}

int Table::lookup_strings(const char *chars, const int64_t *offsets, uint64_t n, uint32_t *out, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (n == 0) return 0;
    const uint64_t nchars = (uint64_t)offsets[n];
    char *d_chars = nullptr;
    int64_t *d_offs = nullptr;
    uint32_t *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_chars, nchars + 16));
    HIPCHK(hipMalloc((void **)&d_offs, (n + 1) * sizeof(int64_t)));
    HIPCHK(hipMalloc((void **)&d_out, n * sizeof(uint32_t)));
    HIPCHK(hipMemcpyAsync(d_chars, chars, nchars, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_offs, offsets, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(lookup_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_chars, d_offs, n, d_out, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    HIPCHK(hipFree(d_chars));
    HIPCHK(hipFree(d_offs));
    HIPCHK(hipFree(d_out));
    return 0;
}

int Table::export_entries(uint64_t *n_out, unsigned long long **d_entries_out, std::string &err) {
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (read_stats(err)) return -1;
    const uint64_t cap = h_stats[ST_DISTINCT];
    unsigned long long *d_e = nullptr, *d_ctr = nullptr;
    HIPCHK(hipMalloc((void **)&d_e, (cap ? cap : 1) * 3 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&d_ctr, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d_ctr, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(export_kernel, dim3(grid_for(nslots, 256 * 16)), dim3(256), 0, stream, d, d_e, d_ctr, cap);
    HIPCHK(hipGetLastError());
    unsigned long long got = 0;
    HIPCHK(hipMemcpyAsync(&got, d_ctr, sizeof got, hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    HIPCHK(hipFree(d_ctr));
    if (got != cap) { (void)hipFree(d_e); err = "export: entry count changed under us"; return -1; }
    *n_out = got;
    *d_entries_out = d_e;
    return 0;
}

int Table::import_entries(const unsigned long long *d_entries, uint64_t n, std::string &err) {
    histo_cached = false;
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (read_stats(err)) return -1;
    uint64_t pos = 0;
    while (pos < n) {
        const uint64_t room = (uint64_t)(0.75 * (double)nslots) > h_stats[ST_DISTINCT] ? (uint64_t)(0.75 * (double)nslots) - h_stats[ST_DISTINCT] : 0;
        uint64_t piece = std::min<uint64_t>(n - pos, std::max<uint64_t>(room, 1u << 20));
        if (piece > room) {
            if (ensure_capacity(piece, err)) return -1;
        }
        hipLaunchKernelGGL(import_kernel, dim3(grid_for(piece, 256)), dim3(256), 0, stream, d_entries + 3 * pos, piece, d);
        HIPCHK(hipGetLastError());
        pos += piece;
        int rc = after_batch(err);
        if (rc) return rc;
    }
    return 0;
}

int Table::reserve(uint64_t min_slots, std::string &err) {
    int ns = d.s;
    while ((1ull << ns) < min_slots && ns < d.B) ++ns;
    if (ns > d.s) return grow(ns, err);
    return 0;
}

// slots that can hold keys of partition part/nparts: partition p = keys whose top-32 hash bits t satisfy
// floor(t * nparts / 2^32) == p, i.e. t in [t_lo, t_hi); their home slots are a contiguous range, and an entry sits
// < MAXPROBE slots behind its home
void Table::part_span(uint32_t part, uint32_t nparts, uint64_t &first, uint64_t &span) const {
    first = 0;
    span = nslots;
    if (nparts <= 1 || d.B < 32) return;        // (B < 32: tiny key space, scan everything)
    const uint64_t t_lo = (((uint64_t)part << 32) + nparts - 1) / nparts;
    const uint64_t t_hi = ((((uint64_t)part + 1) << 32) + nparts - 1) / nparts;   // exclusive
    const int s_ = d.s;
    uint64_t last_excl;
    if (s_ <= 32) { first = t_lo >> (32 - s_); last_excl = ((t_hi - 1) >> (32 - s_)) + 1; }
    else { first = t_lo << (s_ - 32); last_excl = t_hi << (s_ - 32); }
    span = std::min<uint64_t>(nslots, last_excl - first + MAXPROBE);
}

static const char *WIDE_NO_EXCHANGE = "the multi-GPU table exchange packs a key and its count into 16 bytes and holds k <= 48, and a table that other GPUs "
                                      "read keeps whole remainders in its tags (k <= 43, at least 2^(2k-53) slots): run a larger k on one GPU";
// A table that peers read (shards) or that is built region by region from entry lists keeps the whole remainder in the tag
// word: grow it to the smallest such geometry if it is not there yet (2^(2k-53) slots; k <= 43 fits one GPU)
int Table::ensure_narrow(std::string &err) {
    if (!d.ext) return 0;
    const int need = d.B - (63 - OFFBITS);
    if (need > 34) { err = WIDE_NO_EXCHANGE; return -1; }
    return resize(std::max(need, d.s), err);
}
int Table::export_packed(void *d_dst, uint64_t cap, uint64_t *n_out, uint32_t part, uint32_t nparts, std::string &err) {
    if (d.B > 96) { err = WIDE_NO_EXCHANGE; return -1; }
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    uint64_t first = 0, span = nslots;
    part_span(part, nparts, first, span);
    unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(workspace(WS_COUNT + 2, (EXP_BLOCKS + 8) * 8 + 256, err));   // (free between counting calls)
    if (!d_counts) return -1;
    const uint64_t chunk = ((span + EXP_BLOCKS - 1) / EXP_BLOCKS + 255) / 256 * 256;
    hipLaunchKernelGGL(export_packed_count_kernel, dim3(EXP_BLOCKS), dim3(256), 0, stream, d, part, nparts, first, span, chunk, d_counts);
    hipLaunchKernelGGL(export_packed_scan_kernel, dim3(1), dim3(1024), 0, stream, d_counts);
    if (cap) hipLaunchKernelGGL(export_packed_write_kernel, dim3(EXP_BLOCKS), dim3(256), 0, stream, d, (ulonglong2 *)d_dst, cap, part, nparts, first, span,
                                chunk, d_counts);
    HIPCHK(hipGetLastError());
    unsigned long long got = 0;
    HIPCHK(hipMemcpyAsync(&got, d_counts + EXP_BLOCKS, sizeof got, hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    if (read_stats(err)) return -1;
    if (h_stats[ST_FATAL] == 2) { err = "a count does not fit the packed exchange format"; return -2; }
    *n_out = got;   // may exceed cap: the caller sizes its buffer with a first call (cap = 0) or from info()
    return 0;
}

int Table::export_owner(void *d_dst, uint64_t cap, uint32_t nown, int sort_r, uint64_t *counts_out, std::string &err) {
    if (nown < 1 || nown > MAX_SHARDS || sort_r < 0 || sort_r > 60 || sort_r > d.B) { err = "export_owner: 1..8 owners, 0 <= sort_r <= min(60, 2k)"; return -1; }
    if (d.B > 96) { err = WIDE_NO_EXCHANGE; return -1; }
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(workspace(WS_COUNT + 2, (size_t)MAX_SHARDS * EXP_STRIDE * 8 + 256, err));
    if (!d_counts) return -1;
    const uint64_t chunk = ((nslots + EXP_BLOCKS - 1) / EXP_BLOCKS + 255) / 256 * 256;
    hipLaunchKernelGGL(export_owner_count_kernel, dim3(EXP_BLOCKS), dim3(256), 0, stream, d, nown, sort_r, chunk, d_counts);
    hipLaunchKernelGGL(export_packed_scan_kernel, dim3(nown), dim3(1024), 0, stream, d_counts);
    if (cap) hipLaunchKernelGGL(export_owner_write_kernel, dim3(EXP_BLOCKS), dim3(256), 0, stream, d, (ulonglong2 *)d_dst, cap, nown, sort_r, chunk, d_counts);
    HIPCHK(hipGetLastError());
    unsigned long long got[MAX_SHARDS];
    HIPCHK(hipMemcpy2DAsync(got, 8, d_counts + EXP_BLOCKS, (size_t)EXP_STRIDE * 8, 8, nown, hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    if (read_stats(err)) return -1;
    if (h_stats[ST_FATAL] == 2) { err = "a count does not fit the packed exchange format"; return -2; }
    for (uint32_t o = 0; o < nown; ++o) counts_out[o] = got[o];
    return 0;
}

// ---- shards: lookups through this table read the owner's slot array ---------------------------------------------
void Table::detach_shards() {
    for (uint32_t i = 0; i < MAX_SHARDS; ++i) {
        if (ipc_mapped[i]) { (void)hipIpcCloseMemHandle(ipc_mapped[i]); ipc_mapped[i] = nullptr; }
        d.shard[i] = nullptr;
    }
    d.nshard = 0;
}

int Table::ipc_handle(void *out64, std::string &err) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI hands IPC handles over as 64 bytes");
    if (ensure_narrow(err)) return -1;
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, d.slots));
    memcpy(out64, &h, 64);
    exported = true;
    return 0;
}

void Table::release_retired() {
    for (void *p : retired) if (p) (void)hipFree(p);
    retired.clear();
}

// handles64: n handles of 64 bytes, one per owner in owner order (entry `self` is ignored: that is this table).  Every
// owner's table must have this table's geometry -- the caller agrees on it before (dist.shard_tables).
int Table::attach_ipc(const void *handles64, uint32_t n, uint32_t self, std::string &err) {
    if (n < 1 || n > MAX_SHARDS || self >= n) { err = "attach: 1..8 shards, self among them"; return -1; }
    if (ensure_narrow(err)) return -1;
    const bool dbg = getenv("JASPER_SHARD_DEBUG") != nullptr;
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (dbg) { fprintf(stderr, "[attach %u] waiting for the stream\n", self); fflush(stderr); }
    HIPCHK(jk_stream_wait(stream));
    detach_shards();
    for (uint32_t i = 0; i < n; ++i) {
        if (i == self) { d.shard[i] = d.slots; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles64 + 64 * (size_t)i, 64);
        void *p = nullptr;
        if (dbg) { fprintf(stderr, "[attach %u] opening shard %u\n", self, i); fflush(stderr); }
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (dbg) { fprintf(stderr, "[attach %u] shard %u -> %p (%s)\n", self, i, p, hipGetErrorString(e)); fflush(stderr); }
        if (e != hipSuccess) {
            err = std::string("hipIpcOpenMemHandle (shard ") + std::to_string(i) + "): " + hipGetErrorString(e);
            detach_shards();
            return -1;
        }
        ipc_mapped[i] = p;
        d.shard[i] = (const unsigned long long *)p;
        // a mapping that this GPU cannot actually load from would only show as a page fault inside a polishing kernel
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device != device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device, at.device) != hipSuccess || !can) {
                err = "GPU " + std::to_string(device) + " has no peer access to GPU " + std::to_string(at.device) + " (shard " + std::to_string(i) + ")";
                detach_shards();
                return -1;
            }
        }
    }
    d.nshard = n;
    return 0;
}

// the same for shard tables that live in THIS process (one GPU holding several shards: tests, or a table larger than
// one allocation)
int Table::attach_tables(Table *const *peers, uint32_t n, uint32_t self, std::string &err) {
    if (n < 1 || n > MAX_SHARDS || self >= n) { err = "attach: 1..8 shards, self among them"; return -1; }
    if (ensure_narrow(err)) return -1;
    for (uint32_t i = 0; i < n; ++i) if (peers[i] && peers[i] != this && peers[i]->ensure_narrow(err)) return -1;
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    detach_shards();
    for (uint32_t i = 0; i < n; ++i) {
        Table *p = i == self ? this : peers[i];
        if (!p || p->d.s != d.s || p->d.B != d.B) { err = "attach: all shards must have the same k and slot count"; detach_shards(); return -1; }
        if (p != this) {
            if (p->materialize(err)) { detach_shards(); return -1; }
            HIPCHK(jk_stream_wait(p->stream));
        }
        d.shard[i] = p->d.slots;
    }
    d.nshard = n;
    return 0;
}

int Table::import_packed_multi(const void *const *d_srcs, const uint64_t *counts, uint32_t n_src, std::string &err) {
    if (d.B > 96) { err = WIDE_NO_EXCHANGE; return -1; }
    if (ensure_narrow(err)) return -1;
    if (n_src < 1 || n_src > MAX_SHARDS) { err = "import_packed_multi: 1..8 lists"; return -1; }
    histo_cached = false;
    HIPCHK(hipSetDevice(device));
    if (read_stats(err)) return -1;
    MultiSrc S{};
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_src; ++i) { S.p[i] = (const ulonglong2 *)d_srcs[i]; S.n[i] = counts[i]; total += counts[i]; }
    S.n_src = n_src;
    if (!total) return materialize(err);
    // worst case every entry is a new key; if that could overfill the table, grow first (keys shared between the lists make
    // this generous -- callers that know better size the table themselves and never get here)
    if ((double)(h_stats[ST_DISTINCT] + total) > 0.9 * (double)nslots) {
        if (ensure_capacity(total, err)) return -1;
    }
    // an empty table is built region by region in LDS (no global atomic); lists of up to 2^32 entries, tables of >= 2 regions
    bool big = false;
    for (uint32_t i = 0; i < n_src; ++i) big = big || counts[i] >= 0xFFFFFFFFull;
    if (h_stats[ST_DISTINCT] == 0 && h_stats[ST_SPILL] == 0 && !big && nslots >= 2 * IMPR && !getenv("JASPER_IMPORT_ATOMIC")) {
        const uint32_t nreg = (uint32_t)(nslots >> IMPR_LOG);
        const uint64_t dcap = total / 8 + (1u << 20);
        const size_t hwords = (size_t)n_src * (nreg + 1);
        unsigned int *d_ws = reinterpret_cast<unsigned int *>(workspace(WS_COUNT + 2, (hwords + 2) * 4 + 64, err));   // [deferred counter (8 B)][histograms]
        ulonglong2 *d_def = reinterpret_cast<ulonglong2 *>(workspace(WS_COUNT + 0, dcap * 16 + 64, err));
        if (!d_ws || !d_def) return -1;
        unsigned long long *d_defn = reinterpret_cast<unsigned long long *>(d_ws);
        unsigned int *d_b = d_ws + 2;
        slots_dirty = false;               // every slot is written below: a lazily cleared table needs no zeroing first
        HIPCHK(hipMemsetAsync(d_ws, 0, (hwords + 2) * 4, stream));
        hipLaunchKernelGGL(imp_region_hist_kernel, dim3(2048), dim3(256), 0, stream, S, d, nreg, d_b);
        hipLaunchKernelGGL(imp_region_scan_kernel, dim3(n_src), dim3(1024), 0, stream, d_b, nreg);
        hipLaunchKernelGGL(lds_import_kernel, dim3(nreg), dim3(256), 0, stream, S, d, nreg, d_b, d_def, d_defn, (unsigned long long)dcap);
        HIPCHK(hipGetLastError());
        unsigned long long n_def = 0;
        HIPCHK(hipMemcpyAsync(&n_def, d_defn, 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(jk_stream_wait(stream));
        if (getenv("JASPER_COUNT_DEBUG")) fprintf(stderr, "[import] %llu entries into %u LDS regions, %llu deferred\n", (unsigned long long)total, nreg, n_def);
        if (n_def <= dcap) {
            if (n_def) {
                hipLaunchKernelGGL(import_packed_kernel, dim3(grid_for(n_def, 256)), dim3(256), 0, stream, d_def, (uint64_t)n_def, d, 0);
                HIPCHK(hipGetLastError());
            }
            return after_batch(err);
        }
        // the lists were not in slot order (too many entries outside their piece): start over with the atomic kernel
        HIPCHK(hipMemsetAsync(d.stats + ST_DISTINCT, 0, sizeof(unsigned long long), stream));
        if (zero_slots(d.slots, nslots, err)) return -1;
    }
    if (materialize(err)) return -1;
    const uint64_t band = 32ull << 20;                                   // bytes of slots per chunk
    const uint32_t nchunks = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, nslots * 16 / band));
    hipLaunchKernelGGL(import_packed_multi_kernel, dim3(nchunks * IMP_BPC), dim3(256), 0, stream, S, nchunks, d);
    HIPCHK(hipGetLastError());
    return after_batch(err);
}

int Table::import_packed(const void *d_src, uint64_t n, int mode, std::string &err) {
    if (d.B > 96) { err = WIDE_NO_EXCHANGE; return -1; }
    histo_cached = false;
    HIPCHK(hipSetDevice(device));
    if (materialize(err)) return -1;
    if (read_stats(err)) return -1;
    uint64_t pos = 0;
    while (pos < n) {
        const uint64_t room = (uint64_t)(0.75 * (double)nslots) > h_stats[ST_DISTINCT] ? (uint64_t)(0.75 * (double)nslots) - h_stats[ST_DISTINCT] : 0;
        uint64_t piece = std::min<uint64_t>(n - pos, std::max<uint64_t>(room, 1u << 20));
        if (piece > room) {
            if (ensure_capacity(piece, err)) return -1;
        }
        hipLaunchKernelGGL(import_packed_kernel, dim3(grid_for(piece, 256)), dim3(256), 0, stream, (const ulonglong2 *)d_src + pos, piece, d, mode);
        HIPCHK(hipGetLastError());
        pos += piece;
        int rc = after_batch(err);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace jk
