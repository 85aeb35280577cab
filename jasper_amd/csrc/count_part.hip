// count_part.hip -- counting without global atomics on the table ("partitioned" path of K1+K2).
//
// Why: the direct kernel (table.hip: count_kernel) issues one scattered 64-bit atomic per k-mer, and MI355X retires
// only ~20 G scattered atomics/s chip-wide whatever the working set (measured: hashing alone 190 Gk-mers/s, hashing +
// one random 8-B load 48 Gk-mers/s, full insert 18 Gk-mers/s, unchanged when the touched slots fit in L2; see
// DESIGN.md 4.1).  Here the table is updated with LDS atomics instead:
//
//   part1_kernel   bases -> mixed hash -> bucket = top p1 hash bits; the remaining (2k-p1 <= 64) bits go, as one
//                  8-byte record, into the bucket's list.  Per 16 K-base tile: LDS histogram (returning LDS atomics give
//                  every record its rank), block scan, records sorted by bucket inside LDS, copied out in bucket order
//                  into the list's slice, at a place taken by one returning atomic per bucket and tile.
//   part2f_kernel  same scheme on each bucket, by the next p2 hash bits  ->  2^(p1+p2) lists, one per table REGION
//                  of 2^rbits <= 8192 consecutive slots (home slot = top hash bits, so a region is a hash range); slices
//                  written in whole 128-byte lines.  (part2_kernel: the general form -- more than 128 lists per bucket,
//                  lists laid out by key owner for the exchange between GPUs, an owner's extra split pass.)
//   region_insert_kernel  one workgroup per region: an LDS image of the region's slots (started from zeros on a table
//                  that is logically empty), records inserted with LDS compare-and-swap / add, image written back --
//                  and, when this pass produces the final counts of the whole table, binned into the multiplicity
//                  histogram on the way out.  One launch; the owner's side of the exchange reads N senders' slices.
//   A record that finds no room (slice full, probe past the region's end) goes to a deferred list that import3h_kernel
//   drains through the direct path afterwards.
//
// The table layout, tags and probe order are exactly those of the direct path, so lookups, histogram, export, growth
// and the polisher do not know which path filled the table.  HBM traffic per k-mer: 1 B base + 8 B x 2 (part1 list)
// + 8 B x 2 (part2 list) + the table once in, once out -- all of it streaming.
#include "table.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
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

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL stores and loads
// (s_waitcnt vmcnt(0)) -- here that would make every tile wait for the copy-out of the previous phase to reach memory, and
// for the prefetch of the next tile, although nothing in the workgroup reads those bytes back.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int PT_THREADS = 1024;
constexpr int PT_GROUP = 16;
constexpr int PT_TILE = PT_THREADS * PT_GROUP;   // 16384 records per block iteration
constexpr int PT_HALO = 4;
constexpr int PT_MAXBUCKETS = 2048;              // p1, p2 <= 11
constexpr int RG_MAXBITS = 12;                   // region = 4096 slots: 48 KB of LDS in region_insert_kernel, three workgroups per CU (8192 for the largest tables)

struct PartGeom {
    int p1, p2, rbits;       // p1 + p2 + rbits == s
    int recbits;             // 2k - p1  (<= 64): bits kept in a record
    uint32_t nblk1;          // slices per level-1 list: the part1 blocks b, b + nblk1, b + 2 nblk1, ... append to slice b % nblk1 of every list
    uint32_t grid1;          // part1 blocks
    uint32_t nblk2;          // part2 blocks per level-1 bucket: every one owns a slice of each of the bucket's region lists
    uint32_t cap1, cap2;     // slice capacities (records)
};
// level-1 list of bucket b = slices  out1[(b * nblk1 + g) * cap1 ...], filled counts cnt1[g * 2^p1 + b]  (slice-major: a wave's 64
// fill-count atomics in part1 -- lane = bucket -- are then 256 contiguous bytes, four requests to the memory side instead of 8 x nblk1)
// region list of region r  = slices  out2[(r * nblk2 + x) * cap2 ...],   filled counts cnt2[r * nblk2 + x]

__device__ __forceinline__ uint64_t rec_of(u128 h, int recbits) { return recbits >= 64 ? h.lo : (h.lo & ((1ull << recbits) - 1ull)); }
__device__ __forceinline__ u128 hash_of(uint64_t b1, uint64_t rec, int recbits) { return bor(shl(mk(0, b1), recbits), mk(0, rec)); }

// a record that found no room in its list waits for the direct path until every LDS image has been written back
__device__ __forceinline__ void defer_record(unsigned long long *stats, u128 h, unsigned long long *deferred, unsigned long long *deferred_n, uint64_t deferred_cap) {
    const unsigned long long di = atomicAdd(deferred_n, 1ull);
    if (di < deferred_cap) { deferred[3 * di] = h.hi; deferred[3 * di + 1] = h.lo; deferred[3 * di + 2] = 1ull; }
    else atomicExch(&stats[ST_FATAL], 1ull);
}
__device__ __forceinline__ void defer_record(const TableDev &T, u128 h, unsigned long long *deferred, unsigned long long *deferred_n, uint64_t deferred_cap) {
    defer_record(T.stats, h, deferred, deferred_n, deferred_cap);
}

// ---- level 1: bases -> records in 2^p1 bucket lists --------------------------------------------------------
// A tile's records are first sorted by bucket inside LDS (rank from a returning LDS atomic, offsets from a block scan) and then
// copied out in bucket order: consecutive lanes write consecutive records of one slice.  (Writing each 8-B record straight from
// the thread that produced it cost 3.9x the bytes in WRITE_SIZE.)
// A tile's run in one bucket is ~13 records = 108 bytes: it begins and ends inside 128-byte lines.  With a slice per BLOCK every
// line is written by two consecutive tiles of its block, a tile apart, and the open lines of all blocks (256 x 1024 lists x 128 B)
// are as large as the L2s together: the half-written lines are evicted and written twice (round 3: +1.9 ms of 5.7).  So the
// blocks SHARE their slices: block b appends to slice b % nblk1 of every list -- nblk1 = 8, i.e. the blocks of one XCD (blocks are
// dispatched round-robin over the XCDs: tools/probes/xcc_probe.hip), so the writers of a slice share an L2 -- and a tile's run
// gets its place by ONE returning agent-scope atomic add per bucket on the slice's fill count (thread t: bucket t).  The fill
// counts are laid out [slice][bucket]: atomics execute at the memory side, ~20 G requests/s chip-wide, and a wave's 64 atomics are
// then 256 contiguous bytes = 4 requests (laid out [bucket][slice] they were 8 x nblk1 requests, and the 88 M atomics of a
// 47-Mb call were what bounded the kernel: 4.4 ms).  The atomic's result is not needed before the copy-out: it is in flight
// during the scan and the staging.  WRITE_SIZE: 9.15 GB for 8.57 GB of records (round 3: 10.9).
//
// The kernel is bound by vector-ALU issue (without any store it takes 3.3 of its 4.0-4.3 ms), so it is written for few
// instructions per base:
//   * NW = number of 32-bit words of a k-mer is a template parameter: both strands roll by v_alignbit_b32 on words (2 x NW
//     instructions per base; 64-bit shifts by run-time amounts cost several each), the canonical choice is one compare chain;
//   * the hash is ONE 64-bit multiply (kmer.hpp: mix);
//   * no branch in the per-base loop: a window that is no k-mer takes its "rank" from a word of its lane;
//   * the copy-out is per RECORD, not per bucket: a second LDS array holds the bucket of every staged record, so a lane needs
//     three LDS reads and one multiply-add for its destination (the per-bucket loop of round 2 spent ~35 instructions per
//     record on 16-lane groups that were 3/4 full) -- and it is issued inside the NEXT tile's hash loop (below);
//   * four barriers per tile (the rank counters are double-buffered).
// The stage holds P1_STAGE records; a tile with more (only input without read boundaries: a genome) is staged in two rounds.
constexpr int P1_MAXB = 1024;          // part1 handles p1 <= 10; larger p1 falls back to the direct kernel
constexpr int P1_TH = 1024;
constexpr int P1_STAGE = 13824;        // records per staging round (a tile of 150-base reads holds ~12.4 K)
constexpr size_t P1_LDS = (size_t)P1_STAGE * 10 + (size_t)(2 * (P1_MAXB + 4) + 32 + 64) * 4 + (size_t)2 * (P1_TH + PT_HALO) * 4 + (size_t)P1_MAXB * 8;
typedef __attribute__((address_space(1))) uint64_t global_u64;      // a pointer known to be global memory (kept as an integer in LDS)
// Timing experiments on part1's copy-out (builds under ab/ only, `make EXTRA=-DJK_P1_EXP=n`; the lists they leave are garbage):
//   1 = the stores go to a 16-MB window (cache-resident: everything but the HBM writes), 2 = no stores at all,
//   4 = the same bytes as one sequential stream per block
#ifndef JK_P1_EXP
#define JK_P1_EXP 0
#endif
__device__ __forceinline__ void p1_store(uint64_t base, uint32_t idx, uint64_t rr, const uint64_t *out1, uint64_t seq) {
#if JK_P1_EXP == 1
    const uint64_t off = (base + (uint64_t)idx * 8ull - reinterpret_cast<uint64_t>(out1)) & ((16ull << 20) - 8ull);
    *reinterpret_cast<global_u64 *>(reinterpret_cast<uint64_t>(out1) + off) = rr;
#elif JK_P1_EXP == 2
    asm volatile("" :: "v"(base), "v"(idx), "v"(rr));
#elif JK_P1_EXP == 4
    asm volatile("" :: "v"(base));
    reinterpret_cast<global_u64 *>(reinterpret_cast<uint64_t>(out1))[seq + idx] = rr;
#else
    reinterpret_cast<global_u64 *>(base)[idx] = rr;
#endif
}

// inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row shifts inside the rows of 16, then the two row broadcasts)
__device__ __forceinline__ unsigned int wave_scan_incl(unsigned int v) {
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);    // row_bcast:15 -> rows 1 and 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);    // row_bcast:31 -> rows 2 and 3
    return v;
}

struct P1Args {              // what part1_kernel needs of the table and the geometry (few scalar registers: the kernel is short of them)
    int k, p1, recbits;
    uint32_t nblk1, cap1;
    unsigned long long *stats;
};
// KFIX: 0 = k, p1 and the record width are read from P; 37 = the reference's default k (src/jasper.sh:11) with the geometry that
// k implies (2k - 64 = 10 = p1 bucket bits above a 64-bit record): the masks and shift amounts of the per-base loop are then
// immediates instead of scalar registers (the kernel has more loop constants than scalar registers: they were spilled to vector
// lanes and read back by v_readlane in every iteration).
template <int NW, int KFIX = 0>
__global__ __launch_bounds__(P1_TH) void part1_kernel(const uint8_t *__restrict__ bases, uint64_t n, uint64_t ntiles, uint64_t emit_from, P1Args P,
                                                      uint64_t *__restrict__ out1, unsigned int *__restrict__ cnt1, unsigned long long *__restrict__ deferred,
                                                      unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    // (the stage comes first: the copy-out reads stage entry t + 1024 j, and with the stage at offset 0 that is ONE address register
    //  per 64 KB plus the 16-bit offset field of the LDS instruction -- no address arithmetic per record)
    // where stage index 0 of the staged tile goes in my slice of each bucket's list (slice base + (cursor - offset) records), as an integer
    uint64_t *s_base = reinterpret_cast<uint64_t *>(s_raw);                                  // P1_MAXB (at offset 0: its address is one shift of the bucket)
    uint64_t *s_stage = s_base + P1_MAXB;                                                    // P1_STAGE records, bucket order
    unsigned short *s_bkt = reinterpret_cast<unsigned short *>(s_stage + P1_STAGE);          // bucket of each staged record
    // s_cnt2: two copies, used by alternate tiles -- records of the tile per bucket while they are ranked (A), then, in place, the
    // exclusive prefix of those counts (B, C); the copy of the tile before is cleared meanwhile
    unsigned int *s_cnt2 = reinterpret_cast<unsigned int *>(s_bkt + P1_STAGE);               // 2 x (P1_MAXB+4)
    unsigned int *s_wsum = s_cnt2 + 2 * (P1_MAXB + 4);                                       // 16 wave totals, [16] = "a slice overflows", [24..31] the tile's last 64 bases
    unsigned int *s_dummy = s_wsum + 32;                                                     // 64: what the rank atomics of windows that are no k-mer add to (one word per lane)
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_dummy + 64);                           // P1_TH + PT_HALO
    uint32_t *s_inv = s_code + (P1_TH + PT_HALO);
    const int t = threadIdx.x;
    const int k = KFIX ? KFIX : P.k;
    const int p1 = KFIX ? 2 * KFIX - 64 : P.p1;
    const int recbits = KFIX ? 64 : P.recbits;
    const int nb = 1 << p1;
    // word-level constants: the k-mer's 2k bits fill words 0 .. NW-1, `topbits` of them in word NW-1
    const int topbits = 2 * k - 32 * (NW - 1);                                  // 2..32
    const uint32_t topmask = topbits >= 32 ? ~0u : ((1u << topbits) - 1u);
    const int rsh = topbits - 2;                                                // where the complement of a new base enters rc's top word
    const int hb = NW > 2 ? 2 * k - 64 : 0;                                     // hash bits above the low 64 (2k > 64)
    const uint32_t himask = NW > 2 ? (hb >= 32 ? ~0u : ((1u << hb) - 1u)) : 0u;
    const uint64_t lomask = NW > 2 || 2 * k >= 64 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const int hh = k;                                                           // (2k <= 64) xor-shift by half the width
    const int e = NW > 2 ? p1 - hb : 0;                                         // bucket bits taken from the low 64 hash bits (2k > 64: p1 >= hb)
    const uint64_t recmask = recbits >= 64 ? ~0ull : ((1ull << recbits) - 1ull);
    // a window is a k-mer iff none of its k bases is "no base": OR of the invalid-base bits over every window of k positions,
    // for the tile's 16 windows at once (the largest power of two <= k by doubling, then two such windows that overlap)
    int wlog = 0;
    while ((2 << wlog) <= k) ++wlog;                                            // 2^wlog <= k < 2^(wlog+1)
    unsigned long long added = 0;
    const uint32_t grp = blockIdx.x % P.nblk1;                                  // my slice of every list, shared with the blocks grp + j * nblk1
    s_cnt2[t] = 0;
    s_cnt2[P1_MAXB + 4 + t] = 0;
    for (int i = t; i < P1_STAGE; i += P1_TH) s_bkt[i] = 0;                     // (the copy-out reads the bucket of stage entries that hold nothing: any bucket will do)
    if (t == 0) s_wsum[16] = 0;
    // a block takes a run of consecutive tiles: the 64 bases before a tile are then the tail of the tile before it, still in LDS
    // (tile numbers are block-uniform 32-bit scalars: a piece has at most 2^31 bases)
    const uint32_t nt32 = (uint32_t)ntiles;
    const uint32_t per_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nt32 + gridDim.x - 1) / gridDim.x));
    const uint32_t tile_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * per_block));
    const uint32_t tile_end = tile_first + per_block < nt32 ? tile_first + per_block : nt32;
    // THE TILE LOOP IS A PIPELINE OF TWO TILES.  Hashing (phase A) is vector-ALU work, the copy-out (phase D) is LDS reads and
    // global stores: run one after the other by all 16 waves, each leaves the other's units idle (round 3: 15 us per tile, of which
    // the ALU was busy 6).  So the copy-out of tile i is issued INSIDE the hash loop of tile i + 1, one staged record per hashed
    // base: the stage, the bucket ids and the slice places of tile i are not touched by phase A, and the prefix sums of tile
    // i + 1 (B) and its staging (C) come after a barrier behind both.
    // Vector-memory operations of a wave retire in order: the text of tile i + 1 is requested at the top of tile i, BEFORE the
    // stores of D(i - 1) are issued, and turned into codes at the end of tile i together with the slice places (a returning
    // atomic issued in B) -- by then those stores are one scan and one staging phase old.
    uint32_t c = 0, iv = 0xFFFFu;
    if (tile_first < tile_end) {
        uint32_t hc = 0, hiv = 0xFFFFu;
        const int64_t b0 = (int64_t)tile_first * PT_TILE;
        stage16(bases, b0 + (int64_t)t * PT_GROUP, (int64_t)n, c, iv);
        if (t < PT_HALO) stage16(bases, b0 - (int64_t)(PT_HALO - t) * PT_GROUP, (int64_t)n, hc, hiv);
        s_code[t + PT_HALO] = c;
        s_inv[t + PT_HALO] = iv;
        if (t < PT_HALO) { s_code[t] = hc; s_inv[t] = hiv; }
    }
    lds_barrier();
    // staged records [0, pend_nr) wait to be copied out: stage index i goes to position pend_r0 + i behind s_base[bucket]  (block-uniform)
    unsigned int pend_nr = 0, pend_r0 = 0;
    uint64_t exp_seq = (uint64_t)blockIdx.x * per_block * PT_TILE;      // (JK_P1_EXP == 4: where this block's stream stands)
    // the copy-out by itself: a staging round that is not a tile's last, the last tile's, and whenever a slice is full
    auto copy_out = [&](unsigned int r0, unsigned int nr) {
        if (!s_wsum[16]) {                                                          // (block-uniform) no slice of mine is full
#pragma unroll 2
            for (unsigned int i = t; i < nr; i += P1_TH) {
                const uint64_t rr = s_stage[i];
                const uint32_t b = s_bkt[i];
                p1_store(s_base[b], r0 + i, rr, out1, exp_seq);
            }
        } else {
            for (unsigned int i = t; i < nr; i += P1_TH) {
                const uint64_t rr = s_stage[i];
                const uint32_t b = s_bkt[i];
                const uint64_t first = reinterpret_cast<uint64_t>(out1 + ((uint64_t)b * P.nblk1 + grp) * P.cap1);
                const uint64_t pos = (uint64_t)(((int64_t)(s_base[b] - first) >> 3) + (int64_t)(r0 + i));   // position in the bucket's slice (s_base may lie below it)
                if (pos < P.cap1) reinterpret_cast<global_u64 *>(first)[pos] = rr;
                else defer_record(P.stats, hash_of((uint64_t)b, rr, P.recbits), deferred, deferred_n, deferred_cap);   // slice full
            }
        }
    };
    constexpr int NCO = (P1_STAGE + P1_TH - 1) / P1_TH;                             // staged records per thread (14)
    static_assert(NCO + 2 <= PT_GROUP, "the copy-out pipeline must fit the hash loop");
    for (uint32_t tile = tile_first; tile < tile_end; ++tile) {
        const int64_t base0 = (int64_t)tile * PT_TILE;
        const bool has_next = tile + 1 < tile_end;
        unsigned int *s_cnt = s_cnt2 + ((tile - tile_first) & 1u) * (P1_MAXB + 4);
        unsigned int *s_off = s_cnt;
        unsigned int *s_nxt = s_cnt2 + (((tile - tile_first) & 1u) ^ 1) * (P1_MAXB + 4);
        int ta = t;
        asm volatile("" : "+v"(ta));
        // (only a next tile that lies inside the text as a whole: ONE load instruction, no branch with a slow side whose merge
        //  would make the compiler wait for the load right here; the piece's last tile is read at the end of this one instead)
        const bool prefetch = has_next && (uint64_t)(base0 + 2 * (int64_t)PT_TILE) <= n;
        if (pend_nr && s_wsum[16]) {                                                // (block-uniform) a slice is full: record by record, with the bound
            copy_out(pend_r0, pend_nr);
            pend_nr = 0;
        }
        // A. hash my 16 windows, take a rank in the tile's bucket histogram
        uint32_t f[NW], r[NW];
        uint32_t vmask;                    // bit 15 - j: the window ending at my base j is a k-mer
        {
            const uint32_t w4 = s_code[ta], w3 = s_code[ta + 1], w2 = s_code[ta + 2], w1 = s_code[ta + 3];
            const uint64_t ivprev = ((uint64_t)s_inv[ta] << 48) | ((uint64_t)s_inv[ta + 1] << 32) | ((uint64_t)s_inv[ta + 2] << 16) | (uint64_t)s_inv[ta + 3];
            const u128 fwd0 = band(mk(((uint64_t)w4 << 32) | w3, ((uint64_t)w2 << 32) | w1), maskbits(2 * k));
            const u128 rc0 = revcomp(fwd0, k);
            const uint32_t fw[4] = {(uint32_t)fwd0.lo, (uint32_t)(fwd0.lo >> 32), (uint32_t)fwd0.hi, (uint32_t)(fwd0.hi >> 32)};
            const uint32_t rw[4] = {(uint32_t)rc0.lo, (uint32_t)(rc0.lo >> 32), (uint32_t)rc0.hi, (uint32_t)(rc0.hi >> 32)};
#pragma unroll
            for (int w = 0; w < NW; ++w) { f[w] = fw[w]; r[w] = rw[w]; }
            // bit q of z: my base 15 - q (q < 16), or the base q - 15 positions before my first (k <= 37: bits up to 15 + 36 matter)
            uint64_t z = (uint64_t)iv | (ivprev << 16);
            uint64_t sm = z;
#pragma unroll
            for (int b = 0; b < 6; ++b) if (b < wlog) sm |= sm >> (1 << b);         // windows of 2^wlog positions
            sm |= sm >> (k - (1 << wlog));                                          // windows of k positions
            vmask = ~(uint32_t)sm & 0xFFFFu;
        }
        // k-mers that END before emit_from belong to the piece before this one
        if ((uint64_t)base0 < emit_from) {                                          // (block-uniform: the piece's first tiles only)
            const int64_t mine = base0 + (int64_t)ta * PT_GROUP;
            const int jfirst = (int64_t)emit_from > mine ? (int)((int64_t)emit_from - mine < PT_GROUP ? (int64_t)emit_from - mine : PT_GROUP) : 0;
            vmask &= 0xFFFFu >> jfirst;
        }
        uint64_t rec[PT_GROUP];
        uint32_t br[PT_GROUP];     // bucket (0xFFFF = no record) << 16 | rank among the tile's records of that bucket
        uint32_t bprev = 0xFFFFu, rprev = 0;      // (a record's two halves are put together one iteration later: the atomic's latency is covered)
        // D of the tile before, in three steps that are one iteration apart: a staged record's bucket (LDS) -> the record and the
        // bucket's place (LDS) -> the store
        uint64_t co_rr[NCO], co_base[NCO];
        uint32_t co_b[NCO];
#pragma unroll
        for (int j = 0; j < PT_GROUP; ++j) {
            if (j < NCO) {
                unsigned int i = (unsigned int)ta + (unsigned int)j * P1_TH;
                if ((j + 1) * P1_TH > P1_STAGE) i = i < (unsigned int)P1_STAGE ? i : (unsigned int)P1_STAGE - 1u;     // (the stage's last, partial row)
                co_b[j] = s_bkt[i];
            }
            const uint32_t cj = (c >> (30 - 2 * j)) & 3u;
            // forward strand: shift left by one base; reverse complement: shift right, the complement enters on top
#pragma unroll
            for (int w = NW - 1; w > 0; --w) f[w] = __builtin_amdgcn_alignbit(f[w], f[w - 1], 30);
            f[0] = (f[0] << 2) | cj;
            f[NW - 1] &= topmask;
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) r[w] = __builtin_amdgcn_alignbit(r[w + 1], r[w], 2);
            r[NW - 1] = (r[NW - 1] >> 2) | ((cj ^ 3u) << rsh);
            const bool valid = (vmask >> (15 - j)) & 1u;
            // canonical = numeric min of the two strands: borrow chain of r - f from the low word up, then one select per word
            // (written out: the compiler turns the same chain into 2 compares per word plus scalar mask logic)
            uint32_t m[NW];
            if constexpr (NW == 3) {
                uint32_t d;
                asm("v_sub_co_u32_e32 %3, vcc, %4, %7\n\tv_subb_co_u32_e32 %3, vcc, %5, %8, vcc\n\tv_subb_co_u32_e32 %3, vcc, %6, %9, vcc\n\t"
                    "v_cndmask_b32_e32 %0, %7, %4, vcc\n\tv_cndmask_b32_e32 %1, %8, %5, vcc\n\tv_cndmask_b32_e32 %2, %9, %6, vcc"
                    : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(d) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(f[0]), "v"(f[1]), "v"(f[2]) : "vcc");
            } else if constexpr (NW == 2) {
                uint32_t d;
                asm("v_sub_co_u32_e32 %2, vcc, %3, %5\n\tv_subb_co_u32_e32 %2, vcc, %4, %6, vcc\n\t"
                    "v_cndmask_b32_e32 %0, %5, %3, vcc\n\tv_cndmask_b32_e32 %1, %6, %4, vcc"
                    : "=&v"(m[0]), "=&v"(m[1]), "=&v"(d) : "v"(r[0]), "v"(r[1]), "v"(f[0]), "v"(f[1]) : "vcc");
            } else {
                m[0] = r[0] < f[0] ? r[0] : f[0];
            }
            uint32_t b;
            if constexpr (NW > 2) {                                                 // 64 < 2k <= 74 (p1 <= 10 bucket bits cover the bits above 64): kmer.hpp mix()
                const uint64_t lo = mix64(((uint64_t)m[1] << 32) | m[0]);
                const uint32_t lo_hi = (uint32_t)(lo >> 32);
                const uint32_t hi = (m[2] ^ __builtin_amdgcn_alignbit(lo_hi, (uint32_t)lo, 30)) & himask;
                // bucket = top p1 bits of the 2k-bit hash (hi: hb bits, lo: 64 bits), p1 >= hb; record = the rest
                b = (hi << e) | ((lo_hi >> 1) >> (31 - e));
                rec[j] = lo & recmask;
            } else {
                uint64_t v = m[0];
                if constexpr (NW > 1) v |= (uint64_t)m[1] << 32;
                v ^= v >> hh;
                v = (v * JK_C1) & lomask;
                v ^= v >> hh;
                b = (uint32_t)(v >> recbits);
                rec[j] = v & recmask;
            }
            if (j > 0) br[j - 1] = (bprev << 16) | rprev;
            bprev = valid ? b : 0xFFFFu;
            // LDS returning atomic, no branch around it: a window that is no k-mer takes a "rank" nobody looks at from a word of its lane
            rprev = atomicAdd(valid ? &s_cnt[b] : &s_dummy[ta & 63], 1u);
            if (j >= 1 && j - 1 < NCO) {
                unsigned int i = (unsigned int)ta + (unsigned int)(j - 1) * P1_TH;
                if (j * P1_TH > P1_STAGE) i = i < (unsigned int)P1_STAGE ? i : (unsigned int)P1_STAGE - 1u;
                co_rr[j - 1] = s_stage[i];
                co_base[j - 1] = s_base[co_b[j - 1]];
            }
            if (j >= 2 && j - 2 < NCO) {
                const unsigned int i = (unsigned int)ta + (unsigned int)(j - 2) * P1_TH;
                if (i < pend_nr) p1_store(co_base[j - 2], pend_r0 + i, co_rr[j - 2], out1, exp_seq);
            }
        }
        br[PT_GROUP - 1] = (bprev << 16) | rprev;
        if (JK_P1_EXP == 4) exp_seq += pend_nr;
        pend_nr = 0;
        // the next tile's text: asked for behind the copy-out stores and looked at after the staging, together with the slice
        // places (asked for in B) -- what is waited for there is the stores and both of these
        Raw16 raw;
        {   // (unconditional: without a next tile to fetch, the first 16 bytes of the text are read and ignored)
            int tb = t;
            asm volatile("" : "+v"(tb));
            struct __attribute__((packed, aligned(1))) V16 { uint32_t w[4]; };
            const V16 v = *reinterpret_cast<const V16 *>(bases + (prefetch ? base0 + (int64_t)PT_TILE + (int64_t)tb * PT_GROUP : (int64_t)0));
            raw.w[0] = v.w[0]; raw.w[1] = v.w[1]; raw.w[2] = v.w[2]; raw.w[3] = v.w[3];
            // the tail of this tile = the 64 bases before the next (parked in spare words: the head of s_code is still being read)
            if (has_next && tb < PT_HALO) { s_wsum[24 + tb] = s_code[P1_TH + tb]; s_wsum[28 + tb] = s_inv[P1_TH + tb]; }
        }
        lds_barrier();
        // B. exclusive prefix of the bucket counts: thread t owns bucket t (wave scan + wave totals); the place of the tile's run
        //    in the slice is asked for here and looked at after the staging
        unsigned int total;
        unsigned int apos = 0;
        {
            // (the thread number through an opaque copy: what is derived from it here is recomputed per tile -- hoisted out of the
            //  tile loop as loop invariants these addresses and masks end up spilled to scratch, and every reload waits for vmcnt(0))
            int tb = t;
            asm volatile("" : "+v"(tb));
            const unsigned int v = tb < nb ? s_cnt[tb] : 0u;
            if (v) apos = __hip_atomic_fetch_add(&cnt1[((uint32_t)grp << p1) + (uint32_t)tb], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int inc = wave_scan_incl(v);
            if ((tb & 63) == 63) s_wsum[tb >> 6] = inc;
            lds_barrier();
            // the 16 wave totals: scanned again by every wave (lanes 0..15), base of my wave and tile total by lane reads
            const unsigned int ws = wave_scan_incl((tb & 63) < P1_TH / 64 ? s_wsum[tb & 63] : 0u);
            const unsigned int wbase = (tb >> 6) ? (unsigned int)__builtin_amdgcn_readlane((int)ws, (tb >> 6) - 1) : 0u;
            total = (unsigned int)__builtin_amdgcn_readlane((int)ws, P1_TH / 64 - 1);
            const unsigned int ex = wbase + inc - v;
            s_off[tb] = ex;                                                          // (in place: nobody else looks at my bucket's count)
        }
        if (t == 0) added += total;
        lds_barrier();
        // C. records into LDS in bucket order; a tile with more records than the stage holds leaves in several rounds, all but the
        //    last of them copied out right away
        unsigned int r0 = 0;
        auto slice_places = [&]() {
            // where stage index 0 would go in the slice: the run's place has arrived by now
            int tc = t;
            asm volatile("" : "+v"(tc), "+v"(apos));
            const unsigned int ex = s_off[tc];
            const unsigned int v = (tc + 1 < P1_MAXB ? s_off[tc + 1] : total) - ex;      // (buckets past the last one hold the total)
            const uint64_t slice0 = reinterpret_cast<uint64_t>(out1 + ((uint64_t)tc * P.nblk1 + grp) * P.cap1);
            s_base[tc] = slice0 + ((uint64_t)apos - (uint64_t)ex) * 8ull;
            if (v && apos + v > P.cap1) s_wsum[16] = 1;                              // (stays set: the slice stays full)
        };
        if (total <= (unsigned int)P1_STAGE) {                                      // (block-uniform) the usual case: one round holds the tile
            if (total) {
#pragma unroll
                for (int j = 0; j < PT_GROUP; ++j) {
                    const uint32_t b = br[j] >> 16;
                    if (b != 0xFFFFu) {
                        const unsigned int pos = s_off[b] + (br[j] & 0xFFFFu);
                        unsigned int pos2 = pos;
                        asm("" : "+v"(pos2));                                       // (its own shift: derived from pos * 8 it becomes a 64-bit multiply-add)
                        s_stage[pos] = rec[j];
                        s_bkt[pos2] = (unsigned short)b;
                    }
                }
                slice_places();
            }
        } else {
            for (;;) {
#pragma unroll
                for (int j = 0; j < PT_GROUP; ++j) {
                    const uint32_t b = br[j] >> 16;
                    if (b != 0xFFFFu) {
                        const unsigned int pos = s_off[b] + (br[j] & 0xFFFFu) - r0;
                        if (pos < (unsigned int)P1_STAGE) { s_stage[pos] = rec[j]; s_bkt[pos] = (unsigned short)b; }
                    }
                }
                if (r0 == 0) slice_places();
                if (total - r0 <= (unsigned int)P1_STAGE) break;                    // the last round leaves with the next tile's hashing
                lds_barrier();
                copy_out(r0, (unsigned int)P1_STAGE);
                r0 += P1_STAGE;
                lds_barrier();
            }
        }
        pend_r0 = r0;
        pend_nr = total - r0;
        s_nxt[t] = 0;              // the next tile's counters (this copy was last looked at in the tile before this one)
        if (has_next) {
            if (prefetch) {
                asm volatile("" : "+v"(raw.w[0]), "+v"(raw.w[1]), "+v"(raw.w[2]), "+v"(raw.w[3]));     // (not before this point)
                encode16(raw.w, c, iv);
            } else stage16(bases, base0 + (int64_t)PT_TILE + (int64_t)t * PT_GROUP, (int64_t)n, c, iv);      // (the piece's last tile)
            int td = t;
            asm volatile("" : "+v"(td));
            s_code[td + PT_HALO] = c;
            s_inv[td + PT_HALO] = iv;
            if (td < PT_HALO) { s_code[td] = s_wsum[24 + td]; s_inv[td] = s_wsum[28 + td]; }
        }
        lds_barrier();
    }
    if (pend_nr) copy_out(pend_r0, pend_nr);
    if (t == 0 && added) atomicAdd(&P.stats[ST_OCCURRENCES], added);
}
// ---- level 1 for keys that do not fit an 8-byte record (k >= 38) -------------------------------------------------------------------
// The same kernel shape with 16-byte records {low 64 hash bits, the hash bits between them and the bucket bits}.  A thread's
// 16 windows would be 64 + 16 registers of records, so a tile goes through the phases in two HALVES of 8 windows per thread
// (rolling state, codes and validity carry over); a half's ~6.2 K records fit the stage, whose entries are 16 bytes.  The copy-out
// of one half runs inside the hash loop of the next, two steps one iteration apart (7 staged records per thread, 8 iterations).
typedef uint64_t Rec16 __attribute__((ext_vector_type(2)));      // .lo = the low 64 hash bits, .hi = the bits above them (a vector type: stored through global-address-space pointers with one 16-byte store)
constexpr int P1W_STAGE = 6912;         // 16-byte records per staging round
constexpr size_t P1W_LDS = (size_t)P1W_STAGE * 18 + (size_t)(2 * (P1_MAXB + 4) + 32 + 64) * 4 + (size_t)2 * (P1_TH + PT_HALO) * 4 + (size_t)P1_MAXB * 8;
typedef __attribute__((address_space(1))) Rec16 global_rec16;
__device__ __forceinline__ u128 hash_of16(uint64_t b1, Rec16 r, int recbits) { return bor(shl(mk(0, b1), recbits), mk(r.hi, r.lo)); }

template <int NW>          // words of a k-mer: 3 (k <= 48: the bits above the low 64 fit one word), 4 (k <= 64)
__global__ __launch_bounds__(P1_TH) void part1w_kernel(const uint8_t *__restrict__ bases, uint64_t n, uint64_t ntiles, uint64_t emit_from, P1Args P,
                                                       Rec16 *__restrict__ out1, unsigned int *__restrict__ cnt1, unsigned long long *__restrict__ deferred,
                                                       unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    static_assert(NW == 3 || NW == 4, "16-byte records are for keys of 65 .. 128 bits");
    using hi_t = typename std::conditional<NW == 3, uint32_t, uint64_t>::type;
    extern __shared__ __align__(16) unsigned char s_raw[];
    unsigned int *s_cnt2 = reinterpret_cast<unsigned int *>(s_raw);                          // 2 x (P1_MAXB+4): rank counters / offsets of alternate halves
    unsigned int *s_wsum = s_cnt2 + 2 * (P1_MAXB + 4);                                       // 16 wave totals, [16] = "a slice overflows", [24..31] the tile's last 64 bases
    unsigned int *s_dummy = s_wsum + 32;                                                     // 64: what the rank atomics of windows that are no k-mer add to
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_dummy + 64);                           // P1_TH + PT_HALO
    uint32_t *s_inv = s_code + (P1_TH + PT_HALO);
    uint64_t *s_base = reinterpret_cast<uint64_t *>(s_inv + (P1_TH + PT_HALO));              // P1_MAXB: address of stage index 0 in my slice of each list
    unsigned short *s_bkt = reinterpret_cast<unsigned short *>(s_base + P1_MAXB);            // bucket of each staged record
    Rec16 *s_stage = reinterpret_cast<Rec16 *>(s_bkt + P1W_STAGE);                           // P1W_STAGE records, bucket order
    static_assert(((2 * (P1_MAXB + 4) + 32 + 64) * 4 + 2 * (P1_TH + PT_HALO) * 4 + P1_MAXB * 8 + P1W_STAGE * 2) % 16 == 0, "the stage is read and written 16 bytes at a time");
    const int t = threadIdx.x;
    const int k = P.k, p1 = P.p1;
    const int nb = 1 << p1;
    const int topbits = 2 * k - 32 * (NW - 1);                                  // 2..32: bits of the k-mer in its top word
    const uint32_t topmask = topbits >= 32 ? ~0u : ((1u << topbits) - 1u);
    const int rsh = topbits - 2;
    const int hb = 2 * k - 64;                                                  // hash bits above the low 64: 12 .. 64
    const hi_t himask = hb >= (int)(8 * sizeof(hi_t)) ? (hi_t)~(hi_t)0 : (hi_t)(((hi_t)1 << hb) - 1);
    const int bsh = hb - p1;                                                    // the bucket is the top p1 of those bits (p1 <= 10 <= hb)
    const hi_t restmask = bsh >= (int)(8 * sizeof(hi_t)) ? (hi_t)~(hi_t)0 : (hi_t)(((hi_t)1 << bsh) - 1);
    int wlog = 0;
    while ((2 << wlog) <= k) ++wlog;
    unsigned long long added = 0;
    const uint32_t grp = blockIdx.x % P.nblk1;
    s_cnt2[t] = 0;
    s_cnt2[P1_MAXB + 4 + t] = 0;
    for (int i = t; i < P1W_STAGE; i += P1_TH) s_bkt[i] = 0;
    if (t == 0) s_wsum[16] = 0;
    const uint32_t nt32 = (uint32_t)ntiles;
    const uint32_t per_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)((nt32 + gridDim.x - 1) / gridDim.x));
    const uint32_t tile_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * per_block));
    const uint32_t tile_end = tile_first + per_block < nt32 ? tile_first + per_block : nt32;
    uint32_t c = 0, iv = 0xFFFFu;
    if (tile_first < tile_end) {
        uint32_t hc = 0, hiv = 0xFFFFu;
        const int64_t b0 = (int64_t)tile_first * PT_TILE;
        stage16(bases, b0 + (int64_t)t * PT_GROUP, (int64_t)n, c, iv);
        if (t < PT_HALO) stage16(bases, b0 - (int64_t)(PT_HALO - t) * PT_GROUP, (int64_t)n, hc, hiv);
        s_code[t + PT_HALO] = c;
        s_inv[t + PT_HALO] = iv;
        if (t < PT_HALO) { s_code[t] = hc; s_inv[t] = hiv; }
    }
    lds_barrier();
    unsigned int pend_nr = 0, pend_r0 = 0;
    auto copy_out = [&](unsigned int r0, unsigned int nr) {
        for (unsigned int i = t; i < nr; i += P1_TH) {
            const Rec16 rr = s_stage[i];
            const uint32_t b = s_bkt[i];
            if (!s_wsum[16]) reinterpret_cast<global_rec16 *>(s_base[b])[r0 + i] = rr;
            else {                                                              // a slice of mine is full: with the bound
                const uint64_t first = reinterpret_cast<uint64_t>(out1 + ((uint64_t)b * P.nblk1 + grp) * P.cap1);
                const uint64_t pos = (uint64_t)(((int64_t)(s_base[b] - first) >> 4) + (int64_t)(r0 + i));
                if (pos < P.cap1) reinterpret_cast<global_rec16 *>(first)[pos] = rr;
                else defer_record(P.stats, hash_of16((uint64_t)b, rr, P.recbits), deferred, deferred_n, deferred_cap);
            }
        }
    };
    constexpr int HALF = PT_GROUP / 2;
    constexpr int NCO = (P1W_STAGE + P1_TH - 1) / P1_TH;                            // staged records per thread (7)
    static_assert(NCO + 1 <= HALF, "the copy-out pipeline must fit the hash loop of a half");
    uint32_t halfno = 0;                                                            // halves done by this block: which copy of the counters
    for (uint32_t tile = tile_first; tile < tile_end; ++tile) {
        const int64_t base0 = (int64_t)tile * PT_TILE;
        const bool has_next = tile + 1 < tile_end;
        const bool prefetch = has_next && (uint64_t)(base0 + 2 * (int64_t)PT_TILE) <= n;
        int ta = t;
        asm volatile("" : "+v"(ta));
        uint32_t f[NW], r[NW];
        uint32_t vmask;
        {
            const uint32_t w4 = s_code[ta], w3 = s_code[ta + 1], w2 = s_code[ta + 2], w1 = s_code[ta + 3];
            const uint64_t ivprev = ((uint64_t)s_inv[ta] << 48) | ((uint64_t)s_inv[ta + 1] << 32) | ((uint64_t)s_inv[ta + 2] << 16) | (uint64_t)s_inv[ta + 3];
            const u128 fwd0 = band(mk(((uint64_t)w4 << 32) | w3, ((uint64_t)w2 << 32) | w1), maskbits(2 * k));
            const u128 rc0 = revcomp(fwd0, k);
            const uint32_t fw[4] = {(uint32_t)fwd0.lo, (uint32_t)(fwd0.lo >> 32), (uint32_t)fwd0.hi, (uint32_t)(fwd0.hi >> 32)};
            const uint32_t rw[4] = {(uint32_t)rc0.lo, (uint32_t)(rc0.lo >> 32), (uint32_t)rc0.hi, (uint32_t)(rc0.hi >> 32)};
#pragma unroll
            for (int w = 0; w < NW; ++w) { f[w] = fw[w]; r[w] = rw[w]; }
            // bit q of z: my base 15 - q (q < 16), or the base q - 15 positions before my first (k <= 64: bits up to 15 + 63 matter)
            const unsigned __int128 z = (unsigned __int128)iv | ((unsigned __int128)ivprev << 16);
            unsigned __int128 sm = z;
#pragma unroll
            for (int b = 0; b < 6; ++b) if (b < wlog) sm |= sm >> (1 << b);
            sm |= sm >> (k - (1 << wlog));
            vmask = ~(uint32_t)sm & 0xFFFFu;
        }
        if ((uint64_t)base0 < emit_from) {
            const int64_t mine = base0 + (int64_t)ta * PT_GROUP;
            const int jfirst = (int64_t)emit_from > mine ? (int)((int64_t)emit_from - mine < PT_GROUP ? (int64_t)emit_from - mine : PT_GROUP) : 0;
            vmask &= 0xFFFFu >> jfirst;
        }
        Raw16 raw;
        for (int half = 0; half < 2; ++half, ++halfno) {
            unsigned int *s_cnt = s_cnt2 + (halfno & 1u) * (P1_MAXB + 4);
            unsigned int *s_off = s_cnt;
            unsigned int *s_nxt = s_cnt2 + ((halfno & 1u) ^ 1u) * (P1_MAXB + 4);
            if (pend_nr && s_wsum[16]) {                                            // (block-uniform)
                copy_out(pend_r0, pend_nr);
                pend_nr = 0;
            }
            uint64_t rlo[HALF];
            hi_t rhi[HALF];
            uint32_t br[HALF];
            uint32_t bprev = 0xFFFFu, rprev = 0;
            uint32_t co_b[NCO];
#pragma unroll
            for (int jj = 0; jj < HALF; ++jj) {
                const int j = half * HALF + jj;                                     // (uniform; the two halves are the same code)
                if (jj < NCO) {
                    unsigned int i = (unsigned int)ta + (unsigned int)jj * P1_TH;
                    if ((jj + 1) * P1_TH > P1W_STAGE) i = i < (unsigned int)P1W_STAGE ? i : (unsigned int)P1W_STAGE - 1u;
                    co_b[jj] = s_bkt[i];
                }
                const uint32_t cj = (c >> (30 - 2 * j)) & 3u;
#pragma unroll
                for (int w = NW - 1; w > 0; --w) f[w] = __builtin_amdgcn_alignbit(f[w], f[w - 1], 30);
                f[0] = (f[0] << 2) | cj;
                f[NW - 1] &= topmask;
#pragma unroll
                for (int w = 0; w < NW - 1; ++w) r[w] = __builtin_amdgcn_alignbit(r[w + 1], r[w], 2);
                r[NW - 1] = (r[NW - 1] >> 2) | ((cj ^ 3u) << rsh);
                const bool valid = (vmask >> (15 - j)) & 1u;
                // canonical = numeric min of the two strands, compared from the top word down
                bool rc_less = false, decided = false;
#pragma unroll
                for (int w = NW - 1; w >= 0; --w) {
                    if (!decided && r[w] != f[w]) { rc_less = r[w] < f[w]; decided = true; }
                }
                uint32_t m[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) m[w] = rc_less ? r[w] : f[w];
                const uint64_t lo = mix64(((uint64_t)m[1] << 32) | m[0]);
                hi_t hi;
                if constexpr (NW == 3) hi = (m[2] ^ __builtin_amdgcn_alignbit((uint32_t)(lo >> 32), (uint32_t)lo, 30)) & himask;
                else hi = ((((uint64_t)m[3] << 32) | m[2]) ^ rotr64(lo, 30)) & himask;
                const uint32_t b = (uint32_t)(hi >> bsh);
                rlo[jj] = lo;
                rhi[jj] = hi & restmask;
                if (jj > 0) br[jj - 1] = (bprev << 16) | rprev;
                bprev = valid ? b : 0xFFFFu;
                rprev = atomicAdd(valid ? &s_cnt[b] : &s_dummy[ta & 63], 1u);
                if (jj >= 1 && jj - 1 < NCO) {
                    unsigned int i = (unsigned int)ta + (unsigned int)(jj - 1) * P1_TH;
                    const unsigned int ic = jj * P1_TH > P1W_STAGE ? (i < (unsigned int)P1W_STAGE ? i : (unsigned int)P1W_STAGE - 1u) : i;
                    const Rec16 rr = s_stage[ic];
                    const uint64_t bs = s_base[co_b[jj - 1]];
                    if (i < pend_nr) reinterpret_cast<global_rec16 *>(bs)[pend_r0 + i] = rr;
                }
            }
            br[HALF - 1] = (bprev << 16) | rprev;
            pend_nr = 0;
            if (half == 1) {
                // the next tile's text: asked for behind the copy-out stores, looked at after the staging
                int tb = t;
                asm volatile("" : "+v"(tb));
                struct __attribute__((packed, aligned(1))) V16 { uint32_t w[4]; };
                const V16 v = *reinterpret_cast<const V16 *>(bases + (prefetch ? base0 + (int64_t)PT_TILE + (int64_t)tb * PT_GROUP : (int64_t)0));
                raw.w[0] = v.w[0]; raw.w[1] = v.w[1]; raw.w[2] = v.w[2]; raw.w[3] = v.w[3];
                if (has_next && tb < PT_HALO) { s_wsum[24 + tb] = s_code[P1_TH + tb]; s_wsum[28 + tb] = s_inv[P1_TH + tb]; }
            }
            lds_barrier();
            // B. exclusive prefix of the bucket counts, the run's place in the slice
            unsigned int total;
            unsigned int apos = 0;
            {
                int tb = t;
                asm volatile("" : "+v"(tb));
                const unsigned int v = tb < nb ? s_cnt[tb] : 0u;
                if (v) apos = __hip_atomic_fetch_add(&cnt1[((uint32_t)grp << p1) + (uint32_t)tb], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int inc = wave_scan_incl(v);
                if ((tb & 63) == 63) s_wsum[tb >> 6] = inc;
                lds_barrier();
                const unsigned int ws = wave_scan_incl((tb & 63) < P1_TH / 64 ? s_wsum[tb & 63] : 0u);
                const unsigned int wbase = (tb >> 6) ? (unsigned int)__builtin_amdgcn_readlane((int)ws, (tb >> 6) - 1) : 0u;
                total = (unsigned int)__builtin_amdgcn_readlane((int)ws, P1_TH / 64 - 1);
                s_off[tb] = wbase + inc - v;
            }
            if (t == 0) added += total;
            lds_barrier();
            // C. records into LDS in bucket order (several rounds when the half holds more than the stage: text without read boundaries)
            unsigned int r0 = 0;
            for (;;) {
#pragma unroll
                for (int jj = 0; jj < HALF; ++jj) {
                    const uint32_t b = br[jj] >> 16;
                    if (b != 0xFFFFu) {
                        const unsigned int pos = s_off[b] + (br[jj] & 0xFFFFu) - r0;
                        if (pos < (unsigned int)P1W_STAGE) {
                            Rec16 rr;
                            rr.lo = rlo[jj]; rr.hi = (uint64_t)rhi[jj];
                            s_stage[pos] = rr;
                            s_bkt[pos] = (unsigned short)b;
                        }
                    }
                }
                if (r0 == 0 && total) {
                    int tc = t;
                    asm volatile("" : "+v"(tc), "+v"(apos));
                    const unsigned int ex = s_off[tc];
                    const unsigned int v = (tc + 1 < P1_MAXB ? s_off[tc + 1] : total) - ex;
                    const uint64_t slice0 = reinterpret_cast<uint64_t>(out1 + ((uint64_t)tc * P.nblk1 + grp) * P.cap1);
                    s_base[tc] = slice0 + ((uint64_t)apos - (uint64_t)ex) * 16ull;
                    if (v && apos + v > P.cap1) s_wsum[16] = 1;
                }
                if (total - r0 <= (unsigned int)P1W_STAGE || !total) break;
                lds_barrier();
                copy_out(r0, (unsigned int)P1W_STAGE);
                r0 += P1W_STAGE;
                lds_barrier();
            }
            pend_r0 = r0;
            pend_nr = total - r0;
            s_nxt[t] = 0;
            if (half == 1 && has_next) {
                if (prefetch) {
                    asm volatile("" : "+v"(raw.w[0]), "+v"(raw.w[1]), "+v"(raw.w[2]), "+v"(raw.w[3]));
                    encode16(raw.w, c, iv);
                } else {                                                              // (the piece's last tile; through an opaque copy of the
                    int te = t;                                                       //  thread number: hoisted out of the tile loop the byte
                    asm volatile("" : "+v"(te));                                      //  addresses of the slow path end up in scratch memory)
                    stage16(bases, base0 + (int64_t)PT_TILE + (int64_t)te * PT_GROUP, (int64_t)n, c, iv);
                }
                int td = t;
                asm volatile("" : "+v"(td));
                s_code[td + PT_HALO] = c;
                s_inv[td + PT_HALO] = iv;
                if (td < PT_HALO) { s_code[td] = s_wsum[24 + td]; s_inv[td] = s_wsum[28 + td]; }
            }
            lds_barrier();
        }
    }
    if (pend_nr) copy_out(pend_r0, pend_nr);
    if (t == 0 && added) atomicAdd(&P.stats[ST_OCCURRENCES], added);
}

// the fill counts were the slices' cursors: one that ran past its slice's end becomes "full" (what did not fit was deferred)
__global__ __launch_bounds__(256) void clamp_counts_kernel(unsigned int *__restrict__ cnt, uint32_t n, uint32_t cap) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && cnt[i] > cap) cnt[i] = cap;
}
// (a table so small that one level of lists is enough: region_insert_kernel reads a region's slice counts side by side)
__global__ __launch_bounds__(256) void transpose_counts_kernel(const unsigned int *__restrict__ in, unsigned int *__restrict__ out, uint32_t nb1, uint32_t nblk1) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nb1 * nblk1) out[(i % nb1) * nblk1 + i / nb1] = in[i];
}
// one launch site for the three word counts: k <= 16, 17..32, 33..37
static hipError_t launch_part1(hipStream_t stream, int k, const uint8_t *d_piece, uint64_t len, uint64_t ntiles, uint64_t emit_from, const TableDev &d, const PartGeom &G,
                               uint64_t *out1, unsigned int *cnt1, unsigned long long *defer_e, unsigned long long *defer_n, uint64_t deferred_cap) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1_kernel<3, 37>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        attr_set = true;
    }
    P1Args P;
    P.k = k; P.p1 = G.p1; P.recbits = G.recbits; P.nblk1 = G.nblk1; P.cap1 = G.cap1; P.stats = d.stats;
    const uint32_t n_cnt1 = (1u << G.p1) * G.nblk1;
    {
        hipError_t e = hipMemsetAsync(cnt1, 0, (size_t)n_cnt1 * 4, stream);
        if (e != hipSuccess) return e;
    }
#define JK_P1_LAUNCH(...) hipLaunchKernelGGL((part1_kernel<__VA_ARGS__>), dim3(G.grid1), dim3(P1_TH), P1_LDS, stream, d_piece, len, ntiles, emit_from, P, out1, cnt1, defer_e, defer_n, deferred_cap)
    if (k == 37 && G.p1 == 10 && G.recbits == 64 && !getenv("JASPER_EXPERIMENT_NO_KFIX")) JK_P1_LAUNCH(3, 37);
    else if (G.recbits > 64) {                // 16-byte records (k >= 38)
        static bool attrw_set = false;
        if (!attrw_set) {
            hipError_t e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1w_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(part1w_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            attrw_set = true;
        }
        if (k <= 48) hipLaunchKernelGGL((part1w_kernel<3>), dim3(G.grid1), dim3(P1_TH), P1W_LDS, stream, d_piece, len, ntiles, emit_from, P, reinterpret_cast<Rec16 *>(out1), cnt1, defer_e, defer_n, deferred_cap);
        else hipLaunchKernelGGL((part1w_kernel<4>), dim3(G.grid1), dim3(P1_TH), P1W_LDS, stream, d_piece, len, ntiles, emit_from, P, reinterpret_cast<Rec16 *>(out1), cnt1, defer_e, defer_n, deferred_cap);
    }
    else if (k <= 16) JK_P1_LAUNCH(1);
    else if (k <= 32) JK_P1_LAUNCH(2);
    else JK_P1_LAUNCH(3);                     // (k <= 37: 8-byte records while 2k - 64 <= p1 <= 10)
#undef JK_P1_LAUNCH
    hipLaunchKernelGGL(clamp_counts_kernel, dim3((n_cnt1 + 255) / 256), dim3(256), 0, stream, cnt1, n_cnt1, G.cap1);
    return hipGetLastError();
}

// ---- level 2: every bucket list -> 2^p2 region lists ---------------------------------------------------------
// grid (nblk2, buckets): block (x, b1) reads the level-1 slices x, x+nblk2, ... of bucket b1 (as one concatenated
// list) and appends to ITS slice of each of the bucket's 2^p2 region lists.  Same shape as part1: a tile of 16 K records
// is sorted by region inside LDS (rank from a returning LDS atomic, offsets from a block scan) and leaves in region
// order, so a wave writes whole runs of one list.  (The first version appended each record straight from the lane that
// loaded it, with wave-level "match any" ballots to share the cursor atomics: ~100 instructions per record, and every
// store instruction touched ~64 different lines -- the kernel was instruction-bound at 2.3 TB/s.)
constexpr int P2_MAXSL = 512;          // level-1 slices per bucket (= nblk1 <= 512)
// OWN (the multi-GPU exchange, count_exchange below): a bucket is split 2^p2 * nown ways, by (owner_of(hash), next p2
// hash bits), and the lists are laid out owner-major -- list ((o * 2^p1 + b1) << p2) + b2 -- so that everything owner o
// is to receive is ONE contiguous block of out2 / cnt2: the region lists of o's own table, ready for region_insert_kernel<., XCHG>.
// MIN (the owner's extra pass when the senders could not split finely enough, count_exchange below): the input slices of a
// bucket come from nsrc senders, each of which laid out ITS lists of all buckets as one block -- slice j of bucket b =
// sender j / (nblk1/nsrc), its slice j % (nblk1/nsrc): list index ((sender * buckets + b) * (nblk1/nsrc) + that).
template <bool OWN, bool MIN = false>
__global__ __launch_bounds__(PT_THREADS) void part2_kernel(const uint64_t *__restrict__ out1, const unsigned int *__restrict__ cnt1, TableDev T,
                                                            PartGeom G, uint64_t *__restrict__ out2, unsigned int *__restrict__ cnt2,
                                                            unsigned long long *__restrict__ deferred, unsigned long long *__restrict__ deferred_n,
                                                            uint64_t deferred_cap, uint32_t nown, int cbits = 0) {
    // cbits > 0 (MIN only: lists deduplicated by their senders): the bits of a record right above G.recbits -- implied by the
    // input list it is in -- hold (occurrences - 1); they travel on untouched, only a record that finds no room is taken apart
    extern __shared__ __align__(16) unsigned char s_raw[];
    uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_raw);                               // PT_TILE records
    unsigned int *s_cur = reinterpret_cast<unsigned int *>(s_raw + (size_t)PT_TILE * 8);   // PT_MAXBUCKETS slice cursors (per b1)
    unsigned int *s_cnt = s_cur + PT_MAXBUCKETS;                                           // PT_MAXBUCKETS records of this tile per region
    unsigned int *s_off = s_cnt + PT_MAXBUCKETS;                                           // PT_MAXBUCKETS+1 exclusive prefix of s_cnt
    unsigned int *s_wsum = s_off + PT_MAXBUCKETS + 1;                                      // 16 wave totals
    unsigned int *s_pref = s_wsum + 16;                                                    // P2_MAXSL+1 prefix of my slices' lengths
    const int t = threadIdx.x;
    const int nb2 = OWN ? (int)(nown << G.p2) : 1 << G.p2;
    const int shift2 = G.recbits - G.p2;               // the p2 bits right below the level-1 bucket bits
    const uint32_t nmine = (G.nblk1 - blockIdx.x + G.nblk2 - 1) / G.nblk2;                 // slices x, x+nblk2, ... < nblk1
    const uint32_t in_per_src = MIN ? G.nblk1 / nown : G.nblk1;                            // (MIN: nown = the number of senders)
    auto in_list = [&](uint32_t b1, uint32_t j) -> uint64_t {                              // list index of input slice j of bucket b1
        if constexpr (MIN) return ((uint64_t)(j / in_per_src) * (1ull << G.p1) + b1) * in_per_src + j % in_per_src;
        else return (uint64_t)b1 * G.nblk1 + j;
    };
    auto in_cnt = [&](uint32_t b1, uint32_t j) -> uint64_t {                               // where its fill count is (part1's counts are slice-major)
        if constexpr (MIN) return in_list(b1, j);
        else return ((uint64_t)j << G.p1) + b1;
    };
    for (uint32_t b1 = blockIdx.y; b1 < (1u << G.p1); b1 += gridDim.y) {
        for (int i = t; i < nb2; i += PT_THREADS) { s_cur[i] = 0; s_cnt[i] = 0; }
        if (t < 64) {                                  // exclusive prefix of my slices' lengths (wave 0)
            unsigned int carry = 0;
            for (uint32_t x0 = 0; x0 < nmine; x0 += 64) {
                const uint32_t x = x0 + t;
                const unsigned int v = x < nmine ? cnt1[in_cnt(b1, blockIdx.x + x * G.nblk2)] : 0u;
                unsigned int inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if (t >= o) inc += u; }
                if (x < nmine) s_pref[x] = carry + inc - v;
                carry += __shfl(inc, 63);
            }
            if (t == 0) s_pref[nmine] = carry;
        }
        lds_barrier();
        const uint32_t total = s_pref[nmine];
        const uint64_t *src0 = out1 + ((uint64_t)b1 * G.nblk1 + blockIdx.x) * G.cap1;      // slice x; slice x + j*nblk2 is j*nblk2*cap1 further (not MIN)
        for (uint32_t tile0 = 0; tile0 < total; tile0 += PT_TILE) {
            // A. my 16 records (all loads in flight), their region, a rank in the tile's region histogram
            uint64_t rec[PT_GROUP];
            uint32_t br[PT_GROUP];
            uint32_t sl;                                                      // slice of my first record: largest x with s_pref[x] <= i
            {
                const uint32_t i = tile0 + t;
                uint32_t lo = 0, hi = nmine - 1;
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= i) lo = mid; else hi = mid - 1; }
                sl = lo;
            }
#pragma unroll
            for (int j = 0; j < PT_GROUP; ++j) {
                const uint32_t i = tile0 + (uint32_t)j * PT_THREADS + t;
                rec[j] = 0ull;
                br[j] = 0xFFFFFFFFu;
                if (i < total) {
                    while (s_pref[sl + 1] <= i) ++sl;                         // my records are 1024 apart: the slice moves on slowly
                    if constexpr (MIN) rec[j] = out1[in_list(b1, blockIdx.x + sl * G.nblk2) * G.cap1 + (i - s_pref[sl])];
                    else rec[j] = src0[(uint64_t)sl * G.nblk2 * G.cap1 + (i - s_pref[sl])];
                    br[j] = sl;                                               // (reused below)
                }
            }
#pragma unroll
            for (int j = 0; j < PT_GROUP; ++j) {
                if (br[j] == 0xFFFFFFFFu) continue;
                uint32_t b2;
                if constexpr (OWN) {
                    b2 = G.p2 ? (uint32_t)(rec[j] >> shift2) & ((1u << G.p2) - 1u) : 0u;
                    b2 |= owner_of(hash_of(b1, rec[j], G.recbits), nown) << G.p2;
                } else b2 = (uint32_t)(rec[j] >> shift2) & (uint32_t)(nb2 - 1);
                br[j] = (b2 << 16) | (atomicAdd(&s_cnt[b2], 1u) & 0xFFFFu);   // a tile holds 2^14 records
            }
            lds_barrier();
            // B. exclusive prefix of the region counts: thread t owns regions 2t and 2t+1 (nb2 <= 2048)
            {
                const unsigned int v0 = 2 * t < nb2 ? s_cnt[2 * t] : 0u, v1 = 2 * t + 1 < nb2 ? s_cnt[2 * t + 1] : 0u;
                const unsigned int v = v0 + v1;
                unsigned int inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
                if ((t & 63) == 63) s_wsum[t >> 6] = inc;
                lds_barrier();
                unsigned int wbase = 0;
                for (int w = 0; w < (t >> 6); ++w) wbase += s_wsum[w];
                const unsigned int ex = wbase + inc - v;
                if (2 * t < nb2) s_off[2 * t] = ex;
                if (2 * t + 1 < nb2) s_off[2 * t + 1] = ex + v0;
                if (t == PT_THREADS - 1) s_off[nb2] = wbase + inc;
            }
            lds_barrier();
            // C. records into LDS in region order
#pragma unroll
            for (int j = 0; j < PT_GROUP; ++j)
                if (br[j] != 0xFFFFFFFFu) s_stage[s_off[br[j] >> 16] + (br[j] & 0xFFFFu)] = rec[j];
            lds_barrier();
            // D. copy out: `lpb` lanes per region (a tile holds PT_TILE / nb2 records per region on average)
            {
                const int lpb = nb2 <= 256 ? 64 : 16;
                const int g = t / lpb, r = t % lpb;
                for (int b2 = g; b2 < nb2; b2 += PT_THREADS / lpb) {
                    const unsigned int off = s_off[b2], cnt = s_cnt[b2], cur = s_cur[b2];
                    const uint64_t region = OWN ? ((((uint64_t)(b2 >> G.p2) << G.p1) + b1) << G.p2) + (uint64_t)(b2 & ((1 << G.p2) - 1))
                                                : ((uint64_t)b1 << G.p2) + (uint64_t)b2;
                    uint64_t *dst = out2 + (region * G.nblk2 + blockIdx.x) * G.cap2;
                    for (unsigned int q = r; q < cnt; q += lpb) {
                        const unsigned int pos = cur + q;
                        const uint64_t rr = s_stage[off + q];
                        if (pos < G.cap2) dst[pos] = rr;
                        else if (MIN && cbits) {
                            const u128 hh = hash_of(b1, rr & ((1ull << G.recbits) - 1ull), G.recbits);
                            const unsigned long long di = atomicAdd(deferred_n, 1ull);
                            if (di < deferred_cap) { deferred[3 * di] = hh.hi; deferred[3 * di + 1] = hh.lo; deferred[3 * di + 2] = ((rr >> G.recbits) & ((1ull << cbits) - 1ull)) + 1ull; }
                            else atomicExch(&T.stats[ST_FATAL], 1ull);
                        } else defer_record(T, hash_of(b1, rr, G.recbits), deferred, deferred_n, deferred_cap);
                    }
                }
            }
            lds_barrier();
            for (int i = t; i < nb2; i += PT_THREADS) { s_cur[i] += s_cnt[i]; s_cnt[i] = 0; }
            lds_barrier();
        }
        for (int i = t; i < nb2; i += PT_THREADS) {
            const uint64_t region = OWN ? ((((uint64_t)(i >> G.p2) << G.p1) + b1) << G.p2) + (uint64_t)(i & ((1 << G.p2) - 1)) : ((uint64_t)b1 << G.p2) + (uint64_t)i;
            cnt2[region * G.nblk2 + blockIdx.x] = s_cur[i] < G.cap2 ? s_cur[i] : G.cap2;
        }
        lds_barrier();
    }
}
constexpr size_t P2_LDS = (size_t)PT_TILE * 8 + (size_t)(3 * PT_MAXBUCKETS + 1 + 16 + P2_MAXSL + 1) * 4;

// ---- level 2, single-GPU form: whole 128-byte lines only ---------------------------------------------------------------------
// The same job as part2_kernel<false> for the plain case with at most MAXB lists per bucket.  Two things differ:
//  * A list's slice is only ever written in whole, aligned 128-byte lines.  What a round leaves over of a list (< 16 records)
//    waits in LDS (s_carry) and leaves in front of the next round's records; the last round's rest closes the slice.  Measured on
//    part1's copy-out: runs that begin and end inside a line cost twice -- the line is written by two rounds, and
//    between the two the half-written line has to survive in an L2 that the open lines of all blocks fill completely.
//  * few instructions per record: a WAVE streams whole input slices (slice, position and length are scalars: a record's address
//    is one add), a row that does not exist counts as a record of one extra list that sorts behind all others (no per-record
//    branch), the copy-out is one lane per staged record, and the loads of the next round are in flight during this one --
//    waited for before this round's copy-out stores are issued (see part1_kernel).
// Instances: ONE workgroup per CU, with the next round's records asked for a round ahead.  <128, 12>: at most 128 lists per
// bucket (tables up to 2^29 slots), rounds of 12 rows of 1024 records -- 96 KB of stage + 16 KB of carry, 120 registers.  (Until
// round 4 this was <128, 7> with TWO workgroups per CU, one's loads and stores under the other's LDS work, 64 registers per lane
// and no room for a prefetch: 3.71 ms against 3.53 on the 47 Mb workload; 14 rows: 3.50 with 24 bytes per lane spilled.)
// <512, 10>: up to 512 lists (tables of 2^30 .. 2^32 slots, and every shard of the larger configurations): the carry alone is
// 64 KB, next to an 80-KB stage (10 rows; 8 rows: 12.1-12.4 ms against 11.7 on the 140 Mb workload); a round brings a list 16 records on average, i.e. about one line leaves per list and round and
// nearly every record passes through the carry (LDS traffic, which this kernel has to spare).
constexpr int P2F_LINE = 16;           // records per 128-byte line
// per list and round: stage record i leaves to gbase + 8 i if i < lim, else waits in carry slot i + cadd
struct P2Meta { uint64_t gbase; uint32_t lim; int32_t cadd; };
__device__ __forceinline__ u128 rec_hash(uint64_t b1, uint64_t r, int recbits) { return hash_of(b1, r, recbits); }
__device__ __forceinline__ u128 rec_hash(uint64_t b1, Rec16 r, int recbits) { return hash_of16(b1, r, recbits); }
template <int MAXB, int ROWS, int RB = 8> struct P2F {
    static constexpr int TILE = PT_THREADS * ROWS;
    static constexpr int PIECE = 64 * ROWS * 9;          // records per input piece: nine full wave rounds (<128, 12>: 6912 records, 432 lines)
    static constexpr size_t LDS = (size_t)TILE * RB + (size_t)(3 * (MAXB + 4) + 32) * 4 + (size_t)MAXB * sizeof(P2Meta) + (size_t)MAXB * 128;
};
struct P2Args {
    int p1, p2, recbits;
    uint32_t nblk1, vper, nblk2, cap1, cap2;     // vper = pieces per level-1 slice; cap2: a multiple of P2F_LINE
    uint32_t nown;                               // OWN: the number of key owners
    unsigned long long *stats;
};
// OWN (the multi-GPU exchange, as part2_kernel<true>): a bucket is split nown x 2^p2 ways, by (owner of the key, next p2 hash bits),
// and the lists are laid out owner-major.
// REC: uint64_t, or Rec16 for keys that do not fit 8-byte records (k >= 38; a line is then 8 records).
template <int MAXB, int ROWS, bool OWN = false, typename REC = uint64_t>
__global__ __launch_bounds__(PT_THREADS, 4) void part2f_kernel(const REC *__restrict__ out1, const unsigned int *__restrict__ cnt1, P2Args P, REC *__restrict__ out2,
                                                            unsigned int *__restrict__ cnt2, unsigned long long *__restrict__ deferred,
                                                            unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    constexpr int P2F_MAXB = MAXB, P2F_ROWS = ROWS, P2F_PIECE = P2F<MAXB, ROWS>::PIECE;        // (P2F_TILE = 1024 x ROWS records per round)
    constexpr bool W16 = sizeof(REC) == 16;
    constexpr int RB = (int)sizeof(REC), P2F_LINE = 128 / RB;                  // (shadows the 8-byte constant)
    typedef __attribute__((address_space(1))) REC global_rec;
    constexpr int NSW = MAXB / 64;                       // waves that scan the list counts
    extern __shared__ __align__(16) unsigned char s_raw[];
    P2Meta *s_meta = reinterpret_cast<P2Meta *>(s_raw);                                      // P2F_MAXB
    REC *s_carry = reinterpret_cast<REC *>(s_meta + P2F_MAXB);                      // P2F_MAXB x P2F_LINE: what a list has waiting
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_carry + P2F_MAXB * P2F_LINE);    // P2F_MAXB+4  records of this round per list ([nb2] = the padding)
    unsigned int *s_off = s_cnt + P2F_MAXB + 4;                                              // P2F_MAXB+4  exclusive prefix of s_cnt
    unsigned int *s_have = s_off + P2F_MAXB + 4;                                             // P2F_MAXB+4  records waiting per list, bit 31: they leave this round
    unsigned int *s_wsum = s_have + P2F_MAXB + 4;                                            // [0 .. NSW) wave totals, [16] = "a slice overflows", [17] = rounds
    REC *s_stage = reinterpret_cast<REC *>(s_wsum + 32);                           // P2F_TILE records, list order
    const int t = threadIdx.x;
    const int nb2 = OWN ? (int)(P.nown << P.p2) : 1 << P.p2;
    const int shift2 = P.recbits - P.p2;               // the p2 bits right below the level-1 bucket bits
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6), lane = (uint32_t)t & 63u;
    for (uint32_t b1 = blockIdx.y; b1 < (1u << P.p1); b1 += gridDim.y) {
        auto list_of = [&](uint32_t b2) -> uint64_t {                                        // the list's index in out2 / cnt2
            if constexpr (OWN) return ((((uint64_t)(b2 >> P.p2) << P.p1) + b1) << P.p2) + (uint64_t)(b2 & ((1u << P.p2) - 1u));
            else return ((uint64_t)b1 << P.p2) + (uint64_t)b2;
        };
        auto b2_of = [&](REC rr) -> uint32_t {                                               // the list of a record of this bucket
            if constexpr (W16) return (uint32_t)shr(mk(rr.hi, rr.lo), (unsigned)shift2).lo & (uint32_t)(nb2 - 1);
            else if constexpr (OWN) {
                const uint32_t lo = P.p2 ? (uint32_t)(rr >> shift2) & ((1u << P.p2) - 1u) : 0u;
                return lo | (owner_of(hash_of(b1, rr, P.recbits), P.nown) << P.p2);
            } else return (uint32_t)(rr >> shift2) & (uint32_t)(nb2 - 1);
        };
        auto slice_of = [&](uint32_t b2) { return reinterpret_cast<uint64_t>(out2 + (list_of(b2) * P.nblk2 + blockIdx.x) * P.cap2); };
        // The bucket's few long slices are read in PIECES of P2F_PIECE records (P.vper pieces per slice, by its capacity);
        // rounds = the most any wave needs
        if (t < P2F_MAXB + 4) s_cnt[t] = 0;
        if (t == 0) { s_wsum[16] = 0; s_wsum[17] = 0; }
        static_assert(P2F_MAXB + 4 <= PT_THREADS && NSW >= 1 && NSW <= 16, "list counts are scanned by the first MAXB / 64 waves");
        lds_barrier();
        // block x reads the level-1 slices x, x + nblk2, ... WHOLE (whatever their fill: the blocks' shares are as even as the
        // slices are), each in pieces; my wave takes the block's pieces wave, wave + 16, ... -- piece q of the block = piece
        // q % vper of slice x + nblk2 * (q / vper)
        auto piece_len = [&](uint32_t ph_, uint32_t sub_) -> uint32_t {                      // records of piece sub_ of slice ph_ (ph_ < nblk1)
            const uint32_t c = cnt1[((uint64_t)ph_ << P.p1) + b1], first = sub_ * (uint32_t)P2F_PIECE;
            return c > first ? (c - first < (uint32_t)P2F_PIECE ? c - first : (uint32_t)P2F_PIECE) : 0u;
        };
        {
            uint32_t rounds = 0;
            for (uint32_t q = wave + 16u * lane;; q += 16u * 64u) {
                const uint32_t ph_ = blockIdx.x + P.nblk2 * (q / P.vper);
                if (ph_ >= P.nblk1) break;
                rounds += (piece_len(ph_, q % P.vper) + 64u * P2F_ROWS - 1u) / (64u * P2F_ROWS);
            }
            for (int o = 32; o > 0; o >>= 1) rounds += __shfl_xor(rounds, o);
            if (lane == 0) atomicMax(&s_wsum[17], rounds);
        }
        lds_barrier();
        const uint32_t rounds = s_wsum[17];
        uint32_t pos = 0;                                                                    // (scalar) position in the current piece
        uint32_t ph = blockIdx.x + P.nblk2 * (wave / P.vper), sub = wave % P.vper;           // (scalar) my wave's current piece: `sub` of slice `ph`
        auto open_piece = [&](uint32_t &len_out, const REC *&src_out) {
            len_out = 0;
            if (ph < P.nblk1) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt1[((uint64_t)ph << P.p1) + b1]), first = sub * (uint32_t)P2F_PIECE;
                len_out = c > first ? (c - first < (uint32_t)P2F_PIECE ? c - first : (uint32_t)P2F_PIECE) : 0u;
            }
            src_out = out1 + ((uint64_t)b1 * P.nblk1 + ph) * P.cap1 + (uint64_t)sub * P2F_PIECE;
        };
        uint32_t slen;
        const REC *src;
        open_piece(slen, src);
        unsigned int cur = 0, have = 0;                    // thread t < nb2, list t: records written to its slice so far (whole lines) / waiting in s_carry
        REC rec[P2F_ROWS];
        auto fetch = [&](REC (&dst)[P2F_ROWS], uint32_t &valid) {                       // up to 64 x P2F_ROWS records of my wave's stream; valid = rows that exist (bit j: row j)
            while (pos >= slen && ph < P.nblk1) {                                            // (scalar) next piece
                sub += 16u;
                while (sub >= P.vper) { sub -= P.vper; ph += P.nblk2; }
                pos = 0;
                open_piece(slen, src);
            }
            valid = 0;
#pragma unroll
            for (int j = 0; j < P2F_ROWS; ++j) {
                const uint32_t i = pos + (uint32_t)j * 64u + lane;
                if constexpr (W16) { dst[j].lo = ~0ull; dst[j].hi = ~0ull; } else dst[j] = ~0ull;
                if (i < slen) { dst[j] = src[i]; valid |= 1u << j; }
            }
            pos += 64u * P2F_ROWS;
        };
        uint32_t vmask = 0;
        // PF: the records of round r + 1 are asked for at the top of round r and waited for right BEFORE round r's copy-out stores
        // are issued (a wave's vector-memory operations retire in order: waited for at the top of round r + 1 they would wait for
        // those stores as well), see part1_kernel.  (A switch: the two-workgroups-per-CU instance of round 3 had no registers for it.)
        constexpr bool PF = true;
        REC nxt[P2F_ROWS];
        uint32_t vmask_n = 0;
        if (PF && rounds) {
            fetch(rec, vmask);
#pragma unroll
            for (int j = 0; j < P2F_ROWS; ++j) { if constexpr (W16) asm volatile("" : "+v"(rec[j].lo), "+v"(rec[j].hi)); else asm volatile("" : "+v"(rec[j])); }   // (arrived before the loop: no wait for "all loads" at its top)
        }
        for (uint32_t round = 0; round < rounds; ++round) {
            if constexpr (PF) { if (round + 1 < rounds) fetch(nxt, vmask_n); else vmask_n = 0; }
            else fetch(rec, vmask);
            // A. the list of each record, a rank in the round's list histogram (a row that does not exist: the padding list nb2)
            uint32_t br[P2F_ROWS];
#pragma unroll
            for (int j = 0; j < P2F_ROWS; ++j) {
                const uint32_t b2 = (vmask >> j) & 1u ? b2_of(rec[j]) : (uint32_t)nb2;
                br[j] = (b2 << 16) | (atomicAdd(&s_cnt[b2], 1u) & 0xFFFFu);                  // LDS returning atomic; a round holds 2^14 rows
            }
            lds_barrier();
            // B. exclusive prefix of the list counts: thread t owns list t (the first MAXB / 64 waves)
            unsigned int v = 0, inc = 0;
            if (t < P2F_MAXB) {
                v = t < nb2 ? s_cnt[t] : 0u;
                inc = wave_scan_incl(v);
                if (lane == 63) s_wsum[wave] = inc;
                s_cnt[t] = 0;
                if (t == 0) s_cnt[nb2] = 0;
            }
            lds_barrier();
            unsigned int total, wbase;
            if constexpr (NSW == 2) { total = s_wsum[0] + s_wsum[1]; wbase = wave ? s_wsum[0] : 0u; }
            else {                                                                           // the wave totals scanned once more by every wave
                const unsigned int ws = wave_scan_incl(lane < (uint32_t)NSW ? s_wsum[lane] : 0u);
                wbase = wave && wave < (uint32_t)NSW ? (unsigned int)__builtin_amdgcn_readlane((int)ws, (int)wave - 1) : 0u;
                total = (unsigned int)__builtin_amdgcn_readlane((int)ws, NSW - 1);
            }
            unsigned int F = 0;                                                              // records of my list that leave now (whole lines)
            if (t < nb2) {
                const unsigned int ex = inc - v + wbase;
                const unsigned int T = have + v;
                F = T & ~(unsigned int)(P2F_LINE - 1);
                const unsigned int L = F ? F - have : 0u;                                     // of the new records
                s_off[t] = ex;
                P2Meta M;
                M.gbase = slice_of((uint32_t)t) + ((uint64_t)cur + (uint64_t)have - (uint64_t)ex) * (uint64_t)RB;      // stage index i -> slice position cur + have + (i - ex)
                M.lim = ex + L;
                M.cadd = F ? -(int32_t)(ex + L) : (int32_t)have - (int32_t)ex;                // carry slot of a record that stays: i + cadd
                s_meta[t] = M;
                s_have[t] = have | (F ? 0x80000000u : 0u);
                if (cur + F > P.cap2) s_wsum[16] = 1;                                         // (stays set: the slice stays full)
            }
            if (t == 0) s_off[nb2] = total;                                                  // the padding sorts behind every record
            lds_barrier();
            // C. rows into LDS in list order
#pragma unroll
            for (int j = 0; j < P2F_ROWS; ++j) s_stage[s_off[br[j] >> 16] + (br[j] & 0xFFFFu)] = rec[j];
            // D1. what waited goes first: lane q of a list's 16 lanes takes waiting record q (read now, stored after the barrier:
            //     the records that stay this round go into the same slots)
            int td = t;                                           // (an opaque copy: what is derived from it is recomputed here, not kept live -- and spilled -- across the round)
            asm volatile("" : "+v"(td));
            constexpr int NU = P2F_MAXB / 64;                                                // lists per 16-lane group
            REC cw[NU];
            uint32_t cat[NU];                                                                // its position in the list's slice, ~0 = nothing
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if constexpr (W16) { cw[u].lo = 0ull; cw[u].hi = 0ull; } else cw[u] = 0ull;
                cat[u] = ~0u;
                const uint32_t b2 = ((uint32_t)td >> 4) + (uint32_t)u * 64u, q = (uint32_t)td & 15u;
                if (b2 < (uint32_t)nb2) {
                    const unsigned int hv = s_have[b2];
                    if ((hv >> 31) && q < (hv & 0xFFFFu)) {
                        cw[u] = s_carry[b2 * P2F_LINE + q];
                        const P2Meta M = s_meta[b2];
                        cat[u] = (uint32_t)((int64_t)(M.gbase - slice_of(b2)) / RB + (int64_t)s_off[b2]) - (hv & 0xFFFFu) + q;      // cur + q
                    }
                }
            }
            const bool overflow = s_wsum[16] != 0;
            if constexpr (PF) {
#pragma unroll
                for (int j = 0; j < P2F_ROWS; ++j) { if constexpr (W16) asm volatile("" : "+v"(nxt[j].lo), "+v"(nxt[j].hi)); else asm volatile("" : "+v"(nxt[j])); }   // (the next round's records have arrived: not after this point)
            }
            lds_barrier();
            if (!overflow) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (cat[u] != ~0u) reinterpret_cast<global_rec *>(slice_of(((uint32_t)td >> 4) + (uint32_t)u * 64u))[cat[u]] = cw[u];
                // D2. one lane per staged record: into the slice, or into the list's carry
#pragma unroll 2
                for (unsigned int i = (unsigned int)td; i < total; i += PT_THREADS) {
                    const REC rr = s_stage[i];
                    const uint32_t b2 = b2_of(rr);
                    const P2Meta M = s_meta[b2];
                    if (i < M.lim) reinterpret_cast<global_rec *>(M.gbase)[i] = rr;
                    else s_carry[b2 * P2F_LINE + (uint32_t)((int32_t)i + M.cadd)] = rr;
                }
            } else {
                // a slice is full: record by record, with the bound (what does not fit takes the deferred list)
                auto put = [&](uint32_t b2, uint64_t at, REC rr) {
                    if (at < P.cap2) reinterpret_cast<global_rec *>(slice_of(b2))[at] = rr;
                    else defer_record(P.stats, rec_hash(b1, rr, P.recbits), deferred, deferred_n, deferred_cap);
                };
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (cat[u] != ~0u) put(((uint32_t)td >> 4) + (uint32_t)u * 64u, cat[u], cw[u]);
                for (unsigned int i = (unsigned int)td; i < total; i += PT_THREADS) {
                    const REC rr = s_stage[i];
                    const uint32_t b2 = b2_of(rr);
                    const P2Meta M = s_meta[b2];
                    if (i < M.lim) put(b2, (uint64_t)((int64_t)(M.gbase - slice_of(b2)) / RB + (int64_t)i), rr);
                    else s_carry[b2 * P2F_LINE + (uint32_t)((int32_t)i + M.cadd)] = rr;
                }
            }
            if (t < nb2) { cur += F; have = have + v - F; }
            if constexpr (PF) {
#pragma unroll
                for (int j = 0; j < P2F_ROWS; ++j) rec[j] = nxt[j];
                vmask = vmask_n;
            }
            // (the next round's barriers order its writes to the stage, s_meta and s_carry against this copy-out)
        }
        lds_barrier();
        // the rest of every list closes its slice (the one line of a slice that is not written whole)
        if (t < nb2) {
            for (unsigned int q = 0; q < have; ++q) {
                const REC rr = s_carry[(uint32_t)t * P2F_LINE + q];
                if (cur + q < P.cap2) reinterpret_cast<global_rec *>(slice_of((uint32_t)t))[cur + q] = rr;
                else defer_record(P.stats, rec_hash(b1, rr, P.recbits), deferred, deferred_n, deferred_cap);
            }
            const unsigned int n = cur + have;
            cnt2[list_of((uint32_t)t) * P.nblk2 + blockIdx.x] = n < P.cap2 ? n : P.cap2;
        }
        lds_barrier();
    }
}

constexpr int LDS_HBINS = 1024;        // histogram bins kept in LDS by region_insert_kernel (higher multiplicities are rare: global atomics)
constexpr int LI_MAXSL = 64;           // slices per region an owner reads (region_insert_kernel<., XCHG>)
// ---- final: region lists -> LDS image of the region -> table ---------------------------------------------------------------------
// One workgroup per region of R = 2^rbits slots (R <= 4096 here: 48 KB of LDS, three workgroups of 512 threads per CU, so that
// one's record loads and image write-out overlap another's insert loop).  The LDS image keeps the tag (8 bytes, the table's own
// format) and -- on a table that starts empty (FRESH) -- a 32-bit count of what this pass adds; the slot's 16 bytes are put
// together on the way out.  A region is exactly the block's own slots: a record whose probe sequence leaves the region (0.03 % of
// them at load 0.3) goes to the deferred list and takes the direct path afterwards, so all regions run in ONE launch and every
// slot is written, and binned into the multiplicity histogram, exactly once (import3h_kernel moves the bins of the few keys it
// touches).  Round 2's form (lds_insert_kernel, gone: 16-byte LDS slots, a 128-slot halo, even and odd regions in two launches,
// one 1024-thread workgroup per CU) spent 178 wave instructions per record, most of them scalar branch bookkeeping of the probe
// loop; here the first probe of four records is straight-line code and only the lanes that miss it loop.
// a block-uniform word through the scalar cache (the compiler reads `cnt` with a vector load once stores have been issued, and
// then waits for every vector-memory operation of the wave)
__device__ __forceinline__ uint32_t scalar_load_u32(const unsigned int *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
constexpr int RI_TH = 512;
#ifndef JK_RI_PF
#define JK_RI_PF 6
#endif
constexpr int RI_PF = 8;              // records per lane in flight
constexpr int RI_PF_AHEAD = JK_RI_PF;  // ... per batch of the instance that keeps two batches in registers
// XCHG (the owner's side of the list exchange, several GPUs): the region's slices come from nsrc senders, each of which laid out
// ITS lists of all my regions as one block -- slice x of region r = sender x / (nsl/nsrc), its slice x % (nsl/nsrc): list index
// ((sender * nregions + r) * (nsl/nsrc) + that).  They are short (1/nsrc of a region's records each) and are read as ONE
// concatenated list (prefix of their lengths in LDS).  cbits > 0 (lists deduplicated by the sender, list_dedupe_kernel): the top
// fbits of a record's p2 field -- the second-level bits the SENDER resolved, implied by the list it made -- hold
// (occurrences - 1) in their low cbits.
// REC: uint64_t, or Rec16 (keys of 65 .. 128 bits, not with XCHG); a table whose remainders fit the tag (B - s <= 53) only: wide
// tables (a second word per slot) take region_insertw_kernel.
template <bool FRESH, bool XCHG = false, typename REC = uint64_t, bool R12 = false>
__global__ __launch_bounds__(RI_TH) void region_insert_kernel(const REC *__restrict__ lists, const unsigned int *__restrict__ cnt, uint32_t cap, uint32_t nsl,
                                                              TableDev T, PartGeom G, uint32_t nregions, unsigned long long *__restrict__ deferred,
                                                              unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap,
                                                              unsigned long long *__restrict__ histo, uint32_t nsrc = 1, int cbits = 0, int fbits = 0) {
    // (an owner adds up what several senders counted in pieces of up to 2^31 bases each: 64-bit counts there)
    using cnt_t = typename std::conditional<FRESH && !XCHG, unsigned int, unsigned long long>::type;
    constexpr bool W16 = sizeof(REC) == 16;
    static_assert(!(W16 && XCHG), "the exchange ships 8-byte records");
    constexpr bool AHEAD = FRESH && !XCHG && R12 && !W16;                            // (see below)
    constexpr int PF = AHEAD ? RI_PF_AHEAD : RI_PF;                                  // records per lane and batch
    auto rec_zero = []() { REC z; if constexpr (W16) { z.lo = 0ull; z.hi = 0ull; } else z = 0ull; return z; };
    extern __shared__ __align__(16) unsigned long long s_tag[];                      // R tags, R counts, LDS_HBINS bins (, LI_MAXSL + 1 slice offsets)
    const uint32_t R = 1u << G.rbits;                                                // (R12: 4096, the launcher says -- not folded in: as a constant it costs 18 registers)
    cnt_t *s_cnt = reinterpret_cast<cnt_t *>(s_tag + R);
    unsigned int *s_bins = reinterpret_cast<unsigned int *>(s_cnt + R);
    unsigned int *s_pref = s_bins + LDS_HBINS;                                       // XCHG: exclusive prefix of the region's slice lengths
    const int t = threadIdx.x;
    const int rs = T.B - T.s;                                                        // remainder bits (<= 53 by the tag format)
    const uint64_t rmask = (1ull << rs) - 1ull;
    const bool whole = nregions == 1;                                                // the whole table is one region: probes wrap inside it
    unsigned long long fresh = 0;
    if (!XCHG && deferred_n[1]) return;                                              // (uniform) the piece was abandoned: part_decide_kernel
    if (histo) {
        for (int i = t; i < LDS_HBINS; i += RI_TH) s_bins[i] = 0;
    }
    // AHEAD (a table of 8-byte records that starts empty, regions of 4096 slots, one GPU -- the headline workload's instance,
    // which has the registers): the records are a batch ahead of the insert loop.  While one batch of PF records per lane is
    // inserted the next one is on its way, and the first batch of the block's NEXT region is asked for BEFORE this region's image
    // is written out: a wave's vector-memory operations retire in order, so a load issued behind the write-out's stores is not
    // back before every one of them has been acknowledged (part1_kernel).  So that the compiler can tell how many operations
    // lie behind a load it waits for, the loads are unconditional (an index past the slice's end reads record 0 and is ignored),
    // the slice counts come through the scalar cache, the write-out is eight stores per lane, and the region loop is rotated:
    // its body is "ask for the next region's first batch; write this region out; set up and fill the next".
    // Measured (47 Mb workload, one run, A/B): 3.42 -> 3.30 ms.  (With a region's three batches as straight-line code every wait
    // is exact -- and the kernel needs 96 registers, two workgroups per CU instead of three: 3.87 ms.  As a loop, the wait before
    // a region's second request still covers most of the write-out's stores: the loop's exit is, as the compiler lays it
    // out, also a way back to its top.)
    uint32_t region = blockIdx.x;
    uint64_t first = 0, b1 = 0;                                                      // first slot of the region; its level-1 bucket
    uint32_t nrec = 0, total = 0;                                                    // records of the slice in hand; XCHG: of all the region's slices
    const uint32_t per_src = XCHG ? nsl / nsrc : nsl;
    auto slice_of = [&](uint32_t x) -> uint64_t {
        if constexpr (XCHG) return ((uint64_t)(x / per_src) * nregions + region) * per_src + x % per_src;
        else return (uint64_t)region * nsl + x;
    };
    auto count_of = [&](uint32_t region_, uint32_t x_) -> uint32_t {                 // (not XCHG) records in slice x_ of region_
        if constexpr (AHEAD) return scalar_load_u32(cnt + ((uint64_t)region_ * nsl + x_));
        else return cnt[(uint64_t)region_ * nsl + x_];
    };
    REC recs[PF], recs2[PF];
    auto fetch = [&](REC (&dst)[PF], uint32_t region_, uint32_t x_, uint32_t i0_, uint32_t n_) {   // (not XCHG)
        const REC *s = lists + ((uint64_t)region_ * nsl + x_) * cap;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const uint32_t i = i0_ + (uint32_t)u * RI_TH + t;
            if constexpr (AHEAD) dst[u] = s[i < n_ ? i : 0u];                        // (what lies past the end is not looked at: insert_batch)
            else dst[u] = i < n_ ? s[i] : rec_zero();
        }
    };
    // the probe loop of the lanes that did not find their key in its home slot (cur = what the home slot held)
    auto probe_on = [&](REC rec, uint32_t idx, unsigned long long want, unsigned long long cur, cnt_t inc) {
        for (;;) {
            if (cur == 0ull) {
                cur = atomicCAS(&s_tag[idx], 0ull, want);                            // LDS compare-and-swap
                if (cur == 0ull) { ++fresh; cur = want; }
            }
            if (cur == want) { atomicAdd(&s_cnt[idx], inc); return; }                // LDS add
            ++idx;
            ++want;                                                                  // tag_of(rem, off + 1): the offset is the tag's low bits
            if (whole) idx &= R - 1;
            if (idx >= R || (want & (unsigned long long)(MAXPROBE - 1)) == 0ull) {   // leaves the region (or the probe limit): direct path, later
                if constexpr (XCHG) {
                    const u128 hh = hash_of(b1, rec, G.recbits);
                    const unsigned long long di = atomicAdd(deferred_n, 1ull);
                    if (di < deferred_cap) { deferred[3 * di] = hh.hi; deferred[3 * di + 1] = hh.lo; deferred[3 * di + 2] = (unsigned long long)inc; }
                    else atomicExch(&T.stats[ST_FATAL], 1ull);
                } else defer_record(T, rec_hash(b1, rec, G.recbits), deferred, deferred_n, deferred_cap);
                return;
            }
            cur = s_tag[idx];
        }
    };
    auto insert_batch = [&](REC (&rr)[PF], uint32_t i0, uint32_t n) {             // rr[u] = record i0 + u * RI_TH + t of a slice of n
        uint32_t idx[PF];
        unsigned long long want[PF], cur[PF];
        cnt_t inc[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            inc[u] = (cnt_t)1;
            if constexpr (XCHG) {
                if (cbits) {                                                         // (uniform) the count, and the record as it was before the sender put it there
                    const int csh = G.recbits - fbits;
                    const uint64_t fmask = ((1ull << fbits) - 1ull) << csh;
                    inc[u] = (cnt_t)(((rr[u] >> csh) & ((1ull << cbits) - 1ull)) + 1ull);
                    rr[u] = (rr[u] & ~fmask) | ((uint64_t)((region >> (G.p2 - fbits)) & ((1u << fbits) - 1u)) << csh);
                }
            }
            // the record holds the low recbits hash bits; the bits above the slot index of this region are implied by the
            // list it is in, so slot and remainder come from the record alone
            if constexpr (W16) {
                idx[u] = (uint32_t)shr(mk(rr[u].hi, rr[u].lo), (unsigned)rs).lo & (R - 1);
                want[u] = OCC | ((rr[u].lo & rmask) << OFFBITS);                     // (rs <= 53: the remainder lies in the low word)
            } else {
                idx[u] = (uint32_t)(rr[u] >> rs) & (R - 1);
                want[u] = OCC | ((rr[u] & rmask) << OFFBITS);
            }
            cur[u] = s_tag[idx[u]];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (i0 + (uint32_t)u * RI_TH + t < n) {
                if (cur[u] == want[u]) atomicAdd(&s_cnt[idx[u]], inc[u]);
                else probe_on(rr[u], idx[u], want[u], cur[u], inc[u]);
            }
        }
    };
    // what a region brings: false = nothing, and nothing to do for its slots (block-uniform)
    auto open_region = [&]() -> bool {
        first = (uint64_t)region << G.rbits;
        b1 = region >> G.p2;
        if constexpr (XCHG) {
            total = 0;
            for (uint32_t x = 0; x < nsl; ++x) total += cnt[slice_of(x)];
            if (!FRESH && !histo && !total) return false;
            if (t == 0) {
                unsigned int run = 0;
                for (uint32_t x = 0; x < nsl; ++x) { s_pref[x] = run; run += cnt[slice_of(x)]; }
                s_pref[nsl] = run;
            }
            // (the barrier after the image set-up orders this before the first use)
        } else {
            nrec = count_of(region, 0u);
            if (!FRESH && !histo) {                                                  // nothing to add: the slots stay as they are
                uint32_t any = nrec;
                for (uint32_t x = 1; x < nsl; ++x) any |= count_of(region, x);
                if (!any) return false;
            }
        }
        return true;
    };
    auto image_setup = [&]() {
        if (FRESH) {
            for (uint32_t i = t; i < R; i += RI_TH) { s_tag[i] = 0ull; s_cnt[i] = 0; }
        } else {
            for (uint32_t i = t; i < R; i += RI_TH) {
                const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * (first + i));
                s_tag[i] = e.x;
                s_cnt[i] = (cnt_t)e.y;
            }
        }
    };
    // all records of the region into the image; recs: the first batch of its first slice (not XCHG)
    auto insert_region = [&]() {
        if constexpr (XCHG) {
            uint32_t x = 0;                                                          // slice of my current record: my indices only grow
            for (uint32_t i0 = 0; i0 < total; i0 += PF * RI_TH) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * RI_TH + t;
                    recs[u] = rec_zero();
                    if (i < total) {
                        while (s_pref[x + 1] <= i) ++x;                              // (a handful of slices)
                        recs[u] = lists[slice_of(x) * cap + (i - s_pref[x])];
                    }
                }
                insert_batch(recs, i0, total);
            }
        } else {
            uint32_t x = 0, i0 = 0;                                                  // the batch in hand: records i0 .. of slice x, which holds nrec
            auto advance = [&]() -> bool {                                           // (block-uniform) to the region's next batch, if there is one
                i0 += PF * RI_TH;
                while (x < nsl && i0 >= nrec) { ++x; i0 = 0; nrec = x < nsl ? count_of(region, x) : 0u; }
                return x < nsl;
            };
            if constexpr (AHEAD) {
                for (;;) {                                                           // (the two register batches take turns)
                    const uint32_t ia = i0, na = nrec;
                    const bool more_b = advance();
                    fetch(recs2, region, more_b ? x : 0u, more_b ? i0 : 0u, nrec);   // (no more: nrec == 0, a request for nothing)
                    insert_batch(recs, ia, na);
                    if (!more_b) break;
                    const uint32_t ib = i0, nb_ = nrec;
                    const bool more_a = advance();
                    fetch(recs, region, more_a ? x : 0u, more_a ? i0 : 0u, nrec);
                    insert_batch(recs2, ib, nb_);
                    if (!more_a) break;
                }
                // (the last request has long arrived -- said here, or the wait for it is put where its registers are written
                //  next: behind the write-out's stores, and for all of them)
#pragma unroll
                for (int u = 0; u < PF; ++u) asm volatile("" :: "v"(recs[u]), "v"(recs2[u]));
            } else {
                for (;;) {
                    insert_batch(recs, i0, nrec);
                    if (!advance()) break;
                    fetch(recs, region, x, i0, nrec);
                }
            }
        }
    };
    auto write_slot = [&](uint32_t i) {
        const unsigned long long tag = s_tag[i];
        const unsigned long long c64 = (unsigned long long)s_cnt[i];
        *reinterpret_cast<ulonglong2 *>(T.slots + 2 * (first + i)) = make_ulonglong2(tag, c64);
        if (histo) {
            const bool occ = tag != 0ull && c64 != 0ull;
            const uint32_t c = clamp32(c64);
            const uint32_t b = c > 10001u ? 10001u : c;
            // most occupied slots of a read set hold 1 (read errors): those are counted per wave, not by 64 same-address atomics
            const unsigned long long ones = __ballot(occ && b == 1u);
            if (ones && (t & 63) == (int)__builtin_ctzll(ones)) atomicAdd(&s_bins[1], (unsigned int)__popcll(ones));
            if (occ && b != 1u) {
                if (b < (uint32_t)LDS_HBINS) atomicAdd(&s_bins[b], 1u);
                else atomicAdd(&histo[b], 1ull);
            }
        }
    };
    auto write_region = [&]() {
        if constexpr (R12) {
            static_assert(!R12 || (1 << 12) == 8 * RI_TH, "eight slots per lane");
#pragma unroll
            for (int j = 0; j < 8; ++j) write_slot((uint32_t)t + (uint32_t)j * RI_TH);
        } else
            for (uint32_t i = t; i < R; i += RI_TH) write_slot(i);
    };
    if constexpr (AHEAD) {
        if (region < nregions) {
            open_region();
            fetch(recs, region, 0u, 0u, nrec);
            image_setup();
            lds_barrier();
            insert_region();
            for (;;) {
                const uint32_t next = region + gridDim.x;
                const bool has_next = next < nregions;
                const uint32_t n_next = has_next ? count_of(next, 0u) : 0u;
                fetch(recs, has_next ? next : region, 0u, 0u, n_next);
                lds_barrier();
                write_region();
                lds_barrier();
                if (!has_next) break;
                region = next;
                open_region();
                image_setup();
                lds_barrier();
                insert_region();
            }
        }
    } else {
        for (; region < nregions; region += gridDim.x) {
            if (!open_region()) continue;
            if constexpr (!XCHG) fetch(recs, region, 0u, 0u, nrec);                  // (requested before the image is set up)
            image_setup();
            lds_barrier();
            insert_region();
            lds_barrier();
            write_region();
            lds_barrier();
        }
    }
    if (histo) {
        for (int i = t; i < LDS_HBINS; i += RI_TH)
            if (s_bins[i]) atomicAdd(&histo[i], (unsigned long long)s_bins[i]);
    }
    // (one add per workgroup: the waves of the last workgroups all end within microseconds of each other, and thousands of adds
    //  to ONE word are served one after the other, ~12 ns each -- tools/probes/mall_probe.hip's first version found out)
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    lds_barrier();
    if (t == 0) s_tag[0] = 0ull;
    lds_barrier();
    if ((t & 63) == 0 && fresh) atomicAdd(&s_tag[0], fresh);
    lds_barrier();
    if (t == 0 && s_tag[0]) atomicAdd(&T.stats[ST_DISTINCT], s_tag[0]);
}

// ---- the same for a WIDE table (kmer.hpp: wide_rem -- B - s > 53: a slot's remainder does not fit its tag word, the low 64 bits
// live in T.ext) -----------------------------------------------------------------------------------------------------------------------
// The LDS image keeps tag, ext word and count of every slot (20 bytes per slot on a table that starts empty: regions of 2048 slots,
// three workgroups per CU as above).  A key is two words, and LDS has no 128-bit compare-and-swap: a lane claims an empty slot by
// the tag, writes the ext word, and only then adds its count -- a slot's key is complete once its count is non-zero.  A lane that
// meets a matching tag with a zero count cannot tell yet whether that is its key and must not wait in place (the claimant may be
// a lane of its own wave): the loop below is left only when every lane of the wave is done, so a trip is complete for all
// lanes before the next one starts, and the undecided lane looks at the same slot again on the next trip (table.hpp:
// table_put_wide is the same protocol on the table itself).
constexpr int RIW_PF = 4;              // records per lane in flight (16 bytes each)
template <bool FRESH>
__global__ __launch_bounds__(RI_TH) void region_insertw_kernel(const Rec16 *__restrict__ lists, const unsigned int *__restrict__ cnt, uint32_t cap, uint32_t nsl, TableDev T,
                                                               PartGeom G, uint32_t nregions, unsigned long long *__restrict__ deferred,
                                                               unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    using cnt_t = typename std::conditional<FRESH, unsigned int, unsigned long long>::type;
    extern __shared__ __align__(16) unsigned long long s_tag[];                      // R tags, R ext words, R counts
    const uint32_t R = 1u << G.rbits;
    unsigned long long *s_ext = s_tag + R;
    cnt_t *s_cnt = reinterpret_cast<cnt_t *>(s_ext + R);
    const int t = threadIdx.x;
    const int rs = T.B - T.s;                                                        // remainder bits: 54 .. 117
    const bool whole = nregions == 1;
    unsigned long long fresh = 0;
    if (deferred_n[1]) return;                                                       // (uniform) the piece was abandoned: part_decide_kernel
    for (uint32_t region = blockIdx.x; region < nregions; region += gridDim.x) {
        const uint64_t first = (uint64_t)region << G.rbits;
        const uint64_t b1 = region >> G.p2;
        if (!FRESH) {                                                                // (block-uniform) nothing to add: the slots stay as they are
            uint32_t any = 0;
            for (uint32_t x = 0; x < nsl; ++x) any |= cnt[(uint64_t)region * nsl + x];
            if (!any) continue;
        }
        if (FRESH) {
            for (uint32_t i = t; i < R; i += RI_TH) { s_tag[i] = 0ull; s_ext[i] = 0ull; s_cnt[i] = 0; }
        } else {
            for (uint32_t i = t; i < R; i += RI_TH) {
                const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * (first + i));
                s_tag[i] = e.x;
                s_cnt[i] = (cnt_t)e.y;
                s_ext[i] = T.ext[first + i];
            }
        }
        lds_barrier();
        for (uint32_t x = 0; x < nsl; ++x) {
            const uint32_t nrec = cnt[(uint64_t)region * nsl + x];
            const Rec16 *src = lists + ((uint64_t)region * nsl + x) * cap;
            for (uint32_t i0 = 0; i0 < nrec; i0 += RIW_PF * RI_TH) {
                Rec16 recs[RIW_PF];
#pragma unroll
                for (int u = 0; u < RIW_PF; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * RI_TH + t;
                    recs[u].lo = 0ull; recs[u].hi = 0ull;
                    if (i < nrec) recs[u] = src[i];
                }
#pragma unroll
                for (int u = 0; u < RIW_PF; ++u) {
                    const u128 rr = mk(recs[u].hi, recs[u].lo);
                    uint32_t idx = (uint32_t)shr(rr, (unsigned)rs).lo & (R - 1);
                    const u128 rem = band(rr, maskbits((unsigned)rs));
                    unsigned long long want = OCC | (rem.hi << OFFBITS);             // tag_of(tag_rem, 0): the offset is the tag's low bits
                    const unsigned long long ext = rem.lo;
                    bool done = !(i0 + (uint32_t)u * RI_TH + t < nrec);
                    uint32_t waited = 0;
                    for (;;) {
                        if (!done) {
                            unsigned long long cur = __hip_atomic_load(&s_tag[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (cur == 0ull) {
                                cur = atomicCAS(&s_tag[idx], 0ull, want);
                                if (cur == 0ull) {                                   // mine: ext word, then the count that makes the key complete
                                    __hip_atomic_store(&s_ext[idx], ext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(&s_cnt[idx], (cnt_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    ++fresh;
                                    done = true;
                                }
                            }
                            if (!done) {
                                bool next = true;
                                if (cur == want) {
                                    const cnt_t c = __hip_atomic_load(&s_cnt[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (c == 0) {                                    // claimed, key not complete yet: the same slot again on the next trip
                                        next = false;
                                        if (++waited > (1u << 16)) { defer_record(T, hash_of16(b1, recs[u], G.recbits), deferred, deferred_n, deferred_cap); done = true; }
                                    } else if (__hip_atomic_load(&s_ext[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == ext) {
                                        __hip_atomic_fetch_add(&s_cnt[idx], (cnt_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        done = true;
                                    }
                                }
                                if (!done && next) {
                                    ++idx;
                                    ++want;
                                    if (whole) idx &= R - 1;
                                    if (idx >= R || (want & (unsigned long long)(MAXPROBE - 1)) == 0ull) {   // leaves the region (or the probe limit): direct path, later
                                        defer_record(T, hash_of16(b1, recs[u], G.recbits), deferred, deferred_n, deferred_cap);
                                        done = true;
                                    }
                                }
                            }
                        }
                        if (__ballot(!done) == 0ull) break;
                    }
                }
            }
        }
        lds_barrier();
        for (uint32_t i = t; i < R; i += RI_TH) {
            *reinterpret_cast<ulonglong2 *>(T.slots + 2 * (first + i)) = make_ulonglong2(s_tag[i], (unsigned long long)s_cnt[i]);
            T.ext[first + i] = s_ext[i];
        }
        lds_barrier();
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// the deferred records of a piece that went into a wide table: through the table's own insert (no fused histogram there)
__global__ __launch_bounds__(256) void import3w_kernel(const unsigned long long *__restrict__ entries, const unsigned long long *__restrict__ n_ptr, uint64_t cap, TableDev T) {
    if (n_ptr[1]) return;
    const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
    unsigned long long fresh = 0;
    for (uint64_t i0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) & ~63ull; i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + (threadIdx.x & 63);
        int r = 1;
        if (i < n) r = table_add(T, mk(entries[3 * i], entries[3 * i + 1]), entries[3 * i + 2]);       // (whole waves: table_put_wide votes)
        if (r == 2) ++fresh;
        if (r == 0) table_spill(T, mk(entries[3 * i], entries[3 * i + 1]), entries[3 * i + 2]);
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// Between the partition passes and the insert: lists that overflowed more than the deferred list holds (one k-mer that makes up
// a tenth of the input: all its records go to ONE region list) cannot be inserted completely.  Nothing has touched the table yet,
// so the piece is ABANDONED here -- word 1 of the deferred list's header tells region_insert_kernel and import3h_kernel to do
// nothing -- and the host counts it again through the direct kernel (table.hip: count_device).
__global__ void part_decide_kernel(unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap, const unsigned long long *__restrict__ stats) {
    if (deferred_n[0] > deferred_cap || stats[ST_FATAL]) deferred_n[1] = 1ull;
}

// the deferred records through the direct path, with the fused histogram kept exact: a key whose count goes from c to c + inc
// leaves bin(c) and enters bin(c + inc) (every add returns the count it found, so concurrent adds to one key move it bin by bin)
__device__ __forceinline__ uint32_t histo_bin(unsigned long long c) { const uint32_t v = clamp32(c); return v > 10001u ? 10001u : v; }
__global__ __launch_bounds__(256) void import3h_kernel(const unsigned long long *__restrict__ entries, const unsigned long long *__restrict__ n_ptr, uint64_t cap,
                                                       TableDev T, unsigned long long *__restrict__ histo) {
    if (n_ptr[1]) return;                                                            // the piece was abandoned (part_decide_kernel)
    const uint64_t n = *n_ptr < cap ? *n_ptr : cap;
    unsigned long long fresh = 0;
    // the bins' changes are summed per workgroup first (nearly all of them are "leaves bin 1 / 2, enters bin 2 / 3": hundreds of
    // thousands of global atomics on a handful of words were most of this kernel's time)
    constexpr int NB = 64;
    __shared__ int s_delta[NB];
    if (threadIdx.x < NB) s_delta[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) & ~63ull; i0 < n; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + (threadIdx.x & 63);
        const bool have = i < n;
        u128 h = mk(0, 0);
        unsigned long long inc = 0;
        if (have) { h = mk(entries[3 * i], entries[3 * i + 1]); inc = entries[3 * i + 2]; }
        // a wave's 64 entries are often ONE key (what overflows a list is a k-mer far more frequent than the rest, and its records
        // were deferred one after the other): the first lane adds for all of them -- 64 atomics on one address would queue up
        {
            const int lead = (int)__builtin_ctzll(__ballot(have));
            const uint64_t lhi = (uint64_t)__shfl((long long)h.hi, lead), llo = (uint64_t)__shfl((long long)h.lo, lead);
            const bool same = have && h.hi == lhi && h.lo == llo;
            unsigned long long part = same ? inc : 0ull;
            for (int o = 32; o > 0; o >>= 1) part += (unsigned long long)__shfl_xor((long long)part, o);
            if (same) inc = (int)(threadIdx.x & 63) == lead ? part : 0ull;
        }
        if (!have || !inc) continue;
        const uint64_t home = home_of(h, T.B, T.s);
        const uint64_t rem = rem_of(h, T.B, T.s);
        bool done = false;
        for (uint32_t off = 0; off < MAXPROBE && !done; ++off) {
            const uint64_t slot = (home + off) & T.mask;
            const unsigned long long want = tag_of(rem, off);
            unsigned long long *p = T.slots + 2 * slot;
            unsigned long long cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                cur = atomicCAS(p, 0ull, want);
                if (cur == 0ull) { ++fresh; cur = want; }
            }
            if (cur == want) {
                const unsigned long long old = __hip_atomic_fetch_add(p + 1, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t b0 = histo_bin(old), b1 = histo_bin(old + inc);
                if (histo && (old == 0ull || b0 != b1)) {
                    if (old) { if (b0 < (uint32_t)NB) atomicAdd(&s_delta[b0], -1); else atomicAdd(&histo[b0], ~0ull); }      // (minus one)
                    if (b1 < (uint32_t)NB) atomicAdd(&s_delta[b1], 1); else atomicAdd(&histo[b1], 1ull);
                }
                done = true;
            }
        }
        if (!done) {                                                                 // no room within the probe limit: spill list, the table grows
            table_spill(T, h, inc);
            if (histo) histo[10002] = 1ull;                                           // "not complete": the re-insertion after growth does not know the bins
        }
    }
    __syncthreads();
    if (histo && threadIdx.x < NB && s_delta[threadIdx.x]) atomicAdd(&histo[threadIdx.x], (unsigned long long)(long long)s_delta[threadIdx.x]);
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

static uint32_t list_cap(double avg) { return (uint32_t)std::min<double>(4.0e9, avg * 1.25 + 8.0 * std::sqrt(avg) + 64.0); }

// can this piece take the partitioned path, and with which geometry?
bool Table::partition_geometry(uint64_t piece_bases, void *geom_out) const {
    PartGeom &G = *reinterpret_cast<PartGeom *>(geom_out);
    if (getenv("JASPER_COUNT_DIRECT")) return false;
    if (piece_bases < (8u << 20)) return false;          // small pieces: the direct kernel is already latency-hidden
    const int B = d.B, s = d.s;
    // 8-byte records hold the hash below the p1 <= 10 bucket bits: keys of up to 74 bits (k <= 37).  Longer keys travel as
    // 16-byte records (part1w_kernel, part2f_kernel<., ., ., Rec16>, region_insert_kernel<., ., Rec16>): half the records per LDS round.
    const bool rec16 = B > 74;
    if (rec16 && getenv("JASPER_COUNT_NO_REC16")) return false;
    if (d.ext && !rec16) return false;                    // (a tiny table for short keys: its images would need the second word for 8-byte records)
    if (d.ext && getenv("JASPER_COUNT_NO_WIDE")) return false;
    const int need_p1 = rec16 ? std::min(10, s - 12) : (B > 64 ? B - 64 : 0);
    // regions of 2^12 slots (three region_insert workgroups per CU); 2^13 only where the two list levels cannot split finer
    int p1 = 0, p2 = 0;
    bool ok = false;
    // (first choice: at most 512 lists per bucket, which part2f_kernel writes in whole lines)
    // (a wide table's image is 20 bytes per slot: regions of 2^11)
    const int rg0 = d.ext ? RG_MAXBITS - 1 : RG_MAXBITS;
    for (int maxp2 = 9; maxp2 <= (rec16 ? 9 : 11) && !ok; maxp2 += 2)
    for (int rg = rg0; rg <= rg0 + 1 && !ok; ++rg) {
        p1 = std::max(need_p1, (s - rg + 1) / 2);
        if (p1 < 1) p1 = 1;
        if (p1 > 10 || p1 > s - 8 || p1 < 1) continue;    // (a table too small to be worth it)
        p2 = s - rg - p1;
        if (p2 < 0) p2 = 0;
        ok = p2 <= maxp2;
    }
    if (!ok) return false;
    G.p1 = p1; G.p2 = p2; G.rbits = s - p1 - p2; G.recbits = B - p1;
    // one 1024-thread block (159 KB of LDS) per CU; the blocks of an XCD (b, b + 8, ...) share their slices (part1_kernel)
    const uint64_t tile1 = (uint64_t)PT_TILE;
    const uint64_t ntiles = (piece_bases + tile1 - 1) / tile1;
    uint64_t grid1 = piece_bases / ((uint64_t)(1u << p1) * 512);
    grid1 = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(grid1, ntiles), 256));
    static const int ngrp_exp = getenv("JASPER_EXPERIMENT_NGRP") ? atoi(getenv("JASPER_EXPERIMENT_NGRP")) : 0;     // tuning experiments only
    // (measured, 47 Mb workload, round 4's kernel with the fill counts slice-major: 2 to 16 slices per list 4.0-4.1 ms, 32 slices 4.2,
    //  ONE slice 5.3 -- 256 blocks adding to each count; with the counts bucket-major a wave's 64 atomics were 8 x nblk1 requests
    //  to the memory side, which alone took 4.4 ms at 2 slices and more at 8.  8 slices: part2f's 16 waves find slices of their own)
    const uint64_t nblk1 = std::min<uint64_t>(grid1, ngrp_exp > 0 ? (uint64_t)ngrp_exp : 8);
    G.grid1 = (uint32_t)grid1;
    G.nblk1 = (uint32_t)nblk1;
    // (a slice takes the tiles of ceil(grid1 / nblk1) blocks of ceil(ntiles / grid1) tiles each)
    const uint64_t tiles_per_slice = ((grid1 + nblk1 - 1) / nblk1) * ((ntiles + grid1 - 1) / grid1);
    // (a level-1 slice holds half a million records and more: 10 % above the BASES that can fall into it -- records are ~3/4 of the
    //  bases -- is slack for a bucket with a very frequent k-mer; what does not fit is deferred.  Device memory that has been used
    //  before is cleared when it is allocated again, ~28 ms per GB on some boxes: every GB of slack shows in a first call.)
    const double avg1 = (double)std::min<uint64_t>(piece_bases, tiles_per_slice * tile1) / (double)(1u << p1);
    G.cap1 = avg1 >= 65536.0 ? (uint32_t)std::min<double>(4.0e9, avg1 * 1.10 + 8.0 * std::sqrt(avg1) + 64.0) : list_cap(avg1);
    // one slice per region list: part2 runs one 1024-thread block per CU, and 2^p1 >= 256 buckets already fill the chip;
    // region_insert_kernel then reads a region's records as one contiguous list
    static const int nblk2_exp = getenv("JASPER_EXPERIMENT_NBLK2") ? atoi(getenv("JASPER_EXPERIMENT_NBLK2")) : 0;   // tuning experiments only
    G.nblk2 = p2 ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nblk1, nblk2_exp > 0 ? nblk2_exp : ((1u << p1) >= 256 ? 1 : 8))) : 1;
    G.cap2 = p2 ? (list_cap((double)piece_bases / ((double)(1ull << (p1 + p2)) * (double)G.nblk2)) + 15u) & ~15u : 0;      // whole 128-byte lines (part2f_kernel)
    return true;
}

int Table::launch_count_partitioned(const uint8_t *d_piece, uint64_t len, uint64_t emit_from, const void *geom, std::string &err) {
    const PartGeom G = *reinterpret_cast<const PartGeom *>(geom);
    const uint32_t nb1 = 1u << G.p1, nregions = 1u << (G.p1 + G.p2);
    const uint64_t deferred_cap = std::max<uint64_t>(1u << 16, len / 48);      // (24 bytes each; a piece that needs more abandons itself: part_decide_kernel)
    const size_t n_cnt1 = (size_t)nb1 * G.nblk1, n_cnt2 = G.p2 ? (size_t)nregions * G.nblk2 : n_cnt1;      // (one level: the counts once more, bucket-major)
    const bool rec16 = G.recbits > 64;
    const size_t RB = rec16 ? 16 : 8;
    uint64_t *out1 = (uint64_t *)workspace(WS_COUNT + 0, n_cnt1 * G.cap1 * RB, err);
    uint64_t *out2 = G.p2 ? (uint64_t *)workspace(WS_COUNT + 1, n_cnt2 * G.cap2 * RB, err) : nullptr;
    unsigned int *cur = (unsigned int *)workspace(WS_COUNT + 2, (n_cnt1 + n_cnt2 + 4) * 4, err);
    unsigned long long *defer = (unsigned long long *)workspace(WS_COUNT + 3, deferred_cap * 24 + 64, err);
    if (!out1 || (G.p2 && !out2) || !cur || !defer) return -2;
    unsigned int *cnt1 = cur, *cnt2 = cur + n_cnt1;
    unsigned long long *defer_n = defer;                    // first 8 bytes: counter; entries start 64 bytes in
    unsigned long long *defer_e = defer + 8;
    HIPCHK(hipMemsetAsync(defer_n, 0, 64, stream));
    const uint64_t ntiles = (len + (uint64_t)PT_TILE - 1) / (uint64_t)PT_TILE;
    for (int i = 0; i < 6; ++i) if (!ev_stage_t[i]) HIPCHK(hipEventCreate(&ev_stage_t[i]));
    part_stage_n = 4;
    count_path = 1;
    HIPCHK(hipEventRecord(ev_k0, stream));
    HIPCHK(hipEventRecord(ev_stage_t[0], stream));
    HIPCHK(launch_part1(stream, k, d_piece, len, ntiles, emit_from, d, G, out1, cnt1, defer_e, defer_n, deferred_cap));
    HIPCHK(hipEventRecord(ev_stage_t[1], stream));
    const uint64_t *lists = out1;
    const unsigned int *lcnt = cnt1;
    uint32_t lcap = G.cap1, nsl = G.nblk1;
    if (G.p2) {
        dim3 grid(G.nblk2, std::min<uint32_t>(nb1, 2048));
        static bool attr2_set = false;
        if (!attr2_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr2_set = true;
        }
        if (rec16) {
            static bool attr2w_set = false;
            if (!attr2w_set) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2f_kernel<512, 4, false, Rec16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr2w_set = true;
            }
            if ((1 << G.p2) > 512 || G.cap2 % P2F_LINE) { err = "count: no list geometry for 16-byte records"; return -1; }
            P2Args P;
            P.p1 = G.p1; P.p2 = G.p2; P.recbits = G.recbits; P.nblk1 = G.nblk1; P.nblk2 = G.nblk2; P.cap1 = G.cap1; P.cap2 = G.cap2; P.stats = d.stats; P.nown = 1;
            constexpr uint32_t piece = (uint32_t)P2F<512, 4, 16>::PIECE;
            P.vper = (G.cap1 + piece - 1u) / piece;
            constexpr size_t lds2 = P2F<512, 4, 16>::LDS;
            hipLaunchKernelGGL((part2f_kernel<512, 4, false, Rec16>), grid, dim3(PT_THREADS), lds2, stream, reinterpret_cast<const Rec16 *>(out1), cnt1, P, reinterpret_cast<Rec16 *>(out2), cnt2, defer_e, defer_n,
                               deferred_cap);
        } else if ((1 << G.p2) <= 512 && G.cap2 % P2F_LINE == 0 && !getenv("JASPER_EXPERIMENT_OLDP2")) {
            static bool attr2f_set = false;
            if (!attr2f_set) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2f_kernel<512, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2f_kernel<128, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr2f_set = true;
            }
            P2Args P;
            P.p1 = G.p1; P.p2 = G.p2; P.recbits = G.recbits; P.nblk1 = G.nblk1; P.nblk2 = G.nblk2; P.cap1 = G.cap1; P.cap2 = G.cap2; P.stats = d.stats; P.nown = 1;
            if ((1 << G.p2) <= 128) {
                P.vper = (G.cap1 + (uint32_t)P2F<128, 12>::PIECE - 1u) / (uint32_t)P2F<128, 12>::PIECE;
                constexpr size_t lds2 = P2F<128, 12>::LDS;
                hipLaunchKernelGGL((part2f_kernel<128, 12>), grid, dim3(PT_THREADS), lds2, stream, out1, cnt1, P, out2, cnt2, defer_e, defer_n, deferred_cap);
            } else {
                P.vper = (G.cap1 + (uint32_t)P2F<512, 10>::PIECE - 1u) / (uint32_t)P2F<512, 10>::PIECE;
                constexpr size_t lds2 = P2F<512, 10>::LDS;
                hipLaunchKernelGGL((part2f_kernel<512, 10>), grid, dim3(PT_THREADS), lds2, stream, out1, cnt1, P, out2, cnt2, defer_e, defer_n, deferred_cap);
            }
        } else
            hipLaunchKernelGGL(part2_kernel<false>, grid, dim3(PT_THREADS), P2_LDS, stream, out1, cnt1, d, G, out2, cnt2, defer_e, defer_n, deferred_cap, 1u);
        HIPCHK(hipGetLastError());
        lists = out2; lcnt = cnt2; lcap = G.cap2; nsl = G.nblk2;
    } else {
        hipLaunchKernelGGL(transpose_counts_kernel, dim3((uint32_t)((n_cnt1 + 255) / 256)), dim3(256), 0, stream, cnt1, cnt2, nb1, G.nblk1);
        HIPCHK(hipGetLastError());
        lcnt = cnt2;
    }
    hipLaunchKernelGGL(part_decide_kernel, dim3(1), dim3(1), 0, stream, defer_n, deferred_cap, d.stats);
    HIPCHK(hipGetLastError());
    part_defer_header = defer_n;
    part_slots_dirty_before = slots_dirty;
    HIPCHK(hipEventRecord(ev_stage_t[2], stream));
    // fused histogram: asked for by count_device when this piece is the whole input going into an empty table
    if (d.ext) histo_request = false;                      // (a wide table's histogram is read from the table afterwards)
    unsigned long long *histo = histo_request ? d_histo : nullptr;
    if (histo) HIPCHK(hipMemsetAsync(histo, 0, HISTO_WORDS * sizeof(unsigned long long), stream));
    // A table that is logically empty is not read: the images start from zeros and every slot is written (a lazily cleared table
    // is never zeroed in HBM).  Its LDS image counts in 32 bits -- a piece below 2^32 bases cannot overflow them.
    const bool empty_tbl = slots_dirty || histo != nullptr;
    const bool fresh32 = empty_tbl && len < (1ull << 32);
    if (slots_dirty && !fresh32) { if (materialize(err)) return -1; }
    const uint32_t R = 1u << G.rbits;
    const size_t lds = (size_t)R * (fresh32 ? 12 : 16) + (histo ? LDS_HBINS * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<true, false, uint64_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    {
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
        const uint32_t nblk = std::max<uint32_t>(1, std::min<uint32_t>(nregions, 256 * per_cu * 4));
        if (d.ext) {
            static bool attrww_set = false;
            if (!attrww_set) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insertw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insertw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attrww_set = true;
            }
            const size_t ldsw = (size_t)R * (fresh32 ? 20 : 24);
            const uint32_t per_cuw = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / ldsw));
            const uint32_t nblkw = std::max<uint32_t>(1, std::min<uint32_t>(nregions, 256 * per_cuw * 4));
            const Rec16 *lw = reinterpret_cast<const Rec16 *>(lists);
            if (fresh32) hipLaunchKernelGGL(region_insertw_kernel<true>, dim3(nblkw), dim3(RI_TH), ldsw, stream, lw, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap);
            else hipLaunchKernelGGL(region_insertw_kernel<false>, dim3(nblkw), dim3(RI_TH), ldsw, stream, lw, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap);
        } else if (rec16) {
            static bool attrw_set = false;
            if (!attrw_set) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<true, false, Rec16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<false, false, Rec16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attrw_set = true;
            }
            const Rec16 *lw = reinterpret_cast<const Rec16 *>(lists);
            if (fresh32) hipLaunchKernelGGL((region_insert_kernel<true, false, Rec16>), dim3(nblk), dim3(RI_TH), lds, stream, lw, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap, histo, 1u, 0, 0);
            else hipLaunchKernelGGL((region_insert_kernel<false, false, Rec16>), dim3(nblk), dim3(RI_TH), lds, stream, lw, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap, histo, 1u, 0, 0);
        } else if (fresh32 && G.rbits == 12 && !getenv("JASPER_EXPERIMENT_NO_AHEAD"))      // (the instance that runs a batch of records ahead)
            hipLaunchKernelGGL((region_insert_kernel<true, false, uint64_t, true>), dim3(nblk), dim3(RI_TH), lds, stream, lists, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap, histo, 1u, 0, 0);
        else if (fresh32) hipLaunchKernelGGL(region_insert_kernel<true>, dim3(nblk), dim3(RI_TH), lds, stream, lists, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap, histo);
        else hipLaunchKernelGGL(region_insert_kernel<false>, dim3(nblk), dim3(RI_TH), lds, stream, lists, lcnt, lcap, nsl, d, G, nregions, defer_e, defer_n, deferred_cap, histo);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ev_stage_t[3], stream));
    slots_dirty = false;   // every region has been written by the launch above
    if (d.ext) hipLaunchKernelGGL(import3w_kernel, dim3(256), dim3(256), 0, stream, defer_e, defer_n, deferred_cap, d);
    else hipLaunchKernelGGL(import3h_kernel, dim3(256), dim3(256), 0, stream, defer_e, defer_n, deferred_cap, d, histo);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_k1, stream));
    HIPCHK(hipEventRecord(ev_stage_t[4], stream));
    part_stage_pending = true;
    if (getenv("JASPER_COUNT_DEBUG") && atoi(getenv("JASPER_COUNT_DEBUG")) >= 2) {
        HIPCHK(jk_stream_wait(stream));
        unsigned long long dn = 0;
        HIPCHK(hipMemcpy(&dn, defer_n, 8, hipMemcpyDeviceToHost));
        std::vector<unsigned int> hc(n_cnt1 + n_cnt2);
        HIPCHK(hipMemcpy(hc.data(), cur, hc.size() * 4, hipMemcpyDeviceToHost));
        unsigned int mx1 = 0, mx2 = 0;
        for (size_t i = 0; i < n_cnt1; ++i) mx1 = std::max(mx1, hc[i]);
        for (size_t i = 0; i < n_cnt2; ++i) mx2 = std::max(mx2, hc[n_cnt1 + i]);
        fprintf(stderr, "[count] partitioned piece %llu bases: p1 %d p2 %d rbits %d nblk1 %u nblk2 %u cap1 %u cap2 %u | fullest slice1 %u slice2 %u deferred %llu\n",
                (unsigned long long)len, G.p1, G.p2, G.rbits, G.nblk1, G.nblk2, G.cap1, G.cap2, mx1, mx2, dn);
    }
    return 0;
}

// ==================================================================================================
// count_exchange: the partition pipeline as the multi-GPU exchange (role: JF::jellyfish/merge_files.cc:44-96 -- there
// every process counts into a table of its own and the tables are merged; here no rank ever builds a table of its own reads).
//   every rank    xchg_scan        part1: its reads become level-1 lists (in its workspace); returns how many records
//                 xchg_partition   part2<OWN>: the level-1 lists become region lists grouped by the OWNER of the key;
//                                  what owner o is to get is one contiguous block of the send buffers (records + slice counts)
//   the caller    one all_to_all of the blocks (8 B per k-mer occurrence + slack), dist.count_sharded
//   every owner   xchg_insert      region_insert<., XCHG> straight into its shard, fused histogram, deferred records
// All ranks must derive the same geometry: from the shard geometry (one for all owners), the number of owners, piece_max =
// the longest piece any rank scans in this round, and records_max = the most records any rank's scan produced (0: not known,
// piece_max stands in).  The slack of the send lists is what travels, so they are sized from the records actually there, not
// with the 1.25x of the local lists: mean + 6 sigma, where a list's fill is a sum over the distinct keys hashed into it of
// their multiplicities in this sender's reads -- variance = mean x (1 + multiplicity).  The multiplicity is estimated from the
// table the caller sized for the keys (about a third full): records / (keys all shards were sized for).  A list that still
// overflows (a k-mer far more frequent than the rest, or a table sized far too large) spends the deferred list, as everywhere.
static uint32_t xchg_cap(double avg, double mult) { return (uint32_t)std::min<double>(4.0e9, avg + 6.0 * std::sqrt(avg * (1.0 + mult)) + 16.0); }
// G: what the senders and the wire use (G.p2 = the second-level bits the SENDER resolves); p2b: second-level bits left to the
// owner -- a bucket can be split 2048 ways in one pass, and the senders also split by owner, so for very large shards (2^32
// slots on 8 GPUs) the owner runs one more split pass over what it received (part2_kernel<false, MIN>) before its insert.
static int xchg_max_lists() { const char *e = getenv("JASPER_XCHG_TEST_MAXLISTS"); return e ? std::max(2, atoi(e)) : PT_MAXBUCKETS; }   // (tests: force the extra pass on small tables)
static bool xchg_geometry(const Table &t, uint64_t piece_max, uint64_t records_max, uint32_t nown, PartGeom &G, int &p2b) {
    if (nown < 2 || nown > MAX_SHARDS) return false;
    alignas(16) char raw[64];
    // (a feed's last batch may be small: the lists are then laid out as for the smallest piece the passes are tuned for)
    piece_max = std::max<uint64_t>(piece_max, 8u << 20);
    if (!t.partition_geometry(piece_max, raw)) return false;
    G = *reinterpret_cast<const PartGeom *>(raw);
    if (G.recbits > 64) return false;                     // (keys of more than 74 bits: per-GPU tables and the entry exchange instead)
    // the senders split by (owner, second-level bits): more than 512 lists per bucket would leave the whole-line kernel, so the
    // owners take regions of 8192 slots instead of 4096 where that is what it takes -- for up to four owners, whose lists travel
    // deduplicated (dist.dedupe_pays).  Beyond that the lists travel as they are, and an owner inserting 10^9 raw records into
    // 8192-slot regions (one workgroup per CU) takes 8.6 ms against 5.7 into 4096-slot ones, which the general split kernel's
    // 5.7 ms against 4.6 does not eat up (role-play at 8 ranks, profiles/round5/exchange_roleplay_model.txt: 15.7 against 17.3 ms
    // of kernels per rank).  JASPER_EXPERIMENT_XCHG_RB12 / _RB13 force one or the other.
    const bool rb13 = getenv("JASPER_EXPERIMENT_XCHG_RB13") ? true : getenv("JASPER_EXPERIMENT_XCHG_RB12") ? false : nown <= 4;
    if (((uint64_t)nown << G.p2) > 512 && ((uint64_t)nown << (G.p2 - 1)) <= 512 && G.p2 > 0 && G.rbits == RG_MAXBITS && rb13) { --G.p2; ++G.rbits; }
    p2b = 0;
    while (((uint64_t)nown << (G.p2 - p2b)) > (uint64_t)xchg_max_lists() && p2b < G.p2) ++p2b;
    if (((uint64_t)nown << (G.p2 - p2b)) > (uint64_t)PT_MAXBUCKETS) return false;
    G.p2 -= p2b;
    const uint32_t nb1 = 1u << G.p1;
    G.nblk2 = nb1 >= 256 ? 1u : std::min<uint32_t>(G.nblk1, 8u);
    if (nown * G.nblk2 > (uint32_t)LI_MAXSL) return false;
    const double lists = (double)nb1 * (double)((uint64_t)nown << G.p2) * (double)G.nblk2;
    if (records_max) {
        const double rec = (double)std::min(records_max, piece_max);
        const double keys = std::max(1.0, std::min(rec, 0.35 * (double)t.nslots * (double)nown));
        G.cap2 = xchg_cap(rec / lists, rec / keys);
    } else G.cap2 = list_cap((double)piece_max / lists);
    G.cap2 = (G.cap2 + 15u) & ~15u;                       // whole 128-byte lines (part2f_kernel<., ., OWN>)
    return true;
}

int Table::xchg_plan(uint64_t piece_max, uint64_t records_max, uint32_t nown, uint64_t out[8], std::string &err) {
    (void)err;
    PartGeom G;
    int p2b = 0;
    if (!xchg_geometry(*this, piece_max, records_max, nown, G, p2b)) return 1;
    const uint64_t lists_per_owner = (uint64_t)1 << (G.p1 + G.p2);
    out[0] = lists_per_owner * G.nblk2 * G.cap2;               // records (8 B) per owner block
    out[1] = lists_per_owner * G.nblk2;                        // slice counts (4 B) per owner block
    out[2] = std::max<uint64_t>(1u << 16, piece_max / 16);     // deferred entries (24 B) a rank may produce
    out[3] = (uint64_t)G.p1; out[4] = (uint64_t)G.p2 | ((uint64_t)p2b << 8); out[5] = (uint64_t)G.rbits; out[6] = G.nblk2; out[7] = G.cap2;
    return 0;
}

// d_defer: 64-byte header (word 0 = number of entries) + defer_cap entries of 3 words (hash.hi, hash.lo, 1)
int Table::xchg_scan(const uint8_t *d_bases, uint64_t n, uint64_t pos, uint64_t end, uint64_t piece_max, uint32_t nown, void *d_defer, uint64_t defer_cap,
                     uint64_t *records, std::string &err) {
    HIPCHK(hipSetDevice(device));
    PartGeom G;
    int p2b = 0;
    if (!xchg_geometry(*this, piece_max, 0, nown, G, p2b)) { err = "count exchange: no geometry for this table / piece size"; return -1; }
    if (end > n) end = n;
    const uint32_t nb1 = 1u << G.p1;
    unsigned long long *defer_n = (unsigned long long *)d_defer, *defer_e = defer_n + 8;
    HIPCHK(hipMemsetAsync(defer_n, 0, 64, stream));
    for (int i = 0; i <= N_STAGES; ++i) if (!ev_stage_t[i]) HIPCHK(hipEventCreate(&ev_stage_t[i]));
    count_path = 3;
    part_stage_n = 5;
    if (read_stats(err)) return -1;
    const uint64_t occ_before = h_stats[ST_OCCURRENCES];
    HIPCHK(hipEventRecord(ev_stage_t[0], stream));
    xchg_partitioned = true;
    const size_t n_cnt1 = (size_t)nb1 * G.nblk1;
    uint64_t *out1 = (uint64_t *)workspace(WS_COUNT + 0, n_cnt1 * G.cap1 * 8, err);
    unsigned int *cnt1 = (unsigned int *)workspace(WS_COUNT + 2, (n_cnt1 + 4) * 4, err);
    if (!out1 || !cnt1) return -2;
    if (pos >= end) {                                          // nothing of mine in this round: empty lists
        HIPCHK(hipMemsetAsync(cnt1, 0, n_cnt1 * 4, stream));
        HIPCHK(hipEventRecord(ev_stage_t[1], stream));
        if (records) *records = 0;
        return 0;
    }
    if (end - pos > piece_max) { err = "count exchange: piece longer than the agreed maximum"; return -1; }
    const uint64_t halo = (uint64_t)(k - 1);
    const uint64_t misalign = reinterpret_cast<uintptr_t>(d_bases) & 15;
    uint64_t start = pos >= halo ? pos - halo : 0;
    const uint64_t a = (start + misalign) & 15;
    start = start >= a ? start - a : 0;
    const uint64_t len = end - start, emit_from = pos - start;
    const uint64_t ntiles = (len + (uint64_t)PT_TILE - 1) / (uint64_t)PT_TILE;
    HIPCHK(launch_part1(stream, k, d_bases + start, len, ntiles, emit_from, d, G, out1, cnt1, defer_e, defer_n, defer_cap));
    HIPCHK(hipEventRecord(ev_stage_t[1], stream));
    if (read_stats(err)) return -1;                            // (waits for the kernel)
    if (records) *records = h_stats[ST_OCCURRENCES] - occ_before;
    return 0;
}

int Table::xchg_partition(uint64_t piece_max, uint64_t records_max, uint32_t nown, void *d_send, void *d_send_cnt, void *d_defer, uint64_t defer_cap, std::string &err) {
    HIPCHK(hipSetDevice(device));
    PartGeom G;
    int p2b = 0;
    if (!xchg_geometry(*this, piece_max, records_max, nown, G, p2b)) { err = "count exchange: no geometry for this table / piece size"; return -1; }
    if (!xchg_partitioned) { err = "count exchange: partition without a scan before it"; return -1; }
    const uint32_t nb1 = 1u << G.p1;
    unsigned long long *defer_n = (unsigned long long *)d_defer, *defer_e = defer_n + 8;
    const size_t n_cnt1 = (size_t)nb1 * G.nblk1;
    uint64_t *out1 = (uint64_t *)workspace(WS_COUNT + 0, n_cnt1 * G.cap1 * 8, err);       // (what the scan filled: same sizes, nothing is reallocated)
    unsigned int *cnt1 = (unsigned int *)workspace(WS_COUNT + 2, (n_cnt1 + 4) * 4, err);
    if (!out1 || !cnt1) return -2;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (((uint64_t)nown << G.p2) <= 512 && G.cap2 % P2F_LINE == 0 && !getenv("JASPER_EXPERIMENT_OLDP2")) {
        // the sender's split in whole lines, like the single-GPU pass
        static bool attrf_set = false;
        if (!attrf_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2f_kernel<512, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attrf_set = true;
        }
        P2Args P;
        P.p1 = G.p1; P.p2 = G.p2; P.recbits = G.recbits; P.nblk1 = G.nblk1; P.nblk2 = G.nblk2; P.cap1 = G.cap1; P.cap2 = G.cap2; P.stats = d.stats; P.nown = nown;
        P.vper = (G.cap1 + (uint32_t)P2F<512, 10>::PIECE - 1u) / (uint32_t)P2F<512, 10>::PIECE;
        constexpr size_t lds2 = P2F<512, 10>::LDS;
        hipLaunchKernelGGL((part2f_kernel<512, 10, true>), dim3(G.nblk2, std::min<uint32_t>(nb1, 2048)), dim3(PT_THREADS), lds2, stream, out1, cnt1, P, (uint64_t *)d_send,
                           (unsigned int *)d_send_cnt, defer_e, defer_n, defer_cap);
    } else
    hipLaunchKernelGGL(part2_kernel<true>, dim3(G.nblk2, std::min<uint32_t>(nb1, 2048)), dim3(PT_THREADS), P2_LDS, stream, out1, cnt1, d, G, (uint64_t *)d_send,
                       (unsigned int *)d_send_cnt, defer_e, defer_n, defer_cap, nown);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[2], stream));
    return 0;
}

// ---- what travels, made smaller: the sender's lists deduplicated in place --------------------------------------------------
// A read shard repeats its k-mers (coverage / number of GPUs times), and every copy of a k-mer is in the same list.  One
// workgroup per list: chunks of 4096 records go through an LDS hash set with counters and come back as one record per distinct
// key, written over the front of the list, with (occurrences - 1) in the bits of the record that the LIST implies anyway (the
// p2 second-level bits; a key with more than 2^cbits occurrences leaves as several records).  The list counts are updated and
// the fullest list reported: the caller then ships only that many records per list.
constexpr int DD_TH = 256, DD_SLOTS = 4096, DD_CHUNK = 3072, DD_PER = DD_CHUNK / DD_TH;      // (a chunk of all-distinct records fills the set to 3/4)
__global__ __launch_bounds__(DD_TH) void list_dedupe_kernel(uint64_t *__restrict__ lists, unsigned int *__restrict__ cnt, uint32_t cap, uint64_t nlists, int cshift,
                                                            int cbits, unsigned int *__restrict__ max_fill) {
    // 48 KB of LDS: three workgroups per CU.  The thread whose compare-and-swap put a key into the set is its claimant: after
    // the barrier it writes the key's record(s) out and empties the slot again, so the set is never scanned or cleared as a whole.
    __shared__ unsigned long long s_key[DD_SLOTS];
    __shared__ unsigned int s_cnt[DD_SLOTS];
    __shared__ unsigned int s_out, s_all1;
    const int t = threadIdx.x;
    constexpr unsigned long long EMPTY = ~0ull;
    const uint64_t fmask = ((1ull << cbits) - 1ull) << cshift;
    const unsigned int per = 1u << cbits;
    unsigned int fullest = 0;
    for (int i = t; i < DD_SLOTS; i += DD_TH) { s_key[i] = EMPTY; s_cnt[i] = 0; }
    for (uint64_t L = blockIdx.x; L < nlists; L += gridDim.x) {
        const unsigned int n = cnt[L] < cap ? cnt[L] : cap;
        uint64_t *lst = lists + L * (uint64_t)cap;
        if (t == 0) { s_out = 0; s_all1 = 0; }
        __syncthreads();
        for (unsigned int c0 = 0; c0 < n; c0 += DD_CHUNK) {
            const unsigned int m = n - c0 < (unsigned int)DD_CHUNK ? n - c0 : (unsigned int)DD_CHUNK;
            unsigned long long r[DD_PER];
            unsigned short slot[DD_PER];                       // where my record's key sits if I am its claimant, else 0xFFFF
#pragma unroll
            for (int u = 0; u < DD_PER; ++u) { const unsigned int i = (unsigned int)u * DD_TH + t; r[u] = i < m ? lst[c0 + i] : EMPTY; }
#pragma unroll
            for (int u = 0; u < DD_PER; ++u) {
                slot[u] = 0xFFFFu;
                if ((unsigned int)u * DD_TH >= m) continue;                    // (block-uniform: a short chunk)
                const unsigned int i = (unsigned int)u * DD_TH + t;
                if (i >= m) continue;
                if (r[u] == EMPTY) { atomicAdd(&s_all1, 1u); continue; }       // (the one value the set cannot hold)
                uint32_t h = (uint32_t)(r[u] ^ (r[u] >> 13)) & (DD_SLOTS - 1);        // (the low record bits are hash bits already: kmer.hpp mix)
                for (;;) {
                    unsigned long long cur = s_key[h];
                    if (cur == EMPTY) {
                        cur = atomicCAS(&s_key[h], EMPTY, r[u]);
                        if (cur == EMPTY) { cur = r[u]; slot[u] = (unsigned short)h; }
                    }
                    if (cur == r[u]) { atomicAdd(&s_cnt[h], 1u); break; }
                    h = (h + 1) & (DD_SLOTS - 1);
                }
            }
            __syncthreads();                                   // the chunk has been read: its place (and what lies before it) may be written
#pragma unroll
            for (int u = 0; u < DD_PER; ++u) {
                // the claimants of a wave take their places in the output with ONE atomic (64 lanes adding to one LDS word are
                // served one after the other); a key with more occurrences than a record holds takes its further places alone
                if ((unsigned int)u * DD_TH >= m) continue;                    // (block-uniform)
                const bool mine = slot[u] != 0xFFFFu;
                const unsigned long long claim = __ballot(mine);
                if (!claim) continue;
                unsigned int base = 0;
                if ((t & 63) == (int)__builtin_ctzll(claim)) base = atomicAdd(&s_out, (unsigned int)__popcll(claim));
                base = (unsigned int)__shfl((int)base, (int)__builtin_ctzll(claim));
                if (!mine) continue;
                unsigned int left = s_cnt[slot[u]];
                const unsigned int first = left < per ? left : per;
                lst[base + (unsigned int)__popcll(claim & ((1ull << (t & 63)) - 1ull))] = (r[u] & ~fmask) | ((uint64_t)(first - 1u) << cshift);
                left -= first;
                while (left) {
                    const unsigned int take = left < per ? left : per;
                    lst[atomicAdd(&s_out, 1u)] = (r[u] & ~fmask) | ((uint64_t)(take - 1u) << cshift);
                    left -= take;
                }
                s_key[slot[u]] = EMPTY;
                s_cnt[slot[u]] = 0;
            }
            if (t == 0 && s_all1) {
                unsigned int left = s_all1;
                while (left) {
                    const unsigned int take = left < per ? left : per;
                    lst[atomicAdd(&s_out, 1u)] = (EMPTY & ~fmask) | ((uint64_t)(take - 1u) << cshift);
                    left -= take;
                }
                s_all1 = 0;
            }
            __syncthreads();
        }
        if (t == 0) cnt[L] = n ? s_out : 0u;
        fullest = s_out > fullest ? s_out : fullest;
        __syncthreads();
    }
    if (t == 0 && fullest) atomicMax(max_fill, fullest);
}

// 0 done (*max_fill = records in the fullest list now), 1 this geometry has no bits to hold the counts (nothing done)
int Table::xchg_dedupe(uint64_t piece_max, uint64_t records_max, uint32_t nown, void *d_send, void *d_send_cnt, uint32_t *max_fill, int *cbits_out, std::string &err) {
    HIPCHK(hipSetDevice(device));
    PartGeom G;
    int p2b = 0;
    if (!xchg_geometry(*this, piece_max, records_max, nown, G, p2b)) { err = "count exchange: no geometry for this table / piece size"; return -1; }
    if (G.p2 < 1) return 1;                                    // (G.p2: the second-level bits the senders resolve)
    const int cbits = std::min(G.p2, 16);
    const uint64_t nlists = ((uint64_t)nown << (G.p1 + G.p2)) * G.nblk2;
    unsigned int *d_max = (unsigned int *)workspace(WS_XCHG + 3, 64, err);
    if (!d_max) return -2;
    HIPCHK(hipMemsetAsync(d_max, 0, 4, stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nlists, 256u * 3u * 8u);
    hipLaunchKernelGGL(list_dedupe_kernel, dim3(grid), dim3(DD_TH), 0, stream, (uint64_t *)d_send, (unsigned int *)d_send_cnt, G.cap2, nlists, G.recbits - G.p2, cbits, d_max);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[7], stream));             // (stage "dedupe": between the partition and this)
    unsigned int h = 0;
    HIPCHK(hipMemcpyAsync(&h, d_max, 4, hipMemcpyDeviceToHost, stream));
    HIPCHK(jk_stream_wait(stream));
    if (max_fill) *max_fill = h;
    if (cbits_out) *cbits_out = cbits;
    xchg_deduped = true;
    return 0;
}

// deferred records of ALL ranks (they are rare): the ones this rank owns go in through the direct path
__global__ __launch_bounds__(256) void import3_owned_kernel(const unsigned long long *__restrict__ entries, uint64_t n, TableDev T, uint32_t nown, uint32_t self,
                                                            unsigned long long *__restrict__ histo_incomplete) {
    unsigned long long fresh = 0;
    bool any = false;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const u128 h = mk(entries[3 * i], entries[3 * i + 1]);
        if (owner_of(h, nown) != self) continue;
        any = true;
        fresh += table_add_or_spill(T, h, entries[3 * i + 2]);
    }
    if (any && histo_incomplete) *histo_incomplete = 1ull;       // counts changed after the fused histogram
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// d_recv / d_recv_cnt: block src = what rank src's xchg_partition put into ITS block `self`.  whole_input: these lists are
// everything that goes into this (empty) shard -> the multiplicity histogram is taken on the way out.
int Table::xchg_insert(const void *d_recv, const void *d_recv_cnt, uint64_t piece_max, uint64_t records_max, uint32_t nown, uint32_t self, const void *d_defer_all,
                       uint64_t n_defer_all, int whole_input, uint32_t slice_cap, int cbits, std::string &err) {
    HIPCHK(hipSetDevice(device));
    PartGeom G;
    int p2b = 0;
    if (!xchg_geometry(*this, piece_max, records_max, nown, G, p2b)) { err = "count exchange: no geometry for this table / piece size"; return -1; }
    if (read_stats(err)) return -1;
    const uint32_t nregions = 1u << (G.p1 + G.p2 + p2b);
    const bool empty = h_stats[ST_DISTINCT] == 0;
    histo_cached = false;
    unsigned long long *histo = whole_input && empty ? d_histo : nullptr;
    if (histo) HIPCHK(hipMemsetAsync(histo, 0, HISTO_WORDS * sizeof(unsigned long long), stream));
    // (a table that is logically empty is not read: the images start from zeros and every slot is written)
    const bool fresh = slots_dirty || histo != nullptr;
    const size_t lds = (size_t)(1u << G.rbits) * 16 + LDS_HBINS * 4 + (LI_MAXSL + 1) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(region_insert_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // this rank's own deferred list lives in the caller's buffer; the kernel's own overflow (probe beyond the halo) goes to a list of its own
    const uint64_t own_cap = std::max<uint64_t>(1u << 16, piece_max / 16);
    unsigned long long *defer = (unsigned long long *)workspace(WS_COUNT + 3, own_cap * 24 + 64, err);
    if (!defer) return -2;
    unsigned long long *defer_n = defer, *defer_e = defer + 8;
    HIPCHK(hipMemsetAsync(defer_n, 0, 64, stream));
    for (int i = 0; i <= N_STAGES; ++i) if (!ev_stage_t[i]) HIPCHK(hipEventCreate(&ev_stage_t[i]));
    count_path = 3;
    part_stage_n = 5;
    HIPCHK(hipEventRecord(ev_stage_t[3], stream));
    if (cbits && cbits > G.p2) { err = "count exchange: counted records do not fit this geometry"; return -1; }
    if (slice_cap) G.cap2 = slice_cap;                         // (lists packed by the caller after xchg_dedupe)
    const void *lists = d_recv;
    const unsigned int *lcnt = (const unsigned int *)d_recv_cnt;
    uint32_t lcap = G.cap2, nsl = nown * G.nblk2;
    PartGeom GI = G;                                           // what lds_insert sees: all second-level bits resolved
    GI.p2 = G.p2 + p2b;
    if (p2b) {
        // the senders resolved only G.p2 of the second-level bits: one more split pass here, over what arrived.  Its buckets are
        // the received lists (2^(p1 + G.p2) of them, nown * nblk2 slices each), its record bits the ones below those lists' bits
        PartGeom G2 = G;
        G2.p1 = G.p1 + G.p2; G2.p2 = p2b; G2.recbits = G.recbits - G.p2;
        G2.nblk1 = nown * G.nblk2; G2.cap1 = G.cap2; G2.nblk2 = 1;
        G2.cap2 = list_cap((double)std::max<uint64_t>(records_max, 1) / (double)nregions);
        uint64_t *out2 = (uint64_t *)workspace(WS_XCHG + 1, (size_t)nregions * G2.cap2 * 8, err);
        unsigned int *cnt2 = (unsigned int *)workspace(WS_XCHG + 2, ((size_t)nregions + 4) * 4, err);
        if (!out2 || !cnt2) return -2;
        static bool attr2_set = false;
        if (!attr2_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(part2_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr2_set = true;
        }
        hipLaunchKernelGGL((part2_kernel<false, true>), dim3(1, std::min<uint32_t>(1u << G2.p1, 2048)), dim3(PT_THREADS), P2_LDS, stream, (const uint64_t *)d_recv,
                           (const unsigned int *)d_recv_cnt, d, G2, out2, cnt2, defer_e, defer_n, own_cap, nown, cbits);
        HIPCHK(hipGetLastError());
        lists = out2; lcnt = cnt2; lcap = G2.cap2; nsl = 1;
    }
    {
        // one launch over all regions (region_insert_kernel: a probe that leaves its region takes the direct path afterwards)
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
        const uint32_t nblk = std::max<uint32_t>(1, std::min<uint32_t>(nregions, 256 * per_cu * 4));
        const uint32_t nsrc = p2b ? 1u : nown;
        if (fresh) hipLaunchKernelGGL((region_insert_kernel<true, true>), dim3(nblk), dim3(RI_TH), lds, stream, (const uint64_t *)lists, lcnt, lcap, nsl, d, GI, nregions, defer_e, defer_n,
                                      own_cap, histo, nsrc, cbits, G.p2);
        else hipLaunchKernelGGL((region_insert_kernel<false, true>), dim3(nblk), dim3(RI_TH), lds, stream, (const uint64_t *)lists, lcnt, lcap, nsl, d, GI, nregions, defer_e, defer_n,
                                own_cap, histo, nsrc, cbits, G.p2);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev_stage_t[4], stream));
        HIPCHK(hipEventRecord(ev_stage_t[5], stream));
    }
    slots_dirty = false;      // every region has been written by the launch above
    hipLaunchKernelGGL(import3h_kernel, dim3(256), dim3(256), 0, stream, defer_e, defer_n, own_cap, d, histo);      // (keeps the fused histogram exact)
    HIPCHK(hipGetLastError());
    if (n_defer_all) {
        hipLaunchKernelGGL(import3_owned_kernel, dim3(256), dim3(256), 0, stream, (const unsigned long long *)d_defer_all, n_defer_all, d, nown, self, histo ? histo + 10002 : nullptr);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ev_stage_t[6], stream));
    const int rc = after_batch(err);                           // (waits; spilled insertions / growth as on the other paths)
    if (rc) return rc;
    histo_cached = histo != nullptr;
    if (xchg_partitioned && xchg_deduped) {                    // the sender's dedupe pass counts as part of its second stage
        float m = 0;
        if (hipEventElapsedTime(&m, ev_stage_t[2], ev_stage_t[7]) == hipSuccess) { part_stage_ms[1] += m; count_kernel_ms += m; }
        else (void)hipGetLastError();
    }
    xchg_deduped = false;
    const int pairs[5][2] = {{0, 1}, {1, 2}, {3, 4}, {4, 5}, {5, 6}};
    for (int i = xchg_partitioned ? 0 : 2; i < 5; ++i) {       // (a shard that only received: no sender stages of its own)
        float m = 0;
        if (hipEventElapsedTime(&m, ev_stage_t[pairs[i][0]], ev_stage_t[pairs[i][1]]) == hipSuccess) { part_stage_ms[i] += m; count_kernel_ms += m; }
        else (void)hipGetLastError();
    }
    xchg_partitioned = false;
    count_launches += 1;
    ++count_partitioned_launches;
    return 0;
}

}  // namespace jk
