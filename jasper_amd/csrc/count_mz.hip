// count_mz.hip -- counting through minimizer super-k-mers: fewer, fatter records through the partition passes.
//
// count_part.hip moves one 8-byte record per k-mer OCCURRENCE through two LDS-sorted partition passes.  Consecutive k-mers of a
// read share their canonical minimizer (the smallest hashed canonical m-mer inside the k-mer) for ~12 positions; such a run --
// a super-k-mer -- is ONE 16-byte record holding its k-1+n bases 2-bit packed, and all its k-mers go to the same minimizer
// bucket whatever the strand they were read from.  The pipeline:
//
//   mz_part_kernel     bases -> hashed canonical m-mers (LDS) -> sliding minimum over the W = k-m+1 m-mers of each k-mer ->
//                      runs of equal bucket -> super-k-mer records, written straight into the block's own slice of each of
//                      the 2^pc coarse bucket lists (cursor = LDS atomic; 256 open 128-B lines per block stay in L2).
//   mz_split_count_kernel, mz_scan1/2, split16_kernel<0>
//                      coarse list -> 2^pf fine lists: count pass, scan, write pass (exact sizes, all records back to back;
//                      the write pass orders a tile by key inside LDS so that a list's line is filled by consecutive stores).
//   mz_count_kernel    a workgroup streams its share of that array: records are made UNIQUE in an LDS table first (a super-k-mer
//                      inside a read is cut out by the genome's minimizers, so every read covering the place yields the same
//                      16 bytes: 34 % of the records are left at 30x), then each unique record's k-mers are rolled out once,
//                      hashed with the table's mix() and added with the record's copies to an LDS k-mer table, which leaves as
//                      (hash, count) ENTRIES into the block's slice of each of 2^pe1 lists by top hash bits.  A k-mer that
//                      finds no room leaves as an entry of its own: entries are partial counts, the next stage adds them up.
//   split16_kernel<1>  entry list -> 2^pe2 region lists (region = 2^rbits consecutive table slots, as in count_part.hip).
//   lds_insert_kernel  (count_part.hip, entry form) region image in LDS <- entries, image written back, fused histogram.
//   Anything that overflows a slice goes to the deferred list (entries) or is expanded by mz_expand_kernel (records) and
//   takes the direct atomic path after the last image has been written; a call that overflows even those starts over on the
//   other counting path (Table::count_device).
//
// The table layout, tags and probe order are untouched: lookups, histogram, export, growth and the polisher do not know
// which path filled the table.  Semantics preserved: JF::include/jellyfish/mer_iterator.hpp:53-81 (which windows are
// counted, canonical = min(mer, revcomp)), JF::include/jellyfish/large_hash_array.hpp:291 (add 1 per occurrence).
// Where it stands: 19.8 ms for cfg 2 against 17.1 ms of count_part.hip (counting is bound by integer issue, not
// by the bytes this path saves), so k <= 37 keeps count_part.hip; 38 <= k <= 43, which count_part.hip cannot take, count here
// by default at 2.6x the rate of the direct kernel.
#include "table.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

// ---- geometry ---------------------------------------------------------------------------------------------------
struct MzGeom {
    int W, m, nmax;          // m-mers per k-mer (k - m + 1, a multiple of 4), minimizer length, k-mers per record at most
    int pc, pf;              // coarse / fine minimizer-bucket bits (bucket id = low pc+pf.. bits of the minimum, see bucket_of)
    int pe1, pe2, rbits;     // entry lists: top pe1 hash bits, next pe2 hash bits, region = 2^rbits slots (pe1+pe2+rbits == s)
    uint32_t nblkA, capA;    // mz_part: blocks (each owns a slice of capA records in every coarse list)
    uint32_t nsubP;          // mz_split: blocks per coarse bucket
    uint32_t nblkC, capC;    // mz_count: blocks (each owns a slice of capC entries in every hash-coarse list)
    uint32_t nsubE, capE;    // ent_split: blocks per hash-coarse list (each owns a slice of capE entries in every region list)
    uint64_t ovf_cap;        // records that did not fit a slice of mz_part
    int exp;                 // tuning experiments only (JASPER_MZ_EXP): 1 = mz_count hashes but does not insert
};

constexpr int MZ_TH = 512;                     // threads of mz_part_kernel; 16 positions each
constexpr int MZ_TILE = MZ_TH * 16;
constexpr int MZ_HW = 4;                       // halo words of 16 bases in front of a tile (64 >= k-1)
constexpr int MZ_PHW = 2;                      // halo words of hashed m-mers (32 >= W)
constexpr int MZ_MAXC = 256;                   // coarse lists (pc <= 8)

// A super-k-mer record, 128 bits:  [0,5) n-1   [5,15) fine bucket   [15, 15+2L) the L = k-1+n bases, first base in the
// most significant pair (k-mer i of the record = (V >> 2(n-1-i)) & kmask with V = record >> 15).  Needs k-1+n <= 56.
constexpr int MZ_VSH = 15;
constexpr int MZ_IDBITS = 18;                  // bucket id = the low 18 bits of the minimum, cut to pc + pf bits
__device__ __forceinline__ u128 mz_record(u128 V, uint32_t fine, int n) { return bor(shl(V, MZ_VSH), mk(0, ((uint64_t)fine << 5) | (uint64_t)(n - 1))); }
__device__ __forceinline__ u128 mz_bases(ulonglong2 r) { return mk(r.y >> MZ_VSH, (r.x >> MZ_VSH) | (r.y << (64 - MZ_VSH))); }

__device__ __forceinline__ uint32_t revpairs32(uint32_t x) {
    const uint32_t r = __brev(x);
    return ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
}
// order of the canonical m-mers: a multiply and a xor-shift (a bijection on 32 bits).  The minimum of W such values is
// small, but its LOW bits are still uniform (the density of the minimum is smooth at that scale), and they pick the bucket.
__device__ __forceinline__ uint32_t mz_phi(uint32_t canon) {
    uint32_t x = canon * 0x9E3779B1u;
    return x ^ (x >> 15);
}

// ---- A: bases -> super-k-mer records in 2^pc coarse lists --------------------------------------------------------------
template <bool WIDE, int W>
__global__ __launch_bounds__(MZ_TH) void mz_part_kernel(const uint8_t *__restrict__ bases, uint64_t n, uint64_t ntiles, uint64_t emit_from, TableDev T, MzGeom G,
                                                         ulonglong2 *__restrict__ outA, unsigned int *__restrict__ cntA,
                                                         ulonglong2 *__restrict__ ovf, unsigned long long *__restrict__ ovf_n) {
    __shared__ uint32_t s_code[MZ_TH + MZ_HW + 4];
    __shared__ uint32_t s_inv[MZ_TH + MZ_HW + 4];
    __shared__ __align__(16) uint32_t s_phi[(MZ_TH + MZ_PHW) * 16];
    __shared__ uint32_t s_bound[MZ_TH + 4];
    __shared__ uint32_t s_queue[MZ_TILE];
    __shared__ unsigned int s_cur[MZ_MAXC];
    __shared__ unsigned int s_qn;
    const int t = threadIdx.x;
    const int k = T.k, m = G.m;
    const int nb = 1 << G.pc;
    const int idsh = MZ_IDBITS - G.pc - G.pf;                    // bucket id = (min & (2^18 - 1)) >> idsh
    const uint32_t idmask = (1u << MZ_IDBITS) - 1u;
    const uint32_t fmask = (1u << G.pf) - 1u;
    const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    const int rcsh = 2 * (m - 1);
    unsigned long long added = 0;
    for (int i = t; i < nb; i += MZ_TH) s_cur[i] = 0;
    if (t < 4) { s_bound[MZ_TH + t] = 0xFFFFu; s_code[MZ_TH + MZ_HW + t] = 0; s_inv[MZ_TH + MZ_HW + t] = 0xFFFFu; }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base0 = (int64_t)(tile * MZ_TILE);
        __syncthreads();                                         // (previous tile's phase 4 is done with s_code / s_queue)
        {   // 1. my 16 bases as 2-bit codes + "no base" bits; the first threads also stage the halo
            uint32_t c, iv;
            stage16(bases, base0 + (int64_t)t * 16, (int64_t)n, c, iv);
            s_code[t + MZ_HW] = c;
            s_inv[t + MZ_HW] = iv;
            if (t < MZ_HW) {
                stage16(bases, base0 - (int64_t)(MZ_HW - t) * 16, (int64_t)n, c, iv);
                s_code[t] = c;
                s_inv[t] = iv;
            }
            if (t == 0) s_qn = 0;
        }
        __syncthreads();
        // 2. hashed canonical m-mer ENDING at every position of words MZ_HW-MZ_PHW .. MZ_HW+MZ_TH-1
        for (int v = t; v < MZ_TH + MZ_PHW; v += MZ_TH) {
            const int wv = MZ_HW - MZ_PHW + v;
            const uint32_t c = s_code[wv], prev = s_code[wv - 1];
            const uint32_t iv = s_inv[wv], ivp = s_inv[wv - 1];
            uint32_t f = m > 1 ? (prev & (mmask >> 2)) : 0u;                           // the m-1 bases before my first one
            uint32_t r = m > 1 ? ((revpairs32(~f) >> (32 - 2 * (m - 1))) << 2) : 0u;    // their reverse complement, one pair up
            int run = ivp ? (int)__builtin_ctz(ivp) : 16;
            uint32_t ph[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t cj = (c >> (30 - 2 * j)) & 3u;
                const bool bad = (iv >> (15 - j)) & 1u;
                f = ((f << 2) | cj) & mmask;
                r = (r >> 2) | ((3u - cj) << rcsh);
                run = bad ? 0 : run + 1;
                ph[j] = run >= m ? mz_phi(r < f ? r : f) : 0xFFFFFFFFu;
            }
            uint4 *dst = reinterpret_cast<uint4 *>(&s_phi[v * 16]);
            dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            dst[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
            dst[2] = make_uint4(ph[8], ph[9], ph[10], ph[11]);
            dst[3] = make_uint4(ph[12], ph[13], ph[14], ph[15]);
        }
        __syncthreads();
        // 3. minimum over the W m-mers of the k-mer ending at each of my 16 positions (and at the one before them), bucket ids,
        //    run boundaries; the starts of runs go into the tile's queue
        {
            uint32_t A[W], B[16];
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(&s_phi[(t + MZ_PHW) * 16 - W]);     // W is a multiple of 4
#pragma unroll
                for (int i = 0; i < W / 4; ++i) { const uint4 q = src[i]; A[4 * i] = q.x; A[4 * i + 1] = q.y; A[4 * i + 2] = q.z; A[4 * i + 3] = q.w; }
                const uint4 *sb = reinterpret_cast<const uint4 *>(&s_phi[(t + MZ_PHW) * 16]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const uint4 q = sb[i]; B[4 * i] = q.x; B[4 * i + 1] = q.y; B[4 * i + 2] = q.z; B[4 * i + 3] = q.w; }
            }
#pragma unroll
            for (int i = W - 2; i >= 0; --i) A[i] = A[i] < A[i + 1] ? A[i] : A[i + 1];      // A[i] = min(A[i..W-1])
            // validity of the k-mer ending at a position: no "no base" among its k bases (the 64 positions before mine are in
            // the four words in front of my word)
            const uint64_t ivprev = ((uint64_t)s_inv[t] << 48) | ((uint64_t)s_inv[t + 1] << 32) | ((uint64_t)s_inv[t + 2] << 16) | (uint64_t)s_inv[t + 3];
            const uint32_t iv = s_inv[t + MZ_HW];
            int run = ivprev ? (int)__builtin_ctzll(ivprev) : 64;
            const int64_t gpos0 = base0 + (int64_t)t * 16;
            bool prev_emit = run >= k && gpos0 - 1 >= (int64_t)emit_from;
            uint32_t prev_id = (A[0] & idmask) >> idsh;
            uint32_t boundmask = 0, startmask = 0;
            uint32_t ids[16];
            uint32_t pm = 0xFFFFFFFFu;                                                       // min(B[0..j]) when W > j
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t mn;
                if (j < W) {
                    pm = B[j] < pm ? B[j] : pm;
                    mn = pm;
                    if (j + 1 < W) mn = A[j + 1] < mn ? A[j + 1] : mn;
                } else {                                                                      // W <= 16: the window lies inside B
                    mn = B[j];
#pragma unroll
                    for (int u = 1; u < W; ++u) mn = B[j - u] < mn ? B[j - u] : mn;
                }
                const uint32_t id = (mn & idmask) >> idsh;
                ids[j] = id;
                const bool bad = (iv >> (15 - j)) & 1u;
                run = bad ? 0 : run + 1;
                const bool emit = run >= k && (uint64_t)(gpos0 + j) >= emit_from;
                const bool bound = !emit || !prev_emit || id != prev_id || (t == 0 && j == 0);
                boundmask |= (uint32_t)bound << j;
                startmask |= (uint32_t)(emit && bound) << j;
                added += emit ? 1u : 0u;
                prev_emit = emit;
                prev_id = id;
            }
            s_bound[t] = boundmask;
            // queue slots: a wave takes a range of the queue with one LDS atomic
            const int ns = __popc(startmask);
            int inc = ns;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
            unsigned int wbase = 0;
            if ((t & 63) == 63) wbase = atomicAdd(&s_qn, (unsigned int)inc);
            wbase = __shfl(wbase, 63);
            unsigned int qo = wbase + (unsigned int)(inc - ns);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((startmask >> j) & 1u) s_queue[qo++] = (uint32_t)(t * 16 + j) | (ids[j] << 13);
        }
        __syncthreads();
        // 4. one queue entry per lane: the run that starts there, cut into records of at most nmax k-mers
        const unsigned int qn = s_qn;
        for (unsigned int qi = t; qi < qn; qi += MZ_TH) {
            const uint32_t e = s_queue[qi];
            int cur = (int)(e & 0x1FFFu);
            const uint32_t id = e >> 13;
            const uint32_t coarse = id >> G.pf, fine = id & fmask;
            for (;;) {
                // distance to the next boundary after `cur` (positions beyond the tile read as boundaries)
                const int tc = cur >> 4, jc = cur & 15;
                const uint64_t look = ((uint64_t)(s_bound[tc] >> (jc + 1))) | ((uint64_t)s_bound[tc + 1] << (15 - jc)) | ((uint64_t)s_bound[tc + 2] << (31 - jc)) |
                                      ((uint64_t)s_bound[tc + 3] << (47 - jc));
                const int d = look ? (int)__builtin_ctzll(look) + 1 : 64;
                const int nk = d < G.nmax ? d : G.nmax;
                // the L = k-1+nk bases that end at position cur+nk-1, out of the staged codes
                const int P = MZ_HW * 16 + cur - (k - 1);
                const int wi = P >> 4, sh = 2 * (P & 15);
                const uint64_t hi64 = ((uint64_t)s_code[wi] << 32) | s_code[wi + 1], mid64 = ((uint64_t)s_code[wi + 2] << 32) | s_code[wi + 3];
                const uint64_t lo64 = (uint64_t)s_code[wi + 4] << 32;
                u128 top = mk(hi64, mid64);
                if (sh) top = mk((hi64 << sh) | (mid64 >> (64 - sh)), (mid64 << sh) | (lo64 >> (64 - sh)));
                const int L = k - 1 + nk;
                const u128 V = shr(top, 128 - 2 * L);
                const u128 rec = mz_record(V, fine, nk);
                const unsigned int slot = atomicAdd(&s_cur[coarse], 1u);
                if (slot < G.capA) outA[((uint64_t)coarse * G.nblkA + blockIdx.x) * G.capA + slot] = make_ulonglong2(rec.lo, rec.hi);
                else {
                    const unsigned long long oi = atomicAdd(ovf_n, 1ull);
                    if (oi < G.ovf_cap) ovf[oi] = make_ulonglong2(rec.lo, rec.hi);
                    else atomicExch(&T.stats[ST_FATAL], 1ull);
                }
                if (d <= G.nmax) break;
                cur += nk;                                    // the run goes on: same bucket, next record
            }
        }
    }
    __syncthreads();
    for (int i = t; i < nb; i += MZ_TH) cntA[(uint64_t)i * G.nblkA + blockIdx.x] = s_cur[i] < G.capA ? s_cur[i] : G.capA;
    for (int o = 32; o > 0; o >>= 1) added += __shfl_xor(added, o);
    if ((threadIdx.x & 63) == 0 && added) atomicAdd(&T.stats[ST_OCCURRENCES], added);
}

// ---- level 2 of the minimizer buckets: coarse list -> 2^pf fine lists, exact sizes ---------------------------------------
// grid (nsubP, 2^pc): block (x, c) reads slices x, x+nsubP, ... of coarse list c.  WRITE = 0: counts per fine key;
// WRITE = 1: records to base[..] + running cursor.
constexpr int SP_MAXSL = 512;            // slices a split block reads as one concatenated list
// count pass: grid (nsubP, 2^pc): block (x, c) reads slices x, x+nsubP, ... of coarse list c as ONE list (prefix of their
// lengths in LDS, so that all loads of the loop are independent) and counts its records per fine key
__global__ __launch_bounds__(256) void mz_split_count_kernel(const ulonglong2 *__restrict__ outA, const unsigned int *__restrict__ cntA, MzGeom G,
                                                              unsigned int *__restrict__ cntP) {
    __shared__ unsigned int s_cur[1024];
    __shared__ unsigned int s_pref[SP_MAXSL + 1];
    const int t = threadIdx.x;
    const uint32_t c = blockIdx.y, x = blockIdx.x;
    const int nf = 1 << G.pf;
    const uint32_t fmask = (uint32_t)nf - 1u;
    const uint64_t row = ((uint64_t)c * G.nsubP + x) << G.pf;
    for (int i = t; i < nf; i += 256) s_cur[i] = 0u;
    const uint32_t nmine = (G.nblkA - x + G.nsubP - 1) / G.nsubP;
    if (t < 64) {
        unsigned int carry = 0;
        for (uint32_t j0 = 0; j0 < nmine; j0 += 64) {
            const uint32_t j = j0 + t;
            const unsigned int v = j < nmine ? cntA[(uint64_t)c * G.nblkA + x + (uint64_t)j * G.nsubP] : 0u;
            unsigned int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if (t >= o) inc += u; }
            if (j < nmine) s_pref[j] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (t == 0) s_pref[nmine] = carry;
    }
    __syncthreads();
    const uint32_t total = s_pref[nmine];
    const ulonglong2 *src0 = outA + ((uint64_t)c * G.nblkA + x) * G.capA;      // slice x; slice x + j*nsubP is j*nsubP*capA further
    uint32_t sl[4] = {0, 0, 0, 0};
    for (uint32_t i0 = 0; i0 < total; i0 += 4 * 256) {
        unsigned long long lo[4];
        bool have[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256 + t;
            have[u] = i < total;
            lo[u] = 0ull;
            if (have[u]) {
                while (s_pref[sl[u] + 1] <= i) ++sl[u];
                lo[u] = src0[(uint64_t)sl[u] * G.nsubP * G.capA + (i - s_pref[sl[u]])].x;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (have[u]) atomicAdd(&s_cur[(uint32_t)(lo[u] >> 5) & fmask], 1u);
    }
    __syncthreads();
    for (int i = t; i < nf; i += 256) cntP[row + i] = s_cur[i];
}
// Exclusive scan of cntP in the order (coarse, fine, sub-block) -> baseP (same indexing as cntP) and the fine lists'
// boundaries startF[F], F = coarse << pf | fine, startF[NF] = total.  mz_scan1: one block per coarse bucket scans its
// 2^pf * nsubP cells (offsets inside the bucket) and leaves the bucket's total; mz_scan2: one block scans the totals and
// turns the offsets into positions.
__global__ __launch_bounds__(256) void mz_scan1_kernel(const unsigned int *__restrict__ cntP, unsigned int *__restrict__ baseP, unsigned int *__restrict__ ctot, MzGeom G) {
    __shared__ unsigned int s_w[4];
    __shared__ unsigned int s_carry;
    const int t = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t nf = 1u << G.pf;
    const uint32_t cells = nf * G.nsubP;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t o0 = 0; o0 < cells; o0 += 256) {               // cell o = fine * nsubP + x
        const uint32_t o = o0 + t;
        const uint64_t idx = (((uint64_t)c * G.nsubP + (o % G.nsubP)) << G.pf) + (o / G.nsubP);
        const unsigned int v = o < cells ? cntP[idx] : 0u;
        unsigned int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned int u = __shfl_up(inc, d); if ((t & 63) >= d) inc += u; }
        if ((t & 63) == 63) s_w[t >> 6] = inc;
        __syncthreads();
        unsigned int wbase = s_carry;
        for (int w = 0; w < (t >> 6); ++w) wbase += s_w[w];
        if (o < cells) baseP[idx] = wbase + inc - v;
        __syncthreads();
        if (t == 255) s_carry = wbase + inc;
        __syncthreads();
    }
    if (t == 0) ctot[c] = s_carry;
}
__global__ __launch_bounds__(256) void mz_scan2_kernel(unsigned int *__restrict__ baseP, unsigned int *__restrict__ ctot, unsigned int *__restrict__ startF, MzGeom G) {
    __shared__ unsigned int s_base[MZ_MAXC + 1];
    const int t = threadIdx.x;
    const uint32_t nc = 1u << G.pc, nf = 1u << G.pf;
    if (t == 0) {
        unsigned int run = 0;
        for (uint32_t c = 0; c < nc; ++c) { s_base[c] = run; run += ctot[c]; }
        s_base[nc] = run;
    }
    __syncthreads();
    const uint64_t cells = (uint64_t)nc * nf * G.nsubP;
    for (uint64_t i = t + (uint64_t)blockIdx.x * 256; i < cells; i += (uint64_t)gridDim.x * 256) {      // i = ((c * nsubP + x) << pf) + f
        const uint32_t c = (uint32_t)((i >> G.pf) / G.nsubP), x = (uint32_t)((i >> G.pf) % G.nsubP), f = (uint32_t)(i & (nf - 1));
        const unsigned int v = baseP[i] + s_base[c];
        baseP[i] = v;
        if (x == 0) startF[((uint64_t)c << G.pf) + f] = v;
    }
    if (blockIdx.x == 0 && t == 0) startF[(uint64_t)nc << G.pf] = s_base[nc];
}

// ---- C: records in fine-bucket order -> unique records -> LDS k-mer table -> (hash, count) entries -------------------------------
// An entry is 16 bytes: { hash.lo, hash.hi | count << 32 }  (hash.hi < 2^22 for k <= 43, count <= 2^32-1).
// The records of ALL fine buckets lie back to back (mz_split), so a block simply takes its 1/nblkC of that array -- a few
// hundred consecutive buckets -- and streams it in rounds of MC_TH records.  Nothing depends on where a bucket ends: a bucket
// cut by a flush (or by the end of a block's share) leaves some of its keys as two entries, and entries are partial counts.
//
// What bounds this kernel is integer issue (rolling both strands of a two-word k-mer, the canonical choice and the table's
// 2 x 64-bit-multiply hash are ~100 issue slots per k-mer; an LDS insert whose probe loop runs as long as the slowest of 64
// lanes costs more than that again -- DESIGN.md 4.2), so the k-mers are not touched until the RECORDS have been made unique:
// a super-k-mer that lies inside a read is cut out of the genome by its minimizers, not by the read, so every read that
// covers the place yields the same 16 bytes (or their reverse complement: records are stored in the smaller orientation) --
// ~20 copies at 30x coverage; only the records at the two ends of a read and those with a read error are on their own.
//   round:  one record per lane -> canonical orientation -> record table RT (LDS, 512 slots: 16-byte key, 32-bit copies)
//   flush (RT has MC_RT_FLUSH unique records, or the input ends):  every unique record's k-mers are rolled out ONCE, hashed,
//           and added to the k-mer table KT (LDS, 4096 slots) with the record's number of copies; KT leaves as entries.
constexpr int MC_TH = 512;
constexpr int MC_SLOTS = 4096;                 // KT: 16 bytes per slot
constexpr int MC_KTBITS = 12;
constexpr int MC_STEPS = 10;                   // KT slot reads per k-mer at most; then it leaves as a (hash, copies) entry
constexpr int MC_PIECE = 8;                    // k-mers a lane rolls out of a record in one go
constexpr int MC_RT = 512;                     // RT slots
constexpr int MC_RT_FLUSH = 280;               // unique records that trigger a flush at the end of a round
constexpr int MC_RT_STEPS = 24;                // RT slot reads per record at most; then the record waits for the next flush
constexpr int MC_MAXE = 256;                   // hash-coarse lists (pe1 <= 8)

// bits [sh, sh+32) of the hash (hhi : hlo), 1 <= sh
__device__ __forceinline__ uint32_t top_bits(uint64_t hhi, uint64_t hlo, int sh) {
    return sh >= 64 ? (uint32_t)(hhi >> (sh - 64)) : (uint32_t)((hhi << (64 - sh)) | (hlo >> sh));
}

template <bool WIDE>
__global__ __launch_bounds__(MC_TH) void mz_count_kernel(const ulonglong2 *__restrict__ lists, const unsigned int *__restrict__ total_ptr, TableDev T, MzGeom G,
                                                          ulonglong2 *__restrict__ outC, unsigned int *__restrict__ cntC, unsigned long long *__restrict__ deferred,
                                                          unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    // KT slot = { w0 = hash.lo, w1 = (hash.hi + 1) << 32 | count } ; w1 == 0: empty.  A slot is claimed by a compare-and-swap
    // on w1 (which also counts the claimant), w0 follows with a plain store; a lane that meets the slot in between sees
    // w0 == 0 and looks again.  RT slot = { r0 = record.lo, r1 = record.hi + 1 } claimed through r1 the same way.
    __shared__ __align__(16) unsigned long long s_kt[2 * MC_SLOTS];
    __shared__ __align__(16) unsigned long long s_rt[2 * MC_RT];
    __shared__ unsigned int s_rtc[MC_RT];                      // copies of the record in RT slot i
    __shared__ unsigned short s_work[MC_RT * 4];               // (RT slot << 2) | piece
    __shared__ unsigned int s_cur[MC_MAXE];
    __shared__ unsigned int s_w[MC_TH / 64];
    __shared__ unsigned int s_total, s_claims, s_pending;
    const int t = threadIdx.x;
    const int k = T.k, B = T.B;
    const int ne = 1 << G.pe1;
    const u128 kmask = maskbits(2 * k);
    const int hb = WIDE ? B - 64 : 0;
    const uint64_t himask = WIDE ? (hb == 64 ? ~0ull : ((1ull << hb) - 1ull)) : 0ull;
    const int rcins_w = 2 * (k - 1) - 64, rcins_n = 2 * (k - 1);
    const uint64_t lomask = (!WIDE && 2 * k < 64) ? ((1ull << (2 * k)) - 1ull) : ~0ull;
    const int e1sh = B - G.pe1, idxsh = B - MC_KTBITS;
    for (int i = t; i < ne; i += MC_TH) s_cur[i] = 0;
    for (int i = t; i < 2 * MC_SLOTS; i += MC_TH) s_kt[i] = 0ull;
    for (int i = t; i < 2 * MC_RT; i += MC_TH) s_rt[i] = 0ull;
    for (int i = t; i < MC_RT; i += MC_TH) s_rtc[i] = 0u;
    if (t == 0) { s_claims = 0; s_pending = 0; }
    // an entry leaves through my slice of the list its top hash bits select, or through the deferred list when that is full
    auto emit_entry = [&](uint64_t hhi, uint64_t hlo, unsigned long long cnt) {
        const uint32_t e1 = top_bits(hhi, hlo, e1sh);                // top pe1 hash bits
        const unsigned int pos = atomicAdd(&s_cur[e1], 1u);
        if (pos < G.capC) outC[((uint64_t)e1 * G.nblkC + blockIdx.x) * G.capC + pos] = make_ulonglong2(hlo, hhi | (cnt << 32));
        else {
            const unsigned long long di = atomicAdd(deferred_n, 1ull);
            if (di < deferred_cap) { deferred[3 * di] = hhi; deferred[3 * di + 1] = hlo; deferred[3 * di + 2] = cnt; }
            else atomicExch(&T.stats[ST_FATAL], 1ull);
        }
    };
    // the k-mers i0 .. i1-1 of a record, each added `copies` times to KT
    auto roll_out = [&](ulonglong2 rr, int i0, int i1, unsigned int copies) {
        const int nk = (int)(rr.x & 31ull) + 1;
        const u128 V = mz_bases(rr);
        const u128 rest = shr(V, 2 * (nk - i1));                   // the bases up to the end of my last k-mer
        const u128 fwd = band(shr(rest, 2 * (i1 - 1 - i0)), kmask);
        const u128 rc = revcomp(fwd, k);
        uint64_t fl = fwd.lo, fh = fwd.hi, rl = rc.lo, rh = rc.hi;
        const uint32_t tail = (uint32_t)rest.lo;                   // the (i1 - 1 - i0) <= 7 bases after my first k-mer are its low bits
        for (int i = i0; i < i1; ++i) {
            if (i > i0) {
                const uint64_t cj = (tail >> (2 * (i1 - 1 - i))) & 3u;
                if (WIDE) {
                    fh = ((fh << 2) | (fl >> 62)) & himask;
                    fl = (fl << 2) | cj;
                    rl = (rl >> 2) | (rh << 62);
                    rh = (rh >> 2) | ((3ull - cj) << rcins_w);
                } else {
                    fl = ((fl << 2) | cj) & lomask;
                    rl = (rl >> 2) | ((3ull - cj) << rcins_n);
                }
            }
            uint64_t hhi, hlo;
            if (WIDE) {
                const bool take_rc = rh < fh || (rh == fh && rl < fl);
                const uint64_t mh = take_rc ? rh : fh, ml = take_rc ? rl : fl;
                hlo = mix64(ml);                                 // = mix() for 2k > 64 (kmer.hpp)
                hhi = (mh ^ rotr64(hlo, 30)) & himask;
            } else {
                const u128 h = mix(mk(0, rl < fl ? rl : fl), B);
                hhi = 0; hlo = h.lo;
            }
            // KT: home = top 12 hash bits; MC_STEPS bounds probes and second looks together, so nothing here can spin
            const uint32_t want_hi = (uint32_t)hhi + 1u;
            uint32_t idx = top_bits(hhi, hlo, idxsh) & (MC_SLOTS - 1);
            bool done = false;
            if (hlo != 0ull) {                                   // (w0 == 0 is what a slot shows before its key is written)
#pragma unroll 1
                for (int st = 0; st < MC_STEPS; ++st) {
                    const ulonglong2 sl = *reinterpret_cast<const ulonglong2 *>(&s_kt[2 * idx]);
                    if (sl.y == 0ull) {
                        if (atomicCAS(&s_kt[2 * idx + 1], 0ull, ((unsigned long long)want_hi << 32) | copies) == 0ull) {     // claims the slot with my copies
                            s_kt[2 * idx] = hlo;
                            done = true;
                            break;
                        }
                        continue;                                // somebody else has just claimed it: look again
                    }
                    if ((uint32_t)(sl.y >> 32) == want_hi) {
                        if (sl.x == hlo) {
                            atomicAdd(reinterpret_cast<unsigned int *>(&s_kt[2 * idx + 1]), copies);      // the count is the low half of w1
                            done = true;
                            break;
                        }
                        if (sl.x == 0ull) continue;              // claimed, key not visible yet: look again
                    }
                    idx = (idx + 1) & (MC_SLOTS - 1);
                }
            }
            if (!done) emit_entry(hhi, hlo, copies);
        }
    };
    unsigned long long dbg_rolled = 0, dbg_unique = 0;
    const uint64_t total = *total_ptr;
    const uint64_t lo = total * blockIdx.x / gridDim.x, hi = total * (blockIdx.x + 1) / gridDim.x;
    ulonglong2 rec_next = make_ulonglong2(0ull, 0ull);
    if (lo + t < hi) rec_next = lists[lo + t];
    __syncthreads();
    for (uint64_t r0 = lo; r0 < hi || r0 == lo; r0 += MC_TH) {
        // ---- a round: my record, in its smaller orientation, into RT
        ulonglong2 rec = rec_next;
        bool have = r0 + t < hi;
        rec_next = make_ulonglong2(0ull, 0ull);
        if (r0 + MC_TH + t < hi) rec_next = lists[r0 + MC_TH + t];       // in flight while this round is worked on
        if (have) {
            const int nk = (int)(rec.x & 31ull) + 1;
            const int L = k - 1 + nk;
            const u128 V = mz_bases(rec);
            const u128 R = revcomp(V, L);
            if (lt(R, V)) {
                const u128 nr = bor(shl(R, MZ_VSH), mk(0, rec.x & ((1ull << MZ_VSH) - 1ull)));
                rec = make_ulonglong2(nr.lo, nr.hi);
            }
        }
        bool pending = have;
        for (;;) {      // (normally one trip; more only when RT was full: every flush empties it, so the records left over get in)
            unsigned int claims = 0;
            if (pending) {
                uint32_t x = (uint32_t)rec.x ^ (uint32_t)(rec.x >> 32) ^ (uint32_t)rec.y ^ (uint32_t)(rec.y >> 32);
                x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u;
                uint32_t idx = (x >> 16) & (MC_RT - 1);
                const unsigned long long want1 = rec.y + 1ull;
#pragma unroll 1
                for (int st = 0; st < MC_RT_STEPS; ++st) {
                    const ulonglong2 sl = *reinterpret_cast<const ulonglong2 *>(&s_rt[2 * idx]);
                    if (sl.y == 0ull) {
                        if (atomicCAS(&s_rt[2 * idx + 1], 0ull, want1) == 0ull) {       // claimed through the high word (+1: never 0)
                            s_rt[2 * idx] = rec.x;
                            atomicAdd(&s_rtc[idx], 1u);
                            ++claims;
                            pending = false;
                            break;
                        }
                        continue;
                    }
                    if (sl.y == want1) {
                        if (sl.x == rec.x) { atomicAdd(&s_rtc[idx], 1u); pending = false; break; }
                        if (sl.x == 0ull) continue;              // claimed just now, low word not visible yet (or really another record: MC_RT_STEPS bounds the looking)
                    }
                    idx = (idx + 1) & (MC_RT - 1);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) claims += __shfl_xor(claims, o);
            if ((t & 63) == 0 && claims) atomicAdd(&s_claims, claims);
            if (__ballot(pending) != 0ull && (t & 63) == 0) atomicOr(&s_pending, 1u);
            __syncthreads();
            const bool last_round = r0 + MC_TH >= hi;
            const bool flush = s_claims >= (unsigned)MC_RT_FLUSH || s_pending != 0u || last_round;
            if (!flush) break;                                      // (the same answer in every thread)
            // ---- flush: unique records -> pieces of MC_PIECE k-mers (block prefix sum of the piece counts) -> KT -> entries
            {
                ulonglong2 ur = make_ulonglong2(0ull, 0ull);
                if (t < MC_RT) ur = *reinterpret_cast<const ulonglong2 *>(&s_rt[2 * t]);         // RT slot t
                const unsigned int np = ur.y != 0ull ? (((unsigned int)(ur.x & 31ull) + 1u) + MC_PIECE - 1) / MC_PIECE : 0u;
                unsigned int inc = np;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
                if ((t & 63) == 63) s_w[t >> 6] = inc;
                __syncthreads();
                unsigned int wbase = 0;
                for (int w = 0; w < (t >> 6); ++w) wbase += s_w[w];
                const unsigned int o = wbase + inc - np;
                for (unsigned int q = 0; q < np; ++q) s_work[o + q] = (unsigned short)((t << 2) | q);
                if (t == MC_TH - 1) s_total = wbase + inc;
                __syncthreads();
                const unsigned int nwork = s_total;
                for (unsigned int wi = t; wi < nwork; wi += MC_TH) {
                    const unsigned int w = s_work[wi];
                    const unsigned int slot = w >> 2;
                    const ulonglong2 sl = *reinterpret_cast<const ulonglong2 *>(&s_rt[2 * slot]);
                    const ulonglong2 rr = make_ulonglong2(sl.x, sl.y - 1ull);
                    const int nk = (int)(rr.x & 31ull) + 1;
                    const int i0 = (int)(w & 3u) * MC_PIECE;
                    roll_out(rr, i0, i0 + MC_PIECE < nk ? i0 + MC_PIECE : nk, s_rtc[slot]);
                    dbg_rolled += (unsigned long long)((i0 + MC_PIECE < nk ? i0 + MC_PIECE : nk) - i0);
                    dbg_unique += (w & 3u) == 0u;
                }
                __syncthreads();
                for (int i = t; i < MC_SLOTS; i += MC_TH) {
                    const ulonglong2 sl = *reinterpret_cast<const ulonglong2 *>(&s_kt[2 * i]);
                    if (sl.y != 0ull) {
                        emit_entry((sl.y >> 32) - 1ull, sl.x, sl.y & 0xFFFFFFFFull);
                        *reinterpret_cast<ulonglong2 *>(&s_kt[2 * i]) = make_ulonglong2(0ull, 0ull);
                    }
                }
                if (t < MC_RT) { *reinterpret_cast<ulonglong2 *>(&s_rt[2 * t]) = make_ulonglong2(0ull, 0ull); s_rtc[t] = 0u; }
                if (t == 0) { s_claims = 0; s_pending = 0; }
                __syncthreads();
            }
            if (__syncthreads_or(pending ? 1 : 0) == 0) break;      // records that found RT full go into the empty one now
        }
        if (r0 + MC_TH >= hi) break;
    }
    __syncthreads();
    for (int i = t; i < ne; i += MC_TH) cntC[(uint64_t)i * G.nblkC + blockIdx.x] = s_cur[i] < G.capC ? s_cur[i] : G.capC;
    if (G.exp == 9) {     // (tuning: how many k-mers were rolled out, how many unique records)
        for (int o = 32; o > 0; o >>= 1) { dbg_rolled += __shfl_xor(dbg_rolled, o); dbg_unique += __shfl_xor(dbg_unique, o); }
        if ((t & 63) == 0) { atomicAdd(&T.stats[5], dbg_rolled); atomicAdd(&T.stats[6], dbg_unique); }
    }
}

// ---- both second-level splits, with the copy-out staged through LDS ---------------------------------------------------------
// Writing every 16-byte record straight to its list (as mz_part can: 256 lists) leaves each of the block's up-to-1024 open
// 128-byte lines to be filled by eight visits spread over the whole run of the block, and with ~2000 workgroups resident that
// is far more open lines than the L2s hold: the lines leave half filled (such a write pass took 1.7 ms where this one takes
// 1.1, the entry split 2.3 instead of 1.7).  Here a tile of ST_TILE records is first ordered by key inside LDS (rank from a returning LDS atomic, offsets
// from a block scan -- part2_kernel's scheme), and lane t then writes all the tile's records of key t one after the other:
// a list's line is filled by consecutive stores within a few hundred cycles.
// grid (nsub, lists): block (x, b) reads slices x, x+nsub, ... of list b as one concatenated list.
// KIND 0: super-k-mer records, key = fine bucket field, destination = exact positions (baseP, from the count pass);
// KIND 1: (hash, count) entries, key = next pe2 hash bits, destination = the block's slice of each region list (capE, deferred list).
constexpr int ST_TH = 1024, ST_TILE = 4096;
template <int KIND>
__global__ __launch_bounds__(ST_TH) void split16_kernel(const ulonglong2 *__restrict__ src_lists, const unsigned int *__restrict__ src_cnt, TableDev T, MzGeom G,
                                                         const unsigned int *__restrict__ baseP, ulonglong2 *__restrict__ out, unsigned int *__restrict__ out_cnt,
                                                         unsigned long long *__restrict__ deferred, unsigned long long *__restrict__ deferred_n, uint64_t deferred_cap) {
    __shared__ __align__(16) ulonglong2 s_stage[ST_TILE];
    __shared__ unsigned int s_cnt[1024], s_off[1024], s_cur[1024];
    __shared__ unsigned int s_pref[SP_MAXSL + 1];
    __shared__ unsigned int s_w[ST_TH / 64];
    const int t = threadIdx.x;
    const uint32_t bkt = blockIdx.y, x = blockIdx.x;             // coarse bucket (KIND 0) / hash-coarse list (KIND 1), sub-block
    const int nkeys = 1 << (KIND == 0 ? G.pf : G.pe2);
    const uint32_t kmask = (uint32_t)nkeys - 1u;
    const uint32_t nsrc = KIND == 0 ? G.nblkA : G.nblkC, nsub = KIND == 0 ? G.nsubP : G.nsubE, scap = KIND == 0 ? G.capA : G.capC;
    const int sh = T.B - G.pe1 - G.pe2;
    const uint64_t row = ((uint64_t)bkt * nsub + x) << G.pf;     // (KIND 0)
    if (t < nkeys) { s_cur[t] = KIND == 0 ? baseP[row + t] : 0u; s_cnt[t] = 0u; }
    const uint32_t nmine = (nsrc - x + nsub - 1) / nsub;          // slices x, x+nsub, ... as ONE list
    if (t < 64) {
        unsigned int carry = 0;
        for (uint32_t j0 = 0; j0 < nmine; j0 += 64) {
            const uint32_t j = j0 + t;
            const unsigned int v = j < nmine ? src_cnt[(uint64_t)bkt * nsrc + x + (uint64_t)j * nsub] : 0u;
            unsigned int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if (t >= o) inc += u; }
            if (j < nmine) s_pref[j] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (t == 0) s_pref[nmine] = carry;
    }
    __syncthreads();
    const uint32_t total = s_pref[nmine];
    const ulonglong2 *src0 = src_lists + ((uint64_t)bkt * nsrc + x) * scap;
    uint32_t sl[4] = {0, 0, 0, 0};
    for (uint32_t i0 = 0; i0 < total; i0 += ST_TILE) {
        ulonglong2 r[4];
        uint32_t kr[4];                                           // key << 16 | rank in the tile ; 0xFFFFFFFF = nothing
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + (uint32_t)u * ST_TH + t;
            kr[u] = 0xFFFFFFFFu;
            r[u] = make_ulonglong2(0ull, 0ull);
            if (i < total) {
                while (s_pref[sl[u] + 1] <= i) ++sl[u];
                r[u] = src0[(uint64_t)sl[u] * nsub * scap + (i - s_pref[sl[u]])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + (uint32_t)u * ST_TH + t >= total) continue;
            const uint32_t key = KIND == 0 ? ((uint32_t)(r[u].x >> 5) & kmask) : (top_bits(r[u].y & 0xFFFFFFFFull, r[u].x, sh) & kmask);
            kr[u] = (key << 16) | (atomicAdd(&s_cnt[key], 1u) & 0xFFFFu);
        }
        __syncthreads();
        {   // exclusive prefix of the key counts: thread t owns key t
            const unsigned int v = t < nkeys ? s_cnt[t] : 0u;
            unsigned int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
            if ((t & 63) == 63) s_w[t >> 6] = inc;
            __syncthreads();
            unsigned int wbase = 0;
            for (int w = 0; w < (t >> 6); ++w) wbase += s_w[w];
            if (t < nkeys) s_off[t] = wbase + inc - v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (kr[u] != 0xFFFFFFFFu) s_stage[s_off[kr[u] >> 16] + (kr[u] & 0xFFFFu)] = r[u];
        __syncthreads();
        if (t < nkeys) {                                          // lane t: the tile's records of key t, one after the other
            const unsigned int cnt = s_cnt[t], off = s_off[t], cur = s_cur[t];
            if (KIND == 0) {
                for (unsigned int q = 0; q < cnt; ++q) out[cur + q] = s_stage[off + q];
            } else {
                const uint64_t region = ((uint64_t)bkt << G.pe2) + (uint32_t)t;
                ulonglong2 *dst = out + (region * nsub + x) * G.capE;
                for (unsigned int q = 0; q < cnt; ++q) {
                    const ulonglong2 e = s_stage[off + q];
                    if (cur + q < G.capE) dst[cur + q] = e;
                    else {
                        const unsigned long long di = atomicAdd(deferred_n, 1ull);
                        if (di < deferred_cap) { deferred[3 * di] = e.y & 0xFFFFFFFFull; deferred[3 * di + 1] = e.x; deferred[3 * di + 2] = e.y >> 32; }
                        else atomicExch(&T.stats[ST_FATAL], 1ull);
                    }
                }
            }
            s_cur[t] = cur + cnt;
            s_cnt[t] = 0u;
        }
        __syncthreads();
    }
    if (KIND == 1 && t < nkeys) out_cnt[((((uint64_t)bkt << G.pe2) + (uint32_t)t) * nsub) + x] = s_cur[t] < G.capE ? s_cur[t] : G.capE;
}

// ---- records that overflowed a slice of mz_part: every k-mer through the direct path --------------------------------------
__global__ __launch_bounds__(256) void mz_expand_kernel(const ulonglong2 *__restrict__ ovf, const unsigned long long *__restrict__ ovf_n, uint64_t cap, TableDev T,
                                                         unsigned long long *__restrict__ histo_incomplete) {
    const uint64_t n = *ovf_n < cap ? *ovf_n : cap;
    if (histo_incomplete && n != 0ull && blockIdx.x == 0 && threadIdx.x == 0) *histo_incomplete = 1ull;
    const int k = T.k;
    const u128 kmask = maskbits(2 * k);
    unsigned long long fresh = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 rr = ovf[i];
        const int nk = (int)(rr.x & 31ull) + 1;
        const u128 V = mz_bases(rr);
        for (int j = 0; j < nk; ++j) {
            const u128 m = band(shr(V, 2 * (nk - 1 - j)), kmask);
            fresh += table_add_or_spill(T, mix(canonical(m, k), T.B), 1ull);
        }
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

// ==================================================================================================================
static uint32_t slice_cap(double avg, double slack) { return (uint32_t)std::min<double>(4.0e9, avg * slack + 8.0 * std::sqrt(avg) + 64.0); }

bool Table::minimizer_geometry(uint64_t piece_bases, void *geom_out) const {
    MzGeom &G = *reinterpret_cast<MzGeom *>(geom_out);
    // JASPER_COUNT_PATH: 1 (default) = count_part.hip, 2 = this file.  Measured on MI355X (profiles/round2): this
    // path moves 7x fewer bytes but executes MORE vector instructions than count_part.hip (mz_count alone 5.1 G wave
    // instructions against 4.4 G for part1 + part2 + lds_insert together) and is slower (26 ms against 17 ms for cfg 2), so it
    // is kept as the second, parity-tested way to fill a table and as the starting point of a bucket-addressed table.
    // For 38 <= k <= 43 count_part.hip cannot run (its level-1 record holds 64 hash bits below 10 bucket bits) and this path is
    // the atomic-free one: ~50 Gk-mers/s against ~18 of the direct kernel.
    const int mode = getenv("JASPER_COUNT_PATH") ? atoi(getenv("JASPER_COUNT_PATH")) : (k >= 38 ? 2 : 1);
    if (mode != 2 || getenv("JASPER_COUNT_DIRECT") || d.ext || mz_off) return false;
    if (piece_bases < (8u << 20) || piece_bases >= (1ull << 32)) return false;
    const int B = d.B, s = d.s;
    if (k < 15 || k > 43 || s < 16) return false;
    // minimizer length: 10..16 with W = k - m + 1 a multiple of 4 (the sliding minimum reads its window with 16-byte LDS loads),
    // the longest such m that keeps W >= 4
    int m = 0;
    for (int cand = 16; cand >= 10; --cand) { const int w = k - cand + 1; if (w >= 4 && w <= 32 && w % 4 == 0) { m = cand; break; } }
    if (!m) return false;
    G.m = m; G.W = k - m + 1;
    G.nmax = std::min(32, 57 - k);
    if (G.nmax < 4) return false;
    // minimizer buckets: ~4 K k-mer occurrences per fine bucket (at 30x coverage with 0.3 % read errors ~600 distinct k-mers):
    // mz_count streams the buckets through a 4096-slot table and empties it every few buckets; the finer the buckets, the
    // fewer keys are cut in two by a flush
    int nbits = 0;
    while (nbits < MZ_IDBITS && (piece_bases >> nbits) > 6144) ++nbits;
    if (const char *e = getenv("JASPER_MZ_TEST_NBITS")) nbits = std::min(nbits, std::max(8, atoi(e)));   // experiments: coarser buckets
    G.pc = std::min(8, nbits);
    G.pf = std::min(10, nbits - G.pc);
    G.rbits = std::min(13, s);
    const int eb = s - G.rbits;
    G.pe1 = std::min(8, eb);
    G.pe2 = eb - G.pe1;
    if (G.pe2 > 10 || B - 12 < 0) return false;      // (a table beyond 2^31 slots: count_part.hip / the direct kernel)
    const uint64_t ntiles = (piece_bases + MZ_TILE - 1) / MZ_TILE;
    G.nblkA = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 512));
    // records: one per ~min(W, nmax)/2 k-mers on random sequence; sized for 1 per 4 (shorter runs overflow into mz_expand)
    const double recs = (double)piece_bases / 4.0;
    G.capA = slice_cap(recs / ((double)(1u << G.pc) * G.nblkA), 1.3);
    G.nsubP = 8;                            // (a split block reads nblkA / nsubP <= 512 slices as one list)
    G.ovf_cap = std::max<uint64_t>(1u << 16, piece_bases / 64);
    // entries: between the distinct k-mers of the piece and its occurrences.  The share of new keys measured on the previous
    // piece (or promised by the size hint) x2 covers the partial counts of crowded buckets; without either, half the occurrences.
    double frac = 0.5;
    if (dup_ratio < 0.6) frac = std::min(0.6, std::max(0.12, 2.0 * dup_ratio));
    const double ents = (double)piece_bases * frac;
    G.nblkC = 512;
    G.capC = slice_cap(ents / ((double)(1u << G.pe1) * G.nblkC), 1.6);      // (blocks take buckets from a queue: uneven shares)
    G.nsubE = 2;
    G.exp = getenv("JASPER_MZ_EXP") ? atoi(getenv("JASPER_MZ_EXP")) : 0;
    G.capE = slice_cap(ents / ((double)(1ull << (G.pe1 + G.pe2)) * G.nsubE), 1.3);
    if (const char *e = getenv("JASPER_MZ_TEST_CAPS")) {      // tests: "a:c:e" = factors on the three slice capacities, so that slices overflow
        double fa = 1, fc = 1, fe = 1;
        if (sscanf(e, "%lf:%lf:%lf", &fa, &fc, &fe) == 3) {
            G.capA = std::max<uint32_t>(8, (uint32_t)(G.capA * fa));
            G.capC = std::max<uint32_t>(8, (uint32_t)(G.capC * fc));
            G.capE = std::max<uint32_t>(8, (uint32_t)(G.capE * fe));
        }
    }
    return true;
}

int Table::launch_count_minimizer(const uint8_t *d_piece, uint64_t len, uint64_t emit_from, const void *geom, std::string &err) {
    const MzGeom G = *reinterpret_cast<const MzGeom *>(geom);
    const uint32_t nc = 1u << G.pc, NF = 1u << (G.pc + G.pf), ne1 = 1u << G.pe1, nregions = 1u << (G.pe1 + G.pe2);
    const uint64_t deferred_cap = std::max<uint64_t>(1u << 16, len / 16);
    const size_t bytesA = (size_t)nc * G.nblkA * G.capA * 16, bytesC = (size_t)ne1 * G.nblkC * G.capC * 16;
    const size_t bytes2 = (size_t)nc * G.nblkA * G.capA * 16, bytesE = (size_t)nregions * G.nsubE * G.capE * 16;
    ulonglong2 *bufX = (ulonglong2 *)workspace(WS_COUNT + 0, std::max(bytesA, bytesC), err);        // mz_part's lists, later mz_count's
    ulonglong2 *bufY = (ulonglong2 *)workspace(WS_COUNT + 1, std::max(bytes2, bytesE), err);        // fine lists, later region lists
    const size_t n_cntA = (size_t)nc * G.nblkA, n_cntP = (size_t)NF * G.nsubP, n_cntC = (size_t)ne1 * G.nblkC, n_cntE = (size_t)nregions * G.nsubE;
    unsigned int *cur = (unsigned int *)workspace(WS_COUNT + 2, (n_cntA + 2 * n_cntP + (NF + 1) + n_cntC + n_cntE + 16 + MZ_MAXC) * 4, err);
    unsigned long long *defer = (unsigned long long *)workspace(WS_COUNT + 3, deferred_cap * 24 + 64, err);
    ulonglong2 *ovf = (ulonglong2 *)workspace(WS_MZ + 0, G.ovf_cap * 16 + 64, err);
    if (!bufX || !bufY || !cur || !defer || !ovf) return -2;
    unsigned int *cntA = cur, *cntP = cntA + n_cntA, *baseP = cntP + n_cntP, *startF = baseP + n_cntP, *cntC = startF + NF + 1, *cntE = cntC + n_cntC;
    unsigned int *next_bucket = cntE + n_cntE, *ctot = next_bucket + 16;
    unsigned long long *defer_n = defer, *defer_e = defer + 8;       // first 8 bytes: counter; [1]: overflow-record counter; entries start 64 bytes in
    unsigned long long *ovf_n = defer + 1;
    HIPCHK(hipMemsetAsync(defer_n, 0, 64, stream));
    HIPCHK(hipMemsetAsync(next_bucket, 0, 16, stream));
    for (int i = 0; i < 9; ++i) if (!ev_stage_t[i]) HIPCHK(hipEventCreate(&ev_stage_t[i]));
    HIPCHK(hipEventRecord(ev_k0, stream));
    HIPCHK(hipEventRecord(ev_stage_t[0], stream));
    const uint64_t ntiles = (len + MZ_TILE - 1) / MZ_TILE;
#define JK_A(WD, WW) hipLaunchKernelGGL((mz_part_kernel<WD, WW>), dim3(G.nblkA), dim3(MZ_TH), 0, stream, d_piece, len, ntiles, emit_from, d, G, bufX, cntA, ovf, ovf_n)
#define JK_AW(WW) do { if (k > 32) JK_A(true, WW); else JK_A(false, WW); } while (0)
    switch (G.W) {
        case 4: JK_AW(4); break;   case 8: JK_AW(8); break;   case 12: JK_AW(12); break; case 16: JK_AW(16); break;
        case 20: JK_AW(20); break; case 24: JK_AW(24); break; case 28: JK_AW(28); break; default: JK_AW(32); break;
    }
#undef JK_AW
#undef JK_A
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[1], stream));
    hipLaunchKernelGGL(mz_split_count_kernel, dim3(G.nsubP, nc), dim3(256), 0, stream, bufX, cntA, G, cntP);
    hipLaunchKernelGGL(mz_scan1_kernel, dim3(nc), dim3(256), 0, stream, cntP, baseP, ctot, G);
    hipLaunchKernelGGL(mz_scan2_kernel, dim3(64), dim3(256), 0, stream, baseP, ctot, startF, G);
    hipLaunchKernelGGL(split16_kernel<0>, dim3(G.nsubP, nc), dim3(ST_TH), 0, stream, bufX, cntA, d, G, baseP, bufY, (unsigned int *)nullptr, defer_e, defer_n, deferred_cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[2], stream));
    if (k > 32) hipLaunchKernelGGL(mz_count_kernel<true>, dim3(G.nblkC), dim3(MC_TH), 0, stream, bufY, startF + NF, d, G, bufX, cntC, defer_e, defer_n, deferred_cap);
    else hipLaunchKernelGGL(mz_count_kernel<false>, dim3(G.nblkC), dim3(MC_TH), 0, stream, bufY, startF + NF, d, G, bufX, cntC, defer_e, defer_n, deferred_cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[3], stream));
    hipLaunchKernelGGL(split16_kernel<1>, dim3(G.nsubE, ne1), dim3(ST_TH), 0, stream, bufX, cntC, d, G, (const unsigned int *)nullptr, bufY, cntE, defer_e, defer_n, deferred_cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev_stage_t[4], stream));
    if (insert_entry_lists(bufY, cntE, G.capE, G.nsubE, G.pe1 + G.pe2, G.rbits, defer_e, defer_n, deferred_cap, &ev_stage_t[5], err)) return -1;   // records ev 5, 6
    unsigned long long *histo = histo_request ? d_histo : nullptr;
    hipLaunchKernelGGL(mz_expand_kernel, dim3(256), dim3(256), 0, stream, ovf, ovf_n, G.ovf_cap, d, histo ? histo + 10002 : nullptr);
    HIPCHK(hipGetLastError());
    if (finish_deferred(defer_e, defer_n, deferred_cap, err)) return -1;
    HIPCHK(hipEventRecord(ev_k1, stream));
    HIPCHK(hipEventRecord(ev_stage_t[7], stream));
    part_stage_pending = true;
    part_stage_n = 7;
    count_path = 2;
    if (getenv("JASPER_COUNT_DEBUG") && atoi(getenv("JASPER_COUNT_DEBUG")) >= 2) {
        HIPCHK(jk_stream_wait(stream));
        unsigned long long dn[2] = {0, 0};
        HIPCHK(hipMemcpy(dn, defer_n, 16, hipMemcpyDeviceToHost));
        if (G.exp == 9) {
            unsigned long long st[ST_WORDS];
            HIPCHK(hipMemcpy(st, d.stats, sizeof st, hipMemcpyDeviceToHost));
            fprintf(stderr, "[count] mz_count: %llu unique records flushed, %llu k-mers rolled out (cumulative)\n", st[6], st[5]);
        }
        std::vector<unsigned int> hc(n_cntA + 2 * n_cntP + NF + 1 + n_cntC + n_cntE);
        HIPCHK(hipMemcpy(hc.data(), cur, hc.size() * 4, hipMemcpyDeviceToHost));
        unsigned int mxA = 0, mxC = 0, mxE = 0, mxF = 0;
        unsigned long long totA = 0, totC = 0;
        for (size_t i = 0; i < n_cntA; ++i) { mxA = std::max(mxA, hc[i]); totA += hc[i]; }
        const unsigned int *sF = hc.data() + n_cntA + 2 * n_cntP;
        for (size_t i = 0; i < NF; ++i) mxF = std::max(mxF, sF[i + 1] - sF[i]);
        const unsigned int *cC = sF + NF + 1;
        for (size_t i = 0; i < n_cntC; ++i) { mxC = std::max(mxC, cC[i]); totC += cC[i]; }
        for (size_t i = 0; i < n_cntE; ++i) mxE = std::max(mxE, cC[n_cntC + i]);
        fprintf(stderr, "[count] minimizer piece %llu bases: m %d W %d nmax %d | pc %d pf %d pe1 %d pe2 %d rbits %d | records %llu (%.2f k-mers each) fullest sliceA %u / %u, "
                        "fine list %u | entries %llu fullest sliceC %u / %u, sliceE %u / %u | deferred %llu, overflow records %llu\n",
                (unsigned long long)len, G.m, G.W, G.nmax, G.pc, G.pf, G.pe1, G.pe2, G.rbits, totA, totA ? (double)len / (double)totA : 0.0, mxA, G.capA, mxF, totC, mxC,
                G.capC, mxE, G.capE, dn[0], dn[1]);
    }
    return 0;
}

}  // namespace jk
