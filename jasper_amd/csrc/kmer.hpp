// kmer.hpp -- 2-bit k-mer arithmetic shared by the HIP kernels and the host side of libjasper_hip.
//
// Semantics restated from Jellyfish 2.3.0 (cited as JF::path:line inside jellyfish-2.3.0.tar.gz):
//   * base codes A/a=0 C/c=1 G/g=2 T/t=3, anything else is "not DNA"   JF::include/jellyfish/mer_dna.hpp:38-55
//   * the first base of a k-mer is the MOST significant bit pair        JF::include/jellyfish/mer_dna.hpp:525-542
//   * canonical = numeric min(mer, reverse complement)                  JF::include/jellyfish/mer_dna.hpp:428-431
// Layout here is our own: a k-mer (k <= 64) is a 128-bit integer held in two 64-bit halves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JK_HD __host__ __device__ __forceinline__

// Host-side wait for a stream.  hipStreamSynchronize may put the thread to sleep and wake it milliseconds after the
// work finished (observed: 7-10 ms for an empty stream on some hosts); polling hipStreamQuery returns as soon as the GPU
// is done.  The polling loop is bounded by the work itself -- it issues nothing.
static inline hipError_t jk_stream_wait(hipStream_t st) {
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
}

namespace jk {

struct u128 {
    uint64_t lo, hi;
};

JK_HD u128 mk(uint64_t hi, uint64_t lo) { u128 r; r.lo = lo; r.hi = hi; return r; }
JK_HD bool lt(u128 a, u128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
JK_HD bool eq(u128 a, u128 b) { return a.hi == b.hi && a.lo == b.lo; }
JK_HD u128 shl(u128 a, unsigned s) {  // 0 <= s < 128
    if (s == 0) return a;
    if (s >= 64) return mk(a.lo << (s - 64), 0);
    return mk((a.hi << s) | (a.lo >> (64 - s)), a.lo << s);
}
JK_HD u128 shr(u128 a, unsigned s) {  // 0 <= s < 128
    if (s == 0) return a;
    if (s >= 64) return mk(0, a.hi >> (s - 64));
    return mk(a.hi >> s, (a.lo >> s) | (a.hi << (64 - s)));
}
JK_HD u128 band(u128 a, u128 b) { return mk(a.hi & b.hi, a.lo & b.lo); }
JK_HD u128 bor(u128 a, u128 b) { return mk(a.hi | b.hi, a.lo | b.lo); }
JK_HD u128 bxor(u128 a, u128 b) { return mk(a.hi ^ b.hi, a.lo ^ b.lo); }
JK_HD u128 maskbits(unsigned bits) {  // low `bits` ones, 0 <= bits <= 128
    if (bits >= 128) return mk(~0ull, ~0ull);
    if (bits >= 64) return mk(bits == 64 ? 0 : ((1ull << (bits - 64)) - 1), ~0ull);
    return mk(0, bits == 0 ? 0 : ((1ull << bits) - 1));
}

JK_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
// (a * b) mod 2^128
JK_HD u128 mul(u128 a, u128 b) {
    u128 r;
    r.lo = a.lo * b.lo;
    r.hi = mulhi64(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}

// base code or -1
JK_HD int code(unsigned char c) {
    // branch-free: ASCII upper/lower folded by clearing bit 5
    unsigned char u = c & 0xDF;
    int r = -1;
    r = (u == 'A') ? 0 : r;
    r = (u == 'C') ? 1 : r;
    r = (u == 'G') ? 2 : r;
    r = (u == 'T') ? 3 : r;
    return r;
}

JK_HD uint64_t brev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
// reverse the order of the 32 bit PAIRS of a word
JK_HD uint64_t revpairs64(uint64_t x) {
    uint64_t r = brev64(x);
    return ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
}
// reverse complement of a k-mer stored in the low 2k bits (word-level trick, our own formulation)
JK_HD u128 revcomp(u128 m, int k) {
    u128 c = mk(~m.hi, ~m.lo);                               // complement: code -> 3 - code
    u128 r = mk(revpairs64(c.lo), revpairs64(c.hi));         // reverse all 64 pairs
    return shr(r, 128 - 2 * k);                              // keep the k pairs that held data
}
JK_HD u128 canonical(u128 m, int k) {
    u128 r = revcomp(m, k);
    return lt(r, m) ? r : m;
}

// ---- bijective mixing of a B-bit key (B = 2k, even, <= 128) -------------------------------------
// The table stores only the part of the hash that the slot index does not imply, so the mix must be a bijection on
// B-bit values.  (Same idea as Jellyfish's invertible GF(2) matrix, JF::include/jellyfish/rectangular_binary_matrix.hpp,
// but a different function: integer multiply-shift suits the GPU's ALUs, a 2k x 2k bit matrix does not.)
// ONE 64-bit multiply per key: hashing is issue-bound in the counting kernels (a 64-bit multiply is three quarter-rate
// instructions), and what the table needs from the hash is only that the TOP bits (level-1 bucket, region, home slot) depend
// on every key bit -- which a fold of the high half into the low half followed by a multiply gives.
//   B <= 64: v ^= v >> B/2;  v = v * C mod 2^B;  v ^= v >> B/2           -- each step is invertible on B bits.
//   B >  64: one Feistel step over (hi: B-64 bits, lo: 64 bits): lo' = fold-multiply-xorshift of lo, hi' = hi ^ F(lo')
//            with F = bits 30.. of lo' (rotated).  lo -> lo' is a bijection and F depends on lo only, so (hi', lo') is a
//            bijection; keys that differ only in hi get different TOP hash bits (far-apart home slots).  F avoids the top bits
//            of lo': those are the next bits of the home slot, and taking F from them would tie the two together (measured on
//            low-complexity and sequential keys: region fills 15x more uneven).  Spread measured against the previous
//            two-multiply mix on random, error-variant, tandem-repeat and sequential keys: region fill sd/sqrt(mean) 0.98-1.06
//            for both, same probe lengths.
#define JK_C1 0x9E3779B97F4A7C15ull

JK_HD uint64_t rotr64(uint64_t x, int r) { return (x >> r) | (x << ((64 - r) & 63)); }
JK_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 32;
    x *= JK_C1;
    x ^= x >> 29;
    return x;
}

JK_HD u128 mix(u128 x, int B) {
    if (B <= 64) {
        const uint64_t m = B == 64 ? ~0ull : ((1ull << B) - 1);
        const int h = B / 2;
        uint64_t v = x.lo & m;
        v ^= v >> h;
        v = (v * JK_C1) & m;
        v ^= v >> h;
        return mk(0, v);
    }
    const int hb = B - 64;                                   // 2..64
    const uint64_t lo = mix64(x.lo);
    const uint64_t hm = hb == 64 ? ~0ull : ((1ull << hb) - 1);
    return mk((x.hi ^ rotr64(lo, 30)) & hm, lo);
}

// inverse of mix (needed only to write k-mers out again: the .jf writer)
JK_HD uint64_t inv_odd64(uint64_t a) {          // a * inv == 1 mod 2^64 (Newton: doubles the correct bits each round)
    uint64_t x = a;                             // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2 - a * x;
    return x;
}
JK_HD u128 unmix(u128 h, int B) {
    if (B <= 64) {
        const uint64_t m = B == 64 ? ~0ull : ((1ull << B) - 1);
        const int hh = B / 2;
        uint64_t v = h.lo & m;
        v ^= v >> hh;                            // xor-shift by half the width is its own inverse
        v = (v * inv_odd64(JK_C1)) & m;
        v ^= v >> hh;
        return mk(0, v);
    }
    const int hb = B - 64;
    const uint64_t hm = hb == 64 ? ~0ull : ((1ull << hb) - 1);
    uint64_t y = h.lo;
    y = y ^ (y >> 29) ^ (y >> 58);               // inverse of y ^= y >> 29
    y *= inv_odd64(JK_C1);
    y ^= y >> 32;
    return mk((h.hi ^ rotr64(h.lo, 30)) & hm, y);
}

// ---- slot word ------------------------------------------------------------------------------------
// A slot is 16 bytes: { uint64 tag, uint64 count }.  tag == 0  <=>  empty.
// tag = 1<<63 | remainder << OFFBITS | probe_offset, where the B-bit hash is split as
//   home = hash >> (B - s)      (s = log2(#slots))      remainder = hash & (2^(B-s) - 1)
// and the entry lives at slot (home + probe_offset) & (2^s - 1).  Needs B - s <= 63 - OFFBITS.
constexpr int OFFBITS = 10;
constexpr uint32_t MAXPROBE = (1u << OFFBITS);
constexpr uint64_t OCC = 1ull << 63;

JK_HD uint64_t home_of(u128 h, int B, int s) { return shr(h, B - s).lo; }
JK_HD uint64_t rem_of(u128 h, int B, int s) { return band(h, maskbits(B - s)).lo; }
JK_HD uint64_t tag_of(uint64_t rem, uint32_t off) { return OCC | (rem << OFFBITS) | off; }
JK_HD u128 hash_from(uint64_t home, uint64_t rem, int B, int s) { return bor(shl(mk(0, home), B - s), mk(0, rem)); }

// Wide remainders.  The tag word has room for 63 - OFFBITS = 53 remainder bits.  When the slot index leaves more than that
// (B - s > 53: k >= 38 unless the table is huge, every k >= 44) the LOW 64 remainder bits live in a second array, one word
// per slot ("ext"), and the tag keeps the bits above them (B - s - 64 <= 53 for every k <= 64 once s >= 11).  Same role as
// Jellyfish's variable-width key field, which may straddle words (JF::include/jellyfish/large_hash_array.hpp:509-597).
JK_HD bool wide_rem(int B, int s) { return B - s > 63 - OFFBITS; }
JK_HD uint64_t tag_rem_of(u128 h, int B, int s) { const u128 r = band(h, maskbits(B - s)); return wide_rem(B, s) ? r.hi : r.lo; }
JK_HD uint64_t ext_of(u128 h, int B, int s) { return band(h, maskbits(B - s)).lo; }
JK_HD u128 hash_from_wide(uint64_t home, uint64_t tag_rem, uint64_t ext, int B, int s) { return bor(shl(mk(0, home), B - s), mk(tag_rem, ext)); }

}  // namespace jk
