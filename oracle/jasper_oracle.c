/*
 * jasper_oracle.c -- TEST INFRASTRUCTURE ONLY (see jasper_oracle.h).
 *
 * CPU restatement of the reference hot path. Every function cites the reference lines it follows:
 *   "JF::x:n"  = file x inside /root/reference/jellyfish-2.3.0.tar.gz (jellyfish-2.3.0/x), line n
 *   "src/x:n"  = /root/reference/src/x, line n
 * Not a copy: the reference is C++ templates + Python; this is a from-scratch C restatement of the
 * observable semantics (which windows are counted, what a lookup returns, which edits the walk makes).
 *
 * Parity: pinned by tests/golden/ (outputs of the real reference), except the Bio.pairwise2-dependent
 * CSV rows of the ">k" branch (unpinned, see header).
 */
#include "jasper_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------
 * a1: 2-bit codes. JF::include/jellyfish/mer_dna.hpp:38-55 -- A/a=0 C/c=1 G/g=2 T/t=3, everything else
 * is "not DNA" (negative). First base of the string lands in the MOST significant bit pair (:525-542).
 * ---------------------------------------------------------------------------------------------- */
static inline int base_code(unsigned char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static inline u128 kmask(int k) { return (k >= 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1); }

/* reverse complement of a k-mer held in the low 2k bits (JF::include/jellyfish/mer_dna.hpp:401-414):
 * complement of base code c is 3-c, order reversed. Done the slow obvious way on purpose. */
static u128 revcomp(int k, u128 m) {
    u128 r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3 - (m & 3));
        m >>= 2;
    }
    return r;
}

/* JF::include/jellyfish/mer_dna.hpp:428-431 + operator< :227-250 : numeric min of mer and its revcomp */
static inline u128 canonical(int k, u128 m) {
    u128 r = revcomp(k, m);
    return r < m ? r : m;
}

/* Appendix A.3 / JF::include/jellyfish/mer_dna.hpp:525-542 (from_chars returns at the first non-DNA code,
 * the mer having been zero-filled first) + JF::swig/mer_dna.i:15 (MerDNA(const char*) -> no length check:
 * a NUL ends a short string). Returns the padded mer; *taken = number of bases actually read. */
static u128 encode_padded(int k, const char *s, long n, int *taken) {
    u128 m = 0;
    int t = 0;
    while (t < k && t < n) {
        int c = base_code((unsigned char)s[t]);
        if (c < 0) break;
        m = (m << 2) | (u128)c;
        t++;
    }
    if (t < k) m <<= 2 * (k - t); /* the rest stays 0 = 'A' */
    if (taken) *taken = t;
    return m;
}

int jo_encode(int k, const char *s, long n, uint64_t out[2]) {
    int t;
    u128 m = encode_padded(k, s, n, &t);
    out[0] = (uint64_t)m;
    out[1] = (uint64_t)(m >> 64);
    return t;
}
void jo_revcomp(int k, const uint64_t in[2], uint64_t out[2]) {
    u128 m = ((u128)in[1] << 64) | in[0];
    m = revcomp(k, m & kmask(k));
    out[0] = (uint64_t)m;
    out[1] = (uint64_t)(m >> 64);
}
void jo_canonical(int k, const uint64_t in[2], uint64_t out[2]) {
    u128 m = ((u128)in[1] << 64) | in[0];
    m = canonical(k, m & kmask(k));
    out[0] = (uint64_t)m;
    out[1] = (uint64_t)(m >> 64);
}

/* ------------------------------------------------------------------------------------------------
 * a4: the count table. The reference is a lock-free bit-packed hash with exact 64-bit counts
 * (JF::include/jellyfish/large_hash_array.hpp:291,509-597,674-752); the result is a pure function of the
 * read multiset, so the restatement is a plain single-threaded open-addressing map that doubles when
 * 2/3 full (the reference doubles too, JF::include/jellyfish/hash_counter.hpp:200-238).
 * ---------------------------------------------------------------------------------------------- */
struct jo_db {
    int k;
    uint64_t cap, used; /* cap is a power of two */
    u128 *keys;         /* key+1 stored, 0 = empty (key+1 never overflows: 2k <= 126 enforced) */
    uint64_t *vals;
    /* multi-threaded baseline only (jo_mt_*, end of file): the map split into `nshard` maps by key owner, so that the
     * counting threads never share a map (what `jellyfish count -t N` does with CAS on one array). 0 = plain map. */
    int nshard;
    struct jo_db **shard;
};

static inline uint64_t mix128(u128 x) {
    uint64_t a = (uint64_t)x, b = (uint64_t)(x >> 64);
    a ^= b * 0x9E3779B97F4A7C15ull;
    a ^= a >> 32; a *= 0xD6E8FEB86659FD93ull;
    a ^= a >> 32; a *= 0xD6E8FEB86659FD93ull;
    a ^= a >> 32;
    return a;
}

jo_db *jo_db_new(int k) {
    if (k < 1 || k > 63) return NULL;
    jo_db *db = (jo_db *)calloc(1, sizeof *db);
    db->k = k;
    db->cap = 1u << 16;
    db->keys = (u128 *)calloc(db->cap, sizeof(u128));
    db->vals = (uint64_t *)calloc(db->cap, sizeof(uint64_t));
    return db;
}
void jo_db_free(jo_db *db) {
    if (!db) return;
    for (int i = 0; i < db->nshard; i++) jo_db_free(db->shard[i]);
    free(db->shard);
    free(db->keys); free(db->vals); free(db);
}
int jo_db_k(const jo_db *db) { return db->k; }
uint64_t jo_db_distinct(const jo_db *db) {
    uint64_t n = db->used;
    for (int i = 0; i < db->nshard; i++) n += db->shard[i]->used;
    return n;
}
/* owner of a key among n maps: hash bits the in-map position does not use */
static inline int db_owner(u128 key, int n) { return (int)(((mix128(key) >> 40) * (uint64_t)n) >> 24); }

static void db_grow(jo_db *db);
static inline void db_add(jo_db *db, u128 key, uint64_t by) {
    if ((db->used + 1) * 3 > db->cap * 2) db_grow(db);
    uint64_t mask = db->cap - 1, p = mix128(key) & mask;
    u128 kk = key + 1;
    for (;;) {
        if (db->keys[p] == kk) { db->vals[p] += by; return; }
        if (db->keys[p] == 0) { db->keys[p] = kk; db->vals[p] = by; db->used++; return; }
        p = (p + 1) & mask;
    }
}
static void db_grow(jo_db *db) {
    uint64_t ocap = db->cap;
    u128 *ok = db->keys;
    uint64_t *ov = db->vals;
    db->cap = ocap * 2;
    db->keys = (u128 *)calloc(db->cap, sizeof(u128));
    db->vals = (uint64_t *)calloc(db->cap, sizeof(uint64_t));
    db->used = 0;
    for (uint64_t i = 0; i < ocap; i++)
        if (ok[i]) db_add(db, ok[i] - 1, ov[i]);
    free(ok); free(ov);
}
static inline uint64_t db_get(const jo_db *db, u128 key) {
    if (db->nshard) db = db->shard[db_owner(key, db->nshard)];
    uint64_t mask = db->cap - 1, p = mix128(key) & mask;
    u128 kk = key + 1;
    for (;;) {
        if (db->keys[p] == kk) return db->vals[p];
        if (db->keys[p] == 0) return 0;
        p = (p + 1) & mask;
    }
}

/* a3: JF::include/jellyfish/mer_iterator.hpp:53-81 -- rolling forward mer and reverse-complement mer;
 * a valid code shifts both, anything else resets `filled`; emit min(m, rc) once filled >= k (:51). */
uint64_t jo_db_count_bases(jo_db *db, const char *bases, size_t n) {
    const int k = db->k;
    const u128 mask = kmask(k);
    u128 fwd = 0, rc = 0;
    int filled = 0;
    uint64_t added = 0;
    for (size_t i = 0; i < n; i++) {
        int c = base_code((unsigned char)bases[i]);
        if (c < 0) { filled = 0; continue; }
        fwd = ((fwd << 2) | (u128)c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
        if (filled < k) filled++;
        if (filled >= k) { db_add(db, fwd < rc ? fwd : rc, 1); added++; }
    }
    return added;
}

/* a2: JF::include/jellyfish/mer_overlap_sequence_parser.hpp.
 *  - format from the first byte of the (concatenated) stream: '>' FASTA, '@' FASTQ, else error (:134-148)
 *  - FASTA: header lines are skipped, sequence lines are concatenated ('\n' and trailing '\r' dropped,
 *    :260-275), records are separated by an 'N' (:175)
 *  - FASTQ: sequence lines up to the line starting with '+', then as many quality characters as there
 *    were sequence characters are skipped, possibly over several lines (:290-307); next byte must be '@'
 *    or EOF, else "Invalid fastq sequence"; records separated by 'N' (:205)
 * The 4096-byte buffers with (k-1)-base seams (:182-184) only re-join what they split, so they are not
 * modelled. Returns 0 ok, -1 unsupported format, -2 invalid fastq. */
typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_reserve(sbuf *b, size_t extra) {
    if (b->n + extra + 1 > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        while (nc < b->n + extra + 1) nc *= 2;
        b->p = (char *)realloc(b->p, nc);
        b->cap = nc;
    }
}
static void sb_put(sbuf *b, const char *s, size_t n) {
    sb_reserve(b, n);
    memcpy(b->p + b->n, s, n);
    b->n += n;
    b->p[b->n] = 0;
}
static void sb_putc(sbuf *b, char c) { sb_put(b, &c, 1); }

static size_t skip_nl(const char *t, size_t n, size_t i) {
    while (i < n && (t[i] == '\n' || t[i] == '\r')) i++;
    return i;
}
static size_t skip_line(const char *t, size_t n, size_t i) {
    while (i < n && t[i] != '\n') i++;
    return i < n ? i + 1 : n;
}
/* one logical "read_sequence": lines until a line starts with `stop`; appends bases to out */
static size_t read_seq_lines(const char *t, size_t n, size_t i, char stop, sbuf *out, size_t *nbases) {
    i = skip_nl(t, n, i);
    while (i < n && t[i] != stop) {
        size_t e = i;
        while (e < n && t[e] != '\n') e++;
        size_t le = e;
        while (le > i && t[le - 1] == '\r') le--;
        sb_put(out, t + i, le - i);
        *nbases += le - i;
        i = skip_nl(t, n, e);
    }
    return i;
}

int jo_db_count_text(jo_db *db, const char *t, size_t n, uint64_t *n_kmers) {
    sbuf bases = {0};
    size_t i = 0;
    int rc = 0;
    if (n_kmers) *n_kmers = 0;
    if (n == 0) return 0;
    if (t[0] == '>') {
        i = skip_line(t, n, 0);
        while (i < n) {
            size_t nb = 0;
            i = read_seq_lines(t, n, i, '>', &bases, &nb);
            if (i < n && t[i] == '>') {
                if (bases.n > 0) sb_putc(&bases, 'N');
                i = skip_line(t, n, i);
            }
        }
    } else if (t[0] == '@') {
        i = skip_line(t, n, 0);
        while (i < n) {
            size_t nb = 0;
            i = read_seq_lines(t, n, i, '+', &bases, &nb);
            if (i < n && t[i] == '+') {
                i = skip_line(t, n, i); /* '+' line */
                size_t q = 0;
                i = skip_nl(t, n, i);
                while (i < n && q < nb) { /* quality characters, by count, over any number of lines */
                    size_t e = i;
                    while (e < n && t[e] != '\n' && (e - i) < (nb - q)) e++;
                    size_t le = e;
                    while (le > i && t[le - 1] == '\r') le--;
                    q += le - i;
                    i = skip_nl(t, n, e);
                }
                i = skip_nl(t, n, i);
                if (!(q == nb && (i >= n || t[i] == '@'))) { rc = -2; break; }
                if (i < n) {
                    sb_putc(&bases, 'N');
                    i = skip_line(t, n, i);
                }
            }
        }
    } else {
        rc = -1;
    }
    if (rc == 0 && bases.n) {
        uint64_t a = jo_db_count_bases(db, bases.p, bases.n);
        if (n_kmers) *n_kmers = a;
    }
    free(bases.p);
    return rc;
}

static inline uint32_t clamp32(uint64_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

uint32_t jo_db_query(const jo_db *db, const char *s, long n) {
    u128 m = encode_padded(db->k, s, n < 0 ? 0 : n, NULL);
    return clamp32(db_get(db, canonical(db->k, m)));
}

int jo_db_add_kmer(jo_db *db, const char *kmer, uint64_t count) {
    int t;
    u128 m = encode_padded(db->k, kmer, db->k, &t);
    if (t != db->k) return -1;
    db_add(db, canonical(db->k, m), count);
    return 0;
}

static void db_histo_add(const jo_db *db, uint64_t *out);
void jo_db_histo(const jo_db *db, uint64_t *out) {
    memset(out, 0, 10002 * sizeof(uint64_t));
    db_histo_add(db, out);
    for (int i = 0; i < db->nshard; i++) db_histo_add(db->shard[i], out);
}
static void db_histo_add(const jo_db *db, uint64_t *out) {
    for (uint64_t i = 0; i < db->cap; i++) {
        if (!db->keys[i]) continue;
        uint64_t c = clamp32(db->vals[i]);
        if (c == 0) continue;
        out[c > 10001 ? 10001 : c]++; /* JF::sub_commands/histo_main.cc:38-42: low=1, high=10000, ceil bucket */
    }
}

int jo_db_next(const jo_db *db, uint64_t *cursor, char *kmer_out, uint64_t *count_out) {
    while (*cursor < db->cap) {
        uint64_t i = (*cursor)++;
        if (!db->keys[i]) continue;
        u128 m = db->keys[i] - 1;
        for (int b = 0; b < db->k; b++) kmer_out[b] = "ACGT"[(int)((m >> (2 * (db->k - 1 - b))) & 3)];
        kmer_out[db->k] = 0;
        *count_out = db->vals[i];
        return 1;
    }
    return 0;
}

/* a7: src/jellyfish.py:8-22 */
int jo_threshold(const uint64_t *mult, const uint64_t *nd, size_t nrows) {
    long long count = -1;
    long long threshold = 0;
    int first = 1;
    for (size_t r = 0; r < nrows; r++) {
        if (first) { count = (long long)nd[r]; first = 0; continue; } /* :12-13 (count == -1) */
        if (count >= (long long)nd[r]) {                                /* :15-17 */
            count = (long long)nd[r];
            threshold = (long long)(mult[r] / 2);
        } else {                                                        /* :18-22 */
            if (threshold < 2) return -1;
            return (int)threshold;
        }
    }
    return 0; /* loop fell through: nothing printed, exit status 0 */
}

/* ================================================================================================
 * a10-a13: src/jasper.py
 * ============================================================================================== */
typedef struct {
    const jo_db *db;
    int k, step, solid;
    char *s;      /* current chunk */
    long len, cap;
    const char *name;
    sbuf *csv;    /* rows of the current pass */
    uint64_t nlook;
    int fatal;
} ctx_t;

/* Python slice bounds for seq[a:b] (CPython PySlice_AdjustIndices, step 1) */
static void pyslice(long len, long a, long b, long *lo, long *hi) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

/* qf[jf.MerDNA(str).get_canonical()] for an explicit buffer */
static inline uint32_t cnt_buf(ctx_t *c, const char *p, long n) {
    c->nlook++;
    return jo_db_query(c->db, p, n);
}
/* qf[jf.MerDNA(seq[a:b]).get_canonical()] on the chunk, Python slice semantics */
static inline uint32_t cnt_seq(ctx_t *c, long a, long b) {
    long lo, hi;
    pyslice(c->len, a, b, &lo, &hi);
    return cnt_buf(c, c->s + lo, hi - lo);
}

/* seq = seq[:a] + patch + seq[b:]   (a,b already valid, a<=b) */
static void splice(ctx_t *c, long a, long b, const char *patch, long plen) {
    long nl = c->len - (b - a) + plen;
    if (nl + 1 > c->cap) {
        c->cap = nl + 1 + 1024;
        c->s = (char *)realloc(c->s, c->cap);
    }
    memmove(c->s + a + plen, c->s + b, c->len - b);
    memcpy(c->s + a, patch, plen);
    c->len = nl;
    c->s[nl] = 0;
}


/* tiny owned string used for trial sequences */
typedef struct { char *p; long n; } ostr;
static ostr os_new(long cap) { ostr s; s.p = (char *)malloc((size_t)cap + 2); s.n = 0; s.p[0] = 0; return s; }
static void os_free(ostr *s) { free(s->p); s->p = NULL; s->n = 0; }
static void os_cat(ostr *s, const char *p, long n) { memcpy(s->p + s->n, p, (size_t)n); s->n += n; s->p[s->n] = 0; }
static void os_catc(ostr *s, char c) { s->p[s->n++] = c; s->p[s->n] = 0; }

/* src/jasper.py:585-599 check_sequence(trial, qf, k, threshold) */
static int check_sequence(ctx_t *c, const char *t, long n, uint32_t thr) {
    const int k = c->k;
    long lo, hi;
    pyslice(n, 0, k, &lo, &hi);                       /* trial[:k]  :589 */
    if (cnt_buf(c, t + lo, hi - lo) < thr) return 0;
    pyslice(n, -(long)k, n, &lo, &hi);                /* trial[-k:] :592 */
    if (cnt_buf(c, t + lo, hi - lo) < thr) return 0;
    for (long i = c->step; i < n - k; i += c->step) { /* range(step, len(trial)-k, step) :595 */
        pyslice(n, i, i + k, &lo, &hi);
        if (cnt_buf(c, t + lo, hi - lo) < thr) return 0;
    }
    return 1;
}

/* result of fixing_sid (src/jasper.py:226-332): fixed_base / original / fixed_ind */
typedef struct {
    int changed;     /* fixed_base != "nN" */
    int is_list;     /* the ">k" branch returns python lists */
    int n;           /* len(fixed_ind) */
    long ind[2];
    char *newb[2];   /* malloc'ed strings */
    char *orig[2];
} fixres;

static char *dupn(const char *p, long n) { char *r = (char *)malloc((size_t)n + 1); memcpy(r, p, (size_t)n); r[n] = 0; return r; }
static char *tag_str(char tag, const char *p, long n) { char *r = (char *)malloc((size_t)n + 2); r[0] = tag; memcpy(r + 1, p, (size_t)n); r[n + 1] = 0; return r; }
static char *rep_str(char ch, long n) { char *r = (char *)malloc((size_t)n + 1); memset(r, ch, (size_t)n); r[n] = 0; return r; }

/* src/jasper.py:392-406 fix_k_case_sub: returns base or 0; *out = trial */
static char fix_k_case_sub(ctx_t *c, const char *tbf, long L, uint32_t thr, ostr *out) {
    const int k = c->k;
    char bad = tbf[k - 1];
    static const char order[] = "ACTG"; /* :396 */
    for (int b = 0; b < 4; b++) {
        if (order[b] == bad) continue;
        ostr t = os_new(L);
        os_cat(&t, tbf, k - 1); os_catc(&t, order[b]); os_cat(&t, tbf + k, L - k); /* :401 */
        if (check_sequence(c, t.p, t.n, thr)) { *out = t; return order[b]; }
        os_free(&t);
    }
    return 0;
}
/* src/jasper.py:409-419 fix_insert */
static char fix_insert(ctx_t *c, const char *tbf, long L, uint32_t thr, ostr *out) {
    const int k = c->k;
    ostr t = os_new(L);
    os_cat(&t, tbf, k - 1); os_cat(&t, tbf + k, L - k); /* :414 */
    if (check_sequence(c, t.p, t.n, thr)) { *out = t; return tbf[k - 1]; }
    os_free(&t);
    return 0;
}
/* src/jasper.py:422-431 fix_del */
static char fix_del(ctx_t *c, const char *tbf, long L, uint32_t thr, ostr *out) {
    const int k = c->k;
    static const char order[] = "ATCG"; /* :425 */
    for (int a = 0; a < 4; a++) {
        ostr t = os_new(L + 1);
        os_cat(&t, tbf, k - 1); os_catc(&t, order[a]); os_cat(&t, tbf + k - 1, L - (k - 1)); /* :426 */
        if (check_sequence(c, t.p, t.n, thr)) { *out = t; return order[a]; }
        os_free(&t);
    }
    return 0;
}

/* src/jasper.py:340-382 fixdiploid. Returns 0 (None), 's' or 'e'; *left,*right, *out = trial */
static char fixdiploid(ctx_t *c, const char *tbf, long L, uint32_t thr, long gb, long ga, char *left, char *right, ostr *out) {
    const int k = c->k;
    const char *full = c->s;
    const long flen = c->len;
    char left_bad = tbf[L - k], right_bad = tbf[k - 1];                /* :348-349 */
    long gbsi = gb - k + 1; if (gbsi < 0) gbsi = 0;                    /* :352 */
    long h = (long)((double)(k - 1 - L + k) / 2.0);                    /* int((k-1-len+k)/2) :354 */
    long lo, hi, alo, ahi;
    if (ga + k - 1 + h < flen) pyslice(flen, ga + k - 1, ga + k - 1 + h, &alo, &ahi);          /* :355 */
    else { long st = ga + k - 1; if (st > flen - 1) st = flen - 1; pyslice(flen, st, flen, &alo, &ahi); } /* :357 */
    long before_len = ahi - alo;
    long bs = gbsi - before_len + 1; if (bs < 0) bs = 0;
    pyslice(flen, bs, gbsi + 1, &lo, &hi);                             /* :359 */
    static const char order[] = "ACTG";
    for (int xi = 0; xi < 4; xi++) for (int yi = 0; yi < 4; yi++) {
        char x = order[xi], y = order[yi];
        if (x == left_bad && y == right_bad) continue;                 /* :362 */
        if (x != left_bad && y != right_bad) continue;                 /* :364 */
        ostr t = os_new(L + 2);
        long a0, a1;
        pyslice(L, 0, L - k, &a0, &a1); os_cat(&t, tbf + a0, a1 - a0); /* :366 */
        os_catc(&t, x);
        pyslice(L, L - k + 1, k - 1, &a0, &a1); os_cat(&t, tbf + a0, a1 - a0);
        os_catc(&t, y);
        pyslice(L, k, L, &a0, &a1); os_cat(&t, tbf + a0, a1 - a0);
        ostr chk = os_new((hi - lo) + t.n + (ahi - alo));
        os_cat(&chk, full + lo, hi - lo); os_cat(&chk, t.p, t.n); os_cat(&chk, full + alo, ahi - alo); /* :367 */
        int ok = check_sequence(c, chk.p, chk.n, thr);
        os_free(&chk);
        if (ok) {
            *left = x; *right = y;
            *out = t;
            return (x == left_bad) ? 'e' : 's';                         /* :371-375 (third case unreachable) */
        }
        os_free(&t);
    }
    return 0;
}

/* src/jasper.py:434-477 fix_same_base_del. Returns 1 on success: *ridx, *rbase (malloc'ed), *out */
static int fix_same_base_del(ctx_t *c, const char *tbf, long L, uint32_t thr, long *ridx, char **rbase, ostr *out) {
    const int k = c->k;
    if (thr > (uint32_t)c->solid) return 0;           /* :437 */
    char sb = tbf[k - 2];                             /* :439 */
    long inserted = 0, original_bad = L - k + 1, current_bad = original_bad, max_ins = original_bad;
    ostr trial = os_new(L + max_ins + 1);
    os_cat(&trial, tbf, L);
    while (inserted < max_ins) {                      /* :449 */
        long new_bad = 0;
        memmove(trial.p + k, trial.p + k - 1, (size_t)(trial.n - (k - 1)) + 1); /* trial[:k-1]+sb+trial[k-1:] :451 */
        trial.p[k - 1] = sb; trial.n++;
        int fixed = 1;
        inserted++;
        for (long i = 0; i < trial.n - k + 1; i++)    /* :454 */
            if (cnt_buf(c, trial.p + i, k) < thr) { fixed = 0; new_bad++; }
        if (fixed) { *ridx = k - 1; *rbase = rep_str(sb, inserted); *out = trial; return 1; } /* :461 */
        if (new_bad >= current_bad) { inserted = max_ins; break; }
        current_bad = new_bad;
    }
    os_free(&trial);
    static const char order[] = "ATCG";               /* :471 */
    for (int a = 0; a < 4; a++) {
        ostr t = os_new(L + 1);
        os_cat(&t, tbf, k - 2); os_catc(&t, order[a]); os_cat(&t, tbf + k - 2, L - (k - 2)); /* :472 */
        if (check_sequence(c, t.p, t.n, thr)) { *ridx = k - 2; *rbase = rep_str(order[a], 1); *out = t; return 1; }
        os_free(&t);
    }
    return 0;
}

/* src/jasper.py:479-524 fix_same_base_insertion */
static int fix_same_base_insertion(ctx_t *c, const char *tbf, long L, uint32_t thr, long *ridx, char **rbase, ostr *out) {
    const int k = c->k;
    if (thr > (uint32_t)c->solid) return 0;           /* :482 */
    char sb = tbf[k - 1];                             /* :485 */
    long deleted = 0, original_bad = L - k + 1, current_bad = original_bad, max_del = original_bad;
    ostr loc = os_new(L);
    os_cat(&loc, tbf, L);
    while (tbf[k - 1] == sb && deleted < max_del) {   /* :494 (first test is always true) */
        current_bad -= 1;
        deleted += 1;
        memmove(loc.p + k - 1, loc.p + k, (size_t)(loc.n - k) + 1); /* :497 */
        loc.n--;
        if (loc.n == k) break;                        /* :498 */
        int fixed = 1;
        long new_bad = 0;
        for (long i = 0; i < loc.n - k + 1; i++)      /* :502 */
            if (cnt_buf(c, loc.p + i, k) < thr) { fixed = 0; new_bad++; }
        if (fixed) { *ridx = k - 1; *rbase = rep_str(sb, deleted); *out = loc; return 1; } /* :509 */
        if (new_bad >= current_bad) break;
        current_bad = new_bad;
    }
    os_free(&loc);
    for (long i = L - k; i < L - 1; i++) {            /* :517 */
        if (i < 0) continue;                          /* (L >= k on every call path) */
        ostr t = os_new(L);
        os_cat(&t, tbf, i); os_cat(&t, tbf + i + 1, L - i - 1);
        if (check_sequence(c, t.p, t.n, thr)) { *ridx = i; *rbase = rep_str(tbf[i], 1); *out = t; return 1; }
        os_free(&t);
    }
    return 0;
}

/* Python round(): half-to-even on the double */
static inline long pyround(double x) { return (long)nearbyint(x); }

/* src/jasper.py:527-583 base_extension. Returns malloc'ed patch (may be empty string) or NULL (None). */
static char *base_extension(ctx_t *c, long Ltbf, const char *gkb, long gkb_n, const char *gka, long gka_n, uint32_t thr) {
    const int k = c->k;
    if (gkb_n < k || gka_n < k || thr > (uint32_t)c->solid) return NULL;   /* :528 */
    static const char bases[] = "ACGT";                                     /* :530 */
    const long min_overlap = 5, slack = 10;
    long max_ext = pyround((double)(Ltbf - 2 * k) * 1.2) + min_overlap + slack; /* :535 */
    long min_patch_len = pyround((double)(Ltbf - 2 * k) / 1.2) - slack;     /* :536 */
    long np = 1, capp = 64;
    char **paths = (char **)malloc(sizeof(char *) * (size_t)capp);
    long *plen = (long *)malloc(sizeof(long) * (size_t)capp);
    paths[0] = dupn(gkb + k - 1, 1); plen[0] = 1;                           /* :537 */
    char *result = NULL;
    int done = 0;
    char *tmp = (char *)malloc((size_t)(k + max_ext + 3 * k + 16));
    for (long i = 1; i < max_ext && !done; i++) {                           /* :541 */
        long w = 0;
        for (long p = 0; p < np; p++) {                                     /* :542 drop empty paths */
            if (plen[p] > 0) { paths[w] = paths[p]; plen[w] = plen[p]; w++; } else free(paths[p]);
        }
        np = w;
        if (np > 5000) { done = 1; break; }                                 /* :543-546 */
        long last_path = np;
        for (long p = 0; p < last_path && !done; p++) {
            if (plen[p] == 0) continue;
            /* km1 = (start_km1 + paths[p])[-k+1:] :551 */
            char km1[64];
            long tot = (k - 1) + plen[p];
            for (long q = 0; q < k - 1; q++) {
                long idx = tot - (k - 1) + q;
                km1[q] = idx < k - 1 ? gkb[idx] : paths[p][idx - (k - 1)];
            }
            int ext = 0;
            for (int j = 0; j < 4 && !done; j++) {
                km1[k - 1] = bases[j];
                uint32_t score = cnt_buf(c, km1, k);                        /* :554 */
                if (score < thr) continue;
                if (i >= min_overlap && i >= min_patch_len) {               /* :557 */
                    /* last_bases[-5:] == good_k_mer_after[0:5] :558 (k >= 5 assumed as in the reference) */
                    if (k >= min_overlap && memcmp(km1 + k - min_overlap, gka, (size_t)min_overlap) == 0) {
                        /* body = paths[p] (+ext: without its last char) + bases[j] */
                        long bl = ext ? plen[p] - 1 : plen[p];
                        long n = 0;
                        memcpy(tmp + n, gkb, (size_t)(k - 1)); n += k - 1;
                        memcpy(tmp + n, paths[p], (size_t)bl); n += bl;
                        tmp[n++] = bases[j];
                        long lo, hi;
                        pyslice(gka_n, -(long)(k - min_overlap), gka_n, &lo, &hi); /* good_k_mer_after[-(k-5):] */
                        memcpy(tmp + n, gka + lo, (size_t)(hi - lo)); n += hi - lo;
                        pyslice(n, -(long)(2 * k - 1), n, &lo, &hi);        /* [-(2k-1):] :560/:563 */
                        if (check_sequence(c, tmp + lo, hi - lo, thr)) {   /* :567 */
                            if (i == min_overlap) { result = NULL; done = 1; break; } /* :568-571 */
                            /* return_path = (body)[1:-5] :561/:564 */
                            long blen = bl + 1;
                            long rlo, rhi;
                            pyslice(blen, 1, -min_overlap, &rlo, &rhi);
                            char *body = (char *)malloc((size_t)blen + 1);
                            memcpy(body, paths[p], (size_t)bl); body[bl] = bases[j];
                            result = dupn(body + rlo, rhi - rlo);
                            free(body);
                            done = 1; break;
                        }
                    }
                }
                if (!ext) {                                                 /* :576-578 */
                    paths[p] = (char *)realloc(paths[p], (size_t)plen[p] + 2);
                    paths[p][plen[p]++] = bases[j]; paths[p][plen[p]] = 0;
                    ext = 1;
                } else {                                                    /* :579-580 */
                    if (np == capp) {
                        capp *= 2;
                        paths = (char **)realloc(paths, sizeof(char *) * (size_t)capp);
                        plen = (long *)realloc(plen, sizeof(long) * (size_t)capp);
                    }
                    paths[np] = dupn(paths[p], plen[p]);
                    paths[np][plen[p] - 1] = bases[j];
                    plen[np] = plen[p];
                    np++;
                }
            }
            if (!ext && !done) plen[p] = 0;                                 /* :581-582 */
        }
    }
    for (long p = 0; p < np; p++) free(paths[p]);
    free(paths); free(plen); free(tmp);
    return result;
}

/* Stand-in for Bio.pairwise2.align.globalms(a, b, 0, -1, -1, -1)[0] (src/jasper.py:309).
 * PARITY UNPINNED: Biopython is not part of the reference tree. Needleman-Wunsch with match 0 and every
 * mismatch/gap column -1; traceback from the end preferring (1) gap in `a`, (2) diagonal, (3) gap in `b`,
 * never following a gap in `b` by a gap in `a` (pairwise2's documented redundancy rule), first complete
 * path wins. ra/rb are malloc'ed aligned strings of equal length. */
static void align_globalms(const char *a, long n, const char *b, long m, char **ra, char **rb) {
    long W = m + 1;
    int *S = (int *)malloc(sizeof(int) * (size_t)((n + 1) * W));
    for (long i = 0; i <= n; i++) S[i * W] = -(int)i;
    for (long j = 0; j <= m; j++) S[j] = -(int)j;
    for (long i = 1; i <= n; i++)
        for (long j = 1; j <= m; j++) {
            int d = S[(i - 1) * W + j - 1] + (a[i - 1] == b[j - 1] ? 0 : -1);
            int u = S[(i - 1) * W + j] - 1, l = S[i * W + j - 1] - 1;
            int best = d; if (u > best) best = u; if (l > best) best = l;
            S[i * W + j] = best;
        }
    /* depth-first traceback with an explicit stack of (i, j, next option, col_gap, out length) */
    typedef struct { long i, j; int opt; int col_gap; long olen; } fr;
    long cap = n + m + 4;
    fr *st = (fr *)malloc(sizeof(fr) * (size_t)cap);
    char *oa = (char *)malloc((size_t)cap + 1), *ob = (char *)malloc((size_t)cap + 1);
    long sp = 0;
    st[sp++] = (fr){n, m, 0, 0, 0};
    long outlen = 0;
    while (sp > 0) {
        fr *f = &st[sp - 1];
        if (f->i == 0 && f->j == 0) { outlen = f->olen; break; }
        int advanced = 0;
        while (f->opt < 3 && !advanced) {
            int o = f->opt++;
            long i = f->i, j = f->j;
            int cur = S[i * W + j];
            if (o == 0) { /* gap in a: consume b[j-1] */
                if (j > 0 && cur == S[i * W + j - 1] - 1 && !f->col_gap) {
                    oa[f->olen] = '-'; ob[f->olen] = b[j - 1];
                    st[sp++] = (fr){i, j - 1, 0, 0, f->olen + 1}; advanced = 1;
                }
            } else if (o == 1) {
                if (i > 0 && j > 0 && cur == S[(i - 1) * W + j - 1] + (a[i - 1] == b[j - 1] ? 0 : -1)) {
                    oa[f->olen] = a[i - 1]; ob[f->olen] = b[j - 1];
                    st[sp++] = (fr){i - 1, j - 1, 0, 0, f->olen + 1}; advanced = 1;
                }
            } else { /* gap in b: consume a[i-1] */
                if (i > 0 && cur == S[(i - 1) * W + j] - 1) {
                    oa[f->olen] = a[i - 1]; ob[f->olen] = '-';
                    st[sp++] = (fr){i - 1, j, 0, 1, f->olen + 1}; advanced = 1;
                }
            }
        }
        if (!advanced) sp--; /* dead end: backtrack */
    }
    *ra = (char *)malloc((size_t)outlen + 1); *rb = (char *)malloc((size_t)outlen + 1);
    for (long q = 0; q < outlen; q++) { (*ra)[q] = oa[outlen - 1 - q]; (*rb)[q] = ob[outlen - 1 - q]; }
    (*ra)[outlen] = 0; (*rb)[outlen] = 0;
    free(S); free(st); free(oa); free(ob);
}

/* src/jasper.py:226-332 fixing_sid */
static void fixing_sid(ctx_t *c, const char *tbf, long L, uint32_t thr, long n, long gb, long ga, fixres *r) {
    const int k = c->k;
    long s0 = gb - k + 2; if (s0 < 0) s0 = 0;
    memset(r, 0, sizeof *r);
    ostr out = {0};
    if (n == k) {                                                            /* :232 */
        char b = fix_k_case_sub(c, tbf, L, thr, &out);
        if (b) {
            r->changed = 1; r->n = 1; r->ind[0] = ga - 1;
            r->orig[0] = tag_str('s', c->s + ga - 1, 1); r->newb[0] = rep_str(b, 1);   /* :235-237 */
            splice(c, s0, ga + k - 1, out.p, out.n);                                   /* :238 */
        } else {
            b = fix_insert(c, tbf, L, thr, &out);
            if (b) {
                r->changed = 1; r->n = 1; r->ind[0] = ga - 1;
                r->orig[0] = tag_str('i', c->s + ga - 1, 1); r->newb[0] = rep_str('-', 1); /* :242-244 */
                splice(c, s0, ga + k - 1, out.p, out.n);
            }
        }
    } else if (n == k - 1) {                                                 /* :247 */
        char b = fix_del(c, tbf, L, thr, &out);
        if (b) {
            r->changed = 1; r->n = 1; r->ind[0] = ga;
            r->orig[0] = dupn("d-", 2); r->newb[0] = rep_str(b, 1);          /* :250-253 */
            splice(c, s0, ga + k - 1, out.p, out.n);
        } else {
            char left, right;
            char lr = fixdiploid(c, tbf, L, thr, gb, ga, &left, &right, &out);
            if (lr) {
                r->changed = 1; r->n = 1;
                if (lr == 's') { r->orig[0] = tag_str('s', c->s + ga - 1, 1); r->newb[0] = rep_str(left, 1); r->ind[0] = ga - 1; }
                else { r->orig[0] = tag_str('s', c->s + gb + 1, 1); r->newb[0] = rep_str(right, 1); r->ind[0] = gb + 1; }
                splice(c, s0, ga + k - 1, out.p, out.n);                     /* :265 */
            } else {
                long idx; char *bs;
                if (fix_same_base_insertion(c, tbf, L, thr, &idx, &bs, &out)) { /* :267 */
                    r->changed = 1; r->n = 1; r->ind[0] = idx + s0;
                    r->orig[0] = tag_str('i', bs, (long)strlen(bs)); r->newb[0] = rep_str('-', 1);
                    free(bs);
                    splice(c, s0, ga + k - 1, out.p, out.n);
                }
            }
        }
    } else if (n < k - 1 && n > 1 && L >= k) {                               /* :274 */
        long idx; char *bs;
        if (fix_same_base_del(c, tbf, L, thr, &idx, &bs, &out)) {
            r->changed = 1; r->n = 1; r->ind[0] = idx + s0;
            r->orig[0] = dupn("d-", 2); r->newb[0] = bs;                     /* :277-280 */
            splice(c, s0, ga + k - 1, out.p, out.n);
        } else {
            char left, right;
            char lr = fixdiploid(c, tbf, L, thr, gb, ga, &left, &right, &out);
            if (lr) {
                r->changed = 1; r->n = 1;
                if (lr == 's') { r->orig[0] = tag_str('s', c->s + ga - 1, 1); r->newb[0] = rep_str(left, 1); r->ind[0] = ga - 1; }
                else { r->orig[0] = tag_str('s', c->s + gb + 1, 1); r->newb[0] = rep_str(right, 1); r->ind[0] = gb + 1; }
                splice(c, s0, ga + k - 1, out.p, out.n);
            } else if (fix_same_base_insertion(c, tbf, L, thr, &idx, &bs, &out)) { /* :294 */
                r->changed = 1; r->n = 1; r->ind[0] = idx + s0;
                r->orig[0] = tag_str('i', bs, (long)strlen(bs)); r->newb[0] = rep_str('-', 1);
                free(bs);
                splice(c, s0, ga + k - 1, out.p, out.n);
            }
        }
    } else if (n > k) {                                                      /* :301 */
        long blo, bhi, alo, ahi;
        pyslice(c->len, gb - k + 1, gb + 1, &blo, &bhi);                     /* :302 */
        pyslice(c->len, ga, ga + k, &alo, &ahi);                             /* :303 */
        char *fixed_seq = base_extension(c, L, c->s + blo, bhi - blo, c->s + alo, ahi - alo, thr);
        if (fixed_seq) {
            long fl = (long)strlen(fixed_seq);
            long olo, ohi;
            pyslice(c->len, gb + 1, ga, &olo, &ohi);
            char *ra, *rb;
            align_globalms(fixed_seq, fl, c->s + olo, ohi - olo, &ra, &rb);  /* :309 */
            r->changed = 1; r->is_list = 1; r->n = 0;
            long al = (long)strlen(ra);
            for (long q = 0; q < al; q++) {                                  /* :313-329 */
                char ori = rb[q], chg = ra[q];
                if (chg == ori) continue;
                if (r->n < 2) {
                    r->ind[r->n] = q + gb + 1;
                    if (chg == '-') { r->newb[r->n] = rep_str('-', 1); r->orig[r->n] = tag_str('i', &ori, 1); }
                    else if (ori == '-') { r->newb[r->n] = rep_str(chg, 1); r->orig[r->n] = dupn("d-", 2); }
                    else { r->newb[r->n] = rep_str(chg, 1); r->orig[r->n] = tag_str('s', &ori, 1); }
                }
                r->n++;
            }
            free(ra); free(rb);
            splice(c, gb + 1, ga, fixed_seq, fl);                            /* :312 */
            free(fixed_seq);
        }
    }
    if (out.p) os_free(&out);
}

/* csv.writer(delimiter=' ') QUOTE_MINIMAL field */
static void csv_field(sbuf *b, const char *f) {
    int q = 0;
    for (const char *p = f; *p; p++) if (*p == ' ' || *p == '"' || *p == '\r' || *p == '\n') q = 1;
    if (!*f) q = 0;
    if (!q) { sb_put(b, f, strlen(f)); return; }
    sb_putc(b, '"');
    for (const char *p = f; *p; p++) { if (*p == '"') sb_putc(b, '"'); sb_putc(b, *p); }
    sb_putc(b, '"');
}
static void csv_row(ctx_t *c, long ind, const char *newb, const char *orig, int as_list) {
    char num[32];
    csv_field(c->csv, c->name); sb_putc(c->csv, ' ');
    snprintf(num, sizeof num, "%ld", ind);
    sb_put(c->csv, num, strlen(num)); sb_putc(c->csv, ' ');
    if (as_list) { /* str(['x']) */
        sb_put(c->csv, "['", 2); sb_put(c->csv, newb, strlen(newb)); sb_put(c->csv, "']", 2);
        sb_putc(c->csv, ' ');
        sb_put(c->csv, "['", 2); sb_put(c->csv, orig, strlen(orig)); sb_put(c->csv, "']", 2);
    } else {
        csv_field(c->csv, newb); sb_putc(c->csv, ' '); csv_field(c->csv, orig);
    }
    sb_put(c->csv, "\r\n", 2);
}

/* src/jasper.py:150-223 handle_bad_kmers. Returns new i; *brk = break_while_loop */
static long handle_bad_kmers(ctx_t *c, long i, long *wrong, int fix, long rolling_thre, int *brk) {
    const int k = c->k;
    *brk = 0;
    uint32_t thre = (uint32_t)c->solid;
    if (rolling_thre > 0) thre = (uint32_t)rolling_thre;                     /* :151-153 */
    long j = i - 1;
    uint32_t occ = cnt_seq(c, j, k + j);                                     /* :156 */
    while (occ < thre && j >= 0) { j--; occ = cnt_seq(c, j, k + j); }        /* :157-159 */
    long gb = j + k - 1;                                                     /* :160 */
    uint32_t prev = cnt_seq(c, j, k + j);                                    /* :161 */
    uint32_t kc = cnt_seq(c, i, k + i);                                      /* :162 */
    if (j == -1) gb = -1;                                                    /* :164 */
    if (rolling_thre == 0) {
        while (kc < thre && i < c->len - k + 1) { i++; kc = cnt_seq(c, i, k + i); }   /* :168-170 */
    } else {
        while (kc < thre && i < c->len - k + 1) {                            /* :172 */
            if (i - j > k) return i + 1;                                     /* :173-176 */
            i++; kc = cnt_seq(c, i, k + i);
        }
    }
    long ga = i;                                                             /* :179 */
    /* comparisons with solid_thre/2 and prev/2 are int-vs-float in python: x < t/2  <=>  2x < t */
    if (2 * (uint64_t)cnt_seq(c, gb - k + 2, gb + 2) < (uint64_t)c->solid &&
        2 * (uint64_t)cnt_seq(c, gb - k + 3, gb + 3) < (uint64_t)c->solid) {    /* :182 */
        /* too_low_flag only */
    } else if (rolling_thre == 0) {                                          /* :184 */
        while (2 * (uint64_t)cnt_seq(c, gb - k + 2, gb + 2) >= (uint64_t)prev && gb - k + 1 < ga) { /* :185 */
            if (gb == -1) break;                                             /* :186 */
            if (2 * (uint64_t)prev >= (uint64_t)thre &&
                2 * (uint64_t)cnt_seq(c, gb - k + 2, gb + 2) < (uint64_t)thre &&
                2 * (uint64_t)cnt_seq(c, gb - k + 3, gb + 3) < (uint64_t)thre) break;      /* :188-190 */
            prev = cnt_seq(c, gb - k + 2, gb + 2);                           /* :192 */
            gb++;
        }
        if (gb >= c->len - 1) { *brk = 1; return i; }                        /* :194-195 */
    }
    long s0 = gb - k + 2; if (s0 < 0) s0 = 0;
    if (s0 + k + k >= c->len) { *brk = 1; return s0 + k + k; }               /* :197-198 */
    if (cnt_seq(c, s0 + 1, s0 + k + 1) < thre && cnt_seq(c, s0 + k - 2, s0 + k + k - 2) < thre &&
        cnt_seq(c, s0 + k - 1, s0 + k + k - 1) < thre && cnt_seq(c, s0 + k, s0 + k + k) >= thre)
        ga = s0 + k;                                                         /* :199-205 */
    long tlo, thi;
    pyslice(c->len, s0, ga + k - 1, &tlo, &thi);                             /* :206 */
    long n = ga - s0; if (n < 0) n = 0;                                      /* len(range(s0, ga)) :207 */
    *wrong += n;
    if (fix) {
        if (gb < 0) return i;                                                /* :211-212 */
        char *tbf = dupn(c->s + tlo, thi - tlo);
        fixres r;
        fixing_sid(c, tbf, thi - tlo, thre, n, gb, ga, &r);                  /* :213 */
        free(tbf);
        if (r.changed) {
            if (!r.is_list) {
                csv_row(c, r.ind[0], r.newb[0], r.orig[0], 0);               /* :218-219 (strings) */
            } else if (r.n == 1) {
                csv_row(c, r.ind[0], r.newb[0], r.orig[0], 1);               /* :219 with python lists */
            } else if (r.n == 0) {
                c->fatal = -2;                                               /* IndexError at :221 -> sys.exit(1) */
            } else {
                csv_row(c, r.ind[0], r.newb[0], r.orig[0], 0);               /* :221-222 */
                csv_row(c, r.ind[1], r.newb[1], r.orig[1], 0);
            }
        }
        for (int q = 0; q < 2; q++) { free(r.newb[q]); free(r.orig[q]); }
    }
    return i;                                                                /* :223 */
}

/* src/jasper.py:35-137 iteration, one chunk */
static void walk_chunk(ctx_t *c, int fix, int64_t *total_wrong, int64_t *total_kmers) {
    const int k = c->k;
    *total_kmers += c->len - k + 1;                                          /* :51 */
    long i = 0, wrong = 0;
    while (i < c->len - k + 1 && !c->fatal) {                                /* :55 */
        const char *w = c->s + i;
        long N = -1, nn = -1, bad = -1;
        for (int q = 0; q < k; q++) if (w[q] == 'N') { N = q; break; }       /* :57 */
        if (N >= 0) { i += N + 1; continue; }
        for (int q = 0; q < k; q++) if (w[q] == 'n') { nn = q; break; }      /* :61 */
        if (nn >= 0) { i += nn + 1; continue; }
        for (int q = 0; q < k; q++) if (base_code((unsigned char)w[q]) < 0) { bad = q; break; } /* :65 */
        if (bad >= 0) { i += 1; continue; }
        uint32_t occ = cnt_buf(c, w, k);                                     /* :70-71 */
        int brk = 0;
        if (occ < (uint32_t)c->solid) {                                      /* :73 */
            i = handle_bad_kmers(c, i, &wrong, fix, 0, &brk);
            if (brk) break;
        } else {
            int cond2 = 0;
            if (i > 0) {                                                     /* :80 */
                long a = i - k; if (a < 0) a = 0;
                long b = i > k ? i : k;
                uint32_t pc = cnt_seq(c, a, b);
                cond2 = (50 * (uint64_t)occ < (uint64_t)pc);                 /* occ < pc/50 */
            }
            if (cond2) {
                double sum = 0;                                              /* exact: < 2^53 */
                long ind = i - k; if (ind < 0) ind = 0;
                long num = 0;
                while (ind < i) { num++; ind += c->step; sum += (double)cnt_seq(c, ind, k + ind); } /* :85-88 */
                long rolling = pyround(sum / (double)num / 50.0);           /* :89 */
                if ((long)occ < rolling) {                                   /* :90 */
                    i = handle_bad_kmers(c, i, &wrong, fix, pyround(sum / (double)num / 2.0), &brk); /* :93 */
                    if (brk) break;
                } else i += k - 1;                                           /* :97 */
            } else i += k - 1;                                               /* :100 */
        }
    }
    *total_wrong += wrong;                                                   /* :104 */
}

int jo_polish_batch(const jo_db *db, int k, int n_chunks, const char *const *names, char **seqs,
                    int solid_thre, int passes, int fix, char **csv_out, int64_t qv[4], uint64_t *n_lookups) {
    ctx_t c;
    memset(&c, 0, sizeof c);
    c.db = db; c.k = k; c.solid = solid_thre;
    c.step = (int)pyround((double)k / 8.0); if (c.step < 2) c.step = 2;      /* src/jasper.py:20 */
    long *lens = (long *)malloc(sizeof(long) * (size_t)(n_chunks > 0 ? n_chunks : 1));
    for (int q = 0; q < n_chunks; q++) lens[q] = (long)strlen(seqs[q]);
    qv[0] = qv[1] = qv[2] = qv[3] = 0;
    for (int ite = 0; ite <= passes && !c.fatal; ite++) {                    /* :25 */
        int dofix = fix && ite < passes;                                     /* :37-38 */
        sbuf csv = {0};
        sb_reserve(&csv, 1); csv.p[0] = 0;
        c.csv = &csv;
        int64_t tw = 0, tk = 0;
        for (int q = 0; q < n_chunks && !c.fatal; q++) {
            c.s = seqs[q]; c.len = lens[q]; c.cap = lens[q] + 1; c.name = names[q];
            walk_chunk(&c, dofix, &tw, &tk);
            seqs[q] = c.s; lens[q] = c.len;
        }
        if (ite == 0) { qv[0] = tw; qv[1] = tk; }                            /* :107-111 */
        if (ite == passes) { qv[2] = tw; qv[3] = tk; }
        if (ite < passes && csv_out) csv_out[ite] = csv.p; else free(csv.p);
    }
    free(lens);
    if (n_lookups) *n_lookups = c.nlook;
    return c.fatal;
}

void jo_free(void *p) { free(p); }

/* ================================================================================================
 * Multi-threaded driver (bench.py's cpu_baseline on all host cores; tests check it against the plain path).
 * The algorithm is the restatement above, unchanged; only the work is divided the way the reference divides it:
 *   counting  = `jellyfish count -t N` (src/jasper.sh:177; JF::sub_commands/count_main.cc:152-184: N threads pull
 *               pieces of the read stream and add into one table).  Here: every thread rolls the k-mers of its slice of
 *               the base stream (slices end at a non-ACGT byte, so no window is cut) and files them by key owner; then
 *               every thread adds the keys it owns into its own map -- same counts, no locks.
 *   polishing = `xargs -P N` over batch files (src/jasper.sh:212): chunk records are independent, one worker each.
 * ==============================================================================================*/
#include <pthread.h>

jo_db *jo_mt_db_new(int k, int nshard) {
    jo_db *db = jo_db_new(k);
    if (!db || nshard < 1) { jo_db_free(db); return NULL; }
    db->shard = (jo_db **)calloc((size_t)nshard, sizeof(jo_db *));
    for (int i = 0; i < nshard; i++) db->shard[i] = jo_db_new(k);
    db->nshard = nshard;
    return db;
}

typedef struct { u128 *p; size_t n, cap; } kbuf;
typedef struct {
    jo_db *db; const char *bases; size_t lo, hi; int T, me; kbuf *out; /* out[T*T]: [producer][owner] */
    uint64_t added;
} mt_count_arg;

static void *mt_produce(void *v) {
    mt_count_arg *a = (mt_count_arg *)v;
    const int k = a->db->k, T = a->T;
    const u128 mask = kmask(k);
    u128 fwd = 0, rc = 0;
    int filled = 0;
    kbuf *mine = a->out + (size_t)a->me * T;
    for (int o = 0; o < T; o++) mine[o].n = 0;
    a->added = 0;
    for (size_t i = a->lo; i < a->hi; i++) {
        int c = base_code((unsigned char)a->bases[i]);
        if (c < 0) { filled = 0; continue; }
        fwd = ((fwd << 2) | (u128)c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
        if (filled < k) filled++;
        if (filled >= k) {
            u128 key = fwd < rc ? fwd : rc;
            kbuf *b = mine + db_owner(key, T);
            if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->p = (u128 *)realloc(b->p, b->cap * sizeof(u128)); }
            b->p[b->n++] = key;
            a->added++;
        }
    }
    return NULL;
}
static void *mt_consume(void *v) {
    mt_count_arg *a = (mt_count_arg *)v;
    jo_db *sh = a->db->shard[a->me];
    for (int prod = 0; prod < a->T; prod++) {
        const kbuf *b = a->out + (size_t)prod * a->T + a->me;
        for (size_t i = 0; i < b->n; i++) db_add(sh, b->p[i], 1);
    }
    return NULL;
}

/* same result as jo_db_count_bases on a plain map; db must come from jo_mt_db_new (threads = its number of shards) */
uint64_t jo_mt_count_bases(jo_db *db, const char *bases, size_t n) {
    const int T = db->nshard;
    if (T < 1) return jo_db_count_bases(db, bases, n);
    const size_t round_bases = (size_t)T << 22;     /* 4 M bases per thread and round: 64 MB of keys per thread */
    kbuf *out = (kbuf *)calloc((size_t)T * T, sizeof(kbuf));
    mt_count_arg *args = (mt_count_arg *)calloc((size_t)T, sizeof(mt_count_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    uint64_t added = 0;
    size_t pos = 0;
    while (pos < n) {
        size_t end = pos + round_bases < n ? pos + round_bases : n;
        while (end < n && base_code((unsigned char)bases[end]) >= 0) end++;     /* rounds and slices end between windows */
        size_t cut = pos;
        for (int t = 0; t < T; t++) {
            size_t hi = t == T - 1 ? end : pos + (end - pos) / T * (size_t)(t + 1);
            if (hi < cut) hi = cut;
            while (hi < end && base_code((unsigned char)bases[hi]) >= 0) hi++;
            args[t] = (mt_count_arg){db, bases, cut, hi, T, t, out, 0};
            cut = hi;
        }
        for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, mt_produce, &args[t]);
        for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); added += args[t].added; }
        for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, mt_consume, &args[t]);
        for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        pos = end;
    }
    for (size_t i = 0; i < (size_t)T * T; i++) free(out[i].p);
    free(out); free(args); free(th);
    return added;
}

typedef struct {
    const jo_db *db; int k, n_chunks; const char *const *names; char **seqs; int solid, passes, fix;
    char **csv;          /* [chunk * passes + pass] */
    int64_t (*qv)[4]; uint64_t *nlook; int *rc;
    volatile int *next;
} mt_polish_arg;
static void *mt_polish(void *v) {
    mt_polish_arg *a = (mt_polish_arg *)v;
    for (;;) {
        int q = __sync_fetch_and_add(a->next, 1);
        if (q >= a->n_chunks) break;
        a->rc[q] = jo_polish_batch(a->db, a->k, 1, a->names + q, a->seqs + q, a->solid, a->passes, a->fix,
                                   a->csv + (size_t)q * (size_t)(a->passes > 0 ? a->passes : 1), a->qv[q], a->nlook + q);
    }
    return NULL;
}
/* jo_polish_batch with the chunk records handed to `threads` workers; outputs as jo_polish_batch (rows of a pass in
 * chunk order, counters summed) */
int jo_mt_polish_batch(const jo_db *db, int k, int n_chunks, const char *const *names, char **seqs, int solid_thre, int passes,
                       int fix, char **csv_out, int64_t qv[4], uint64_t *n_lookups, int threads) {
    if (threads < 1) threads = 1;
    const size_t P = (size_t)(passes > 0 ? passes : 1), N = (size_t)(n_chunks > 0 ? n_chunks : 1);
    char **csv = (char **)calloc(N * P, sizeof(char *));
    int64_t (*q4)[4] = (int64_t (*)[4])calloc(N, sizeof(int64_t[4]));
    uint64_t *nl = (uint64_t *)calloc(N, sizeof(uint64_t));
    int *rc = (int *)calloc(N, sizeof(int));
    volatile int next = 0;
    mt_polish_arg a = {db, k, n_chunks, names, seqs, solid_thre, passes, fix, csv, q4, nl, rc, &next};
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, mt_polish, &a);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    int ret = 0;
    qv[0] = qv[1] = qv[2] = qv[3] = 0;
    uint64_t looks = 0;
    for (int q = 0; q < n_chunks; q++) {
        if (rc[q] && !ret) ret = rc[q];
        for (int j = 0; j < 4; j++) qv[j] += q4[q][j];
        looks += nl[q];
    }
    for (int p = 0; p < passes; p++) {
        sbuf all = {0};
        sb_reserve(&all, 1); all.p[0] = 0;
        for (int q = 0; q < n_chunks; q++) {
            char *r = csv[(size_t)q * P + (size_t)p];
            if (r) { sb_put(&all, r, strlen(r)); free(r); }
        }
        if (csv_out) csv_out[p] = all.p; else free(all.p);
    }
    if (n_lookups) *n_lookups = looks;
    free(csv); free(q4); free(nl); free(rc); free(th);
    return ret;
}
