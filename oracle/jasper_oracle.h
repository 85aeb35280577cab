/*
 * jasper_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference hot path
 *     reads -> canonical k-mer counts -> histogram -> threshold -> stride walk / repair -> (bad,total)
 * as implemented by Jellyfish 2.3.0 (vendored tarball, cited "JF::path:line") and by
 * src/jasper.py / src/jellyfish.py of alguoo314/JASPER (cited "src/...:line").
 *
 * Nothing in the product path (jasper_amd/, include/) may include, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED against outputs of the reference itself (real Jellyfish 2.3.0 + unmodified
 * src/jasper.py, run in the build container; see tests/golden/make_golden.py and tests/golden/).
 * UNPINNED corner: the rows written to the fix CSV by the ">k bad k-mers" branch depend on Biopython's
 * Bio.pairwise2 tie-breaking (third party, not vendored, version unpinned; src/jasper.py:309).
 */
#ifndef JASPER_ORACLE_H
#define JASPER_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jo_db jo_db;

/* count database: canonical k-mer -> exact 64-bit count (JF::sub_commands/count_main.cc:152-184) */
jo_db   *jo_db_new(int k);
void     jo_db_free(jo_db *db);
int      jo_db_k(const jo_db *db);
uint64_t jo_db_distinct(const jo_db *db);
/* a3: every window of k consecutive ACGTacgt bytes is counted canonically; any other byte resets
 * (JF::include/jellyfish/mer_iterator.hpp:53-81). Windows do not span calls. Returns #k-mers added. */
uint64_t jo_db_count_bases(jo_db *db, const char *bases, size_t n);
/* a2: FASTA/FASTQ text (format sniffed from first byte) -> records joined by 'N' -> jo_db_count_bases
 * (JF::include/jellyfish/mer_overlap_sequence_parser.hpp:120-307). Returns 0, or <0 on format error.
 * *n_kmers (optional) receives the number of k-mer occurrences added. */
int      jo_db_count_text(jo_db *db, const char *text, size_t n, uint64_t *n_kmers);
/* a8 + Appendix A.3: count of canonical(pad(s)) clamped to 2^32-1; s is truncated at the first
 * non-ACGTacgt byte (or at k) and right-filled with 'A' (JF::include/jellyfish/mer_dna.hpp:525-542,
 * JF::swig/mer_file.i:41, JF::include/jellyfish/binary_dumper.hpp:36-40). */
uint32_t jo_db_query(const jo_db *db, const char *s, long n);
/* add `count` to a canonical k-mer given as exactly k ACGT chars (used to load `jellyfish dump` fixtures) */
int      jo_db_add_kmer(jo_db *db, const char *kmer, uint64_t count);
/* a6: hist[m], m=1..10000, hist[10001] = #distinct with (clamped) count >= 10001; out has 10002 entries
 * (JF::sub_commands/histo_main.cc:34-44,64-84) */
void     jo_db_histo(const jo_db *db, uint64_t *out10002);
/* iterate distinct k-mers: writes k chars + NUL into kmer_out; returns 0 when exhausted */
int      jo_db_next(const jo_db *db, uint64_t *cursor, char *kmer_out, uint64_t *count_out);

/* a1 helpers exposed for known-answer tests: 2-bit encode (first base in MSBs), revcomp, canonical.
 * Words are little-endian 64-bit limbs of the 2k-bit integer: out[0] = low 64 bits. */
int      jo_encode(int k, const char *s, long n, uint64_t out[2]);      /* returns #bases taken before padding */
void     jo_revcomp(int k, const uint64_t in[2], uint64_t out[2]);
void     jo_canonical(int k, const uint64_t in[2], uint64_t out[2]);

/* a7: src/jellyfish.py:8-22. rows = (multiplicity, n_distinct) pairs in file order.
 * Returns the threshold (>=2), 0 when the script prints nothing and exits 0 (no local minimum),
 * or -1 when it exits 1 (threshold < 2). */
int      jo_threshold(const uint64_t *mult, const uint64_t *ndistinct, size_t nrows);

/* a10-a13: src/jasper.py main/iteration/handle_bad_kmers/fixing_sid/... over one batch of chunks.
 *   seqs[i]   in: malloc'ed NUL-terminated chunk i; out: malloc'ed polished chunk (after `passes` fixing passes)
 *   csv_out   receives `passes` malloc'ed strings: the data rows (no header) of _iter{p}_*.fix.csv, CRLF ends
 *   qv        receives bad0,total0,badP,totalP  (src/jasper.py:107-111)
 * Returns 0, or the reference's failure mode as a negative code (-2: IndexError at src/jasper.py:221). */
int      jo_polish_batch(const jo_db *db, int k, int n_chunks, const char *const *names,
                         char **seqs, int solid_thre, int passes, int fix,
                         char **csv_out, int64_t qv[4], uint64_t *n_lookups);
void     jo_free(void *p);

/* Multi-threaded driver around the same restatement (bench.py's cpu_baseline on all host cores): the work divided as
 * `jellyfish count -t N` (src/jasper.sh:177) and `xargs -P N` over batch files (src/jasper.sh:212) divide it.
 * jo_mt_db_new: a map split into `threads` maps by key owner; jo_db_query / jo_db_histo / jo_db_distinct /
 * jo_polish_batch work on it, jo_db_next does not. */
jo_db   *jo_mt_db_new(int k, int threads);
uint64_t jo_mt_count_bases(jo_db *db, const char *bases, size_t n);
int      jo_mt_polish_batch(const jo_db *db, int k, int n_chunks, const char *const *names, char **seqs, int solid_thre,
                            int passes, int fix, char **csv_out, int64_t qv[4], uint64_t *n_lookups, int threads);

#ifdef __cplusplus
}
#endif
#endif
