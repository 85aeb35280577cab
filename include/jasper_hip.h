/*
 * jasper_hip.h -- C-ABI of libjasper_hip.so, the MI355X-native replacement for the hot path of
 * alguoo314/JASPER:   reads -> canonical k-mer counts (HBM table) -> histogram -> scan / lookup / fix -> (bad,total).
 *
 * Plain C, opaque handles, caller-owned outputs, no exceptions, no torch types.  Every entry point returns
 * 0 on success and a negative code on failure; jasper_last_error() then returns a message (thread-local).
 * A handle may be used from one host thread at a time; all work of a handle is issued on its own HIP stream
 * of its own device.
 *
 * What each entry point replaces in the reference ("src/..." = /root/reference/src, "JF::..." = inside the
 * vendored jellyfish-2.3.0.tar.gz):
 *
 *   jasper_table_create / _destroy     `jellyfish count -s SIZE -m K` table set-up          JF::sub_commands/count_main.cc:258-283
 *                                      and jf.QueryMerFile(path) open / close               JF::swig/mer_file.i:18-36
 *   jasper_count_reads_files           `zcat -f READS | jellyfish count -C -m K /dev/stdin` src/jasper.sh:177
 *   jasper_count_reads_text            the same on an in-memory FASTA/FASTQ stream          JF::include/jellyfish/mer_overlap_sequence_parser.hpp:120-307
 *   jasper_count_bases[_device]        mer_counter_base::start hot loop on parsed bases     JF::sub_commands/count_main.cc:152-184
 *   jasper_histogram                   `jellyfish histo`                                    JF::sub_commands/histo_main.cc:34-44,64-84 (src/jasper.sh:177,189)
 *   jasper_lookup                      qf[jf.MerDNA(s).get_canonical()]                     JF::swig/mer_file.i:41, JF::swig/mer_dna.i:12-19 (src/jasper.py:70-71 ...)
 *   jasper_table_export/_import[_device]  `jellyfish merge` (sum by key)                    JF::jellyfish/merge_files.cc:44-176 -> multi-GPU table merge
 *   jasper_polish_batch + jasper_result_*   one `jasper.py --db DB --query BATCH ...` process   src/jasper.py:12-137 (invoked at src/jasper.sh:207-212)
 *   jasper_asm_*                       the perl one-liners around it: batch size, split, join        src/jasper.sh:132,155-156,220; src/jasper.py:120-128
 */
#ifndef JASPER_HIP_H
#define JASPER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jasper_table jasper_table;
typedef struct jasper_result jasper_result;

/* one repair, as src/jasper.py:218-222 records it ([seqname, index, fixed_base, original]) */
typedef struct jasper_fixrec {
    int64_t index;     /* Base_coord: chunk- and pass-relative */
    uint32_t chunk;    /* position of the chunk in the batch */
    uint32_t seqno;    /* emission order inside (chunk, pass) */
    uint8_t pass;
    uint8_t kind;      /* 's' substitution  'i' inserted base(s) removed  'd' deleted base(s) restored  'x' path extension */
    uint8_t newc;      /* 's': new base   'd': restored base, repeated rep times */
    uint8_t oldc;      /* 's': old base   'i': removed base, repeated rep times */
    uint32_t rep;      /* 'x': length of the replaced original segment */
    uint32_t aux_off;  /* 'x': offset of  patch ++ original segment  in the chunk's aux bytes */
    uint32_t aux_len;  /* 'x': length of the patch */
} jasper_fixrec;

#define JASPER_OK 0
#define JASPER_ERR -1           /* HIP / IO / argument error */
#define JASPER_ERR_CAPACITY -2  /* table or scratch too small (message says which) */
#define JASPER_ERR_FORMAT -3    /* "Unsupported format" / "Invalid fastq sequence" (Jellyfish's wording) */
#define JASPER_ERR_REFERENCE_EXIT -4 /* the reference itself would exit(1) on this input (src/jasper.py:221 IndexError) */

const char *jasper_last_error(void);
int jasper_device_count(int *n);
/* on != 0: the long-running calls of OTHER threads (jasper_count_reads_files, jasper_table_write_jf) return JASPER_ERR ("cancelled")
 * at their next chunk / block instead of finishing -- for a driver that exits on an error elsewhere, as src/jasper.sh:23-28 kills its
 * children (`trap abort`); on == 0 re-arms.  Process-wide. */
int jasper_request_cancel(int on);
/* free and total device memory in bytes (hipMemGetInfo): lets the driver that replaces src/jasper.sh decide whether two stages
 * that the reference runs one after the other (`tee $JF_DB` at :177, then the jasper.py processes at :207-212) fit side by side */
int jasper_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* k in [1,64]; min_slots is a size hint like `jellyfish count -s` (rounded up to a power of two; the table
 * doubles by itself when half full).  k > 26 needs at least 2^(2k-53) slots (k=37: 2^21 = 32 MiB). */
int jasper_table_create(int k, uint64_t min_slots, int device, jasper_table **out);
/* open a Jellyfish "binary/sorted" database (jasper.sh -j, or an existing mer_counts$K.jf) into a new HBM table; k comes from
 * the file header like jf.QueryMerFile(path) does (JF::swig/mer_file.i:18-36); errors use Jellyfish's wording */
int jasper_table_load_jf(const char *path, int device, jasper_table **out);
void jasper_table_destroy(jasper_table *t);
/* the table as a Jellyfish "binary/sorted" database: what `jellyfish count -o mer_counts$K.jf` leaves behind
 * (src/jasper.sh:177; format JF::include/jellyfish/file_header.hpp:26-108, binary_dumper.hpp:36-40,148-199), readable by
 * jellyfish 2.3.0 query/dump/histo, QueryMerFile and jasper_table_load_jf.  cmdline[] is recorded in the header. */
/* records [n*part/nparts, n*(part+1)/nparts) of the database only: one GPU's shard of an existing DB (dist.shard_tables
 * then routes every key to its owner) */
int jasper_table_load_jf_part(const char *path, int device, uint32_t part, uint32_t nparts, jasper_table **out);
int jasper_table_write_jf(jasper_table *t, const char *path, const char *const *cmdline, int n_cmdline);
/* test hook (host arithmetic only, no GPU): the table's bijective k-mer hash (inverse = 0) or its inverse (1) */
int jasper_debug_mix(int k, int inverse, uint64_t hi, uint64_t lo, uint64_t out2[2]);
/* distinct = keys in this table; occurrences = k-mer occurrences this table's counting calls have SCANNED.  For an owner shard
 * filled by the exchange (jasper_count_exchange_*) that is what this GPU read and sent to all owners, not the occurrences of the
 * keys the shard holds (those are the sum of its counts: jasper_histogram). */
int jasper_table_info(jasper_table *t, int *k, uint64_t *slots, uint64_t *distinct, uint64_t *occurrences);
int jasper_table_sync(jasper_table *t);
/* forget every k-mer (slots and counters zeroed in place; capacity kept) */
int jasper_table_clear(jasper_table *t);

/* bases: concatenated read sequences, records separated by any non-ACGTacgt byte; windows do not span calls */
int jasper_count_bases(jasper_table *t, const char *bases, uint64_t n);
/* the same with the bases already resident in this table's device memory (16-byte aligned for full speed) */
int jasper_count_bases_device(jasper_table *t, const void *d_bases, uint64_t n);
int jasper_count_reads_text(jasper_table *t, const char *text, uint64_t n);
int jasper_count_reads_files(jasper_table *t, const char *const *paths, int n_paths);
/* the same over a byte range [begins[i], ends[i]) of every file (ends[i] < 0: to its end) -- one GPU's shard of the reads.
 * Ranges must start at record boundaries (jasper_amd/dist.py plan_read_shards finds them); a gzip file cannot be cut: it
 * is read whole if its range starts at 0 and skipped otherwise.  Counts are sums over reads, so the shards' tables add up
 * to the table of the whole input. */
int jasper_count_reads_file_ranges(jasper_table *t, const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths);

/* of the last jasper_count_reads_files call: text bytes parsed by the GPU kernels / by the host state machine (the
 * fallback for multi-line records, DOS line ends, malformed input and stream tails) */
int jasper_last_ingest(jasper_table *t, uint64_t *gpu_bytes, uint64_t *host_bytes);
int jasper_histogram(jasper_table *t, uint64_t *out10002);
/* the same over the keys of ONE owner partition (as in jasper_table_export_packed): after a multi-GPU merge every rank
 * bins the range it owns and the 10002 bins are summed over ranks, instead of every rank scanning the whole table */
int jasper_histogram_part(jasper_table *t, uint32_t part, uint32_t nparts, uint64_t *out10002);
/* 1 if the histogram is already known because the last counting call binned the final counts while it wrote them
 * (one partitioned pass over the whole input into an empty table); jasper_histogram then costs one small copy */
int jasper_histogram_is_fused(jasper_table *t);
/* string i is chars[offsets[i] .. offsets[i+1]); out[i] = count of canonical(pad(string i)) clamped to 2^32-1 */
int jasper_lookup(jasper_table *t, const char *chars, const int64_t *offsets, uint64_t n, uint32_t *out);

/* entries are 3 x uint64 each: mixed-hash high word, low word, exact count.  Any table with the same k accepts them. */
int jasper_table_export(jasper_table *t, uint64_t *n_entries, uint64_t *host_entries /* NULL: size query */);
int jasper_table_import(jasper_table *t, const uint64_t *host_entries, uint64_t n_entries);
int jasper_table_export_device(jasper_table *t, uint64_t *n_entries, void **d_entries);
int jasper_table_import_device(jasper_table *t, const void *d_entries, uint64_t n_entries);
/* export into caller-owned device memory (e.g. a torch tensor handed to RCCL); cap_entries = room in d_dst */
int jasper_table_export_to(jasper_table *t, void *d_dst, uint64_t cap_entries, uint64_t *n_entries);
int jasper_device_free(jasper_table *t, void *d_ptr);
/* multi-GPU exchange format: 16-byte entries { hash.lo, hash.hi | count << max(0, 2k-64) } in device memory.
 * export: keys whose home slot lies in slot-range partition `part` of `nparts` (nparts = 1: all); *n_entries is the
 * number that exists (may exceed cap_entries: call once with cap 0 to size the buffer).  import mode 0 adds the
 * counts (key-wise sum), mode 1 sets them (an owner's final counts replace this table's partial ones). */
int jasper_table_export_packed(jasper_table *t, void *d_dst, uint64_t cap_entries, uint64_t *n_entries, uint32_t part, uint32_t nparts);
int jasper_table_import_packed(jasper_table *t, const void *d_src, uint64_t n_entries, int mode);
/* add up to 8 entry lists in ONE sweep over the table (an owner adding what every rank sent it): the lists are in slot order
 * of same-hash tables, so their c-th parts land in the same band of slots, which stays in cache while all lists update it */
int jasper_table_import_packed_multi(jasper_table *t, const void *const *d_srcs, const uint64_t *counts, uint32_t n_src);
/* grow to at least min_slots slots (ranks agree on one geometry before exchanging slot-range partitions) */
int jasper_table_reserve(jasper_table *t, uint64_t min_slots);
/* rehash into the smallest slot count that holds the present keys at a load of at most max_load (0.05 .. 0.9); may shrink */
int jasper_table_fit(jasper_table *t, double max_load);

/* Owner-sharded table ("keep the table key-sharded and route lookups").  Instead of replicating the merged
 * table on every GPU, owner o of n keeps ONLY the keys with jasper_owner_of(hash) == o, and every lookup -- the polishing
 * kernels', jasper_lookup's -- reads the owner's slot array directly: its own HBM, or a peer's over xGMI.
 *   jasper_table_export_owner: all entries of t in the exchange format above, grouped by owner; segment o starts at
 *     d_dst + o * cap_entries * 16 bytes and holds counts[o] entries (if any counts[o] > cap_entries nothing beyond cap was
 *     written: call again with more room).  One pass over the table whatever n_owners is.
 *   jasper_table_ipc_handle: 64 bytes that let another PROCESS map this table's slot array (hipIpcGetMemHandle).
 *   jasper_table_attach_ipc: handles = n x 64 bytes in owner order (entry `self` unused).  All owners' tables must have
 *     the geometry of t (same k, same slot count: jasper_table_reserve after agreeing on the maximum).
 *   jasper_table_attach_tables: the same for shard tables living in this process.
 *   jasper_table_detach: back to a whole table.  Growing a table detaches it.
 * The owners' tables must not be written while any GPU reads them; that ordering is the caller's (a barrier). */
int jasper_table_export_owner(jasper_table *t, void *d_dst, uint64_t cap_entries, uint32_t n_owners, uint64_t *counts);
/* The read files as a FEED of base batches in HBM (role of `zcat -f $READS |` in front of a counter that is driven from outside,
 * src/jasper.sh:177): the reader / inflater / parsers of jasper_count_reads_file_ranges run in a thread of their own, but every
 * batch of bases they would have counted into t is handed to the caller instead.  `t` only lends its device and buffers (its
 * table is not touched).  begins / ends: per-file byte ranges as for jasper_count_reads_file_ranges, or both NULL.
 *   jasper_read_feed_next     waits for the next batch: *d_bases (text bases with a non-base byte between records, as
 *                             jasper_count_bases_device takes them; no k-mer spans two batches), *n bytes.  *n == 0: the stream
 *                             has ended and the feed is closed; a reader / parser error is returned here.
 *   jasper_read_feed_release  the caller is done with the batch: its memory is reused for the next one. */
int jasper_read_feed_start(jasper_table *t, const char *const *paths, const int64_t *begins, const int64_t *ends, int n_paths);
int jasper_read_feed_next(jasper_table *t, const void **d_bases, uint64_t *n);
int jasper_read_feed_release(jasper_table *t);
/* Counting on several GPUs WITHOUT per-GPU tables that are merged afterwards (role of `jellyfish count` over all reads,
 * src/jasper.sh:177, and of JF::jellyfish/merge_files.cc:44-96): the reads of every GPU go through the two partition passes
 * of the atomic-free counting path; the second pass also groups by owner, so what owner o is to receive is one contiguous
 * block of the send buffers -- the region lists of o's own table.  After ONE all_to_all of those blocks (8 bytes per k-mer
 * occurrence plus the slack of the lists) every owner inserts what it received straight into its shard `t`.  All ranks call
 * with shard tables of one geometry and the same piece_max (the longest piece [pos, end) any of them scans in this round)
 * and records_max (the most records any rank's scan returned; 0 = not known, piece_max stands in: the lists, and with them
 * the bytes that travel, are then sized for the worst case).
 *   jasper_count_exchange_plan      out8 = { records (8 B) per owner block, slice counts (4 B) per owner block, deferred entries
 *                                   (24 B) to provide room for, p1, p2 (+ 256 x the second-level bits left to an extra pass
 *                                   on the owner: very large shards), region bits, slices per list, slice capacity };
 *                                   returns 1 (not an error) when this table / piece size / k has no such geometry: count into a
 *                                   table per GPU and use jasper_table_export_owner instead.
 *   jasper_count_exchange_scan      first pass over bases [pos, end) of d_bases (n bytes, text bases as for
 *                                   jasper_count_bases_device; the k-1 bases before pos are read as context) into lists kept
 *                                   inside t; *records = k-mer occurrences found.  d_deferred: 64-byte header (word 0 =
 *                                   entries), then entries of 3 words hash.hi, hash.lo, increment -- the few records that found
 *                                   no room in their list, here or in the next call.  Returns when the pass is done.
 *   jasper_count_exchange_partition second pass: the lists of the scan -> d_send (n_owners blocks of records), d_send_counts
 *                                   (n_owners blocks of counts); asynchronous like the counting calls (jasper_table_sync).
 *   jasper_count_exchange_dedupe    optional, between partition and the all_to_all: every list of d_send is deduplicated in
 *                                   place -- one record per distinct key, (occurrences - 1) in *count_bits of the record's bits
 *                                   that its list implies -- the counts are updated and *max_fill = records in the fullest
 *                                   list: only that many per list need to travel (the caller packs [list][slice capacity] to
 *                                   [list][max over ranks of max_fill]).  Returns 1 (not an error, nothing done) when the
 *                                   geometry has no such bits.  Returns when the pass is done.
 *   jasper_count_exchange_insert    d_recv / d_recv_counts: block s = what rank s put into its block `self`; d_deferred_all: the
 *                                   deferred entries of ALL ranks back to back (the ones owned by `self` are added);
 *                                   whole_input != 0: these lists are all that goes into the (empty) shard, so the multiplicity
 *                                   histogram is taken on the way (jasper_histogram_is_fused).  slice_cap / count_bits: 0, or the
 *                                   packed slice capacity and the count bits after jasper_count_exchange_dedupe. */
int jasper_count_exchange_plan(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, uint64_t *out8);
int jasper_count_exchange_scan(jasper_table *t, const void *d_bases, uint64_t n, uint64_t pos, uint64_t end, uint64_t piece_max, uint32_t n_owners, void *d_deferred,
                               uint64_t deferred_cap, uint64_t *records);
int jasper_count_exchange_partition(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, void *d_send, void *d_send_counts, void *d_deferred,
                                    uint64_t deferred_cap);
int jasper_count_exchange_dedupe(jasper_table *t, uint64_t piece_max, uint64_t records_max, uint32_t n_owners, void *d_send, void *d_send_counts, uint32_t *max_fill,
                                 int *count_bits);
int jasper_count_exchange_insert(jasper_table *t, const void *d_recv, const void *d_recv_counts, uint64_t piece_max, uint64_t records_max, uint32_t n_owners,
                                 uint32_t self, const void *d_deferred_all, uint64_t n_deferred_all, int whole_input, uint32_t slice_cap, int count_bits);
/* A binary/sorted database written by several GPUs: the file order (pos, key) with `size` = 2^size_log2 is the numeric order
 * of the key rotated right by size_log2 bits, so cutting the value range of the key's low size_log2 bits into n_ranges equal
 * parts cuts the file into n_ranges consecutive pieces.  jasper_table_export_file_ranges groups the entries of t by that
 * range (same layout and conventions as jasper_table_export_owner); after an all_to_all each GPU holds one range, sorts it
 * and writes it with jasper_table_write_jf_piece (what = 1: records only; 2: the header only; 0: both). */
int jasper_table_export_file_ranges(jasper_table *t, void *d_dst, uint64_t cap_entries, uint32_t n_ranges, int size_log2, uint64_t *counts);
int jasper_table_write_jf_piece(jasper_table *t, const char *path, const char *const *cmdline, int n_cmdline, int size_log2, int what);
int jasper_table_ipc_handle(jasper_table *t, void *out64);
int jasper_table_attach_ipc(jasper_table *t, const void *handles, uint32_t n, uint32_t self);
int jasper_table_attach_tables(jasper_table *t, jasper_table *const *shards, uint32_t n, uint32_t self);
int jasper_table_detach(jasper_table *t);
/* open and close the peers' handles without a table: run from a throw-away process with a time limit before the real
 * attach (jasper_amd/dist.py), so that a mapping call that never returns costs a killed helper, not a hung GPU process */
int jasper_ipc_probe(int device, const void *handles, uint32_t n, uint32_t self);
/* owner of a packed entry's hash among n (host-side restatement of the device function, for tests and routing) */
uint32_t jasper_owner_of(uint64_t hash_lo, uint64_t hash_hi, uint32_t n);

/* one batch of chunk records through `passes` fixing passes + the final QV pass (src/jasper.py:25-26) */
int jasper_polish_batch(jasper_table *t, int n_chunks, const char *const *seqs, const int64_t *lens,
                        int solid_thre, int passes, int fix, jasper_result **out);
/* the same with the chunk records already in HBM on the table's device (chunk c = d_text[offsets[c] .. offsets[c+1]),
 * offsets is a host array of n_chunks+1 entries).  The polished text is left in HBM: jasper_result_seq_device points at
 * it, jasper_result_seq copies it to the host on first use.  It lies in the table's workspace, so the library copies it
 * to the host by itself before the next polish call on the same table (or jasper_table_destroy) would overwrite it;
 * after that jasper_result_seq_device fails and jasper_result_seq still works. */
int jasper_polish_batch_device(jasper_table *t, int n_chunks, const void *d_text, const int64_t *offsets,
                               int solid_thre, int passes, int fix, jasper_result **out);
int jasper_result_num_chunks(const jasper_result *r);
int jasper_result_seq(const jasper_result *r, int chunk, const char **seq, int64_t *len);
int jasper_result_seq_len(const jasper_result *r, int chunk, int64_t *len);
int jasper_result_seq_device(const jasper_result *r, int chunk, const void **d_seq, int64_t *len);
int jasper_result_records(const jasper_result *r, const jasper_fixrec **recs, uint64_t *n);
int jasper_result_aux(const jasper_result *r, int chunk, const char **aux, uint64_t *n);
int jasper_result_qv(const jasper_result *r, int64_t out4[4]); /* bad0,total0,badP,totalP  (src/jasper.py:107-111) */
/* the same four counters for one chunk record (a caller that polishes several batch files in one call splits them again) */
int jasper_result_qv_chunk(const jasper_result *r, int chunk, int64_t out4[4]);
int jasper_result_lookups(const jasper_result *r, uint64_t *n);
double jasper_result_seconds(const jasper_result *r);         /* device time of the passes (HIP events) */
/* how the batch was parallelised: segments walked over all passes, chunks redone unsegmented after a failed speculation */
int jasper_result_segments(const jasper_result *r, uint64_t *n_segments, uint64_t *n_respeculated);
/* 1 if the batch had to be repeated with larger internal buffers (results are the same either way) */
int jasper_result_retried(const jasper_result *r);
void jasper_result_free(jasper_result *r);

/* The assembly side of src/jasper.sh, natively and by several host threads (no GPU call except jasper_asm_polish):
 *   jasper_asm_open          the assembly FASTA read once into ONE host arena (line ends taken out, contigs back to back).  Returns 1
 *                            (not an error, *out = NULL) for anything but the ordinary file -- '\r', a first byte that is not '>',
 *                            blanks / tabs / non-printable / non-ASCII bytes in sequence lines, odd bytes in header lines, a contig
 *                            name that occurs twice -- for which the caller applies the line-by-line rules of the perl one-liners itself.
 *   jasper_asm_info          *sequence_bytes = `grep -v '^>' $QUERY | tr -d '\n' | wc` third column      src/jasper.sh:132
 *   jasper_asm_contig        name = first whitespace token of the header line WITH its '>' (perl -ane $F[0])  src/jasper.sh:155
 *   jasper_asm_split         perl #1: chunk records ">name:offset" of <= batch_size bases at offsets 0, bs, 2bs ..; perl #2: batch files
 *                            `$prefix.batch.N.fa`, a new one at a record once MORE than batch_size bases are in the current one
 *                            (src/jasper.sh:155-156).  write_files != 0: the files (all, or only_files[0..n_only)) are written by a
 *                            thread of the job while the caller goes on; jasper_asm_split_wait joins it and reports its failure
 *                            ("Splitting files failed", src/jasper.sh:159).
 *   jasper_asm_chunks        per record: contig index, offset, length, batch file (caller's arrays of n_chunks; any may be NULL)
 *   jasper_asm_file_bytes    size of every batch file (what `ls -l` would show; dist.assign_chunks balances by it)
 *   jasper_asm_chunk_text    a record's text: the input (polished = 0) or what jasper_asm_take kept (polished = 1)
 *   jasper_asm_polish        jasper_polish_batch on the records of the listed batch files, read straight from the arena: one
 *                            `jasper.py --query $prefix.batch.N.fa` process per listed file (src/jasper.sh:207-212); result chunk i =
 *                            the i-th record of the files in list order
 *   jasper_asm_pin           optional, any thread (it waits for the GPU runtime to start): registers the arena with the runtime and
 *                            allocates one pinned buffer for the polished text, so that record text crosses PCIe without staging
 *                            copies in either direction
 *   jasper_asm_take          moves the polished text out of the result into the job (for the two writers below); a result of
 *                            jasper_asm_polish leaves its text in HBM until this call copies it
 *   jasper_asm_put           the polished text of one record given by the caller instead (read back from an `_iter*.fixed.fa` of
 *                            an interrupted run; tests)
 *   jasper_asm_write_fixed   `_iter{P-1}_<batch>.fixed.fa`: ">name:offset" + lines of 60            src/jasper.py:120-128,142-147
 *   jasper_asm_polished_lens per record: length of the polished text held (0 if none), and whether it is held
 *   jasper_asm_join          `$QUERY_FN.polished.fasta`: per contig ">name", its records in offset order on ONE line   src/jasper.sh:220
 *                            (contigs in input order; the reference's order is perl's hash order).  all_lens: the polished length of
 *                            EVERY record (NULL: this job holds them all).  mode 1 creates the file at its final size, mode 2 writes
 *                            the records this job holds at their places (several processes, one per GPU, into one file), 3 = both. */
typedef struct jasper_asm jasper_asm;
int jasper_asm_open(const char *path, int threads, jasper_asm **out);
void jasper_asm_close(jasper_asm *a);
int jasper_asm_info(const jasper_asm *a, uint64_t *sequence_bytes, uint64_t *n_contigs, uint64_t *n_bases);
int jasper_asm_contig(const jasper_asm *a, uint64_t i, const char **name, uint64_t *name_len, uint64_t *n_bases);
int jasper_asm_split(jasper_asm *a, uint64_t batch_size, const char *prefix, const uint32_t *only_files, uint32_t n_only, int write_files, int threads,
                     uint64_t *n_chunks, uint64_t *n_files);
int jasper_asm_split_wait(jasper_asm *a);
int jasper_asm_chunks(const jasper_asm *a, uint32_t *contig, uint64_t *ci, uint64_t *len, uint32_t *file);
int jasper_asm_file_bytes(const jasper_asm *a, uint64_t *bytes);
int jasper_asm_chunk_text(const jasper_asm *a, uint64_t chunk, int polished, const char **text, uint64_t *len);
int jasper_asm_polish(jasper_table *t, jasper_asm *a, const uint32_t *files, uint32_t n_files, int solid_thre, int passes, int fix, jasper_result **out);
int jasper_asm_pin(jasper_asm *a, int device);
int jasper_asm_take(jasper_asm *a, jasper_result *r, const uint32_t *files, uint32_t n_files);
int jasper_asm_put(jasper_asm *a, uint64_t chunk, const char *text, uint64_t len);
int jasper_asm_write_fixed(jasper_asm *a, const uint32_t *files, const char *const *out_paths, uint32_t n_files, int threads);
int jasper_asm_polished_lens(const jasper_asm *a, uint64_t *lens, uint8_t *have);
int jasper_asm_join(jasper_asm *a, const char *out_path, const uint64_t *all_lens, int mode, int threads);
/* `$QUERY_FN.fixes.csv` from the per-batch fix CSVs (src/jasper.sh:222-226: awk | awk -F: | sort -k1,1 -k2,2n -k3,3n | awk), byte order
 * for names and for sort's last-resort whole-line comparison.  Returns 1 (not an error, nothing written) when a line holds bytes
 * other than printable ASCII / blank / tab / '\r' or a number of more than 18 digits: the caller applies the rules itself. */
int jasper_merge_fix_csvs(const char *const *paths, uint32_t n_paths, const char *out_path);

/* kernel timing for bench.py: HIP-event time of the last counting call on this table, and its launch count */
int jasper_last_count_timing(jasper_table *t, double *kernel_ms, uint64_t *launches);
/* the same split by kernel stage of the atomic-free counting paths, and which path the last piece took (*path):
 *   1 = one record per occurrence (count_part.hip): part1, part2, region_insert, deferred, (unused x4)
 *   3 = the same with the region lists exchanged between GPUs: part1, part2 by owner (+ dedupe), region_insert, (unused), deferred
 *   0 = count_kernel (global atomics; small pieces)
 * *partitioned_launches of the counted launches took path 1 or 3 (the others ran count_kernel) */
int jasper_last_count_stages(jasper_table *t, double stage_ms[8], uint64_t *partitioned_launches, int *path);

/* A slot array whose IPC handle was given out (jasper_table_ipc_handle) and that the table has outgrown since is kept until
 * this call: the owners make it after they have all attached to the new arrays (jasper_amd/dist.py: shard_tables), so that
 * nothing a peer may still have mapped is ever freed under it.  jasper_table_destroy releases them too. */
int jasper_table_release_retired(jasper_table *t);

/* Host only (no GPU touched): one gzip file inflated by `threads` threads into out_path (or nowhere when out_path is null),
 * the way jasper_count_reads_files reads a large .gz -- the role of `zcat -f` in src/jasper.sh:177.  chunk_bytes = compressed
 * bytes per unit of work (0: default 4 MiB).  *n_out = inflated bytes; *parallel = 1 when the many-thread reader handled the
 * file, 0 when it declined (small file, not a regular gzip file) and zlib's reader was used. */
int jasper_inflate_file(const char *path, int threads, uint64_t chunk_bytes, const char *out_path, uint64_t *n_out, int *parallel);

#ifdef __cplusplus
}
#endif
#endif
