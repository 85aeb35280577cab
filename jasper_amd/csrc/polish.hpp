// polish.hpp -- data structures shared by the polishing kernels (polish.hip) and the C-ABI (capi.hip).
#pragma once
#include "table.hpp"

namespace jk {

// one fix record (src/jasper.py:218-222 appends [seqname, index, fixed_base, original]); the host turns
// these into CSV rows. 32 bytes.
struct FixRec {
    int64_t index;     // Base_coord (chunk- and pass-relative, as in the reference)
    uint32_t chunk;
    uint32_t seqno;    // order of emission inside (chunk, pass)
    uint8_t pass;
    uint8_t kind;      // 's' substitution, 'i' inserted base(s) removed, 'd' deleted base(s) restored, 'x' path extension
    uint8_t newc;      // 's': new base; 'd': restored base (repeated `rep` times)
    uint8_t oldc;      // 's': old base; 'i': removed base (repeated `rep` times)
    uint32_t rep;      // 'x': length of the original segment
    uint32_t aux_off;  // 'x': offset of  patch ++ original segment  in the aux bytes
    uint32_t aux_len;  // 'x': length of the patch
};
static_assert(sizeof(FixRec) == 32, "FixRec layout");

enum PolishStatus {
    PS_OK = 0,
    PS_GAP_EXHAUSTED = 1,   // text grew beyond its slack
    PS_REC_OVERFLOW = 2,    // more fix records than reserved
    PS_AUX_OVERFLOW = 3,
    PS_BFS_ARENA = 4,       // path-extension search ran out of scratch
    PS_REF_INDEXERROR = 5,  // the reference raises IndexError here (src/jasper.py:221) and exits 1
    PS_STRING_TOO_LONG = 6
};

// position classes written by the dense scan (one byte per window start of a chunk, per pass)
enum PosClass : uint8_t {
    PC_CLEAN = 0,  // valid window, count >= solid_thre, and not (count < count(window k back)/50): the walk does `i += k-1`
    PC_BAD = 1,    // valid window, count < solid_thre
    PC_OTHER = 2   // non-ACGT window, or the k-back test fires / cannot be decided densely: evaluate exactly
};

// scratch of the path-extension search (src/jasper.py:527-583), pooled: a wave takes a slot only while it searches
struct ScratchPool {
    unsigned int *locks;   // 0 free, 1 taken
    uint8_t *base;
    size_t stride;         // bytes per slot
    uint32_t nslots;
    uint32_t node_cap;     // trie arena entries (uint32)
    uint32_t front_cap;    // frontier entries (80 B)
    uint32_t patch_cap;    // bytes
    size_t off_front, off_patch;
};

// One SEGMENT of a chunk record for one pass.  A chunk is cut at "sync points": starts of runs of >= k-1 bad k-mers
// that are preceded by >= 4k clean positions.  Whatever stride phase the reference's walk arrives with, it lands inside
// such a run and handle_bad_kmers() then does the same thing -- so the walk to the right of a sync point does not
// depend on anything to its left except a coordinate shift, and segments can be walked concurrently.
// one text replacement of a segment walk, in the segment's local coordinates AT THE TIME of the edit: local
// [a, a+oldlen) became plen new bytes.  The stitch step replays the list backwards to carry the position classes of
// untouched windows over to the next pass instead of recomputing them.
struct EditRec {
    int64_t a;
    int32_t plen, oldlen;
};

struct SegDev {
    uint8_t *buf;        // gap buffer holding a private copy of the segment's text (local coordinates)
    int64_t cap;
    int64_t len;         // logical length (local)
    int64_t gs, glen;    // gap start / length
    int64_t len0;        // logical length before the pass
    int64_t seg_lo;      // chunk coordinate (at pass start) of local position 0
    int64_t start_i;     // local position where the walk starts (0 for the first segment of a chunk)
    int64_t stop_orig;   // the walk ends when it arrives at an event whose pass-start chunk coordinate is >= this
    int32_t first, last; // first / last segment of its chunk: chunk-start / chunk-end semantics are real there
    // A segment whose left boundary is a CLEAN-ZONE boundary (not a sync point) does not know where the walk arrives: it
    // takes the arrival position its predecessor publishes (chain_in = 1).  Every segment publishes the chunk coordinate
    // (pass-start text) at which it handed over in *arrive: ARRIVE_PENDING until then, ARRIVE_FAIL if it gave up.
    int32_t chain_in;
    int32_t took_shortcut;   // out: the range held no event, the arrival was computed instead of walked
    long long *arrive;   // this segment's slot; the predecessor's is arrive[-1]
    const uint8_t *cls;  // PosClass per window start of the chunk at pass start (chunk coordinates), cls_n entries
    int64_t cls_n;
    FixRec *recs;
    uint32_t rec_cap, nrec;
    uint8_t *aux;
    uint32_t aux_cap, naux;
    EditRec *edits;
    uint32_t edit_cap, nedit;
    uint32_t chunk;
    int32_t status;
    int32_t spec_fail;   // the walk touched text outside what the segment may assume -> redo the chunk unsegmented
    int64_t wrong;       // bad k-mers counted in this segment during this pass
    uint64_t lookups;
    uint64_t ticks;      // wall_clock64() ticks (100 MHz) this segment's walk took -- tuning aid
    uint64_t tk[12];     // of which: [0] skipping good k-mers, [1] finding the bad run, [2] choosing a fix, [3] splicing the text;
                         // [4..10] inside [2]: fix_k_case_sub, fix_insert, fix_del, fixdiploid, fix_same_base_del,
                         // fix_same_base_insertion, base_extension
    // filled before stitching (seg_summary_kernel, or the host on the slow path)
    int64_t own_lo, own_hi;   // local range of the polished text this segment contributes
    int64_t own_hi0;          // own_hi before this segment's own change of length is added (set when the segment is built)
    int64_t out_off;          // where it goes in the chunk's new text
};

// what the host needs of a pass's walks, per chunk record (seg_summary_kernel): everything else stays on the device
struct ChunkSummary {
    int64_t newlen;           // length of the chunk's stitched text
    int64_t wrong;            // bad k-mers counted (src/jasper.py:207)
    uint64_t lookups;
    uint32_t nrec, naux;      // fix records / aux bytes of the chunk's segments
    int32_t bad_seg;          // index of the first segment whose status is not PS_OK, or -1
    int32_t spec_fail;        // some segment's speculation failed
};

constexpr long long ARRIVE_PENDING = (long long)0x8080808080808080ull;   // what hipMemset(0x80) leaves
constexpr long long ARRIVE_FAIL = -2, ARRIVE_END = -1;
// what a segment puts into its slot when it STARTS (round 4): CLEAN = chained and nothing can happen in its range, its arrival follows
// from any earlier segment's (successors look past it without waiting for it); WALKING = it walks, its arrival comes at its end
constexpr long long ARRIVE_CLEAN = -3, ARRIVE_WALKING = -4;

struct PolishParams {
    int k;
    int step;       // max(2, round(k/8))            src/jasper.py:20
    uint32_t solid; // solid_thre                    src/jasper.py:23
    int passes;     // P fixing passes; pass P is the QV-only pass (src/jasper.py:25,37-38)
    int fix;
};

// per-chunk arguments of the dense scan / classify / sync-point kernels (one launch covers all chunks of a batch)
struct ScanChunk {
    const uint8_t *text;
    int64_t len;
    uint8_t *cls;
    int64_t *cand;             // sync-point candidates of the WHOLE batch: (chunk << 40) | position, unordered
    unsigned int *cand_count;  // one counter for the batch
    unsigned int cand_cap;
    int32_t want_sync;
    const uint8_t *flags;   // one byte per 64 text positions: the previous pass changed text there (rescan only)
    int64_t *clean_cand;    // per CLEAN_CELL positions: a position whose windows [p-4k, p+k) are all CLEAN, or -1
    uint32_t n_cells;
};
constexpr int64_t CLEAN_CELL = 65536;
// pass 0: dense scan (counts stay in LDS) + classes + sync-point candidates of every chunk
void launch_scan_batch(const TableDev &T, const ScanChunk *d_chunks, int n_chunks, int k, uint32_t solid, hipStream_t stream);
// later passes: classes were carried over by the stitch; recompute the 64-window tiles next to changed text, then candidates
void launch_rescan_batch(const TableDev &T, const ScanChunk *d_chunks, int n_chunks, int k, uint32_t solid, hipStream_t stream);

void launch_copy_chunks(uint8_t *const *d_chunk_ptr, uint8_t *d_pack, const int64_t *d_offs, int n, bool to_chunks, hipStream_t stream);
void launch_seg_init(SegDev *d_segs, int n_segs, const uint8_t *const *d_chunk_text, hipStream_t stream);
// d_ticket: one device word (zeroed by the call); d_segs[i].arrive must point into an array preset to ARRIVE_PENDING
void launch_seg_walk(const TableDev &T, SegDev *d_segs, int n_segs, PolishParams pp, int pass, ScratchPool pool, unsigned int *d_ticket,
                     hipStream_t stream);
// d_cls_out / d_flags may be null (last pass): then only the text is stitched
void launch_seg_stitch(const SegDev *d_segs, int n_segs, uint8_t *const *d_chunk_out, uint8_t *const *d_cls_out, uint8_t *const *d_flags,
                       hipStream_t stream);
// After the walks: per chunk (its segments are consecutive, in text order) the stitched coordinates -- own_hi / out_off of every
// segment written back into d_segs -- the coordinate bases the gather needs (idx_base, seq_base, rec_off, aux_off: n_segs each)
// and one ChunkSummary per chunk.  first_seg[c] .. first_seg[c+1] = the chunk's segments.
void launch_seg_summary(SegDev *d_segs, int n_segs, const int32_t *d_first_seg, int n_chunks, int64_t *idx_base, uint32_t *seq_base, uint32_t *rec_off,
                        uint32_t *aux_off, ChunkSummary *d_out, hipStream_t stream);
void launch_seg_gather(const SegDev *d_segs, int n_segs, const int64_t *idx_base, const uint32_t *seq_base, const uint32_t *rec_off,
                       const uint32_t *aux_off, FixRec *out_recs, uint8_t *out_aux, hipStream_t stream);

}  // namespace jk
