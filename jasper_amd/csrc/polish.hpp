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
    uint32_t aux_off;  // 'x': offset of  patch ++ original segment  in the chunk's aux bytes
    uint32_t aux_len;  // 'x': length of the patch
};
static_assert(sizeof(FixRec) == 32, "FixRec layout");

enum PolishStatus {
    PS_OK = 0,
    PS_GAP_EXHAUSTED = 1,   // chunk grew beyond its slack
    PS_REC_OVERFLOW = 2,    // more fix records than reserved
    PS_AUX_OVERFLOW = 3,
    PS_BFS_ARENA = 4,       // path-extension search ran out of scratch
    PS_REF_INDEXERROR = 5,  // the reference raises IndexError here (src/jasper.py:221) and exits 1
    PS_STRING_TOO_LONG = 6
};

// per chunk state in HBM
struct ChunkDev {
    uint8_t *buf;        // gap buffer: logical text = buf[0,gs) ++ buf[gs+glen, cap)
    int64_t cap;
    int64_t len;         // logical length
    int64_t gs;          // gap start
    int64_t glen;        // gap length
    FixRec *recs;        // this chunk's records (all passes), capacity rec_cap
    uint32_t rec_cap;
    uint32_t nrec;
    uint8_t *aux;        // bytes referenced by 'x' records
    uint32_t aux_cap;
    uint32_t naux;
    // scratch for the path-extension search (src/jasper.py:527-583)
    uint32_t *nodes;     // trie arena: parent << 2 | base
    uint32_t node_cap;
    uint8_t *front;      // frontier entries
    uint32_t front_cap;  // entries
    uint8_t *patch;      // reconstructed patch
    uint32_t patch_cap;
    int32_t status;
    int64_t wrong[2];    // bad k-mers counted in pass 0 and in the final (QV) pass
    int64_t total[2];    // total k-mers (len-k+1) in pass 0 and in the final pass
    uint64_t lookups;    // table probes issued (informational)
};

struct PolishParams {
    int k;
    int step;       // max(2, round(k/8))            src/jasper.py:20
    uint32_t solid; // solid_thre                    src/jasper.py:23
    int passes;     // P fixing passes; pass P is the QV-only pass (src/jasper.py:25,37-38)
    int fix;
};

void launch_polish_pass(const TableDev &T, ChunkDev *d_chunks, int n_chunks, PolishParams pp, int pass, hipStream_t stream);
void launch_pack(ChunkDev *d_chunks, int n_chunks, uint8_t *d_out, const int64_t *d_out_off, hipStream_t stream);

}  // namespace jk
