// polish_host.hip -- host orchestration of one batch of chunk records through P fixing passes + the QV pass
// (src/jasper.py:25-26 `for ite in range(num_iter+1): iteration(...)`).
//
// Per pass:  dense scan + classes (all chunks)  ->  sync points  ->  segments  ->  concurrent segment walks
//            ->  [redo chunks whose speculation failed as one segment]  ->  gather fix records, stitch the new text.
#include "polish_host.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace jk {

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                     \
            return -1;                                                                \
        }                                                                             \
    } while (0)

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

namespace {
struct DevBuf {  // a slot of the table's grow-only workspace (not owned) or a temporary (owned)
    void *p = nullptr;
    bool owned = false;
    ~DevBuf() { if (p && owned) (void)hipFree(p); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
}  // namespace

// one lane: these chunk records through all passes, on stream `st`, in the lane's own workspace slots and pinned buffers
static int run_polish_lane(Table &T, const int lane, hipStream_t st, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre, int passes, int fix,
                           PolishOut &R, std::string &err, bool device_in, bool keep_on_device, int roomy) {
    // `roomy`: every slack / bound below is a guess about how much a pass can add; the default guesses are generous for
    // real polishing (edits are ~0.1 % of the text).  If one is exceeded the call fails cleanly with -2 and the caller
    // repeats it with roomy = 1: 8x the slack (nothing of a failed call is kept, so the repeat is exact).
    const int64_t RM = roomy ? 8 : 1;
    const bool tight = !roomy && getenv("JASPER_POLISH_TIGHT") != nullptr;   // tests: make the first attempt run out of room
    const int k = T.k;
    const int64_t W = 4ll * k;        // clean window required left of a sync point
    const int64_t M = 3ll * k;        // text kept right of the next sync point
    const int64_t TMIN = 1024;        // minimum distance between sync points
    const int64_t CMIN = 32768;       // minimum distance between a clean-zone boundary and its neighbours
    HIPCHK(hipSetDevice(T.device));
    const int ws_base = lane ? Table::WS_LANE0 + (lane - 1) * Table::WS_POLISH_MAX : 0, pin_base = lane * Table::PIN_PER_LANE;
    R.seqs.assign(n_chunks, std::string());
    R.d_seqs.clear();
    R.d_lens.clear();
    R.aux.assign(n_chunks, std::string());
    R.recs.clear();
    R.qv[0] = R.qv[1] = R.qv[2] = R.qv[3] = 0;
    R.qv_chunk.assign((size_t)4 * n_chunks, 0);
    R.lookups = 0;
    R.seconds = 0;
    R.n_segments = 0;
    R.n_respeculated = 0;
    if (n_chunks == 0) return 0;

    const bool dbg = getenv("JASPER_POLISH_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // ---------------- layout ----------------
    std::vector<int64_t> len(lens, lens + n_chunks), cap(n_chunks);
    std::vector<size_t> off_text(n_chunks), off_pos(n_chunks), off_cand(n_chunks), off_flag(n_chunks), off_cell(n_chunks);
    std::vector<uint32_t> n_cells(n_chunks);
    size_t cell_items = 0;
    std::vector<uint32_t> cand_cap(n_chunks);
    size_t text_bytes = 0, pos_items = 0, cand_items = 0, flag_items = 0, seg_text_bound = 0, seg_rec_bound = 0, seg_aux_bound = 0;
    int64_t max_segs = 0;
    for (int c = 0; c < n_chunks; ++c) {
        cap[c] = len[c] + RM * std::max<int64_t>(4096, len[c] / 8) + 64;
        off_text[c] = text_bytes;  text_bytes += al256((size_t)cap[c]);
        off_pos[c] = pos_items;    pos_items += al256((size_t)cap[c]);
        cand_cap[c] = (uint32_t)std::min<int64_t>(1 << 28, cap[c] / (4 * k) + 16);
        off_cand[c] = cand_items;  cand_items += cand_cap[c];
        off_flag[c] = flag_items;  flag_items += al256((size_t)(cap[c] >> 6) + 2);
        n_cells[c] = (uint32_t)(cap[c] / CLEAN_CELL + 1);
        off_cell[c] = cell_items;  cell_items += n_cells[c];
        const int64_t ms = cap[c] / TMIN + 2;
        max_segs += ms;
        const int64_t tb = cap[c] + ms * (W + M + 64);
        seg_text_bound += (size_t)(tb + RM * (tb / 8 + ms * 1280));
        seg_rec_bound += (size_t)(RM * (2 * tb / k + 16 * ms));
        seg_aux_bound += (size_t)(RM * (2 * tb + 1024 * ms));
    }
    DevBuf b_textA, b_textB, b_pack, b_cls, b_clsB, b_flags, b_segedit, b_cells, b_arrive, b_ptrA, b_segs, b_segtext, b_segrec, b_segaux, b_pool, b_locks, b_scan, b_ticket;
    int ws_next = 0;
    auto dmalloc = [&](DevBuf &b, size_t bytes) -> bool {      // persistent: slot of the table's workspace
        if (ws_next >= Table::WS_POLISH_MAX) { err = "polish: workspace slots exhausted"; return false; }
        b.p = T.workspace(ws_base + ws_next++, bytes, err);
        b.owned = false;
        return b.p != nullptr;
    };
    if (!dmalloc(b_textA, text_bytes) || !dmalloc(b_textB, text_bytes) || 
        !dmalloc(b_cls, pos_items) || !dmalloc(b_clsB, pos_items) || !dmalloc(b_flags, flag_items) ||
        !dmalloc(b_segedit, seg_rec_bound * sizeof(EditRec)) || !dmalloc(b_cells, (2 + cell_items + cand_items) * 8) ||      // [count | clean-zone cells | sync-point candidates]: one copy brings the first two and the candidates' head
        !dmalloc(b_arrive, ((size_t)max_segs + 4) * 8) ||
        !dmalloc(b_ptrA, (size_t)6 * n_chunks * sizeof(void *)) ||
        !dmalloc(b_segs, (size_t)max_segs * sizeof(SegDev)) || !dmalloc(b_segtext, seg_text_bound) ||
        !dmalloc(b_segrec, seg_rec_bound * sizeof(FixRec)) || !dmalloc(b_segaux, seg_aux_bound))
        return -2;
    ScratchPool pool;
    pool.nslots = 256;
    pool.node_cap = (uint32_t)(RM << 18);
    pool.front_cap = 20480;            // (the search gives up beyond 5000 live paths, src/jasper.py:543-546; each adds <= 3 siblings per level)
    pool.patch_cap = (uint32_t)(RM << 16);
    if (roomy) pool.nslots = 64;
    if (const char *e = getenv("JASPER_POLISH_TEST_NSLOTS")) pool.nslots = (uint32_t)std::max(1, std::min(atoi(e), 256));     // (tests: slots handed from wave to wave all the time)
    pool.off_front = al256((size_t)pool.node_cap * 4);
    pool.off_patch = pool.off_front + al256((size_t)pool.front_cap * 80);
    pool.stride = pool.off_patch + al256(pool.patch_cap);
    if (!dmalloc(b_pool, pool.stride * pool.nslots) || !dmalloc(b_locks, pool.nslots * 4) || !dmalloc(b_scan, sizeof(ScanChunk) * n_chunks) ||
        !dmalloc(b_ticket, 256)) return -2;
    // the bookkeeping after a pass's walks happens on the device (seg_summary_kernel); the host gets one ChunkSummary per chunk
    DevBuf b_first, b_sum, b_idx, b_seq, b_ro, b_ao;
    if (!dmalloc(b_first, ((size_t)n_chunks + 1) * 4) || !dmalloc(b_sum, (size_t)n_chunks * sizeof(ChunkSummary)) || !dmalloc(b_idx, (size_t)max_segs * 8) ||
        !dmalloc(b_seq, (size_t)max_segs * 4) || !dmalloc(b_ro, (size_t)max_segs * 4) || !dmalloc(b_ao, (size_t)max_segs * 4)) return -2;
    // pinned host memory for everything that crosses PCIe inside the pass loop (kept with the table)
    const size_t PIN_RECS = 1u << 18, PIN_AUX = 8u << 20;            // records / aux bytes of ALL passes that travel without a host wait
    SegDev *segs_p = reinterpret_cast<SegDev *>(T.pinned(pin_base + 0, (size_t)max_segs * sizeof(SegDev), err));
    int32_t *first_p = reinterpret_cast<int32_t *>(T.pinned(pin_base + 1, ((size_t)n_chunks + 1) * 4, err));
    ChunkSummary *sum_p = reinterpret_cast<ChunkSummary *>(T.pinned(pin_base + 2, (size_t)n_chunks * sizeof(ChunkSummary), err));
    int64_t *cand_p = reinterpret_cast<int64_t *>(T.pinned(pin_base + 3, (std::min<size_t>(cand_items, 32768) + cell_items + 8) * 8, err));   // [count (2 words) | cells | first candidates]
    FixRec *recs_p = reinterpret_cast<FixRec *>(T.pinned(pin_base + 4, PIN_RECS * sizeof(FixRec), err));
    uint8_t *aux_p = reinterpret_cast<uint8_t *>(T.pinned(pin_base + 5, PIN_AUX, err));
    ScanChunk *sc_p = reinterpret_cast<ScanChunk *>(T.pinned(pin_base + 6, sizeof(ScanChunk) * (size_t)n_chunks, err));
    if (!segs_p || !first_p || !sum_p || !cand_p || !recs_p || !aux_p || !sc_p) return -2;
    // very many small records in host memory: their texts cross PCIe packed, one transfer each way (a transfer per record costs
    // ~10 us: 100 000 contigs of a fragmented assembly took 2.8 s for 70 ms of kernels)
    int64_t total_len = 0;
    for (int c = 0; c < n_chunks; ++c) total_len += len[c];
    const bool packed_io = !device_in && n_chunks > 256 && total_len / n_chunks < 65536 && !getenv("JASPER_POLISH_NO_PACKED_IO");
    DevBuf b_iooffs;
    std::vector<int64_t> io_offs;
    uint8_t *io_stage = nullptr;
    if (!packed_io) ws_next += 2;                    // (the two slots stay theirs: what follows keeps its slots from call to call)
    if (packed_io) {
        if (!dmalloc(b_iooffs, ((size_t)n_chunks + 1) * 8) || !dmalloc(b_pack, text_bytes)) return -2;
        io_offs.resize((size_t)n_chunks + 1);
        // (the texts may grow on the way: the way back is sized like the arenas)
        io_stage = reinterpret_cast<uint8_t *>(T.pinned(pin_base + 7, std::max<size_t>((size_t)total_len, 1) + (size_t)(RM * std::max<int64_t>(4096, total_len / 8)) + 64, err));
        if (!io_stage) return -2;
    }
    size_t pin_recs_used = 0, pin_aux_used = 0;
    struct PinnedPass { size_t rec_at, nrec, aux_at, naux, r0, pass_i; };      // what a pass left in the pinned record buffers: copied out after the last pass
    std::vector<PinnedPass> pinned_passes;
    pool.base = b_pool.as<uint8_t>();
    pool.locks = b_locks.as<unsigned int>();
    HIPCHK(hipMemsetAsync(pool.locks, 0, pool.nslots * 4, st));

    uint8_t *clsIn = b_cls.as<uint8_t>(), *clsOut = b_clsB.as<uint8_t>();
    // Pointer tables (one upload): [0] the text the next pass READS per chunk, [1] where it WRITES, [2] the third arena's
    // turn, [3] classes A, [4] classes B, [5] changed-text flags.  Chunk records that are already in HBM are read where
    // they lie by pass 0 (U -> B -> A -> B ...); host records are copied into arena A first (A -> B -> A ...).
    std::vector<uint8_t *> tables((size_t)6 * n_chunks);
    uint8_t **hIn = tables.data(), **hOut = hIn + n_chunks, **hSpare = hOut + n_chunks, **hCA = hSpare + n_chunks, **hCB = hCA + n_chunks,
            **hF = hCB + n_chunks;
    for (int c = 0; c < n_chunks; ++c) {
        uint8_t *a = b_textA.as<uint8_t>() + off_text[c];
        hIn[c] = device_in ? reinterpret_cast<uint8_t *>(const_cast<char *>(seqs[c])) : a;
        hOut[c] = b_textB.as<uint8_t>() + off_text[c];
        hSpare[c] = a;
        hCA[c] = b_cls.as<uint8_t>() + off_pos[c];
        hCB[c] = b_clsB.as<uint8_t>() + off_pos[c];
        hF[c] = b_flags.as<uint8_t>() + off_flag[c];
        if (len[c] && !device_in && !packed_io) HIPCHK(hipMemcpyAsync(a, seqs[c], (size_t)len[c], hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(b_ptrA.p, tables.data(), tables.size() * sizeof(void *), hipMemcpyHostToDevice, st));
    uint8_t **dIn = b_ptrA.as<uint8_t *>(), **dOut = dIn + n_chunks, **dSpare = dOut + n_chunks;
    uint8_t **ptrClsIn = dSpare + n_chunks, **ptrClsOut = ptrClsIn + n_chunks, **ptrFlags = ptrClsOut + n_chunks;
    if (packed_io) {
        // all the texts in one transfer: packed on the host, spread to their arena places by a kernel
        int64_t at = 0;
        for (int c = 0; c < n_chunks; ++c) { io_offs[c] = at; if (len[c]) memcpy(io_stage + at, seqs[c], (size_t)len[c]); at += len[c]; }
        io_offs[n_chunks] = at;
        HIPCHK(hipMemcpyAsync(b_iooffs.p, io_offs.data(), io_offs.size() * 8, hipMemcpyHostToDevice, st));
        if (at) HIPCHK(hipMemcpyAsync(b_pack.p, io_stage, (size_t)at, hipMemcpyHostToDevice, st));
        launch_copy_chunks(dIn, b_pack.as<uint8_t>(), b_iooffs.as<int64_t>(), n_chunks, true, st);
        HIPCHK(hipGetLastError());
        HIPCHK(jk_stream_wait(st));                    // (io_offs and the stage are reused for the way back)
    }

    if (dbg) { (void)jk_stream_wait(st); fprintf(stderr, "[polish] setup + H2D: %.2f ms\n", now() - t_begin); }
    const double t_loop = now();
    PolishParams pp;
    pp.k = k;
    pp.step = std::max(2, (int)std::nearbyint((double)k / 8.0));   // src/jasper.py:20 (python round = half-to-even)
    pp.solid = (uint32_t)solid_thre;
    pp.passes = passes;
    pp.fix = fix ? 1 : 0;

    hipEvent_t ev0, ev1;
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    HIPCHK(hipEventRecord(ev0, st));

    std::vector<SegDev> segs;
    std::vector<ScanChunk> sc(n_chunks);            // lives across passes: the async H2D copy reads it after the call returns
    int64_t *const d_count = b_cells.as<int64_t>(), *const d_cells = d_count + 2, *const d_cands = d_cells + cell_items;
    std::vector<int64_t> all_cands(cand_items);     // every chunk's sync-point candidates, fetched with one copy per pass
    std::vector<int64_t> all_cells(cell_items);     // clean-zone boundary candidates, one per CLEAN_CELL positions
    std::vector<std::vector<int64_t>> chunk_cands(n_chunks);
    std::vector<int64_t> cands;
    std::vector<uint32_t> aux_total(n_chunks, 0);   // aux bytes gathered so far per chunk (all passes)
    std::vector<std::vector<uint8_t>> aux_pass;      // raw aux bytes per pass, regrouped per chunk at the end
    std::vector<size_t> rec_pass_begin;
    int rc = 0;

    for (int pass = 0; pass <= passes && rc == 0; ++pass) {                                   // src/jasper.py:25
        // ---- 1. position classes + sync-point candidates: a dense scan in pass 0; afterwards the stitch has carried the
        //         classes of untouched windows over and only the tiles next to changed text are recomputed
        HIPCHK(hipMemsetAsync(d_count, 0, 4, st));
        {
            for (int c = 0; c < n_chunks; ++c) {
                ScanChunk &S = sc[c];
                S.text = hIn[c];
                S.len = len[c];
                S.cls = clsIn + off_pos[c];
                S.flags = b_flags.as<uint8_t>() + off_flag[c];
                S.cand = d_cands;
                S.cand_count = reinterpret_cast<unsigned int *>(d_count);
                S.cand_cap = (unsigned int)std::min<size_t>(cand_items, 0x7fffffffu);
                S.want_sync = (len[c] - k + 1) > 2 * TMIN;
                S.clean_cand = d_cells + off_cell[c];
                S.n_cells = (uint32_t)std::min<int64_t>(n_cells[c], std::max<int64_t>(0, len[c] - k + 1) / CLEAN_CELL + 1);
            }
            memcpy(sc_p, sc.data(), sizeof(ScanChunk) * n_chunks);
            HIPCHK(hipMemcpyAsync(b_scan.p, sc_p, sizeof(ScanChunk) * n_chunks, hipMemcpyHostToDevice, st));
            if (pass == 0) launch_scan_batch(T.d, b_scan.as<ScanChunk>(), n_chunks, k, pp.solid, st);
            else launch_rescan_batch(T.d, b_scan.as<ScanChunk>(), n_chunks, k, pp.solid, st);
        }
        HIPCHK(hipGetLastError());
        // the candidate list is fetched optimistically: its count and its first CAND_PRE entries in one round trip
        const size_t CAND_PRE = std::min<size_t>(cand_items, 32768);
        unsigned int n_cand = 0;
        {   // (pinned: [0] the count, [1 ..] the first CAND_PRE candidates, then the clean-zone cells)
            HIPCHK(hipMemcpyAsync(cand_p, d_count, (2 + cell_items + CAND_PRE) * 8, hipMemcpyDeviceToHost, st));       // ONE copy command
            HIPCHK(jk_stream_wait(st));
            n_cand = *reinterpret_cast<const unsigned int *>(cand_p);
            if (cell_items) memcpy(all_cells.data(), cand_p + 2, cell_items * 8);
            if (CAND_PRE) memcpy(all_cands.data(), cand_p + 2 + cell_items, std::min<size_t>(n_cand, CAND_PRE) * 8);
        }
        n_cand = (unsigned int)std::min<size_t>(n_cand, cand_items);
        if (n_cand > CAND_PRE) {
            HIPCHK(hipMemcpyAsync(all_cands.data() + CAND_PRE, d_cands + CAND_PRE, (n_cand - CAND_PRE) * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(jk_stream_wait(st));
        }
        for (int c = 0; c < n_chunks; ++c) chunk_cands[c].clear();
        for (unsigned int i = 0; i < n_cand; ++i) {
            const int64_t v = all_cands[i];
            const int64_t c = v >> 40;
            if (c >= 0 && c < n_chunks) chunk_cands[c].push_back(v & ((1ll << 40) - 1));
        }

        // ---- 2. segments
        // segments of `chunks` into out[0 .. n_out) (room for max_segs); first_of (optional): first_of[i] = index of chunks[i]'s first segment
        auto build_segments = [&](const std::vector<int> &chunks, bool speculate, SegDev *out, size_t &n_out, int32_t *first_of, std::string &e2) -> int {
            n_out = 0;
            size_t tpos = 0, rpos = 0, apos = 0, ci = 0;
            for (int c : chunks) {
                if (first_of) first_of[ci++] = (int32_t)n_out;
                std::vector<int64_t> sync;
                if (speculate && !chunk_cands[c].empty()) {
                    cands = chunk_cands[c];
                    std::sort(cands.begin(), cands.end());
                    int64_t last = 0;
                    for (int64_t p : cands)
                        if (p - last >= TMIN && len[c] - p >= TMIN) { sync.push_back(p); last = p; }
                }
                // clean-zone boundaries inside the stretches the sync points leave long: the segment to the right of one
                // is chained to its left neighbour (it starts where that one's walk arrives)
                std::vector<char> chained(sync.size(), 0);
                if (speculate && len[c] - k + 1 > 2 * CMIN) {
                    const uint32_t nc = sc[c].n_cells;
                    std::vector<int64_t> merged;
                    std::vector<char> mtype;
                    size_t si = 0;
                    int64_t last = 0;
                    for (uint32_t g = 0; g <= nc; ++g) {
                        const int64_t p = g < nc ? all_cells[off_cell[c] + g] : INT64_MAX;
                        while (si < sync.size() && sync[si] <= p) { merged.push_back(sync[si]); mtype.push_back(0); last = sync[si]; ++si; }
                        if (g == nc || p < 0) continue;
                        const int64_t next = si < sync.size() ? sync[si] : len[c];
                        if (p - last >= CMIN && next - p >= CMIN) { merged.push_back(p); mtype.push_back(1); last = p; }
                    }
                    sync.swap(merged);
                    chained.swap(mtype);
                }
                const int m = (int)sync.size();
                if ((int64_t)n_out + m + 1 > max_segs) { e2 = "polish: internal segment arena bound exceeded"; return -2; }
                for (int j = 0; j <= m; ++j) {
                    SegDev &S = out[n_out++];
                    memset(&S, 0, sizeof S);
                    S.chunk = (uint32_t)c;
                    S.first = (j == 0);
                    S.last = (j == m);
                    S.seg_lo = j == 0 ? 0 : sync[j - 1] - W;
                    S.start_i = j == 0 ? 0 : W;
                    S.chain_in = j > 0 && chained[j - 1];
                    S.arrive = b_arrive.as<long long>() + n_out;             // (= 1 + its index; slot 0 is never read: a first segment is not chained)
                    S.stop_orig = j == m ? INT64_MAX : sync[j];
                    const int64_t seg_hi = j == m ? len[c] : std::min<int64_t>(len[c], sync[j] + M);
                    S.len0 = seg_hi - S.seg_lo;
                    S.len = S.len0;
                    S.cap = S.len0 + (tight ? 2 : RM * std::max<int64_t>(1024, S.len0 / 8));
                    S.gs = 0;
                    S.glen = S.cap - S.len0;
                    S.cls = (len[c] - k + 1 > 0) ? clsIn + off_pos[c] : nullptr;
                    S.cls_n = std::max<int64_t>(0, len[c] - k + 1);
                    S.rec_cap = (uint32_t)std::min<int64_t>(0x7fffffff, RM * (2 * S.len0 / k + 16));
                    S.aux_cap = (uint32_t)std::min<int64_t>(0x7fffffff, RM * (2 * S.len0 + 1024));
                    S.buf = b_segtext.as<uint8_t>() + tpos;            tpos += al256((size_t)S.cap);
                    S.recs = b_segrec.as<FixRec>() + rpos;
                    S.edits = b_segedit.as<EditRec>() + rpos;          // one edit list entry per possible record
                    S.edit_cap = S.rec_cap;                            rpos += S.rec_cap;
                    S.aux = b_segaux.as<uint8_t>() + apos;             apos += al256(S.aux_cap);
                    // owned part of the text: chunk coordinates [B_j, B_{j+1}), B = sync - 2k
                    S.own_lo = j == 0 ? 0 : 2ll * k;                   // local (no edit can precede it)
                    S.own_hi = j == m ? -1 : (sync[j] - 2ll * k) - S.seg_lo;   // local, before adding this segment's delta
                    S.own_hi0 = S.own_hi;
                }
            }
            if (first_of) first_of[ci] = (int32_t)n_out;
            if (tpos > seg_text_bound || rpos > seg_rec_bound || apos > seg_aux_bound || (int64_t)n_out > max_segs) {
                e2 = "polish: internal segment arena bound exceeded"; return -2;
            }
            return 0;
        };
        // upload, walk; then either the whole table comes back (sv is overwritten) or -- `summary` -- the bookkeeping is done on
        // the device and one ChunkSummary per chunk comes back (sum_p)
        auto run_segments = [&](SegDev *sv, size_t nsv, bool summary, std::string &e2) -> int {
            if (!nsv) return 0;
            const bool fine = dbg && getenv("JASPER_POLISH_DEBUG") && atoi(getenv("JASPER_POLISH_DEBUG")) >= 2;   // (waits between the steps: their times, not the pass's)
            double tf[5] = {fine ? now() : 0, 0, 0, 0, 0};
            if (hipMemcpyAsync(b_segs.p, sv, nsv * sizeof(SegDev), hipMemcpyHostToDevice, st) != hipSuccess) { e2 = "polish: H2D segs"; return -1; }
            if (summary && hipMemcpyAsync(b_first.p, first_p, ((size_t)n_chunks + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess) { e2 = "polish: H2D first"; return -1; }
            if (fine) { (void)jk_stream_wait(st); tf[1] = now(); }
            launch_seg_init(b_segs.as<SegDev>(), (int)nsv, (const uint8_t *const *)dIn, st);
            if (hipMemsetAsync(b_arrive.p, 0x80, (nsv + 2) * 8, st) != hipSuccess) { e2 = "polish: memset"; return -1; }   // ARRIVE_PENDING
            if (fine) { (void)jk_stream_wait(st); tf[2] = now(); }
            launch_seg_walk(T.d, b_segs.as<SegDev>(), (int)nsv, pp, pass, pool, b_ticket.as<unsigned int>(), st);
            if (hipGetLastError() != hipSuccess) { e2 = "polish: kernel launch failed"; return -1; }
            if (fine) { (void)jk_stream_wait(st); tf[3] = now(); }
            if (summary) {
                launch_seg_summary(b_segs.as<SegDev>(), (int)nsv, b_first.as<int32_t>(), n_chunks, b_idx.as<int64_t>(), b_seq.as<uint32_t>(), b_ro.as<uint32_t>(),
                                   b_ao.as<uint32_t>(), b_sum.as<ChunkSummary>(), st);
                if (hipMemcpyAsync(sum_p, b_sum.p, (size_t)n_chunks * sizeof(ChunkSummary), hipMemcpyDeviceToHost, st) != hipSuccess) { e2 = "polish: D2H summary"; return -1; }
            } else if (hipMemcpyAsync(sv, b_segs.p, nsv * sizeof(SegDev), hipMemcpyDeviceToHost, st) != hipSuccess) { e2 = "polish: D2H segs"; return -1; }
            if (jk_stream_wait(st) != hipSuccess) { e2 = "polish: kernel execution failed"; return -1; }
            if (fine) {
                tf[4] = now();
                fprintf(stderr, "[polish]     %zu segments x %zu B: H2D %.3f ms, init + memset %.3f ms, walk %.3f ms, %s %.3f ms\n", nsv, sizeof(SegDev), tf[1] - tf[0], tf[2] - tf[1],
                        tf[3] - tf[2], summary ? "summary + D2H" : "D2H", tf[4] - tf[3]);
            }
            return 0;
        };
        std::vector<int> all(n_chunks);
        for (int c = 0; c < n_chunks; ++c) all[c] = c;
        const double tb0 = dbg ? now() : 0;
        size_t ns = 0;
        if ((rc = build_segments(all, true, segs_p, ns, first_p, err))) break;
        const double tb1 = dbg ? now() : 0;
        if ((rc = run_segments(segs_p, ns, true, err))) break;
        if (dbg) fprintf(stderr, "[polish]   host: build %zu segments %.3f ms, upload + walk + summary %.3f ms\n", ns, tb1 - tb0, now() - tb1);

        // ---- 3. + 4. the bookkeeping per chunk came back as summaries.  Anything out of the ordinary -- a segment that ran out of
        //         room, the reference's own IndexError, a speculation that failed -- takes the host's own bookkeeping (below),
        //         on the whole segment table; so does the debug output, which wants every segment's counters.
        bool ordinary = !getenv("JASPER_POLISH_DEBUG") && !getenv("JASPER_POLISH_HOST_BOOKKEEPING");      // (the second: tests take the rare path every time)
        for (int c = 0; c < n_chunks && ordinary; ++c) ordinary = sum_p[c].bad_seg < 0 && !sum_p[c].spec_fail;
        std::vector<int64_t> idx_base;
        std::vector<uint32_t> seq_base, rec_off, aux_off;
        std::vector<int64_t> newlen(n_chunks, 0), shift(n_chunks, 0);
        std::vector<uint32_t> seqc(n_chunks, 0);
        size_t nrec_pass = 0, naux_pass = 0;
        if (ordinary) {
            R.n_segments += ns;
            for (int c = 0; c < n_chunks; ++c) {
                const ChunkSummary &S = sum_p[c];
                newlen[c] = S.newlen;
                nrec_pass += S.nrec;
                naux_pass += S.naux;
                R.lookups += S.lookups;
                if (pass == 0) { R.qv[0] += S.wrong; R.qv_chunk[4 * (size_t)c + 0] += S.wrong; }
                if (pass == passes) { R.qv[2] += S.wrong; R.qv_chunk[4 * (size_t)c + 2] += S.wrong; }
            }
        } else {
        segs.resize(ns);
        HIPCHK(hipMemcpyAsync(segs.data(), b_segs.p, ns * sizeof(SegDev), hipMemcpyDeviceToHost, st));
        HIPCHK(jk_stream_wait(st));
        R.n_segments += segs.size();

        // ---- 3. chunks whose speculation failed are redone as a single segment (= the plain sequential walk)
        std::vector<char> failed(n_chunks, 0);
        bool any_failed = false;
        for (const SegDev &S : segs)
            if (S.spec_fail && S.status == PS_OK) { failed[S.chunk] = 1; any_failed = true; }
        if (any_failed) {
            std::vector<int> redo;
            for (int c = 0; c < n_chunks; ++c) if (failed[c]) redo.push_back(c);
            R.n_respeculated += redo.size();
            // results of the good chunks must survive: stitch / gather them first, then reuse the arenas for the redo
            std::vector<SegDev> good, bad;
            for (const SegDev &S : segs) if (!failed[S.chunk]) good.push_back(S);
            // (the redo gets fresh arena space; the good segments' buffers stay untouched because run_segments
            // only writes through the SegDev table it uploads -- so place the redo segments BEHIND the good ones)
            size_t tpos = 0, rpos = 0, apos = 0;
            for (const SegDev &S : segs) { tpos += al256((size_t)S.cap); rpos += S.rec_cap; apos += al256(S.aux_cap); }
            std::vector<SegDev> redo_segs((size_t)max_segs);
            size_t nredo = 0;
            if ((rc = build_segments(redo, false, redo_segs.data(), nredo, nullptr, err))) break;
            redo_segs.resize(nredo);
            size_t t2 = tpos, r2 = rpos, a2 = apos;
            for (SegDev &S : redo_segs) {
                S.buf = b_segtext.as<uint8_t>() + t2;  t2 += al256((size_t)S.cap);
                S.recs = b_segrec.as<FixRec>() + r2;
                S.edits = b_segedit.as<EditRec>() + r2; r2 += S.rec_cap;
                S.aux = b_segaux.as<uint8_t>() + a2;   a2 += al256(S.aux_cap);
            }
            if (t2 > seg_text_bound || r2 > seg_rec_bound || a2 > seg_aux_bound) {
                // not enough spare room: fall back to redoing EVERYTHING unsegmented (always fits)
                segs.assign((size_t)max_segs, SegDev{});
                size_t nall = 0;
                if ((rc = build_segments(all, false, segs.data(), nall, nullptr, err))) break;
                segs.resize(nall);
                if ((rc = run_segments(segs.data(), segs.size(), false, err))) break;
            } else {
                if ((rc = run_segments(redo_segs.data(), redo_segs.size(), false, err))) break;
                segs = good;
                segs.insert(segs.end(), redo_segs.begin(), redo_segs.end());
                std::stable_sort(segs.begin(), segs.end(), [](const SegDev &a, const SegDev &b) {
                    return a.chunk != b.chunk ? a.chunk < b.chunk : a.seg_lo < b.seg_lo;
                });
            }
        }
        if (getenv("JASPER_POLISH_DEBUG")) {
            uint64_t mx = 0, sum = 0, mxl = 0; int64_t mxlen = 0; size_t nrecs = 0, mxrec = 0;
            uint64_t tks[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mxtk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (const SegDev &S : segs) {
                sum += S.ticks; nrecs += S.nrec;
                for (int q = 0; q < 12; ++q) tks[q] += S.tk[q];
                if (S.ticks > mx) { mx = S.ticks; mxl = S.lookups; mxlen = S.len0; mxrec = S.nrec; for (int q = 0; q < 12; ++q) mxtk[q] = S.tk[q]; }
            }
            fprintf(stderr, "[polish] pass %d: %zu segments, %zu records, walk ticks(10ns): mean %.0f max %llu (that segment: len %lld, %llu lookups, %zu records)\n",
                    pass, segs.size(), nrecs, segs.empty() ? 0.0 : (double)sum / segs.size(), (unsigned long long)mx, (long long)mxlen,
                    (unsigned long long)mxl, mxrec);
            fprintf(stderr, "[polish]   of all walk ticks: skip_good %.1f%%, find run %.1f%%, choose fix %.1f%% (of which splice %.1f%%)\n",
                    100.0 * tks[0] / (sum + 1), 100.0 * tks[1] / (sum + 1), 100.0 * tks[2] / (sum + 1), 100.0 * tks[3] / (sum + 1));
            fprintf(stderr, "[polish]   inside choose fix: k_case_sub %.1f%%, insert %.1f%%, del %.1f%%, diploid %.1f%%, same_base_del %.1f%%, same_base_ins %.1f%%, path search %.1f%%\n",
                    100.0 * tks[4] / (sum + 1), 100.0 * tks[5] / (sum + 1), 100.0 * tks[6] / (sum + 1), 100.0 * tks[7] / (sum + 1),
                    100.0 * tks[8] / (sum + 1), 100.0 * tks[9] / (sum + 1), 100.0 * tks[10] / (sum + 1));
            fprintf(stderr, "[polish]   slowest segment (10ns ticks): skip %llu, find run %llu, choose fix %llu [sub %llu ins %llu del %llu diploid %llu sb_del %llu sb_ins %llu path %llu], splice %llu\n",
                    (unsigned long long)mxtk[0], (unsigned long long)mxtk[1], (unsigned long long)mxtk[2], (unsigned long long)mxtk[4], (unsigned long long)mxtk[5],
                    (unsigned long long)mxtk[6], (unsigned long long)mxtk[7], (unsigned long long)mxtk[8], (unsigned long long)mxtk[9], (unsigned long long)mxtk[10],
                    (unsigned long long)mxtk[3]);
            // histogram of segment times in ms buckets
            int hb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (const SegDev &S : segs) { double ms = S.ticks * 1e-5; int b = ms < 0.1 ? 0 : ms < 0.3 ? 1 : ms < 1 ? 2 : ms < 2 ? 3 : ms < 4 ? 4 : ms < 8 ? 5 : ms < 16 ? 6 : 7; hb[b]++; }
            fprintf(stderr, "[polish]   segments by walk time: <0.1ms %d, <0.3 %d, <1 %d, <2 %d, <4 %d, <8 %d, <16 %d, more %d\n", hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6], hb[7]);
        }

        // ---- 4. bookkeeping per chunk: status, counters, coordinates of the stitched text
        ns = segs.size();
        idx_base.assign(ns, 0);
        seq_base.assign(ns, 0); rec_off.assign(ns, 0); aux_off.assign(ns, 0);
        for (size_t s = 0; s < ns && rc == 0; ++s) {
            SegDev &S = segs[s];
            const int c = (int)S.chunk;
            if (S.status != PS_OK) {
                static const char *names[] = {"ok", "text grew beyond its slack", "fix-record buffer overflow", "aux buffer overflow",
                                              "path-extension scratch exhausted", "reference IndexError (src/jasper.py:221)",
                                              "trial string too long"};
                err = std::string("polish: chunk ") + std::to_string(c) + ": " + names[S.status];
                rc = S.status == PS_REF_INDEXERROR ? -4 : -2;
                break;
            }
            const int64_t d = S.len - S.len0;
            S.own_hi = S.last ? S.len : S.own_hi0 + d;        // (from own_hi0: the device's summary pass may have written own_hi already)
            S.out_off = newlen[c];
            newlen[c] += S.own_hi - S.own_lo;
            idx_base[s] = S.seg_lo + shift[c];
            shift[c] += d;
            seq_base[s] = seqc[c];
            seqc[c] += S.nrec;
            rec_off[s] = (uint32_t)nrec_pass;
            aux_off[s] = (uint32_t)naux_pass;
            nrec_pass += S.nrec;
            naux_pass += S.naux;
            R.lookups += S.lookups;
            if (pass == 0) { R.qv[0] += S.wrong; R.qv_chunk[4 * (size_t)c + 0] += S.wrong; }
            if (pass == passes) { R.qv[2] += S.wrong; R.qv_chunk[4 * (size_t)c + 2] += S.wrong; }
        }
        if (rc) break;
        }       // (the host's own bookkeeping)
        if (rc) break;
        for (int c = 0; c < n_chunks; ++c) {
            if (pass == 0) { R.qv[1] += len[c] - k + 1; R.qv_chunk[4 * (size_t)c + 1] += len[c] - k + 1; }   // src/jasper.py:51,107-111
            if (pass == passes) { R.qv[3] += len[c] - k + 1; R.qv_chunk[4 * (size_t)c + 3] += len[c] - k + 1; }
            if (newlen[c] > cap[c]) { err = "polish: chunk grew beyond its slack"; rc = -2; }
        }
        if (rc) break;

        // ---- 5. gather records / aux, stitch the new text
        rec_pass_begin.push_back(R.recs.size());
        aux_pass.emplace_back();
        const int64_t *g_idx = b_idx.as<int64_t>();
        const uint32_t *g_seq = b_seq.as<uint32_t>(), *g_ro = b_ro.as<uint32_t>(), *g_ao = b_ao.as<uint32_t>();
        if (!ordinary) {                            // the host's bookkeeping goes up: the segment table and the four coordinate bases
            HIPCHK(hipMemcpyAsync(b_segs.p, segs.data(), ns * sizeof(SegDev), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(b_idx.p, idx_base.data(), ns * 8, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(b_seq.p, seq_base.data(), ns * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(b_ro.p, rec_off.data(), ns * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(b_ao.p, aux_off.data(), ns * 4, hipMemcpyHostToDevice, st));
        }
        if (nrec_pass) {
            DevBuf d_recs, d_aux;
            int ws_save = ws_next;          // the same two workspace slots every pass
            if (!dmalloc(d_recs, nrec_pass * sizeof(FixRec)) || !dmalloc(d_aux, naux_pass)) { rc = -2; break; }
            ws_next = ws_save;
            launch_seg_gather(b_segs.as<SegDev>(), (int)ns, g_idx, g_seq, g_ro, g_ao, d_recs.as<FixRec>(), d_aux.as<uint8_t>(), st);
            HIPCHK(hipGetLastError());
            const size_t r0 = R.recs.size();
            R.recs.resize(r0 + nrec_pass);
            aux_pass.back().resize(naux_pass);
            if (pin_recs_used + nrec_pass <= PIN_RECS && pin_aux_used + naux_pass <= PIN_AUX) {
                // through pinned memory, without waiting: the stream orders the copy before the next pass's gather reuses d_recs
                HIPCHK(hipMemcpyAsync(recs_p + pin_recs_used, d_recs.p, nrec_pass * sizeof(FixRec), hipMemcpyDeviceToHost, st));
                if (naux_pass) HIPCHK(hipMemcpyAsync(aux_p + pin_aux_used, d_aux.p, naux_pass, hipMemcpyDeviceToHost, st));
                pinned_passes.push_back(PinnedPass{pin_recs_used, nrec_pass, pin_aux_used, naux_pass, r0, aux_pass.size() - 1});
                pin_recs_used += nrec_pass;
                pin_aux_used += naux_pass;
            } else {
                HIPCHK(hipMemcpyAsync(&R.recs[r0], d_recs.p, nrec_pass * sizeof(FixRec), hipMemcpyDeviceToHost, st));
                if (naux_pass) HIPCHK(hipMemcpyAsync(aux_pass.back().data(), d_aux.p, naux_pass, hipMemcpyDeviceToHost, st));
                HIPCHK(jk_stream_wait(st));
            }
        }
        const bool carry = pass < passes;           // another pass follows: carry the classes over, flag changed text
        if (carry) HIPCHK(hipMemsetAsync(b_flags.p, 0, flag_items, st));
        launch_seg_stitch(b_segs.as<SegDev>(), (int)ns, (uint8_t *const *)dOut, carry ? (uint8_t *const *)ptrClsOut : nullptr,
                          carry ? (uint8_t *const *)ptrFlags : nullptr, st);
        HIPCHK(hipGetLastError());
        for (int c = 0; c < n_chunks; ++c) len[c] = newlen[c];
        // The host's own bookkeeping went up from vectors of THIS iteration (segs, idx_base, seq_base, rec_off, aux_off: ordinary host
        // memory): an asynchronous copy may read its source when the stream gets to it, not when it is issued -- the vectors must
        // outlive that.  (Round 5: without this wait a chunk whose speculation had failed now and then came back with the records
        // or the text of freed memory: 1 run in ~1 000 of one fuzz case, on some boxes.)  The rare path can afford the wait.
        if (!ordinary) HIPCHK(jk_stream_wait(st));
        // the text just written is read next; the next pass writes into the arena that is free (never into the caller's buffer)
        std::swap(hIn, hOut);
        std::swap(dIn, dOut);
        if (pass == 0) { hOut = hSpare; dOut = dSpare; }
        std::swap(clsIn, clsOut);
        std::swap(ptrClsIn, ptrClsOut);
    }
    if (dbg) { (void)jk_stream_wait(st); fprintf(stderr, "[polish] passes: %.2f ms\n", now() - t_loop); }
    const double t_out = now();
    if (rc == 0) {
        HIPCHK(hipEventRecord(ev1, st));
        // ---- results to the host (or left where they are)
        if (keep_on_device) {
            R.d_seqs.resize(n_chunks);
            R.d_lens.assign(len.begin(), len.end());
            for (int c = 0; c < n_chunks; ++c) R.d_seqs[c] = hIn[c];
        }
        bool packed_back = false;
        if (packed_io && !keep_on_device) {
            int64_t at = 0;
            for (int c = 0; c < n_chunks; ++c) { io_offs[c] = at; at += len[c]; }
            io_offs[n_chunks] = at;
            if ((size_t)at <= text_bytes && (size_t)at <= (size_t)total_len + (size_t)(RM * std::max<int64_t>(4096, total_len / 8)) + 64) {
                packed_back = true;
                HIPCHK(hipMemcpyAsync(b_iooffs.p, io_offs.data(), io_offs.size() * 8, hipMemcpyHostToDevice, st));
                launch_copy_chunks(dIn, b_pack.as<uint8_t>(), b_iooffs.as<int64_t>(), n_chunks, false, st);
                HIPCHK(hipGetLastError());
                if (at) HIPCHK(hipMemcpyAsync(io_stage, b_pack.p, (size_t)at, hipMemcpyDeviceToHost, st));
            }
        }
        for (int c = 0; c < n_chunks && !keep_on_device && !packed_back; ++c) {
            R.seqs[c].resize((size_t)len[c]);
            if (len[c]) HIPCHK(hipMemcpyAsync(&R.seqs[c][0], hIn[c], (size_t)len[c], hipMemcpyDeviceToHost, st));
        }
        HIPCHK(jk_stream_wait(st));
        if (packed_back)
            for (int c = 0; c < n_chunks; ++c) R.seqs[c].assign(reinterpret_cast<const char *>(io_stage) + io_offs[c], (size_t)len[c]);
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
        R.seconds = ms * 1e-3;
        for (const PinnedPass &P : pinned_passes) {                                             // what travelled through pinned memory
            memcpy(&R.recs[P.r0], recs_p + P.rec_at, P.nrec * sizeof(FixRec));
            if (P.naux) memcpy(aux_pass[P.pass_i].data(), aux_p + P.aux_at, P.naux);
        }
        // records: order by chunk, pass, emission; regroup the aux bytes of 'x' records per chunk
        for (size_t p = 0; p < rec_pass_begin.size(); ++p) {
            const size_t b = rec_pass_begin[p], e = p + 1 < rec_pass_begin.size() ? rec_pass_begin[p + 1] : R.recs.size();
            for (size_t i = b; i < e; ++i) {
                FixRec &f = R.recs[i];
                if (f.kind == 'x') {
                    std::string &a = R.aux[f.chunk];
                    const size_t n = (size_t)f.aux_len + f.rep;
                    const uint32_t no = (uint32_t)a.size();
                    a.append(reinterpret_cast<const char *>(aux_pass[p].data()) + f.aux_off, n);
                    f.aux_off = no;
                }
            }
        }
        std::stable_sort(R.recs.begin(), R.recs.end(), [](const FixRec &a, const FixRec &b) {
            if (a.chunk != b.chunk) return a.chunk < b.chunk;
            if (a.pass != b.pass) return a.pass < b.pass;
            return a.seqno < b.seqno;
        });
    }
    (void)jk_stream_wait(st);
    if (dbg) fprintf(stderr, "[polish] D2H + record sort: %.2f ms; total %.2f ms\n", now() - t_out, now() - t_begin);
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    return rc;
}

// A batch as several lanes.  Chunk records are independent (the reference runs one process per batch file, src/jasper.sh:207-212),
// and one lane leaves most of the chip idle most of the time: its walks are chains of dependent lookups on a few thousand waves,
// the slowest segment of a pass (a path search: 0.6 ms) or the chain through a chunk's clean zones (0.6 ms in the later passes)
// sets the pass's length whatever the number of chunks, and between the passes the host cuts the next segments.  Groups of
// chunks, each with its own stream, host thread and workspace, fill each other's gaps: one's walk tail and host work run under
// another's kernels.  Measured (47 Mb in 18 chunk records): 4.7 ms in one lane, 4.55-4.7 in two, 4.35-4.65 in three, worse in
// four -- the HBM-bound scans only share the bandwidth and every lane still has its own chain of walks -- while the first
// call pays the workspace allocations once per lane (the drop-in end to end: 0.80 -> 0.97 s).  So a table's FIRST polishing call
// runs in one lane -- the drop-in polishes a genome of up to ~1 Gbase in one call -- and a table that is polished again (the
// further groups of a large genome, a resident service, the bench's steady state) takes three lanes from the second call on
// when the batch has a dozen chunks or more: 3.70 -> 3.59 ms per call with round 4's kernels.  JASPER_POLISH_LANES=n overrides.
// Results are per chunk, the same in any number of lanes -- only the records have to be merged back into batch order
// (tests/test_gpu_parity.py, tests/test_gpu_determinism.py).
int run_polish(Table &T, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre, int passes, int fix,
               PolishOut &R, std::string &err, bool device_in, bool keep_on_device, int roomy) {
    HIPCHK(hipSetDevice(T.device));
    if (T.materialize(err)) return -1;
    int lanes = T.polish_calls++ >= 1 && n_chunks >= 12 ? 3 : 1;
    if (const char *e = getenv("JASPER_POLISH_LANES")) lanes = atoi(e);                   // (tests, tuning)
    if (getenv("JASPER_POLISH_DEBUG")) lanes = 1;                                         // (its statistics are one lane's)
    lanes = std::max(1, std::min(std::min(lanes, (int)Table::POLISH_LANES_MAX), n_chunks / 2));
    if (lanes == 1) return run_polish_lane(T, 0, T.stream, n_chunks, seqs, lens, solid_thre, passes, fix, R, err, device_in, keep_on_device, roomy);

    // groups of about the same length: longest first to the lightest lane; inside a lane, batch order
    std::vector<int> order(n_chunks);
    for (int c = 0; c < n_chunks; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    std::vector<std::vector<int>> mine(lanes);
    std::vector<int64_t> load(lanes, 0);
    for (int c : order) {
        int l = 0;
        for (int j = 1; j < lanes; ++j) if (load[j] < load[l]) l = j;
        mine[l].push_back(c);
        load[l] += lens[c];
    }
    for (auto &m : mine) std::sort(m.begin(), m.end());
    // the lanes' streams start behind what the table's stream has been given so far (the counting)
    if (!T.polish_ev) HIPCHK(hipEventCreateWithFlags(&T.polish_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(T.polish_ev, T.stream));
    for (int l = 1; l < lanes; ++l) {
        if (!T.polish_stream[l]) HIPCHK(hipStreamCreateWithFlags(&T.polish_stream[l], hipStreamNonBlocking));
        HIPCHK(hipStreamWaitEvent(T.polish_stream[l], T.polish_ev, 0));
    }
    std::vector<PolishOut> out(lanes);
    std::vector<std::string> errs(lanes);
    std::vector<int> rcs(lanes, 0);
    std::vector<std::vector<const char *>> lseqs(lanes);
    std::vector<std::vector<int64_t>> llens(lanes);
    for (int l = 0; l < lanes; ++l)
        for (int c : mine[l]) { lseqs[l].push_back(seqs[c]); llens[l].push_back(lens[c]); }
    auto run = [&](int l) {
        rcs[l] = run_polish_lane(T, l, l ? T.polish_stream[l] : T.stream, (int)mine[l].size(), lseqs[l].data(), llens[l].data(), solid_thre, passes, fix, out[l],
                                 errs[l], device_in, keep_on_device, roomy);
    };
    {
        std::vector<std::thread> th;
        for (int l = 1; l < lanes; ++l) th.emplace_back(run, l);
        run(0);
        for (auto &t : th) t.join();
    }
    // (every lane has waited for its stream; a failure of any lane fails the call: HIP error, then "the reference exits 1", then capacity)
    for (int want : {-1, -4, -2})
        for (int l = 0; l < lanes; ++l)
            if (rcs[l] == want) { err = errs[l]; return want; }
    for (int l = 0; l < lanes; ++l)
        if (rcs[l]) { err = errs[l]; return rcs[l]; }
    R.seqs.assign(n_chunks, std::string());
    R.aux.assign(n_chunks, std::string());
    R.d_seqs.clear();
    R.d_lens.clear();
    if (keep_on_device) { R.d_seqs.assign(n_chunks, nullptr); R.d_lens.assign(n_chunks, 0); }
    R.recs.clear();
    R.qv[0] = R.qv[1] = R.qv[2] = R.qv[3] = 0;
    R.qv_chunk.assign((size_t)4 * n_chunks, 0);
    R.lookups = 0; R.seconds = 0; R.n_segments = 0; R.n_respeculated = 0;
    for (int l = 0; l < lanes; ++l) {
        PolishOut &O = out[l];
        for (size_t j = 0; j < mine[l].size(); ++j) {
            const int c = mine[l][j];
            R.seqs[c].swap(O.seqs[j]);
            R.aux[c].swap(O.aux[j]);
            if (keep_on_device) { R.d_seqs[c] = O.d_seqs[j]; R.d_lens[c] = O.d_lens[j]; }
            for (int q = 0; q < 4; ++q) R.qv_chunk[4 * (size_t)c + q] = O.qv_chunk[4 * j + q];
        }
        for (FixRec f : O.recs) { f.chunk = (uint32_t)mine[l][f.chunk]; R.recs.push_back(f); }
        for (int q = 0; q < 4; ++q) R.qv[q] += O.qv[q];
        R.lookups += O.lookups;
        R.n_segments += O.n_segments;
        R.n_respeculated += O.n_respeculated;
        R.seconds = std::max(R.seconds, O.seconds);
    }
    std::stable_sort(R.recs.begin(), R.recs.end(), [](const FixRec &a, const FixRec &b) {
        if (a.chunk != b.chunk) return a.chunk < b.chunk;
        if (a.pass != b.pass) return a.pass < b.pass;
        return a.seqno < b.seqno;
    });
    return 0;
}

}  // namespace jk
