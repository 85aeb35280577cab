// pgunzip.hpp -- ONE gzip stream inflated by many threads (host side of the read ingest).
//
// `zcat -f reads.fq.gz | jellyfish count /dev/stdin` (src/jasper.sh:177) inflates on one core: ~0.3 GB/s of text, while the GPU
// parses text at 10 GB/s -- for a real read set (two large .fastq.gz files) that pipe is the wall.  A deflate stream has no
// index, and a block may copy bytes from the 32 KB before it, so it cannot simply be cut.  What is done here:
//
//   1. the compressed file is cut at nominal offsets every CHUNK bytes; from each, a worker searches forward, bit by bit,
//      for something that parses as the header of a dynamic-Huffman block (complete code-length code, complete literal
//      and distance codes, an end-of-block symbol -- the checks of zlib's inflate_table) and from which zlib really inflates
//      (a gzip member header followed by such a block is accepted too: there the window is known to be empty);
//   2. every chunk is inflated from its boundary to the next chunk's boundary TWICE, with two different made-up 32 KB
//      dictionaries A and C in place of the unknown window, chosen so that A[p] != C[p] for every window position p and
//      (A[p], C[p]) -> p is one-to-one.  The compressed bits parse the same way whatever the window holds, so the two outputs
//      differ exactly at the bytes that (directly or through copies of copies) come out of the unknown window, and the pair
//      of bytes there names the window position;
//   3. the chunks are stitched in order: chunk 0 starts at the real beginning; a chunk is accepted only if the chunk before it
//      ended exactly on its boundary (so a false boundary is simply never reached: the decoder before it runs on to the next
//      one), and its marked bytes are filled in from the last 32 KB of the text before it.
//   Nothing is assumed about the text (no "ASCII only"); the gzip trailer's CRC-32 and length are checked per member as zlib does.
// Same idea as the two-pass parallel decoders for FASTQ (pugz); the symbolic window is replaced by the two-dictionary
// trick so that zlib's own inflate does all the decoding.
#pragma once
#include <zlib.h>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace jk {

class ParallelGunzip {
  public:
    static constexpr size_t WIN = 32768;
    // threads >= 2; chunk = compressed bytes per unit of work
    ParallelGunzip(const char *path, int threads, size_t chunk = 4u << 20) : path_(path), threads_(std::max(2, threads)), chunk_(std::max<size_t>(chunk, 1u << 16)) {
        for (size_t p = 0; p < WIN; ++p) {
            dictA_[p] = (uint8_t)(p & 0xFF);
            dictC_[p] = (uint8_t)((dictA_[p] + 1 + (p >> 8)) & 0xFF);       // 1 + (p >> 8) is in [1, 128]: never equal to A[p]
        }
    }
    ~ParallelGunzip() { close_file(); }
    // compressed bytes per unit of work for a file of this size: 4 MiB, less for files that would otherwise not keep the
    // threads busy for a few waves (not below 256 KiB: a boundary search costs ~1 ms, a chunk should cost far more)
    static size_t chunk_for(size_t file_bytes, int threads) {
        const size_t want = file_bytes / ((size_t)std::max(2, threads) * 3);
        return std::min<size_t>(4u << 20, std::max<size_t>(256u << 10, want));
    }
    const std::string &error() const { return err_; }

    // opens the file; false when it is not something this class handles (not a regular gzip file, too small to be worth it):
    // the caller then reads it the ordinary way
    bool open() {
        fd_ = ::open(path_.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < 4 * chunk_) { close_file(); return false; }
        n_ = (size_t)st.st_size;
        void *m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) { close_file(); return false; }
        data_ = (const uint8_t *)m;
        (void)madvise(m, n_, MADV_SEQUENTIAL);
        size_t hdr = gzip_header_len(0);
        if (!hdr) { close_file(); return false; }
        cur_bit_ = 8ull * hdr;
        next_nominal_ = 0;
        crc_ = crc32(0L, Z_NULL, 0);
        member_len_ = 0;
        return true;
    }

    // the next pieces of the inflated stream, in order (one wave: up to `threads` chunks); false at the end or on error
    // (error() non-empty)
    bool next(std::vector<std::vector<uint8_t>> &pieces) {
        pieces.clear();
        if (done_ || !data_) return false;
        if (!run_wave(pieces)) { done_ = true; return false; }
        return true;
    }

  private:
    struct Chunk {
        uint64_t start_bit = 0;          // boundary found (bit position of a block start)
        bool fresh_window = false;        // the boundary is the first block of a gzip member: nothing before it can be referenced
        const uint8_t *known_window = nullptr;   // the real 32 KB before the boundary, when they are known (the first chunk of a wave)
        bool found = false;
        std::vector<uLong> part_crc;      // CRC-32 of the text up to each member end, and of the rest (after the marked bytes are filled in)
        // result of the speculative decode
        bool decoded = false, failed = false;
        uint64_t end_bit = 0;             // where the decode stopped: a later chunk's start_bit, or the end of the stream
        bool hit_eof = false;
        std::vector<uint8_t> out;         // run A; its bytes that differ from run C come out of the window before the chunk
        std::vector<uint8_t> outC;        // run C (empty when the window was known)
        uint8_t window[WIN];              // the 32 KB before the chunk's text, once they are known (stitching)
        size_t window_n = 0;              // how many of them exist (less than 32 KB only near the start of the stream)
        bool bad_ref = false;
        // member bookkeeping for the trailer checks: the text of this chunk is split at member ends
        std::vector<size_t> member_ends;  // offsets in out where a member ended
        std::vector<std::pair<uint32_t, uint32_t>> trailers;   // (crc, isize) read at each of them
    };

    std::string path_, err_;
    int threads_;
    size_t chunk_;
    int fd_ = -1;
    const uint8_t *data_ = nullptr;
    size_t n_ = 0;
    bool done_ = false;
    uint64_t cur_bit_ = 0;                // where the accepted text ends in the compressed stream (a block start)
    size_t next_nominal_ = 0;             // next nominal cut (byte offset) not yet searched
    uint8_t tail_[WIN];                   // the last WIN bytes of accepted text
    size_t tail_n_ = 0;
    bool cur_fresh_ = true;               // cur_bit_ is the first block of a member
    uLong crc_ = 0;
    uint64_t member_len_ = 0;
    uint8_t dictA_[WIN], dictC_[WIN];

    void close_file() {
        if (data_) munmap((void *)data_, n_);
        data_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }

    // ---- gzip member header at byte offset o: its length, or 0 if it is not one (RFC 1952) -------------------------
    size_t gzip_header_len(size_t o) const {
        if (o + 18 > n_) return 0;
        const uint8_t *p = data_ + o;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
        const int flg = p[3];
        size_t q = o + 10;
        if (flg & 4) { if (q + 2 > n_) return 0; const size_t xlen = p[10] | (p[11] << 8); q += 2 + xlen; }
        if (flg & 8) { while (q < n_ && data_[q]) ++q; ++q; }
        if (flg & 16) { while (q < n_ && data_[q]) ++q; ++q; }
        if (flg & 2) q += 2;
        return q < n_ ? q - o : 0;
    }

    // ---- does a dynamic-Huffman block header start at bit position b?  (RFC 1951 3.2.7; the completeness rules are those of
    //      zlib's inflate_table: the code-length code and the literal/length code must be complete, a distance code may be
    //      incomplete only if it has a single code) -----------------------------------------------------------------------------
    struct Bits {
        const uint8_t *d; size_t n; uint64_t pos;
        bool ok = true;
        uint32_t get(int k) {
            uint32_t v = 0;
            for (int i = 0; i < k; ++i) {
                const uint64_t by = pos >> 3;
                if (by >= n) { ok = false; return 0; }
                v |= (uint32_t)((d[by] >> (pos & 7)) & 1u) << i;
                ++pos;
            }
            return v;
        }
    };
    static bool complete_code(const uint8_t *len, int n, int maxbits, bool allow_single) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[len[i]]++;
        if (count[0] == n) return allow_single;            // no codes at all (a distance code of a block without matches)
        int left = 1;
        for (int l = 1; l <= maxbits; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left == 0) return true;
        return allow_single && (n - count[0]) == 1 && count[1] == 1;
    }
    bool dynamic_header_at(uint64_t b) const {
        Bits r{data_, n_, b};
        if (r.get(1) != 0) return false;                    // (the last block of a member is not worth cutting at)
        if (r.get(2) != 2) return false;
        const int hlit = (int)r.get(5) + 257, hdist = (int)r.get(5) + 1, hclen = (int)r.get(4) + 4;
        if (!r.ok || hlit > 286 || hdist > 30) return false;
        static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)r.get(3);
        if (!r.ok || !complete_code(cl, 19, 7, false)) return false;
        // canonical code of the 19 code-length symbols
        int count[8] = {0}, offs[8], symbol[19];
        for (int i = 0; i < 19; ++i) count[cl[i]]++;
        offs[1] = 0;
        for (int l = 1; l < 7; ++l) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < 19; ++i) if (cl[i]) symbol[offs[cl[i]]++] = i;
        auto decode = [&]() -> int {
            int code = 0, first = 0, index = 0;
            for (int l = 1; l <= 7; ++l) {
                code |= (int)r.get(1);
                const int c = count[l];
                if (code - c < first) return symbol[index + (code - first)];
                index += c; first += c; first <<= 1; code <<= 1;
            }
            return -1;
        };
        uint8_t lens[286 + 30];
        int i = 0;
        while (i < hlit + hdist) {
            const int s = decode();
            if (s < 0 || !r.ok) return false;
            if (s < 16) lens[i++] = (uint8_t)s;
            else {
                int rep, val = 0;
                if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)r.get(2); }
                else if (s == 17) rep = 3 + (int)r.get(3);
                else rep = 11 + (int)r.get(7);
                if (i + rep > hlit + hdist) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (!r.ok || lens[256] == 0) return false;
        if (!complete_code(lens, hlit, 15, false)) return false;
        if (!complete_code(lens + hlit, hdist, 15, true)) return false;
        return true;
    }
    // zlib agrees: 64 KB of output (or the end of the block sequence) come out of it without an error
    bool inflates_from(uint64_t b) const {
        z_stream s;
        memset(&s, 0, sizeof s);
        if (inflateInit2(&s, -15) != Z_OK) return false;
        size_t by = (size_t)(b >> 3);
        const int bit = (int)(b & 7);
        if (bit) { inflatePrime(&s, 8 - bit, data_[by] >> bit); ++by; }
        inflateSetDictionary(&s, dictA_, WIN);
        std::vector<uint8_t> tmp(1u << 16);
        s.next_in = const_cast<Bytef *>(data_ + by);
        s.avail_in = (uInt)std::min<size_t>(n_ - by, 1u << 20);
        s.next_out = tmp.data();
        s.avail_out = (uInt)tmp.size();
        const int rc = inflate(&s, Z_SYNC_FLUSH);
        inflateEnd(&s);
        return rc == Z_OK || rc == Z_STREAM_END || (rc == Z_BUF_ERROR && s.avail_out == 0);
    }
    // first acceptable boundary at or after byte offset o (and before byte offset lim); false if there is none
    bool find_boundary(size_t o, size_t lim, Chunk &c) const {
        lim = std::min(lim, n_);
        for (uint64_t b = 8ull * o; b < 8ull * lim; ++b) {
            if ((b & 7) == 0) {                                         // a gzip member header here?
                const size_t by = (size_t)(b >> 3);
                if (data_[by] == 0x1f && by + 18 < n_ && data_[by + 1] == 0x8b) {
                    const size_t h = gzip_header_len(by);
                    if (h && inflates_from(8ull * (by + h))) { c.start_bit = 8ull * (by + h); c.fresh_window = true; c.found = true; return true; }
                }
            }
            // cheap prefix test before the full parse: BFINAL = 0, BTYPE = 10b  ->  bits 0, 0, 1
            const uint64_t by = b >> 3;
            const int sh = (int)(b & 7);
            const uint32_t w = (uint32_t)data_[by] | ((by + 1 < n_ ? (uint32_t)data_[by + 1] : 0u) << 8);
            if (((w >> sh) & 7u) != 4u) continue;
            if (dynamic_header_at(b) && inflates_from(b)) { c.start_bit = b; c.fresh_window = false; c.found = true; return true; }
        }
        return false;
    }

    // ---- inflate from start_bit until the position is one of `stops` (sorted bit positions > start_bit) or the stream ends --------
    // dict: the 32 KB to use as the window (null: the window is empty, the start is a member's first block)
    struct RunOut { std::vector<uint8_t> out; uint64_t end_bit = 0; bool eof = false, ok = false;
                    std::vector<size_t> member_ends; std::vector<std::pair<uint32_t, uint32_t>> trailers; };
    size_t run_in_max() const { return (size_t)(threads_ + 2) * chunk_; }                  // compressed bytes one run may consume ...
    size_t run_out_max() const { return std::max<size_t>(32 * chunk_, 64u << 20); }        // ... and text it may produce, before it ends at a block end
    void inflate_run(uint64_t start_bit, const uint8_t *dict, const std::vector<uint64_t> &stops, RunOut &R) const {
        R.ok = false;
        z_stream s;
        memset(&s, 0, sizeof s);
        if (inflateInit2(&s, -15) != Z_OK) return;
        size_t by = (size_t)(start_bit >> 3);
        const int bit = (int)(start_bit & 7);
        if (bit) { inflatePrime(&s, 8 - bit, data_[by] >> bit); ++by; }
        if (dict) inflateSetDictionary(&s, dict, WIN);
        R.out.resize(std::max<size_t>(chunk_ * 5, 1u << 20));
        size_t produced = 0;
        s.next_in = const_cast<Bytef *>(data_ + by);
        size_t in_left = n_ - by;
        s.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
        in_left -= s.avail_in;
        size_t stop_i = 0;
        for (;;) {
            if (produced == R.out.size()) R.out.resize(R.out.size() + R.out.size() / 2);
            s.next_out = R.out.data() + produced;
            const size_t room = std::min<size_t>(R.out.size() - produced, 1u << 30);
            s.avail_out = (uInt)room;
            if (s.avail_in == 0 && in_left) { s.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30); in_left -= s.avail_in; }
            const int rc = inflate(&s, Z_BLOCK);
            produced += room - s.avail_out;
            if (rc == Z_STREAM_END) {
                // end of a member: trailer (CRC-32, ISIZE), then possibly another member
                size_t pos = (size_t)(s.next_in - data_);
                if (pos + 8 > n_) break;                                    // truncated
                const uint32_t crc = data_[pos] | (data_[pos + 1] << 8) | (data_[pos + 2] << 16) | ((uint32_t)data_[pos + 3] << 24);
                const uint32_t isz = data_[pos + 4] | (data_[pos + 5] << 8) | (data_[pos + 6] << 16) | ((uint32_t)data_[pos + 7] << 24);
                R.member_ends.push_back(produced);
                R.trailers.emplace_back(crc, isz);
                pos += 8;
                while (pos < n_ && data_[pos] == 0) ++pos;                  // (zero padding between members, as gzip tolerates)
                if (pos >= n_) { R.eof = true; R.end_bit = 8ull * n_; R.ok = true; break; }
                const size_t h = gzip_header_len(pos);
                if (!h) break;                                              // trailing garbage: let the ordinary reader report it
                const uint64_t nb = 8ull * (pos + h);
                while (stop_i < stops.size() && stops[stop_i] < nb) ++stop_i;
                if (stop_i < stops.size() && stops[stop_i] == nb) { R.end_bit = nb; R.ok = true; break; }
                inflateReset2(&s, -15);
                s.next_in = const_cast<Bytef *>(data_ + pos + h);
                in_left = n_ - (pos + h);
                s.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
                in_left -= s.avail_in;
                continue;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) break;                     // data error: not a real boundary, or a damaged file
            if (rc == Z_BUF_ERROR && s.avail_in == 0 && in_left == 0 && s.avail_out != 0) break;      // truncated stream
            if ((s.data_type & 128) && !(s.data_type & 64)) {               // just after a block (not the last one)
                const uint64_t posb = 8ull * (uint64_t)(s.next_in - data_) - (uint64_t)(s.data_type & 63);
                while (stop_i < stops.size() && stops[stop_i] < posb) ++stop_i;
                if (stop_i < stops.size() && stops[stop_i] == posb) { R.end_bit = posb; R.ok = true; break; }
                // A run is bounded: with no usable boundary ahead (streams of stored or fixed-Huffman blocks, every candidate a false
                // positive) it would otherwise inflate to the end of the file into ONE growing vector.  It ends at the first block
                // end past the bound -- a real block start, so the next wave carries on from there (the chunks of this wave that
                // started beyond it are dropped by the stitch like any chunk whose predecessor did not end on its start).
                if (produced >= run_out_max() || posb - start_bit >= 8ull * run_in_max()) { R.end_bit = posb; R.ok = true; break; }
            }
        }
        R.out.resize(produced);
        inflateEnd(&s);
    }

    void decode_chunk(Chunk &c, const std::vector<uint64_t> &stops) const {
        RunOut A, C;
        inflate_run(c.start_bit, c.fresh_window ? nullptr : (c.known_window ? c.known_window : dictA_), stops, A);
        if (!A.ok) { c.failed = true; c.decoded = true; return; }
        if (!c.fresh_window && !c.known_window) {
            inflate_run(c.start_bit, dictC_, stops, C);
            if (!C.ok || C.end_bit != A.end_bit || C.out.size() != A.out.size()) { c.failed = true; c.decoded = true; return; }
            c.outC.swap(C.out);
        }
        c.out.swap(A.out);
        c.end_bit = A.end_bit;
        c.hit_eof = A.eof;
        c.member_ends.swap(A.member_ends);
        c.trailers.swap(A.trailers);
        c.decoded = true;
    }
    // the bytes of [lo, hi) that came out of the window: where the two runs differ, the pair of bytes names the window position
    static void resolve(Chunk &c, size_t lo, size_t hi) {
        if (c.outC.empty()) return;
        uint8_t *a = c.out.data();
        const uint8_t *cc = c.outC.data();
        const size_t missing = WIN - c.window_n;
        for (size_t i = lo; i < hi; ++i) {
            if (a[i] == cc[i]) continue;
            const size_t p = (size_t)a[i] | ((((unsigned)cc[i] - a[i] - 1u) & 0xFFu) << 8);
            if (p < missing) { c.bad_ref = true; continue; }       // refers to before the start of the stream
            a[i] = c.window[p];
        }
    }

    // CRC-32 of the parts of a chunk's text (cut at member ends), computed by a worker once the text is final
    static void crc_parts(Chunk &c) {
        c.part_crc.clear();
        size_t done = 0;
        for (size_t m = 0; m <= c.member_ends.size(); ++m) {
            const size_t upto = m < c.member_ends.size() ? c.member_ends[m] : c.out.size();
            uLong v = crc32(0L, Z_NULL, 0);
            size_t q = done;
            while (q < upto) { const size_t step = std::min<size_t>(upto - q, 1u << 30); v = crc32(v, c.out.data() + q, (uInt)step); q += step; }
            c.part_crc.push_back(v);
            done = upto;
        }
    }
    // accepted text: the members' CRC-32 / length against their trailers (as zlib's gzread checks them)
    bool check_trailers(const Chunk &c) {
        size_t done = 0;
        for (size_t m = 0; m <= c.member_ends.size(); ++m) {
            const size_t upto = m < c.member_ends.size() ? c.member_ends[m] : c.out.size();
            crc_ = crc32_combine(crc_, c.part_crc[m], (z_off_t)(upto - done));
            member_len_ += upto - done;
            done = upto;
            if (m < c.member_ends.size()) {
                if ((uint32_t)crc_ != c.trailers[m].first || (uint32_t)member_len_ != c.trailers[m].second) { err_ = "crc or length error in " + path_; return false; }
                crc_ = crc32(0L, Z_NULL, 0);
                member_len_ = 0;
            }
        }
        return true;
    }
    void keep_tail(const uint8_t *p, size_t n) {
        if (n >= WIN) { memcpy(tail_, p + n - WIN, WIN); tail_n_ = WIN; }
        else {
            const size_t keep = std::min(tail_n_, WIN - n);
            memmove(tail_, tail_ + (tail_n_ - keep), keep);
            memcpy(tail_ + keep, p, n);
            tail_n_ = keep + n;
        }
    }

    // one wave: boundaries for the next `threads_` nominal cuts, speculative decodes in parallel, stitch
    bool run_wave(std::vector<std::vector<uint8_t>> &pieces) {
        // chunk 0 of the wave starts where the accepted text ends (a real block start, its window known); the others at
        // boundaries found from the nominal cuts; one more boundary (the first cut of the NEXT wave) only serves as a stop
        const size_t base = std::max<size_t>(next_nominal_, (size_t)(cur_bit_ >> 3) + chunk_ / 2);
        std::vector<Chunk> ch_store((size_t)threads_ + 1);
        Chunk *ch = ch_store.data();
        const size_t nch = ch_store.size();
        uint8_t window0[WIN];
        memset(window0, 0, WIN - tail_n_);
        memcpy(window0 + (WIN - tail_n_), tail_, tail_n_);
        ch[0].start_bit = cur_bit_; ch[0].found = true; ch[0].fresh_window = cur_fresh_; ch[0].known_window = window0;
        {
            std::vector<std::thread> th;
            for (int i = 1; i <= threads_; ++i) {
                const size_t o = base + (size_t)(i - 1) * chunk_;
                if (o + 64 >= n_) continue;
                th.emplace_back([this, ch, i, o] { find_boundary(o, o + chunk_, ch[(size_t)i]); });
            }
            for (auto &t : th) t.join();
        }
        next_nominal_ = base + (size_t)threads_ * chunk_;
        std::vector<uint64_t> stops;
        for (size_t i = 1; i < nch; ++i) if (ch[i].found && ch[i].start_bit > cur_bit_) stops.push_back(ch[i].start_bit);
        std::sort(stops.begin(), stops.end());
        stops.erase(std::unique(stops.begin(), stops.end()), stops.end());
        {
            std::vector<std::thread> th;
            for (size_t i = 0; i + 1 < nch; ++i) {           // (the last one is only a stop)
                if (!ch[i].found || (i > 0 && ch[i].start_bit <= cur_bit_)) continue;
                std::vector<uint64_t> later;
                for (uint64_t s : stops) if (s > ch[i].start_bit) later.push_back(s);
                th.emplace_back([this, ch, i, later] { decode_chunk(ch[i], later); });
            }
            for (auto &t : th) t.join();
        }
        // stitch: follow the chain of ends from the accepted position
        std::vector<Chunk *> chain;
        for (;;) {
            Chunk *c = nullptr;
            for (size_t i = 0; i + 1 < nch; ++i)
                if (ch[i].found && ch[i].decoded && ch[i].start_bit == cur_bit_) { c = &ch[i]; break; }
            if (!c) break;                                          // the chain continues in the next wave
            if (c->failed) {
                // chunk 0 always starts at a real block with its real window: if IT fails, the file is damaged or truncated.
                // A later chunk on the chain also starts at a real block, but is simply decoded again as chunk 0 of the next wave.
                if (c == &ch[0]) { err_ = "read error in " + path_; return false; }
                break;
            }
            // the 32 KB before this chunk are known now; its own last 32 KB are filled in at once (the next chunk needs them),
            // the rest by the workers below
            memset(c->window, 0, WIN - tail_n_);
            memcpy(c->window + (WIN - tail_n_), tail_, tail_n_);
            c->window_n = tail_n_;
            const size_t m = c->out.size();
            resolve(*c, m > WIN ? m - WIN : 0, m);
            keep_tail(c->out.data(), m);
            cur_bit_ = c->end_bit;
            cur_fresh_ = !c->member_ends.empty() && c->member_ends.back() == m;
            c->decoded = false;                                      // consumed
            chain.push_back(c);
            if (c->hit_eof) { done_ = true; break; }
        }
        if (chain.empty()) { if (err_.empty()) err_ = "read error in " + path_; return false; }
        {   // per chunk in parallel: the remaining window bytes, then the checksums of the final text
            std::vector<std::thread> th;
            for (Chunk *c : chain) th.emplace_back([c] {
                const size_t m = c->out.size();
                resolve(*c, 0, m > WIN ? m - WIN : 0);
                std::vector<uint8_t>().swap(c->outC);
                crc_parts(*c);
            });
            for (auto &t : th) t.join();
        }
        for (Chunk *c : chain) {
            if (c->bad_ref) { err_ = "read error in " + path_; return false; }
            if (!check_trailers(*c)) return false;
            pieces.emplace_back(std::move(c->out));
        }
        return true;
    }
};

}  // namespace jk
