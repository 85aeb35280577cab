// jfwrite.hip -- the table as a Jellyfish "binary/sorted" database (what `jellyfish count -o` leaves in
// mer_counts$K.jf, src/jasper.sh:177), readable by jellyfish 2.3.0's own query / dump / histo and by QueryMerFile.
//
// Format (JF::include/jellyfish/generic_file_header.hpp:88-111, file_header.hpp:26-108, binary_dumper.hpp:36-40):
//   9 ASCII digits = length of what follows up to the records, terse JSON header, NUL padding to a multiple of 8 bytes,
//   then records [key: ceil(2k/8) bytes, little-endian words][count: counter_len = 4 bytes LE, min(count, 2^32-1)].
// Order: records are sorted by (pos, key), pos = matrix1 * key & (size-1), because the reader binary-searches on pos and
// breaks ties by key (binary_dumper.hpp:148-199).  Jellyfish draws a random invertible GF(2) matrix per run and stores it
// in the header; the reader takes whatever the header says, including the "identity" form (file_header.hpp:36-47:
// pos = the low r bits of the key).  We write that form: no matrix product is needed on either side, and (pos, key)
// order becomes the numeric order of the key rotated right by r bits -- one radix sort on the GPU.
// (Consequence: `jellyfish merge` refuses to mix such a file with files from another matrix, as it does for any two
// Jellyfish runs with different matrices.)
#include "table.hpp"
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>

namespace jk {

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                     \
            rc = -1;                                                                  \
            goto done;                                                                \
        }                                                                             \
    } while (0)

struct HiCnt { unsigned long long hi; unsigned int cnt; unsigned int pad; };
struct LoCnt { unsigned long long lo; unsigned int cnt; unsigned int pad; };

// every stored key -> K = key rotated right by r within B bits (its low r bits, the reader's `pos`, become the top bits)
__global__ __launch_bounds__(256) void export_rotated_kernel(TableDev T, int r, unsigned long long *__restrict__ klo, HiCnt *__restrict__ hic,
                                                             unsigned long long *__restrict__ counter, uint64_t cap) {
    const uint64_t nslots = T.mask + 1;
    const int B = T.B;
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x; i0 < nslots; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + threadIdx.x;
        ulonglong2 e = make_ulonglong2(0ull, 0ull);
        if (i < nslots) e = *reinterpret_cast<const ulonglong2 *>(T.slots + 2 * i);
        const bool have = e.x != 0ull && e.y != 0ull;
        const uint64_t m = __ballot(have);                       // one atomic per wave, not per key
        if (!m) continue;
        const int leader = (int)__builtin_ctzll(m);
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        if (!have) continue;
        const unsigned long long idx = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        const u128 key = unmix(slot_hash(T, i, e.x), B);
        const u128 K = r == 0 || r == B ? key : bor(shl(band(key, maskbits(r)), B - r), shr(key, r));
        if (idx < cap) {
            klo[idx] = K.lo;
            HiCnt h;
            h.hi = K.hi; h.cnt = clamp32(e.y); h.pad = 0;
            hic[idx] = h;
        }
    }
}
__global__ __launch_bounds__(256) void swap_words_kernel(const unsigned long long *__restrict__ klo, const HiCnt *__restrict__ hic, uint64_t n,
                                                         unsigned long long *__restrict__ khi, LoCnt *__restrict__ loc) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        khi[i] = hic[i].hi;
        LoCnt l;
        l.lo = klo[i]; l.cnt = hic[i].cnt; l.pad = 0;
        loc[i] = l;
    }
}

// the file's record bytes, formatted on the device: [key: kb bytes LE][count: 4 bytes LE] per sorted entry
__global__ __launch_bounds__(256) void format_records_kernel(const unsigned long long *__restrict__ key_word, const HiCnt *__restrict__ other, uint64_t n0,
                                                             uint64_t m, int B, int r, int kb, uint8_t *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const HiCnt o = other[n0 + i];                                             // (the other word, count) -- HiCnt and LoCnt share a layout
        const u128 K = B > 64 ? mk(key_word[n0 + i], o.hi) : mk(0, key_word[n0 + i]);
        const u128 key = r == 0 || r == B ? K : bor(shl(band(K, maskbits(B - r)), r), shr(K, B - r));
        uint8_t *dst = out + i * (uint64_t)(kb + 4);
        for (int b = 0; b < kb; ++b) dst[b] = (uint8_t)((b < 8 ? key.lo >> (8 * b) : key.hi >> (8 * (b - 8))) & 0xFFu);
        for (int b = 0; b < 4; ++b) dst[kb + b] = (uint8_t)((o.cnt >> (8 * b)) & 0xFFu);
    }
}

// the reverse: raw file records -> table.  Keys are inserted AS STORED (no canonicalisation), so that a lookup of
// canonical(query) hits exactly when the reference's binary search would (JF::include/jellyfish/binary_dumper.hpp:148-199)
__global__ __launch_bounds__(256) void add_jf_records_kernel(const uint8_t *__restrict__ raw, uint64_t n, int kb, int cl, TableDev T) {
    unsigned long long fresh = 0;
    const u128 kmask = maskbits(T.B);
    const int rec = kb + cl;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *p = raw + i * (uint64_t)rec;
        unsigned long long lo = 0, hi = 0, c = 0;
        for (int b = 0; b < kb && b < 8; ++b) lo |= (unsigned long long)p[b] << (8 * b);
        for (int b = 8; b < kb; ++b) hi |= (unsigned long long)p[b] << (8 * (b - 8));
        for (int b = 0; b < cl; ++b) c |= (unsigned long long)p[kb + b] << (8 * b);
        if (c) fresh += table_add_or_spill(T, mix(band(mk(hi, lo), kmask), T.B), c);
    }
    for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&T.stats[ST_DISTINCT], fresh);
}

static std::string json_str(const char *s) {
    std::string o = "\"";
    for (; *s; ++s) {
        const unsigned char c = (unsigned char)*s;
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    return o + "\"";
}

int Table::write_jf(const char *path, const char *const *cmdline, int n_cmd, std::string &err, int r_bits, int what) {
    int rc = 0;
    hipStream_t stream = this->stream;       // (the table's own until the writer's is there: see below)
    FILE *f = nullptr;
    unsigned long long *d_klo[2] = {nullptr, nullptr}, *d_khi[2] = {nullptr, nullptr}, *d_counter = nullptr;
    HiCnt *d_hic[2] = {nullptr, nullptr};
    LoCnt *d_loc[2] = {nullptr, nullptr};
    void *d_tmp = nullptr;
    uint8_t *d_fmt = nullptr, *h_fmt = nullptr;
    hipEvent_t ev_fmt[2] = {nullptr, nullptr};
    const int B = d.B;
    const int r = r_bits >= 0 ? (r_bits < B ? r_bits : B) : (d.s < B ? d.s : B);   // rows of the identity matrix = log2(size)
    const int kb = (B + 7) / 8;
    uint64_t n = 0;
    if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return -1; }
    if (materialize(err)) return -1;
    if (read_stats(err)) return -1;
    // (a stream of its own: the table is only read from here on, and the caller may be polishing through the table's own stream
    //  meanwhile -- jasper_amd/cli.py writes the database file beside the stages that follow the counting; read_stats has waited
    //  for what the table's stream held)
    if (!jf_stream) HIPCHK(hipStreamCreateWithFlags(&jf_stream, hipStreamNonBlocking));
    stream = jf_stream;
    if (what != 2) {
        const uint64_t cap = h_stats[ST_DISTINCT] + 1;
        unsigned long long zero = 0, cnt = 0;
        HIPCHK(hipMalloc((void **)&d_counter, 8));
        HIPCHK(hipMemcpyAsync(d_counter, &zero, 8, hipMemcpyHostToDevice, stream));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc((void **)&d_klo[i], cap * 8));
            HIPCHK(hipMalloc((void **)&d_hic[i], cap * sizeof(HiCnt)));
        }
        hipLaunchKernelGGL(export_rotated_kernel, dim3(2048), dim3(256), 0, stream, d, r, d_klo[0], d_hic[0], d_counter, cap);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&cnt, d_counter, 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(jk_stream_wait(stream));
        if (cnt > cap) { err = "internal: more keys than the table reports"; rc = -1; goto done; }
        n = cnt;
        // stable LSD radix sort of the B-bit K: low word first, then (B > 64) the high word
        const int lo_bits = B < 64 ? B : 64;
        size_t tmp1 = 0, tmp2 = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp1, d_klo[0], d_klo[1], d_hic[0], d_hic[1], n, 0, lo_bits, stream));
        if (B > 64) {
            HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp2, d_klo[0], d_klo[1], (LoCnt *)nullptr, (LoCnt *)nullptr, n, 0, B - 64, stream));
        }
        HIPCHK(hipMalloc(&d_tmp, std::max(tmp1, tmp2) + 256));
        if (n) HIPCHK(rocprim::radix_sort_pairs(d_tmp, tmp1, d_klo[0], d_klo[1], d_hic[0], d_hic[1], n, 0, lo_bits, stream));
        // result: d_klo[1], d_hic[1]
        if (B > 64 && n) {
            // reuse the "0" buffers: khi keys + (lo, cnt) payload
            d_khi[0] = d_klo[0];
            d_loc[0] = reinterpret_cast<LoCnt *>(d_hic[0]);
            HIPCHK(hipMalloc((void **)&d_khi[1], n * 8));
            HIPCHK(hipMalloc((void **)&d_loc[1], n * sizeof(LoCnt)));
            hipLaunchKernelGGL(swap_words_kernel, dim3(2048), dim3(256), 0, stream, d_klo[1], d_hic[1], n, d_khi[0], d_loc[0]);
            HIPCHK(hipGetLastError());
            HIPCHK(rocprim::radix_sort_pairs(d_tmp, tmp2, d_khi[0], d_khi[1], d_loc[0], d_loc[1], n, 0, B - 64, stream));
        }
        HIPCHK(jk_stream_wait(stream));
    }
    {
        // ---- header (field set of a real jellyfish 2.3.0 header; values that describe OUR file)
        std::string h = "{\"alignment\":8,\"canonical\":true,\"cmdline\":[";
        for (int i = 0; i < n_cmd; ++i) { if (i) h += ","; h += json_str(cmdline[i]); }
        h += "],\"counter_len\":4,\"exe_path\":\"libjasper_hip.so\",\"format\":\"binary/sorted\",\"hostname\":\"\",\"key_len\":" + std::to_string(B);
        h += ",\"matrix1\":{\"c\":" + std::to_string(B) + ",\"identity\":true,\"r\":" + std::to_string(r) + "},\"max_reprobe\":126,\"pwd\":\"\",\"reprobes\":[";
        for (int i = 0; i <= 126; ++i) { if (i) h += ","; h += std::to_string(i == 0 ? 1 : i * (i + 1) / 2); }
        char tbuf[64] = "";
        {
            const time_t now = time(nullptr);
            struct tm tmv;
            if (localtime_r(&now, &tmv)) strftime(tbuf, sizeof tbuf, "%a %b %e %H:%M:%S %Y", &tmv);
        }
        h += "],\"size\":" + std::to_string(1ull << r) + ",\"time\":" + json_str(tbuf) + ",\"val_len\":7}";
        size_t hlen = h.size();
        const size_t rem = (9 + hlen) % 8;
        const size_t pad = rem ? 8 - rem : 0;
        hlen += pad;
        f = fopen(path, "wb");
        if (!f) { err = std::string("cannot create ") + path; rc = -1; goto done; }
        if (what != 1) {
            char digits[16];
            snprintf(digits, sizeof digits, "%09zu", hlen);
            if (fwrite(digits, 1, 9, f) != 9 || fwrite(h.data(), 1, h.size(), f) != h.size()) { err = "write error"; rc = -1; goto done; }
            const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pad && fwrite(zeros, 1, pad, f) != pad) { err = "write error"; rc = -1; goto done; }
        }
    }
    if (what != 2) {
        // ---- records: formatted on the device, copied through two pinned buffers; the fwrite of a block overlaps the
        //      formatting + copy of the next one
        const uint64_t BLK = 1u << 22;
        const size_t rec = (size_t)kb + 4;
        const unsigned long long *src_key = B > 64 ? d_khi[1] : d_klo[1];
        const HiCnt *src_pay = B > 64 ? reinterpret_cast<const HiCnt *>(d_loc[1]) : d_hic[1];
        HIPCHK(hipMalloc((void **)&d_fmt, 2 * BLK * rec));
        HIPCHK(hipHostMalloc((void **)&h_fmt, 2 * BLK * rec, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&ev_fmt[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_fmt[1], hipEventDisableTiming));
        const uint64_t nblk = (n + BLK - 1) / BLK;
        auto issue = [&](uint64_t b) -> hipError_t {
            const uint64_t pos = b * BLK, m = std::min<uint64_t>(BLK, n - pos);
            uint8_t *d = d_fmt + (b & 1) * BLK * rec;
            hipLaunchKernelGGL(format_records_kernel, dim3(2048), dim3(256), 0, stream, src_key, src_pay, pos, m, B, r, kb, d);
            hipError_t e = hipMemcpyAsync(h_fmt + (b & 1) * BLK * rec, d, m * rec, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipEventRecord(ev_fmt[b & 1], stream);
            return e;
        };
        if (nblk) HIPCHK(issue(0));
        for (uint64_t b = 0; b < nblk; ++b) {
            if (g_cancel.load(std::memory_order_relaxed)) { err = "cancelled"; rc = -1; goto done; }
            if (b + 1 < nblk) HIPCHK(issue(b + 1));
            for (;;) {                                   // poll: hipEventSynchronize may sleep for milliseconds
                const hipError_t q = hipEventQuery(ev_fmt[b & 1]);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { err = std::string("event: ") + hipGetErrorString(q); rc = -1; goto done; }
            }
            const uint64_t m = std::min<uint64_t>(BLK, n - b * BLK);
            if (fwrite(h_fmt + (b & 1) * BLK * rec, rec, m, f) != m) { err = "write error"; rc = -1; goto done; }
        }
    }
done:
    if (f && fclose(f) != 0 && rc == 0) { err = "write error"; rc = -1; }
    (void)jk_stream_wait(stream);
    if (d_khi[1]) (void)hipFree(d_khi[1]);
    if (d_loc[1]) (void)hipFree(d_loc[1]);
    for (int i = 0; i < 2; ++i) { if (d_klo[i]) (void)hipFree(d_klo[i]); if (d_hic[i]) (void)hipFree(d_hic[i]); }
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_counter) (void)hipFree(d_counter);
    if (d_fmt) (void)hipFree(d_fmt);
    if (h_fmt) (void)hipHostFree(h_fmt);
    for (int i = 0; i < 2; ++i) if (ev_fmt[i]) (void)hipEventDestroy(ev_fmt[i]);
    return rc;
}

}  // namespace jk

namespace jk {

// records of a binary/sorted file (after its header) -> table: raw blocks through a pinned buffer, unpacked on the device
int Table::load_jf_records(const char *path, uint64_t data_offset, uint64_t n_records, int key_len_bits, int counter_len, std::string &err) {
    int rc = 0;
    histo_cached = false;
    if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return -1; }
    if (materialize(err)) return -1;
    const int kb = (key_len_bits + 7) / 8, cl = counter_len;
    const size_t rec = (size_t)kb + (size_t)cl;
    const uint64_t BLK = 1u << 22;
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("Can't open file '") + path + "'"; return -1; }
    uint8_t *h_raw = nullptr, *d_raw = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    uint64_t left = n_records;
    int b = 0;
    if (fseek(f, (long)data_offset, SEEK_SET) != 0) { err = "truncated Jellyfish database"; rc = -1; goto done; }
    if (ensure_capacity(n_records, err)) { rc = -1; goto done; }
    HIPCHK(hipHostMalloc((void **)&h_raw, 2 * BLK * rec, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&d_raw, 2 * BLK * rec));
    HIPCHK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    for (; left; b ^= 1) {
        const uint64_t m = std::min<uint64_t>(left, BLK);
        if (ev[b]) {                                   // the copy that last used this half must have left the pinned buffer
            for (;;) { const hipError_t q = hipEventQuery(ev[b]); if (q == hipSuccess) break; if (q != hipErrorNotReady) { err = "event query failed"; rc = -1; goto done; } }
        }
        if (fread(h_raw + (size_t)b * BLK * rec, rec, m, f) != m) { err = "truncated Jellyfish database"; rc = -1; goto done; }
        HIPCHK(hipMemcpyAsync(d_raw + (size_t)b * BLK * rec, h_raw + (size_t)b * BLK * rec, m * rec, hipMemcpyHostToDevice, stream));
        HIPCHK(hipEventRecord(ev[b], stream));
        hipLaunchKernelGGL(add_jf_records_kernel, dim3(2048), dim3(256), 0, stream, d_raw + (size_t)b * BLK * rec, m, kb, cl, d);
        HIPCHK(hipGetLastError());
        left -= m;
    }
    rc = after_batch(err);
done:
    (void)jk_stream_wait(stream);
    fclose(f);
    if (h_raw) (void)hipHostFree(h_raw);
    if (d_raw) (void)hipFree(d_raw);
    for (int i = 0; i < 2; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
    return rc;
}

}  // namespace jk
