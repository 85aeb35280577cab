// asmio.hpp -- the assembly side of src/jasper.sh on the host, natively and by several threads:
//   src/jasper.sh:132       BS  = bytes of the non-header lines            -> AsmJob::sequence_bytes
//   src/jasper.sh:155       perl #1: chunk records ">name:offset" of <= BATCH_SIZE bases
//   src/jasper.sh:156       perl #2: batch files QUERY_FN.batch.N.fa
//   src/jasper.py:120-128   _iter{P-1}_<batch>.fixed.fa (60 columns)
//   src/jasper.sh:220       the join: ">name" + the contig on ONE line
// The sequence of every contig (line ends taken out) lies back to back in ONE host arena, so a chunk record is a slice of it:
// the polisher reads chunk text from there, the batch files are written from there, and nothing is copied into per-record
// strings.  Only the ORDINARY file takes this path (jasper_asm_open returns 1 otherwise and the caller applies the reference's
// line-by-line rules itself): no '\r', first byte '>', sequence lines of printable ASCII without blanks, header lines of
// printable ASCII / blanks / tabs, contig names unique.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

struct AsmContig {
    std::string name;           // first whitespace token of the header line, with its '>'
    size_t seq_off = 0;         // in the arena
    size_t seq_len = 0;
    size_t first_chunk = 0, n_chunks = 0;
};

struct AsmChunk {
    uint32_t contig = 0;
    uint32_t file = 0;
    uint64_t ci = 0;            // offset of the record in its contig (the number after the ':')
    uint64_t len = 0;
};

struct jasper_asm {
    std::string path;
    uint64_t sequence_bytes = 0;            // src/jasper.sh:132
    uint8_t *arena = nullptr;               // mmap'ed, arena_cap bytes
    size_t arena_len = 0, arena_cap = 0;
    std::vector<AsmContig> contigs;         // the contigs perl #1 emits records for (non-empty sequence), in input order
    // after jasper_asm_split
    uint64_t batch_size = 0;
    std::string prefix;                     // $QUERY_FN
    std::vector<AsmChunk> chunks;
    std::vector<size_t> file_first;         // file f = chunks [file_first[f], file_first[f + 1])
    std::vector<uint64_t> file_bytes;       // size of batch file f
    uint64_t own_bases = 0;                 // bases of the records of the batch files this job writes (all, or only_files): what it will polish
    std::thread writer;                     // the batch files are written beside whatever the caller does next
    bool writer_running = false;
    std::atomic<int> writer_failed{0};
    std::string writer_err;
    // polished text per chunk: a view into out_pinned (jasper_asm_take of a result whose text was still on the device), or `own`
    // (moved out of a host result; jasper_asm_put)
    struct Polished {
        const char *p = nullptr;
        size_t n = 0;
        std::string own;
        const char *data() const { return p ? p : own.data(); }
        size_t size() const { return p ? n : own.size(); }
    };
    std::vector<Polished> polished;
    std::vector<uint8_t> have;
    // jasper_asm_pin (capi.hip): the arena registered with the GPU runtime (chunk text is then copied to the device without a
    // staging copy), and ONE pinned buffer that receives the polished text
    bool arena_registered = false;
    char *out_pinned = nullptr;
    size_t out_cap = 0, out_used = 0;
    void (*gpu_release)(jasper_asm *) = nullptr;      // set by whoever pinned something: called by the destructor
    ~jasper_asm();
};

std::string &jasper_err_ref();          // capi.hip: the thread-local message behind jasper_last_error()
