// ingest.hpp -- host-side FASTA/FASTQ reader feeding the counting kernel.
//
// Restates which windows `jellyfish count` sees (JF::include/jellyfish/mer_overlap_sequence_parser.hpp):
//   * the concatenation of all input files is ONE stream (src/jasper.sh:177 pipes `zcat -f $READS`), its
//     format is decided by the first byte: '>' FASTA, '@' FASTQ, anything else "Unsupported format" (:134-148)
//   * FASTA: header lines skipped, sequence lines concatenated with '\n' and trailing '\r' removed (:260-275)
//   * FASTQ: sequence lines up to the line starting with '+'; then as many quality characters as sequence
//     characters are skipped, over any number of lines; the next byte must be '@' or EOF, otherwise
//     "Invalid fastq sequence" (:290-307)
//   * records are separated by one 'N' so that no k-mer spans two reads (:175,205)
// Our own design: an incremental line state machine (the reference pulls 4 KB buffers through an istream).
#pragma once
#include <cstdint>
#include <functional>
#include <string>

namespace jk {

class FastxParser {
  public:
    // sink(bases, n) is called with whole records only (each followed by its 'N' separator)
    explicit FastxParser(std::function<int(const char *, size_t)> sink, size_t flush_bytes = 32u << 20)
        : sink_(std::move(sink)), flush_bytes_(flush_bytes) {}
    int feed(const char *data, size_t n);  // 0 ok, <0 error (message in error())
    int finish();
    // continue a stream whose beginning somebody else has parsed: `mode` 0 = unknown (decide from the first byte),
    // 1 = FASTA at a line start, 2 = FASTQ at a record start (ingest_gpu.hip hands over like this)
    void resume(int mode);
    const std::string &error() const { return err_; }
    uint64_t records() const { return records_; }

  private:
    enum Mode { UNKNOWN, FASTA, FASTQ };
    enum FqState { FQ_HEADER, FQ_SEQ, FQ_QUAL };
    int line(const char *p, size_t n, bool has_newline);
    int flush(bool force);
    std::function<int(const char *, size_t)> sink_;
    size_t flush_bytes_;
    std::string carry_;   // partial line
    std::string out_;     // parsed bases awaiting the sink
    std::string err_;
    Mode mode_ = UNKNOWN;
    FqState fq_ = FQ_HEADER;
    uint64_t seq_len_ = 0, qual_len_ = 0;
    uint64_t records_ = 0;
};

// reads plain or gzip files (zlib gzread passes plain data through, like `zcat -f`) as one stream
int parse_files(const char *const *paths, int n_paths, FastxParser &parser, std::string &err);

// Jellyfish "binary/sorted" database (JF::include/jellyfish/generic_file_header.hpp:88-143,
// JF::include/jellyfish/binary_dumper.hpp:36-40,103-108): 9 ASCII digits = header length, JSON header (NUL padded),
// then records [key: ceil(key_len/8) bytes LE][count: counter_len bytes LE] (unpacked on the device: jfwrite.hip).
struct JfHeader {
    int key_len = 0;        // bits = 2k
    int counter_len = 0;    // bytes
    bool canonical = false;
    uint64_t data_offset = 0, n_records = 0;
};
int jf_read_header(const char *path, JfHeader &h, std::string &err);

}  // namespace jk
