// ingest.cpp -- see ingest.hpp
#include "ingest.hpp"
#include <cstring>
#include <vector>
#include <zlib.h>

namespace jk {

int FastxParser::flush(bool force) {
    if (out_.empty()) return 0;
    if (!force && out_.size() < flush_bytes_) return 0;
    int rc = sink_(out_.data(), out_.size());
    out_.clear();
    return rc;
}

// one line without its '\n'
int FastxParser::line(const char *p, size_t n, bool /*has_newline*/) {
    while (n && p[0] == '\r') { ++p; --n; }          // skip_newlines() eats leading '\r' too
    while (n && p[n - 1] == '\r') --n;               // trailing '\r' of DOS files
    if (mode_ == UNKNOWN) {
        if (n == 0) return 0;                         // (an empty first line: peek() would see '\n' -> unsupported)
        if (p[0] == '>') { mode_ = FASTA; ++records_; return 0; }
        if (p[0] == '@') { mode_ = FASTQ; fq_ = FQ_SEQ; seq_len_ = 0; ++records_; return 0; }
        err_ = "Unsupported format";
        return -3;
    }
    if (n == 0) return 0;
    if (mode_ == FASTA) {
        if (p[0] == '>') {
            out_.push_back('N');
            ++records_;
            return flush(false);
        }
        out_.append(p, n);
        return 0;
    }
    // FASTQ
    switch (fq_) {
    case FQ_HEADER:
        if (p[0] != '@') { err_ = "Invalid fastq sequence"; return -3; }
        out_.push_back('N');
        ++records_;
        fq_ = FQ_SEQ;
        seq_len_ = 0;
        return flush(false);
    case FQ_SEQ:
        if (p[0] == '+') {
            qual_len_ = 0;
            fq_ = seq_len_ ? FQ_QUAL : FQ_HEADER;
            return 0;
        }
        out_.append(p, n);
        seq_len_ += n;
        return 0;
    case FQ_QUAL:
        qual_len_ += n;
        if (qual_len_ == seq_len_) { fq_ = FQ_HEADER; return 0; }
        if (qual_len_ > seq_len_) { err_ = "Invalid fastq sequence"; return -3; }
        return 0;
    }
    return 0;
}

int FastxParser::feed(const char *data, size_t n) {
    size_t i = 0;
    while (i < n) {
        const char *nl = (const char *)memchr(data + i, '\n', n - i);
        if (!nl) {
            carry_.append(data + i, n - i);
            return 0;
        }
        size_t len = (size_t)(nl - (data + i));
        int rc;
        if (!carry_.empty()) {
            carry_.append(data + i, len);
            rc = line(carry_.data(), carry_.size(), true);
            carry_.clear();
        } else {
            rc = line(data + i, len, true);
        }
        if (rc) return rc;
        i += len + 1;
    }
    return 0;
}

int FastxParser::finish() {
    if (!carry_.empty()) {
        int rc = line(carry_.data(), carry_.size(), false);
        carry_.clear();
        if (rc) return rc;
    }
    if (mode_ == FASTQ && fq_ == FQ_QUAL) { err_ = "Invalid fastq sequence"; return -3; }
    return flush(true);
}

int parse_files(const char *const *paths, int n_paths, FastxParser &parser, std::string &err) {
    std::vector<char> buf(8u << 20);
    for (int f = 0; f < n_paths; ++f) {
        gzFile g = gzopen(paths[f], "rb");
        if (!g) { err = std::string("cannot open ") + paths[f]; return -1; }
        gzbuffer(g, 1u << 20);
        for (;;) {
            int got = gzread(g, buf.data(), (unsigned)buf.size());
            if (got < 0) { err = std::string("read error in ") + paths[f]; gzclose(g); return -1; }
            if (got == 0) break;
            int rc = parser.feed(buf.data(), (size_t)got);
            if (rc) { err = parser.error(); gzclose(g); return rc; }
        }
        gzclose(g);
    }
    int rc = parser.finish();
    if (rc) err = parser.error();
    return rc;
}

}  // namespace jk
