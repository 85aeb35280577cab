// ingest.cpp -- see ingest.hpp
#include "ingest.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
        if (n == 0) { err_ = "Unsupported format"; return -3; }   // an empty first line: the reference's peek() sees '\n' (:134-148)
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

void FastxParser::resume(int mode) {
    mode_ = mode == 1 ? FASTA : mode == 2 ? FASTQ : UNKNOWN;
    fq_ = FQ_HEADER;
    seq_len_ = qual_len_ = 0;
    carry_.clear();
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


// The header is a JSON object (JF::include/jellyfish/generic_file_header.hpp:88-111) that also carries strings the user
// controls (cmdline, pwd, exe_path, hostname): it is walked token by token, and only TOP-LEVEL members count -- a path or an
// argument that contains "key_len" or "binary/sorted" must not be taken for the member of that name.
namespace {
struct JsonTop {
    const std::string &j;
    size_t p = 0;
    explicit JsonTop(const std::string &s) : j(s) {}
    void ws() { while (p < j.size() && (j[p] == ' ' || j[p] == '\n' || j[p] == '\t' || j[p] == '\r')) ++p; }
    bool str(std::string &out) {
        if (p >= j.size() || j[p] != '"') return false;
        ++p;
        out.clear();
        while (p < j.size() && j[p] != '"') {
            if (j[p] == '\\') { if (p + 1 >= j.size()) return false; out.push_back(j[p + 1]); p += 2; }
            else out.push_back(j[p++]);
        }
        if (p >= j.size()) return false;
        ++p;
        return true;
    }
    // skips any value; scalars are returned as text in `scalar` (strings unquoted)
    bool value(std::string &scalar) {
        ws();
        scalar.clear();
        if (p >= j.size()) return false;
        if (j[p] == '"') return str(scalar);
        if (j[p] == '{' || j[p] == '[') {
            const char open = j[p], close = open == '{' ? '}' : ']';
            ++p;
            for (;;) {
                ws();
                if (p >= j.size()) return false;
                if (j[p] == close) { ++p; return true; }
                if (j[p] == ',') { ++p; continue; }
                std::string tmp;
                if (open == '{') { if (!str(tmp)) return false; ws(); if (p >= j.size() || j[p] != ':') return false; ++p; }
                if (!value(tmp)) return false;
            }
        }
        const size_t a = p;
        while (p < j.size() && j[p] != ',' && j[p] != '}' && j[p] != ']' && j[p] != ' ' && j[p] != '\n' && j[p] != '\t' && j[p] != '\r') ++p;
        scalar = j.substr(a, p - a);
        return p > a;
    }
};
}  // namespace

int jf_read_header(const char *path, JfHeader &h, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("Can't open file '") + path + "'"; return -1; }     // JF::swig/mer_file.i:21
    char digits[10] = {0};
    if (fread(digits, 1, 9, f) != 9) { fclose(f); err = "Unsupported format"; return -3; }
    for (int i = 0; i < 9; ++i) if (digits[i] < '0' || digits[i] > '9') { fclose(f); err = "Unsupported format"; return -3; }
    const long hlen = strtol(digits, nullptr, 10);
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    if (hlen <= 0 || 9 + hlen > fsize) { fclose(f); err = "Unsupported format"; return -3; }      // (checked before anything of that size is allocated)
    fseek(f, 9, SEEK_SET);
    std::string j((size_t)hlen, '\0');
    if (fread(&j[0], 1, (size_t)hlen, f) != (size_t)hlen) { fclose(f); err = "Unsupported format"; return -3; }
    fclose(f);
    std::string format, key_len, counter_len, canonical;
    {
        JsonTop t(j);
        t.ws();
        if (t.p >= j.size() || j[t.p] != '{') { err = "Unsupported format"; return -3; }
        ++t.p;
        for (;;) {
            t.ws();
            if (t.p >= j.size()) { err = "Unsupported format"; return -3; }
            if (j[t.p] == '}') break;
            if (j[t.p] == ',') { ++t.p; continue; }
            std::string name, val;
            if (!t.str(name)) { err = "Unsupported format"; return -3; }
            t.ws();
            if (t.p >= j.size() || j[t.p] != ':') { err = "Unsupported format"; return -3; }
            ++t.p;
            if (!t.value(val)) { err = "Unsupported format"; return -3; }
            if (name == "format") format = val;
            else if (name == "key_len") key_len = val;
            else if (name == "counter_len") counter_len = val;
            else if (name == "canonical") canonical = val;
        }
    }
    if (format != "binary/sorted") { err = "Unsupported format"; return -3; }           // JF::swig/mer_file.i:34
    char *e1 = nullptr, *e2 = nullptr;
    const long kl = strtol(key_len.c_str(), &e1, 10), cl = strtol(counter_len.c_str(), &e2, 10);
    if (key_len.empty() || counter_len.empty() || *e1 || *e2 || kl <= 0 || kl > 128 || (kl & 1) || cl <= 0 || cl > 8) { err = "Unsupported format"; return -3; }
    h.key_len = (int)kl;
    h.counter_len = (int)cl;
    h.canonical = canonical == "true";
    h.data_offset = 9 + (uint64_t)hlen;
    const uint64_t rec = (uint64_t)((kl + 7) / 8) + (uint64_t)cl;
    if (((uint64_t)fsize - h.data_offset) % rec != 0) { err = "Unsupported format"; return -3; }      // a whole number of records
    h.n_records = ((uint64_t)fsize - h.data_offset) / rec;
    return 0;
}

}  // namespace jk
