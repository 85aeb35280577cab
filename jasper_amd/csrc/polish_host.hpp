// polish_host.hpp -- host entry of the batch polisher (see polish_host.hip)
#pragma once
#include "polish.hpp"
#include <string>
#include <vector>

namespace jk {

struct PolishOut {
    std::vector<std::string> seqs;   // polished chunk texts, batch order
    std::vector<FixRec> recs;        // ordered by chunk, pass, emission; index in chunk coordinates
    std::vector<std::string> aux;    // per chunk: bytes referenced by its 'x' records
    int64_t qv[4];                   // bad0, total0, badP, totalP   (src/jasper.py:107-111)
    uint64_t lookups;
    double seconds;                  // device time of all passes (HIP events on the table's stream)
    uint64_t n_segments;             // segments walked (all passes)
    uint64_t n_respeculated;         // chunks redone unsegmented because a segment's assumption did not hold
};

// returns 0, -1 (HIP error), -2 (capacity), -4 (the reference itself would exit 1)
int run_polish(Table &T, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre, int passes, int fix,
               PolishOut &R, std::string &err);

}  // namespace jk
