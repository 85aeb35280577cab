// polish_host.hpp -- host entry of the batch polisher (see polish_host.hip)
#pragma once
#include "polish.hpp"
#include <string>
#include <vector>

namespace jk {

struct PolishOut {
    std::vector<std::string> seqs;   // polished chunk texts, batch order (empty strings while the text is kept in HBM)
    std::vector<const uint8_t *> d_seqs;   // keep_on_device: where each polished chunk lies in the table's workspace
    std::vector<int64_t> d_lens;
    std::vector<FixRec> recs;        // ordered by chunk, pass, emission; index in chunk coordinates
    std::vector<std::string> aux;    // per chunk: bytes referenced by its 'x' records
    int64_t qv[4];                   // bad0, total0, badP, totalP   (src/jasper.py:107-111)
    std::vector<int64_t> qv_chunk;   // the same four per chunk record (4 * n_chunks): callers that group chunks into files
    uint64_t lookups;
    double seconds;                  // device time of all passes (HIP events on the table's stream)
    uint64_t n_segments;             // segments walked (all passes)
    uint64_t n_respeculated;         // chunks redone unsegmented because a segment's assumption did not hold
};

// returns 0, -1 (HIP error), -2 (capacity), -4 (the reference itself would exit 1)
// seqs[c] are host pointers, or device pointers when `device_in`.  With `keep_on_device` the polished text is not copied
// to the host: R.d_seqs/d_lens point into the table's workspace, valid until the next run_polish on the same table.
int run_polish(Table &T, int n_chunks, const char *const *seqs, const int64_t *lens, int solid_thre, int passes, int fix,
               PolishOut &R, std::string &err, bool device_in = false, bool keep_on_device = false, int roomy = 0);

}  // namespace jk
