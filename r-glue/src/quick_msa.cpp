// Shim for .Call quick_msa (body it replaces: /root/reference/src/quick_msa.cpp:15-80).  With it the
// package no longer includes SeqAn: drop RSeqAn from LinkingTo (DESCRIPTION:20).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

#include <algorithm>

SEXP quick_msa(SEXP groupings, SEXP sequences, SEXP match, SEXP mismatch, SEXP gapExtension, SEXP gapOpening, SEXP bandwidth) {
    BEGIN_RCPP
    Csr g = csr_from_list(groupings);
    Flat s = flatten(sequences, true);
    const int64_t ng = g.n(), nseq = s.n();
    const double ma = check_numeric_scalar(match, "match score"), mm = check_numeric_scalar(mismatch, "mismatch score");
    const double gx = check_numeric_scalar(gapExtension, "gap extension score");
    const double go = check_numeric_scalar(gapOpening, "gap opening score");
    const int bw = check_integer_scalar(bandwidth, "bandwidth");

    // One pass in the common case: gapped rows are rarely more than 1.5x the reads they hold.  A call
    // whose buffer is too small fails after filling out_off[] with the exact sizes (the alignments are
    // what costs, so a separate sizing call with out == NULL would double the work every time).
    std::vector<int32_t> width(std::max<int64_t>(ng, 1));
    std::vector<int64_t> ooff(ng + 1);
    int64_t member_bases = 0;
    for (int64_t k = 0; k < g.off.back(); ++k) {
        const int32_t id = g.val[k];
        if (id >= 1 && id <= nseq) member_bases += s.off[id] - s.off[id - 1];   // bad ids are reported by the library
    }
    std::vector<char> rows((size_t)(member_bases + member_bases / 2) + 1024);
    if (sarlacc_quick_msa(g.off.data(), g.val.data(), ng, s.chars.data(), s.off.data(), nseq, ma, mm, gx, go, bw,
                          width.data(), ooff.data(), rows.data(), (int64_t)rows.size())) {
        if ((int64_t)rows.size() >= ooff[ng]) throw std::runtime_error(sarlacc_last_error());   // a real error
        rows.resize((size_t)ooff[ng]);
        SL_CHECK(sarlacc_quick_msa(g.off.data(), g.val.data(), ng, s.chars.data(), s.off.data(), nseq, ma, mm, gx, go, bw,
                                   width.data(), ooff.data(), rows.data(), (int64_t)rows.size()));
    }
    Rcpp::List out(ng);
    for (int64_t k = 0; k < ng; ++k) {                   // rows of group k: equal width, back to back
        const int64_t m = g.off[k + 1] - g.off[k];
        Rcpp::StringVector v(m);
        for (int64_t r = 0; r < m; ++r) v[r] = std::string(rows.data() + ooff[k] + r * width[k], (size_t)width[k]);
        out[k] = v;
    }
    return out;
    END_RCPP
}
