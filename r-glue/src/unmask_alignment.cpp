// Shim for .Call unmask_alignment (body it replaces: /root/reference/src/unmask_alignment.cpp:12-59).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

SEXP unmask_alignment(SEXP alignments, SEXP originals) {
    BEGIN_RCPP
    Flat a = flatten(alignments, true), o = flatten(originals, true);
    std::vector<char> out((size_t)a.total() + 1);
    SL_CHECK(sarlacc_unmask_alignment(a.chars.data(), a.off.data(), a.n(), o.chars.data(), o.off.data(), o.n(), out.data()));
    return strings_from_flat(out.data(), a.off.data(), a.n());   // same offsets as the alignment
    END_RCPP
}
