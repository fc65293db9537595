// Shim for .Call mask_bad_bases (body it replaces: /root/reference/src/mask_bad_bases.cpp:10-52).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

SEXP mask_bad_bases(SEXP sequences, SEXP qualities, SEXP encoding, SEXP threshold) {
    BEGIN_RCPP
    Flat s, q;
    flatten_pair(sequences, qualities, s, q);
    Enc enc = flatten_encoding(encoding);
    const double maxerr = check_numeric_scalar(threshold, "quality threshold");
    std::vector<char> out((size_t)s.total() + 1);
    SL_CHECK(sarlacc_mask_bad_bases(s.chars.data(), s.off.data(), q.chars.data(), q.off.data(), s.n(),
                                    enc.err.data(), enc.names.data(), enc.n(), maxerr, out.data()));
    return strings_from_flat(out.data(), s.off.data(), s.n());   // same offsets as the input
    END_RCPP
}
