// Shim for .Call general_align (body it replaces: /root/reference/src/general_align.cpp:10-62).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

SEXP general_align(SEXP inputseq, SEXP inputqual, SEXP encoding, SEXP gapopen, SEXP gapext, SEXP reference, SEXP edit_only) {
    BEGIN_RCPP
    const std::string ref = check_string(reference, "reference sequence");
    const double go = check_numeric_scalar(gapopen, "gap opening penalty");
    const double ge = check_numeric_scalar(gapext, "gap extension penalty");
    Flat s, q;
    flatten_pair(inputseq, inputqual, s, q);
    const bool only_edit = check_logical_scalar(edit_only, "edit-only specification");
    Enc enc = flatten_encoding(encoding);
    const int64_t n = s.n();

    Rcpp::NumericVector scores(n);
    Rcpp::IntegerVector edits(n);
    // a gapped pair is never longer than read + reference
    const int64_t cap = s.total() + n * (int64_t)ref.size() + 1;
    std::vector<char> aref(only_edit ? 1 : cap), aquery(only_edit ? 1 : cap);
    std::vector<int64_t> aoff(n + 1);
    SL_CHECK(sarlacc_general_align(s.chars.data(), s.off.data(), q.chars.data(), q.off.data(), n,
                                   enc.err.data(), enc.names.data(), enc.n(), go, ge,
                                   ref.data(), (int)ref.size(), only_edit ? 1 : 0,
                                   scores.begin(), edits.begin(),
                                   only_edit ? NULL : aref.data(), only_edit ? NULL : aquery.data(), aoff.data(), cap));
    if (only_edit) return Rcpp::List::create(scores, edits, Rcpp::StringVector(0), Rcpp::StringVector(0));
    return Rcpp::List::create(scores, edits, strings_from_flat(aref.data(), aoff.data(), n),
                              strings_from_flat(aquery.data(), aoff.data(), n));
    END_RCPP
}
