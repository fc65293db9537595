// Shim for .Call barcode_align (body it replaces: /root/reference/src/barcode_align.cpp:10-44).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

SEXP barcode_align(SEXP barcodeseq, SEXP barcodequal, SEXP encoding, SEXP gapopen, SEXP gapext, SEXP reference) {
    BEGIN_RCPP
    const std::string ref = check_string(reference, "barcode sequence");
    const double go = check_numeric_scalar(gapopen, "gap opening penalty");
    const double ge = check_numeric_scalar(gapext, "gap extension penalty");
    Flat s, q;
    flatten_pair(barcodeseq, barcodequal, s, q);
    Enc enc = flatten_encoding(encoding);
    Rcpp::NumericVector scores(s.n());
    SL_CHECK(sarlacc_barcode_align(s.chars.data(), s.off.data(), q.chars.data(), q.off.data(), s.n(),
                                   enc.err.data(), enc.names.data(), enc.n(), go, ge,
                                   ref.data(), (int)ref.size(), scores.begin()));
    return scores;
    END_RCPP
}
