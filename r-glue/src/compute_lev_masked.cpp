// Shim for .Call compute_lev_masked (body it replaces: /root/reference/src/compute_lev_masked.cpp:13-64).
#include "sarlacc.h"
#include "flatten.h"

SEXP compute_lev_masked(SEXP sequences) {
    BEGIN_RCPP
    Flat s = flatten(sequences, true);
    const int64_t n = s.n();
    Rcpp::NumericVector out(n > 0 ? n * (n - 1) / 2 : 0);
    std::vector<double> none(1);
    SL_CHECK(sarlacc_compute_lev_masked(s.chars.data(), s.off.data(), n, out.size() ? out.begin() : none.data()));
    return out;
    END_RCPP
}
