// Shims for .Call adaptor_align and adaptor_align_score_only
// (bodies they replace: /root/reference/src/adaptor_align.cpp:11-77 and :79-110).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

#include <algorithm>

SEXP adaptor_align(SEXP readseq, SEXP readqual, SEXP encoding, SEXP gapopen, SEXP gapext,
                   SEXP adaptor, SEXP sec_starts, SEXP sec_ends) {
    BEGIN_RCPP
    const std::string ad = check_string(adaptor, "adaptor sequence");
    const double go = check_numeric_scalar(gapopen, "gap opening penalty");
    const double ge = check_numeric_scalar(gapext, "gap extension penalty");
    Flat s, q;
    flatten_pair(readseq, readqual, s, q);
    Enc enc = flatten_encoding(encoding);
    Rcpp::IntegerVector ss(sec_starts), se(sec_ends);
    if (ss.size() != se.size()) throw std::runtime_error("section starts and ends should have the same length");
    const int64_t n = s.n();
    const int ns = ss.size();

    Rcpp::NumericVector scores(n);
    Rcpp::IntegerVector starts(n), ends(n);
    std::vector<int32_t> so((size_t)std::max(ns, 1) * std::max<int64_t>(n, 1)), sw(so.size());
    SL_CHECK(sarlacc_adaptor_align(s.chars.data(), s.off.data(), q.chars.data(), q.off.data(), n,
                                   enc.err.data(), enc.names.data(), enc.n(), go, ge,
                                   ad.data(), (int)ad.size(), ss.begin(), se.begin(), ns,
                                   scores.begin(), starts.begin(), ends.begin(), so.data(), sw.data()));
    Rcpp::List sl(ns), wl(ns);
    for (int k = 0; k < ns; ++k) {
        sl[k] = Rcpp::IntegerVector(so.begin() + k * n, so.begin() + (k + 1) * n);
        wl[k] = Rcpp::IntegerVector(sw.begin() + k * n, sw.begin() + (k + 1) * n);
    }
    return Rcpp::List::create(scores, starts, ends, sl, wl);
    END_RCPP
}

SEXP adaptor_align_score_only(SEXP readseq, SEXP readqual, SEXP encoding, SEXP gapopen, SEXP gapext, SEXP adaptor) {
    BEGIN_RCPP
    const std::string ad = check_string(adaptor, "adaptor sequence");
    const double go = check_numeric_scalar(gapopen, "gap opening penalty");
    const double ge = check_numeric_scalar(gapext, "gap extension penalty");
    Flat s, q;
    flatten_pair(readseq, readqual, s, q);
    Enc enc = flatten_encoding(encoding);
    Rcpp::NumericVector scores(s.n());
    SL_CHECK(sarlacc_adaptor_align_score_only(s.chars.data(), s.off.data(), q.chars.data(), q.off.data(), s.n(),
                                              enc.err.data(), enc.names.data(), enc.n(), go, ge,
                                              ad.data(), (int)ad.size(), scores.begin()));
    return scores;
    END_RCPP
}
