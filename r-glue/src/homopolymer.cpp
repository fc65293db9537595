// Shims for .Call find_homopolymers and match_homopolymers
// (bodies they replace: /root/reference/src/homopolymer.cpp:87-134 and :141-209; the rle_walker above them goes away).
#include "sarlacc.h"
#include "flatten.h"

#include <algorithm>

static Rcpp::StringVector single_chars(const std::vector<char>& c, int64_t n) {
    Rcpp::StringVector out(n);
    for (int64_t k = 0; k < n; ++k) out[k] = std::string(1, c[k]);
    return out;
}

SEXP find_homopolymers(SEXP sequences) {
    BEGIN_RCPP
    Flat s = flatten(sequences, true);
    int64_t count = 0;
    std::vector<int32_t> idx(1024), pos(1024), size(1024);
    std::vector<char> base(1024);
    SL_CHECK(sarlacc_find_homopolymers(s.chars.data(), s.off.data(), s.n(), idx.data(), pos.data(), size.data(), base.data(), (int64_t)idx.size(), &count));
    if (count > (int64_t)idx.size()) {                     // sizing protocol: second call with the exact size
        idx.resize(count); pos.resize(count); size.resize(count); base.resize(count);
        SL_CHECK(sarlacc_find_homopolymers(s.chars.data(), s.off.data(), s.n(), idx.data(), pos.data(), size.data(), base.data(), count, &count));
    }
    return Rcpp::List::create(Rcpp::IntegerVector(idx.begin(), idx.begin() + count), Rcpp::IntegerVector(pos.begin(), pos.begin() + count),
                              Rcpp::IntegerVector(size.begin(), size.begin() + count), single_chars(base, count));
    END_RCPP
}

SEXP match_homopolymers(SEXP ref_align, SEXP read_align) {
    BEGIN_RCPP
    Flat r = flatten(ref_align, true), q = flatten(read_align, true);
    int64_t count = 0;
    std::vector<int32_t> idx(1024), pos(1024), rlen(1024);
    SL_CHECK(sarlacc_match_homopolymers(r.chars.data(), r.off.data(), r.n(), q.chars.data(), q.off.data(), q.n(), idx.data(), pos.data(), rlen.data(), (int64_t)idx.size(), &count));
    if (count > (int64_t)idx.size()) {
        idx.resize(count); pos.resize(count); rlen.resize(count);
        SL_CHECK(sarlacc_match_homopolymers(r.chars.data(), r.off.data(), r.n(), q.chars.data(), q.off.data(), q.n(), idx.data(), pos.data(), rlen.data(), count, &count));
    }
    return Rcpp::List::create(Rcpp::IntegerVector(idx.begin(), idx.begin() + count), Rcpp::IntegerVector(pos.begin(), pos.begin() + count),
                              Rcpp::IntegerVector(rlen.begin(), rlen.begin() + count));
    END_RCPP
}
