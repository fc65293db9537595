// Shim for .Call find_errors (body it replaces: /root/reference/src/find_errors.cpp:9-121).
#include "sarlacc.h"
#include "flatten.h"

#include <algorithm>

SEXP find_errors(SEXP ref_align, SEXP read_align) {
    BEGIN_RCPP
    Flat r = flatten(ref_align, true), q = flatten(read_align, true);
    // the first reference string bounds the number of bases; insertions: two-call sizing
    const int64_t cap = r.n() ? r.off[1] - r.off[0] : 0;
    std::vector<char> bases(std::max<int64_t>(cap, 1));
    Rcpp::IntegerVector a(cap), c(cap), g(cap), t(cap), d(cap);
    std::vector<int32_t> ipos(1024), ilen(1024);
    int64_t sl = 0, nins = 0;
    SL_CHECK(sarlacc_find_errors(r.chars.data(), r.off.data(), r.n(), q.chars.data(), q.off.data(), q.n(), &sl, bases.data(), a.begin(), c.begin(),
                                 g.begin(), t.begin(), d.begin(), cap, ipos.data(), ilen.data(), (int64_t)ipos.size(), &nins));
    if (nins > (int64_t)ipos.size()) {
        ipos.resize(nins); ilen.resize(nins);
        SL_CHECK(sarlacc_find_errors(r.chars.data(), r.off.data(), r.n(), q.chars.data(), q.off.data(), q.n(), &sl, bases.data(), a.begin(), c.begin(),
                                     g.begin(), t.begin(), d.begin(), cap, ipos.data(), ilen.data(), nins, &nins));
    }
    Rcpp::StringVector b(sl);
    for (int64_t k = 0; k < sl; ++k) b[k] = std::string(1, bases[k]);
    auto head = [&](Rcpp::IntegerVector v) { return Rcpp::IntegerVector(v.begin(), v.begin() + sl); };
    return Rcpp::List::create(b, head(a), head(c), head(g), head(t), head(d), Rcpp::IntegerVector(ipos.begin(), ipos.begin() + nins),
                              Rcpp::IntegerVector(ilen.begin(), ilen.begin() + nins));
    END_RCPP
}
