// Shims for the four consensus routines (bodies they replace: /root/reference/src/create_consensus.cpp
// :137-148 create_consensus_basic, :150-170 create_consensus_basic_loop, :274-285 create_consensus_quality,
// :287-308 create_consensus_quality_loop).  The single-alignment (test) forms are the loop forms
// on a list of length one, returning the natural-log errors instead of the Phred string.
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

namespace {

struct Consensus {
    std::vector<char> cons, phred;
    std::vector<int64_t> off;
    std::vector<double> lerr;
};

Consensus run_basic(const FlatList& a, double mincov, double pseudo, bool want_lerr) {
    Consensus c;
    const int64_t cap = a.rows.total() + 1;              // an alignment's width bounds its consensus
    c.cons.resize(cap); c.phred.resize(cap); c.off.resize(a.n() + 1);
    if (want_lerr) c.lerr.resize(cap);
    SL_CHECK(sarlacc_create_consensus_basic_loop(a.rows.chars.data(), a.rows.off.data(), a.ranges.data(), a.n(),
                                                 mincov, pseudo, c.cons.data(), c.phred.data(), c.off.data(),
                                                 want_lerr ? c.lerr.data() : NULL));
    return c;
}

Consensus run_quality(const FlatList& a, const FlatList& q, double mincov, const Enc& enc, bool want_lerr) {
    Consensus c;
    const int64_t cap = a.rows.total() + 1;
    c.cons.resize(cap); c.phred.resize(cap); c.off.resize(a.n() + 1);
    if (want_lerr) c.lerr.resize(cap);
    SL_CHECK(sarlacc_create_consensus_quality_loop(a.rows.chars.data(), a.rows.off.data(), a.ranges.data(), a.n(),
                                                   q.rows.chars.data(), q.rows.off.data(), q.ranges.data(), mincov,
                                                   enc.err.data(), enc.names.data(), enc.n(),
                                                   c.cons.data(), c.phred.data(), c.off.data(),
                                                   want_lerr ? c.lerr.data() : NULL));
    return c;
}

}

SEXP create_consensus_basic(SEXP alignments, SEXP min_cov, SEXP pseudo_count) {
    BEGIN_RCPP
    const double mincov = check_numeric_scalar(min_cov, "minimum coverage");
    const double pseudo = check_numeric_scalar(pseudo_count, "pseudo count");
    Consensus c = run_basic(flatten_single(alignments, true), mincov, pseudo, true);
    return Rcpp::List::create(Rcpp::String(std::string(c.cons.data(), (size_t)c.off[1])),
                              Rcpp::NumericVector(c.lerr.begin(), c.lerr.begin() + c.off[1]));
    END_RCPP
}

SEXP create_consensus_basic_loop(SEXP alignments, SEXP min_cov, SEXP pseudo_count) {
    BEGIN_RCPP
    const double mincov = check_numeric_scalar(min_cov, "minimum coverage");
    const double pseudo = check_numeric_scalar(pseudo_count, "pseudo count");
    FlatList a = flatten_list(Rcpp::List(alignments), true);
    Consensus c = run_basic(a, mincov, pseudo, false);
    return Rcpp::List::create(strings_from_flat(c.cons.data(), c.off.data(), a.n()),
                              strings_from_flat(c.phred.data(), c.off.data(), a.n()));
    END_RCPP
}

SEXP create_consensus_quality(SEXP alignments, SEXP min_cov, SEXP qualities, SEXP encoding) {
    BEGIN_RCPP
    const double mincov = check_numeric_scalar(min_cov, "minimum coverage");
    Enc enc = flatten_encoding(encoding);
    Consensus c = run_quality(flatten_single(alignments, true), flatten_single(qualities, false), mincov, enc, true);
    return Rcpp::List::create(Rcpp::String(std::string(c.cons.data(), (size_t)c.off[1])),
                              Rcpp::NumericVector(c.lerr.begin(), c.lerr.begin() + c.off[1]));
    END_RCPP
}

SEXP create_consensus_quality_loop(SEXP alignments, SEXP min_cov, SEXP qualities, SEXP encoding) {
    BEGIN_RCPP
    Rcpp::List al(alignments), ql(qualities);
    if (al.size() != ql.size()) throw std::runtime_error("alignments and qualities should have the same length");
    const double mincov = check_numeric_scalar(min_cov, "minimum coverage");
    Enc enc = flatten_encoding(encoding);
    FlatList a = flatten_list(al, true), q = flatten_list(ql, false);
    Consensus c = run_quality(a, q, mincov, enc, false);
    return Rcpp::List::create(strings_from_flat(c.cons.data(), c.off.data(), a.n()),
                              strings_from_flat(c.phred.data(), c.off.data(), a.n()));
    END_RCPP
}
