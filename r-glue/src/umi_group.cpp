// Shim for .Call umi_group (body it replaces: /root/reference/src/umi_group.cpp:14-116).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

#include <algorithm>

SEXP umi_group(SEXP umi1, SEXP thresh1, SEXP umi2, SEXP thresh2, SEXP pregroups) {
    BEGIN_RCPP
    Flat u1 = flatten(umi1, true);
    const bool two = umi2 != R_NilValue;
    Flat u2;
    if (two) u2 = flatten(umi2, true);
    if (two && u2.n() != u1.n()) throw std::runtime_error("'umi1' and 'umi2' should have the same length");
    const int t1 = check_integer_scalar(thresh1, "threshold 1");
    const int t2 = two ? check_integer_scalar(thresh2, "threshold 2") : t1;
    Csr g = csr_from_list(pregroups);
    const int64_t total = g.off.back();

    std::vector<int64_t> co(total + 2);
    std::vector<int32_t> cl(std::max<int64_t>(total, 1));
    int64_t ncl = 0;
    SL_CHECK(sarlacc_umi_group(u1.chars.data(), u1.off.data(), two ? u2.chars.data() : NULL, two ? u2.off.data() : NULL,
                               u1.n(), t1, t2, g.off.data(), g.val.data(), g.n(), &ncl, co.data(), cl.data()));
    // The ABI returns the clusters of all pre-groups as one flat list, i.e. what R/umiGroup.R:22
    // builds with unlist(out, recursive=FALSE); wrapped once so that unlist() there is the identity.
    return Rcpp::List::create(list_from_csr(co.data(), cl.data(), ncl));
    END_RCPP
}
