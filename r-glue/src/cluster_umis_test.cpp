// Shim for .Call cluster_umis_test (body it replaces: /root/reference/src/cluster_umis_test.cpp:8-30).
#include "sarlacc.h"
#include "flatten.h"

#include <algorithm>

SEXP cluster_umis_test(SEXP links) {
    BEGIN_RCPP
    Csr l = csr_from_list(links);
    const int64_t n = l.n();
    std::vector<int64_t> co(n + 2);
    std::vector<int32_t> cl(std::max<int64_t>(n, 1));
    int64_t ncl = 0;
    SL_CHECK(sarlacc_cluster_umis_test(l.off.data(), l.val.data(), n, &ncl, co.data(), cl.data()));
    return list_from_csr(co.data(), cl.data(), ncl);
    END_RCPP
}
