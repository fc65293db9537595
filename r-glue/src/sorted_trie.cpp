// Shim for .Call fast_levdist_test (body it replaces: /root/reference/src/sorted_trie.cpp:304-337;
// the sorted_trie class above it in that file is no longer needed by any routine).
#include "sarlacc.h"
#include "utils.h"
#include "flatten.h"

#include <algorithm>

SEXP fast_levdist_test(SEXP input, SEXP limit, SEXP sorted) {
    BEGIN_RCPP
    Flat s = flatten(input, true);
    const int lim = check_integer_scalar(limit, "limit");
    check_logical_scalar(sorted, "sort specification");   // processing order only: never visible in the output
    const int64_t n = s.n();
    std::vector<int64_t> off(n + 1);
    std::vector<int32_t> nbr((size_t)std::max<int64_t>(32 * n, 1024));
    int64_t need = 0;
    SL_CHECK(sarlacc_fast_levdist_test(s.chars.data(), s.off.data(), n, lim, off.data(), nbr.data(), (int64_t)nbr.size(), &need));
    if (need > (int64_t)nbr.size()) {                      // sizing protocol: second call with the exact size
        nbr.resize((size_t)need);
        SL_CHECK(sarlacc_fast_levdist_test(s.chars.data(), s.off.data(), n, lim, off.data(), nbr.data(), (int64_t)nbr.size(), &need));
    }
    return list_from_csr(off.data(), nbr.data(), n);
    END_RCPP
}
