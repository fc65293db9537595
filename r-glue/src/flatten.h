// flatten.h -- SEXP <-> flat arrays of the C ABI (include/sarlacc_amd.h).  Takes the place of
// /root/reference/src/DNA_input.{h,cpp} and src/quality_encoding.{h,cpp} in the shimmed package:
// the shims only marshal, every check that needs the data itself happens behind the ABI.
// Not built in this repository (no R / Rcpp / Biostrings in the build image); tests/test_rglue.py checks every
// sarlacc_* call below against the header and runs every shim through g++ -fsyntax-only against declaration-only
// stand-ins for the Rcpp / Biostrings headers (tests/rglue_stubs) and the real include/sarlacc_amd.h.
#ifndef SARLACC_FLATTEN_H
#define SARLACC_FLATTEN_H

#include "Rcpp.h"
extern "C" {
#include "Biostrings_interface.h"
}
#include "sarlacc_amd.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#define SL_CHECK(call) do { if (call) throw std::runtime_error(sarlacc_last_error()); } while (0)

// XStringSet / character vector -> concatenated characters + n+1 offsets
struct Flat {
    std::vector<char> chars;
    std::vector<int64_t> off;
    Flat() : off(1, 0) {}
    int64_t n() const { return (int64_t)off.size() - 1; }
    int64_t total() const { return off.back(); }
    void append(const Flat& o) {
        chars.insert(chars.end(), o.chars.begin(), o.chars.begin() + o.total());
        for (int64_t r = 1; r <= o.n(); ++r) off.push_back(off.back() + o.off[r] - o.off[r - 1]);
    }
};

inline Flat flatten(Rcpp::RObject x, bool decode_dna) {
    Flat f;
    if (x.isS4()) {                                   // DNAStringSet / BStringSet / PhredQuality
        XStringSet_holder h = hold_XStringSet(SEXP(x));
        const int n = get_length_from_XStringSet_holder(&h);
        for (int i = 0; i < n; ++i) {
            Chars_holder c = get_elt_from_XStringSet_holder(&h, i);
            for (int k = 0; k < c.length; ++k) f.chars.push_back(decode_dna ? DNAdecode(c.ptr[k]) : c.ptr[k]);
            f.off.push_back((int64_t)f.chars.size());
        }
    } else {                                          // plain character vector
        Rcpp::StringVector v(x);
        for (R_xlen_t i = 0; i < v.size(); ++i) {
            const char* s = CHAR(STRING_ELT(v, i));
            f.chars.insert(f.chars.end(), s, s + Rf_length(STRING_ELT(v, i)));
            f.off.push_back((int64_t)f.chars.size());
        }
    }
    f.chars.push_back(0);                             // keeps .data() valid for empty input
    return f;
}

// the two XStringSets of every alignment routine; the per-string length check is done by the library
inline void flatten_pair(SEXP seq, SEXP qual, Flat& s, Flat& q) {
    s = flatten(seq, true);
    q = flatten(qual, false);
    if (s.n() != q.n()) throw std::runtime_error("sequence and quality vectors should have the same length");
}

// named numeric `encoding` (R/utils.R:.create_encoding_vector) -> errors + one-character names
struct Enc {
    std::vector<double> err;
    std::string names;
    int n() const { return (int)err.size(); }
};

inline Enc flatten_encoding(SEXP encoding) {
    Rcpp::NumericVector e(encoding);
    Enc out;
    if (e.size() == 0 || !e.hasAttribute("names")) throw std::runtime_error("encoding vector must be non-empty and named");
    Rcpp::StringVector nm = e.names();
    for (R_xlen_t i = 0; i < e.size(); ++i) {
        std::string s = Rcpp::as<std::string>(nm[i]);
        if (s.size() != 1) throw std::runtime_error("names of encoding vector must be one character in length");
        out.names.push_back(s[0]);
        out.err.push_back(e[i]);
    }
    return out;
}

// list of integer vectors <-> CSR (1-based values kept as they are)
struct Csr {
    std::vector<int64_t> off;
    std::vector<int32_t> val;
    Csr() : off(1, 0) {}
    int64_t n() const { return (int64_t)off.size() - 1; }
};

inline Csr csr_from_list(Rcpp::List l) {
    Csr c;
    for (R_xlen_t g = 0; g < l.size(); ++g) {
        Rcpp::IntegerVector v(l[g]);
        c.val.insert(c.val.end(), v.begin(), v.end());
        c.off.push_back((int64_t)c.val.size());
    }
    c.val.push_back(0);
    return c;
}

inline Rcpp::List list_from_csr(const int64_t* off, const int32_t* val, int64_t n) {
    Rcpp::List out(n);
    for (int64_t k = 0; k < n; ++k) out[k] = Rcpp::IntegerVector(val + off[k], val + off[k + 1]);
    return out;
}

// concatenated strings + offsets -> character vector
inline Rcpp::StringVector strings_from_flat(const char* chars, const int64_t* off, int64_t n) {
    Rcpp::StringVector out(n);
    for (int64_t k = 0; k < n; ++k) out[k] = std::string(chars + off[k], (size_t)(off[k + 1] - off[k]));
    return out;
}

// list of alignments (each an XStringSet / character vector of rows) -> all rows + row ranges
struct FlatList {
    Flat rows;
    std::vector<int64_t> ranges;
    FlatList() : ranges(1, 0) {}
    int64_t n() const { return (int64_t)ranges.size() - 1; }
};

inline FlatList flatten_list(Rcpp::List l, bool decode_dna) {
    FlatList out;
    for (R_xlen_t k = 0; k < l.size(); ++k) {
        out.rows.append(flatten(l[k], decode_dna));
        out.ranges.push_back(out.rows.n());
    }
    out.rows.chars.push_back(0);
    return out;
}

inline FlatList flatten_single(SEXP x, bool decode_dna) {    // one alignment = a list of length 1
    FlatList out;
    out.rows = flatten(x, decode_dna);
    out.ranges.push_back(out.rows.n());
    return out;
}

#endif
