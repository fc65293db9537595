// Declaration-only stand-in for the reference package's src/sarlacc.h (it stays in the shimmed package): the 18 .Call
// routines with the arities of src/init.cpp:9-35.  See README.md in this directory (test infrastructure).
#ifndef RGLUE_STUB_SARLACC_H
#define RGLUE_STUB_SARLACC_H
#include "Rcpp.h"
extern "C" {
SEXP adaptor_align(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP adaptor_align_score_only(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP barcode_align(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP general_align(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP mask_bad_bases(SEXP, SEXP, SEXP, SEXP);
SEXP unmask_alignment(SEXP, SEXP);
SEXP create_consensus_basic(SEXP, SEXP, SEXP);
SEXP create_consensus_basic_loop(SEXP, SEXP, SEXP);
SEXP create_consensus_quality(SEXP, SEXP, SEXP, SEXP);
SEXP create_consensus_quality_loop(SEXP, SEXP, SEXP, SEXP);
SEXP umi_group(SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP fast_levdist_test(SEXP, SEXP, SEXP);
SEXP cluster_umis_test(SEXP);
SEXP compute_lev_masked(SEXP);
SEXP quick_msa(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP find_homopolymers(SEXP);
SEXP match_homopolymers(SEXP, SEXP);
SEXP find_errors(SEXP, SEXP);
}
#endif
