/*
 * oracle.h -- CPU restatement of sarlacc's alignment-and-consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported CPU baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - the reference itself cannot be built in this image (it needs Rcpp, R,
 *     Biostrings and SeqAn headers which are absent, and stand-in headers are
 *     not allowed), so every function below is pinned against the literal
 *     known answers in the reference's own testthat files and against
 *     restatements of the R-side test oracles (tests/test_oracle_*.py);
 *   - orc_msa_* has NO reference counterpart that can be pinned (the reference
 *     delegates to SeqAn's T-Coffee and has no test of it): PARITY UNPINNED.
 *
 * String sets are passed as one concatenated byte buffer plus n+1 int64
 * offsets.  All functions return 0 on success; on failure they return nonzero
 * and orc_last_error() holds the message (same text as the reference throws).
 */
#ifndef SARLACC_ORACLE_H
#define SARLACC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* orc_last_error(void);

/* ---- quality encoding (reference src/quality_encoding.cpp:5-47) ---- */
/* names[i] is the single-character name of errors[i]. */
int orc_check_encoding(const double* errors, const char* names, int n);

/* ---- DP cost tables (reference src/reference_align.cpp:21-52) ---- */
/* match/mismatch: [4][n] row-major. */
int orc_cost_tables(const double* errors, int n, double* match, double* mismatch);

/* ---- quality-weighted affine DP (reference src/reference_align.cpp:54-181) ---- */
/* dirs: (L+1)*(R+1) int32, column-major [c*(L+1)+i]; may be NULL for score only. */
int orc_align_one(const char* ref, int R, const char* seq, const char* qual, int L,
                  const double* errors, const char* names, int nenc,
                  double gapopen, double gapext, int local,
                  double* score, int32_t* dirs);

/* .Call adaptor_align (reference src/adaptor_align.cpp:11-77) */
int orc_adaptor_align(const char* seq, const int64_t* seq_off,
                      const char* qual, const int64_t* qual_off, int64_t n,
                      const double* errors, const char* names, int nenc,
                      double gapopen, double gapext,
                      const char* adaptor, int adaptor_len,
                      const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                      double* scores, int32_t* starts, int32_t* ends,
                      int32_t* sec_start_out, int32_t* sec_width_out);

/* .Call adaptor_align_score_only (:79-110) and barcode_align (src/barcode_align.cpp:10-44) */
int orc_align_scores(const char* seq, const int64_t* seq_off,
                     const char* qual, const int64_t* qual_off, int64_t n,
                     const double* errors, const char* names, int nenc,
                     double gapopen, double gapext,
                     const char* ref, int ref_len, int local, double* scores);

/* .Call general_align (src/general_align.cpp:10-62).
 * aln_ref/aln_query: caller-provided buffers, aln_off[n+1] receives offsets
 * (strings are NOT NUL-terminated); pass aln_ref=NULL for edit_only. */
int orc_general_align(const char* seq, const int64_t* seq_off,
                      const char* qual, const int64_t* qual_off, int64_t n,
                      const double* errors, const char* names, int nenc,
                      double gapopen, double gapext,
                      const char* ref, int ref_len,
                      double* scores, int32_t* edits,
                      char* aln_ref, char* aln_query, int64_t* aln_off, int64_t aln_cap);

/* ---- masking (reference src/mask_bad_bases.cpp:10-52) ---- */
int orc_mask_bad_bases(const char* seq, const int64_t* seq_off,
                       const char* qual, const int64_t* qual_off, int64_t n,
                       const double* errors, const char* names, int nenc,
                       double threshold, char* out);

/* ---- unmasking (reference src/unmask_alignment.cpp:12-59) ---- */
int orc_unmask_alignment(const char* aln, const int64_t* aln_off, int64_t naln, const char* orig,
                         const int64_t* orig_off, int64_t norig, char* out);

/* ---- alignment profiling (reference src/homopolymer.cpp, src/find_errors.cpp) ----
 * Variable-length outputs: two-call protocol, *count is always the full number of entries. */
int orc_find_homopolymers(const char* seq, const int64_t* off, int64_t n, int32_t* idx, int32_t* pos, int32_t* size,
                          char* base, int64_t cap, int64_t* count);
int orc_match_homopolymers(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                           int64_t nread, int32_t* idx, int32_t* pos, int32_t* rlen, int64_t cap, int64_t* count);
/* returns 2 when cap_bases is smaller than *standard_len (nothing else filled) */
int orc_find_errors(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                    int64_t nread, int64_t* standard_len, char* bases, int32_t* to_a, int32_t* to_c, int32_t* to_g,
                    int32_t* to_t, int32_t* deletions, int64_t cap_bases, int32_t* ins_pos, int32_t* ins_len, int64_t cap_ins,
                    int64_t* nins);

/* per-read shuffle of the scrambled-control callers (our generator, see align.c) */
int orc_scramble(const char* seq, const char* qual, const int64_t* off, int64_t n, uint64_t seed,
                 char* oseq, char* oqual);

/* ---- masked Levenshtein ---- */
/* dense lower triangle, i-major (reference src/compute_lev_masked.cpp:13-64) */
int orc_compute_lev_masked(const char* seq, const int64_t* off, int64_t n, double* out);
/* trie neighbour search (reference src/sorted_trie.cpp; test hook :304-337).
 * Output CSR: nbr_off[n+1], nbr[] (0-based), capacity nbr_cap; returns needed size in *nbr_need. */
int orc_fast_levdist(const char* seq, const int64_t* off, int64_t n, int limit,
                     int64_t* nbr_off, int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need);

/* ---- greedy clustering (reference src/cluster_umis.cpp:7-112) ---- */
/* in: CSR links (0-based); out: CSR clusters (clu_off has at most n+1 entries). */
int orc_cluster_umis(const int64_t* link_off, const int32_t* links, int64_t n,
                     int64_t* nclusters, int64_t* clu_off, int32_t* clu);
/* same result, O(E log) bucket implementation for large n (validated against the above) */
int orc_cluster_umis_fast(const int64_t* link_off, const int32_t* links, int64_t n,
                          int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* ---- umi_group (reference src/umi_group.cpp:14-116) ---- */
/* pregroups: CSR of 1-based read ids.  Output: flattened list of clusters of
 * 1-based read ids (R side does unlist(out, recursive=FALSE), R/umiGroup.R:22),
 * clu_off must hold total_reads+1 entries, clu total_reads entries. */
int orc_umi_group(const char* umi1, const int64_t* off1,
                  const char* umi2, const int64_t* off2, /* umi2 may be NULL */
                  int64_t n, int thresh1, int thresh2,
                  const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                  int fast_cluster,
                  int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* ---- consensus (reference src/create_consensus.cpp) ---- */
/* one alignment: rows concatenated; cons (cap >= width+1) NUL-terminated, lerr[width] */
int orc_consensus_basic(const char* aln, const int64_t* off, int64_t nrows,
                        double mincov, double pseudo,
                        char* cons, double* lerr, int64_t* conlen);
int orc_consensus_quality(const char* aln, const int64_t* off, int64_t nrows,
                          const char* qual, const int64_t* qoff, int64_t nquals,
                          double mincov,
                          const double* errors, const char* names, int nenc,
                          char* cons, double* lerr, int64_t* conlen);
/* Phred+33 string from log-errors (reference src/create_consensus.cpp:18-32) */
void orc_errors_to_string(const double* lerr, int64_t n, char* out);

/* ---- MSA: own specification (DESIGN.md "MSA spec v1"), PARITY UNPINNED ---- */
/* One group: reads (concatenated + offsets, nreads), integer scores.
 * out: nreads rows of equal width *width, written row-major into out (cap bytes). */
int orc_msa_group(const char* seq, const int64_t* off, int64_t nreads,
                  int match, int mismatch, int gapopen, int gapext, int bandwidth,
                  char* out, int64_t cap, int64_t* width);

/* Banded global Gotoh of read r (rows) against centre c (columns), both already Dna5: the pairwise
 * stage shared by spec v1 and spec v2.  ins_cnt[p] (p <= lc): read characters inserted before centre
 * position p; aligned[p] (p < lc): read position matched to centre position p, or -1. */
/* widest band (diagonals) any pairwise alignment of the MSA stage uses; see orc_msa_pairwise */
#define ORC_MSA_MAXBAND 1024
int orc_msa_pairwise(const char* r, int64_t lr, const char* c, int64_t lc,
                     int ma, int mm, int go, int ge, int bw, int32_t* ins_cnt, int64_t* aligned);

/* ---- MSA spec v2: T-Coffee as SeqAn's globalMsaAlignment runs it for small
 * groups, at base resolution -- all-pairs banded global alignments, primary library, full triplet
 * extension, neighbour-joining guide tree, progressive heaviest-common-subsequence alignment.
 * PARITY UNPINNED (SeqAn is absent, the reference has no test of quick_msa).  Same interface as
 * orc_msa_group. */
int orc_msa2_group(const char* seq, const int64_t* off, int64_t nreads,
                   int match, int mismatch, int gapopen, int gapext, int bandwidth,
                   char* out, int64_t cap, int64_t* width);
/* Inspection hooks for the tests: the guide tree (joins[2*k], joins[2*k+1] = node ids merged by join k;
 * leaves 0..n-1, join k creates node n+k) and the pairwise distances (n*n doubles). */
/* spec v2's own rules (row cap, noise filter) switched off for A/B runs, and counters of how often they act:
 * joins, rows, rows with candidates, rows capped, candidates ignored by the cap, entries before the filter,
 * entries filtered, rows filtered, entries kept, (a,p,b) triples, triples naming >= 2 / >= 3 positions of b,
 * triples whose direct edge is a gap, candidates, rows with > 1 entry, most entries in a row, partner positions the
 * bounded library ignored. */
void orc_msa2_set_rules(int nocap, int nofilter);
/* other partner positions kept per (a, p, b) beside the direct one (spec: 3); -1 = the unbounded library of rounds 2-4 */
void orc_msa2_set_library(int others);
int orc_msa2_stats(int64_t* out, int64_t cap, int reset);
int orc_msa2_tree(const char* seq, const int64_t* off, int64_t nreads,
                  int match, int mismatch, int gapopen, int gapext, int bandwidth,
                  int32_t* joins, double* dist);

#ifdef __cplusplus
}
#endif
#endif
