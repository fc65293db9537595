/*
 * sarlacc_amd.h -- C ABI of the MI355X-native sarlacc hot path (libsarlacc_amd.so).
 *
 * Every entry point replaces one `.Call` routine of the reference
 * (/root/reference/src/init.cpp:9-35, prototypes src/sarlacc.h:14-36); the
 * citation next to each declaration names the routine it stands in for.  The
 * reference's SEXP arguments become flat arrays:
 *
 *   XStringSet / character vector  ->  const char* chars (concatenated, no NULs)
 *                                      + const int64_t* off (n+1 offsets)
 *   named numeric `encoding`       ->  const double* enc_errors + const char* enc_names
 *                                      (names[i] is the single-character name of errors[i])
 *   list of integer vectors        ->  CSR: int64 off[n+1] + int32 values (1-based as in R)
 *   numeric / integer scalars      ->  double / int
 *
 * Two levels are exported:
 *   sarlacc_<name>(...)      host pointers in, host pointers out (what a cgo/.Call/ctypes
 *                            shim binds; does the H2D/D2H copies itself);
 *   sarlacc_dev_<name>(...)  device pointers in/out, asynchronous on a caller stream
 *                            (inputs already resident in HBM; used by bench.py and by a
 *                            pipeline that keeps reads on the GPU between stages).
 *
 * All functions return 0 on success.  On failure they return nonzero and
 * sarlacc_last_error() returns the message; messages that the reference throws
 * are reproduced verbatim (src/utils.cpp, src/reference_align.cpp, ...).
 * There is NO CPU fallback: without a usable HIP device every compute entry
 * point fails with an error.
 */
#ifndef SARLACC_AMD_H
#define SARLACC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* library / device management (no reference counterpart)               */
const char* sarlacc_last_error(void);
int sarlacc_version(void);
/* Number of visible HIP devices (0 when none / runtime unavailable). */
int sarlacc_device_count(void);
/* Select the device used by this thread's subsequent calls (default 0). */
int sarlacc_set_device(int device);
/* Release cached device workspaces (and the idle page-locked host blocks of sarlacc_host_alloc). */
void sarlacc_release_workspace(void);
/* Frees only the cached device buffers of the umi_group stage (the neighbour search, the sorted link lists, the clustering:
 * 17 GB after a call on 8 x 10^6 UMIs) -- for pipelines that go on to the MSA stage in the same process and want that memory
 * for it.  Returns the number of bytes given back. */
int64_t sarlacc_release_umi_workspace(void);
/* What the library's cached device buffers hold right now: "name bytes" lines, largest first, into buf (truncated at cap);
 * returns the total in bytes (diagnostics: bench.py reports the largest ones of the giant pre-group leg). */
int64_t sarlacc_workspace_report(char* buf, int64_t cap);
/* Duration in ms of a named group of kernel launches of the last call that ran it (HIP events on
 * the launch stream, summed over the batches of the call): "msa_pairwise", "msa_merge", "consensus", "umi_pairs";
 * <0 if it never ran. */
double sarlacc_stage_ms(const char* name);
/* Work counters of the last call that set them (for rooflines): "msa_pairs", "msa_cells" (banded DP
 * cells of the pairwise alignments), "consensus_cells" (rows x width); <0 if unset. */
double sarlacc_stage_count(const char* name);
/* Duration in ms of the DP kernel launches recorded by the last
 * sarlacc_dev_* align call (HIP events on the launch stream); <0 if none. */
double sarlacc_last_kernel_ms(void);

/* ------------------------------------------------------------------ */
/* quality-weighted read-vs-reference DP                                 */

/* Reference (adaptor / barcode) length, all entry points of this section: up to 1 024 columns an alignment runs
 * inside a wavefront (k_align), beyond that one workgroup holds it (k_align_wide: strips of 8 192 columns for
 * longer references); beyond 2^20 columns a call fails with a message.  The reference itself (src/reference_align.cpp:7-13) has no limit. */

/* replaces .Call adaptor_align  (src/adaptor_align.cpp:11-77)
 * sec_starts 0-based, sec_ends 1-based (as passed by R/adaptorAlign.R:158).
 * Outputs: scores[n], starts[n], ends[n] (1-based, 0/0 when empty),
 *          sec_start_out / sec_width_out [nsec][n] row-major. */
int sarlacc_adaptor_align(const char* seq, const int64_t* seq_off,
                          const char* qual, const int64_t* qual_off, int64_t n,
                          const double* enc_errors, const char* enc_names, int enc_n,
                          double gapopen, double gapext,
                          const char* adaptor, int adaptor_len,
                          const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                          double* scores, int32_t* starts, int32_t* ends,
                          int32_t* sec_start_out, int32_t* sec_width_out);

/* replaces .Call adaptor_align_score_only  (src/adaptor_align.cpp:79-110) */
int sarlacc_adaptor_align_score_only(const char* seq, const int64_t* seq_off,
                                     const char* qual, const int64_t* qual_off, int64_t n,
                                     const double* enc_errors, const char* enc_names, int enc_n,
                                     double gapopen, double gapext,
                                     const char* adaptor, int adaptor_len, double* scores);

/* replaces .Call barcode_align  (src/barcode_align.cpp:10-44), global mode */
int sarlacc_barcode_align(const char* seq, const int64_t* seq_off,
                          const char* qual, const int64_t* qual_off, int64_t n,
                          const double* enc_errors, const char* enc_names, int enc_n,
                          double gapopen, double gapext,
                          const char* reference, int reference_len, double* scores);

/* replaces .Call general_align  (src/general_align.cpp:10-62), global mode.
 * aln_ref / aln_query receive the gapped strings back to back, aln_off[n+1] the
 * offsets; pass edit_only != 0 (then aln_* may be NULL) for scores + edit distances.
 * aln_cap is the capacity of each string buffer (sum(L)+n*R always suffices). */
int sarlacc_general_align(const char* seq, const int64_t* seq_off,
                          const char* qual, const int64_t* qual_off, int64_t n,
                          const double* enc_errors, const char* enc_names, int enc_n,
                          double gapopen, double gapext,
                          const char* reference, int reference_len, int edit_only,
                          double* scores, int32_t* edits,
                          char* aln_ref, char* aln_query, int64_t* aln_off, int64_t aln_cap);

/* replaces .Call mask_bad_bases  (src/mask_bad_bases.cpp:10-52); out has seq_off[n] bytes */
int sarlacc_mask_bad_bases(const char* seq, const int64_t* seq_off,
                           const char* qual, const int64_t* qual_off, int64_t n,
                           const double* enc_errors, const char* enc_names, int enc_n,
                           double threshold, char* out);

/* Device-resident form of the three score/trace entry points above.
 * d_seq/d_qual/d_off are device pointers (one shared offset array: lengths were
 * validated when the batch was uploaded); stream is a hipStream_t (NULL = default).
 *   mode: 0 = local (adaptor_align*), 1 = global (barcode_align/general_align)
 *   d_starts/d_ends/d_sec_* may be NULL for score-only.
 * max_len = longest read in the batch (host-known). */
int sarlacc_dev_align(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off,
                      int64_t n, int32_t max_len,
                      const double* enc_errors, const char* enc_names, int enc_n,
                      double gapopen, double gapext,
                      const char* reference, int reference_len, int mode,
                      const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                      double* d_scores, int32_t* d_starts, int32_t* d_ends,
                      int32_t* d_sec_start_out, int32_t* d_sec_width_out,
                      void* stream);

/* Resident read format of the DP kernel: 2-bit packed bases (4 per byte, base i of the
 * concatenated batch in bits 2*(i%4) of byte i/4: A=0 C=1 G=2 T=3) plus one exception bit per
 * base (8 per byte) set where the character is not A/C/G/T (such bases mismatch every A/C/G/T
 * reference column, exactly like the reference's `ref==obs` test, src/reference_align.cpp:188).
 * d_packed needs ceil(total/4)+1 bytes, d_nmask ceil(total/8) bytes. */
int sarlacc_dev_pack_reads(const uint8_t* d_seq, int64_t total, uint8_t* d_packed, uint8_t* d_nmask,
                           void* stream);

/* sarlacc_dev_align on the packed format (offsets still count bases). */
int sarlacc_dev_align_packed(const uint8_t* d_packed, const uint8_t* d_nmask, const uint8_t* d_qual,
                             const int64_t* d_off, int64_t n, int32_t max_len,
                             const double* enc_errors, const char* enc_names, int enc_n,
                             double gapopen, double gapext,
                             const char* reference, int reference_len, int mode,
                             const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                             double* d_scores, int32_t* d_starts, int32_t* d_ends,
                             int32_t* d_sec_start_out, int32_t* d_sec_width_out,
                             void* stream);

/* ---- resident batches for the scrambled-control callers ----
 * tuneAlignment (R/tuneAlignment.R:30-72) and getAdaptorThresholds
 * (R/getAdaptorThresholds.R:35-48,105-128) align the same read windows many times; these
 * helpers keep them in HBM: plain device allocation / copies, .get_front_and_back
 * (R/adaptorAlign.R:86-95) and .scramble_input (R/getAdaptorThresholds.R:68-92) on the device. */
int sarlacc_dev_malloc(void** p, int64_t bytes);
int sarlacc_dev_free(void* p);
/* (blocks of 1 MB and more are pooled by size -- powers of two, at most 4 GB idle per device --: a freed block waits for the
 * next request of its size; sarlacc_dev_pool_release, and sarlacc_release_workspace, give the idle ones back) */
int sarlacc_dev_pool_release(void);
int sarlacc_dev_upload(void* d, const void* h, int64_t bytes);
int sarlacc_dev_download(void* h, const void* d, int64_t bytes);
/* Page-locked host blocks for results (no reference counterpart: R hands its own vectors to .Call).  A device -> host copy
 * into such a block is one DMA transfer; into pageable memory it goes through the runtime's staging buffer.  Blocks are
 * pooled by size (powers of two from 1 MB); sarlacc_host_free returns a block to the pool, sarlacc_host_release frees the
 * idle blocks.  Any host pointer of this interface may point into one. */
int sarlacc_host_alloc(void** p, int64_t bytes);
int sarlacc_host_free(void* p);
int sarlacc_host_release(void);
/* which = 0: first min(tol,width) bases; which = 1: reverse complement of the last
 * min(tol,width) bases with reversed qualities.  d_woff: offsets of the windows (n+1). */
int sarlacc_dev_windows(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                        const int64_t* d_woff, int which, uint8_t* d_oseq, uint8_t* d_oqual, void* stream);
/* .resolve_strand and the row selection of .align_AA_internal (R/adaptorAlign.R:112-122, :190-207) on the results of four
 * sarlacc_dev_align calls that are still in HBM: cs / ce = adaptor 1 on the front windows / adaptor 2 on the back windows,
 * rs / re = adaptor 1 on the back / adaptor 2 on the front.  Each is ONE block of n alignments with S sections (S = nsec1 for
 * cs, rs, out1; nsec2 for ce, re, out2): double scores[n] | int32 starts[n] | int32 ends[n] | int32 section starts[max(S,1)][n]
 * | int32 section widths[max(S,1)][n] (the five outputs of sarlacc_dev_align at these offsets).  d_rev[r] = 1 iff
 * max(cs,0) + max(ce,0) < max(rs,0) + max(re,0); out1 row r = rs or cs, out2 row r = re or ce. */
int sarlacc_dev_choose_strand(const void* d_cs, const void* d_ce, const void* d_rs, const void* d_re, int64_t n, int nsec1, int nsec2,
                              void* d_out1, void* d_out2, uint8_t* d_rev, void* stream);
/* Sub-sequences of resident reads, straight to host strings (XVector::subseq in .align_and_extract, R/adaptorAlign.R:160-174):
 * output r = width[r] bases from the 1-based position start[r] of read r of batch A -- or of batch B where from_b[r] != 0
 * (from_b may be NULL; adaptorAlign's strand choice takes, per read, the alignment on the front or on the back window).
 * start / width / from_b are host arrays; out_off (n + 1) and out_chars (out_cap bytes) are filled on the host. */
int sarlacc_dev_subseq(const uint8_t* d_seq_a, const int64_t* d_off_a, const uint8_t* d_seq_b, const int64_t* d_off_b,
                       const uint8_t* from_b, const int32_t* start, const int32_t* width, int64_t n, char* out_chars,
                       int64_t out_cap, int64_t* out_off, void* stream);
/* realizeReads on resident reads (R/realizeReads.R:28-43): output r = read d_idx[r] (0-based),
 * reverse-complemented when d_rev[r] != 0 (qualities reversed), cut to the oriented positions
 * d_tstart[r] .. d_tstart[r] + width - 1 (1-based), widths given by the output offsets d_ooff. */
int sarlacc_dev_realize(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, const int64_t* d_idx,
                        const uint8_t* d_rev, const int32_t* d_tstart, int64_t n_out, const int64_t* d_ooff,
                        uint8_t* d_oseq, uint8_t* d_oqual, void* stream);
/* Per-read Fisher-Yates shuffle (bases and qualities together), splitmix64 stream per read. */
int sarlacc_dev_scramble(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                         uint64_t seed, uint8_t* d_oseq, uint8_t* d_oqual, void* stream);

/* replaces .Call unmask_alignment  (src/unmask_alignment.cpp:12-59): every 'N'/'n'
 * of a gapped row becomes the base at the same ungapped position of the original sequence.
 * out has the layout of aln. */
int sarlacc_unmask_alignment(const char* aln, const int64_t* aln_off, int64_t naln,
                             const char* orig, const int64_t* orig_off, int64_t norig, char* out);

/* ---- alignment profiling: variable-length results, two-call protocol -- *count / *nins always
 * receive the full number of entries; the arrays are filled only when their capacity suffices. ---- */

/* replaces .Call find_homopolymers  (src/homopolymer.cpp:87-134): runs of two or more equal bases of every (possibly
 * gapped) sequence: idx (0-based sequence), pos (1-based start in the ungapped sequence), size, base. */
int sarlacc_find_homopolymers(const char* seq, const int64_t* off, int64_t n, int32_t* idx, int32_t* pos, int32_t* size,
                              char* base, int64_t cap, int64_t* count);

/* replaces .Call match_homopolymers  (src/homopolymer.cpp:141-209): per alignment and homopolymer of the reference string
 * the longest run of the same base in the read string that overlaps it: idx (0-based alignment), pos (as above), rlen. */
int sarlacc_match_homopolymers(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                               int64_t nread, int32_t* idx, int32_t* pos, int32_t* rlen, int64_t cap, int64_t* count);

/* replaces .Call find_errors  (src/find_errors.cpp:9-121): per base of the (first alignment's) reference the number of
 * alignments whose read shows A, C, G, T or a deletion there, and one (position of the next reference base, 0-based;
 * length) entry per insertion.  *standard_len = bases of the reference; with cap_bases below it (or the count arrays
 * NULL) nothing else is computed. */
int sarlacc_find_errors(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                        int64_t nread, int64_t* standard_len, char* bases, int32_t* to_a, int32_t* to_c, int32_t* to_g,
                        int32_t* to_t, int32_t* deletions, int64_t cap_bases, int32_t* ins_pos, int32_t* ins_len,
                        int64_t cap_ins, int64_t* nins);

/* FASTQ text already in device memory -> resident read batch.  Replaces the
 * host-side ShortRead::FastqStreamer + .FASTQ2QSDS conversion (R/adaptorAlign.R:26-37,:104-110;
 * R/realizeReads.R:15-26).  4-line records, LF or CRLF, trailing blank lines ignored; sequences
 * are upper-cased, read names are the header lines without the '@'.
 * Step 1 indexes the text and reports the sizes the caller has to allocate; step 2 fills
 * d_seq/d_qual (total_bases bytes each), d_off (n+1), d_names (total_name_bytes, may be NULL
 * together with d_name_off) for the text indexed last on this thread. */
int sarlacc_dev_fastq_index(const uint8_t* d_text, int64_t nbytes, int64_t* n_records, int64_t* total_bases,
                            int64_t* total_name_bytes, void* stream);
/* Chunked streaming (the `number` argument of adaptorAlign, R/adaptorAlign.R:8,:26: FastqStreamer(filepath, n=number)):
 * d_text holds a block of the file that may end inside a record.  Reports how many of its leading
 * complete records (all four lines terminated by a newline), at most max_records, form the next
 * chunk and how many bytes they occupy; the caller indexes exactly those bytes, then continues at
 * d_text + consumed_bytes (appending more of the file when n_records < max_records). */
int sarlacc_dev_fastq_split(const uint8_t* d_text, int64_t nbytes, int64_t max_records, int64_t* n_records,
                            int64_t* consumed_bytes, void* stream);
int sarlacc_dev_fastq_extract(const uint8_t* d_text, uint8_t* d_seq, uint8_t* d_qual, int64_t* d_off,
                              uint8_t* d_names, int64_t* d_name_off, void* stream);

/* ------------------------------------------------------------------ */
/* masked Levenshtein, neighbour search, clustering                      */

/* replaces .Call compute_lev_masked  (src/compute_lev_masked.cpp:13-64);
 * out has n(n-1)/2 doubles, i-major lower triangle (R 'dist' order). */
int sarlacc_compute_lev_masked(const char* seq, const int64_t* off, int64_t n, double* out);

/* replaces .Call fast_levdist_test  (src/sorted_trie.cpp:304-337).
 * Neighbour lists (1-based) in the reference's trie order.  Two-call protocol:
 * nbr==NULL returns the required size in *nbr_need and fills nbr_off. */
int sarlacc_fast_levdist_test(const char* seq, const int64_t* off, int64_t n, int limit,
                              int64_t* nbr_off, int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need);

/* replaces .Call cluster_umis_test  (src/cluster_umis_test.cpp:8-30).
 * links: CSR of 1-based neighbour lists; clusters returned as CSR of 1-based ids
 * (clu_off needs n+1 entries, clu n entries). */
int sarlacc_cluster_umis_test(const int64_t* link_off, const int32_t* links, int64_t n,
                              int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* replaces .Call umi_group  (src/umi_group.cpp:14-116) followed by the
 * unlist(out, recursive=FALSE) of R/umiGroup.R:22.
 * umi2 may be NULL.  pregroups: CSR of 1-based read ids.  Output clusters as CSR
 * of 1-based read ids (clu_off: total+1 entries, clu: total entries, where
 * total = grp_off[ngroups]). */
int sarlacc_umi_group(const char* umi1, const int64_t* off1,
                      const char* umi2, const int64_t* off2, int64_t n,
                      int thresh1, int thresh2,
                      const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                      int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* One giant pre-group across several GPUs: the row tiles of the all-pairs
 * matrix shard; every shard returns its neighbour pairs (rank_i << 32 | rank_j, i < j, ranks in the
 * trie order of the whole set, which every shard computes identically), the pair lists are
 * exchanged (all-gather) and the clustering of umi_group runs on their concatenation.
 * Two-call sizing protocol for `pairs` as in sarlacc_fast_levdist_test. */
int sarlacc_umi_pairs_shard(const char* umi, const int64_t* off, int64_t n, int limit,
                            int shard_index, int shard_count,
                            uint64_t* pairs, int64_t cap, int64_t* npairs);
int sarlacc_umi_group_from_pairs(const char* umi, const int64_t* off, int64_t n, int limit,
                                 const uint64_t* pairs, int64_t npairs,
                                 int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* The same exchange with the pairs kept in HBM (at 8 x 10^6 reads the lists hold 10^8 pairs; nothing of them crosses
 * PCIe): sarlacc_dev_umi_pairs_shard searches the shard's row tiles and leaves its pairs in the library's workspace on the
 * device, reporting their number; sarlacc_dev_umi_pairs_fetch copies them (device to device) into a caller buffer of at
 * least that many entries -- e.g. the send buffer of an RCCL all-gather -- and must be the next library call of the thread;
 * sarlacc_dev_umi_group_from_pairs clusters from a DEVICE array of pairs (validated on the device; the call waits for the
 * device first, so a collective on another stream has finished writing them).
 * Limits shared by every umi_group entry point: strings of at most 1024 bases (longer ones only alone in their pre-group),
 * characters ACGTN only, and at most 2^32 - 1 neighbour links (self links included) per call -- the lists are explicit as in
 * src/umi_group.cpp:59-103; a threshold that joins most of a large set must be lowered or the reads split into pre-groups. */
int sarlacc_dev_umi_pairs_shard(const char* umi, const int64_t* off, int64_t n, int limit,
                                int shard_index, int shard_count, int64_t* npairs);
int sarlacc_dev_umi_pairs_fetch(uint64_t* d_pairs, int64_t cap);
int sarlacc_dev_umi_group_from_pairs(const char* umi, const int64_t* off, int64_t n, int limit,
                                     const uint64_t* d_pairs, int64_t npairs,
                                     int64_t* nclusters, int64_t* clu_off, int32_t* clu);

/* ------------------------------------------------------------------ */
/* per-group MSA and consensus                                           */

/* Which written specification the MSA stage follows (the reference delegates to SeqAn's
 * T-Coffee, which cannot be run or pinned here): 2 (default) = consistency-based progressive alignment --
 * all-pairs banded alignments, primary library, triplet extension into a bounded library (the direct partner and three
 * further positions per pair and base), neighbour-joining guide tree, progressive
 * heaviest-common-subsequence merging -- for groups of up to 64 reads, 1 = centre-star (also used by spec 2 for
 * larger groups, for reads beyond 65 471 bases and for alignments wider than 65 535 columns).  0 restores the default, spec 2 (SARLACC_MSA_SPEC only sets the
 * value the process starts with: the environment is read once). */
int sarlacc_set_msa_spec(int spec);

/* A/B switches of the tests and the perf tools (every default, 0, is the product path; the table is kOptNames in
 * csrc/common.cpp, the meanings are at enum Opt in csrc/common.hpp and in INTEGRATION.md "Options"):
 *   "msa_spec" (1 / 2), "msa2_general_rows", "msa2_chain_hbm", "msa2_waves_per_cu" (a count), "msa2_single_wave",
 *   "msa2_batches" (a count), "msa2_tight_profiles", "align_pensel", "align_chunks" (a count), "align_k" (columns per lane),
 *   "align_waves_per_cu" (a count), "align_interleave" (-1 never / 1 with align_k), "consensus_chars", "consensus_generic",
 *   "msa_int32", "msa_affine", "msa_bitvector" (-1 never, 2 one kernel for fill and walk), "msa_bitvector_core" (-1 whole
 *   records, 1 one word), "msa_bitvector_tile_gb" (GB), "umi_full_rounds", "umi_tile_search", "umi_split_min" (a set size),
 *   "umi_scan_single", "align_wide_barrier" (references beyond 1 024 columns: the kernel with a barrier per step everywhere),
 *   "align_wide_band" (rows either side of the main diagonal whose cells carry traceback codes in k_align_wide_q's first launch; -1: all),
 *   "msa2_budget_gb" (GB a batch of groups may take), "msa2_max_columns" (a lower ceiling of spec v2's profiles),
 *   "msa2_simple_extend" (the extended library by the one-position-per-lane kernel everywhere), "msa2_wide_extend" (largest
 *   group size of the four-positions-per-lane kernel).
 * The environment (SARLACC_<NAME>) is read once, when the first option is asked for; afterwards only this call changes a
 * value.  Nothing in the reference corresponds. */
int sarlacc_set_option(const char* name, int value);

/* replaces .Call quick_msa  (src/quick_msa.cpp:15-80); argument order as there
 * (the R caller passes -gapOpening as gap_extension and -gapExtension as
 * gap_opening, R/multiReadAlign.R:47).  Groups: CSR of 1-based read ids.
 * Output: for group g, rows_g = size(g) strings of equal width width_out[g],
 * written row-major at out + out_off[g]; out_off has ngroups+1 entries.
 * Two-call protocol: out==NULL fills width_out/out_off only. */
int sarlacc_quick_msa(const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                      const char* seq, const int64_t* seq_off, int64_t nseq,
                      double match, double mismatch, double gap_extension, double gap_opening,
                      int bandwidth,
                      int32_t* width_out, int64_t* out_off, char* out, int64_t out_cap);

/* replaces .Call create_consensus_basic_loop  (src/create_consensus.cpp:150-170)
 * and, with ngroups==1 and lerr!=NULL, create_consensus_basic (:137-148).
 * Alignments: aln_off[nrows_total+1] string offsets, grp_rows[ngroups+1] row
 * ranges per alignment.  cons/phred: concatenated outputs with cons_off[ngroups+1];
 * capacity of each is the total alignment width.  lerr (optional) gets the
 * natural-log error of every kept column, same offsets. */
int sarlacc_create_consensus_basic_loop(const char* aln, const int64_t* aln_off,
                                        const int64_t* grp_rows, int64_t ngroups,
                                        double min_cov, double pseudo_count,
                                        char* cons, char* phred, int64_t* cons_off, double* lerr);

/* replaces .Call create_consensus_quality_loop  (src/create_consensus.cpp:287-308)
 * and create_consensus_quality (:274-285).  qual strings are per alignment row,
 * ungapped (consumed positionally, N included). */
int sarlacc_create_consensus_quality_loop(const char* aln, const int64_t* aln_off,
                                          const int64_t* grp_rows, int64_t ngroups,
                                          const char* qual, const int64_t* qual_off,
                                          const int64_t* qgrp_rows,
                                          double min_cov,
                                          const double* enc_errors, const char* enc_names, int enc_n,
                                          char* cons, char* phred, int64_t* cons_off, double* lerr);

/* multiReadAlign + consensusReadSeq in one call (R/multiReadAlign.R:16-47 followed by
 * R/consensusReadSeq.R:14-21; i.e. .Call quick_msa then .Call create_consensus_*_loop): the
 * gapped rows never leave HBM and the quality strings stay in read order (rows find theirs
 * through the group lists), so nothing is re-marshalled between the two stages.  Results are
 * those of sarlacc_quick_msa + sarlacc_create_consensus_{quality,basic}_loop on the same input.
 * qual == NULL selects the basic vote (pseudo_count used), otherwise the quality-weighted vote.
 * cons/phred need the KEPT columns of every group (at most the alignment's width; the exact total is reported in the error
 * message when cons_cap is too small -- the vote has run by then; 1.5 x the longest member per group is a safe first guess:
 * same-molecule groups keep about one read length, and a cluster of several molecules, whose alignment is several reads
 * wide, keeps the columns its majority covers). */
int sarlacc_msa_consensus(const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                          const char* seq, const int64_t* seq_off,
                          const char* qual, const int64_t* qual_off, int64_t nseq,
                          double match, double mismatch, double gap_extension, double gap_opening,
                          int bandwidth, double min_cov, double pseudo_count,
                          const double* enc_errors, const char* enc_names, int enc_n,
                          char* cons, char* phred, int64_t* cons_off, int64_t cons_cap);

/* sarlacc_msa_consensus with the reads (and, for the quality vote, their quality strings) already
 * resident in HBM, e.g. left there by sarlacc_dev_fastq_extract / sarlacc_dev_realize: d_seq and
 * d_qual are device pointers to the concatenated strings, `off` the HOST copy of their n+1 offsets
 * (byte 0 of d_seq is off[0]; the job list is built from the lengths on the host).  d_qual == NULL
 * selects the basic vote.  Group lists and outputs are host arrays as in sarlacc_msa_consensus. */
int sarlacc_dev_msa_consensus(const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                              const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* off, int64_t nseq,
                              double match, double mismatch, double gap_extension, double gap_opening,
                              int bandwidth, double min_cov, double pseudo_count,
                              const double* enc_errors, const char* enc_names, int enc_n,
                              char* cons, char* phred, int64_t* cons_off, int64_t cons_cap);

#ifdef __cplusplus
}
#endif
#endif
