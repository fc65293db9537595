// common.hpp -- shared host-side plumbing of libsarlacc_amd.so: error reporting,
// per-thread device context, cached device workspaces, upload helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace sarlacc {

// ---- errors -------------------------------------------------------------
std::string& last_error();
int fail(const char* fmt, ...);

#define SL_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess)                                                    \
            return ::sarlacc::fail("HIP error %s at %s:%d (%s)", hipGetErrorName(e__), \
                                   __FILE__, __LINE__, #expr);                    \
    } while (0)

#define SL_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__) return rc__;    \
    } while (0)

// ---- options --------------------------------------------------------------
// The A/B switches the tests and the perf tools use; every default (0) is the product path.  Read ONCE from
// the environment (SARLACC_<NAME IN CAPITALS>) the first time any option is asked for, afterwards only changed
// through sarlacc_set_option -- no kernel launch path evaluates the environment.
enum Opt {
    OPT_MSA_SPEC,             // 1: centre-star, 2: consistency-based progressive (0 = default = 2)
    OPT_MSA2_GENERAL_ROWS,    // spec v2: the any-weights records and row lists also for unit weights
    OPT_MSA2_CHAIN_HBM,       // spec v2: the chain's prefix maxima in HBM from the start (the fallback of the LDS ring)
    OPT_MSA2_WAVES_PER_CU,    // spec v2: resident wavefronts of the merge kernel per CU (perf sweeps)
    OPT_MSA2_SINGLE_WAVE,     // spec v2: one wavefront per group whatever its size (A/B of the 4- and 8-wavefront workgroups)
    OPT_MSA2_BATCHES,         // spec v2: number of pipelined batches of a call (default: as few as memory allows)
    OPT_ALIGN_PENSEL,         // quality DP: the instantiation with explicit penalty selects also for gapopen >= 0
    OPT_ALIGN_CHUNKS,         // host-pointer DP: number of upload chunks
    OPT_ALIGN_K,              // quality DP: reference columns per lane (perf sweeps)
    OPT_ALIGN_WAVES_PER_CU,   // quality DP: grid size (perf sweeps)
    OPT_CONSENSUS_CHARS,      // fused MSA + consensus on character rows instead of vote codes
    OPT_CONSENSUS_GENERIC,    // quality vote on character rows: the generic kernel only
    OPT_MSA_INT32,            // pairwise MSA alignments by the 32-bit kernel
    OPT_MSA_AFFINE,           // pairwise MSA alignments by the full affine recurrence also where open <= extend makes it linear
    OPT_UMI_FULL_ROUNDS,      // greedy clustering of dense graphs: every round walks every live list (A/B of the candidate-set rounds)
    OPT_UMI_TILE_SEARCH,      // neighbour search at thresholds 1 to 3 by the all-tile-pairs kernel (A/B of the split-key search)
    OPT_MSA_BITVECTOR,        // pairwise MSA alignments: -1 never by the bit-vector kernel of the unit-cost linear regime (A/B)
    OPT_MSA_BITVECTOR_CORE,      // bit-vector pairwise kernel: -1 whole traceback records always; 1 walks may use ONE word of the partial records (tests of the second run)
    OPT_MSA_BITVECTOR_TILE_GB,   // bit-vector pairwise kernel: GB of traceback records per chunk of batches (default 48; perf sweeps)
    OPT_ALIGN_INTERLEAVE,     // quality DP: -1 never two 8-lane alignments per DPP row (A/B of that shape); 1 with align_k: that shape
    OPT_UMI_SPLIT_MIN,        // smallest set (and half of it: smallest average pre-group) that takes the split-key search (default 32768; tests lower it)
    OPT_MSA2_TIGHT_PROFILES,  // spec v2: first-pass profile capacity of 2 read lengths whatever the group size (tests of the second pass with exact capacity)
    OPT_UMI_SCAN_SINGLE,      // split-key search: one candidate column per lane also where two fit (A/B of k_sk_scan_pk)
    OPT_ALIGN_WIDE_BARRIER,   // references beyond 1 024 columns: the kernel with a barrier per step also where the queue kernel applies (A/B)
    OPT_MSA2_BUDGET_GB,       // spec v2: cap (GB) of the memory one batch of groups may take (default 96; at most half of what is free)
    OPT_MSA2_MAX_COLUMNS,     // spec v2: alignments wider than this go to spec v1 (default and maximum 65535: 16-bit columns; tests lower it)
    OPT_ALIGN_WIDE_BAND,      // k_align_wide_q with traceback: rows either side of the main diagonal that carry codes in the first launch (default 96; -1: every cell)
    OPT_MSA2_SIMPLE_EXTEND,   // spec v2: the extended library by the one-position-per-lane kernel also for unit weights (A/B, tests)
    OPT_MSA2_WIDE_EXTEND,     // spec v2: largest group size that takes the four-positions-per-lane extension kernel (default 12; A/B)
    OPT_N
};
int option(Opt o);
int set_option(const char* name, int value);

// ---- device context -------------------------------------------------------
// One per host thread (a .Call runs on R's main thread; BiocParallel workers are
// separate processes, /root/reference/R/adaptorAlign.R:126-134).
struct Workspace {
    void* ptr = nullptr;
    size_t cap = 0;
};

struct Context {
    int device = 0;
    bool ready = false;
    int num_cu = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    std::map<std::string, Workspace> ws;
    // named stage timers (HIP events on the launch stream; read back by sarlacc_stage_ms)
    // A stage may run in several segments (batches); sarlacc_stage_ms adds them up.  stage_reset starts a new call.
    struct StageTimer { std::vector<std::pair<hipEvent_t, hipEvent_t>> segs; size_t used = 0; bool open = false; };
    std::map<std::string, StageTimer> stages;
    std::map<std::string, double> counts;   // work counters of the last call (cells, jobs ...), sarlacc_stage_count
    void stage_reset(const char* name);
    int stage_begin(const char* name, hipStream_t s);
    int stage_end(const char* name, hipStream_t s);

    // Returns a cached device buffer of at least `bytes` (grown geometrically).
    int buffer(const char* name, size_t bytes, void** out);
    void release();
};

Context& ctx();
// Makes sure a HIP device is usable; fails loudly otherwise (no CPU fallback).
int ensure_device();

template <typename T>
int upload(const char* name, const T* host, size_t count, T** dev, hipStream_t s) {
    void* p = nullptr;
    SL_TRY(ctx().buffer(name, (count ? count : 1) * sizeof(T), &p));
    if (count) SL_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    *dev = static_cast<T*>(p);
    return 0;
}

template <typename T>
int scratch(const char* name, size_t count, T** dev) {
    void* p = nullptr;
    SL_TRY(ctx().buffer(name, (count ? count : 1) * sizeof(T), &p));
    *dev = static_cast<T*>(p);
    return 0;
}

// ---- MSA stage with the rows left on the device (msa.hip) ---------------------
// Rows as vote codes (fused MSA + quality consensus): the row writers know the read position of every cell, so they
// can emit, instead of the character, the 16-bit code the vote kernel needs -- the index of the (quality, base)
// entry in its LDS table, bit 15 for a gap (consensus.hip: k_consensus_code).  The qualities must be laid out like
// the reads (same relative offsets).
struct CodeSpec {
    bool want = false;
    const uint8_t* const* qual = nullptr;   // *qual: the qualities in HBM, valid once `ready` has fired
    hipEvent_t ready = nullptr;             // null: already there
    int qoffset = 0, navail = 0;
    int* d_bad = nullptr;                   // device: smallest flat row with a quality below the encoding (INT_MAX: none)
};
struct MsaResult {
    std::vector<int32_t> width;     // per group (caller sizes it: ngroups)
    std::vector<int64_t> out_off;   // per group start of its rows in d_out (caller sizes it: ngroups + 1)
    uint8_t* d_out = nullptr;       // gapped rows, group after group, equal width inside a group
    int32_t* d_members = nullptr;   // flattened 1-based read ids, one per row
    CodeSpec code;                  // in: code rows wanted instead of characters
    uint16_t* d_codes = nullptr;    // out: the rows as codes (offsets as for d_out, in cells)
};
// d_seq_resident (optional): the concatenated reads already in HBM (byte 0 = seq_off[0]); `seq` is then
// not read and nothing is uploaded but the offsets.
// `overlap` (optional) runs on the host right after the pairwise kernels are launched, i.e. while
// they execute: the place for transfers the next stage needs (on a stream of their own).
int msa_run(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq, const int64_t* seq_off,
            int64_t nseq, double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
            bool want_rows, int64_t out_cap, MsaResult* res, const std::function<int()>* overlap = nullptr,
            const uint8_t* d_seq_resident = nullptr);
// spec v1 (centre-star, msa.hip) with the same interface; msa_run (msa2.hip) applies spec v2 and hands the
// groups v2 does not take to it
int msa1_run(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq, const int64_t* seq_off,
             int64_t nseq, double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
             bool want_rows, int64_t out_cap, MsaResult* res, const std::function<int()>* overlap = nullptr,
             const uint8_t* d_seq_resident = nullptr, bool accumulate = false);

// ---- quality encoding (reference src/quality_encoding.cpp:5-33) -------------
int check_encoding(const double* errors, const char* names, int n);

}  // namespace sarlacc
