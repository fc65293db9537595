// msa_common.hpp -- job descriptors shared by the pairwise kernels (msa_pairwise.hip) and the two
// MSA drivers built on them (msa.hip: spec v1 centre-star; msa2.hip: spec v2 consistency-based
// progressive alignment).  Both stand in for /root/reference/src/quick_msa.cpp:15-80.
#pragma once

#include "common.hpp"

#include <vector>

namespace sarlacc {

// One banded global alignment: `read` (rows i) against `centre` (columns j).
struct MsaJob {
    long long read_off;   // into seq
    long long ctr_off;
    long long out_off;    // OUT 0: into ins[] (lc+1 entries) and aln[] (lc entries, same base)
                          // OUT 1: into map[]: lc entries, centre position -> read position (0xFFFF = gap)
    long long out2_off;   // OUT 1: into map[]: lr entries, read position -> centre position
    int lr, lc;
};

struct MsaArgs {
    const uint8_t* seq;
    const MsaJob* jobs;
    const int* order;       // optional: the njobs job indices this launch works on
    int njobs;
    int ma, mm, go, ge, bw;
    uint16_t* ins;          // OUT 0: per pair, insertions before each centre position
    uint8_t* aln;           // OUT 0: per pair, 1 if the centre base is matched to a read base
    uint16_t* map;          // OUT 1: position maps of both directions
    int2* stats;            // OUT 1: per job (aligned pairs with equal bases, aligned pairs)
    void* tb;               // per-wave traceback tile
    unsigned long long tb_per_wave;  // in tile words
    uint32_t* moves;        // bit-vector kernel: per job, the traceback as a move string (see k_msa_pairwise_bv)
    unsigned moves_stride;  // words per job
    int batch0, batch1;     // bit-vector kernel: the batches (of 64 jobs) this launch works on
    int core_lo, core_hi;   // bit-vector kernel with partial records: the words a walk may use (inside the kept ones; tests narrow it)
    int skip_wide;          // bit-vector kernel: the job list holds jobs of wider band classes too; their lanes stay idle
    int* stuck;             // set when a traceback exceeds its step bound (cannot happen with consistent codes;
                            // the bound is what guarantees that every wave leaves the walk)
};

// Band cap of the MSA specification (both versions; oracle/msa.c orc_msa_pairwise): a pair is aligned inside
// at most MSA_MAXBAND diagonals.  Where |lc - lr| + 2 bandwidth + 1 exceeds that, the bandwidth of that pair
// shrinks to the largest that fits; where even |lc - lr| + 1 does not fit, the pair gets the diagonal
// alignment (position p opposite position p, the tail of the longer sequence unaligned) without any DP.
// Returns the pair's bandwidth, or -1 for the diagonal alignment.
constexpr int MSA_MAXBAND = 1024;
__host__ __device__ __forceinline__ int msa_pair_bandwidth(int bandwidth, int lr, int lc) {
    const long long dl = lc > lr ? lc - lr : lr - lc;
    if (dl + 2LL * bandwidth + 1 <= MSA_MAXBAND) return bandwidth;
    return MSA_MAXBAND - 1 - dl >= 0 ? static_cast<int>((MSA_MAXBAND - 1 - dl) / 2) : -1;
}
// diagonals the pair's DP covers (1 for the diagonal alignment)
__host__ __device__ __forceinline__ int msa_pair_band(int bandwidth, int lr, int lc) {
    const int bw = msa_pair_bandwidth(bandwidth, lr, lc);
    return bw < 0 ? 1 : (lc > lr ? lc - lr : lr - lc) + 2 * bw + 1;
}

// What the launcher needs to know about a job list (band classes: up to 256 / 512 / 1024 diagonals); a caller that builds
// the list with several threads fills it on the way (msa2.hip), otherwise the launcher walks the list itself.
struct MsaJobSummary {
    size_t n[3] = {0, 0, 0};
    int lr[3] = {0, 0, 0}, lc[3] = {0, 0, 0}, band[3] = {1, 1, 1};
    double cells = 0;   // rows x diagonals over all jobs
    double cols[3] = {0, 0, 0};   // centre columns per class
    std::vector<int> wide[2];     // indices of the jobs of classes 1 and 2 (few), when the caller passes the job's index to add()
    bool wide_listed = false;
    void add(int bandwidth, int jlr, int jlc, long long index = -1) {
        const int b = msa_pair_band(bandwidth, jlr, jlc);   // capped at MSA_MAXBAND by the spec
        const int cls = b <= 256 ? 0 : (b <= 512 ? 1 : 2);
        ++n[cls];
        cols[cls] += jlc;
        if (index >= 0) { wide_listed = true; if (cls > 0) wide[cls - 1].push_back(static_cast<int>(index)); }
        lr[cls] = jlr > lr[cls] ? jlr : lr[cls];
        lc[cls] = jlc > lc[cls] ? jlc : lc[cls];
        band[cls] = b > band[cls] ? b : band[cls];
        cells += static_cast<double>(jlr) * b;
    }
    void merge(const MsaJobSummary& o) {
        for (int k = 0; k < 3; ++k) {
            n[k] += o.n[k];
            lr[k] = o.lr[k] > lr[k] ? o.lr[k] : lr[k];
            lc[k] = o.lc[k] > lc[k] ? o.lc[k] : lc[k];
            band[k] = o.band[k] > band[k] ? o.band[k] : band[k];
            cols[k] += o.cols[k];
        }
        for (int k = 0; k < 2; ++k) wide[k].insert(wide[k].end(), o.wide[k].begin(), o.wide[k].end());
        wide_listed = wide_listed || o.wide_listed;
        cells += o.cells;
    }
};

// Launches the pairwise kernels for the njobs jobs at d_jobs on stream s.  `jobs` is the host's copy of the table, read
// for band classes only: it may be null when `summary` comes with the wide jobs listed (msa2.hip builds its table on the
// device and never holds one).  out_mode 0: ins/aln (spec v1), 1: maps + stats (spec v2).
int msa_pairwise_launch(const MsaJob* jobs, size_t njobs, const MsaJob* d_jobs, const uint8_t* d_seq,
                        double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
                        int out_mode, uint16_t* d_ins, uint8_t* d_aln, uint16_t* d_map, int2* d_stats, hipStream_t s,
                        const MsaJobSummary* summary = nullptr, bool reset_stuck = true);
inline int msa_pairwise_launch(const std::vector<MsaJob>& jobs, const MsaJob* d_jobs, const uint8_t* d_seq,
                               double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
                               int out_mode, uint16_t* d_ins, uint8_t* d_aln, uint16_t* d_map, int2* d_stats, hipStream_t s,
                               const MsaJobSummary* summary = nullptr, bool reset_stuck = true) {
    return msa_pairwise_launch(jobs.data(), jobs.size(), d_jobs, d_seq, match, mismatch, gap_extension, gap_opening, bandwidth,
                               out_mode, d_ins, d_aln, d_map, d_stats, s, summary, reset_stuck);
}
// (reset_stuck = false: the "traceback exceeded its step bound" flag of earlier launches of the same call is kept, the
// caller reads it once at the end -- pipelined batches, msa2.hip)

__device__ __forceinline__ uint8_t dna5_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
    }
    return 4;
}

// ---- vote codes (CodeSpec, common.hpp).  Table entry = quality index * 8 + slot: the vote kernel reads the four
// consecutive entries from the code's index; per quality the strip is (w w w r w w w w), bases in the order
// A, C, T, G read from slot 3 - base, any other character from slot 4 (wrong for all four); the row after the last
// quality is all zeros (N: counted as a base, adds nothing; gap: the same with bit 15 set). ----
constexpr int CODE_STRIP = 8;
constexpr unsigned CODE_GAPBIT = 0x8000u;
__host__ __device__ __forceinline__ unsigned code_zero_index(int navail) { return static_cast<unsigned>(navail) * CODE_STRIP; }
// ch: the row's character (upper case for aligned rows; verbatim for single-read groups); q: its quality character
__device__ __forceinline__ uint16_t vote_code(uint8_t ch, uint8_t q, int qoffset, int navail, bool& bad) {
    if (ch == 'N') return static_cast<uint16_t>(code_zero_index(navail));
    int qi = static_cast<int>(static_cast<signed char>(q)) - qoffset;
    if (qi < 0) { bad = true; qi = 0; }
    if (qi >= navail) qi = navail - 1;
    int slot = 4;                       // a character other than A, C, G, T, N
    switch (ch) {
        case 'A': slot = 3; break;      // base order of the vote kernel: A 0, C 1, T 2, G 3
        case 'C': slot = 2; break;
        case 'T': slot = 1; break;
        case 'G': slot = 0; break;
    }
    return static_cast<uint16_t>(qi * CODE_STRIP + slot);
}

}  // namespace sarlacc
