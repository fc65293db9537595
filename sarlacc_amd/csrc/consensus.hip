// consensus.hip -- per-column consensus vote over per-group alignments on gfx950.
//
// Replaces internal_create_consensus_basic / _quality and their _loop wrappers
// (/root/reference/src/create_consensus.cpp:61-135, :178-272, :150-170, :287-308)
// and errorsToString (:18-32).
//
// Mapping: one wavefront per alignment (group); lanes own 64 consecutive columns,
// rows are streamed in the reference's row order so every per-column fp64 sum adds
// the same terms in the same order (bit-identical sums => identical base calls).
// The position of a base inside its ungapped quality string is a running prefix
// count of non-gap characters: per 64-column chunk it is ballot + popcount, the
// per-row carry sits in LDS.  Kept columns are compacted with the same trick.
// log1p(-e) / log(e/3) come from host-built tables (host libm); the device only
// evaluates transcendentals for the per-column error (log1p / exp), and columns
// whose Phred value falls within 1e-9 of a rounding boundary are re-evaluated on
// the host with the host libm so the emitted Phred characters never depend on
// device-vs-host libm differences.
//
// This is the one streaming, HBM-bound stage of the path: 2 B in per alignment
// cell (gapped base + quality), 2 B out per kept column.
#include "msa_common.hpp"

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace sarlacc {

struct ConsArgs {
    const uint8_t* aln;
    const int64_t* aln_off;    // per row
    const int64_t* grp_rows;   // [ngroups+1]
    long long ngroups;
    const uint8_t* qual;       // quality mode only
    const int64_t* qual_off;   // per row (same row numbering as aln), or per read when row_read is set
    const int32_t* row_read;   // optional: 1-based read id of every row (qualities stay in read order)
    const double* right;       // [navail] log1p(-e)
    const double* wrong;       // [navail] log(e/3)
    const double* vec;         // k_consensus_q4: [(5 * navail + 1)][4] per-(read character, quality) additions to the A,C,G,T scores
    int qoffset, navail, max_rows;
    double mincov, pseudo, ln10;
    const int64_t* out_off;    // [ngroups] start of the group's output (= offset of its first row)
    uint8_t* cons;
    uint8_t* phred;
    double* lerr;              // optional
    int32_t* cons_len;         // [ngroups]
    int8_t* row_status;        // quality: 0 ok, 1 bad quality char, 2 shorter, 3 longer
    unsigned long long* first_bad_char;  // basic: min byte index of an unknown character
    // near-boundary columns to be resolved on the host
    int* fix_count;
    int fix_cap;
    long long* fix_pos;        // output byte index
    double* fix_val;           // 4 per entry: sorted scores (quality) or {max,total,0,0} (basic)
    int* fix_grp;              // k_consensus_qf: group of the entry (entries of groups handed to the generic kernel are void)
    // k_consensus_qf (fast path) and its hand-over to k_consensus_q4
    const double* strip;       // [(navail + 1)][QF_STRIP][QF_SLOTS]: per quality the strip (w w w r w w w), replicated; last row zeros
    int* gflag;                // per group: 1 = the fast kernel met something it does not handle; the generic kernel redoes the group
    int only_flagged;          // k_consensus_q4: skip groups whose flag is 0
    long long aln_bytes, qual_bytes;   // sizes of the two buffers (the fast kernel reads whole dwords)
    // k_consensus_code: the rows as 16-bit vote codes written by the MSA stage (CodeSpec, common.hpp)
    const uint16_t* codes;     // same offsets as aln, in cells
    const double* strip8;      // [(navail + 1)][CODE_STRIP][QF_SLOTS]: (w w w r w w w w) per quality, zero row last
    int* code_bad;             // smallest row with a quality below the encoding (INT_MAX: none)
};

__device__ __forceinline__ double dev_log1pexp(double x) {
    // R nmath log1pexp: x <= 18 -> log1p(exp x); 18 < x <= 33.3 -> x + exp(-x); else x
    if (x <= 18.) return log1p(exp(x));
    if (x > 33.3) return x;
    return x + exp(-x);
}

template <bool QUALITY>
__global__ void __launch_bounds__(64) k_consensus(const ConsArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_right = reinterpret_cast<double*>(smem);
    double* s_wrong = s_right + (QUALITY ? A.navail : 0);
    long long* s_roff = reinterpret_cast<long long*>(s_wrong + (QUALITY ? A.navail : 0));
    long long* s_qoff = s_roff + A.max_rows;
    int* s_pos = reinterpret_cast<int*>(s_qoff + A.max_rows);
    int* s_bad = s_pos + A.max_rows;
    int* s_qlen = s_bad + A.max_rows;
    constexpr int RB = 8;

    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (QUALITY)
        for (int x = lane; x < A.navail; x += 64) { s_right[x] = A.right[x]; s_wrong[x] = A.wrong[x]; }
    __syncthreads();

    for (long long g = blockIdx.x; g < A.ngroups; g += gridDim.x) {
        const long long row0 = A.grp_rows[g];
        const int nrows = static_cast<int>(A.grp_rows[g + 1] - row0);
        if (nrows == 0) {
            if (lane == 0) A.cons_len[g] = 0;
            continue;
        }
        const long long W = A.aln_off[row0 + 1] - A.aln_off[row0];
        const double thresh = static_cast<double>(nrows) * A.mincov;
        const long long obase = A.out_off[g];
        // per-row offsets once per group: no dependent scalar loads inside the column loop
        for (int r = lane; r < nrows; r += 64) {
            s_pos[r] = 0;
            s_bad[r] = 0;
            s_roff[r] = A.aln_off[row0 + r];
            if (QUALITY) {
                const long long q = A.row_read ? A.row_read[row0 + r] - 1 : row0 + r;
                s_qoff[r] = A.qual_off[q];
                s_qlen[r] = static_cast<int>(A.qual_off[q + 1] - A.qual_off[q]);
            }
        }
        __syncthreads();
        int outpos = 0;

        for (long long c0 = 0; c0 < W; c0 += 64) {
            const long long col = c0 + lane;
            const bool active = col < W;
            double sA = 0, sC = 0, sG = 0, sT = 0;
            int inc = 0;
            // rows in batches of RB: all alignment bytes of a batch are requested before any is
            // used, then all quality bytes -- two memory round trips per batch instead of per row.
            // Accumulation stays in row order, so the fp64 sums are the reference's.
            for (int r0 = 0; r0 < nrows; r0 += RB) {
                uint8_t ch[RB];
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int r = r0 + b;
                    ch[b] = (active && r < nrows) ? A.aln[s_roff[r < nrows ? r : 0] + col] : static_cast<uint8_t>('-');
                }
                int pq[RB];
                uint8_t qv8[RB];
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int r = r0 + b;
                    const bool nongap = ch[b] != '-';
                    const unsigned long long mask = __ballot(nongap);
                    pq[b] = -1;
                    if (QUALITY && r < nrows) {
                        const int p = s_pos[r] + __popcll(mask & lt);
                        if (lane == 0) s_pos[r] += __popcll(mask);
                        if (nongap && ch[b] != 'N' && p < s_qlen[r]) pq[b] = p;
                    }
                }
                if (QUALITY) {
#pragma unroll
                    for (int b = 0; b < RB; ++b) {
                        const int r = r0 + b;
                        qv8[b] = (pq[b] >= 0) ? A.qual[s_qoff[r < nrows ? r : 0] + pq[b]] : static_cast<uint8_t>(0);
                    }
                }
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int r = r0 + b;
                    const uint8_t c = ch[b];
                    if (c == '-') continue;
                    ++inc;
                    if (QUALITY) {
                        if (pq[b] >= 0) {
                            int qi = static_cast<int>(static_cast<signed char>(qv8[b])) - A.qoffset;
                            if (qi < 0) { s_bad[r] = 1; qi = 0; }
                            if (qi >= A.navail) qi = A.navail - 1;
                            const double right = s_right[qi], wrong = s_wrong[qi];
                            sA += (c == 'A') ? right : wrong;
                            sC += (c == 'C') ? right : wrong;
                            sG += (c == 'G') ? right : wrong;
                            sT += (c == 'T') ? right : wrong;
                        }
                    } else if (c != 'N') {
                        if (c == 'A') sA += 1;
                        else if (c == 'C') sC += 1;
                        else if (c == 'G') sG += 1;
                        else if (c == 'T') sT += 1;
                        else atomicMin(A.first_bad_char, static_cast<unsigned long long>(s_roff[r] + col));
                    }
                }
            }
            const bool keep = active && !(inc < thresh);
            // first maximum in A,C,G,T order (std::max_element)
            int best = 0;
            double bv = sA;
            if (sC > bv) { bv = sC; best = 1; }
            if (sG > bv) { bv = sG; best = 2; }
            if (sT > bv) { bv = sT; best = 3; }
            double le;
            double f0, f1, f2 = 0, f3 = 0;
            if (QUALITY) {
                // ascending sort of the four log-probabilities, then running log-sum-exp
                double a = sA, b = sC, c = sG, d = sT, t;
                if (a > b) { t = a; a = b; b = t; }
                if (c > d) { t = c; c = d; d = t; }
                if (a > c) { t = a; a = c; c = t; }
                if (b > d) { t = b; b = d; d = t; }
                if (b > c) { t = b; b = c; c = t; }
                double denom = a;
                denom += dev_log1pexp(b - denom);
                denom += dev_log1pexp(c - denom);
                const double err3 = denom;
                denom += dev_log1pexp(d - denom);
                le = err3 - denom;
                f0 = a; f1 = b; f2 = c; f3 = d;
            } else {
                int total = 0;
                total = static_cast<int>(total + sA);
                total = static_cast<int>(total + sC);
                total = static_cast<int>(total + sG);
                total = static_cast<int>(total + sT);
                const double p = (bv + A.pseudo / 4) / (static_cast<double>(total) + A.pseudo);
                le = log1p(-p);
                f0 = bv; f1 = static_cast<double>(total);
            }
            const double x = -10 * le / A.ln10;
            double qv = round(x);
            if (qv > 93.0) qv = 93.0;
            const double frac = x - floor(x);
            const bool near = keep && x < 93.4 && fabs(frac - 0.5) < 1e-9;

            const unsigned long long kmask = __ballot(keep);
            const int o = outpos + __popcll(kmask & lt);
            if (keep) {
                A.cons[obase + o] = "ACGT"[best];
                A.phred[obase + o] = static_cast<uint8_t>(static_cast<int>(qv) + 33);
                if (A.lerr) A.lerr[obase + o] = le;
                if (near) {
                    const int slot = atomicAdd(A.fix_count, 1);
                    if (slot < A.fix_cap) {
                        A.fix_pos[slot] = obase + o;
                        A.fix_val[4 * slot + 0] = f0; A.fix_val[4 * slot + 1] = f1;
                        A.fix_val[4 * slot + 2] = f2; A.fix_val[4 * slot + 3] = f3;
                    }
                }
            }
            outpos += __popcll(kmask);
        }
        if (lane == 0) A.cons_len[g] = outpos;
        if (QUALITY) {
            for (int r = lane; r < nrows; r += 64) {
                const int qlen = s_qlen[r];
                const int used = s_pos[r];
                A.row_status[row0 + r] = s_bad[r] ? 1 : (used == qlen ? 0 : (used > qlen ? 2 : 3));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_consensus_q4: the quality-weighted vote with 4 columns per lane.
//
// Same arithmetic as k_consensus<true> (rows in order, so every per-column fp64 sum adds the
// reference's terms in the reference's order) but laid out for bandwidth: a wavefront covers
// 256 columns per step, a lane reads its 4 alignment characters of a row with one (unaligned)
// dword load and, because the non-gap characters of a lane are consecutive in the ungapped
// quality string, their qualities with one more; rows are fetched in batches so several loads
// are in flight.  The (character, quality) -> four additions map is a table of 32-byte vectors
// in LDS shared by the 4 wavefronts of a workgroup (two ds_read_b128 + four v_add_f64 per cell;
// gaps, N and out-of-range positions read a zero vector, which leaves the sums bit-identical).
// The per-column error needs three log1pexp: they are first evaluated with the hardware fp32
// exp/log (absolute error < 1e-5 in the log-error), which decides the Phred character unless
// the value lies within 2e-4 of a rounding boundary; only those columns take the fp64 path,
// and columns within 1e-9 of a boundary go to the host libm as before.
typedef uint32_t __attribute__((aligned(1))) cons_u32_unaligned;

// exact fp64 log error of a column from its four sorted scores (out of line: rare, register-hungry)
__device__ __noinline__ double exact_log_error(double a, double b, double c, double d) {
    double denom = a;
    denom += dev_log1pexp(b - denom);
    denom += dev_log1pexp(c - denom);
    const double err3 = denom;
    denom += dev_log1pexp(d - denom);
    return err3 - denom;
}

__device__ __forceinline__ double fast_log1pexp(double x) {
    if (x > 33.3) return x;
    if (x > 18.) return x + static_cast<double>(__expf(static_cast<float>(-x)));
    const float y = __expf(static_cast<float>(x));
    return static_cast<double>(__logf(1.0f + y));
}

constexpr int Q4_RB = 5;   // rows fetched per batch

__global__ void __launch_bounds__(256, 4) k_consensus_q4(const ConsArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nvec = 5 * A.navail + 1;
    double* const s_vec = reinterpret_cast<double*>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long* const s_roff = reinterpret_cast<long long*>(s_vec + 4 * nvec) + wave * 2 * A.max_rows;
    long long* const s_qoff = s_roff + A.max_rows;
    int* const s_int = reinterpret_cast<int*>(reinterpret_cast<long long*>(s_vec + 4 * nvec) + 4 * 2 * A.max_rows) + wave * 3 * A.max_rows;
    int* const s_qlen = s_int;
    int* const s_pos = s_int + A.max_rows;
    int* const s_bad = s_int + 2 * A.max_rows;
    for (int x = threadIdx.x; x < 4 * nvec; x += 256) s_vec[x] = A.vec[x];
    __syncthreads();
    const int zero_entry = 5 * A.navail;

    for (long long g = static_cast<long long>(blockIdx.x) * 4 + wave; g < A.ngroups; g += static_cast<long long>(gridDim.x) * 4) {
        if (A.only_flagged && !A.gflag[g]) continue;
        const long long row0 = A.grp_rows[g];
        const int nrows = static_cast<int>(A.grp_rows[g + 1] - row0);
        if (nrows == 0) {
            if (lane == 0) A.cons_len[g] = 0;
            continue;
        }
        const long long W = A.aln_off[row0 + 1] - A.aln_off[row0];
        const double thresh = static_cast<double>(nrows) * A.mincov;
        const long long obase = A.out_off[g];
        for (int r = lane; r < nrows; r += 64) {
            s_pos[r] = 0;
            s_bad[r] = 0;
            s_roff[r] = A.aln_off[row0 + r];
            const long long q = A.row_read ? A.row_read[row0 + r] - 1 : row0 + r;
            s_qoff[r] = A.qual_off[q];
            s_qlen[r] = static_cast<int>(A.qual_off[q + 1] - A.qual_off[q]);
        }
        int outpos = 0;

        for (long long c0 = 0; c0 < W; c0 += 256) {
            const long long col = c0 + 4 * lane;
            const long long remain = W - col;          // characters of the row at and after `col`
            double acc[4][4];
            int inc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { inc[k] = 0; acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0; }

            for (int r0 = 0; r0 < nrows; r0 += Q4_RB) {
                uint32_t aw[Q4_RB], qw[Q4_RB];
                int pq[Q4_RB];
#pragma unroll
                for (int b = 0; b < Q4_RB; ++b) {
                    const int r = r0 + b;
                    uint32_t v = 0x2d2d2d2du;   // "----"
                    if (r < nrows && remain > 0) {
                        const uint8_t* src = A.aln + s_roff[r] + col;
                        if (remain >= 4) v = *reinterpret_cast<const cons_u32_unaligned*>(src);
                        else
                            for (int k = 0; k < remain; ++k) v = (v & ~(0xffu << (8 * k))) | (static_cast<uint32_t>(src[k]) << (8 * k));
                    }
                    aw[b] = v;
                }
#pragma unroll
                for (int b = 0; b < Q4_RB; ++b) {
                    const int r = r0 + b;
                    // position of this lane's first non-gap character in the row's ungapped string
                    int below = 0, total = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned long long m = __ballot(((aw[b] >> (8 * k)) & 0xffu) != 0x2du);
                        below += __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0));
                        total += __popcll(m);
                    }
                    pq[b] = 0;
                    if (r < nrows) {
                        pq[b] = s_pos[r] + below;
                        if (lane == 0) s_pos[r] += total;
                    }
                }
#pragma unroll
                for (int b = 0; b < Q4_RB; ++b) {
                    const int r = r0 + b;
                    uint32_t v = 0;
                    if (r < nrows) {
                        const int avail = s_qlen[r] - pq[b];
                        if (avail > 0) {
                            const uint8_t* src = A.qual + s_qoff[r] + pq[b];
                            if (avail >= 4) v = *reinterpret_cast<const cons_u32_unaligned*>(src);
                            else
                                for (int k = 0; k < avail; ++k) v |= static_cast<uint32_t>(src[k]) << (8 * k);
                        }
                    }
                    qw[b] = v;
                }
#pragma unroll
                for (int b = 0; b < Q4_RB; ++b) {
                    const int r = r0 + b;
                    if (r >= nrows) break;
                    const int avail = s_qlen[r] - pq[b];
                    int jq = 0;   // non-gap characters of this lane seen so far
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t c = (aw[b] >> (8 * k)) & 0xffu;
                        int entry = zero_entry;
                        if (c != 0x2du) {
                            ++inc[k];
                            if (c != 'N' && jq < avail) {
                                int qi = static_cast<int>(static_cast<signed char>((qw[b] >> (8 * jq)) & 0xffu)) - A.qoffset;
                                if (qi < 0) { s_bad[r] = 1; qi = 0; }
                                if (qi >= A.navail) qi = A.navail - 1;
                                const int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
                                entry = code * A.navail + qi;
                            }
                            ++jq;
                        }
                        const double* vp = s_vec + 4 * entry;
                        acc[k][0] += vp[0]; acc[k][1] += vp[1]; acc[k][2] += vp[2]; acc[k][3] += vp[3];
                    }
                }
            }

            // ---- per-column result ----
            int nkept = 0;
            int below_kept = 0;
            bool keepk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                keepk[k] = (k < remain) && !(inc[k] < thresh);
                const unsigned long long m = __ballot(keepk[k]);
                below_kept += __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0));
                nkept += __popcll(m);
            }
            int o = outpos + below_kept;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!keepk[k]) continue;
                const double sA = acc[k][0], sC = acc[k][1], sG = acc[k][2], sT = acc[k][3];
                int best = 0;
                double bv = sA;
                if (sC > bv) { bv = sC; best = 1; }
                if (sG > bv) { bv = sG; best = 2; }
                if (sT > bv) { bv = sT; best = 3; }
                double a = sA, b = sC, c = sG, d = sT, t;
                if (a > b) { t = a; a = b; b = t; }
                if (c > d) { t = c; c = d; d = t; }
                if (a > c) { t = a; a = c; c = t; }
                if (b > d) { t = b; b = d; d = t; }
                if (b > c) { t = b; b = c; c = t; }
                double denom = a;
                denom += fast_log1pexp(b - denom);
                denom += fast_log1pexp(c - denom);
                double err3 = denom;
                denom += fast_log1pexp(d - denom);
                double le = err3 - denom;
                double x = -10 * le / A.ln10;
                double frac = x - floor(x);
                bool near = false;
                if (x < 93.4 && fabs(frac - 0.5) < 2e-4) {   // too close for the fp32 estimate: exact fp64 evaluation
                    le = exact_log_error(a, b, c, d);
                    x = -10 * le / A.ln10;
                    frac = x - floor(x);
                    near = x < 93.4 && fabs(frac - 0.5) < 1e-9;
                }
                double qv = round(x);
                if (qv > 93.0) qv = 93.0;
                A.cons[obase + o] = "ACGT"[best];
                A.phred[obase + o] = static_cast<uint8_t>(static_cast<int>(qv) + 33);
                if (near) {
                    const int slot = atomicAdd(A.fix_count, 1);
                    if (slot < A.fix_cap) {
                        A.fix_pos[slot] = obase + o;
                        A.fix_val[4 * slot + 0] = a; A.fix_val[4 * slot + 1] = b;
                        A.fix_val[4 * slot + 2] = c; A.fix_val[4 * slot + 3] = d;
                    }
                }
                ++o;
            }
            outpos += nkept;
        }
        if (lane == 0) A.cons_len[g] = outpos;
        for (int r = lane; r < nrows; r += 64) {
            const int qlen = s_qlen[r];
            const int used = s_pos[r];
            A.row_status[row0 + r] = s_bad[r] ? 1 : (used == qlen ? 0 : (used > qlen ? 2 : 3));
        }
    }
}

// ---------------------------------------------------------------------------
// k_consensus_qf: the quality vote on clean data, built around the instruction count per cell.
//
// Same sums in the same order as k_consensus_q4 (4 columns per lane, rows in order).  What is different:
//  * everything that can be done on the four characters of a lane at once is done on the dword: gap
//    detection, non-gap count, the position of every character in the quality string (byte-wise prefix
//    by one multiplication), the expansion of the lane's quality bytes to the slots of its characters
//    (one v_perm), range and alphabet checks;
//  * the wave prefix of the non-gap counts is one DPP scan for three rows (10-bit fields);
//  * the four addends of a cell come from LDS with ONE address: per quality value the table holds the
//    strip (w w w r w w w) of log(e/3) and log1p(-e); a cell of base code c reads strip[3 - c + b] for
//    base b = 0..3, i.e. four ds_read_b64 at immediate offsets from (row of q) + (3 - c) -- no select.
//    Base codes are (char >> 1) & 3: A 0, C 1, T 2, G 3.  Every double is replicated for 16 lane slots.
//    Gaps read the all-zero row (adding +0.0 leaves the sums bit-identical).
// It handles alignments of A, C, G, T and '-' with qualities inside the encoding, at most 64 rows stored
// back to back; anything else (N or other characters, a quality outside the table, quality strings
// shorter or longer than their rows, the last bytes of a buffer) sets the group's flag and the group is
// redone by k_consensus_q4 -- detection costs a few SWAR operations per dword and is exact.
constexpr int QF_SLOTS = 16;
constexpr int QF_STRIP = 7;
constexpr int QF_ROWB = QF_STRIP * QF_SLOTS * 8;   // 896 bytes per quality value
constexpr int QF_RB = 6;                           // rows per batch: two scans of three packed counts
constexpr int QF_THREADS = 1024;

__device__ __forceinline__ unsigned qf_scan(unsigned x) {   // inclusive prefix sum over the wavefront
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xf, 0xf, false));   // row_shr:1
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xf, 0xf, false));   // row_shr:2
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xf, 0xf, false));   // row_shr:4
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xf, 0xf, false));   // row_shr:8
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xa, 0xf, false));   // row_bcast:15
    x += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xc, 0xf, false));   // row_bcast:31
    return x;
}

__global__ void __launch_bounds__(QF_THREADS) k_consensus_qf(const ConsArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    {
        const uint4* src = reinterpret_cast<const uint4*>(A.strip);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const int n16 = (A.navail + 1) * (QF_ROWB / 16);
        for (int x = threadIdx.x; x < n16; x += QF_THREADS) dst[x] = src[x];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* const lanebase = smem + (lane & 15) * 8;
    const unsigned DASH4 = 0x2d2d2d2du;
    const unsigned QOFF4 = static_cast<unsigned>(A.qoffset) * 0x01010101u;
    const unsigned QZERO4 = static_cast<unsigned>(A.qoffset + A.navail) * 0x01010101u;   // the quality "character" of the zero row
    const unsigned KHI = static_cast<unsigned>(0x80 - A.navail) * 0x01010101u;           // q + KHI has bit 7 set iff q >= navail
    const unsigned LUT = 0x47544341u;                                                    // code -> character: A C T G

    for (long long g = static_cast<long long>(blockIdx.x) * (QF_THREADS / 64) + wave; g < A.ngroups;
         g += static_cast<long long>(gridDim.x) * (QF_THREADS / 64)) {
        const long long row0 = A.grp_rows[g];
        const int nrows = static_cast<int>(A.grp_rows[g + 1] - row0);
        if (nrows == 0) {
            if (lane == 0) { A.cons_len[g] = 0; A.gflag[g] = 0; }
            continue;
        }
        const long long abase = A.aln_off[row0];
        const long long W = A.aln_off[row0 + 1] - abase;
        // row descriptors: lane r holds row r
        long long qo = 0;
        int qlen = 0;
        bool rowbad = false;
        if (lane < nrows && nrows <= 64) {
            const long long q = A.row_read ? A.row_read[row0 + lane] - 1 : row0 + lane;
            qo = A.qual_off[q];
            qlen = static_cast<int>(A.qual_off[q + 1] - qo);
            rowbad = A.aln_off[row0 + lane] != abase + static_cast<long long>(lane) * W || qo + qlen + 4 > A.qual_bytes;
        }
        if (nrows > 64 || W >= (1ll << 30) || abase + static_cast<long long>(nrows) * W + 4 > A.aln_bytes || __ballot(rowbad)) {
            if (lane == 0) A.gflag[g] = 1;
            continue;
        }
        const int qo_lo = static_cast<int>(qo), qo_hi = static_cast<int>(qo >> 32);
        int vpos = 0;
        unsigned bad = 0;
        const double thresh = static_cast<double>(nrows) * A.mincov;
        const long long obase = A.out_off[g];
        int outpos = 0;

        unsigned vn[QF_RB];
#pragma unroll
        for (int b = 0; b < QF_RB; ++b) {
            unsigned w = DASH4;
            if (b < nrows && 4 * lane < static_cast<int>(W)) w = *reinterpret_cast<const cons_u32_unaligned*>(A.aln + abase + static_cast<long long>(b) * W + 4 * lane);
            vn[b] = w;
        }
        for (long long c0 = 0; c0 < W; c0 += 256) {
            const int col = static_cast<int>(c0) + 4 * lane;
            const int remain = static_cast<int>(W) - col;          // characters of the row at and after `col`
            const unsigned keepmask = remain >= 4 ? 0xffffffffu : (remain <= 0 ? 0u : ((1u << (8 * remain)) - 1u));
            double acc[4][4];
            unsigned inc4 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;

            for (int r0 = 0; r0 < nrows; r0 += QF_RB) {
                unsigned v[QF_RB], ng[QF_RB], qw[QF_RB];
#pragma unroll
                for (int b = 0; b < QF_RB; ++b) v[b] = (vn[b] & keepmask) | (DASH4 & ~keepmask);   // requested one batch ahead
                unsigned pk[2] = {0u, 0u};
#pragma unroll
                for (int b = 0; b < QF_RB; ++b) {
                    const unsigned x = v[b] ^ DASH4;
                    const unsigned t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;
                    ng[b] = t & 0x80808080u;                        // bit 7 of every non-gap byte
                    pk[b / 3] |= static_cast<unsigned>(__builtin_popcount(ng[b])) << (10 * (b % 3));
                }
                const unsigned sc0 = qf_scan(pk[0]), sc1 = qf_scan(pk[1]);
                const unsigned tot0 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(sc0), 63));
                const unsigned tot1 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(sc1), 63));
#pragma unroll
                for (int b = 0; b < QF_RB; ++b) {
                    const int r = r0 + b;
                    const int rr = r < 63 ? r : 63;
                    const unsigned incl = ((b < 3 ? sc0 : sc1) >> (10 * (b % 3))) & 0x3ffu;
                    const unsigned total = ((b < 3 ? tot0 : tot1) >> (10 * (b % 3))) & 0x3ffu;
                    const int n = __builtin_popcount(ng[b]);
                    const int carry = __builtin_amdgcn_readlane(vpos, rr);
                    const int ql = __builtin_amdgcn_readlane(qlen, rr);
                    const int pos = carry + static_cast<int>(incl) - n;
                    if (r < nrows && lane == rr) vpos = carry + static_cast<int>(total);
                    bad |= (r < nrows && pos + n > ql) ? 0x80u : 0u;
                    unsigned w = QZERO4;
                    if (r < nrows && n > 0) {
                        const long long qb = (static_cast<long long>(__builtin_amdgcn_readlane(qo_hi, rr)) << 32) |
                                             static_cast<unsigned>(__builtin_amdgcn_readlane(qo_lo, rr));
                        w = *reinterpret_cast<const cons_u32_unaligned*>(A.qual + qb + (pos < ql ? pos : ql));
                    }
                    qw[b] = w;
                }
                {   // characters of the next batch (next rows of this step, or the first rows of the next step)
                    const bool wrap = r0 + QF_RB >= nrows;
                    const int nr0 = wrap ? 0 : r0 + QF_RB;
                    const int ncol = wrap ? col + 256 : col;
#pragma unroll
                    for (int b = 0; b < QF_RB; ++b) {
                        const int r = nr0 + b;
                        unsigned w = DASH4;
                        if (r < nrows && ncol < static_cast<int>(W)) w = *reinterpret_cast<const cons_u32_unaligned*>(A.aln + abase + static_cast<long long>(r) * W + ncol);
                        vn[b] = w;
                    }
                }
#pragma unroll
                for (int b = 0; b < QF_RB; ++b) {
                    if (r0 + b >= nrows) break;
                    const unsigned one = ng[b] >> 7;                                     // 1 per non-gap byte
                    const unsigned ff = (ng[b] << 1) - one;                              // 0xff per non-gap byte
                    const unsigned presel = (one << 8) + (one << 16) + (one << 24);      // byte k: non-gap bytes below k
                    const unsigned sel = (presel & ff) | (0x04040404u & ~ff);
                    const unsigned qe = __builtin_amdgcn_perm(QZERO4, qw[b], sel);       // quality of every character, zero row for gaps
                    const unsigned qi = qe - QOFF4;
                    bad |= (qi | (qi + KHI)) & ng[b];
                    const unsigned selw = (v[b] >> 1) & 0x03030303u;                     // base codes
                    bad |= (__builtin_amdgcn_perm(LUT, LUT, selw) ^ v[b]) & ff;          // characters other than A, C, G, T
                    inc4 += one;
                    const unsigned tst = 0x03030303u - selw;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned off = __umul24((qi >> (8 * k)) & 0xffu, QF_ROWB) + (((tst >> (8 * k)) & 0xffu) << 7);
                        const unsigned char* cell = lanebase + off;
                        acc[k][0] += *reinterpret_cast<const double*>(cell);
                        acc[k][1] += *reinterpret_cast<const double*>(cell + 128);
                        acc[k][2] += *reinterpret_cast<const double*>(cell + 256);
                        acc[k][3] += *reinterpret_cast<const double*>(cell + 384);
                    }
                }
            }

            // ---- per-column result (accumulators are in code order A, C, T, G) ----
            int nkept = 0;
            int below_kept = 0;
            bool keepk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int inck = static_cast<int>((inc4 >> (8 * k)) & 0xffu);
                keepk[k] = (k < remain) && !(inck < thresh);
                const unsigned long long m = __ballot(keepk[k]);
                below_kept += __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0));
                nkept += __popcll(m);
            }
            int o = outpos + below_kept;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!keepk[k]) continue;
                const double sA = acc[k][0], sC = acc[k][1], sG = acc[k][3], sT = acc[k][2];
                // first maximum in the order A, C, G, T (std::max_element)
                const double mx = fmax(fmax(sA, sC), fmax(sG, sT));
                const int best = sA == mx ? 0 : (sC == mx ? 1 : (sG == mx ? 2 : 3));
                // Estimate of the Phred value in fp32: with u = score - max, S = sum of exp(u) over the three other
                // bases, the log error is log(S / (1 + S)) and the Phred value 10 log10(2) (log2(1 + S) - log2(S)).
                // Mathematically the value of the reference's log1pexp chain; the fp32 error (< 1e-4 in the Phred
                // value) only matters next to a rounding boundary, where the exact chain is evaluated instead.
                const float NEG = -1.0e30f;
                const float uA = best == 0 ? NEG : static_cast<float>(sA - mx), uC = best == 1 ? NEG : static_cast<float>(sC - mx);
                const float uG = best == 2 ? NEG : static_cast<float>(sG - mx), uT = best == 3 ? NEG : static_cast<float>(sT - mx);
                const float L2E = 1.44269504088896341f;
                const float S = (__builtin_amdgcn_exp2f(uA * L2E) + __builtin_amdgcn_exp2f(uC * L2E)) +
                                (__builtin_amdgcn_exp2f(uG * L2E) + __builtin_amdgcn_exp2f(uT * L2E));
                float xf = 3.01029995663981195f * (__builtin_amdgcn_logf(1.0f + S) - __builtin_amdgcn_logf(S));   // v_log_f32 = log2
                float fr = xf - floorf(xf);
                int qv;
                bool near = false;
                double a = 0, b = 0, c = 0, d = 0;
                if (xf < 93.4f && fabsf(fr - 0.5f) < 4e-4f) {     // too close for the estimate: the reference's chain in fp64
                    double t;
                    a = sA; b = sC; c = sG; d = sT;
                    if (a > b) { t = a; a = b; b = t; }
                    if (c > d) { t = c; c = d; d = t; }
                    if (a > c) { t = a; a = c; c = t; }
                    if (b > d) { t = b; b = d; d = t; }
                    if (b > c) { t = b; b = c; c = t; }
                    const double le = exact_log_error(a, b, c, d);
                    const double x = -10 * le / A.ln10;
                    const double frac = x - floor(x);
                    near = x < 93.4 && fabs(frac - 0.5) < 1e-9;
                    double q = round(x);
                    if (q > 93.0) q = 93.0;
                    qv = static_cast<int>(q);
                } else {
                    xf = fminf(xf, 93.0f);                          // also takes +inf (S underflowed to 0)
                    qv = static_cast<int>(floorf(xf + 0.5f));
                }
                A.cons[obase + o] = "ACGT"[best];
                A.phred[obase + o] = static_cast<uint8_t>(qv + 33);
                if (near) {
                    const int slot = atomicAdd(A.fix_count, 1);
                    if (slot < A.fix_cap) {
                        A.fix_pos[slot] = obase + o;
                        A.fix_grp[slot] = static_cast<int>(g);
                        A.fix_val[4 * slot + 0] = a; A.fix_val[4 * slot + 1] = b;
                        A.fix_val[4 * slot + 2] = c; A.fix_val[4 * slot + 3] = d;
                    }
                }
                ++o;
            }
            outpos += nkept;
        }
        // every quality string consumed exactly, nothing unusual seen: the group is done
        const bool dirty = __ballot(bad != 0u || (lane < nrows && vpos != qlen)) != 0ull;
        if (lane == 0) {
            A.gflag[g] = dirty ? 1 : 0;
            if (!dirty) A.cons_len[g] = outpos;
        }
    }
}

// ---------------------------------------------------------------------------
// k_consensus_code: the quality vote on rows the MSA stage wrote as 16-bit vote codes (fused multiReadAlign +
// consensusReadSeq).  A code is the index of the cell's (quality, base) entry in the LDS table -- the row writers
// know the read position of every cell, so the quality lookup, the gap / N / alphabet handling and the clamping
// are theirs -- with bit 15 set for a gap.  What is left per cell: one address (code * 128 + the lane's slot), four
// 8-byte LDS reads at immediate offsets, four additions in row order; per row and lane one 8-byte load of four
// codes, independent of everything else, so the next batch of rows is always in flight.  Same sums in the same
// order as the other kernels.
constexpr int QC_RB = 5;
constexpr int QC_ROWB = CODE_STRIP * QF_SLOTS * 8;   // 1024 bytes per quality value
typedef unsigned long long __attribute__((aligned(1))) cons_u64_unaligned;

// The rows of one 256-column step: per row and lane one 8-byte load of four codes, requested one batch of rows
// ahead; per cell one address, four 8-byte LDS reads, four additions.  Loads are unconditional -- rows past the last
// one re-read it, lanes past the row end read its start, both masked afterwards -- so that the compiler can count them
// and wait only for the batch it is about to use.  MASK: the step contains the row end (cells beyond it become gaps).
template <bool MASK>
__device__ __forceinline__ void qc_rows(const uint16_t* rows, long long W, int nrows, int col, int remain, const unsigned char* lanebase,
                                        unsigned long long GAP4, double (&acc)[4][4], unsigned& gaps01, unsigned& gaps23) {
    const unsigned long long keep = !MASK || remain >= 4 ? ~0ull : (remain <= 0 ? 0ull : ((1ull << (16 * remain)) - 1ull));
    const int lcol = (!MASK || remain > 0) ? col : 0;
    unsigned long long wn[QC_RB];
#pragma unroll
    for (int b = 0; b < QC_RB; ++b) wn[b] = *reinterpret_cast<const cons_u64_unaligned*>(rows + static_cast<long long>(min(b, nrows - 1)) * W + lcol);
    for (int r0 = 0; r0 < nrows; r0 += QC_RB) {
        unsigned long long w[QC_RB];
#pragma unroll
        for (int b = 0; b < QC_RB; ++b) w[b] = MASK ? ((wn[b] & keep) | (GAP4 & ~keep)) : wn[b];
#pragma unroll
        for (int b = 0; b < QC_RB; ++b)                 // the next batch (nothing depends on the data)
            wn[b] = *reinterpret_cast<const cons_u64_unaligned*>(rows + static_cast<long long>(min(r0 + QC_RB + b, nrows - 1)) * W + lcol);
#pragma unroll
        for (int b = 0; b < QC_RB; ++b) {
            if (r0 + b >= nrows) break;
            const unsigned lo = static_cast<unsigned>(w[b]), hi = static_cast<unsigned>(w[b] >> 32);
            gaps01 += (lo >> 15) & 0x00010001u;
            gaps23 += (hi >> 15) & 0x00010001u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned h = ((k < 2 ? lo : hi) >> (16 * (k & 1))) & 0x7fffu;
                // four single 8-byte reads (2 LDS cycles each); merged into two ds_read2_b64 they take 8 cycles each
                // (MI355X_MICROARCH.md, LDS table) -- volatile keeps the compiler from pairing them
                typedef const volatile double __attribute__((address_space(3))) lds_cvd;
                lds_cvd* cell = reinterpret_cast<lds_cvd*>(reinterpret_cast<uintptr_t>(lanebase + (h << 7)) & 0xffffffffu);
                const double t0 = cell[0], t1 = cell[16], t2 = cell[32], t3 = cell[48];
                acc[k][0] += t0; acc[k][1] += t1; acc[k][2] += t2; acc[k][3] += t3;
            }
        }
    }
}

__global__ void __launch_bounds__(QF_THREADS) k_consensus_code(const ConsArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    {
        const uint4* src = reinterpret_cast<const uint4*>(A.strip8);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const int n16 = (A.navail + 1) * (QC_ROWB / 16);
        for (int x = threadIdx.x; x < n16; x += QF_THREADS) dst[x] = src[x];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* const lanebase = smem + (lane & 15) * 8;
    const unsigned gap1 = CODE_GAPBIT | code_zero_index(A.navail);
    const unsigned long long GAP4 = gap1 * 0x0001000100010001ull;

    for (long long g = static_cast<long long>(blockIdx.x) * (QF_THREADS / 64) + wave; g < A.ngroups;
         g += static_cast<long long>(gridDim.x) * (QF_THREADS / 64)) {
        const long long row0 = A.grp_rows[g];
        const int nrows = static_cast<int>(A.grp_rows[g + 1] - row0);
        if (nrows == 0) {
            if (lane == 0) A.cons_len[g] = 0;
            continue;
        }
        const long long cbase = A.aln_off[row0];
        const long long W = A.aln_off[row0 + 1] - cbase;
        const double thresh = static_cast<double>(nrows) * A.mincov;
        const long long obase = A.out_off[g];
        int outpos = 0;
        const uint16_t* const rows = A.codes + cbase;

        for (long long c0 = 0; c0 < W; c0 += 256) {
            const int col = static_cast<int>(c0) + 4 * lane;
            const int remain = static_cast<int>(W) - col;          // cells of the row at and after `col`
            double acc[4][4];
            unsigned gaps01 = 0, gaps23 = 0;                        // packed 16-bit gap counts of the lane's four columns
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
            // the rows of this step: a variant without the row-end mask for the steps that lie wholly inside the row
            if (c0 + 256 <= W) qc_rows<false>(rows, W, nrows, col, remain, lanebase, GAP4, acc, gaps01, gaps23);
            else qc_rows<true>(rows, W, nrows, col, remain, lanebase, GAP4, acc, gaps01, gaps23);

            // ---- per-column result (accumulators are in code order A, C, T, G) ----
            int nkept = 0;
            int below_kept = 0;
            bool keepk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int inck = nrows - static_cast<int>(((k < 2 ? gaps01 : gaps23) >> (16 * (k & 1))) & 0xffffu);
                keepk[k] = (k < remain) && !(inck < thresh);
                const unsigned long long m = __ballot(keepk[k]);
                below_kept += __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0));
                nkept += __popcll(m);
            }
            int o = outpos + below_kept;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!keepk[k]) continue;
                const double sA = acc[k][0], sC = acc[k][1], sG = acc[k][3], sT = acc[k][2];
                const double mx = fmax(fmax(sA, sC), fmax(sG, sT));
                const int best = sA == mx ? 0 : (sC == mx ? 1 : (sG == mx ? 2 : 3));
                // fp32 estimate of the Phred value, the reference's chain in fp64 next to a rounding boundary (see k_consensus_qf)
                const float NEG = -1.0e30f;
                const float uA = best == 0 ? NEG : static_cast<float>(sA - mx), uC = best == 1 ? NEG : static_cast<float>(sC - mx);
                const float uG = best == 2 ? NEG : static_cast<float>(sG - mx), uT = best == 3 ? NEG : static_cast<float>(sT - mx);
                const float L2E = 1.44269504088896341f;
                const float S = (__builtin_amdgcn_exp2f(uA * L2E) + __builtin_amdgcn_exp2f(uC * L2E)) +
                                (__builtin_amdgcn_exp2f(uG * L2E) + __builtin_amdgcn_exp2f(uT * L2E));
                float xf = 3.01029995663981195f * (__builtin_amdgcn_logf(1.0f + S) - __builtin_amdgcn_logf(S));
                const float fr = xf - floorf(xf);
                int qv;
                bool near = false;
                double a = 0, b = 0, c = 0, d = 0;
                if (xf < 93.4f && fabsf(fr - 0.5f) < 4e-4f) {
                    double t;
                    a = sA; b = sC; c = sG; d = sT;
                    if (a > b) { t = a; a = b; b = t; }
                    if (c > d) { t = c; c = d; d = t; }
                    if (a > c) { t = a; a = c; c = t; }
                    if (b > d) { t = b; b = d; d = t; }
                    if (b > c) { t = b; b = c; c = t; }
                    const double le = exact_log_error(a, b, c, d);
                    const double x = -10 * le / A.ln10;
                    const double frac = x - floor(x);
                    near = x < 93.4 && fabs(frac - 0.5) < 1e-9;
                    double q = round(x);
                    if (q > 93.0) q = 93.0;
                    qv = static_cast<int>(q);
                } else {
                    xf = fminf(xf, 93.0f);
                    qv = static_cast<int>(floorf(xf + 0.5f));
                }
                A.cons[obase + o] = "ACGT"[best];
                A.phred[obase + o] = static_cast<uint8_t>(qv + 33);
                if (near) {
                    const int slot = atomicAdd(A.fix_count, 1);
                    if (slot < A.fix_cap) {
                        A.fix_pos[slot] = obase + o;
                        A.fix_val[4 * slot + 0] = a; A.fix_val[4 * slot + 1] = b;
                        A.fix_val[4 * slot + 2] = c; A.fix_val[4 * slot + 3] = d;
                    }
                }
                ++o;
            }
            outpos += nkept;
        }
        if (lane == 0) A.cons_len[g] = outpos;
    }
}

// kept columns of every alignment -> contiguous output (one block per alignment)
__global__ void k_consensus_compact(const uint8_t* cons, const uint8_t* phred, const double* lerr, const int64_t* out_off,
                                    const int32_t* len, const long long* dst_off, long long ngroups, uint8_t* dcons,
                                    uint8_t* dphred, double* dlerr) {
    const long long g = blockIdx.x;
    if (g >= ngroups) return;
    const long long src = out_off[g], dst = dst_off[g];
    for (int p = threadIdx.x; p < len[g]; p += blockDim.x) {
        dcons[dst + p] = cons[src + p];
        dphred[dst + p] = phred[src + p];
        if (lerr) dlerr[dst + p] = lerr[src + p];
    }
}

// ---------------------------------------------------------------------------
static double host_log1pexp(double x) {
    if (x <= 18.) return std::log1p(std::exp(x));
    if (x > 33.3) return x;
    return x + std::exp(-x);
}

static char phred_char(double le) {
    const double q = std::min(std::round(-10 * le / std::log(10)), 93.0);
    return static_cast<char>(static_cast<int>(q) + 33);
}

// Device part of the consensus: `a` arrives with the alignment rows, their offsets, the group
// table, the output offsets and (quality mode) the quality strings already in HBM; this adds
// the tables and scratch, launches the vote and resolves results and errors on the host.
// aln_host (optional) is the host copy of the rows, used only to quote an unknown character.
static int consensus_core(bool quality, ConsArgs a, int64_t ngroups, int64_t ng_eval, int64_t rows_eval, int64_t total,
                          const std::vector<int64_t>& out_off, const char* aln_host, int struct_err_kind,
                          double min_cov, double pseudo, const double* enc_errors, const char* enc_names, int enc_n,
                          char* cons, char* phred, int64_t* cons_off, double* lerr, hipStream_t s, int64_t cons_cap = -1) {
    Context& c = ctx();
    int64_t* d_ooff = const_cast<int64_t*>(a.out_off);
    std::vector<double> right, wrong;
    if (quality) {
        double* d_r; double* d_w;
        right.resize(enc_n);
        wrong.resize(enc_n);
        for (int k = 0; k < enc_n; ++k) {  // (src/create_consensus.cpp:14,:218-226)
            double e = enc_errors[k];
            if (e > 0.99999999) e = 0.99999999;
            else if (e < 0.00000001) e = 0.00000001;
            right[k] = std::log1p(-e);
            wrong[k] = std::log(e / 3);
        }
        SL_TRY(upload("cons.right", right.data(), right.size(), &d_r, s));
        SL_TRY(upload("cons.wrong", wrong.data(), wrong.size(), &d_w, s));
        a.right = d_r; a.wrong = d_w;
        // k_consensus_q4: what a cell adds to (A, C, G, T) by read character (A, C, G, T, other) and
        // quality; the last entry (gap, N, no quality left) adds nothing
        std::vector<double> vec(static_cast<size_t>(5 * enc_n + 1) * 4, 0.0);
        for (int code = 0; code < 5; ++code)
            for (int k = 0; k < enc_n; ++k)
                for (int bidx = 0; bidx < 4; ++bidx)
                    vec[(static_cast<size_t>(code) * enc_n + k) * 4 + bidx] = (bidx == code) ? right[k] : wrong[k];
        double* d_vec;
        SL_TRY(upload("cons.vec", vec.data(), vec.size(), &d_vec, s));
        a.vec = d_vec;
        a.qoffset = static_cast<int>(enc_names[0]); a.navail = enc_n;
        // k_consensus_qf: per quality the strip (w w w r w w w), every double replicated for QF_SLOTS lanes; zero row last
        std::vector<double> strip(static_cast<size_t>(enc_n + 1) * QF_STRIP * QF_SLOTS, 0.0);
        for (int k = 0; k < enc_n; ++k)
            for (int i = 0; i < QF_STRIP; ++i)
                for (int sl = 0; sl < QF_SLOTS; ++sl)
                    strip[(static_cast<size_t>(k) * QF_STRIP + i) * QF_SLOTS + sl] = (i == 3) ? right[k] : wrong[k];
        double* d_strip;
        SL_TRY(upload("cons.strip", strip.data(), strip.size(), &d_strip, s));
        a.strip = d_strip;
        if (a.codes) {   // k_consensus_code: (w w w r w w w w) per quality, zero row last
            std::vector<double> strip8(static_cast<size_t>(enc_n + 1) * CODE_STRIP * QF_SLOTS, 0.0);
            for (int k = 0; k < enc_n; ++k)
                for (int i = 0; i < CODE_STRIP; ++i)
                    for (int sl = 0; sl < QF_SLOTS; ++sl)
                        strip8[(static_cast<size_t>(k) * CODE_STRIP + i) * QF_SLOTS + sl] = (i == 3) ? right[k] : wrong[k];
            double* d_strip8;
            SL_TRY(upload("cons.strip8", strip8.data(), strip8.size(), &d_strip8, s));
            a.strip8 = d_strip8;
        }
    }
    a.ngroups = ng_eval;
    a.mincov = min_cov; a.pseudo = pseudo; a.ln10 = std::log(10);
    const int max_rows = a.max_rows;

    uint8_t* d_cons; uint8_t* d_phred; double* d_lerr = nullptr; int32_t* d_len; int8_t* d_status;
    unsigned long long* d_badchar; int* d_fixn; long long* d_fixpos; double* d_fixval;
    const int fix_cap = 4096;
    SL_TRY(scratch("cons.out", static_cast<size_t>(total), &d_cons));
    SL_TRY(scratch("cons.phred", static_cast<size_t>(total), &d_phred));
    if (lerr) SL_TRY(scratch("cons.lerr", static_cast<size_t>(total), &d_lerr));
    SL_TRY(scratch("cons.len", static_cast<size_t>(ngroups), &d_len));
    SL_TRY(scratch("cons.status", static_cast<size_t>(std::max<int64_t>(rows_eval, 1)), &d_status));
    SL_TRY(scratch("cons.badchar", 1, &d_badchar));
    SL_TRY(scratch("cons.fixn", 1, &d_fixn));
    SL_TRY(scratch("cons.fixpos", fix_cap, &d_fixpos));
    SL_TRY(scratch("cons.fixval", 4 * fix_cap, &d_fixval));
    SL_HIP(hipMemsetAsync(d_status, 0, static_cast<size_t>(std::max<int64_t>(rows_eval, 1)), s));
    SL_HIP(hipMemsetAsync(d_badchar, 0xff, sizeof(unsigned long long), s));
    SL_HIP(hipMemsetAsync(d_fixn, 0, sizeof(int), s));
    a.cons = d_cons; a.phred = d_phred; a.lerr = d_lerr; a.cons_len = d_len; a.row_status = d_status;
    a.first_bad_char = d_badchar; a.fix_count = d_fixn; a.fix_cap = fix_cap; a.fix_pos = d_fixpos; a.fix_val = d_fixval;
    int* d_fixgrp; int* d_gflag;
    SL_TRY(scratch("cons.fixgrp", fix_cap, &d_fixgrp));
    SL_TRY(scratch("cons.gflag", static_cast<size_t>(std::max<int64_t>(ngroups, 1)), &d_gflag));
    SL_HIP(hipMemsetAsync(d_fixgrp, 0xff, sizeof(int) * fix_cap, s));
    a.fix_grp = d_fixgrp; a.gflag = d_gflag; a.only_flagged = 0;

    if (ng_eval > 0) {
        const size_t lds = (quality ? 2 * sizeof(double) * enc_n : 0) + (2 * sizeof(long long) + 3 * sizeof(int)) * static_cast<size_t>(max_rows) + 16;
        if (lds > 160 * 1024) return fail("sarlacc_amd: alignment with %d rows does not fit the consensus kernel", max_rows);
        const int grid = static_cast<int>(std::min<int64_t>(ng_eval, static_cast<int64_t>(c.num_cu) * 32));
        // quality vote without the per-column log errors: 4 columns per lane, 4 alignments per workgroup
        const size_t lds4 = sizeof(double) * 4 * (5 * static_cast<size_t>(enc_n) + 1) + 4 * (2 * sizeof(long long) + 3 * sizeof(int)) * static_cast<size_t>(max_rows) + 16;
        const bool q4 = quality && !lerr && lds4 <= 48 * 1024;
        SL_HIP(hipEventRecord(c.ev_start, s));
        c.counts["consensus_cells"] = static_cast<double>(total);
        c.stage_reset("consensus");
        SL_TRY(c.stage_begin("consensus", s));
        if (a.codes) {
            // rows written as vote codes by the MSA stage: nothing to decode
            const size_t ldsc = static_cast<size_t>(enc_n + 1) * QC_ROWB;
            SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_consensus_code), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsc)));
            const int gridc = static_cast<int>(std::min<int64_t>((ng_eval + QF_THREADS / 64 - 1) / (QF_THREADS / 64), c.num_cu));
            hipLaunchKernelGGL(k_consensus_code, dim3(gridc), dim3(QF_THREADS), ldsc, s, a);
        } else if (q4) {
            // clean groups on the fast kernel (one 1024-thread workgroup per CU around a ~85 KB strip table in LDS);
            // whatever it flags is redone by the generic kernel
            const size_t ldsf = static_cast<size_t>(enc_n + 1) * QF_ROWB;
            const bool qf = a.aln_bytes > 0 && a.qual_bytes > 0 && enc_n <= 127 && a.qoffset + enc_n <= 255 && ldsf <= 150 * 1024 &&
                            !option(OPT_CONSENSUS_GENERIC);
            if (qf) {
                SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_consensus_qf), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsf)));
                const int gridf = static_cast<int>(std::min<int64_t>((ng_eval + QF_THREADS / 64 - 1) / (QF_THREADS / 64), c.num_cu));
                hipLaunchKernelGGL(k_consensus_qf, dim3(gridf), dim3(QF_THREADS), ldsf, s, a);
                SL_HIP(hipGetLastError());
                a.only_flagged = 1;
            }
            const int grid4 = static_cast<int>(std::min<int64_t>((ng_eval + 3) / 4, static_cast<int64_t>(c.num_cu) * 64));
            hipLaunchKernelGGL(k_consensus_q4, dim3(grid4), dim3(256), lds4, s, a);
        } else {
            if (lds > 48 * 1024) {   // alignments of thousands of rows: more than the default 64 KB of dynamic LDS
                SL_HIP(hipFuncSetAttribute(quality ? reinterpret_cast<const void*>(&k_consensus<true>) : reinterpret_cast<const void*>(&k_consensus<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            }
            if (quality) hipLaunchKernelGGL(k_consensus<true>, dim3(grid), dim3(64), lds, s, a);
            else hipLaunchKernelGGL(k_consensus<false>, dim3(grid), dim3(64), lds, s, a);
        }
        SL_HIP(hipGetLastError());
        SL_HIP(hipEventRecord(c.ev_stop, s));
        SL_TRY(c.stage_end("consensus", s));
        c.timed = true;
    }

    // ---- results and error resolution ----
    std::vector<int32_t> len(static_cast<size_t>(ngroups), 0);
    std::vector<int8_t> status(static_cast<size_t>(std::max<int64_t>(rows_eval, 1)), 0);
    unsigned long long badchar = ~0ull;
    int fixn = 0;
    if (ng_eval > 0) {
        SL_HIP(hipMemcpy(len.data(), d_len, sizeof(int32_t) * static_cast<size_t>(ng_eval), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(status.data(), d_status, static_cast<size_t>(std::max<int64_t>(rows_eval, 1)), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(&badchar, d_badchar, sizeof badchar, hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(&fixn, d_fixn, sizeof fixn, hipMemcpyDeviceToHost));
    }
    // earliest error in the reference's processing order
    if (!quality && badchar != ~0ull) {
        char bc = '?';
        if (aln_host) bc = aln_host[static_cast<int64_t>(badchar)];
        else SL_HIP(hipMemcpy(&bc, a.aln + badchar, 1, hipMemcpyDeviceToHost));
        char msg[96];
        snprintf(msg, sizeof msg, "unknown character '%c' in alignment string", bc);
        return fail("%s", msg);
    }
    if (a.codes && a.code_bad) {
        int bad_row = 0x7fffffff;
        SL_HIP(hipMemcpy(&bad_row, a.code_bad, sizeof bad_row, hipMemcpyDeviceToHost));
        if (bad_row != 0x7fffffff) return fail("quality cannot be lower than smallest encoded value");
    }
    if (quality)
        for (int64_t r = 0; r < rows_eval; ++r) {
            if (status[r] == 1) return fail("quality cannot be lower than smallest encoded value");
            if (status[r] == 2) return fail("quality vector is shorter than the alignment sequence");
            if (status[r] == 3) return fail("quality vector is longer than the alignment sequence");
        }
    if (struct_err_kind == 1) return fail("alignment strings should have the same length");
    if (struct_err_kind == 2) return fail("alignments and qualities have different numbers of entries");
    if (fixn > fix_cap) return fail("sarlacc_amd: too many Phred values on a rounding boundary (%d)", fixn);

    // compact the kept columns on the device, then copy only those back
    std::vector<long long> dst(static_cast<size_t>(ngroups) + 1, 0);
    for (int64_t g = 0; g < ngroups; ++g) dst[g + 1] = dst[g] + len[g];
    const long long kept = dst[ngroups];
    // (the caller's buffers hold the KEPT columns: their number is known only now -- the alignments of clusters of several
    // molecules are several reads wide and keep one read's worth of columns, so a bound from the widths would refuse, and make a
    // caller repeat, calls whose results fit easily)
    if (cons_cap >= 0 && kept > cons_cap) return fail("sarlacc_amd: consensus output buffer too small (%lld needed)", kept);
    // results land directly in the caller's buffers (no intermediate host copies)
    uint8_t* const hc = reinterpret_cast<uint8_t*>(cons);
    uint8_t* const hp = reinterpret_cast<uint8_t*>(phred);
    double* const hl = lerr;
    if (kept) {
        long long* d_dst; uint8_t* d_cc; uint8_t* d_cp; double* d_cl = nullptr;
        SL_TRY(upload("cons.dst", dst.data(), dst.size(), &d_dst, s));
        SL_TRY(scratch("cons.cc", static_cast<size_t>(kept), &d_cc));
        SL_TRY(scratch("cons.cp", static_cast<size_t>(kept), &d_cp));
        if (lerr) SL_TRY(scratch("cons.cl", static_cast<size_t>(kept), &d_cl));
        hipLaunchKernelGGL(k_consensus_compact, dim3(static_cast<unsigned>(ng_eval)), dim3(256), 0, s, d_cons, d_phred, d_lerr,
                           d_ooff, d_len, d_dst, static_cast<long long>(ng_eval), d_cc, d_cp, d_cl);
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpy(hc, d_cc, static_cast<size_t>(kept), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(hp, d_cp, static_cast<size_t>(kept), hipMemcpyDeviceToHost));
        if (lerr) SL_HIP(hipMemcpy(hl, d_cl, sizeof(double) * static_cast<size_t>(kept), hipMemcpyDeviceToHost));
    }
    if (fixn > 0) {  // exact host-libm re-evaluation of boundary columns
        std::vector<long long> fp(static_cast<size_t>(fixn));
        std::vector<double> fv(static_cast<size_t>(fixn) * 4);
        SL_HIP(hipMemcpy(fp.data(), d_fixpos, sizeof(long long) * fp.size(), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(fv.data(), d_fixval, sizeof(double) * fv.size(), hipMemcpyDeviceToHost));
        // entries the fast kernel made for a group it later handed over are void (the generic kernel made its own)
        std::vector<int> fg(static_cast<size_t>(fixn)), gflag;
        SL_HIP(hipMemcpy(fg.data(), d_fixgrp, sizeof(int) * fg.size(), hipMemcpyDeviceToHost));
        if (a.only_flagged) {
            gflag.resize(static_cast<size_t>(ng_eval));
            SL_HIP(hipMemcpy(gflag.data(), d_gflag, sizeof(int) * gflag.size(), hipMemcpyDeviceToHost));
        }
        for (int k = 0; k < fixn; ++k) {
            if (fg[k] >= 0 && !gflag.empty() && gflag[static_cast<size_t>(fg[k])]) continue;
            double le;
            if (quality) {
                double denom = fv[4 * k];
                denom += host_log1pexp(fv[4 * k + 1] - denom);
                denom += host_log1pexp(fv[4 * k + 2] - denom);
                const double err3 = denom;
                denom += host_log1pexp(fv[4 * k + 3] - denom);
                le = err3 - denom;
            } else {
                le = std::log1p(-((fv[4 * k] + pseudo / 4) / (fv[4 * k + 1] + pseudo)));
            }
            // fix_pos is an index into the uncompacted layout: find its alignment, then its slot
            const long long pos = fp[k];
            const int64_t g = static_cast<int64_t>(std::upper_bound(out_off.begin(), out_off.begin() + ng_eval, pos) - out_off.begin()) - 1;
            const size_t at = static_cast<size_t>(dst[g] + (pos - out_off[g]));
            hp[at] = static_cast<uint8_t>(phred_char(le));
            if (lerr) hl[at] = le;
        }
    }
    for (int64_t g = 0; g < ngroups; ++g) cons_off[g + 1] = dst[g + 1];
    return 0;
}

static int run_consensus(bool quality, const char* aln, const int64_t* aln_off, const int64_t* grp_rows,
                         int64_t ngroups, const char* qual, const int64_t* qual_off, const int64_t* qgrp_rows,
                         double min_cov, double pseudo, const double* enc_errors, const char* enc_names, int enc_n,
                         char* cons, char* phred, int64_t* cons_off, double* lerr) {
    if (ngroups < 0) return fail("sarlacc_amd: negative number of alignments");
    if (quality) SL_TRY(check_encoding(enc_errors, enc_names, enc_n));
    cons_off[0] = 0;
    if (ngroups == 0) return 0;
    const int64_t nrows_total = grp_rows[ngroups];

    // host-side structural checks, in the reference's order (group by group):
    // equal row widths (src/DNA_input.cpp:90-104), then matching entry counts (:186-190)
    int64_t struct_err_group = -1;
    int struct_err_kind = 0;
    int max_rows = 1;
    for (int64_t g = 0; g < ngroups && struct_err_group < 0; ++g) {
        const int64_t r0 = grp_rows[g], r1 = grp_rows[g + 1];
        max_rows = static_cast<int>(std::max<int64_t>(max_rows, r1 - r0));
        for (int64_t r = r0 + 1; r < r1; ++r)
            if (aln_off[r + 1] - aln_off[r] != aln_off[r0 + 1] - aln_off[r0]) { struct_err_group = g; struct_err_kind = 1; break; }
        if (struct_err_group < 0 && quality && (qgrp_rows[g + 1] - qgrp_rows[g]) != (r1 - r0)) {
            struct_err_group = g;
            struct_err_kind = 2;
        }
    }
    // Only groups before the first structural error are evaluated on the device.
    const int64_t ng_eval = struct_err_group >= 0 ? struct_err_group : ngroups;
    const int64_t rows_eval = grp_rows[ng_eval];
    if (quality)
        for (int64_t g = 0; g < ng_eval; ++g)
            if (qgrp_rows[g] != grp_rows[g]) return fail("sarlacc_amd: alignment and quality row numbering differ");

    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    const int64_t total = aln_off[nrows_total] - aln_off[0];
    const int64_t base = aln_off[0];

    std::vector<int64_t> rel(static_cast<size_t>(nrows_total) + 1), out_off(static_cast<size_t>(std::max<int64_t>(ngroups, 1)));
    for (int64_t r = 0; r <= nrows_total; ++r) rel[r] = aln_off[r] - base;
    for (int64_t g = 0; g < ngroups; ++g) out_off[g] = rel[grp_rows[g]];

    ConsArgs a{};
    uint8_t* d_aln; int64_t* d_aoff; int64_t* d_grows; int64_t* d_ooff;
    SL_TRY(upload("cons.aln", reinterpret_cast<const uint8_t*>(aln) + base, static_cast<size_t>(total), &d_aln, s));
    SL_TRY(upload("cons.aoff", rel.data(), rel.size(), &d_aoff, s));
    SL_TRY(upload("cons.grows", grp_rows, static_cast<size_t>(ngroups) + 1, &d_grows, s));
    SL_TRY(upload("cons.ooff", out_off.data(), out_off.size(), &d_ooff, s));
    a.aln = d_aln; a.aln_off = d_aoff; a.grp_rows = d_grows; a.out_off = d_ooff; a.max_rows = max_rows;
    a.aln_bytes = total;
    if (quality) {
        const int64_t qrows = qgrp_rows[ng_eval];
        const int64_t qbase = qual_off[0];
        const int64_t qtotal = qual_off[qrows] - qbase;
        std::vector<int64_t> qrel(static_cast<size_t>(qrows) + 1);
        for (int64_t r = 0; r <= qrows; ++r) qrel[r] = qual_off[r] - qbase;
        uint8_t* d_q; int64_t* d_qoff;
        SL_TRY(upload("cons.qual", reinterpret_cast<const uint8_t*>(qual) + qbase, static_cast<size_t>(qtotal), &d_q, s));
        SL_TRY(upload("cons.qoff", qrel.data(), qrel.size(), &d_qoff, s));
        a.qual = d_q; a.qual_off = d_qoff; a.qual_bytes = qtotal;
    }
    return consensus_core(quality, a, ngroups, ng_eval, rows_eval, total, out_off, aln + base, struct_err_kind, min_cov, pseudo,
                          enc_errors, enc_names, enc_n, cons, phred, cons_off, lerr, s);
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" {

int sarlacc_create_consensus_basic_loop(const char* aln, const int64_t* aln_off, const int64_t* grp_rows,
                                        int64_t ngroups, double min_cov, double pseudo_count, char* cons,
                                        char* phred, int64_t* cons_off, double* lerr) {
    return run_consensus(false, aln, aln_off, grp_rows, ngroups, nullptr, nullptr, nullptr, min_cov, pseudo_count,
                         nullptr, nullptr, 0, cons, phred, cons_off, lerr);
}

// Shared body of sarlacc_msa_consensus (host pointers) and sarlacc_dev_msa_consensus (reads and
// qualities already in HBM: d_seq / d_qual non-null, seq / qual unused).
static int msa_consensus_impl(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq,
                              const int64_t* seq_off, const char* qual, const int64_t* qual_off, int64_t nseq, double match,
                              double mismatch, double gap_extension, double gap_opening, int bandwidth, double min_cov,
                              double pseudo_count, const double* enc_errors, const char* enc_names, int enc_n, char* cons,
                              char* phred, int64_t* cons_off, int64_t cons_cap, const uint8_t* d_seq_res, const uint8_t* d_qual_res,
                              bool quality) {
    if (ngroups < 0 || nseq < 0) return fail("sarlacc_amd: negative sizes");
    if (quality) SL_TRY(check_encoding(enc_errors, enc_names, enc_n));
    cons_off[0] = 0;
    if (ngroups == 0) return 0;
    MsaResult res;
    res.width.assign(static_cast<size_t>(ngroups), 0);
    res.out_off.assign(static_cast<size_t>(ngroups) + 1, 0);
    // the quality strings travel to the device on a stream of their own while the pairwise
    // alignments run (qualities stay in read order; every row finds its string through the member list)
    uint8_t* d_q = nullptr; int64_t* d_qoff = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t q_ready = nullptr;
    std::vector<int64_t> qrel;
    // Rows as vote codes (k_consensus_code) when the quality strings are laid out like the reads -- every read as long as
    // its quality string, which is also what the reference demands -- and the table fits LDS; otherwise characters.
    bool codes = quality && static_cast<size_t>(enc_n + 1) * QC_ROWB <= 150 * 1024 && static_cast<int>(enc_names[0]) + enc_n <= 255 &&
                 !option(OPT_CONSENSUS_CHARS);
    for (int64_t r = 0; codes && r < nseq; ++r)
        if (qual_off[r + 1] - qual_off[r] != seq_off[r + 1] - seq_off[r]) codes = false;
    const uint8_t* d_q_const = nullptr;
    if (codes) {
        SL_TRY(ensure_device());
        int* d_bad;
        SL_TRY(scratch("cons.codebad", 1, &d_bad));
        const int none = 0x7fffffff;
        SL_HIP(hipMemcpy(d_bad, &none, sizeof none, hipMemcpyHostToDevice));
        SL_HIP(hipEventCreateWithFlags(&q_ready, hipEventDisableTiming));
        res.code.want = true; res.code.qual = &d_q_const; res.code.ready = q_ready;
        res.code.qoffset = static_cast<int>(enc_names[0]); res.code.navail = enc_n; res.code.d_bad = d_bad;
    }
    const std::function<int()> upload_quals = [&]() -> int {
        if (!quality) return 0;
        const int64_t qbase = qual_off[0];
        qrel.resize(static_cast<size_t>(nseq) + 1);
        for (int64_t r = 0; r <= nseq; ++r) qrel[r] = qual_off[r] - qbase;
        SL_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        if (d_qual_res) d_q = const_cast<uint8_t*>(d_qual_res);
        else SL_TRY(upload("cons.qual", reinterpret_cast<const uint8_t*>(qual) + qbase, static_cast<size_t>(qrel[nseq]), &d_q, copy_stream));
        SL_TRY(upload("cons.qoff", qrel.data(), qrel.size(), &d_qoff, copy_stream));
        d_q_const = d_q;
        if (q_ready) SL_HIP(hipEventRecord(q_ready, copy_stream));
        return 0;
    };
    struct EventGuard { hipEvent_t* e; ~EventGuard() { if (*e) (void)hipEventDestroy(*e); } } q_ready_guard{&q_ready};
    const int msa_rc = msa_run(grp_off, grp, ngroups, seq, seq_off, nseq, match, mismatch, gap_extension, gap_opening, bandwidth, true,
                               -1, &res, &upload_quals, d_seq_res);
    if (copy_stream) {
        const hipError_t e1 = hipStreamSynchronize(copy_stream), e2 = hipStreamDestroy(copy_stream);
        if (!msa_rc && (e1 != hipSuccess || e2 != hipSuccess)) return fail("HIP error while uploading the quality strings");
    }
    if (msa_rc) return msa_rc;
    if (quality && !d_q) return fail("sarlacc_amd: quality upload did not run");
    const int64_t total = res.out_off[ngroups];
    if (total == 0) {
        for (int64_t g = 0; g < ngroups; ++g) cons_off[g + 1] = 0;
        return 0;
    }
    hipStream_t s = nullptr;
    // row table of the alignments the MSA stage left in HBM
    const int64_t nrows = grp_off[ngroups] - grp_off[0];
    std::vector<int64_t> rel(static_cast<size_t>(nrows) + 1), grows(static_cast<size_t>(ngroups) + 1);
    std::vector<int64_t> out_off(static_cast<size_t>(ngroups));
    int max_rows = 1;
    int64_t row = 0;
    for (int64_t g = 0; g < ngroups; ++g) {
        const int64_t m = grp_off[g + 1] - grp_off[g];
        grows[g] = row;
        out_off[g] = res.out_off[g];
        max_rows = static_cast<int>(std::max<int64_t>(max_rows, m));
        for (int64_t r = 0; r < m; ++r) rel[row++] = res.out_off[g] + r * res.width[g];
    }
    grows[ngroups] = row;
    rel[nrows] = total;
    ConsArgs a{};
    int64_t* d_aoff; int64_t* d_grows; int64_t* d_ooff;
    SL_TRY(upload("cons.aoff", rel.data(), rel.size(), &d_aoff, s));
    SL_TRY(upload("cons.grows", grows.data(), grows.size(), &d_grows, s));
    SL_TRY(upload("cons.ooff", out_off.data(), out_off.size(), &d_ooff, s));
    a.aln = res.d_out; a.aln_off = d_aoff; a.grp_rows = d_grows; a.out_off = d_ooff; a.max_rows = max_rows;
    a.aln_bytes = total;
    if (codes) { a.codes = res.d_codes; a.code_bad = res.code.d_bad; }
    if (quality) { a.qual = d_q; a.qual_off = d_qoff; a.row_read = res.d_members; a.qual_bytes = qrel[static_cast<size_t>(nseq)]; }
    return consensus_core(quality, a, ngroups, ngroups, nrows, total, out_off, nullptr, 0, min_cov, pseudo_count, enc_errors,
                          enc_names, enc_n, cons, phred, cons_off, nullptr, s, cons_cap);
}

int sarlacc_msa_consensus(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq,
                          const int64_t* seq_off, const char* qual, const int64_t* qual_off, int64_t nseq, double match,
                          double mismatch, double gap_extension, double gap_opening, int bandwidth, double min_cov,
                          double pseudo_count, const double* enc_errors, const char* enc_names, int enc_n, char* cons,
                          char* phred, int64_t* cons_off, int64_t cons_cap) {
    return msa_consensus_impl(grp_off, grp, ngroups, seq, seq_off, qual, qual_off, nseq, match, mismatch, gap_extension,
                              gap_opening, bandwidth, min_cov, pseudo_count, enc_errors, enc_names, enc_n, cons, phred,
                              cons_off, cons_cap, nullptr, nullptr, qual != nullptr);
}

int sarlacc_dev_msa_consensus(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const uint8_t* d_seq,
                              const uint8_t* d_qual, const int64_t* off, int64_t nseq, double match, double mismatch,
                              double gap_extension, double gap_opening, int bandwidth, double min_cov, double pseudo_count,
                              const double* enc_errors, const char* enc_names, int enc_n, char* cons, char* phred,
                              int64_t* cons_off, int64_t cons_cap) {
    if (!d_seq) return fail("sarlacc_amd: sarlacc_dev_msa_consensus needs the reads in device memory");
    return msa_consensus_impl(grp_off, grp, ngroups, nullptr, off, nullptr, off, nseq, match, mismatch, gap_extension,
                              gap_opening, bandwidth, min_cov, pseudo_count, enc_errors, enc_names, enc_n, cons, phred,
                              cons_off, cons_cap, d_seq, d_qual, d_qual != nullptr);
}

int sarlacc_create_consensus_quality_loop(const char* aln, const int64_t* aln_off, const int64_t* grp_rows,
                                          int64_t ngroups, const char* qual, const int64_t* qual_off,
                                          const int64_t* qgrp_rows, double min_cov, const double* enc_errors,
                                          const char* enc_names, int enc_n, char* cons, char* phred,
                                          int64_t* cons_off, double* lerr) {
    return run_consensus(true, aln, aln_off, grp_rows, ngroups, qual, qual_off, qgrp_rows, min_cov, 0.0, enc_errors,
                         enc_names, enc_n, cons, phred, cons_off, lerr);
}
}
