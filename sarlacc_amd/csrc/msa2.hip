// msa2.hip -- per-group multiple sequence alignment, "MSA spec v2" (DESIGN.md section 5), on gfx950.
//
// Stands in for the reference's quick_msa (/root/reference/src/quick_msa.cpp:15-80), which hands each
// group to SeqAn's T-Coffee (globalMsaAlignment, :66).  Spec v2 is that pipeline restated at base
// resolution (checker: oracle/msa2.c, where every step is written down with the SeqAn routine it
// follows): all-pairs banded global alignments -> distances -> neighbour-joining guide tree ->
// primary library + full triplet extension -> progressive merging of profiles by the heaviest common
// subsequence of their columns.  PARITY UNPINNED (SeqAn is absent from the reference tree and the
// image, the reference has no test of quick_msa).
//
// Data per batch of groups (all in HBM):
//   map   uint16  for every ordered pair (a, b) of a group: position of b aligned to position p of a
//                 (0xFFFF = gap) -- both directions of every pairwise alignment, written by the walk of
//                 msa_pairwise.hip (OUT 1);
//   col   uint16  column of every base in the profile that currently holds its read;
//   pos   uint16  per group n x wcap: position of read a at column c of its profile (0xFFFF = gap);
//   ext   uint32  the extended library: per group, pair and position of the pair's first-child member the partner positions
//                 and weights of spec v2 step 5 (k_m2_extend, once per batch, before the merging).
// Groups are independent (src/quick_msa.cpp:39); only the joins INSIDE a group are ordered.  So after the pairwise
// alignments and the guide trees ONE launch does all the merging: k_m2_group, one wavefront per group, groups
// pulled from an atomic counter in order of decreasing size, and for every join of its group the wavefront runs
//   rows      lane = column i of the first child; reads the extended library of every member pair at the lane's
//             position, sums the weights per partner column in a private list (registers), applies the
//             row cap and the noise filter of spec v2 step 5 and appends the row's entries, by column, to a
//             compact match list (row, column, weight) in the wavefront's scratch;
//   chain     heaviest chain over the match list, 64 matches per step: prefix maxima over the second child's
//             columns in an LDS ring behind the front, dependencies inside the block resolved in registers,
//             predecessors stored per match; then the walk back through the predecessors;
//   renumber  new column numbers (first child's unmatched columns before the second child's between two
//             matched pairs), col / pos of every member updated.
// Nothing is exchanged between wavefronts, so there is no waiting and no launch per join (round 2 launched five
// kernels per join for all groups of a batch and spent most of the stage with a few long wavefronts on the chip).
// A row keeps the first M2_CAP distinct partner columns (spec v2, step 5); groups whose profiles outgrow the
// fast capacity are redone with profiles as wide as the sum of the read lengths.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "msa_common.hpp"

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <numeric>
#include <type_traits>
#include <thread>
#include <utility>
#include <vector>

namespace sarlacc {

constexpr int M2_MAXN = 64;          // group sizes aligned by spec v2 (member sets are 64-bit masks; 32-bit ones in the kernels of groups of up to 32)
constexpr int M2_N32 = 32;
typedef unsigned long long m2_mask;
#ifndef M2_WAVES_EU
#define M2_WAVES_EU 8   // wavefronts per SIMD the merge kernel is compiled for (its registers are capped accordingly)
#endif
#ifndef M2_NWD
#define M2_NWD 4   // wavefronts per group of 33 .. 64 reads
#endif
constexpr int M2_NB = 24;   // groups of up to M2_NB reads: one wavefront; up to 32: 8 wavefronts; 33 .. 64: M2_NWD (k_m2_group; sweeps: profiles/r03_exp_m2_class_thresholds_v1.txt, r04_exp_m2_classes_v1.txt -- a 4-wavefront class between the first two lost in every sweep and is gone)
constexpr int M2_CAP = 16;           // partner columns per row (spec v2, step 5)
constexpr unsigned M2_NONE = 0xFFFFu;
// profile capacity of the first pass: same-molecule reads grow a profile by 10-20 %, one or two unrelated reads in
// the cluster by their length each
// ... by group size: a cluster of n reads out of umi_group holds about n / 10 molecules (UMI collisions), whose profiles do
// not align -- one more read length per 8 reads beyond 11.  (With 3 maxlen + 64 for every size the clusters of three and
// more molecules, the most expensive groups of a call, ran out of columns and were done twice, all-pairs alignments
// included: 85 ms of the 657 ms merge stage of bench.py's pipeline workload.)
static inline long long m2_fast_width(long long n, long long maxlen) {
    if (option(OPT_MSA2_TIGHT_PROFILES)) return 2 * maxlen + 64;
    return (3 + std::max(0LL, (n - 4) / 8)) * maxlen + 64;
}

struct M2Member {         // one read of a group
    long long seq_off;    // into d_seq
    long long map_base;   // into d_map: (n - 1) arrays of `len` entries, the other members in member order
    long long col_base;   // into d_col
    long long ext_base;   // the group's (M2Group::ext_base): k_m2_extend works from the member table alone
    int len;
    int first_member;     // of its group
    int n;                // reads in its group
    int lmax;             // the group's longest read
};
struct M2Group {
    int first_member;     // member a of the group is members[first_member + a]; joins at first_member + k
    int n;
    int wcap;             // capacity of a profile in columns
    int lmax;             // the group's longest read: stride of the extended library's records per pair
    long long pos_base;   // into d_pos: n * wcap
    long long first_job;  // pairwise job of (a, b), a < b: first_job + a n - a (a + 1) / 2 + b - a - 1
    long long dist_base;  // into the tree kernel's scratch: n * n + n doubles
    long long ext_base;   // into a plane of the extended library: n (n - 1) / 2 pairs x lmax records
    long long map0, col0; // map_base / col_base of the group's first member
};

// counters of one call (sarlacc_stage_count): how often spec v2's own rules act, and the chain's fallback
enum { M2C_ROWS, M2C_ROWS_CAPPED, M2C_ENT_FILTERED, M2C_ROWS_FILTERED, M2C_ENT_KEPT, M2C_JOINS, M2C_JOINS_HBMQ,
       M2C_GATHERS,   // wave-wide gather instructions of the rows phase (positions, map entries, records, columns): a lower bound
       // where the wavefronts' time goes (s_memtime cycles summed over the wavefronts) and when they leave (s_memrealtime, 100 MHz)
       M2C_CYC_ROWS, M2C_CYC_CHAIN, M2C_CYC_WALK, M2C_CYC_RENUMBER, M2C_T_START, M2C_T_FIRST_EXIT, M2C_T_LAST_EXIT,
       M2C_T_EXIT1, M2C_T_EXIT4, M2C_T_EXIT8,   // last exit of the instantiations: one wavefront per group / groups of 33 .. 64 reads / 8 wavefronts
       M2C_N };

struct M2Args {
    const uint8_t* seq;
    const M2Group* groups;
    const M2Member* members;
    int ngroups;
    int ma, mm;
    const uint16_t* map;
    const int2* stats;
    double* dist;
    int2* joins;
    m2_mask* first;                // per member: the members it meets as part of the FIRST child (k_m2_first)
    uint32_t* ext;                 // the extended library (k_m2_extend): plane 0 here, planes 1 .. 2 (unit weights) or 1 .. 3 at
    long long ext_off1, ext_plane; //   ext + ext_off1 + (k - 1) ext_plane -- plane 0 may live in a buffer of its own (m2_merge)
    uint16_t* col;                 // (a column is < 65535: 16 bits halve the bytes of the walk's column gathers)
    uint16_t* pos;
    int* ovf;                      // per group: a profile outgrew its capacity
    int32_t* width;                // per group: columns of the final profile
    // ---- k_m2_group: the launch's range of groups [g0, g1), work counter, counters, per-workgroup scratch (w_rows columns each) ----
    int g0, g1;
    int* next;
    unsigned long long* counters;
    unsigned long long* w_ent;     // match list: (row << 48) | (column << 32) | weight; M2_CAP entries per column
    unsigned* w_pred;              // per match: 1 + index of its predecessor on its best chain (0: none)
    int* w_part;                   // partner column of column i of the first child, -1 if unmatched
    int* w_nca;                    // new numbers of the first child's columns
    int* w_ncb;                    // ... of the second child's
    int* w_pb;                     // partner row of column j of the second child
    unsigned long long* w_q;       // prefix maxima over all columns (the chain's fallback when the LDS ring cannot answer)
    long long w_rows;
    int chain_hbm;                 // testing: the prefix maxima in HBM from the start (what the LDS ring falls back to)
    // ---- row writer ----
    uint8_t* out;                  // gapped rows
    uint16_t* out16;               // rows as vote codes instead of characters (CodeSpec, common.hpp): out16 != nullptr
    const uint8_t* qual;           // laid out like seq
    int qoffset, navail;
    int* bad;
};

__device__ __forceinline__ int m2_w0(int x, int y, int ma, int mm) {
    const int s = (x == y) ? ma : mm;
    return s > 1 ? s : 1;
}

// ---- guide tree: one wavefront per group (oracle/msa2.c nj_tree, operation for operation) ----
// Lane i = slot i of the distance matrix (n <= 64 slots, the matrix in LDS).  Per round: every live lane sums its row in slot
// order (the oracle's order of additions), finds the first minimum of the Saitou-Nei criterion among its partners j > i, the
// wavefront takes the smallest value and, among equal ones, the smallest (i, j) -- the oracle's "first minimum in (i, j) order" --
// and the live lanes update their distance to the joined node.  (Round 3 ran one THREAD per group: a 39-read cluster kept the
// stage waiting 11 ms for 59 000 dependent iterations; a 64-read one would take 50.)
__global__ void __launch_bounds__(64) k_m2_tree(M2Args A) {
    __shared__ double s_D[M2_MAXN * M2_MAXN];
    __shared__ double s_R[M2_MAXN];
    const int g = blockIdx.x;
    if (g >= A.ngroups) return;
    const M2Group G = A.groups[g];
    const int n = G.n;
    const int fm = G.first_member;
    const int lane = threadIdx.x;
    if (lane == 0) A.width[g] = n == 1 ? A.members[fm].len : 0;   // (groups of two and more: written by k_m2_group)
    if (n < 2) return;
    if (lane < n) {
        s_D[lane * n + lane] = 0.0;
        const int lena = A.members[fm + lane].len;
        for (int b = lane + 1; b < n; ++b) {
            const int2 st = A.stats[G.first_job + static_cast<long long>(lane) * n - static_cast<long long>(lane) * (lane + 1) / 2 + b - lane - 1];
            const long long alen = static_cast<long long>(lena) + A.members[fm + b].len - st.y;
            const double d = alen > 0 ? 1.0 - static_cast<double>(st.x) / static_cast<double>(alen) : 0.0;
            s_D[lane * n + b] = d;
            s_D[b * n + lane] = d;
        }
    }
    __syncthreads();
    m2_mask active = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    int node = lane;   // the tree node in this lane's slot
    int r = n, nj = 0;
    while (r > 3) {
        const bool live = lane < n && ((active >> lane) & 1ull);
        if (live) {
            double sum = 0.0;
            for (int k = 0; k < n; ++k)
                if (((active >> k) & 1ull) && k != lane) sum = sum + s_D[lane * n + k];
            s_R[lane] = sum;
        }
        __syncthreads();
        // this lane's first minimum over its partners j > lane
        const double rm2 = static_cast<double>(r - 2);
        double best = 0.0;
        int bj = -1;
        if (live) {
            const double Ri = s_R[lane];
            for (int j = lane + 1; j < n; ++j) {
                if (!((active >> j) & 1ull)) continue;
                const double q = (rm2 * s_D[lane * n + j] - Ri) - s_R[j];
                if (bj < 0 || q < best) { best = q; bj = j; }
            }
        }
        // the wavefront's: smallest value, then smallest lane (a lane without partners takes no part)
        double wbest = best;
        int wi = bj >= 0 ? lane : 0x7fffffff, wj = bj;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ob = __shfl_xor(wbest, d);
            const int oi = __shfl_xor(wi, d), oj = __shfl_xor(wj, d);
            const bool take = oi != 0x7fffffff && (wi == 0x7fffffff || ob < wbest || (ob == wbest && oi < wi));
            wbest = take ? ob : wbest; wi = take ? oi : wi; wj = take ? oj : wj;
        }
        const int bi = __builtin_amdgcn_readfirstlane(wi), bjj = __builtin_amdgcn_readfirstlane(wj);
        const int node_i = __builtin_amdgcn_readlane(node, bi), node_j = __builtin_amdgcn_readlane(node, bjj);
        if (lane == 0) A.joins[fm + nj] = make_int2(node_i, node_j);
        const double dij = s_D[bi * n + bjj];
        __syncthreads();
        if (live && lane != bi && lane != bjj) {
            const double v = ((s_D[bi * n + lane] + s_D[bjj * n + lane]) - dij) * 0.5;
            s_D[bi * n + lane] = v;
            s_D[lane * n + bi] = v;
        }
        active &= ~(1ull << bjj);
        if (lane == bi) node = n + nj;
        ++nj; --r;
        __syncthreads();
    }
    // the remaining two or three nodes: lowest slots first
    int l[3], c = 0;
    for (int i = 0; i < n && c < 3; ++i)
        if ((active >> i) & 1ull) l[c++] = i;
    if (c >= 2) {
        const int n0 = __builtin_amdgcn_readlane(node, l[0]), n1 = __builtin_amdgcn_readlane(node, l[1]);
        if (lane == 0) A.joins[fm + nj] = make_int2(n0, n1);
        if (lane == l[0]) node = n + nj;
        ++nj;
    }
    if (c == 3) {
        const int n0 = __builtin_amdgcn_readlane(node, l[0]), n2 = __builtin_amdgcn_readlane(node, l[2]);
        if (lane == 0) A.joins[fm + nj] = make_int2(n0, n2);
    }
}

// ---- leaves: col = position, pos = identity ----
__global__ void k_m2_init(M2Args A, const int* member_group, int nmembers) {
    const int m = blockIdx.y;
    if (m >= nmembers) return;
    const M2Group G = A.groups[member_group[m]];
    const M2Member Me = A.members[m];
    const int a = m - G.first_member;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Me.len; p += gridDim.x * blockDim.x) {
        A.col[Me.col_base + p] = static_cast<uint16_t>(p);
        A.pos[G.pos_base + static_cast<long long>(a) * G.wcap + p] = static_cast<uint16_t>(p);
    }
}

// =============================================================================================
// k_m2_group: all joins of one group on one wavefront

typedef unsigned long long m2_u64;

__device__ __forceinline__ int m2_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// The lane index, recomputed where a phase starts: `volatile`, so nothing derived from it (lane addresses, lane masks)
// is computed once at the top of the kernel and kept in registers through every other phase.
__device__ __forceinline__ int m2_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ unsigned m2_readlane(unsigned v, int l) { return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), l)); }
__device__ __forceinline__ m2_u64 m2_readlane64(m2_u64 v, int l) {
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), l));
    return (static_cast<m2_u64>(hi) << 32) | lo;
}
__device__ __forceinline__ m2_u64 m2_readlane(m2_u64 v, int l) { return m2_readlane64(v, l); }
// Wave-wide scans of an int by DPP: row shifts inside the rows of 16 lanes, then the last lane of row 0 / 2 broadcast
// into row 1 / 3 and lane 31 into rows 2 and 3; lanes without a source read the identity.  (Not __shfl: its
// ds_bpermute needs a lane-address register per step, and the compiler keeps those alive through the whole kernel.)
template <int OP>   // 0: sum, 1: maximum, 2: minimum
__device__ __forceinline__ int m2_scan_i32(int v, const int identity) {
#define M2_ISTEP(CTRL, ROWMASK)                                                                      \
    {                                                                                                \
        const int o = __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWMASK, 0xf, false);           \
        v = OP == 0 ? v + o : (OP == 1 ? max(v, o) : min(v, o));                                     \
    }
    M2_ISTEP(0x111, 0xf) M2_ISTEP(0x112, 0xf) M2_ISTEP(0x114, 0xf) M2_ISTEP(0x118, 0xf)
    M2_ISTEP(0x142, 0xa) M2_ISTEP(0x143, 0xc)
#undef M2_ISTEP
    return v;
}
__device__ __forceinline__ int m2_incl_sum(int v) { return m2_scan_i32<0>(v, 0); }
__device__ __forceinline__ int m2_wave_max(int v) { return __builtin_amdgcn_readlane(m2_scan_i32<1>(v, static_cast<int>(0x80000000)), 63); }
__device__ __forceinline__ int m2_wave_min(int v) { return __builtin_amdgcn_readlane(m2_scan_i32<2>(v, 0x7fffffff), 63); }
__device__ __forceinline__ int m2_excl_max(int v, const int identity) {   // exclusive prefix maximum over the lanes
    const int x = m2_scan_i32<1>(v, static_cast<int>(0x80000000));
    return max(__builtin_amdgcn_update_dpp(identity, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false), identity);
}
// inclusive prefix maximum of a 64-bit value over the wavefront: DPP row shifts inside the rows of 16, then the last
// lane of row 0 / 2 broadcast into row 1 / 3 and lane 31 into rows 2 and 3 (lanes without a source read 0, the identity)
__device__ __forceinline__ m2_u64 m2_scan_max64(m2_u64 v) {
#define M2_SCAN_STEP(CTRL, ROWMASK)                                                                                              \
    {                                                                                                                            \
        const unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);                                       \
        const unsigned olo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), CTRL, ROWMASK, 0xf, false)); \
        const unsigned ohi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), CTRL, ROWMASK, 0xf, false)); \
        const m2_u64 o = (static_cast<m2_u64>(ohi) << 32) | olo;                                                                 \
        v = o > v ? o : v;                                                                                                       \
    }
    M2_SCAN_STEP(0x111, 0xf) M2_SCAN_STEP(0x112, 0xf) M2_SCAN_STEP(0x114, 0xf) M2_SCAN_STEP(0x118, 0xf)
    M2_SCAN_STEP(0x142, 0xa) M2_SCAN_STEP(0x143, 0xc)
#undef M2_SCAN_STEP
    return v;
}

// ---- the extended library (spec v2, step 5), once per group before the merging ----
// For an ordered pair of reads (a, b) and a position p of a the library holds the direct partner q0 = map(a -> b)[p] with
// its weight and the first M2_LIB OTHER positions of b named by a triplet p - r - q through a third read c (c ascending),
// each with the sum of its triplets' weights: a function of (a, p, b) alone.  Rounds 2-4 walked the triplets inside
// k_m2_group, per join and per column of the first child, with two dependent 2-byte gathers and a 16-entry list insertion per
// (a, b, c): n^3 L gathers from maps that no cache held (1.6 TB of fabric traffic per merge stage at bench.py's pipeline
// workload, 18 x the maps) at ~90 instructions per candidate.  Now k_m2_extend does the n^3 L part once, one wavefront per 64
// positions of a: the positions of its lanes in every third read staged in LDS, one nearly contiguous 2-byte gather per
// candidate, four compares; the merging reads the result, |A| |B| records per column instead of |A| |B| n candidates.
// Only the pairs' orientation that the merging reads is computed: a in the FIRST child of the join that brings a and b
// together (k_m2_first: one 64-bit mask of such b per read).
//   record of (a, b, p), unit weights: plane 0 = w0 (6 bits) | others (2) | w1 (8) | q1 (16); plane 1 = w2 | q2 << 16; plane 2 = w3 | q3 << 16
//                        any weights:  plane 0 = w0 | others << 16; plane k = wk | qk << 16 (k = 1 .. 3)
//   at ext + plane * ext_plane + group's ext_base + pair_index(a, b) * lmax + p   (lmax: the group's longest read)
constexpr int M2_LIB = 3;   // oracle/msa2.c MSA2_LIBRARY
constexpr int M2_EXT_WAVES = 4;
static_assert(M2_LIB == 3, "the records of k_m2_extend hold three further partner positions");
typedef const __attribute__((address_space(1))) uint16_t m2_gu16;
__host__ __device__ __forceinline__ long long m2_pair_index(int a, int b, int n) {   // a != b
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return static_cast<long long>(lo) * n - static_cast<long long>(lo) * (lo + 1) / 2 + hi - lo - 1;
}
static inline int m2_ext_planes(bool unitw) { return unitw ? 3 : 4; }

// first[m] = the members b of m's group for which m is in the first child of the join that brings the two together
__global__ void __launch_bounds__(64) k_m2_first(M2Args A) {
    __shared__ m2_mask s_mask[2 * M2_MAXN];
    const int g = blockIdx.x;
    const M2Group G = A.groups[g];
    const int n = G.n, fm = G.first_member;
    const int lane = threadIdx.x;
    if (n < 2) {
        if (lane < n) A.first[fm + lane] = 0;
        return;
    }
    if (lane < n) s_mask[lane] = 1ull << lane;
    __syncthreads();
    m2_mask F = 0;
    for (int k = 0; k + 1 < n; ++k) {
        const int2 jn = A.joins[fm + k];
        const m2_mask mA = s_mask[jn.x], mB = s_mask[jn.y];
        if ((mA >> lane) & 1ull) F |= mB;
        if (lane == 0) s_mask[n + k] = mA | mB;   // (a new slot: nobody reads it before the barrier)
        __syncthreads();
    }
    if (lane < n) A.first[fm + lane] = F;
}

// One wavefront per 64 positions of member m (blockIdx.y = m; the block's 4 wavefronts take 4 consecutive windows, the grid's
// x extent strides over the read).  gridDim.x = 8 puts the same windows of consecutive members -- the reads of one group, which
// gather from the same maps -- on the same XCD, i.e. behind the same L2.
// (What bounds this kernel is the SCALAR unit, one instruction per SIMD every four cycles: the first version took a candidate's
// map base and length out of the lanes by v_readlane and did its 64-bit address arithmetic, the skip tests and two wave-wide
// branches per candidate on it -- 31 scalar instructions per gather against 22 vector ones, 0.22 per cycle and SIMD
// (rocprofv3 --pmc SQ_INSTS_SALU / SQ_INSTS_VALU / SQ_INSTS_VMEM).  Now the members' map addresses and lengths sit in an LDS
// table read by broadcast, the lanes compute their own addresses, and a batch of eight candidates that all name the direct
// partner -- the rule among same-molecule reads -- is recognised by ONE wave-wide test.)
struct __attribute__((aligned(16))) M2XMember {   // LDS: what a lane needs to know about member c
    const uint16_t* maps;   // its n - 1 position maps
    const uint8_t* seq;     // its bases
    int len, pad[3];
};
template <bool UNITW>
__global__ void __launch_bounds__(64 * M2_EXT_WAVES) k_m2_extend(M2Args A, const int* member_group, int m0, int m1, int maxn) {   // members [m0, m1)
    extern __shared__ __align__(16) unsigned char m2_xs[];
    const int m = m0 + static_cast<int>(blockIdx.y);
    if (m >= m1) return;
    const m2_mask F = A.first[m];
    if (!F) return;   // (the whole workgroup: F belongs to the member)
    const M2Member Ma = A.members[m];
    const int n = Ma.n, fm = Ma.first_member, a = m - fm;
    const int lane = threadIdx.x & 63;
    const int wave = m2_rfl(static_cast<int>(threadIdx.x >> 6));
    M2XMember* const s_mem = reinterpret_cast<M2XMember*>(m2_xs);
    uint16_t* const s_r = reinterpret_cast<uint16_t*>(m2_xs + static_cast<size_t>(maxn) * sizeof(M2XMember)) + static_cast<size_t>(wave) * maxn * 64 + lane;   // [c * 64]: this wavefront's, this lane's column
    if (threadIdx.x < static_cast<unsigned>(n)) {
        const M2Member Mc = A.members[fm + threadIdx.x];
        M2XMember X;
        X.maps = A.map + Mc.map_base; X.seq = A.seq + Mc.seq_off; X.len = Mc.len; X.pad[0] = X.pad[1] = X.pad[2] = 0;
        s_mem[threadIdx.x] = X;
    }
    __syncthreads();
    const int nwin = (Ma.len + 63) / 64;
    for (int w = blockIdx.x * M2_EXT_WAVES + wave; w < nwin; w += gridDim.x * M2_EXT_WAVES) {
        const int p = w * 64 + lane;
        const bool in = p < Ma.len;
        const unsigned pidx = in ? static_cast<unsigned>(p) : 0u;
        for (int c = 0; c < n; ++c) {
            const int slot = c < a ? c : c - 1;
            const unsigned r = c == a ? M2_NONE : A.map[Ma.map_base + static_cast<long long>(c == a ? 0 : slot) * Ma.len + pidx];
            s_r[c * 64] = static_cast<uint16_t>(in ? r : M2_NONE);
        }
        int xa = 0;
        if (!UNITW) xa = dna5_code(A.seq[Ma.seq_off + pidx]);
        for (int b = 0; b < n; ++b) {
            if (!((F >> b) & 1ull)) continue;
            const uint8_t* const seqb = s_mem[b].seq;
            // Comparisons as integer arithmetic in vector registers (a difference is zero where two positions agree; boolean
            // algebra on lane masks runs on the scalar unit): an invalid candidate is 0xFFFF, an empty slot 0x1FFFF, a missing
            // direct partner 0x2FFFF -- no two of them equal, none equal to a position.
            const unsigned q0r = s_r[b * 64];
            const unsigned q0 = q0r != M2_NONE ? q0r : 0x2FFFFu;
            unsigned w0 = 0;
            if (q0r != M2_NONE) w0 = UNITW ? 1u : static_cast<unsigned>(m2_w0(xa, dna5_code(seqb[q0r]), A.ma, A.mm));
            unsigned q1 = 0x1FFFFu, q2 = 0x1FFFFu, q3 = 0x1FFFFu, w1 = 0, w2 = 0, w3 = 0, nalt = 0;
            // the third reads eight at a time: their gathers are requested back to back, then the records updated in order of c
            for (int c0 = 0; c0 < n; c0 += 8) {
                unsigned qq[8], ri[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = min(c0 + u, n - 1);
                    const bool use = c0 + u < n && c != a && c != b;   // (wave-uniform)
                    const M2XMember X = s_mem[c];   // (one broadcast read)
                    const unsigned r = use ? s_r[c * 64] : M2_NONE;
                    ri[u] = r;
                    const unsigned off = static_cast<unsigned>(c == b ? 0 : (b < c ? b : b - 1)) * static_cast<unsigned>(X.len) + min(r, static_cast<unsigned>(max(X.len - 1, 0)));
                    const unsigned q = X.maps[off];   // (a gap reads the last entry of a valid map)
                    qq[u] = r != M2_NONE ? q : M2_NONE;
                }
                unsigned other = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) other |= (qq[u] ^ M2_NONE) ? (qq[u] ^ q0) : 0u;   // nonzero: a valid candidate that is not the direct partner
                if (UNITW && !__ballot(other != 0u)) {   // every triplet of the batch names the direct partner (or nothing)
#pragma unroll
                    for (int u = 0; u < 8; ++u) w0 += qq[u] != M2_NONE ? 1u : 0u;
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned q = qq[u];
                    unsigned wt = 1;
                    if (!UNITW) {
                        const bool v = q != M2_NONE;
                        const int c = min(c0 + u, n - 1);
                        const int xc = dna5_code(s_mem[c].seq[v ? ri[u] : 0u]);
                        const int xb = dna5_code(seqb[v ? q : 0u]);
                        const int wac = m2_w0(xa, xc, A.ma, A.mm), wcb = m2_w0(xc, xb, A.ma, A.mm);
                        wt = static_cast<unsigned>(wac < wcb ? wac : wcb);
                    }
                    const unsigned x0 = q ^ q0, x1 = q ^ q1, x2 = q ^ q2, x3 = q ^ q3;
                    w0 += x0 == 0u ? wt : 0u; w1 += x1 == 0u ? wt : 0u; w2 += x2 == 0u ? wt : 0u; w3 += x3 == 0u ? wt : 0u;
                    // nonzero: a position not seen yet (no slot agrees, the candidate is valid) and a free slot for it
                    const unsigned fresh = min(min(min(x0, x1), min(x2, x3)), min(q ^ M2_NONE, 3u - nalt));
                    if (__ballot(fresh != 0u)) {   // (among unrelated reads the three slots are taken after three candidates)
                        const bool f = fresh != 0u;
                        const bool t1 = f && nalt == 0, t2 = f && nalt == 1, t3 = f && nalt == 2;
                        q1 = t1 ? q : q1; w1 = t1 ? wt : w1;
                        q2 = t2 ? q : q2; w2 = t2 ? wt : w2;
                        q3 = t3 ? q : q3; w3 = t3 ? wt : w3;
                        nalt += f ? 1u : 0u;
                    }
                }
            }
            q1 &= 0xffffu; q2 &= 0xffffu; q3 &= 0xffffu;   // (an empty slot leaves as 0xFFFF)
            if (in) {
                uint32_t* const E = A.ext + Ma.ext_base + m2_pair_index(a, b, n) * Ma.lmax + p;
                if (UNITW) {
                    E[0] = w0 | (nalt << 6) | (w1 << 8) | (q1 << 16);
                    if (nalt >= 2) E[A.ext_off1] = w2 | (q2 << 16);
                    if (nalt >= 3) E[A.ext_off1 + A.ext_plane] = w3 | (q3 << 16);
                } else {
                    E[0] = w0 | (nalt << 16);
                    if (nalt >= 1) E[A.ext_off1] = w1 | (q1 << 16);
                    if (nalt >= 2) E[A.ext_off1 + A.ext_plane] = w2 | (q2 << 16);
                    if (nalt >= 3) E[A.ext_off1 + 2 * A.ext_plane] = w3 | (q3 << 16);
                }
            }
        }
    }
}

// Unit weights (the default scores): the same records from 256 positions per wavefront, FOUR consecutive positions per lane.
// The positions r of the lane's bases in a third read c are four consecutive entries of map(a -> c) -- one 8-byte load per
// read, staged in LDS once per window for every b --, and their partners in b four entries of map(c -> b) that lie within a
// few positions of each other: ONE 16-byte load at the first valid r covers r .. r + 7 (2-byte aligned: gfx950 takes it), a
// lane whose four positions span more than that (an insertion of more than four bases inside four positions) gathers the
// stragglers one by one.  So a (b, c) candidate costs one wide load per 256 positions instead of four 2-byte gathers, and the
// records leave as 16-byte stores.  (The first version, one position per lane and one gather per candidate like
// k_m2_extend<false> below, spent 150 ms per 10^6 reads waiting for its round trips; the request rate of its gathers alone
// would have allowed 15.)
struct __attribute__((packed, aligned(2))) m2_u16x4 { unsigned lo, hi; };
struct __attribute__((packed, aligned(2))) m2_u16x8 { unsigned x, y, z, w; };
struct __attribute__((packed, aligned(4))) m2_u32x4 { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned m2_pick16(const m2_u16x8& W, unsigned d) {   // entry d (0 .. 7) of a 16-byte window
    const unsigned long long lo = (static_cast<unsigned long long>(W.y) << 32) | W.x, hi = (static_cast<unsigned long long>(W.w) << 32) | W.z;
    const unsigned long long h = (d & 4u) ? hi : lo;
    return static_cast<unsigned>(h >> ((d & 3u) * 16u)) & 0xffffu;
}
constexpr int M2_XC = 4;   // third reads requested side by side
__global__ void __launch_bounds__(64) k_m2_extend_unit(M2Args A, int m0, int m1) {   // members [m0, m1)
    extern __shared__ __align__(16) unsigned char m2_xs[];
    const int m = m0 + static_cast<int>(blockIdx.y);
    if (m >= m1) return;
    const m2_mask F = A.first[m];
    const M2Member Ma = A.members[m];
    if (!F) return;
    const int n = Ma.n, fm = Ma.first_member, a = m - fm;
    const int lane = threadIdx.x;
    uint16_t* const s_r = reinterpret_cast<uint16_t*>(m2_xs) + 4 * lane;   // [c * 256]: this lane's four positions in read c
    const M2Member Ml = A.members[fm + min(lane, n - 1)];                   // lane c = member c
    const int nwin = (Ma.len + 255) / 256;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int P = w * 256 + 4 * lane;
        const int nin = max(0, min(4, Ma.len - P));   // of the lane's four positions, those inside the read
        for (int c = 0; c < n; ++c) {
            m2_u16x4 v;
            v.lo = v.hi = 0xFFFFFFFFu;
            if (c != a && nin > 0) {
                v = *reinterpret_cast<const m2_u16x4*>(A.map + Ma.map_base + static_cast<long long>(c < a ? c : c - 1) * Ma.len + P);
                if (nin < 4) {   // (the load ran into the next array: its slack, the next map, is readable)
                    v.hi = nin <= 2 ? 0xFFFFFFFFu : (v.hi | 0xFFFF0000u);
                    v.lo = nin <= 1 ? (v.lo | 0xFFFF0000u) : v.lo;
                }
            }
            *reinterpret_cast<m2_u16x4*>(s_r + c * 256) = v;
        }
        for (int b = 0; b < n; ++b) {
            if (!((F >> b) & 1ull)) continue;
            const m2_u16x4 dv = *reinterpret_cast<const m2_u16x4*>(s_r + b * 256);
            unsigned q0[4] = {dv.lo & 0xffffu, dv.lo >> 16, dv.hi & 0xffffu, dv.hi >> 16};
            unsigned wp[4], q1[4], q2[4], q3[4], nalt[4];   // wp: the four weights of a position, a byte each
#pragma unroll
            for (int k = 0; k < 4; ++k) { wp[k] = q0[k] != M2_NONE ? 1u : 0u; q1[k] = q2[k] = q3[k] = M2_NONE; nalt[k] = 0; }
            // the third reads: every member but a and b, M2_XC at a time
            const int lo_ab = min(a, b), hi_ab = max(a, b);
            for (int k0 = 0; k0 < n - 2; k0 += M2_XC) {
                m2_u16x8 W[M2_XC];
                m2_u16x4 rv[M2_XC];
                unsigned first[M2_XC];
#pragma unroll
                for (int u = 0; u < M2_XC; ++u) {
                    const bool use = k0 + u < n - 2;
                    int c = min(k0 + u, n - 3);
                    c += c >= lo_ab ? 1 : 0;
                    c += c >= hi_ab ? 1 : 0;
                    const int lenc = __builtin_amdgcn_readlane(Ml.len, c);
                    const long long basec = static_cast<long long>(m2_readlane64(static_cast<m2_u64>(Ml.map_base), c)) + static_cast<long long>(b < c ? b : b - 1) * lenc;
                    rv[u] = *reinterpret_cast<const m2_u16x4*>(s_r + c * 256);
                    if (!use) rv[u].lo = rv[u].hi = 0xFFFFFFFFu;
                    const unsigned r0 = rv[u].lo & 0xffffu, r1 = rv[u].lo >> 16, r2 = rv[u].hi & 0xffffu, r3 = rv[u].hi >> 16;
                    // (positions ascend along a pairwise alignment: the first valid one is the smallest)
                    first[u] = r0 != M2_NONE ? r0 : (r1 != M2_NONE ? r1 : (r2 != M2_NONE ? r2 : (r3 != M2_NONE ? r3 : 0u)));
                    W[u] = *reinterpret_cast<const m2_u16x8*>(A.map + basec + first[u]);
                }
#pragma unroll
                for (int u = 0; u < M2_XC; ++u) {
                    const unsigned rr[4] = {rv[u].lo & 0xffffu, rv[u].lo >> 16, rv[u].hi & 0xffffu, rv[u].hi >> 16};
                    unsigned q[4];
                    bool far = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned d = rr[k] - first[u];
                        q[k] = rr[k] != M2_NONE ? m2_pick16(W[u], d & 7u) : M2_NONE;
                        far = far || (rr[k] != M2_NONE && d > 7u);
                    }
                    if (__ballot(far)) {   // beyond the window: gathers of their own
                        int c = min(k0 + u, n - 3);
                        c += c >= lo_ab ? 1 : 0;
                        c += c >= hi_ab ? 1 : 0;
                        const int lenc = __builtin_amdgcn_readlane(Ml.len, c);
                        const long long basec = static_cast<long long>(m2_readlane64(static_cast<m2_u64>(Ml.map_base), c)) + static_cast<long long>(b < c ? b : b - 1) * lenc;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (rr[k] != M2_NONE && rr[k] - first[u] > 7u) q[k] = A.map[basec + rr[k]];
                    }
                    bool other = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool m0 = q[k] != M2_NONE && q[k] == q0[k];
                        wp[k] += m0 ? 1u : 0u;
                        other = other || (q[k] != M2_NONE && !m0);
                    }
                    if (__ballot(other)) {   // (same-molecule reads: nearly every triplet names the direct partner)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool v = q[k] != M2_NONE && q[k] != q0[k];
                            const bool m1 = v && q[k] == q1[k], m2 = v && q[k] == q2[k], m3 = v && q[k] == q3[k];
                            wp[k] += (m1 ? 0x100u : 0u) + (m2 ? 0x10000u : 0u) + (m3 ? 0x1000000u : 0u);
                            const bool fresh = v && !m1 && !m2 && !m3 && nalt[k] < 3u;   // a position not seen yet and a free slot for it
                            if (__ballot(fresh)) {
                                const bool t1 = fresh && nalt[k] == 0, t2 = fresh && nalt[k] == 1, t3 = fresh && nalt[k] == 2;
                                wp[k] += (t1 ? 0x100u : 0u) + (t2 ? 0x10000u : 0u) + (t3 ? 0x1000000u : 0u);
                                q1[k] = t1 ? q[k] : q1[k];
                                q2[k] = t2 ? q[k] : q2[k];
                                q3[k] = t3 ? q[k] : q3[k];
                                nalt[k] += fresh ? 1u : 0u;
                            }
                        }
                    }
                }
            }
            uint32_t* const E = A.ext + Ma.ext_base + m2_pair_index(a, b, n) * Ma.lmax + P;
            unsigned e0[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e0[k] = (wp[k] & 63u) | (nalt[k] << 6) | (((wp[k] >> 8) & 255u) << 8) | (q1[k] << 16);
            if (nin == 4) {
                m2_u32x4 o;
                o.x = e0[0]; o.y = e0[1]; o.z = e0[2]; o.w = e0[3];
                *reinterpret_cast<m2_u32x4*>(E) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nin) E[k] = e0[k];
            }
            if (__ballot(nin > 0 && (nalt[0] | nalt[1] | nalt[2] | nalt[3]) >= 2u)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nin && nalt[k] >= 2u) E[A.ext_off1 + k] = ((wp[k] >> 16) & 255u) | (q2[k] << 16);
                    if (k < nin && nalt[k] >= 3u) E[A.ext_off1 + A.ext_plane + k] = (wp[k] >> 24) | (q3[k] << 16);
                }
            }
        }
    }
}

template <typename MASK>
struct M2JoinT {      // wave-uniform description of one join (MASK: unsigned for groups of up to 32 reads, m2_mask beyond)
    int n, fm;
    MASK maskA, maskB;
    int nA, nB;
};
__device__ __forceinline__ int m2_popc(unsigned m) { return __popc(m); }
__device__ __forceinline__ int m2_popc(m2_mask m) { return __popcll(m); }
__device__ __forceinline__ int m2_ctz(unsigned m) { return __builtin_ctz(m); }
__device__ __forceinline__ int m2_ctz(m2_mask m) { return __builtin_ctzll(m); }

// A row's list -> filter, order by column, append to the match list.  ej[k] < 0: empty slot.
// (macro: the lists must stay in registers, every index a compile-time constant)
#define M2_FINISH_ROW()                                                                                                  \
    {                                                                                                                    \
        int wmax = 0;                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < M2_CAP; ++k) wmax = max(wmax, ew[k]);                                      \
        int kept = 0;                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < M2_CAP; ++k) {                                                             \
            /* noise filter (spec v2, step 5): entries lighter than half the row's heaviest are dropped */              \
            const bool keep = k < cnt && 2 * ew[k] >= wmax;                                                              \
            if (!keep) ej[k] = -1;                                                                                       \
            kept += keep ? 1 : 0;                                                                                        \
        }                                                                                                                \
        if (!row) kept = 0;                                                                                              \
        st_filtered += row ? static_cast<unsigned>(cnt - kept) : 0u;                                                     \
        st_rowsf += (row && kept < cnt) ? 1u : 0u;                                                                       \
        const int incl = m2_incl_sum(kept);                                                                        \
        m2_u64* const mine = ent + ne + (incl - kept);                                                                   \
        if (row) {                                                                                                       \
            _Pragma("unroll") for (int k = 0; k < M2_CAP; ++k) {                                                         \
                if (ej[k] >= 0) {   /* by column: rank of the entry among the kept ones (columns are distinct) */       \
                    int rank = 0;                                                                                        \
                    _Pragma("unroll") for (int q = 0; q < M2_CAP; ++q) rank += (ej[q] >= 0 && ej[q] < ej[k]) ? 1 : 0;    \
                    mine[rank] = (static_cast<m2_u64>(static_cast<unsigned>(i)) << 48) |                                 \
                                 (static_cast<m2_u64>(static_cast<unsigned>(ej[k])) << 32) | static_cast<unsigned>(ew[k]); \
                }                                                                                                        \
            }                                                                                                            \
            part[i] = -1;                                                                                                \
        }                                                                                                                \
        ne += __builtin_amdgcn_readlane(incl, 63);                                                                       \
    }

// ---- rows of one join: the library of its member pairs summed per column of the first child ----
// Lane = column i of the first child.  For its members a (ascending) with a base p in this column and the second child's
// members b (ascending): the record of (a, b, p) -- the direct partner from the map, the others from the extension --, each
// partner position turned into its column through col[] and added to the lane's private list of at most M2_CAP partner columns
// (spec v2, step 5: the first M2_CAP distinct columns in this order count).
// Unit weights: a list entry is one register, (column << 16) | weight (a column is < 65535, a weight at most
// |A| |B| (n - 1) <= 64512); 0xFFFF0000 is an empty slot, an invalid candidate carries a column no entry can hold.
// The columns of a list are distinct, so at most one entry matches.  Three tiers, each behind a wave-wide test:
// entries 0-3; entries 4-7 and the append into 0-7; entries 8-15.
constexpr unsigned M2_EMPTY = 0xFFFF0000u;
#ifndef M2_RB1
#define M2_RB1 1   // members of the second child whose records a row requests together: one wavefront per group (64 registers) ...
#endif
#ifndef M2_RBN
#define M2_RBN 1   // ... and the workgroups of several wavefronts.  (2 / 4 and 4 / 4: 490 and 563 ms against 449 at bench.py's pipeline workload --
                   // the registers of a batch cost more resident wavefronts than its shorter chain of round trips gains)
#endif
#define M2_MATCH1(K0, K1)                                                                              \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                           \
        const bool m_ = (pe[k_] >> 16) == j_;                                                          \
        pe[k_] += m_ ? w_ : 0u;                                                                        \
        hit_ = hit_ || m_;                                                                             \
    }
#define M2_APPEND1(K0, K1)                                                                             \
    {                                                                                                  \
        const bool app_ = !hit_ && cnt < (K1);                                                         \
        _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) pe[k_] = (app_ && cnt == k_) ? ((j_ << 16) | w_) : pe[k_]; \
        cnt += app_ ? 1 : 0;                                                                           \
        hit_ = hit_ || app_;                                                                           \
    }
#define M2_ADD1(J, W, VALID)                                                                           \
    {                                                                                                  \
        const bool v_ = (VALID);                                                                       \
        const unsigned j_ = v_ ? static_cast<unsigned>(J) : 0x1FFFFu;                                  \
        const unsigned w_ = static_cast<unsigned>(W);                                                  \
        bool hit_ = !v_;                                                                               \
        M2_MATCH1(0, 4)                                                                                \
        if (__ballot(!hit_)) {                                                                         \
            M2_MATCH1(4, 8)                                                                            \
            M2_APPEND1(0, 8)                                                                           \
            if (__ballot(!hit_)) {                                                                     \
                M2_MATCH1(8, M2_CAP)                                                                   \
                /* a new column; beyond M2_CAP distinct columns it is ignored (spec v2, step 5) */     \
                M2_APPEND1(8, M2_CAP)                                                                  \
                capped = capped || !hit_;                                                              \
            }                                                                                          \
        }                                                                                              \
    }
// any weights: columns and weights in registers of their own
#define M2_ADD(J, W, VALID)                                                                            \
    {                                                                                                  \
        const int j_ = static_cast<int>(J), w_ = static_cast<int>(W);                                  \
        bool hit_ = !(VALID);                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                             \
            const bool m_ = !hit_ && ej[k_] == j_;                                                     \
            ew[k_] += m_ ? w_ : 0;                                                                     \
            hit_ = hit_ || m_;                                                                         \
        }                                                                                              \
        if (__ballot(!hit_)) {                                                                         \
            _Pragma("unroll") for (int k_ = 3; k_ < M2_CAP; ++k_) {                                    \
                const bool m_ = !hit_ && ej[k_] == j_;                                                 \
                ew[k_] += m_ ? w_ : 0;                                                                 \
                hit_ = hit_ || m_;                                                                     \
            }                                                                                          \
            /* a new column; beyond M2_CAP distinct columns it is ignored (spec v2, step 5) */         \
            const bool app_ = !hit_ && cnt < M2_CAP;                                                   \
            capped = capped || (!hit_ && !app_);                                                       \
            _Pragma("unroll") for (int k_ = 0; k_ < M2_CAP; ++k_) {                                    \
                const bool s_ = app_ && k_ == cnt;                                                     \
                ej[k_] = s_ ? j_ : ej[k_];                                                             \
                ew[k_] = s_ ? w_ : ew[k_];                                                             \
            }                                                                                          \
            cnt += app_ ? 1 : 0;                                                                       \
        }                                                                                              \
    }

template <bool UNITW, int M2_RB, typename MASK>
__device__ __forceinline__ int m2_rows_ext(const M2Args& A, const M2Group& G, const M2JoinT<MASK>& J, int i_lo, int i_hi, m2_u64* ent, int* part,
                                           unsigned& st_capped, unsigned& st_filtered, unsigned& st_rowsf) {
    const int lane = m2_lane();
    const int n = J.n, fm = J.fm, nA = J.nA;
    const bool leafB = m2_popc(J.maskB) == 1;   // (a single member in the second child is a leaf of the guide tree: col = position)
    int ne = 0;
    for (int i0 = i_lo; i0 < i_hi; i0 += 64) {
        const int i = i0 + lane;
        unsigned pe[M2_CAP];
        int ej[M2_CAP], ew[M2_CAP];   // (the instantiation's other list is never touched and disappears)
        if constexpr (UNITW) {
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) pe[k] = M2_EMPTY;
        } else {
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) { ej[k] = -1; ew[k] = 0; }
        }
        int cnt = 0;
        bool capped = false;
        const bool row = i < nA;
        // the first child's members in ascending order; the positions of the NEXT member are requested before this one's pairs
        MASK restA = J.maskA;
        int a_next = m2_ctz(restA);
        restA &= restA - 1;
        unsigned p_next = row ? A.pos[G.pos_base + static_cast<long long>(a_next) * G.wcap + i] : M2_NONE;
        for (bool more_a = true; more_a;) {
            const int a = a_next;
            const unsigned p = p_next;
            more_a = restA != 0;
            if (more_a) {
                a_next = m2_ctz(restA);
                restA &= restA - 1;
                p_next = row ? A.pos[G.pos_base + static_cast<long long>(a_next) * G.wcap + i] : M2_NONE;
            }
            const bool havep = p != M2_NONE;
            if (!__ballot(havep)) continue;
            const M2Member Ma = A.members[fm + a];
            const unsigned pidx = havep ? p : 0u;
            // the second child's members in ascending order, M2_RB at a time: their map entries and records are requested together,
            // then the partners' columns and the further records, then the columns of those -- three round trips per M2_RB members
            // (one member at a time, each with up to six dependent round trips, left the wavefront waiting most of the time)
#define M2_ADD_ANY(Jx, Wx, Vx) { if constexpr (UNITW) M2_ADD1(Jx, Wx, Vx) else M2_ADD(Jx, Wx, Vx) }
            MASK rest = J.maskB;
            while (rest) {
                int bb[M2_RB], nbb = 0;
#pragma unroll
                for (int u = 0; u < M2_RB; ++u) {
                    const bool has = rest != 0;
                    bb[u] = has ? m2_ctz(rest) : bb[u > 0 ? u - 1 : 0];
                    rest = has ? (rest & (rest - 1)) : rest;
                    nbb += has ? 1 : 0;
                }
                unsigned q0[M2_RB], e0[M2_RB];
#pragma unroll
                for (int u = 0; u < M2_RB; ++u) {   // (a batch that is not full repeats its last member: harmless loads)
                    const int b = bb[u];
                    q0[u] = A.map[Ma.map_base + static_cast<long long>(b < a ? b : b - 1) * Ma.len + pidx];
                    e0[u] = A.ext[G.ext_base + m2_pair_index(a, b, n) * G.lmax + pidx];
                }
                unsigned j0[M2_RB], x1[M2_RB], x2[M2_RB], x3[M2_RB], nal[M2_RB];
                bool any1[M2_RB], any2[M2_RB], any3[M2_RB];
#pragma unroll
                for (int u = 0; u < M2_RB; ++u) {
                    const M2Member Mb = A.members[fm + bb[u]];
                    const uint16_t* const colb = A.col + Mb.col_base;
                    const unsigned lenb1 = static_cast<unsigned>(max(Mb.len - 1, 0));
                    const uint32_t* const E = A.ext + G.ext_base + m2_pair_index(a, bb[u], n) * G.lmax + pidx;
                    nal[u] = (havep && u < nbb) ? (UNITW ? ((e0[u] >> 6) & 3u) : (e0[u] >> 16)) : 0u;
                    any1[u] = __ballot(nal[u] >= 1u) != 0; any2[u] = __ballot(nal[u] >= 2u) != 0; any3[u] = __ballot(nal[u] >= 3u) != 0;
                    j0[u] = leafB ? q0[u] : colb[min(q0[u], lenb1)];
                    x1[u] = x2[u] = x3[u] = 0;
                    if (UNITW) {
                        const unsigned q1 = e0[u] >> 16;
                        if (any1[u]) x1[u] = ((leafB ? q1 : colb[min(q1, lenb1)]) << 16) | ((e0[u] >> 8) & 255u);
                    } else if (any1[u]) x1[u] = nal[u] >= 1u ? E[A.ext_off1] : 0u;
                    if (any2[u]) x2[u] = nal[u] >= 2u ? E[A.ext_off1 + (UNITW ? 0 : 1) * A.ext_plane] : 0u;
                    if (any3[u]) x3[u] = nal[u] >= 3u ? E[A.ext_off1 + (UNITW ? 1 : 2) * A.ext_plane] : 0u;
                }
#pragma unroll
                for (int u = 0; u < M2_RB; ++u) {   // the columns of the further partners: x = (column << 16) | weight from here on
                    const M2Member Mb = A.members[fm + bb[u]];
                    const uint16_t* const colb = A.col + Mb.col_base;
                    const unsigned lenb1 = static_cast<unsigned>(max(Mb.len - 1, 0));
                    if (!UNITW && any1[u]) x1[u] = ((leafB ? (x1[u] >> 16) : colb[min(x1[u] >> 16, lenb1)]) << 16) | (x1[u] & 0xffffu);
                    if (any2[u]) x2[u] = ((leafB ? (x2[u] >> 16) : colb[min(x2[u] >> 16, lenb1)]) << 16) | (x2[u] & 0xffffu);
                    if (any3[u]) x3[u] = ((leafB ? (x3[u] >> 16) : colb[min(x3[u] >> 16, lenb1)]) << 16) | (x3[u] & 0xffffu);
                }
#pragma unroll
                for (int u = 0; u < M2_RB; ++u) {
                    if (u >= nbb) break;
                    const unsigned w0 = UNITW ? (e0[u] & 63u) : (e0[u] & 0xffffu);
                    M2_ADD_ANY(j0[u], w0, havep && q0[u] != M2_NONE)
                    if (any1[u]) {
                        M2_ADD_ANY(x1[u] >> 16, x1[u] & 0xffffu, nal[u] >= 1u)
                        if (any2[u]) M2_ADD_ANY(x2[u] >> 16, x2[u] & 0xffffu, nal[u] >= 2u)
                        if (any3[u]) M2_ADD_ANY(x3[u] >> 16, x3[u] & 0xffffu, nal[u] >= 3u)
                    }
                }
            }
#undef M2_ADD_ANY
        }
        st_capped += (row && capped) ? 1u : 0u;
        if constexpr (UNITW) {   // filter, order by column, append -- on the packed entries: (column << 16) | weight orders by column
            unsigned wmax = 0;
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) wmax = max(wmax, k < cnt ? (pe[k] & 0xffffu) : 0u);
            int kept = 0;
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) {
                // noise filter (spec v2, step 5): entries lighter than half the row's heaviest are dropped
                const bool keep = k < cnt && 2u * (pe[k] & 0xffffu) >= wmax;
                pe[k] = keep ? pe[k] : 0xFFFFFFFFu;   // (sorts behind every kept entry)
                kept += keep ? 1 : 0;
            }
            if (!row) kept = 0;
            st_filtered += row ? static_cast<unsigned>(cnt - kept) : 0u;
            st_rowsf += (row && kept < cnt) ? 1u : 0u;
            const int incl = m2_incl_sum(kept);
            m2_u64* const mine = ent + ne + (incl - kept);
            if (row) {
                const m2_u64 rowbits = static_cast<m2_u64>(static_cast<unsigned>(i)) << 48;
#pragma unroll
                for (int k = 0; k < M2_CAP; ++k) {
                    const unsigned ek = pe[k];
                    if (ek != 0xFFFFFFFFu) {   // rank of the entry among the kept ones (columns are distinct)
                        int rank = 0;
#pragma unroll
                        for (int q = 0; q < M2_CAP; ++q) rank += pe[q] < ek ? 1 : 0;
                        mine[rank] = rowbits | (static_cast<m2_u64>(ek >> 16) << 32) | (ek & 0xffffu);
                    }
                }
                part[i] = -1;
            }
            ne += __builtin_amdgcn_readlane(incl, 63);
        } else {
            M2_FINISH_ROW()
        }
    }
    return ne;
}
#undef M2_ADD1
#undef M2_MATCH1
#undef M2_APPEND1
#undef M2_ADD
#undef M2_FINISH_ROW

// ---- heaviest chain of one join over the match list, 64 matches per step ----
// f(m) = w(m) + the best f over the matches with a smaller row AND a smaller column, "best" = larger f, then the
// earlier match (oracle/msa2.c): a node is (f << 32) | ~id with id = 1 + the match's index in the list (row-major,
// columns ascending inside a row), so an unsigned maximum picks it.  All matches of a row see the state before the row.
//   Q[c] = best node among the matches entered so far with column <= c, kept for the columns up to the front F (the
// largest column entered; everything beyond it equals Q[F], one scalar).  A block of up to 64 matches that ends on a
// row boundary is (1) looked up in Q -- the matches of earlier blocks; (2) resolved against each other in registers:
// lanes are in list order, so when lane s broadcasts its finished node every lane with a larger row and a larger column
// takes the maximum, and lane s + 1 is finished in turn; (3) entered: Q is extended to the new front, the block's nodes
// are written with LDS maxima and the prefix maximum is re-established from the block's smallest column to the front
// by a DPP scan per 64 columns.  With RING the columns live in an LDS ring of M2_QW entries behind the front; a block
// that reaches further back (unrelated reads in the cluster) makes the function return false and the join's chain is
// redone with Q over all columns in HBM (RING = false; atomics and L2-coherent loads, rare).
constexpr int M2_QW = 512;

// The match list of a join comes in segments: wavefront w of the workgroup wrote the rows of its range at
// ent + w * stride, cnt[w] matches; pfx[w] = matches before segment w (the index of a match counts through the segments).
struct M2Segs {
    int nseg;
    long long stride;
    const int* cnt;   // LDS
    const int* pfx;   // LDS, nseg + 1 entries
};

// (single wavefront: LDS operations of a wavefront execute in order, so no barrier is needed between them -- only the
// compiler must keep the order; the HBM variant waits for its stores and atomics)
template <bool RING>
__device__ __forceinline__ void m2_chain_order() {
    if (!RING) __threadfence();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool RING>
__device__ __forceinline__ bool m2_chain_forward(const m2_u64* ent, const M2Segs& S, unsigned* pred, m2_u64* Q, m2_u64& tail, int& err) {
    const int lane = m2_lane();
    int F = -1;
    m2_u64 QF = 0;
    auto q_load = [&](int c) -> m2_u64 {
        if (RING) return Q[c & (M2_QW - 1)];
        return __hip_atomic_load(&Q[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto q_store = [&](int c, m2_u64 v) {
        if (RING) Q[c & (M2_QW - 1)] = v;
        else __hip_atomic_store(&Q[c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (int sg = 0; sg < S.nseg; ++sg) {
        const m2_u64* const es = ent + sg * S.stride;
        const int ne = m2_rfl(S.cnt[sg]);
        const int base = m2_rfl(S.pfx[sg]);
        int m0 = 0;
        while (m0 < ne) {
            const int m = m0 + lane;
            const bool in = m < ne;
            const m2_u64 e = es[min(m, ne - 1)];
            // a block ends on a row boundary: the matches of the row that continues into the next block wait for it
            // (a segment ends on one: segments are ranges of rows)
            const int inext = m0 + 64 < ne ? static_cast<int>(es[m0 + 64] >> 48) : -1;
            const int i = static_cast<int>(e >> 48), j = static_cast<int>((e >> 32) & 0xffffu);
            const unsigned w = static_cast<unsigned>(e);
            const bool act = in && i != inext;
            const int nb = __popcll(__ballot(act));   // (the waiting matches are the last lanes: the list is in row order)
            if (nb == 0) { err = 1; return true; }   // cannot happen: a row holds at most M2_CAP < 64 matches
            const int jmax = m2_wave_max(act ? j : -1), jmin = m2_wave_min(act ? j : 0x7fffffff);
            const int newF = max(F, jmax);
            if (RING && newF - jmin >= M2_QW - 2) return false;
            // (1) state before the block
            m2_u64 best = 0;
            if (act && j > 0) best = (j - 1 > F) ? QF : q_load(j - 1);
            // (2) the block's own matches, in list order
            const unsigned ij = (static_cast<unsigned>(i) << 16) | static_cast<unsigned>(j);
            for (int s = 0; s < nb; ++s) {
                const unsigned bh = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(best >> 32), s));
                const unsigned ws = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(w), s));
                const unsigned ijs = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(ij), s));
                const m2_u64 nvs = (static_cast<m2_u64>(bh + ws) << 32) | static_cast<unsigned>(~static_cast<unsigned>(base + m0 + s + 1));
                const bool dom = act && i > static_cast<int>(ijs >> 16) && j > static_cast<int>(ijs & 0xffffu);
                best = (dom && nvs > best) ? nvs : best;
            }
            const unsigned id = static_cast<unsigned>(base + m) + 1u;
            const m2_u64 nv = act ? ((static_cast<m2_u64>(w + static_cast<unsigned>(best >> 32)) << 32) | static_cast<unsigned>(~id)) : 0ull;
            if (act) pred[base + m] = best ? ~static_cast<unsigned>(best) : 0u;
            // (3) enter the block
            for (int c0 = RING ? max(F + 1, newF - M2_QW + 1) : F + 1; c0 <= newF; c0 += 64)
                if (c0 + lane <= newF) q_store(c0 + lane, QF);
            m2_chain_order<RING>();
            if (act) {
                if (RING) atomicMax(&Q[j & (M2_QW - 1)], nv);
                else atomicMax(&Q[j], nv);
            }
            m2_chain_order<RING>();
            m2_u64 carry = jmin > 0 ? q_load(jmin - 1) : 0ull;
            for (int c0 = jmin; c0 <= newF; c0 += 64) {
                const int c = c0 + lane;
                m2_u64 x = c <= newF ? q_load(c) : 0ull;
                x = m2_scan_max64(x);
                x = x > carry ? x : carry;
                if (c <= newF) q_store(c, x);
                carry = m2_readlane64(x, 63);
            }
            m2_chain_order<RING>();
            QF = carry;
            F = newF;
            m0 += nb;
        }
    }
    tail = QF;
    return true;
}

// walk back through the predecessors: part[row] = column for the matches of the chain
__device__ __forceinline__ void m2_chain_walk(const m2_u64* ent, const M2Segs& S, const unsigned* pred, m2_u64 tail, int* part, int& err) {
    const int lane = m2_lane();
    unsigned cur = tail ? ~static_cast<unsigned>(tail) : 0u;
    cur = static_cast<unsigned>(m2_rfl(static_cast<int>(cur)));
    int sg = S.nseg - 1;
    int steps = m2_rfl(S.pfx[S.nseg]) + 2;   // a predecessor has a smaller index: the walk visits every match at most once
    while (cur != 0u && steps > 0) {
        while (sg > 0 && static_cast<int>(cur - 1u) < m2_rfl(S.pfx[sg])) --sg;
        const int base = m2_rfl(S.pfx[sg]), ne = m2_rfl(S.cnt[sg]);
        const int mb = (static_cast<int>(cur - 1u) - base) & ~63;   // block of 64 matches inside the segment
        const int m = min(mb + lane, ne - 1);
        const m2_u64 e = ent[sg * S.stride + m];
        const unsigned p = pred[base + m];
        unsigned long long visited = 0;
        do {
            const int l = static_cast<int>(cur - 1u) - base - mb;
            visited |= 1ull << l;
            cur = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(p), l));
            --steps;
        } while (cur != 0u && static_cast<int>(cur - 1u) >= base + mb && steps > 0);
        if ((visited >> lane) & 1ull) part[static_cast<int>(e >> 48)] = static_cast<int>((e >> 32) & 0xffffu);
    }
    if (cur != 0u) err = 2;
}

// ---- new column numbers, col / pos of every member: the scans on the first wavefront, the rest on all NW ----
// returns the width of the joined profile, or -1 when it exceeds the capacity
template <int NW, typename MASK>
__device__ __forceinline__ int m2_renumber(const M2Args& A, const M2Group& G, const M2JoinT<MASK>& J, const int* part, int* nca, int* ncb, int* pb,
                                           int* s_newW) {
    const int lane = m2_lane();
    const int wave = m2_rfl(static_cast<int>(threadIdx.x >> 6));
    const int tid = wave * 64 + lane;
    constexpr int NT = 64 * NW;
    const int n = J.n, fm = J.fm, nA = J.nA, nB = J.nB;
    // partner rows of the second child's columns
    for (int j = tid; j < nB; j += NT) pb[j] = -1;
    __threadfence_block();
    __syncthreads();
    for (int i0 = tid; i0 < nA; i0 += 4 * NT) {   // (four loads in flight; see the scatter below)
        int pj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pj[u] = i0 + u * NT < nA ? part[i0 + u * NT] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pj[u] >= 0) pb[pj[u]] = i0 + u * NT;
    }
    __threadfence_block();
    __syncthreads();
    if (wave == 0) {
        // first child: column i -> i + (columns of the second child up to the previous matched partner) - matches before
        // (in the three scans the NEXT block's entry is requested before this block's is used: the running values -- matches so
        // far, last partner -- chain the blocks, the loads need not wait for them)
        int jprev = -1, t0 = 0;
        int pj_next = lane < nA ? part[lane] : -1;
        for (int i0 = 0; i0 < nA; i0 += 64) {
            const int i = i0 + lane;
            const int pj = pj_next;
            pj_next = i + 64 < nA ? part[i + 64] : -1;
            const unsigned long long ball = __ballot(pj >= 0);
            const int before = __popcll(ball & ((1ull << lane) - 1ull));
            const int pm = max(m2_excl_max(pj, -1), jprev);
            const int t = t0 + before;
            if (i < nA) nca[i] = pj >= 0 ? i + pj - t : i + (pm + 1) - t;
            t0 += __popcll(ball);
            jprev = max(jprev, m2_wave_max(pj));
        }
        const int nm = t0;
        // second child: column j -> (row of the next matched pair, or nA) + j - matches before; descending for "next"
        int inext = nA;
        const int jtop = ((nB - 1) / 64) * 64;
        int pi_next = (nB > 0 && jtop + 63 - lane < nB) ? pb[jtop + 63 - lane] : -1;
        for (int j0 = jtop; j0 >= 0 && nB > 0; j0 -= 64) {
            const int j = j0 + 63 - lane;     // (descending over the lanes: "at or after j" is a prefix)
            const int pi = pi_next;
            pi_next = j0 >= 64 ? pb[j - 64] : -1;   // (a full block: every lane inside)
            const int sc = m2_scan_i32<2>(pi >= 0 ? pi : 0x7fffffff, 0x7fffffff);
            if (j < nB) ncb[j] = min(sc, inext);          // provisional: the row of the next matched pair at or after j
            inext = min(inext, __builtin_amdgcn_readlane(sc, 63));
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        int tb = 0;
        int pi2_next = lane < nB ? pb[lane] : -1, nc_next = lane < nB ? ncb[lane] : 0;
        for (int j0 = 0; j0 < nB; j0 += 64) {
            const int j = j0 + lane;
            const int pi = pi2_next, ncj = nc_next;
            pi2_next = j + 64 < nB ? pb[j + 64] : -1;
            nc_next = j + 64 < nB ? ncb[j + 64] : 0;
            const unsigned long long ball = __ballot(pi >= 0);
            const int before = tb + __popcll(ball & ((1ull << lane) - 1ull));
            if (j < nB) ncb[j] = ncj + j - before;   // matched: next = its own row
            tb += __popcll(ball);
        }
        if (lane == 0) *s_newW = nA + nB - nm;
    }
    __threadfence_block();
    __syncthreads();
    const int newW = m2_rfl(*s_newW);
    if (newW > G.wcap) return -1;
    // clear the members' rows of pos, then scatter the new columns
    const MASK both = J.maskA | J.maskB;
    for (int a = 0; a < n; ++a) {
        if (!((both >> a) & 1)) continue;
        uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
        for (int c = tid; c < newW; c += NT) row[c] = static_cast<uint16_t>(M2_NONE);
    }
    __threadfence_block();
    __syncthreads();
    for (int a = 0; a < n; ++a) {
        const bool inA = (J.maskA >> a) & 1;
        if (!((both >> a) & 1)) continue;
        const M2Member Me = A.members[fm + a];
        const int* nc = inA ? nca : ncb;
        uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
        // (four independent lookups in flight: col -> new number is a chain of two round trips, and one position per trip left the
        // wavefront waiting through len / 64 of them per member)
        for (int p0 = tid; p0 < Me.len; p0 += 4 * NT) {
            unsigned oc[4];
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) oc[u] = A.col[Me.col_base + min(p0 + u * NT, Me.len - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = nc[oc[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * NT;
                if (p < Me.len) {
                    A.col[Me.col_base + p] = static_cast<uint16_t>(c[u]);
                    row[c[u]] = static_cast<uint16_t>(p);
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    return newW;
}

// LDS of a workgroup: the chain's ring (first wavefront).  (Until round 5 the rows phase staged positions and candidate
// tables here: 5 KB per wavefront, 37 KB for the workgroups of 33 .. 64 reads.)
constexpr int M2_LDS = M2_QW * 8;

// One workgroup of NW wavefronts per group (groups of up to NMAX reads; NW = 1 takes any).  The rows of a join are
// cut into NW ranges, one per wavefront; the chain runs on the first wavefront; the renumbering's copies on all.
template <bool UNITW, int NW, int NMAX>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 1 ? (UNITW ? M2_WAVES_EU : 4) : 4, 8))) k_m2_group(M2Args A) {
    __shared__ __align__(16) unsigned char smem[M2_LDS];
    __shared__ int s_cnt[NW], s_pfx[NW + 1], s_ctl[4];
    typedef typename std::conditional<(NMAX > 32), m2_mask, unsigned>::type MASK;
    constexpr bool TWO = NMAX > 32;
    const int lane = threadIdx.x & 63;
    const int wave = m2_rfl(static_cast<int>(threadIdx.x >> 6));
    const long long wb = static_cast<long long>(blockIdx.x) * A.w_rows;
    m2_u64* const ent = A.w_ent + wb * M2_CAP;
    unsigned* const pred = A.w_pred + wb * M2_CAP;
    int* const part = A.w_part + wb;
    int* const nca = A.w_nca + wb;
    int* const ncb = A.w_ncb + wb;
    int* const pb = A.w_pb + wb;
    m2_u64* const qg = A.w_q + wb;
    unsigned st_capped = 0, st_filtered = 0, st_rowsf = 0;
    unsigned long long st_rows = 0, st_kept = 0, st_joins = 0, st_hbmq = 0, st_gath = 0;
    unsigned long long cy_rows = 0, cy_chain = 0, cy_walk = 0, cy_renum = 0;
    if (threadIdx.x == 0) atomicMin(&A.counters[M2C_T_START], __builtin_amdgcn_s_memrealtime());
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_ctl[0] = A.g0 + atomicAdd(A.next, 1);
        __syncthreads();
        const int g = m2_rfl(s_ctl[0]);
        if (g >= A.g1) break;
        const M2Group G = A.groups[g];
        const int n = G.n, fm = G.first_member;
        if (n < 2) continue;
        // the nodes of the guide tree live in the lanes (of every wavefront): lane k = node k (leaves 0 .. n - 1, join k creates n + k)
        // (a group of up to 32 reads has at most 63 nodes: one per lane, 32-bit member masks; beyond, nodes 64 .. 126 live in a
        // second pair of registers and the masks have 64 bits)
        MASK nmask = lane < n ? (static_cast<MASK>(1) << lane) : static_cast<MASK>(0), nmask2 = 0;
        int ncols = lane < n ? A.members[fm + lane].len : 0, ncols2 = 0;
        int err = 0, width = 0;
        bool over = n > NMAX;   // (the host sends a group to an instantiation that holds it)
        for (int round = 0; round + 1 < n && !over; ++round) {
            const int2 jn = A.joins[fm + round];
            const int jx = m2_rfl(jn.x), jy = m2_rfl(jn.y);
            M2JoinT<MASK> J;
            J.n = n; J.fm = fm;
            J.maskA = (TWO && jx >= 64) ? m2_readlane(nmask2, jx - 64) : m2_readlane(nmask, jx);
            J.maskB = (TWO && jy >= 64) ? m2_readlane(nmask2, jy - 64) : m2_readlane(nmask, jy);
            J.nA = (TWO && jx >= 64) ? __builtin_amdgcn_readlane(ncols2, jx - 64) : __builtin_amdgcn_readlane(ncols, jx);
            J.nB = (TWO && jy >= 64) ? __builtin_amdgcn_readlane(ncols2, jy - 64) : __builtin_amdgcn_readlane(ncols, jy);
            __syncthreads();
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            // ---- rows: wavefront w takes the blocks of 64 rows [w bpw, (w + 1) bpw) and writes its matches at ent + w stride ----
            const int bpw = ((J.nA + 63) / 64 + NW - 1) / NW;
            const long long stride = static_cast<long long>(bpw) * 64 * M2_CAP;
            const int i_lo = min(wave * bpw * 64, J.nA), i_hi = min((wave + 1) * bpw * 64, J.nA);
            const int ne = m2_rows_ext<UNITW, (NW == 1 ? M2_RB1 : M2_RBN), MASK>(A, G, J, i_lo, i_hi, ent + wave * stride, part, st_capped, st_filtered, st_rowsf);
            // wave-wide gather instructions of this wavefront's rows: per block of 64 columns and member of the first child its
            // positions, then per member of the second child the direct partner, the record and the partner's column (further
            // partner positions of a record: not counted, a lower bound)
            st_gath += static_cast<unsigned long long>((i_hi - i_lo + 63) / 64) * static_cast<unsigned>(m2_popc(J.maskA)) *
                       static_cast<unsigned>(1 + m2_popc(J.maskB) * (m2_popc(J.maskB) == 1 ? 2 : 3));
            if (lane == 0) s_cnt[wave] = ne;
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 0) {
                int acc = 0;
                for (int w = 0; w < NW; ++w) { s_pfx[w] = acc; acc += s_cnt[w]; }
                s_pfx[NW] = acc;
            }
            __syncthreads();
            const int ne_all = m2_rfl(s_pfx[NW]);
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            unsigned long long t2 = t1;
            if (wave == 0) {
                st_rows += static_cast<unsigned>(J.nA);
                st_kept += static_cast<unsigned>(ne_all);
                ++st_joins;
                if (ne_all > 0) {
                    M2Segs S;
                    S.nseg = NW; S.stride = stride; S.cnt = s_cnt; S.pfx = s_pfx;
                    m2_u64 tail = 0;
                    m2_u64* const ring = reinterpret_cast<m2_u64*>(smem);
                    if (A.chain_hbm || !m2_chain_forward<true>(ent, S, pred, ring, tail, err)) {
                        ++st_hbmq;
                        m2_chain_order<false>();
                        m2_chain_forward<false>(ent, S, pred, qg, tail, err);
                    }
                    m2_chain_order<false>();
                    t2 = __builtin_amdgcn_s_memtime();
                    m2_chain_walk(ent, S, pred, tail, part, err);
                }
                if (lane == 0) s_ctl[1] = err;
            }
            __threadfence_block();
            __syncthreads();
            err = m2_rfl(s_ctl[1]) | err;
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            const int newW = m2_renumber<NW, MASK>(A, G, J, part, nca, ncb, pb, &s_ctl[2]);
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            if (wave == 0) { cy_rows += t1 - t0; cy_chain += t2 - t1; cy_walk += t3 - t2; cy_renum += t4 - t3; }
            if (newW < 0 || err) { over = true; break; }
            if (!TWO || n + round < 64) { if (lane == n + round) { nmask = J.maskA | J.maskB; ncols = newW; } }
            else if (lane == n + round - 64) { nmask2 = J.maskA | J.maskB; ncols2 = newW; }
            width = newW;
        }
        if (threadIdx.x == 0) {
            if (over) A.ovf[g] = err || n > NMAX ? 2 : 1;
            else A.width[g] = width;
        }
    }
    // the wavefronts' counters
    st_capped = static_cast<unsigned>(m2_incl_sum(static_cast<int>(st_capped)));
    st_filtered = static_cast<unsigned>(m2_incl_sum(static_cast<int>(st_filtered)));
    st_rowsf = static_cast<unsigned>(m2_incl_sum(static_cast<int>(st_rowsf)));
    if (lane == 63) {
        atomicAdd(&A.counters[M2C_ROWS_CAPPED], static_cast<unsigned long long>(st_capped));
        atomicAdd(&A.counters[M2C_ENT_FILTERED], static_cast<unsigned long long>(st_filtered));
        atomicAdd(&A.counters[M2C_ROWS_FILTERED], static_cast<unsigned long long>(st_rowsf));
        atomicAdd(&A.counters[M2C_GATHERS], st_gath);
        if (wave == 0) {
            atomicAdd(&A.counters[M2C_ROWS], st_rows);
            atomicAdd(&A.counters[M2C_ENT_KEPT], st_kept);
            atomicAdd(&A.counters[M2C_JOINS], st_joins);
            atomicAdd(&A.counters[M2C_JOINS_HBMQ], st_hbmq);
            atomicAdd(&A.counters[M2C_CYC_ROWS], cy_rows);
            atomicAdd(&A.counters[M2C_CYC_CHAIN], cy_chain);
            atomicAdd(&A.counters[M2C_CYC_WALK], cy_walk);
            atomicAdd(&A.counters[M2C_CYC_RENUMBER], cy_renum);
            const unsigned long long tx = __builtin_amdgcn_s_memrealtime();
            atomicMin(&A.counters[M2C_T_FIRST_EXIT], tx);
            atomicMax(&A.counters[M2C_T_LAST_EXIT], tx);
            atomicMax(&A.counters[NW == 1 ? M2C_T_EXIT1 : (NMAX > M2_N32 ? M2C_T_EXIT4 : M2C_T_EXIT8)], tx);   // (EXIT4: the class of 33 .. 64 reads)
        }
    }
}

// one block per (member, chunk of columns): the gapped row of a member
__global__ void k_m2_write(M2Args A, const int* member_group, int nmembers, const long long* out_off) {
    const int m = blockIdx.y;
    if (m >= nmembers) return;
    const int g = member_group[m];
    const M2Group G = A.groups[g];
    const M2Member Me = A.members[m];
    const int a = m - G.first_member;
    const int W = A.width[g];
    if (W <= 0 && G.n != 1) return;   // (a group sent to the next pass)
    const uint8_t* src = A.seq + Me.seq_off;
    if (A.out16) {   // vote codes: the cell's quality is the read position's
        uint16_t* dst16 = A.out16 + out_off[g] + static_cast<long long>(a) * W;
        const uint8_t* ql = A.qual + Me.seq_off;
        const uint16_t gapcode = static_cast<uint16_t>(CODE_GAPBIT | code_zero_index(A.navail));
        bool badq = false;
        if (G.n == 1) {
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) dst16[c] = vote_code(src[c], ql[c], A.qoffset, A.navail, badq);
        } else {
            const uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
            for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) {
                const unsigned p = row[c];
                dst16[c] = p == M2_NONE ? gapcode : vote_code("ACGTN"[dna5_code(src[p])], ql[p], A.qoffset, A.navail, badq);
            }
        }
        if (badq) atomicMin(A.bad, m);
        return;
    }
    uint8_t* dst = A.out + out_off[g] + static_cast<long long>(a) * W;
    if (G.n == 1) {   // verbatim (src/quick_msa.cpp:46-50)
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) dst[c] = src[c];
        return;
    }
    const uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) {
        const unsigned p = row[c];
        dst[c] = p == M2_NONE ? '-' : "ACGTN"[dna5_code(src[p])];
    }
}


// =============================================================================================
// host side

static int msa_spec() { return option(OPT_MSA_SPEC) == 1 ? 1 : 2; }

static inline unsigned m2_blocks(long long n, int bs) { return static_cast<unsigned>((n + bs - 1) / bs); }
static inline double m2_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// host seconds of the MSA stage outside its kernels, by part (sarlacc_stage_count "msa_host_<part>_s")
static void m2_host_time(const char* part, double t0) {
    std::map<std::string, double>& cn = ctx().counts;
    const std::string k = std::string("msa_host_") + part + "_s";
    cn[k] = (cn.count(k) ? cn[k] : 0.0) + (m2_now() - t0);
}

// One batch of groups (`ids`: indices into the caller's group list) through the v2 kernels.
// exact_w = false: fast profile capacity (3 maxlen + 64 columns); groups that outgrow it come back flagged.
// exact_w = true: profiles as wide as the sum of the read lengths, nothing can overflow.
// Leaves the batch's state (pos, members ...) in the "m2*" workspaces for the row writer.
struct M2Batch {
    std::vector<int64_t> ids;      // caller's group indices, by decreasing group size
    std::vector<size_t> slot;      // position of each in the caller's order (msa2_core)
    std::vector<M2Group> groups;
    std::vector<M2Member> members;
    std::vector<int> member_group;
    size_t njobs = 0;             // pairwise jobs of the batch (the table itself exists on the device only, k_m2_jobs)
    std::vector<int32_t> width;   // per group of the batch
    std::vector<int> ovf;
    M2Args a{};                   // device pointers
    int* d_member_group = nullptr;
    int max_len = 0, max_wcap = 0, max_n = 0;
    long long ext_n = 0;          // records per plane of the extended library
    bool alias_tiles = false;     // plane 0 of the library in the pairwise kernel's tile buffer (a call of one batch)
    MsaJobSummary jsum;           // band classes and cell count of `jobs`
    hipEvent_t pair_done = nullptr;   // the all-pairs alignments of the batch have finished (m2_prepare -> m2_merge)
};

// The pairwise job table of a batch (4.4 million jobs, 180 MB at C4) is a function of the group and member tables: it is
// written on the device (the host would fill it in 30 ms and the copy would hold the stream for 40 more; the host only counts
// the band classes, MsaJobSummary).  One wavefront per group; job of (a, b), a < b, at M2Group::first_job + a n - a (a + 1) / 2 + b - a - 1,
// rows = member b, columns = member a.
__global__ __launch_bounds__(256) void k_m2_jobs(const M2Group* __restrict__ groups, const M2Member* __restrict__ members, int ngroups,
                                                 MsaJob* __restrict__ jobs) {
    const int q = static_cast<int>((static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= ngroups) return;
    const M2Group G = groups[q];
    const M2Member* const M = members + G.first_member;
    const int n = G.n;
    MsaJob* row = jobs + G.first_job;
    for (int a = 0; a + 1 < n; ++a) {
        const M2Member Ma = M[a];
        for (int b = a + 1 + lane; b < n; b += 64) {
            const M2Member Mb = M[b];
            MsaJob J;
            J.read_off = Mb.seq_off; J.ctr_off = Ma.seq_off;
            J.lr = Mb.len; J.lc = Ma.len;
            J.out_off = Ma.map_base + static_cast<long long>(b - 1) * Ma.len;    // b among the others of a
            J.out2_off = Mb.map_base + static_cast<long long>(a) * Mb.len;       // a among the others of b
            row[b - a - 1] = J;
        }
        row += n - a - 1;
    }
}

// Host tables of a batch: groups and members, and the band classes of the pairwise jobs (4.4 million at C4).  The offsets come
// from one serial pass over the groups; the members are then filled and the jobs counted by a few threads over disjoint ranges
// of groups.
static int m2_plan(M2Batch& B, const int64_t* grp_off, const int32_t* grp, const int64_t* rel, const long long* gsum, const int* gmx,
                   bool exact_w, int bandwidth) {
    const size_t ngr = B.ids.size();
    B.groups.assign(ngr, M2Group{});
    B.max_len = 0; B.max_wcap = 0; B.max_n = 0;
    long long map_pos = 0, col_pos = 0, pos_pos = 0, dist_pos = 0, ext_pos = 0, mem_pos = 0, job_pos = 0;
    for (size_t q = 0; q < ngr; ++q) {
        const int64_t g = B.ids[q];
        const int n = static_cast<int>(grp_off[g + 1] - grp_off[g]);
        M2Group& G = B.groups[q];
        G.first_member = static_cast<int>(mem_pos);
        G.n = n;
        const long long sum = gsum[B.slot[q]];   // sum and maximum of the group's read lengths (msa2_core)
        const int mx = gmx[B.slot[q]];
        const long long fast_w = m2_fast_width(n, mx);
        // (65535 columns is the ceiling of spec v2: positions and columns are 16-bit; only reachable when the sum of
        // the read lengths exceeds it AND the alignment really is that wide)
        // (msa2_max_columns: a lower ceiling, for the tests of the hand-over to spec v1)
        const long long ceiling = option(OPT_MSA2_MAX_COLUMNS) > 0 ? std::min(65535, option(OPT_MSA2_MAX_COLUMNS)) : 65535;
        G.wcap = static_cast<int>(std::min<long long>(ceiling, std::min<long long>(sum, exact_w ? sum : fast_w)));
        if (G.wcap < 1) G.wcap = 1;
        G.pos_base = pos_pos;
        G.first_job = job_pos;
        G.dist_base = dist_pos;
        G.ext_base = ext_pos;
        G.lmax = mx;
        G.map0 = map_pos;
        G.col0 = col_pos;
        pos_pos += static_cast<long long>(n) * G.wcap;
        dist_pos += static_cast<long long>(n) * n + n;
        ext_pos += static_cast<long long>(n) * (n - 1) / 2 * mx;
        map_pos += static_cast<long long>(std::max(0, n - 1)) * sum;
        col_pos += sum;
        mem_pos += n;
        job_pos += static_cast<long long>(n) * (n - 1) / 2;
        B.max_len = std::max(B.max_len, mx);
        B.max_wcap = std::max(B.max_wcap, G.wcap);
        B.max_n = std::max(B.max_n, n);
    }
    B.ext_n = ext_pos;
    B.members.assign(static_cast<size_t>(mem_pos), M2Member{});
    B.member_group.assign(static_cast<size_t>(mem_pos), 0);
    B.njobs = static_cast<size_t>(job_pos);
    auto fill = [&](size_t q0, size_t q1, MsaJobSummary* sum) {
        for (size_t q = q0; q < q1; ++q) {
            const M2Group& G = B.groups[q];
            const int32_t* mem = grp + grp_off[B.ids[q]];
            const int n = G.n;
            M2Member* const M = B.members.data() + G.first_member;
            long long mp = G.map0, cp = G.col0;
            for (int a = 0; a < n; ++a) {
                M2Member& Me = M[a];
                Me.seq_off = rel[mem[a] - 1];
                Me.len = static_cast<int>(rel[mem[a]] - rel[mem[a] - 1]);
                Me.map_base = mp;
                Me.col_base = cp;
                Me.ext_base = G.ext_base; Me.first_member = G.first_member; Me.n = n; Me.lmax = G.lmax;
                mp += static_cast<long long>(std::max(0, n - 1)) * Me.len;
                cp += Me.len;
                B.member_group[static_cast<size_t>(G.first_member) + a] = static_cast<int>(q);
            }
            // band classes of the group's pairs (rows = b, columns = a: k_m2_jobs).  Where the longest and the shortest read of the
            // group already fit the narrow class -- nearly every group -- no pair needs a look of its own: the summary's sums
            // and maxima in closed form, the cell count from a branch-free loop.
            if (n < 2) continue;
            int lmin = M[0].len, lmax = M[0].len;
            for (int a = 1; a < n; ++a) { lmin = std::min(lmin, M[a].len); lmax = std::max(lmax, M[a].len); }
            const long long widest = static_cast<long long>(lmax - lmin) + 2LL * bandwidth + 1;
            if (widest <= 256) {
                sum->wide_listed = true;
                sum->n[0] += static_cast<size_t>(n) * (n - 1) / 2;
                sum->band[0] = std::max(sum->band[0], static_cast<int>(widest));
                const long long base = 2LL * bandwidth + 1;
                double cells = 0;
                for (int a = 0; a < n; ++a) {
                    const int la = M[a].len;
                    if (a + 1 < n) { sum->cols[0] += static_cast<double>(la) * (n - 1 - a); sum->lc[0] = std::max(sum->lc[0], la); }
                    if (a > 0) sum->lr[0] = std::max(sum->lr[0], la);
                    long long acc = 0;
                    for (int b = a + 1; b < n; ++b) { const long long lb = M[b].len; acc += lb * (base + (lb > la ? lb - la : la - lb)); }
                    cells += static_cast<double>(acc);
                }
                sum->cells += cells;
                continue;
            }
            long long j = G.first_job;
            for (int a = 0; a < n; ++a)
                for (int b = a + 1; b < n; ++b, ++j) sum->add(bandwidth, M[b].len, M[a].len, j);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nthreads = job_pos < 200000 ? 1 : std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, 8));
    B.jsum = MsaJobSummary{};
    if (nthreads == 1) fill(0, ngr, &B.jsum);
    else {
        // ranges of groups with about the same number of jobs each
        std::vector<std::thread> pool;
        std::vector<MsaJobSummary> part(nthreads);
        size_t q0 = 0;
        for (size_t t = 0; t < nthreads; ++t) {
            const long long want = job_pos * static_cast<long long>(t + 1) / static_cast<long long>(nthreads);
            size_t q1 = q0;
            while (q1 < ngr && (t + 1 == nthreads || B.groups[q1].first_job + static_cast<long long>(B.groups[q1].n) * (B.groups[q1].n - 1) / 2 <= want)) ++q1;
            pool.emplace_back(fill, q0, q1, &part[t]);
            q0 = q1;
        }
        for (std::thread& th : pool) th.join();
        for (const MsaJobSummary& ps : part) B.jsum.merge(ps);
    }
    return 0;
}

// streams of the side-by-side instantiations of k_m2_group (created once per process and device, non-blocking: the
// caller's stream may be the legacy default one)
struct M2Streams {
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> join;
    hipEvent_t fork = nullptr;
    hipEvent_t pair[2] = {nullptr, nullptr};   // "alignments of the batch in workspace set k are done"
    int device = -1;
    int ensure(int n) {
        if (device != ctx().device) {   // (streams and events belong to the device that was current when they were made)
            for (hipStream_t x : st) (void)hipStreamDestroy(x);
            for (hipEvent_t e : join) (void)hipEventDestroy(e);
            if (fork) (void)hipEventDestroy(fork);
            for (int k = 0; k < 2; ++k) { if (pair[k]) (void)hipEventDestroy(pair[k]); pair[k] = nullptr; }
            st.clear(); join.clear(); fork = nullptr;
            device = ctx().device;
        }
        if (!fork) SL_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k)
            if (!pair[k]) SL_HIP(hipEventCreateWithFlags(&pair[k], hipEventDisableTiming));
        while (static_cast<int>(st.size()) < n) {
            hipStream_t x; hipEvent_t e;
            SL_HIP(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
            SL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            st.push_back(x); join.push_back(e);
        }
        return 0;
    }
};
static M2Streams& m2_streams() { static M2Streams m; return m; }

// First half of a batch: tables to the device, workspaces, the all-pairs alignments -- on stream `sp`, which runs ahead of
// the merging: while the wavefronts of k_m2_group wait for memory (most of their time) the alignments of the NEXT batch
// keep the vector units busy.  `B.pair_done` fires when the batch's library is complete.
static int m2_prepare(M2Batch& B, const std::string& pf, const uint8_t* d_seq, double match, double mismatch, double gap_extension,
                      double gap_opening, int bandwidth, double* cells, bool first_of_call, hipStream_t s) {
    Context& c = ctx();
    const size_t ng = B.groups.size(), nm = B.members.size();
    if (ng == 0) return 0;
    long long map_n = 0, col_n = 0, pos_n = 0;
    for (const M2Member& Me : B.members) col_n += Me.len;
    for (const M2Group& G : B.groups) pos_n += static_cast<long long>(G.n) * G.wcap;
    if (!B.members.empty()) { const M2Member& L = B.members.back(); const M2Group& G = B.groups[B.member_group.back()]; map_n = L.map_base + static_cast<long long>(std::max(0, G.n - 1)) * L.len; }
    M2Args& a = B.a;
    a = M2Args{};
    M2Group* d_groups; M2Member* d_members; MsaJob* d_jobs; int* d_mg;
    double th = m2_now();
    SL_TRY(upload((pf + ".groups").c_str(), B.groups.data(), ng, &d_groups, s));
    SL_TRY(upload((pf + ".members").c_str(), B.members.data(), nm, &d_members, s));
    SL_TRY(upload((pf + ".mg").c_str(), B.member_group.data(), nm, &d_mg, s));
    SL_TRY(scratch((pf + ".jobs").c_str(), B.njobs + 1, &d_jobs));
    hipLaunchKernelGGL(k_m2_jobs, dim3(static_cast<unsigned>((ng + 3) / 4)), dim3(256), 0, s, d_groups, d_members, static_cast<int>(ng), d_jobs);
    SL_HIP(hipGetLastError());
    B.d_member_group = d_mg;
    uint16_t* d_map; int2* d_stats; double* d_dist; int2* d_joins; m2_mask* d_first; uint16_t* d_col; uint16_t* d_pos; int* d_ovf; int32_t* d_width;
    SL_TRY(scratch((pf + ".map").c_str(), static_cast<size_t>(map_n) + 1, &d_map));
    SL_TRY(scratch((pf + ".stats").c_str(), B.njobs + 1, &d_stats));
    d_dist = nullptr;   // (the tree kernel keeps its distance matrix in LDS since round 4)
    SL_TRY(scratch((pf + ".joins").c_str(), nm + 1, &d_joins));
    SL_TRY(scratch((pf + ".first").c_str(), nm + 1, &d_first));
    SL_TRY(scratch((pf + ".col").c_str(), static_cast<size_t>(col_n) + 1, &d_col));
    SL_TRY(scratch((pf + ".pos").c_str(), static_cast<size_t>(pos_n) + 1, &d_pos));
    SL_TRY(scratch((pf + ".ovf").c_str(), ng, &d_ovf));
    SL_TRY(scratch((pf + ".width").c_str(), ng, &d_width));
    SL_HIP(hipMemsetAsync(d_ovf, 0, sizeof(int) * ng, s));
    a.seq = d_seq; a.groups = d_groups; a.members = d_members; a.ngroups = static_cast<int>(ng);
    a.ma = static_cast<int>(match); a.mm = static_cast<int>(mismatch);
    a.map = d_map; a.stats = d_stats; a.dist = d_dist; a.joins = d_joins; a.first = d_first;
    a.col = d_col; a.pos = d_pos; a.ovf = d_ovf; a.width = d_width;

    m2_host_time("upload_alloc", th);
    if (B.alias_tiles && B.ext_n > 0) {   // (before the alignments take their pointer to it)
        void* tb;
        SL_TRY(c.buffer("msa.tb0", static_cast<size_t>(B.ext_n) * 4 + 16, &tb));
    }
    // ---- all pairs ----
    th = m2_now();
    *cells += B.jsum.cells;
    SL_TRY(c.stage_begin("msa_pairwise", s));
    SL_TRY(msa_pairwise_launch(nullptr, B.njobs, d_jobs, d_seq, match, mismatch, gap_extension, gap_opening, bandwidth, 1, nullptr, nullptr,
                               d_map, d_stats, s, &B.jsum, first_of_call));
    SL_TRY(c.stage_end("msa_pairwise", s));
    m2_host_time("pairwise_launch", th);
    SL_HIP(hipEventRecord(B.pair_done, s));
    return 0;
}

// Second half: guide trees, the extended library, all the merging, widths back to the host -- on stream `s` (and the streams of
// the side-by-side instantiations) once the batch's alignments are done.
static int m2_merge(M2Batch& B, const std::string& pf, double* counters, hipStream_t s) {
    Context& c = ctx();
    const size_t ng = B.groups.size(), nm = B.members.size();
    if (ng == 0) return 0;
    M2Args& a = B.a;
    int* const d_mg = B.d_member_group;
    int32_t* const d_width = a.width;
    int* const d_ovf = a.ovf;
    SL_HIP(hipStreamWaitEvent(s, B.pair_done, 0));
    // ---- guide trees, leaves ----
    SL_TRY(c.stage_begin("msa_merge", s));
    hipLaunchKernelGGL(k_m2_tree, dim3(static_cast<unsigned>(ng)), dim3(64), 0, s, a);
    if (nm) hipLaunchKernelGGL(k_m2_init, dim3(std::min(2u, std::max(1u, m2_blocks(B.max_len, 256))), static_cast<unsigned>(nm)), dim3(256), 0, s, a, d_mg, static_cast<int>(nm));
    SL_HIP(hipGetLastError());
    const bool unitw = a.ma <= 1 && a.mm <= 1 && !option(OPT_MSA2_GENERAL_ROWS);
    unsigned long long* d_cnt;
    SL_TRY(scratch((pf + ".cnt").c_str(), M2C_N, &d_cnt));
    {
        unsigned long long init[M2C_N] = {};
        init[M2C_T_START] = init[M2C_T_FIRST_EXIT] = ~0ull;
        SL_HIP(hipMemcpyAsync(d_cnt, init, sizeof init, hipMemcpyHostToDevice, s));
        SL_HIP(hipStreamSynchronize(s));   // (init is on this frame's stack)
    }
    a.counters = d_cnt;
    // ---- the extended library of every group (spec v2, step 5) ----
    if (B.ext_n > 0) {
        const int planes = m2_ext_planes(unitw);
        a.ext_plane = B.ext_n;
        // A call that runs as ONE batch keeps plane 0 in the tile buffer of the pairwise kernel, whose records are dead by now
        // (m2_prepare made sure it is large enough); with pipelined batches the next batch's alignments are writing there.
        uint32_t* hi;
        SL_TRY(scratch((pf + ".ext").c_str(), static_cast<size_t>(B.ext_n) * (planes - (B.alias_tiles ? 1 : 0)) + 1, &hi));
        if (B.alias_tiles) {
            const auto it = c.ws.find("msa.tb0");
            if (it == c.ws.end() || it->second.cap < static_cast<size_t>(B.ext_n) * 4) return fail("sarlacc_amd: internal error: the tile buffer does not hold the extended library");
            a.ext = static_cast<uint32_t*>(it->second.ptr);
            a.ext_off1 = hi - a.ext;   // (both 4-byte aligned device addresses)
        } else {
            a.ext = hi;
            a.ext_off1 = B.ext_n;
        }
        hipLaunchKernelGGL(k_m2_first, dim3(static_cast<unsigned>(ng)), dim3(64), 0, s, a);
        SL_HIP(hipGetLastError());
    }
    // One launch of the extension per size class (the batch is ordered by decreasing size): groups of up to 12 reads through the
    // four-positions kernel (unit weights; 6 KB of LDS per wavefront), larger ones through the one-position kernel (256 positions
    // of 64 reads would be 32 KB per wavefront, five wavefronts per CU; msa2_wide_extend moves the limit, msa2_simple_extend sends
    // everything through the one-position kernel).  The class of more than 24 reads goes first: its groups' merging -- the longest
    // launches of the stage, a few workgroups waiting for memory -- then runs UNDER the extension of the other classes.
    auto ext_class = [&](int cap) -> int {
        if (B.ext_n <= 0) return 0;
        const int lower = cap == M2_MAXN ? 24 : (cap == 24 ? 12 : 0);   // this launch: groups of lower < n <= cap
        size_t q = 0, q1;
        while (q < ng && B.groups[q].n > cap) ++q;
        q1 = q;
        while (q1 < ng && B.groups[q1].n > lower && B.groups[q1].n >= 2) ++q1;
        if (q1 == q) return 0;
        const int m0 = B.groups[q].first_member, m1 = B.groups[q1 - 1].first_member + B.groups[q1 - 1].n;
        const int wide_max = !unitw || option(OPT_MSA2_SIMPLE_EXTEND) ? 0 : (option(OPT_MSA2_WIDE_EXTEND) > 0 ? option(OPT_MSA2_WIDE_EXTEND) : 12);
        const int cn = std::min(cap, std::max(B.max_n, 1));
        if (cap > wide_max) {
            const unsigned gx1 = std::max(1u, std::min(8u, m2_blocks(B.max_len, 64 * M2_EXT_WAVES)));
            const size_t lds = static_cast<size_t>(M2_EXT_WAVES) * cn * 128 + static_cast<size_t>(cn) * sizeof(M2XMember);
            if (unitw) hipLaunchKernelGGL((k_m2_extend<true>), dim3(gx1, static_cast<unsigned>(m1 - m0)), dim3(64 * M2_EXT_WAVES), lds, s, a, d_mg, m0, m1, cn);
            else hipLaunchKernelGGL((k_m2_extend<false>), dim3(gx1, static_cast<unsigned>(m1 - m0)), dim3(64 * M2_EXT_WAVES), lds, s, a, d_mg, m0, m1, cn);
        } else {
            const unsigned gx = std::max(1u, std::min(8u, m2_blocks(B.max_len, 256)));
            hipLaunchKernelGGL(k_m2_extend_unit, dim3(gx, static_cast<unsigned>(m1 - m0)), dim3(64), static_cast<size_t>(cn) * 512, s, a, m0, m1);
        }
        SL_HIP(hipGetLastError());
        return 0;
    };
    SL_TRY(ext_class(M2_MAXN));
    // ---- progressive merging: every join of every group in ONE round of launches ----
    // Groups are ordered by decreasing size.  A group is merged by one workgroup: one wavefront for the bulk (up to M2_NB reads),
    // 8 wavefronts up to 32 reads -- the cost of a group grows with the cube of its size, and the longest group sets the length of
    // the launch --, M2_NWD wavefronts for groups of 33 to M2_MAXN reads, which have an instantiation of their own (64-bit member
    // masks, up to 127 tree nodes).  The instantiations run side by side on streams of their own.
    size_t nmulti = 0;
    while (nmulti < ng && B.groups[nmulti].n >= 2) ++nmulti;
    if (nmulti) {
        size_t iD = 0, iC = 0;
        while (iD < nmulti && B.groups[iD].n > M2_N32) ++iD;
        iC = iD;
        if (!option(OPT_MSA2_SINGLE_WAVE))
            while (iC < nmulti && B.groups[iC].n > M2_NB) ++iC;
        int* d_next;
        SL_TRY(scratch((pf + ".next").c_str(), 4, &d_next));
        SL_HIP(hipMemsetAsync(d_next, 0, 4 * sizeof(int), s));
        a.chain_hbm = option(OPT_MSA2_CHAIN_HBM) ? 1 : 0;

        M2Streams& MS = m2_streams();
        SL_TRY(MS.ensure(4));
        SL_HIP(hipEventRecord(MS.fork, s));
        struct Cls { size_t lo, hi; int nw; const char* tag; int stream; };   // stream: index into MS.st (2 is the alignments' own), -1 = s
        const Cls cls[3] = {{0, iD, M2_NWD, ".d", 3}, {iD, iC, 8, ".c", 0}, {iC, nmulti, 1, ".a", -1}};
        for (int k = 0; k < 3; ++k) {
            if (k == 2) {   // (the side streams have their launches; the caller's stream goes on with the library of the smaller groups)
                SL_TRY(ext_class(24));
                SL_TRY(ext_class(12));
            }
            if (cls[k].lo >= cls[k].hi) continue;
            // scratch of a resident workgroup: as wide as the widest profile capacity of the class
            int class_wcap = 1;
            for (size_t q = cls[k].lo; q < cls[k].hi; ++q) class_wcap = std::max(class_wcap, B.groups[q].wcap);
            const long long w_rows = (static_cast<long long>(class_wcap) + 63) / 64 * 64 + 64 * 9;   // (+ the slack of 8 row ranges)
            const long long per_wg = w_rows * (M2_CAP * 12 + 4 * 4 + 8);
            hipStream_t sk = cls[k].stream >= 0 ? MS.st[cls[k].stream] : s;
            if (cls[k].stream >= 0) SL_HIP(hipStreamWaitEvent(sk, MS.fork, 0));
            const void* fn = unitw ? (k == 0 ? reinterpret_cast<const void*>(&k_m2_group<true, M2_NWD, M2_MAXN>)
                                      : k == 1 ? reinterpret_cast<const void*>(&k_m2_group<true, 8, M2_N32>)
                                               : reinterpret_cast<const void*>(&k_m2_group<true, 1, M2_N32>))
                                   : (k == 0 ? reinterpret_cast<const void*>(&k_m2_group<false, M2_NWD, M2_MAXN>)
                                      : k == 1 ? reinterpret_cast<const void*>(&k_m2_group<false, 8, M2_N32>)
                                               : reinterpret_cast<const void*>(&k_m2_group<false, 1, M2_N32>));
            int per_cu = 0;
            SL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * cls[k].nw, 0));
            per_cu = std::max(1, std::min(per_cu, 32 / cls[k].nw));
            if (option(OPT_MSA2_WAVES_PER_CU) > 0) per_cu = std::max(1, std::min(per_cu, option(OPT_MSA2_WAVES_PER_CU) / cls[k].nw));
            long long wgs = std::min<long long>(static_cast<long long>(per_cu) * std::max(1, c.num_cu), static_cast<long long>(cls[k].hi - cls[k].lo));
            // (scratch of the resident workgroups: at most 16 GB per instantiation -- 8 GB left the one-wavefront instantiation with
            // 6 000 of its 8 192 workgroups at 2-kb reads and cost 18 % of the merge stage)
            wgs = std::max<long long>(1, std::min(wgs, (16LL << 30) / per_wg));
            M2Args am = a;
            am.w_rows = w_rows;
            const std::string q = std::string("m2w") + cls[k].tag;   // (shared by the batches: their merging runs one after the other)
            SL_TRY(scratch((q + ".w_ent").c_str(), static_cast<size_t>(wgs * w_rows * M2_CAP), &am.w_ent));
            SL_TRY(scratch((q + ".w_pred").c_str(), static_cast<size_t>(wgs * w_rows * M2_CAP), &am.w_pred));
            SL_TRY(scratch((q + ".w_part").c_str(), static_cast<size_t>(wgs * w_rows), &am.w_part));
            SL_TRY(scratch((q + ".w_nca").c_str(), static_cast<size_t>(wgs * w_rows), &am.w_nca));
            SL_TRY(scratch((q + ".w_ncb").c_str(), static_cast<size_t>(wgs * w_rows), &am.w_ncb));
            SL_TRY(scratch((q + ".w_pb").c_str(), static_cast<size_t>(wgs * w_rows), &am.w_pb));
            SL_TRY(scratch((q + ".w_q").c_str(), static_cast<size_t>(wgs * w_rows), &am.w_q));
            am.g0 = static_cast<int>(cls[k].lo); am.g1 = static_cast<int>(cls[k].hi);
            am.next = d_next + k;
            const dim3 grid(static_cast<unsigned>(wgs)), block(64 * cls[k].nw);
            if (unitw) {
                if (k == 0) hipLaunchKernelGGL((k_m2_group<true, M2_NWD, M2_MAXN>), grid, block, 0, sk, am);
                else if (k == 1) hipLaunchKernelGGL((k_m2_group<true, 8, M2_N32>), grid, block, 0, sk, am);
                else hipLaunchKernelGGL((k_m2_group<true, 1, M2_N32>), grid, block, 0, sk, am);
            } else {
                if (k == 0) hipLaunchKernelGGL((k_m2_group<false, M2_NWD, M2_MAXN>), grid, block, 0, sk, am);
                else if (k == 1) hipLaunchKernelGGL((k_m2_group<false, 8, M2_N32>), grid, block, 0, sk, am);
                else hipLaunchKernelGGL((k_m2_group<false, 1, M2_N32>), grid, block, 0, sk, am);
            }
            SL_HIP(hipGetLastError());
            if (cls[k].stream >= 0) SL_HIP(hipEventRecord(MS.join[cls[k].stream], sk));
        }
        // the caller's stream joins the side streams only now, AFTER its own instantiation is queued (waiting inside the loop
        // made the one-wavefront launch -- last in the loop, on `s` -- start when the others had finished: the classes ran one
        // after the other, 0.18 s of the 0.58 s merge stage of bench.py's pipeline workload)
        for (int k = 0; k < 3; ++k)
            if (cls[k].lo < cls[k].hi && cls[k].stream >= 0) SL_HIP(hipStreamWaitEvent(s, MS.join[cls[k].stream], 0));
    }
    SL_TRY(c.stage_end("msa_merge", s));
    B.width.resize(ng);
    B.ovf.resize(ng);
    unsigned long long hc[M2C_N] = {};
    SL_HIP(hipMemcpyAsync(B.width.data(), d_width, sizeof(int32_t) * ng, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(B.ovf.data(), d_ovf, sizeof(int) * ng, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(hc, a.counters, sizeof hc, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (nmulti) {
        for (int k = 0; k < M2C_T_START; ++k) counters[k] += static_cast<double>(hc[k]);
        // seconds from the first wavefront's start to the first / the last wavefront's exit (100 MHz counter)
        counters[M2C_T_FIRST_EXIT] += static_cast<double>(hc[M2C_T_FIRST_EXIT] - hc[M2C_T_START]) * 1e-8;
        counters[M2C_T_LAST_EXIT] += static_cast<double>(hc[M2C_T_LAST_EXIT] - hc[M2C_T_START]) * 1e-8;
        for (int k = M2C_T_EXIT1; k <= M2C_T_EXIT8; ++k)
            if (hc[k]) counters[k] += static_cast<double>(hc[k] - hc[M2C_T_START]) * 1e-8;
        counters[M2C_T_START] += static_cast<double>(hc[M2C_JOINS] ? 1 : 0);   // (launches)
    }
    for (size_t q = 0; q < ng; ++q)
        if (B.ovf[q] > 1) return fail("sarlacc_amd: internal error: the chain search of an MSA join did not finish");
    return 0;
}

// rows of the batch's groups (except those sent to the next pass, whose width is 0) at out + off[group of the caller's list]
static int m2_write_batch(M2Batch& B, const std::string& pf, const std::vector<long long>& off_of_batch_group, void* d_out,
                          const CodeSpec& code, hipStream_t s) {
    if (B.members.empty()) return 0;
    long long* d_off;
    SL_TRY(upload((pf + ".ooff").c_str(), off_of_batch_group.data(), off_of_batch_group.size(), &d_off, s));
    if (code.want) {
        B.a.out = nullptr; B.a.out16 = static_cast<uint16_t*>(d_out);
        B.a.qual = *code.qual; B.a.qoffset = code.qoffset; B.a.navail = code.navail; B.a.bad = code.d_bad;
    } else {
        B.a.out = static_cast<uint8_t*>(d_out); B.a.out16 = nullptr;
    }
    int maxw = 1;
    for (int32_t w : B.width) maxw = std::max(maxw, static_cast<int>(w));
    // two workgroups per row at most (the kernel strides over the row): with one workgroup per 256 columns of the WIDEST group, most
    // of the ~20 million workgroups of a 10^6-read batch found nothing to do and the launch rate, not the bytes, set the time
    hipLaunchKernelGGL(k_m2_write, dim3(std::min(2u, std::max(1u, m2_blocks(maxw, 256))), static_cast<unsigned>(B.members.size())), dim3(256), 0, s, B.a,
                       B.d_member_group, static_cast<int>(B.members.size()), d_off);
    SL_HIP(hipGetLastError());
    return 0;
}

// spec v2 on the groups `ids` of the caller's list; rows land in *d_rows at off[k] for ids[k] (processing order:
// the offsets are not cumulative in k), widths in width[k].
static int msa2_core(const int64_t* grp_off, const int32_t* grp, const std::vector<int64_t>& ids, const uint8_t* d_seq,
                     const std::vector<int64_t>& rel, double match, double mismatch, double gap_extension, double gap_opening,
                     int bandwidth, std::vector<int32_t>& width, std::vector<long long>& off, uint8_t** d_rows,
                     const std::function<int()>* overlap, const CodeSpec& code, hipStream_t s, std::vector<size_t>* gave_up) {
    Context& c = ctx();
    const size_t cs = code.want ? 2 : 1;   // bytes per cell of the row buffer (offsets and widths stay in cells)
    bool waited = false;
    width.assign(ids.size(), 0);
    off.assign(ids.size() + 1, 0);
    *d_rows = nullptr;
    // row buffer: grown when a batch does not fit (contents are kept)
    Workspace& rows_ws = c.ws["msa2.rows"];
    auto rows_reserve = [&](size_t used, size_t need) -> int {
        if (rows_ws.cap >= need) return 0;
        const size_t want = std::max(need, rows_ws.cap + rows_ws.cap / 2) + 4096;
        void* np = nullptr;
        if (hipMalloc(&np, want) != hipSuccess) {
            size_t fb = 0, tb = 0;
            (void)hipGetLastError();
            (void)hipMemGetInfo(&fb, &tb);
            size_t held = 0;
            for (const auto& kv : c.ws) held += kv.second.cap;
            return fail("sarlacc_amd: cannot allocate %zu bytes of device memory for the alignment rows (%zu free of %zu, the library's workspaces hold %zu)", want, fb, tb, held);
        }
        if (rows_ws.ptr) {
            if (used) SL_HIP(hipMemcpyAsync(np, rows_ws.ptr, used, hipMemcpyDeviceToDevice, s));
            SL_HIP(hipStreamSynchronize(s));
            SL_HIP(hipFree(rows_ws.ptr));
        }
        rows_ws.ptr = np;
        rows_ws.cap = want;
        return 0;
    };
    // Batches.  What a batch holds per group is the library (both maps of every pair: 4 (n - 1) bytes per base), the positions /
    // columns of its profiles and the extended library (12 or 16 bytes per pair and position) -- about 165 GB for C4 (10^5 groups x 10 reads
    // x 2 kb), sized for an HBM of 288 GB and bounded by a share of what is free now.  A large call is cut into a few
    // batches that are PIPELINED: the all-pairs alignments of batch k + 1 (vector-unit bound) run on a stream of their
    // own under the merging of batch k (memory-latency bound); two sets of workspaces alternate.
    size_t free_b = 0, total_b = 0;
    SL_HIP(hipMemGetInfo(&free_b, &total_b));
    // the workspaces an earlier call left are grown in place, not added: everything this stage holds -- the batches' maps, positions,
    // columns and tables, but also the pairwise kernel's traceback records and move strings, the merging's per-workgroup scratch and
    // the rows (counting only the first four made the SECOND of two equal calls see half the room and cut itself into 5 batches
    // where the first took 3)
    long long reusable = 0;
    for (const auto& kv : c.ws)
        if (kv.first.compare(0, 2, "m2") == 0 || kv.first.compare(0, 4, "msa.") == 0 || kv.first.compare(0, 5, "msa2.") == 0)
            reusable += static_cast<long long>(kv.second.cap);
    // (msa2_budget_gb: the cap in GB, tests and sweeps; the default leaves the other half of what is free to the traceback records of
    // the pairwise kernel, the per-workgroup scratch of the merging and the rows)
    const long long budget_cap = option(OPT_MSA2_BUDGET_GB) > 0 ? static_cast<long long>(option(OPT_MSA2_BUDGET_GB)) << 30 : 96LL << 30;
    const long long mem_budget = std::max<long long>(1LL << 30, std::min<long long>(budget_cap, (static_cast<long long>(free_b) + reusable) / 2));
    const long long job_budget = 12000000;
    double cells = 0, pairs = 0;
    double counters[M2C_N] = {};
    const bool unitw_plan = static_cast<int>(match) <= 1 && static_cast<int>(mismatch) <= 1 && !option(OPT_MSA2_GENERAL_ROWS);   // (m2_merge)
    double second_pass = 0;   // groups whose profiles outgrew the first-pass capacity
    double nbatches = 0;      // batches of the call (from two on the stage timers of alignments and merging overlap)
    long long used = 0;
    M2Streams& MS = m2_streams();
    SL_TRY(MS.ensure(3));
    hipStream_t sp = MS.st[2];
    SL_HIP(hipEventRecord(MS.fork, s));          // whatever the caller queued on `s` (the reads) comes first
    SL_HIP(hipStreamWaitEvent(sp, MS.fork, 0));
    bool first = true;
    // Pass 0: every group with the fast profile capacity.  Pass 1: the groups whose profiles outgrew it (several
    // unrelated reads in one cluster), with profiles as wide as the sum of the read lengths.
    std::vector<size_t> todo(ids.size());
    std::iota(todo.begin(), todo.end(), size_t(0));
    // sum and maximum of the read lengths of every group, once, by a few threads: the batch sizes below and every batch's plan
    // start from them (a serial pass over the 10^6 reads of C4 is 5 ms during which the device has nothing to run)
    std::vector<long long> gsum(ids.size());
    std::vector<int> gmx(ids.size());
    {
        auto lens = [&](size_t q0, size_t q1) {
            for (size_t q = q0; q < q1; ++q) {
                const int32_t* mem = grp + grp_off[ids[q]];
                const long long n = grp_off[ids[q] + 1] - grp_off[ids[q]];
                long long sum = 0, mx = 0;
                for (long long a = 0; a < n; ++a) { const long long len = rel[mem[a]] - rel[mem[a] - 1]; sum += len; mx = std::max(mx, len); }
                gsum[q] = sum; gmx[q] = static_cast<int>(mx);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = ids.size() < 20000 ? 1 : std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, 8));
        if (nt == 1) lens(0, ids.size());
        else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; ++t) th.emplace_back(lens, ids.size() * t / nt, ids.size() * (t + 1) / nt);
            for (std::thread& x : th) x.join();
        }
    }
    for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
        const bool exact_w = pass == 1;
        // processing order: by decreasing group size -- the workgroups of k_m2_group take the groups in this order, the
        // longest first
        std::stable_sort(todo.begin(), todo.end(), [&](size_t x, size_t y) {
            return grp_off[ids[x] + 1] - grp_off[ids[x]] > grp_off[ids[y] + 1] - grp_off[ids[y]];
        });
        std::vector<size_t> again;
        long long mem_all = 0, jobs_all = 0, ext_all = 0, cells_est = 0;
        std::vector<long long> cmem(todo.size() + 1, 0), cjobs(todo.size() + 1, 0);   // cumulative over the sorted list
        for (size_t x = 0; x < todo.size(); ++x) {
            const size_t q = todo[x];
            const int64_t g = ids[q];
            const long long n = grp_off[g + 1] - grp_off[g];
            const long long sum = gsum[q], mx = gmx[q];
            const long long wc = exact_w ? sum : std::min(sum, m2_fast_width(n, mx));
            mem_all += 2 * (n - 1) * sum + 2 * n * wc + 2 * sum + n * (n - 1) / 2 * mx * 4 * m2_ext_planes(unitw_plan);
            ext_all += n * (n - 1) / 2 * mx * 4 * m2_ext_planes(unitw_plan);
            cells_est += n * std::min(sum, 2 * mx);   // (rows: about the longest read, some more for clusters of several molecules)
            mem_all += n * (n - 1) / 2 * ((2 * mx) / 16 + 2) * 4;   // the move strings of the bit-vector pairwise kernel (msa_pairwise.hip)
            jobs_all += n * (n - 1) / 2;
            cmem[x + 1] = mem_all; cjobs[x + 1] = jobs_all;
        }
        // batches: as many as memory and the job list demand, a few more for the overlap once there is enough work.  (Until round 5
        // every batch got every nb-th group of the sorted list, "the same mix of sizes" -- and so every batch held some of the
        // largest groups, whose merging is the longest chain of dependent joins of the call: a 64-read cluster keeps its workgroup
        // busy for about 2 s, and the merge stage of a call took nb x that -- the same clusters in 1 / 3 / 5 batches: 2.96 / 6.6 /
        // 9.6 s.  Now the LARGE groups go to the batches in contiguous pieces of the sorted list -- the largest share the first
        // batch, the longest group of every later batch is shorter than that of the one before -- and only the bulk of
        // one-wavefront groups is dealt round robin.)
        // (one batch when everything fits -- then the stage timers do not overlap either; from two on the two workspace
        // sets share the budget.  Measured at C4 with two pipelined batches: pure groups 5 % faster end to end, clusters
        // of several molecules unchanged -- the alignments saturate the vector units by themselves.)
        c.counts["msa2_mem_all_gb"] = static_cast<double>(mem_all) / 1073741824.0;
        c.counts["msa2_mem_budget_gb"] = static_cast<double>(mem_budget) / 1073741824.0;
        long long nb = std::max<long long>(1, std::max((mem_all + mem_budget - 1) / mem_budget, (jobs_all + job_budget - 1) / job_budget));
        // One batch whenever the device holds it: what the stage needs beside a batch's arrays is bounded -- the tile buffer of the
        // alignments (48 GB at most, and it doubles as plane 0 of the extended library), the per-workgroup scratch of the merging
        // (16 GB per class at most), the rows.  (Until the library was stored, a batch could have half of what was free and
        // bench.py's pipeline workload fitted; with it the half-rule cut that call into 4 batches and the merging's launches, each
        // as long as its longest group, into 4 rounds: 647 ms against 420.)
        bool one_batch = false;
        if (option(OPT_MSA2_BUDGET_GB) <= 0 && option(OPT_MSA2_BATCHES) <= 1 && jobs_all <= job_budget) {
            const long long plane0 = ext_all / m2_ext_planes(unitw_plan);
            const long long rows_est = 2 * cells_est;
            const long long extra = std::max<long long>(48LL << 30, plane0) - plane0 + (48LL << 30) + rows_est + (8LL << 30);
            one_batch = mem_all + extra <= static_cast<long long>(free_b) + reusable;
            if (one_batch) nb = 1;
        }
        const long long want = option(OPT_MSA2_BATCHES) > 0 ? option(OPT_MSA2_BATCHES) : 1;
        if (!one_batch && std::max(nb, want) > 1) nb = std::max<long long>(want, std::max((2 * mem_all + mem_budget - 1) / mem_budget, (jobs_all + job_budget - 1) / job_budget));
        nb = std::min<long long>(nb, static_cast<long long>(todo.size()));
        std::vector<M2Batch> batches(static_cast<size_t>(nb));
        {
            // groups of more than M2_NB reads (several wavefronts each, the long chains of joins): contiguous pieces of the sorted
            // list, by their share of the memory and of the jobs of all such groups; the rest: every nb-th group, so that every
            // batch has the same bulk of one-wavefront groups to fill the chip beside its large ones
            size_t nbig = 0;
            while (nbig < todo.size() && grp_off[ids[todo[nbig]] + 1] - grp_off[ids[todo[nbig]]] > M2_NB) ++nbig;
            const long long bmem = cmem[nbig], bjobs = cjobs[nbig];
            size_t k = 0;
            for (size_t x = 0; x < nbig; ++x) {
                const double share = std::max(bmem > 0 ? static_cast<double>(cmem[x]) / static_cast<double>(bmem) : 0.0,
                                              bjobs > 0 ? static_cast<double>(cjobs[x]) / static_cast<double>(bjobs) : 0.0);
                while (k + 1 < static_cast<size_t>(nb) && share * static_cast<double>(nb) >= static_cast<double>(k + 1)) ++k;
                batches[k].ids.push_back(ids[todo[x]]);
                batches[k].slot.push_back(todo[x]);
            }
            for (size_t x = nbig; x < todo.size(); ++x) {
                M2Batch& B = batches[(x - nbig) % static_cast<size_t>(nb)];
                B.ids.push_back(ids[todo[x]]);
                B.slot.push_back(todo[x]);
            }
            // (a call with fewer groups than batches, or one huge group taking the share of two pieces, leaves a batch without groups)
            batches.erase(std::remove_if(batches.begin(), batches.end(), [](const M2Batch& B) { return B.ids.empty(); }), batches.end());
        }
        nbatches += static_cast<double>(batches.size());
        const char* const pfs[2] = {"m2a", "m2b"};
        auto prepare = [&](size_t k) -> int {
            M2Batch& B = batches[k];
            double th = m2_now();
            SL_TRY(m2_plan(B, grp_off, grp, rel.data(), gsum.data(), gmx.data(), exact_w, bandwidth));
            m2_host_time("plan", th);
            B.pair_done = MS.pair[k & 1];
            B.alias_tiles = batches.size() == 1;
            SL_TRY(m2_prepare(B, pfs[k & 1], d_seq, match, mismatch, gap_extension, gap_opening, bandwidth, &cells, first, sp));
            pairs += static_cast<double>(B.njobs);
            if (first && overlap) SL_TRY((*overlap)());
            first = false;
            return 0;
        };
        SL_TRY(prepare(0));
        for (size_t k = 0; k < batches.size(); ++k) {
            M2Batch& B = batches[k];
            if (k + 1 < batches.size()) SL_TRY(prepare(k + 1));   // (its workspaces: those of batch k - 1, finished and read back)
            SL_TRY(m2_merge(B, pfs[k & 1], counters, s));
            double th = m2_now();
            long long need = used;
            std::vector<long long> boff(B.groups.size(), 0);
            std::vector<int32_t> bw = B.width;
            // The rows of a batch are laid out in the CALLER's group order (the batch itself is ordered by group size): when one
            // batch holds every group -- the usual case -- the buffer is then already what msa_run hands on, and its gathering
            // copy (4.6 GB of vote codes at 10^6 reads: 5 ms) is skipped.
            std::vector<size_t> byslot(B.groups.size());
            std::iota(byslot.begin(), byslot.end(), size_t(0));
            std::sort(byslot.begin(), byslot.end(), [&](size_t x, size_t y) { return B.slot[x] < B.slot[y]; });
            for (size_t q : byslot) {
                if (B.ovf[q]) {   // profile capacity exceeded: next pass
                    // (with profiles as wide as 16-bit columns allow and still too narrow: the alignment is wider than 65 535 columns,
                    // beyond spec v2 -- the caller hands the group to spec v1)
                    if (exact_w) gave_up->push_back(B.slot[q]); else again.push_back(B.slot[q]);
                    bw[q] = 0;
                    continue;
                }
                width[B.slot[q]] = bw[q];
                off[B.slot[q]] = need;
                boff[q] = need;
                need += static_cast<long long>(bw[q]) * B.groups[q].n;
            }
            SL_TRY(rows_reserve(static_cast<size_t>(used) * cs, (static_cast<size_t>(need) + 1) * cs));
            SL_HIP(hipMemcpyAsync(B.a.width, bw.data(), sizeof(int32_t) * bw.size(), hipMemcpyHostToDevice, s));
            B.width = bw;
            if (code.want && code.ready && !waited) { SL_HIP(hipStreamWaitEvent(s, code.ready, 0)); waited = true; }   // qualities in HBM
            SL_TRY(m2_write_batch(B, pfs[k & 1], boff, rows_ws.ptr, code, s));
            SL_HIP(hipStreamSynchronize(s));   // the batch's host vectors and workspaces are reused by the batch after the next
            m2_host_time("rows", th);
            used = need;
        }
        if (pass == 0) second_pass = static_cast<double>(again.size());
        todo.swap(again);
    }
    SL_HIP(hipStreamSynchronize(sp));
    if (pairs > 0) {
        int* d_stuck;
        int stuck = 0;
        SL_TRY(scratch("msa.stuck", 1, &d_stuck));
        SL_HIP(hipMemcpy(&stuck, d_stuck, sizeof stuck, hipMemcpyDeviceToHost));
        if (stuck) return fail("sarlacc_amd: internal error: an MSA traceback exceeded its step bound");
    }
    auto add = [&](const char* name, double v) { c.counts[name] = (c.counts.count(name) ? c.counts[name] : 0.0) + v; };
    add("msa_pairs", pairs);
    add("msa_cells", cells);
    add("msa2_rows", counters[M2C_ROWS]);
    add("msa2_rows_capped", counters[M2C_ROWS_CAPPED]);
    add("msa2_entries_filtered", counters[M2C_ENT_FILTERED]);
    add("msa2_rows_filtered", counters[M2C_ROWS_FILTERED]);
    add("msa2_entries_kept", counters[M2C_ENT_KEPT]);
    add("msa2_joins", counters[M2C_JOINS]);
    add("msa2_joins_chain_in_hbm", counters[M2C_JOINS_HBMQ]);
    add("msa2_gathers", counters[M2C_GATHERS]);
    add("msa2_groups_second_pass", second_pass);
    add("msa2_batches", nbatches);
    add("msa2_cycles_rows", counters[M2C_CYC_ROWS]);
    add("msa2_cycles_chain", counters[M2C_CYC_CHAIN]);
    add("msa2_cycles_walk", counters[M2C_CYC_WALK]);
    add("msa2_cycles_renumber", counters[M2C_CYC_RENUMBER]);
    add("msa2_launches", counters[M2C_T_START]);
    add("msa2_first_exit_s", counters[M2C_T_FIRST_EXIT]);
    add("msa2_last_exit_s", counters[M2C_T_LAST_EXIT]);
    add("msa2_exit_s_1wave", counters[M2C_T_EXIT1]);
    add("msa2_exit_s_4waves", counters[M2C_T_EXIT4]);
    add("msa2_exit_s_8waves", counters[M2C_T_EXIT8]);
    if (!rows_ws.ptr) SL_TRY(rows_reserve(0, 16));
    *d_rows = static_cast<uint8_t*>(rows_ws.ptr);
    return 0;
}

__global__ void k_rows_copy(const uint8_t* src, const long long* src_off, uint8_t* dst, const long long* dst_off, const long long* nbytes) {
    const long long g = blockIdx.x;
    const long long n = nbytes[g];
    const uint8_t* s = src + src_off[g];
    uint8_t* d = dst + dst_off[g];
    for (long long k = threadIdx.x; k < n; k += blockDim.x) d[k] = s[k];
}

// The MSA stage of quick_msa: spec v2 for groups of up to M2_MAXN reads whose profiles fit 65535 columns,
// spec v1 (msa.hip) for the rest and when spec 1 is selected (sarlacc_set_msa_spec / SARLACC_MSA_SPEC=1).
int msa_run(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq, const int64_t* seq_off,
            int64_t nseq, double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
            bool want_rows, int64_t out_cap, MsaResult* res, const std::function<int()>* overlap,
            const uint8_t* d_seq_resident) {
    if (msa_spec() == 1)
        return msa1_run(grp_off, grp, ngroups, seq, seq_off, nseq, match, mismatch, gap_extension, gap_opening, bandwidth, want_rows,
                        out_cap, res, overlap, d_seq_resident);
    int32_t* width_out = res->width.data();
    int64_t* out_off = res->out_off.data();
    res->d_out = nullptr;
    res->d_members = nullptr;
    out_off[0] = 0;
    if (bandwidth < 0) return fail("sarlacc_amd: negative bandwidth");
    const int64_t nmemb = grp_off[ngroups] - grp_off[0];
    for (int64_t i = 0; i < nmemb; ++i) {
        const int32_t v = grp[grp_off[0] + i];
        if (v < 1 || v > nseq) return fail("sarlacc_amd: group index %d outside 1..%lld", v, static_cast<long long>(nseq));
    }
    std::vector<int64_t> rel(static_cast<size_t>(nseq) + 1);
    for (int64_t i = 0; i <= nseq; ++i) rel[i] = (nseq ? seq_off[i] : 0) - (nseq ? seq_off[0] : 0);
    // which groups does spec v2 take
    std::vector<int64_t> v2, v1;
    for (int64_t g = 0; g < ngroups; ++g) {
        const int64_t n = grp_off[g + 1] - grp_off[g];
        int64_t mx = 0;
        for (int64_t a = 0; a < n; ++a) { const int32_t id = grp[grp_off[g] + a]; mx = std::max<int64_t>(mx, rel[id] - rel[id - 1]); }
        if (mx > 60000) return fail("sarlacc_amd: reads longer than 60000 bases are not supported by the MSA stage");
        // spec v2: up to M2_MAXN reads, each short enough for 16-bit positions (a profile that outgrows 65 535 COLUMNS sends its group
        // to spec v1 afterwards: msa2_core's gave_up)
        if (n <= M2_MAXN && mx + 64 <= 65535) v2.push_back(g); else v1.push_back(g);
    }
    SL_TRY(ensure_device());
    Context& c = ctx();
    hipStream_t s = nullptr;
    c.stage_reset("msa_pairwise");
    c.stage_reset("msa_merge");
    for (const char* nm : {"msa_host_plan_s", "msa_host_upload_alloc_s", "msa_host_pairwise_launch_s", "msa_host_rows_s", "msa_host_select_s",
                           "msa_host_total_s"})
        c.counts[nm] = 0;
    const double t_run = m2_now();
    for (const char* nm : {"msa_pairs", "msa_cells", "msa2_rows", "msa2_rows_capped", "msa2_entries_filtered", "msa2_rows_filtered",
                           "msa2_entries_kept", "msa2_joins", "msa2_joins_chain_in_hbm", "msa2_gathers", "msa2_groups_second_pass", "msa2_batches", "msa2_cycles_rows", "msa2_cycles_chain",
                           "msa2_cycles_walk", "msa2_cycles_renumber", "msa2_launches", "msa2_first_exit_s", "msa2_last_exit_s",
                           "msa2_exit_s_1wave", "msa2_exit_s_4waves", "msa2_exit_s_8waves", "msa_pairs_bitvector", "msa_bitvector_tile_bytes",
                           "msa_bitvector_split", "msa_bitvector_redone"})
        c.counts[nm] = 0;
    if (v2.empty())
        return msa1_run(grp_off, grp, ngroups, seq, seq_off, nseq, match, mismatch, gap_extension, gap_opening, bandwidth, want_rows,
                        out_cap, res, overlap, d_seq_resident);
    const int64_t total = rel[nseq];
    uint8_t* d_seq;
    if (d_seq_resident) d_seq = const_cast<uint8_t*>(d_seq_resident);
    else SL_TRY(upload("msa.seq", reinterpret_cast<const uint8_t*>(seq) + (nseq ? seq_off[0] : 0), static_cast<size_t>(total), &d_seq, s));
    int32_t* d_mem;
    SL_TRY(upload("msa.mem.all", grp + grp_off[0], static_cast<size_t>(nmemb), &d_mem, s));   // (msa1_run uploads its own "msa.mem")
    res->d_members = d_mem;

    std::vector<int32_t> w2;
    std::vector<long long> o2;
    uint8_t* d_rows2 = nullptr;
    std::vector<size_t> gave_up;
    SL_TRY(msa2_core(grp_off, grp, v2, d_seq, rel, match, mismatch, gap_extension, gap_opening, bandwidth, w2, o2, &d_rows2, overlap, res->code, s, &gave_up));
    if (!gave_up.empty()) {   // alignments wider than 65 535 columns: spec v1 (the list stays in group order)
        for (size_t q : gave_up) v1.push_back(v2[q]);
        std::sort(v1.begin(), v1.end());
    }
    c.counts["msa_v1_fallback"] = static_cast<double>(v1.size());
    c.counts["msa_v1_fallback_too_wide"] = static_cast<double>(gave_up.size());
    // spec v1 part on a compacted group list: more than M2_MAXN reads, reads too long, or dropped by the guard
    MsaResult r1;
    std::vector<int64_t> g1off(v1.size() + 1, 0);
    std::vector<int32_t> g1;
    if (!v1.empty()) {
        for (size_t q = 0; q < v1.size(); ++q) {
            const int64_t g = v1[q];
            for (int64_t k = grp_off[g]; k < grp_off[g + 1]; ++k) g1.push_back(grp[k]);
            g1off[q + 1] = static_cast<int64_t>(g1.size());
        }
        r1.width.assign(v1.size(), 0);
        r1.out_off.assign(v1.size() + 1, 0);
        r1.code = res->code;
        SL_TRY(msa1_run(g1off.data(), g1.data(), static_cast<int64_t>(v1.size()), seq, seq_off, nseq, match, mismatch, gap_extension,
                        gap_opening, bandwidth, true, -1, &r1, nullptr, d_seq, true));
    }
    // the rows of spec v2 are in processing order, those of spec v1 in a buffer of their own: gather both into one in group order
    std::vector<long long> src_off(static_cast<size_t>(ngroups)), dst_off(static_cast<size_t>(ngroups)), nbytes(static_cast<size_t>(ngroups));
    std::vector<char> from1(static_cast<size_t>(ngroups), 0);
    {
        size_t a1 = 0, a2 = 0;
        for (int64_t g = 0; g < ngroups; ++g) {
            const int64_t n = grp_off[g + 1] - grp_off[g];
            const bool in2 = a2 < v2.size() && v2[a2] == g;
            if (a1 < v1.size() && v1[a1] == g) { width_out[g] = r1.width[a1]; src_off[g] = r1.out_off[a1]; from1[g] = 1; ++a1; }
            else { width_out[g] = w2[a2]; src_off[g] = o2[a2]; }
            if (in2) ++a2;
            nbytes[g] = static_cast<long long>(width_out[g]) * n;
            dst_off[g] = out_off[g];
            out_off[g + 1] = out_off[g] + nbytes[g];
        }
    }
    if (want_rows && out_cap >= 0 && out_cap < out_off[ngroups]) return fail("sarlacc_amd: MSA output buffer too small (%lld needed)", static_cast<long long>(out_off[ngroups]));
    // spec v2's buffer already in group order (one batch, no group by spec v1, none that needed the second pass): it IS the result
    bool in_place = v1.empty();
    for (int64_t g = 0; g < ngroups && in_place; ++g) in_place = src_off[g] == dst_off[g];
    if (in_place) {
        SL_HIP(hipStreamSynchronize(s));
        if (res->code.want) res->d_codes = reinterpret_cast<uint16_t*>(d_rows2);
        else res->d_out = d_rows2;
        m2_host_time("total", t_run);
        return 0;
    }
    // (offsets and sizes so far are in cells; a cell is one character, or one 16-bit vote code)
    const long long cs = res->code.want ? 2 : 1;
    uint8_t* d_final;
    SL_TRY(scratch(res->code.want ? "msa.final16" : "msa.final", static_cast<size_t>(out_off[ngroups] * cs) + 8, &d_final));
    if (cs > 1)
        for (int64_t g = 0; g < ngroups; ++g) { src_off[g] *= cs; dst_off[g] *= cs; nbytes[g] *= cs; }
    // two launches, one per source buffer (groups of the other kind copy nothing)
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<long long> nb(nbytes);
        for (int64_t g = 0; g < ngroups; ++g)
            if ((from1[g] != 0) != (pass == 0)) nb[g] = 0;
        long long *d_so, *d_do, *d_nb;
        SL_TRY(upload(pass == 0 ? "msa.cp.so1" : "msa.cp.so2", src_off.data(), src_off.size(), &d_so, s));
        SL_TRY(upload(pass == 0 ? "msa.cp.do1" : "msa.cp.do2", dst_off.data(), dst_off.size(), &d_do, s));
        SL_TRY(upload(pass == 0 ? "msa.cp.nb1" : "msa.cp.nb2", nb.data(), nb.size(), &d_nb, s));
        if (pass == 0 && v1.empty()) continue;
        const uint8_t* src1 = res->code.want ? reinterpret_cast<const uint8_t*>(r1.d_codes) : r1.d_out;
        hipLaunchKernelGGL(k_rows_copy, dim3(static_cast<unsigned>(ngroups)), dim3(256), 0, s, pass == 0 ? src1 : d_rows2, d_so, d_final, d_do, d_nb);
        SL_HIP(hipGetLastError());
    }
    SL_HIP(hipStreamSynchronize(s));
    if (res->code.want) res->d_codes = reinterpret_cast<uint16_t*>(d_final);
    else res->d_out = d_final;
    m2_host_time("total", t_run);
    return 0;
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" int sarlacc_set_msa_spec(int spec) {
    if (spec != 0 && spec != 1 && spec != 2) return fail("sarlacc_amd: MSA spec must be 1 (centre-star) or 2 (consistency-based progressive), 0 = default");
    return sarlacc::set_option("msa_spec", spec);
}
