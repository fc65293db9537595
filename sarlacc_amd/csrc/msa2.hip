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
//   col   int32   column of every base in the profile that currently holds its read;
//   pos   uint16  per group n x wcap: position of read a at column c of its profile (0xFFFF = gap);
//   rows  per profile column of the first child of a merge: up to `cap` (partner column, weight) entries.
// Kernels, per merge round k (every group performs its k-th join in the same launches):
//   k_m2_gather  lane = column i of the first child; walks the library (map -> map -> col) for every
//                member pair and third sequence, sums the weights per partner column in a private list
//                (registers), writes the list sorted by column; with unit weights (the default scores)
//                k_m2_candidates + k_m2_gather_unit: the round's (c, b) pairs as a table of scalar descriptors;
//   k_m2_chain   one wavefront per group: heaviest chain over the lists in row order with a Fenwick tree
//                of prefix maxima in LDS (16 lanes read / update the <= 16 nodes of one match in one
//                instruction), then the traceback through the stored predecessors;
//   k_m2_merge   one workgroup per group: new column numbers (first child's unmatched columns before the
//                second child's between two matched pairs), col / pos of every member updated.
// A row keeps the first M2_CAP distinct partner columns (spec v2, step 5); groups whose profiles outgrow the
// fast capacity are redone with profiles as wide as the sum of the read lengths.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "msa_common.hpp"

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <numeric>
#include <vector>

namespace sarlacc {

constexpr int M2_MAXN = 32;          // group sizes aligned by spec v2 (member sets are 32-bit masks)
constexpr int M2_CAP = 16;           // partner columns per row on the fast path (private lists in LDS)
constexpr unsigned M2_NONE = 0xFFFFu;
// profile capacity of the first pass: same-molecule reads grow a profile by 10-20 %, one or two unrelated reads in
// the cluster by their length each
#define M2_FASTW(maxlen) (3 * (maxlen) + 64)

struct M2Member {         // one read of a group
    long long seq_off;    // into d_seq
    long long map_base;   // into d_map: (n - 1) arrays of `len` entries, the other members in member order
    long long col_base;   // into d_col
    int len;
    int pad;
};
struct M2Group {
    int first_member;     // member a of the group is members[first_member + a]; joins at first_member + k,
                          // tree nodes (leaves 0..n-1, join k creates n + k) at 2 * first_member + node
    int n;
    int wcap;             // capacity of a profile in columns
    int cap;              // entries per row list
    long long row_base;   // into the per-column arrays (lists, partners, renumbering): wcap entries
    long long pos_base;   // into d_pos: n * wcap
    long long first_job;  // pairwise job of (a, b), a < b: first_job + a n - a (a + 1) / 2 + b - a - 1
    long long dist_base;  // into the tree kernel's scratch: n * n + n doubles
    long long out_off;    // the group's rows in the row buffer
};

struct M2Args {
    int g0;                        // first group of this launch (the round kernels run on sub-ranges of a batch, one per stream)
    const uint8_t* seq;
    const M2Group* groups;
    const M2Member* members;
    int ngroups;
    int ma, mm;
    unsigned long long* clk;       // SARLACC_MSA2_CLOCKS: cycles of the chain kernel's phases, first wave of the last launch
    unsigned long long* xdbg;      // SARLACC_MSA2_EXACTDBG: per group (cycles, rows << 32 | entries) of k_m2_chain_exact
    const uint16_t* map;
    const int2* stats;
    double* dist;
    int2* joins;
    uint32_t* nodemask;
    int* ncols;
    int* col;
    uint16_t* pos;
    uint16_t* row_cnt;             // entries in a row's list
    unsigned long long* row_ent;   // (partner column << 32) | weight
    unsigned* row_pred;            // id of the predecessor of every entry on its best chain (0: none)
    int* part;                     // partner column of column i of the first child, -1 if unmatched
    int* ovf;                      // per group: a capacity was exceeded
    int* redo;                     // per group: this round's chain has to be done by k_m2_chain_exact
    int32_t* width;                // per group: columns of the final profile
    uint8_t* out;                  // gapped rows
    // rows as vote codes instead of characters (CodeSpec, common.hpp): out16 != nullptr
    uint16_t* out16;
    const uint8_t* qual;           // laid out like seq
    int qoffset, navail;
    int* bad;
};

__device__ __forceinline__ int m2_w0(int x, int y, int ma, int mm) {
    const int s = (x == y) ? ma : mm;
    return s > 1 ? s : 1;
}

// ---- guide tree: one thread per group (oracle/msa2.c nj_tree, operation for operation) ----
__global__ void k_m2_tree(M2Args A) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.ngroups) return;
    const M2Group G = A.groups[g];
    const int n = G.n;
    const int fm = G.first_member;
    for (int a = 0; a < n; ++a) {
        A.nodemask[2 * fm + a] = 1u << a;
        A.ncols[2 * fm + a] = A.members[fm + a].len;
    }
    if (n < 2) return;
    double* D = A.dist + G.dist_base;
    double* R = D + static_cast<long long>(n) * n;
    for (int a = 0; a < n; ++a) {
        D[a * n + a] = 0.0;
        for (int b = a + 1; b < n; ++b) {
            const int2 st = A.stats[G.first_job + static_cast<long long>(a) * n - static_cast<long long>(a) * (a + 1) / 2 + b - a - 1];
            const long long alen = static_cast<long long>(A.members[fm + a].len) + A.members[fm + b].len - st.y;
            const double d = alen > 0 ? 1.0 - static_cast<double>(st.x) / static_cast<double>(alen) : 0.0;
            D[a * n + b] = d;
            D[b * n + a] = d;
        }
    }
    unsigned active = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    int node[M2_MAXN];
    for (int i = 0; i < n; ++i) node[i] = i;
    int r = n, nj = 0;
    while (r > 3) {
        for (int i = 0; i < n; ++i) {
            if (!((active >> i) & 1u)) continue;
            double sum = 0.0;
            for (int k = 0; k < n; ++k)
                if (((active >> k) & 1u) && k != i) sum = sum + D[i * n + k];
            R[i] = sum;
        }
        int bi = -1, bj = -1;
        double best = 0.0;
        const double rm2 = static_cast<double>(r - 2);
        for (int i = 0; i < n; ++i) {
            if (!((active >> i) & 1u)) continue;
            for (int j = i + 1; j < n; ++j) {
                if (!((active >> j) & 1u)) continue;
                const double q = (rm2 * D[i * n + j] - R[i]) - R[j];
                if (bi < 0 || q < best) { best = q; bi = i; bj = j; }
            }
        }
        A.joins[fm + nj] = make_int2(node[bi], node[bj]);
        const double dij = D[bi * n + bj];
        for (int k = 0; k < n; ++k) {
            if (!((active >> k) & 1u) || k == bi || k == bj) continue;
            const double v = ((D[bi * n + k] + D[bj * n + k]) - dij) * 0.5;
            D[bi * n + k] = v;
            D[k * n + bi] = v;
        }
        active &= ~(1u << bj);
        node[bi] = n + nj;
        ++nj; --r;
    }
    int l[3], c = 0;
    for (int i = 0; i < n && c < 3; ++i)
        if ((active >> i) & 1u) l[c++] = i;
    if (c >= 2) {
        A.joins[fm + nj] = make_int2(node[l[0]], node[l[1]]);
        node[l[0]] = n + nj;
        ++nj;
    }
    if (c == 3) A.joins[fm + nj] = make_int2(node[l[0]], node[l[2]]);
}

// ---- leaves: col = position, pos = identity ----
__global__ void k_m2_init(M2Args A, const int* member_group, int nmembers) {
    const int m = blockIdx.y;
    if (m >= nmembers) return;
    const M2Group G = A.groups[member_group[m]];
    const M2Member Me = A.members[m];
    const int a = m - G.first_member;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Me.len; p += gridDim.x * blockDim.x) {
        A.col[Me.col_base + p] = p;
        A.pos[G.pos_base + static_cast<long long>(a) * G.wcap + p] = static_cast<uint16_t>(p);
    }
}

// ---- match lists of one merge round ----
// lane = column i of the first child.  The row's list (first M2_CAP distinct partner columns in the canonical
// enumeration order of spec v2, step 5: a ascending; direct edges b ascending; then c ascending, b ascending)
// lives in registers; a candidate is compared with the first entries before anything else (same-molecule reads
// agree on 1-3 columns).  Loads are issued in independent batches: the positions r_c of every third sequence
// first (LDS), then up to M2_BATCH (c, b) pairs at a time -- the dependent chain map -> map -> col of one
// candidate is three memory latencies long, so the pairs of a batch are looked up side by side.
constexpr int M2_BATCH = 4;

__device__ __forceinline__ long long m2_uniform64(long long v) {   // a wave-uniform 64-bit value into scalar registers
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)));
    return static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <bool UNITW>
__global__ void __launch_bounds__(64) k_m2_gather(M2Args A, int round) {
    __shared__ uint16_t s_r[M2_MAXN][64];   // position of the lane's base in every other member (0xFFFF: gap)
    __shared__ long long s_mapbase[M2_MAXN], s_colbase[M2_MAXN], s_seqoff[M2_MAXN];   // the members' descriptors
    __shared__ int s_len[M2_MAXN];
    __shared__ int s_b[M2_MAXN];   // the second child's members, ascending
    const int g = A.g0 + blockIdx.y;
    const M2Group G = A.groups[g];
    if (round >= G.n - 1 || A.ovf[g] != 0) return;
    const int n = G.n, fm = G.first_member;
    const int2 jn = A.joins[fm + round];
    const unsigned maskA = A.nodemask[2 * fm + jn.x], maskB = A.nodemask[2 * fm + jn.y];
    const int nA = A.ncols[2 * fm + jn.x];
    const int lane = threadIdx.x;
    if (lane < M2_MAXN) {
        const M2Member Me = A.members[fm + min(lane, n - 1)];
        s_mapbase[lane] = Me.map_base; s_colbase[lane] = Me.col_base; s_seqoff[lane] = Me.seq_off; s_len[lane] = lane < n ? Me.len : 0;
        if ((maskB >> lane) & 1u) s_b[__popc(maskB & ((1u << lane) - 1u))] = lane;
    }
    const int nbm = __popc(maskB);
    __syncthreads();
    const unsigned long long gclk0 = __builtin_amdgcn_s_memtime();
    unsigned long long gcand = 0;
    for (int i0 = blockIdx.x * 64; i0 < nA; i0 += gridDim.x * 64) {
    const int i = i0 + lane;
    int ej[M2_CAP], ew[M2_CAP];
#pragma unroll
    for (int k = 0; k < M2_CAP; ++k) { ej[k] = -1; ew[k] = 0; }
    int cnt = 0;
    // add(j, w): selects only, every index a compile-time constant -- the lists must stay in registers (a lambda
    // capturing the arrays, or an index the compiler cannot resolve, sends them to scratch memory, whose loads
    // then wait behind every lookup in flight)
#define M2_ADD(J, W, VALID)                                                                            \
    {                                                                                                  \
        const int j_ = (J), w_ = (W);                                                                  \
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
            _Pragma("unroll") for (int k_ = 0; k_ < M2_CAP; ++k_) {                                    \
                const bool s_ = app_ && k_ == cnt;                                                     \
                ej[k_] = s_ ? j_ : ej[k_];                                                             \
                ew[k_] = s_ ? w_ : ew[k_];                                                             \
            }                                                                                          \
            cnt += app_ ? 1 : 0;                                                                       \
        }                                                                                              \
    }
    const bool row = i < nA;
    for (int a = 0; a < n; ++a) {
        if (!((maskA >> a) & 1u)) continue;
        const unsigned p = row ? A.pos[G.pos_base + static_cast<long long>(a) * G.wcap + i] : M2_NONE;
        const bool havep = p != M2_NONE;
        if (!__ballot(havep)) continue;
        const M2Member Ma = A.members[fm + a];
        const int xa = (UNITW || !havep) ? 0 : dna5_code(A.seq[Ma.seq_off + p]);
        // positions in every other member: unconditional loads (clamped), selected afterwards -- see below
        {
            const unsigned pidx = havep ? p : 0u;
            for (int c0 = 0; c0 < n; c0 += 8) {
                uint16_t rv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = min(c0 + u, n - 1);
                    const int slot = c == a ? 0 : (c < a ? c : c - 1);
                    rv[u] = A.map[Ma.map_base + static_cast<long long>(slot) * Ma.len + pidx];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + u < n) s_r[c0 + u][lane] = (c0 + u != a && havep) ? rv[u] : static_cast<uint16_t>(M2_NONE);
            }
        }
        // canonical order of the candidates: the direct edges (c = b) for b in B ascending, then for c ascending the
        // triplets (c, b), b in B ascending, b != c -- a flat list of nbm + (n - 1) nbm entries, looked up M2_BATCH at a
        // time side by side (the second child is often a single read: batching over its members alone would leave
        // one candidate per memory latency).
        const int total = nbm * n;   // direct pass (ci = 0) + the n - 1 third sequences
        int ci = 0, bi = 0;          // position in the flat list: pass ci (0 = direct; ci >= 1: c = ci - 1, skipping a), member bi
        for (int f0 = 0; f0 < total; f0 += M2_BATCH) {
            // every load of the batch is unconditional (indices clamped to valid ones, results selected afterwards):
            // a branch around a load makes the compiler wait for it before it issues the next one
            unsigned qq[M2_BATCH];
            int jj[M2_BATCH], ww[M2_BATCH], bq[M2_BATCH];
            bool ok[M2_BATCH];
            // phase 1: the M2_BATCH map lookups are requested back to back -- nothing between them uses a loaded value,
            // so the compiler does not wait for one before it issues the next (uses in the same loop body would)
            unsigned rr[M2_BATCH], mm[M2_BATCH], rix[M2_BATCH];
            int cuq[M2_BATCH], flg[M2_BATCH];   // flg: bit 0 candidate exists, bit 1 direct edge (c = b), bit 2 counted (direct or b != c)
#pragma unroll
            for (int u = 0; u < M2_BATCH; ++u) {
                const bool in = f0 + u < total;
                // the candidate's members are the same for every lane: their descriptors are moved to scalar registers
                // so that the address arithmetic runs on the scalar unit (values read from LDS arrive in VGPRs)
                const int b = __builtin_amdgcn_readfirstlane(s_b[bi]);
                const int cu = ci == 0 ? b : ((ci - 1) + ((ci - 1) >= a ? 1 : 0));
                const unsigned r = s_r[cu][lane];
                const int bslot = cu == b ? 0 : (b < cu ? b : b - 1);
                const int lencu = __builtin_amdgcn_readfirstlane(s_len[cu]);
                const long long mbase = m2_uniform64(s_mapbase[cu]) + static_cast<long long>(bslot) * lencu;
                const unsigned ridx = min(r != M2_NONE ? r : 0u, static_cast<unsigned>(max(lencu - 1, 0)));
                mm[u] = A.map[mbase + ridx];
                rr[u] = r; rix[u] = ridx; cuq[u] = cu; bq[u] = b;
                flg[u] = (in ? 1 : 0) | (cu == b ? 2 : 0) | ((ci == 0 || b != cu) ? 4 : 0);
                bi = in ? bi + 1 : bi;
                const bool wrap = bi == nbm;
                bi = wrap ? 0 : bi;
                ci = wrap ? ci + 1 : ci;
            }
#pragma unroll
            for (int u = 0; u < M2_BATCH; ++u) {
                qq[u] = (flg[u] & 2) ? rr[u] : mm[u];
                ok[u] = (flg[u] & 1) && (flg[u] & 4) && rr[u] != M2_NONE && qq[u] != M2_NONE;
                ww[u] = 1;
                if (!UNITW) {
                    const int xc = dna5_code(A.seq[s_seqoff[cuq[u]] + rix[u]]);
                    ww[u] = (xc << 16) | (((flg[u] & 2) ? 1 : 0) << 15) | m2_w0(xa, xc, A.ma, A.mm);   // (finished below)
                }
            }
#pragma unroll
            for (int u = 0; u < M2_BATCH; ++u) {
                const int b = __builtin_amdgcn_readfirstlane(bq[u]);
                const int lenb = __builtin_amdgcn_readfirstlane(s_len[b]);
                const unsigned qidx = min(ok[u] ? qq[u] : 0u, static_cast<unsigned>(max(lenb - 1, 0)));
                jj[u] = A.col[m2_uniform64(s_colbase[b]) + qidx];
                if (!UNITW) {
                    const int xc = ww[u] >> 16, wac = ww[u] & 0x7fff;
                    const bool direct = (ww[u] >> 15) & 1;
                    const int wcb = m2_w0(xc, dna5_code(A.seq[s_seqoff[b] + qidx]), A.ma, A.mm);
                    ww[u] = direct ? wac : (wac < wcb ? wac : wcb);
                }
            }
#pragma unroll
            for (int u = 0; u < M2_BATCH; ++u)
                if (f0 + u < total) M2_ADD(jj[u], ww[u], ok[u])
        }
        gcand += total;
    }
    if (row) {
        // noise filter (spec v2, step 5): entries lighter than half the row's heaviest are dropped -- the partner
        // columns reached only through reads unrelated to the rest of the cluster
        int wmax = 0;
#pragma unroll
        for (int k = 0; k < M2_CAP; ++k) wmax = max(wmax, ew[k]);
        {
            int kept = 0;
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) {
                const bool keep = k < cnt && 2 * ew[k] >= wmax;
                if (!keep) ej[k] = 0x7fffffff;   // sorts behind every kept column
                kept += keep ? 1 : 0;
            }
            cnt = kept;
        }
        // by column: rank of every entry among the valid ones (the columns of a list are distinct)
        unsigned long long* const mine = A.row_ent + (G.row_base + i) * static_cast<long long>(G.cap);
#pragma unroll
        for (int k = 0; k < M2_CAP; ++k) {
            if (ej[k] != 0x7fffffff && ej[k] >= 0) {
                int rank = 0;
#pragma unroll
                for (int q = 0; q < M2_CAP; ++q)
                    if (ej[q] >= 0 && ej[q] < ej[k]) ++rank;
                mine[rank] = (static_cast<unsigned long long>(static_cast<unsigned>(ej[k])) << 32) | static_cast<unsigned>(ew[k]);
            }
        }
        A.row_cnt[G.row_base + i] = static_cast<uint16_t>(cnt);
    }
    }
    if (A.clk && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { A.clk[4] = __builtin_amdgcn_s_memtime() - gclk0; A.clk[5] = gcand; A.clk[6] = static_cast<unsigned long long>(nA); A.clk[7] = static_cast<unsigned long long>(n); }
}

// ---- unit weights (the default scores): the candidates of a round as a table of scalar descriptors ----
// Entry e = pass * |B| + bi of a group's table: pass 0 the direct edges (c = b), pass 1 + c the triplets through
// member c, each for the second child's members b ascending -- the canonical candidate order of k_m2_gather, with the
// block of c == a left in: the lane's position "in a itself" is a gap by construction, so those entries add nothing.
//   M2Cand { map of (c -> b), columns of b, len(c) - 1, len(b) - 1, LDS row of c }
// A direct edge reads the identity map (position in b = the staged position itself); an entry that does not count
// (c == b in a triplet pass) and the M2_UBATCH - 1 padding entries behind the table point at an LDS row that holds only
// gaps.  The gather therefore has no flags and no special cases: every entry is
//   r = row[c][lane];  q = map[min(r, len(c) - 1)];  j = col[min(q, len(b) - 1)];  valid = r, q are not gaps
// and, read with wave-uniform indices, the descriptors arrive by scalar loads -- no descriptor arithmetic on the
// vector unit, nothing moved through LDS and v_readfirstlane.
constexpr int M2_UBATCH = 4;   // candidates looked up side by side by k_m2_gather_unit
typedef const __attribute__((address_space(1))) uint16_t m2_gu16;
typedef const __attribute__((address_space(1))) int m2_gi32;
struct __attribute__((aligned(32))) M2Cand {
    const uint16_t* map;
    const int* col;
    int lenc_m1, lenb_m1;
    int row_off;    // byte offset of the LDS row (M2_MAXN: the all-gaps row)
    int pad;
};
static_assert(sizeof(M2Cand) == 32, "M2Cand is read as one 8-dword scalar load");

__global__ void k_m2_identity(uint16_t* ident) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < 65536) ident[x] = static_cast<uint16_t>(x);
}

__global__ void __launch_bounds__(64) k_m2_candidates(M2Args A, int round, M2Cand* tab, const long long* tab_off, const uint16_t* ident) {
    __shared__ int s_b[M2_MAXN];
    const int g = A.g0 + blockIdx.x;
    const M2Group G = A.groups[g];
    if (round >= G.n - 1 || A.ovf[g] != 0) return;
    const int n = G.n, fm = G.first_member;
    const int2 jn = A.joins[fm + round];
    const unsigned maskB = A.nodemask[2 * fm + jn.y];
    const int lane = threadIdx.x;
    if (lane < M2_MAXN && ((maskB >> lane) & 1u)) s_b[__popc(maskB & ((1u << lane) - 1u))] = lane;
    const int nbm = __popc(maskB);
    __syncthreads();
    M2Cand* const T = tab + tab_off[g];
    const int E = nbm * (n + 1);
    for (int e = lane; e < E + M2_UBATCH - 1; e += 64) {
        M2Cand C;
        C.pad = 0;
        if (e >= E) {   // padding: the batch that holds the last entries reads up to M2_UBATCH - 1 more
            C.map = ident; C.col = A.col + A.members[fm].col_base; C.lenc_m1 = 0; C.lenb_m1 = 0; C.row_off = M2_MAXN * 128;
        } else {
            const int pass = e / nbm, b = s_b[e % nbm];
            const int cu = pass == 0 ? b : pass - 1;
            const M2Member Mc = A.members[fm + cu], Mb = A.members[fm + b];
            C.col = A.col + Mb.col_base;
            C.lenb_m1 = max(Mb.len - 1, 0);
            if (pass == 0) { C.map = ident; C.lenc_m1 = 65534; C.row_off = b * 128; }
            else {
                const int bslot = cu == b ? 0 : (b < cu ? b : b - 1);
                C.map = A.map + Mc.map_base + static_cast<long long>(bslot) * Mc.len;
                C.lenc_m1 = max(Mc.len - 1, 0);
                C.row_off = (cu == b ? M2_MAXN : cu) * 128;
            }
        }
        T[e] = C;
    }
}

// Unit weights: a list entry is one register, (column << 16) | weight (a column is < 65535, a weight at most
// |A| |B| (n + 1) <= 8448); 0xFFFF0000 is an empty slot, an invalid candidate carries a column no entry can hold.
// The columns of a list are distinct, so at most one entry matches.  Three tiers, each behind a wave-wide test:
// entries 0-3; entries 4-7 and the append into 0-7 (some lane of the wave meets a new column at most steps, so
// this tier has to be short); entries 8-15.
constexpr unsigned M2_EMPTY = 0xFFFF0000u;
#define M2_MATCH1(K0, K1)                                                                              \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                           \
        const bool m_ = (pe[k_] >> 16) == j_;                                                          \
        pe[k_] += m_ ? 1u : 0u;                                                                        \
        hit_ = hit_ || m_;                                                                             \
    }
#define M2_APPEND1(K0, K1)                                                                             \
    {                                                                                                  \
        const bool app_ = !hit_ && cnt < (K1);                                                         \
        _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) pe[k_] = (app_ && cnt == k_) ? ((j_ << 16) | 1u) : pe[k_]; \
        cnt += app_ ? 1 : 0;                                                                           \
        hit_ = hit_ || app_;                                                                           \
    }
#define M2_ADD1(J, VALID)                                                                              \
    {                                                                                                  \
        const bool v_ = (VALID);                                                                       \
        const unsigned j_ = v_ ? static_cast<unsigned>(J) : 0x1FFFFu;                                  \
        bool hit_ = !v_;                                                                               \
        M2_MATCH1(0, 4)                                                                                \
        if (__ballot(!hit_)) {                                                                         \
            M2_MATCH1(4, 8)                                                                            \
            M2_APPEND1(0, 8)                                                                           \
            if (__ballot(!hit_)) {                                                                     \
                M2_MATCH1(8, M2_CAP)                                                                   \
                /* a new column; beyond M2_CAP distinct columns it is ignored (spec v2, step 5) */     \
                M2_APPEND1(8, M2_CAP)                                                                  \
            }                                                                                          \
        }                                                                                              \
    }

__global__ void __launch_bounds__(64) k_m2_gather_unit(M2Args A, int round, const M2Cand* __restrict__ tab, const long long* __restrict__ tab_off) {
    __shared__ uint16_t s_r[M2_MAXN + 1][64];   // position of the lane's base in every other member (0xFFFF: gap); last row: gaps
    const int g = A.g0 + blockIdx.y;
    const M2Group G = A.groups[g];
    if (round >= G.n - 1 || A.ovf[g] != 0) return;
    const int n = G.n, fm = G.first_member;
    const int2 jn = A.joins[fm + round];
    const unsigned maskA = A.nodemask[2 * fm + jn.x], maskB = A.nodemask[2 * fm + jn.y];
    const int nA = A.ncols[2 * fm + jn.x];
    const int E = __popc(maskB) * (n + 1);
    const M2Cand* const T = tab + tab_off[g];
    const int lane = threadIdx.x;
    s_r[M2_MAXN][lane] = static_cast<uint16_t>(M2_NONE);   // (every lane reads its own column of s_r only)
    const unsigned char* const s_rlane = reinterpret_cast<const unsigned char*>(&s_r[0][lane]);
    for (int i0 = blockIdx.x * 64; i0 < nA; i0 += gridDim.x * 64) {
        const int i = i0 + lane;
        unsigned pe[M2_CAP];
#pragma unroll
        for (int k = 0; k < M2_CAP; ++k) pe[k] = M2_EMPTY;
        int cnt = 0;
        const bool row = i < nA;
        for (int a = 0; a < n; ++a) {
            if (!((maskA >> a) & 1u)) continue;
            const unsigned p = row ? A.pos[G.pos_base + static_cast<long long>(a) * G.wcap + i] : M2_NONE;
            const bool havep = p != M2_NONE;
            if (!__ballot(havep)) continue;
            const M2Member Ma = A.members[fm + a];
            {   // positions in every other member: unconditional loads (clamped), selected afterwards
                const unsigned pidx = havep ? p : 0u;
                for (int c0 = 0; c0 < n; c0 += 8) {
                    uint16_t rv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = min(c0 + u, n - 1);
                        const int slot = c == a ? 0 : (c < a ? c : c - 1);
                        rv[u] = A.map[Ma.map_base + static_cast<long long>(slot) * Ma.len + pidx];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c0 + u < n) s_r[c0 + u][lane] = (c0 + u != a && havep) ? rv[u] : static_cast<uint16_t>(M2_NONE);
                }
            }
            for (int f0 = 0; f0 < E; f0 += M2_UBATCH) {
                unsigned rr[M2_UBATCH], qq[M2_UBATCH];
                int jj[M2_UBATCH];
                // the map lookups of the batch, then the column lookups, each requested back to back
#pragma unroll
                for (int u = 0; u < M2_UBATCH; ++u) {
                    const M2Cand& C = T[f0 + u];
                    rr[u] = *reinterpret_cast<const uint16_t*>(s_rlane + C.row_off);
                    // (the pointers are global memory: said explicitly, a pointer read from memory would be dereferenced
                    // by flat loads, which wait on the LDS counter as well)
                    qq[u] = ((m2_gu16*)C.map)[min(rr[u], static_cast<unsigned>(C.lenc_m1))];   // (a gap clamps to a valid index)
                }
#pragma unroll
                for (int u = 0; u < M2_UBATCH; ++u)
                    jj[u] = ((m2_gi32*)T[f0 + u].col)[min(qq[u], static_cast<unsigned>(T[f0 + u].lenb_m1))];
#pragma unroll
                for (int u = 0; u < M2_UBATCH; ++u) M2_ADD1(jj[u], rr[u] != M2_NONE && qq[u] != M2_NONE)
            }
        }
        if (row) {
            // noise filter and ordering by column: as in k_m2_gather
            int ej[M2_CAP], ew[M2_CAP];
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) { ej[k] = k < cnt ? static_cast<int>(pe[k] >> 16) : -1; ew[k] = k < cnt ? static_cast<int>(pe[k] & 0xffffu) : 0; }
            int wmax = 0;
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) wmax = max(wmax, ew[k]);
            {
                int kept = 0;
#pragma unroll
                for (int k = 0; k < M2_CAP; ++k) {
                    const bool keep = k < cnt && 2 * ew[k] >= wmax;
                    if (!keep) ej[k] = 0x7fffffff;
                    kept += keep ? 1 : 0;
                }
                cnt = kept;
            }
            unsigned long long* const mine = A.row_ent + (G.row_base + i) * static_cast<long long>(G.cap);
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) {
                if (ej[k] != 0x7fffffff && ej[k] >= 0) {
                    int rank = 0;
#pragma unroll
                    for (int q = 0; q < M2_CAP; ++q)
                        if (ej[q] >= 0 && ej[q] < ej[k]) ++rank;
                    mine[rank] = (static_cast<unsigned long long>(static_cast<unsigned>(ej[k])) << 32) | static_cast<unsigned>(ew[k]);
                }
            }
            A.row_cnt[G.row_base + i] = static_cast<uint16_t>(cnt);
        }
    }
}
#undef M2_ADD1
#undef M2_MATCH1
#undef M2_APPEND1
#undef M2_ADD

// maximum of a 64-bit value over the 16 lanes of a DPP row
__device__ __forceinline__ unsigned long long m2_rowmax16(unsigned long long v) {
    // maximum over the 16 lanes of a DPP row, delivered to every lane of the row (row_ror 8, 4, 2, 1)
#define M2_STEP(CTRL)                                                                                                         \
    {                                                                                                                         \
        const unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);                                    \
        const unsigned olo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), CTRL, 0xf, 0xf, false)); \
        const unsigned ohi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), CTRL, 0xf, 0xf, false)); \
        const unsigned long long o = (static_cast<unsigned long long>(ohi) << 32) | olo;                                      \
        v = o > v ? o : v;                                                                                                    \
    }
    M2_STEP(0x128) M2_STEP(0x124) M2_STEP(0x122) M2_STEP(0x121)
#undef M2_STEP
    return v;
}

// ---- heaviest chain of one merge round: four groups per wavefront, 16 lanes each ----
// Node value = (f << 32) | ~id, so that the maximum prefers the larger f and then the earlier match
// (id = row * M2_CAP + index in the row + 1; 0 = no match).  All matches of a row are looked up before any of
// them is entered.
// Prefix maxima over the second child's columns are kept for the M2_PWIN = 512 columns behind the front in three
// levels of PREFIX arrays: L0[c] = best over the columns of c's block of 8 up to c, L1[b] = best over the blocks of
// b's superblock (64 columns) up to b, L2[s] = best over everything up to superblock s (a new superblock starts
// with its predecessor's value); what lies behind the 8 superblocks ending at the front's is one scalar.  A match
// with column j reads THREE values -- L0[j - 1], L1[block - 1], L2[superblock - 1], where they exist -- and
// entering it raises the at most 8 + 8 + 8 entries from its column, block and superblock to the end of their
// block, superblock and the front (LDS maxima without return).  (A step function over single columns raised from
// j up to the front, which this replaces, degenerates when the front runs ahead of the heavy matches: clusters of
// two molecules.)  The 16 lanes of a group serve the 16 entries of one row.  When a match lies in a superblock that
// has left the window (448-511 columns behind the largest column seen up to and including its row), the window
// cannot answer: the group's round is flagged (redo) and done by k_m2_chain_exact, which keeps the prefix maxima
// of ALL columns in a Fenwick tree.
constexpr int M2_PWIN = 512;
constexpr int M2_L1 = M2_PWIN, M2_L2 = M2_PWIN + M2_PWIN / 8, M2_PSIZE = M2_PWIN + M2_PWIN / 8 + M2_PWIN / 64;

__device__ __forceinline__ int m2_qmax_i32(int v) {   // maximum over the 16 lanes of a DPP row, in every lane
    // one v_max_i32_dpp per rotation (every lane of a rotation has a source, so there is no `old` operand to keep; written
    // as asm because the compiler emits a move + a DPP move + a maximum for the builtin).  s_nop 1: a DPP operand read
    // needs two wait states after the VALU write of that register.
#define M2_QSTEP(ROR) asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_ror:" #ROR " row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(v));
    M2_QSTEP(8) M2_QSTEP(4) M2_QSTEP(2) M2_QSTEP(1)
#undef M2_QSTEP
    return v;
}

__global__ void __launch_bounds__(64) k_m2_chain_q(M2Args A, int round, int nactive) {
    __shared__ unsigned long long s_buf[4][M2_PSIZE > 256 ? M2_PSIZE : 256];   // the three levels, later the traceback stage
    const int lane = threadIdx.x, qd = lane >> 4, t = lane & 15;
    const int g = A.g0 + blockIdx.x * 4 + qd;
    bool act = g < nactive;
    M2Group G{};
    if (act) G = A.groups[g];
    if (act && round >= G.n - 1) act = false;
    if (act && A.ovf[g] != 0) act = false;   // dropped (coherence guard, or a profile outgrew its capacity)
    int nA = 0, nB = 0;
    if (act) {
        const int2 jn = A.joins[G.first_member + round];
        nA = A.ncols[2 * G.first_member + jn.x];
        nB = A.ncols[2 * G.first_member + jn.y];
    }
    (void)nB;
    unsigned long long* const P = &s_buf[qd][0];
    for (int x = t; x < M2_PSIZE; x += 16) P[x] = 0;
    unsigned long long pbelow = 0;   // best over the superblocks that left the window
    int sbT = 0;                     // superblock of the front: the window holds superblocks sbT - 7 .. sbT
    unsigned long long* const ent = A.row_ent + G.row_base * static_cast<long long>(M2_CAP);
    unsigned* const prd = A.row_pred + G.row_base * static_cast<long long>(M2_CAP);
    for (int i = t; i < nA; i += 16) A.part[G.row_base + i] = -1;
    int top = -1;
    unsigned long long lbest = 0;    // best node entered by this lane (the chain's last match is the overall maximum)
    bool bad = false;
    const int nAmax = max(max(__shfl(nA, 0), __shfl(nA, 16)), max(__shfl(nA, 32), __shfl(nA, 48)));
    // rows in blocks of 16: lane t holds entry t and the count of each of the block's rows (the counts are
    // broadcast loads: all 16 lanes of a group read the same address)
    unsigned long long eb[16];
    int cb[16];
    auto load_block = [&](int ib, unsigned long long (&e)[16], int (&c)[16]) {
        // unconditional loads (a row index clamped into the group's rows): a select around each load would make
        // the compiler wait for every one of them in turn
        const int last = max(nA - 1, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = ent[static_cast<long long>(min(ib + r, last)) * M2_CAP + t];
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = static_cast<int>(A.row_cnt[G.row_base + min(ib + r, last)]);
    };
    load_block(0, eb, cb);
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
    for (int ib = 0; ib < nAmax; ib += 16) {
        unsigned long long en[16];
        int cn[16];
        load_block(ib + 16, en, cn);   // next block in flight while this one is processed
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ib + r;
            const bool mk0 = act && i < nA && t < cb[r];
            const unsigned long long e = eb[r];
            const int j = static_cast<int>(e >> 32);
            // the window follows the largest column seen up to and including this row
            const int newtop = m2_qmax_i32(mk0 ? j : -1);
            const int T = max(top, newtop);
            top = T;
            if ((T >> 6) > sbT) {
                // new superblocks start with the prefix of everything before them; the slots they take over held the
                // superblocks 8 further back, whose prefix becomes the scalar for "behind the window"
                const unsigned long long carried = P[M2_L2 + (sbT & 7)];
                const int adv = min((T >> 6) - sbT, 8);
                for (int q = 1; q <= adv; ++q) {
                    const int slot = ((T >> 6) - adv + q) & 7;
                    pbelow = P[M2_L2 + slot];
                    if (t == 0) P[M2_L2 + slot] = carried;
                    if (t < 8) P[M2_L1 + slot * 8 + t] = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) P[slot * 64 + u * 16 + t] = 0;
                }
                if ((T >> 6) - sbT >= 8) pbelow = carried;
                sbT = T >> 6;
            }
            if (mk0 && (j >> 6) < sbT - 7) bad = true;
            const bool mk = mk0 && !bad;
            // query: best over the columns < j, state before the row (unconditional reads, selected afterwards)
            const int jb = j >> 3, js = j >> 6;
            const unsigned long long r0 = P[(j - 1) & (M2_PWIN - 1)];
            const unsigned long long r1 = P[M2_L1 + ((jb - 1) & 63)];
            const unsigned long long r2 = P[M2_L2 + ((js - 1) & 7)];
            const unsigned long long q0 = (j & 7) ? r0 : 0ull;
            const unsigned long long q1 = (jb & 7) ? r1 : 0ull;
            const unsigned long long q2 = (js >= 1 && js - 1 >= sbT - 7) ? r2 : pbelow;
            const unsigned long long q01 = q0 > q1 ? q0 : q1;
            unsigned long long v = q01 > q2 ? q01 : q2;
            if (!mk) v = 0;
            const unsigned f = static_cast<unsigned>(e) + static_cast<unsigned>(v >> 32);
            const unsigned pred = v ? ~static_cast<unsigned>(v) : 0u;
            const unsigned id = static_cast<unsigned>(i) * M2_CAP + static_cast<unsigned>(t) + 1u;
            const unsigned long long nv = mk ? ((static_cast<unsigned long long>(f) << 32) | static_cast<unsigned>(~id)) : 0ull;
            if (mk) prd[static_cast<long long>(i) * M2_CAP + t] = pred;
            // enter the matches (all queries of the row were issued before: LDS operations of a wave execute in order)
            // One predicated block, no inner branches: positions past the end of the block are clamped onto its last
            // entry (entered more than once, harmless for a maximum).  Only lanes with a match take part -- maxima on one
            // LDS address serialise, and the stale entries of the other lanes would all name the same few slots.
            if (mk) {
                const int c0 = (j & ~7) & (M2_PWIN - 1), b0 = (jb & ~7) & 63;
                const int cj = j & 7, cb8 = jb & 7;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    atomicMax(&P[c0 + min(cj + d, 7)], nv);
                    atomicMax(&P[M2_L1 + b0 + min(cb8 + d, 7)], nv);
                }
                for (int sb = js; sb <= sbT; ++sb) atomicMax(&P[M2_L2 + (sb & 7)], nv);
            }
            lbest = nv > lbest ? nv : lbest;
        }
        bad = m2_qmax_i32(bad ? 1 : 0) != 0;   // (a failed group keeps going on garbage until here; its round is redone)
#pragma unroll
        for (int r = 0; r < 16; ++r) { eb[r] = en[r]; cb[r] = cn[r]; }
    }
    unsigned long long best = m2_rowmax16(lbest);
    if (act && bad && t == 0) A.redo[g] = 1;
    if (bad) { nA = 0; best = 0; }
    if (!act) { nA = 0; best = 0; }
    // traceback through the stored predecessors, 16 rows staged at a time (the P window is free now)
    __threadfence();
    unsigned long long* const stage = &s_buf[qd][0];   // [16 rows][16 entries]
    const unsigned long long clk1 = __builtin_amdgcn_s_memtime();
    unsigned id = best ? ~static_cast<unsigned>(best) : 0u;
    int steps = 2 * nA + 64;
    while (__builtin_amdgcn_ballot_w64(id != 0 && steps > 0)) {
        const bool go = id != 0 && steps > 0;
        const int itop = go ? static_cast<int>((id - 1u) / M2_CAP) : 0;
        const int i0 = max(0, itop - 15);
        {
            // all loads first, then the LDS writes (interleaved, every load would wait for the write before it)
            unsigned long long ev[16];
            unsigned pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // (rows above itop are never followed: clamped, not masked)
                ev[r] = ent[static_cast<long long>(min(i0 + r, itop)) * M2_CAP + t];
                pv[r] = prd[static_cast<long long>(min(i0 + r, itop)) * M2_CAP + t];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[r * 16 + t] = (ev[r] & 0xffffffff00000000ull) | pv[r];
        }
        --steps;
        while (__builtin_amdgcn_ballot_w64(go && id != 0 && steps > 0 && static_cast<int>((id - 1u) / M2_CAP) >= i0)) {
            const bool in = go && id != 0 && steps > 0 && static_cast<int>((id - 1u) / M2_CAP) >= i0;
            if (in) {
                const int i = static_cast<int>((id - 1u) / M2_CAP), k = static_cast<int>((id - 1u) % M2_CAP);
                const unsigned long long e = stage[(i - i0) * 16 + k];
                if (t == 0) A.part[G.row_base + i] = static_cast<int>(e >> 32);
                id = static_cast<unsigned>(e);
                --steps;
            }
        }
    }
    if (A.clk && blockIdx.x == 0 && lane == 0) {
        const unsigned long long clk2 = __builtin_amdgcn_s_memtime();
        A.clk[0] = clk1 - clk0; A.clk[1] = clk2 - clk1; A.clk[2] = static_cast<unsigned long long>(nAmax);
    }
}

// ---- heaviest chain, exact for any pattern of matches: one wavefront per flagged group ----
// Fenwick tree over the second child's columns: a match with column j reads the prefix maximum of nodes j,
// j - lowbit(j), .. (columns < j) and afterwards raises nodes j + 1, (j + 1) + lowbit, ..; sub-groups of 16 lanes
// serve one match each, one node per lane.  All matches of a row are looked up before any of them is entered.
// GBIT: the tree lives in HBM (wcap + 1 nodes per group at row_base + g) -- profiles too wide for LDS.
template <bool GBIT>
__global__ void __launch_bounds__(64) k_m2_chain_exact(M2Args A, int round, unsigned long long* gbit) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int g = A.g0 + blockIdx.x;
    if (!A.redo[g]) return;
    const M2Group G = A.groups[g];
    const int fm = G.first_member;
    const int2 jn = A.joins[fm + round];
    const int nA = A.ncols[2 * fm + jn.x], nB = A.ncols[2 * fm + jn.y];
    const int lane = threadIdx.x;
    const unsigned long long xclk0 = __builtin_amdgcn_s_memtime();
    unsigned long long xent = 0;
    unsigned long long* const s_stage = reinterpret_cast<unsigned long long*>(smem);   // [64 rows][M2_CAP]
    unsigned long long* const s_nv = s_stage + 64 * M2_CAP;                              // [M2_CAP]
    int* const s_j = reinterpret_cast<int*>(s_nv + M2_CAP);                             // [M2_CAP]
    int* const s_cnt = s_j + M2_CAP;                                                    // [64]
    unsigned long long* const bit = GBIT ? gbit + G.row_base + g : reinterpret_cast<unsigned long long*>(s_cnt + 64);   // [nB + 1]
    for (int x = lane; x <= nB; x += 64) bit[x] = 0;
    if (GBIT) __threadfence();
    for (int i = lane; i < nA; i += 64) A.part[G.row_base + i] = -1;
    const unsigned long long* const ent = A.row_ent + G.row_base * static_cast<long long>(M2_CAP);
    unsigned* const prd = A.row_pred + G.row_base * static_cast<long long>(M2_CAP);
    const int sub = lane >> 4, t = lane & 15;
    const unsigned nBu = static_cast<unsigned>(nB);
    const int last = max(nA - 1, 0);
    unsigned long long best = 0;
    auto bit_load = [&](unsigned x) -> unsigned long long {
        if (GBIT) return __hip_atomic_load(&bit[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the atomics act in L2)
        return bit[x];
    };
    for (int i0 = 0; i0 < nA; i0 += 64) {
        __syncthreads();
        {   // 64 rows: lane = row; unconditional loads (clamped row index)
            const int rowi = min(i0 + lane, last);
            unsigned long long ev[M2_CAP];
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) ev[k] = ent[static_cast<long long>(rowi) * M2_CAP + k];
            const int c = static_cast<int>(A.row_cnt[G.row_base + rowi]);
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) s_stage[lane * M2_CAP + k] = ev[k];
            s_cnt[lane] = (i0 + lane < nA) ? c : 0;
        }
        __syncthreads();
        const int rows = min(64, nA - i0);
        for (int r = 0; r < rows; ++r) {
            const int c = min(__builtin_amdgcn_readfirstlane(s_cnt[r]), M2_CAP);
            if (c == 0) continue;
            xent += static_cast<unsigned>(c);
            const int i = i0 + r;
            for (int k0 = 0; k0 < c; k0 += 4) {          // queries
                const int k = k0 + sub;
                const bool act = k < c;
                const unsigned long long e = act ? s_stage[r * M2_CAP + k] : 0ull;
                const int j = static_cast<int>(e >> 32);
                unsigned x = act ? static_cast<unsigned>(j) : 0u;
#pragma unroll
                for (int q = 0; q < 15; ++q) x = (q < t) ? (x & (x - 1u)) : x;
                unsigned long long v = x ? bit_load(x) : 0ull;
                v = m2_rowmax16(v);
                const unsigned f = static_cast<unsigned>(e) + static_cast<unsigned>(v >> 32);
                const unsigned pred = v ? ~static_cast<unsigned>(v) : 0u;
                const unsigned id = static_cast<unsigned>(i) * M2_CAP + static_cast<unsigned>(k) + 1u;
                const unsigned long long nv = (static_cast<unsigned long long>(f) << 32) | static_cast<unsigned>(~id);
                if (act) {
                    best = nv > best ? nv : best;
                    if (t == 0) { prd[static_cast<long long>(i) * M2_CAP + k] = pred; s_j[k] = j; s_nv[k] = nv; }
                }
            }
            __syncthreads();
            for (int k0 = 0; k0 < c; k0 += 4) {          // updates
                const int k = k0 + sub;
                if (k < c) {
                    const unsigned long long nv = s_nv[k];
                    unsigned y = static_cast<unsigned>(s_j[k]) + 1u;
#pragma unroll
                    for (int q = 0; q < 15; ++q) y = (q < t && y <= nBu) ? y + (y & (0u - y)) : y;
                    if (y <= nBu) atomicMax(&bit[y], nv);
                }
            }
            __syncthreads();
        }
    }
    {   // the chain's last match: maximum over the four sub-groups
        const unsigned long long o1 = (static_cast<unsigned long long>(static_cast<unsigned>(__shfl_xor(static_cast<int>(best >> 32), 16))) << 32) |
                                      static_cast<unsigned>(__shfl_xor(static_cast<int>(best), 16));
        best = o1 > best ? o1 : best;
        const unsigned long long o2 = (static_cast<unsigned long long>(static_cast<unsigned>(__shfl_xor(static_cast<int>(best >> 32), 32))) << 32) |
                                      static_cast<unsigned>(__shfl_xor(static_cast<int>(best), 32));
        best = o2 > best ? o2 : best;
    }
    // traceback through the stored predecessors, 64 rows staged at a time
    __threadfence();
    unsigned id = best ? ~static_cast<unsigned>(best) : 0u;
    id = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(id)));
    int steps = 2 * nA + 64;   // a chain has at most one match per row; staging a block also counts one step
    while (id && --steps >= 0) {
        const int itop = static_cast<int>((id - 1u) / M2_CAP);
        const int i0 = max(0, itop - 63);
        __syncthreads();
        {
            const int rowi = min(i0 + lane, itop);
            unsigned long long ev[M2_CAP];
            unsigned pv[M2_CAP];
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) { ev[k] = ent[static_cast<long long>(rowi) * M2_CAP + k]; pv[k] = prd[static_cast<long long>(rowi) * M2_CAP + k]; }
#pragma unroll
            for (int k = 0; k < M2_CAP; ++k) s_stage[lane * M2_CAP + k] = (ev[k] & 0xffffffff00000000ull) | pv[k];
        }
        __syncthreads();
        while (id && --steps >= 0) {
            const int i = static_cast<int>((id - 1u) / M2_CAP), k = static_cast<int>((id - 1u) % M2_CAP);
            if (i < i0) break;
            const unsigned long long e = s_stage[(i - i0) * M2_CAP + k];
            if (lane == 0) A.part[G.row_base + i] = static_cast<int>(e >> 32);
            id = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(e))));
        }
    }
    if (lane == 0) A.redo[g] = 0;
    if (A.xdbg && lane == 0) { A.xdbg[2 * g] = __builtin_amdgcn_s_memtime() - xclk0; A.xdbg[2 * g + 1] = (static_cast<unsigned long long>(nA) << 32) | xent; }
}

// ---- wave helpers for the renumbering (performance is irrelevant here) ----
__device__ __forceinline__ int m2_excl_max(int v, int identity) {   // exclusive prefix maximum over the lanes
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(x, d);
        if (static_cast<int>(threadIdx.x & 63) >= d) x = max(x, o);
    }
    const int e = __shfl_up(x, 1);
    return (threadIdx.x & 63) == 0 ? identity : e;
}
__device__ __forceinline__ int m2_suffix_min_incl(int v) {          // inclusive suffix minimum over the lanes
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(x, d);
        if (static_cast<int>(threadIdx.x & 63) + d < 64) x = min(x, o);
    }
    return x;
}

// ---- new column numbers, col / pos of every member: one workgroup of 256 threads per group ----
__global__ void __launch_bounds__(256) k_m2_merge(M2Args A, int round, int* ncA, int* ncB, int* partB) {
    const int g = A.g0 + blockIdx.x;
    const M2Group G = A.groups[g];
    if (round >= G.n - 1 || A.ovf[g] != 0) return;
    const int n = G.n, fm = G.first_member;
    const int2 jn = A.joins[fm + round];
    const unsigned maskA = A.nodemask[2 * fm + jn.x], maskB = A.nodemask[2 * fm + jn.y];
    const int nA = A.ncols[2 * fm + jn.x], nB = A.ncols[2 * fm + jn.y];
    const int* part = A.part + G.row_base;
    int* const nca = ncA + G.row_base;
    int* const ncb = ncB + G.row_base;
    int* const pb = partB + G.row_base;
    __shared__ int s_newW;
    const int lane = threadIdx.x & 63;
    // partner rows of the second child's columns
    for (int j = threadIdx.x; j < nB; j += blockDim.x) pb[j] = -1;
    __syncthreads();
    for (int i = threadIdx.x; i < nA; i += blockDim.x) {
        const int pj = part[i];
        if (pj >= 0) pb[pj] = i;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) {
        // first child: column i -> i + (columns of the second child up to the previous matched partner) - matches before
        int jprev = -1, t0 = 0;
        for (int i0 = 0; i0 < nA; i0 += 64) {
            const int i = i0 + lane;
            const int pj = i < nA ? part[i] : -1;
            const unsigned long long ball = __ballot(pj >= 0);
            const int before = __popcll(ball & ((1ull << lane) - 1ull));
            const int pm = max(m2_excl_max(pj, -1), jprev);
            const int t = t0 + before;
            if (i < nA) nca[i] = pj >= 0 ? i + pj - t : i + (pm + 1) - t;
            t0 += __popcll(ball);
            int mx = pj;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
            jprev = max(jprev, mx);
        }
        const int nm = t0;
        // second child: column j -> (row of the next matched pair, or nA) + j - matches before; descending for "next"
        int inext = nA;
        for (int j0 = ((nB - 1) / 64) * 64; j0 >= 0 && nB > 0; j0 -= 64) {
            const int j = j0 + lane;
            const int pi = j < nB ? pb[j] : -1;
            int nx = m2_suffix_min_incl(pi >= 0 ? pi : 0x7fffffff);
            nx = min(nx, inext);
            if (j < nB) ncb[j] = nx;          // provisional: the row of the next matched pair at or after j
            int mn = pi >= 0 ? pi : 0x7fffffff;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mn = min(mn, __shfl_xor(mn, d));
            inext = min(inext, mn);
        }
        int tb = 0;
        for (int j0 = 0; j0 < nB; j0 += 64) {
            const int j = j0 + lane;
            const int pi = j < nB ? pb[j] : -1;
            const unsigned long long ball = __ballot(pi >= 0);
            const int before = tb + __popcll(ball & ((1ull << lane) - 1ull));
            if (j < nB) ncb[j] = ncb[j] + j - before;   // matched: next = its own row
            tb += __popcll(ball);
        }
        if (lane == 0) s_newW = nA + nB - nm;
    }
    __threadfence_block();
    __syncthreads();
    const int newW = s_newW;
    if (newW > G.wcap) {
        if (threadIdx.x == 0) { A.ovf[g] = 1; A.nodemask[2 * fm + n + round] = maskA | maskB; A.ncols[2 * fm + n + round] = G.wcap; }
        return;
    }
    // clear the members' rows of pos, then scatter the new columns
    for (int a = 0; a < n; ++a) {
        if (!(((maskA | maskB) >> a) & 1u)) continue;
        uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
        for (int c = threadIdx.x; c < newW; c += blockDim.x) row[c] = static_cast<uint16_t>(M2_NONE);
    }
    __threadfence_block();
    __syncthreads();
    for (int a = 0; a < n; ++a) {
        const bool inA = (maskA >> a) & 1u, inB = (maskB >> a) & 1u;
        if (!inA && !inB) continue;
        const M2Member Me = A.members[fm + a];
        const int* nc = inA ? nca : ncb;
        uint16_t* row = A.pos + G.pos_base + static_cast<long long>(a) * G.wcap;
        for (int p = threadIdx.x; p < Me.len; p += blockDim.x) {
            const int c = nc[A.col[Me.col_base + p]];
            A.col[Me.col_base + p] = c;
            row[c] = static_cast<uint16_t>(p);
        }
    }
    if (threadIdx.x == 0) { A.nodemask[2 * fm + n + round] = maskA | maskB; A.ncols[2 * fm + n + round] = newW; }
}

__global__ void k_m2_width(M2Args A) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.ngroups) return;
    const M2Group G = A.groups[g];
    int w = 0;
    if (G.n == 1) w = A.members[G.first_member].len;
    else if (G.n >= 2) w = A.ncols[2 * G.first_member + 2 * G.n - 2];
    A.width[g] = w;
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

static int g_msa_spec = 0;   // 0: not set (SARLACC_MSA_SPEC or 2)

static int msa_spec() {
    if (g_msa_spec) return g_msa_spec;
    if (const char* e = std::getenv("SARLACC_MSA_SPEC")) return std::atoi(e) == 1 ? 1 : 2;
    return 2;
}

static inline unsigned m2_blocks(long long n, int bs) { return static_cast<unsigned>((n + bs - 1) / bs); }

// One batch of groups (`ids`: indices into the caller's group list) through the v2 kernels.
// exact = false: fast capacities (lists of M2_CAP entries in LDS, profiles of 2 maxlen + 64 columns); groups
// that outgrow them come back in `redo`.  exact = true: worst-case capacities, nothing can overflow.
// Leaves the batch's state (pos, members ...) in the "m2*" workspaces with prefix `pf` for the row writer.
struct M2Batch {
    std::vector<int64_t> ids;      // caller's group indices, by decreasing group size
    std::vector<size_t> slot;      // position of each in the caller's order (msa2_core)
    std::vector<M2Group> groups;
    std::vector<M2Member> members;
    std::vector<int> member_group;
    std::vector<MsaJob> jobs;
    std::vector<int32_t> width;   // per group of the batch
    std::vector<int> ovf;
    M2Args a{};                   // device pointers
    int* d_member_group = nullptr;
    int max_len = 0, max_wcap = 0, max_n = 0;
};

static int m2_plan(M2Batch& B, const int64_t* grp_off, const int32_t* grp, const int64_t* rel, bool exact_w) {
    long long map_pos = 0, col_pos = 0, row_pos = 0, pos_pos = 0, dist_pos = 0;
    B.groups.clear(); B.members.clear(); B.member_group.clear(); B.jobs.clear();
    B.max_len = 0; B.max_wcap = 0; B.max_n = 0;
    for (size_t q = 0; q < B.ids.size(); ++q) {
        const int64_t g = B.ids[q];
        const int32_t* mem = grp + grp_off[g];
        const int n = static_cast<int>(grp_off[g + 1] - grp_off[g]);
        M2Group G{};
        G.first_member = static_cast<int>(B.members.size());
        G.n = n;
        long long sum = 0;
        int mx = 0;
        for (int a = 0; a < n; ++a) {
            const int len = static_cast<int>(rel[mem[a]] - rel[mem[a] - 1]);
            sum += len;
            mx = std::max(mx, len);
        }
        const long long fast_w = M2_FASTW(mx);
        // (65535 columns is the ceiling of spec v2: positions and the 16-level Fenwick tree; only reachable when the
        // sum of the read lengths exceeds it AND the alignment really is that wide)
        G.wcap = static_cast<int>(std::min<long long>(65535, std::min<long long>(sum, exact_w ? sum : fast_w)));
        if (G.wcap < 1) G.wcap = 1;
        G.cap = M2_CAP;
        G.row_base = row_pos;
        G.pos_base = pos_pos;
        G.first_job = static_cast<long long>(B.jobs.size());
        G.dist_base = dist_pos;
        row_pos += G.wcap;
        pos_pos += static_cast<long long>(n) * G.wcap;
        dist_pos += static_cast<long long>(n) * n + n;
        for (int a = 0; a < n; ++a) {
            M2Member Me{};
            Me.seq_off = rel[mem[a] - 1];
            Me.len = static_cast<int>(rel[mem[a]] - rel[mem[a] - 1]);
            Me.map_base = map_pos;
            Me.col_base = col_pos;
            map_pos += static_cast<long long>(std::max(0, n - 1)) * Me.len;
            col_pos += Me.len;
            B.members.push_back(Me);
            B.member_group.push_back(static_cast<int>(q));
        }
        if (n >= 2)
            for (int a = 0; a < n; ++a)
                for (int b = a + 1; b < n; ++b) {
                    const M2Member& Ma = B.members[G.first_member + a];
                    const M2Member& Mb = B.members[G.first_member + b];
                    MsaJob J{};
                    J.read_off = Mb.seq_off; J.ctr_off = Ma.seq_off;   // rows = b, columns = a
                    J.lr = Mb.len; J.lc = Ma.len;
                    J.out_off = Ma.map_base + static_cast<long long>(b - 1) * Ma.len;    // b among the others of a
                    J.out2_off = Mb.map_base + static_cast<long long>(a) * Mb.len;       // a among the others of b
                    B.jobs.push_back(J);
                }
        B.max_len = std::max(B.max_len, mx);
        B.max_wcap = std::max(B.max_wcap, G.wcap);
        B.max_n = std::max(B.max_n, n);
        B.groups.push_back(G);
    }
    return 0;
}

// streams of the round loop (created once per process, non-blocking: the caller's stream may be the legacy default one)
struct M2Streams {
    std::vector<hipStream_t> st;
    std::vector<hipEvent_t> join;
    hipEvent_t fork = nullptr;
    int device = -1;
    int ensure(int n) {
        if (device != ctx().device) {   // (streams and events belong to the device that was current when they were made)
            for (hipStream_t x : st) (void)hipStreamDestroy(x);
            for (hipEvent_t e : join) (void)hipEventDestroy(e);
            if (fork) (void)hipEventDestroy(fork);
            st.clear(); join.clear(); fork = nullptr;
            device = ctx().device;
        }
        if (!fork) SL_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
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

static int m2_run_batch(M2Batch& B, const std::string& pf, const uint8_t* d_seq, double match, double mismatch, double gap_extension,
                        double gap_opening, int bandwidth, bool exact, const std::function<int()>* overlap, double* cells, hipStream_t s) {
    Context& c = ctx();
    const size_t ng = B.groups.size(), nm = B.members.size();
    if (ng == 0) return 0;
    long long map_n = 0, col_n = 0, row_n = 0, pos_n = 0, dist_n = 0, ent_n = 0;
    for (const M2Member& Me : B.members) col_n += Me.len;
    for (const M2Group& G : B.groups) {
        row_n += G.wcap; pos_n += static_cast<long long>(G.n) * G.wcap; dist_n += static_cast<long long>(G.n) * G.n + G.n;
    }
    if (!B.members.empty()) { const M2Member& L = B.members.back(); const M2Group& G = B.groups[B.member_group.back()]; map_n = L.map_base + static_cast<long long>(std::max(0, G.n - 1)) * L.len; }
    ent_n = row_n * static_cast<long long>(M2_CAP);
    M2Args& a = B.a;
    a = M2Args{};
    M2Group* d_groups; M2Member* d_members; MsaJob* d_jobs; int* d_mg;
    SL_TRY(upload((pf + ".groups").c_str(), B.groups.data(), ng, &d_groups, s));
    SL_TRY(upload((pf + ".members").c_str(), B.members.data(), nm, &d_members, s));
    SL_TRY(upload((pf + ".mg").c_str(), B.member_group.data(), nm, &d_mg, s));
    SL_TRY(upload((pf + ".jobs").c_str(), B.jobs.data(), B.jobs.size(), &d_jobs, s));
    B.d_member_group = d_mg;
    uint16_t* d_map; int2* d_stats; double* d_dist; int2* d_joins; uint32_t* d_mask; int* d_ncols; int* d_col; uint16_t* d_pos;
    uint16_t* d_cnt; unsigned long long* d_ent; int* d_part; int* d_nca; int* d_ncb; int* d_pb; int* d_ovf; int32_t* d_width;
    SL_TRY(scratch((pf + ".map").c_str(), static_cast<size_t>(map_n) + 1, &d_map));
    SL_TRY(scratch((pf + ".stats").c_str(), B.jobs.size() + 1, &d_stats));
    SL_TRY(scratch((pf + ".dist").c_str(), static_cast<size_t>(dist_n) + 1, &d_dist));
    SL_TRY(scratch((pf + ".joins").c_str(), nm + 1, &d_joins));
    SL_TRY(scratch((pf + ".mask").c_str(), 2 * nm + 2, &d_mask));
    SL_TRY(scratch((pf + ".ncols").c_str(), 2 * nm + 2, &d_ncols));
    SL_TRY(scratch((pf + ".col").c_str(), static_cast<size_t>(col_n) + 1, &d_col));
    SL_TRY(scratch((pf + ".pos").c_str(), static_cast<size_t>(pos_n) + 1, &d_pos));
    SL_TRY(scratch((pf + ".cnt").c_str(), static_cast<size_t>(row_n) + 1, &d_cnt));
    SL_TRY(scratch((pf + ".ent").c_str(), static_cast<size_t>(ent_n) + 1, &d_ent));
    unsigned* d_pred;
    SL_TRY(scratch((pf + ".pred").c_str(), static_cast<size_t>(ent_n) + 1, &d_pred));
    SL_TRY(scratch((pf + ".part").c_str(), static_cast<size_t>(row_n) + 1, &d_part));
    SL_TRY(scratch((pf + ".nca").c_str(), static_cast<size_t>(row_n) + 1, &d_nca));
    SL_TRY(scratch((pf + ".ncb").c_str(), static_cast<size_t>(row_n) + 1, &d_ncb));
    SL_TRY(scratch((pf + ".pb").c_str(), static_cast<size_t>(row_n) + 1, &d_pb));
    SL_TRY(scratch((pf + ".ovf").c_str(), ng, &d_ovf));
    int* d_redo;
    SL_TRY(scratch((pf + ".redo").c_str(), ng, &d_redo));
    SL_HIP(hipMemsetAsync(d_redo, 0, sizeof(int) * ng, s));
    SL_TRY(scratch((pf + ".width").c_str(), ng, &d_width));
    SL_HIP(hipMemsetAsync(d_ovf, 0, sizeof(int) * ng, s));
    a.seq = d_seq; a.groups = d_groups; a.members = d_members; a.ngroups = static_cast<int>(ng);
    a.ma = static_cast<int>(match); a.mm = static_cast<int>(mismatch);
    unsigned long long* d_clk = nullptr;
    if (std::getenv("SARLACC_MSA2_CLOCKS")) { SL_TRY(scratch((pf + ".clk").c_str(), 8, &d_clk)); SL_HIP(hipMemsetAsync(d_clk, 0, 64, s)); }
    a.clk = d_clk;
    if (std::getenv("SARLACC_MSA2_EXACTDBG")) { SL_TRY(scratch((pf + ".xdbg").c_str(), 2 * ng, &a.xdbg)); SL_HIP(hipMemsetAsync(a.xdbg, 0, sizeof(unsigned long long) * 2 * ng, s)); }
    a.map = d_map; a.stats = d_stats; a.dist = d_dist; a.joins = d_joins; a.nodemask = d_mask; a.ncols = d_ncols;
    a.col = d_col; a.pos = d_pos; a.row_cnt = d_cnt; a.row_ent = d_ent; a.row_pred = d_pred; a.part = d_part; a.ovf = d_ovf; a.redo = d_redo; a.width = d_width;

    // ---- all pairs ----
    for (const MsaJob& J : B.jobs) *cells += static_cast<double>(J.lr) * msa_pair_band(bandwidth, J.lr, J.lc);
    SL_TRY(c.stage_begin("msa_pairwise", s));
    SL_TRY(msa_pairwise_launch(B.jobs, d_jobs, d_seq, match, mismatch, gap_extension, gap_opening, bandwidth, 1, nullptr, nullptr,
                               d_map, d_stats, s));
    SL_TRY(c.stage_end("msa_pairwise", s));
    if (overlap) SL_TRY((*overlap)());
    // ---- guide trees, leaves ----
    SL_TRY(c.stage_begin("msa_merge", s));
    hipLaunchKernelGGL(k_m2_tree, dim3(m2_blocks(static_cast<long long>(ng), 64)), dim3(64), 0, s, a);
    if (nm) hipLaunchKernelGGL(k_m2_init, dim3(std::max(1u, m2_blocks(B.max_len, 256)), static_cast<unsigned>(nm)), dim3(256), 0, s, a, d_mg, static_cast<int>(nm));
    SL_HIP(hipGetLastError());
    // ---- progressive merging, one round per join ----
    // The groups of a batch are ordered by size (m2_plan), so the groups that still have a join to do in round r
    // are a prefix of the batch.
    const bool unitw = a.ma <= 1 && a.mm <= 1;
    // unit weights: per-round candidate descriptors (k_m2_candidates), < n^2 entries of two int4 per group
    const bool old_gather = std::getenv("SARLACC_MSA2_OLDGATHER") != nullptr;
    M2Cand* d_tab = nullptr; long long* d_toff = nullptr; uint16_t* d_ident = nullptr;
    if (unitw && !old_gather) {
        std::vector<long long> toff(ng + 1, 0);
        for (size_t q = 0; q < ng; ++q) toff[q + 1] = toff[q] + static_cast<long long>(B.groups[q].n) * B.groups[q].n + M2_UBATCH;   // (n - 1)(n + 1) entries + padding
        SL_TRY(upload((pf + ".toff").c_str(), toff.data(), toff.size(), &d_toff, s));
        SL_TRY(scratch((pf + ".tab").c_str(), static_cast<size_t>(toff[ng]) + 1, &d_tab));
        SL_TRY(scratch("msa2.ident", 65536, &d_ident));
        hipLaunchKernelGGL(k_m2_identity, dim3(256), dim3(256), 0, s, d_ident);
    }
    // exact chain kernel: Fenwick tree in LDS while it fits 64 KB, in HBM for wider profiles
    const size_t exact_base = sizeof(unsigned long long) * (64 * M2_CAP + M2_CAP) + sizeof(int) * (M2_CAP + 64);
    const bool exact_gbit = exact_base + sizeof(unsigned long long) * (static_cast<size_t>(B.max_wcap) + 2) > 64 * 1024;
    const size_t exact_lds = exact_gbit ? exact_base : exact_base + sizeof(unsigned long long) * (static_cast<size_t>(B.max_wcap) + 2);
    unsigned long long* d_gbit = nullptr;
    if (exact_gbit) SL_TRY(scratch((pf + ".gbit").c_str(), static_cast<size_t>(row_n) + ng + 1, &d_gbit));
    else if (exact_lds > 48 * 1024)
        SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_m2_chain_exact<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(exact_lds)));
    // The groups of a batch are cut into NS contiguous ranges, each with a stream of its own: the chain kernels are
    // serial per group (few, long wavefronts), the gather is wide, so a range's chain runs under another range's
    // gather.  A group stays in its range for all rounds (its kernels stay in order on one stream).
    int NS = 2;   // (measured at C4: 1 stream 1.04 s, 2 streams 0.90 s, 3 and 4 streams 1.11-1.12 s -- beyond the hardware queues a process gets)
    size_t min_groups = 64;   // below that a batch is not worth two queues
    if (const char* e = std::getenv("SARLACC_MSA2_STREAMS")) { NS = std::min(8, std::max(1, std::atoi(e))); min_groups = 8; }   // (testing: small batches too)
    if (a.xdbg || std::getenv("SARLACC_MSA2_DEBUG") || a.clk) NS = 1;
    if (ng < min_groups) NS = 1;
    const bool stagger = !std::getenv("SARLACC_MSA2_NOSTAGGER");
    M2Streams& MS = m2_streams();
    if (NS > 1) SL_TRY(MS.ensure(NS));
    std::vector<int> cut(static_cast<size_t>(NS) + 1, 0);
    for (int k = 0; k <= NS; ++k) cut[k] = static_cast<int>((static_cast<long long>(ng) * k / NS) / 4 * 4);
    cut[NS] = static_cast<int>(ng);
    if (NS > 1) {
        SL_HIP(hipEventRecord(MS.fork, s));
        for (int k = 0; k < NS; ++k) SL_HIP(hipStreamWaitEvent(MS.st[k], MS.fork, 0));
    }
    for (int round = 0; round + 1 < B.max_n; ++round) {
        int nactive = 0;
        while (nactive < static_cast<int>(ng) && B.groups[nactive].n - 1 > round) ++nactive;
        if (nactive == 0) break;
        for (int k = 0; k < NS; ++k) {
            const int lo = cut[k], hi = std::min(cut[k + 1], nactive);
            if (lo >= hi) continue;
            const unsigned cnt = static_cast<unsigned>(hi - lo);
            hipStream_t sk = NS > 1 ? MS.st[k] : s;
            M2Args ak = a;
            ak.g0 = lo;
            const dim3 ggrid(std::min(128u, m2_blocks(B.max_wcap, 64)), cnt);
            // first round: a range starts when the one before it has finished its gather, so that from then on the
            // ranges are out of phase (their kernels have the same lengths: started together they would stay together)
            if (NS > 1 && round == 0 && k > 0 && stagger) SL_HIP(hipStreamWaitEvent(sk, MS.join[k - 1], 0));
            if (unitw && !old_gather) {
                hipLaunchKernelGGL(k_m2_candidates, dim3(cnt), dim3(64), 0, sk, ak, round, d_tab, d_toff, d_ident);
                hipLaunchKernelGGL(k_m2_gather_unit, ggrid, dim3(64), 0, sk, ak, round, d_tab, d_toff);
            } else if (unitw) hipLaunchKernelGGL(k_m2_gather<true>, ggrid, dim3(64), 0, sk, ak, round);
            else hipLaunchKernelGGL(k_m2_gather<false>, ggrid, dim3(64), 0, sk, ak, round);
            if (NS > 1 && round == 0 && stagger) SL_HIP(hipEventRecord(MS.join[k], sk));
            hipLaunchKernelGGL(k_m2_chain_q, dim3(m2_blocks(cnt, 4)), dim3(64), 0, sk, ak, round, hi);
            // rounds the window could not answer (unrelated reads in the cluster): exact chain search
            if (exact_gbit) hipLaunchKernelGGL(k_m2_chain_exact<true>, dim3(cnt), dim3(64), exact_lds, sk, ak, round, d_gbit);
            else hipLaunchKernelGGL(k_m2_chain_exact<false>, dim3(cnt), dim3(64), exact_lds, sk, ak, round, d_gbit);
            if (NS > 1) hipLaunchKernelGGL(k_m2_merge, dim3(cnt), dim3(256), 0, sk, ak, round, d_nca, d_ncb, d_pb);
        }
        SL_HIP(hipGetLastError());
        if (a.xdbg) {
            SL_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> hx(2 * ng);
            SL_HIP(hipMemcpy(hx.data(), a.xdbg, sizeof(unsigned long long) * hx.size(), hipMemcpyDeviceToHost));
            SL_HIP(hipMemsetAsync(a.xdbg, 0, sizeof(unsigned long long) * hx.size(), s));
            unsigned long long mx = 0, sum = 0, mxinfo = 0; int cntf = 0, mxg = -1;
            for (size_t q = 0; q < ng; ++q) if (hx[2 * q]) { ++cntf; sum += hx[2 * q]; if (hx[2 * q] > mx) { mx = hx[2 * q]; mxinfo = hx[2 * q + 1]; mxg = static_cast<int>(q); } }
            fprintf(stderr, "exact round %d: active %d flagged %d  cycles max %llu (group %d n=%d rows %llu entries %llu) sum %llu\n", round, nactive, cntf, mx,
                    mxg, mxg >= 0 ? B.groups[mxg].n : 0, mxinfo >> 32, mxinfo & 0xffffffffull, sum);
        }
        if (NS == 1) hipLaunchKernelGGL(k_m2_merge, dim3(static_cast<unsigned>(nactive)), dim3(256), 0, s, a, round, d_nca, d_ncb, d_pb);
        SL_HIP(hipGetLastError());
        if (std::getenv("SARLACC_MSA2_DEBUG")) {   // first group of the batch, for comparison with ORC_MSA2_DEBUG of the oracle
            SL_HIP(hipStreamSynchronize(s));
            const M2Group& G = B.groups[0];
            if (round < G.n - 1) {
                std::vector<int2> hj(G.n);
                std::vector<int> hn(2 * G.n), hp(G.wcap);
                std::vector<uint16_t> hc(G.wcap);
                std::vector<unsigned long long> he(static_cast<size_t>(G.wcap) * G.cap);
                SL_HIP(hipMemcpy(hj.data(), d_joins + G.first_member, sizeof(int2) * G.n, hipMemcpyDeviceToHost));
                SL_HIP(hipMemcpy(hn.data(), d_ncols + 2 * G.first_member, sizeof(int) * 2 * G.n, hipMemcpyDeviceToHost));
                SL_HIP(hipMemcpy(hp.data(), d_part + G.row_base, sizeof(int) * G.wcap, hipMemcpyDeviceToHost));
                SL_HIP(hipMemcpy(hc.data(), d_cnt + G.row_base, sizeof(uint16_t) * G.wcap, hipMemcpyDeviceToHost));
                SL_HIP(hipMemcpy(he.data(), d_ent + G.row_base * G.cap, sizeof(unsigned long long) * he.size(), hipMemcpyDeviceToHost));
                const int nA = hn[hj[round].x];
                fprintf(stderr, "GPU round %d: join %d %d nA %d nB %d -> %d\n", round, hj[round].x, hj[round].y, nA, hn[hj[round].y], hn[G.n + round]);
                for (int i = 0; i < nA; ++i) {
                    fprintf(stderr, "  row %d part %d :", i, hp[i]);
                    for (int k = 0; k < hc[i]; ++k) fprintf(stderr, " (%d w %u)", static_cast<int>(he[static_cast<size_t>(i) * G.cap + k] >> 32), static_cast<unsigned>(he[static_cast<size_t>(i) * G.cap + k]));
                    fprintf(stderr, "\n");
                }
            }
        }
    }
    if (NS > 1)
        for (int k = 0; k < NS; ++k) {
            SL_HIP(hipEventRecord(MS.join[k], MS.st[k]));
            SL_HIP(hipStreamWaitEvent(s, MS.join[k], 0));
        }
    if (d_clk) {
        unsigned long long hc[8];
        SL_HIP(hipMemcpy(hc, d_clk, sizeof hc, hipMemcpyDeviceToHost));
        fprintf(stderr, "gather kernel, block (0,0) of the last launch: %llu cycles, %llu candidate steps (x64 lanes), %llu rows in the profile, n = %llu\n", hc[4], hc[5], hc[6], hc[7]);
        fprintf(stderr, "chain kernel, first wave of the last LDS launch: forward %llu cycles, traceback %llu cycles, %llu rows\n", hc[0], hc[1], hc[2]);
    }
    hipLaunchKernelGGL(k_m2_width, dim3(m2_blocks(static_cast<long long>(ng), 256)), dim3(256), 0, s, a);
    SL_HIP(hipGetLastError());
    SL_TRY(c.stage_end("msa_merge", s));
    B.width.resize(ng);
    B.ovf.resize(ng);
    SL_HIP(hipMemcpyAsync(B.width.data(), d_width, sizeof(int32_t) * ng, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(B.ovf.data(), d_ovf, sizeof(int) * ng, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    int stuck = 0;
    if (!B.jobs.empty()) {
        int* d_stuck;
        SL_TRY(scratch("msa.stuck", 1, &d_stuck));
        SL_HIP(hipMemcpy(&stuck, d_stuck, sizeof stuck, hipMemcpyDeviceToHost));
        if (stuck) return fail("sarlacc_amd: internal error: an MSA traceback exceeded its step bound");
    }
    return 0;
}

// rows of the batch's groups (except those flagged in `skip`) at out + off[group of the caller's list]
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
    hipLaunchKernelGGL(k_m2_write, dim3(std::max(1u, m2_blocks(maxw, 256)), static_cast<unsigned>(B.members.size())), dim3(256), 0, s, B.a,
                       B.d_member_group, static_cast<int>(B.members.size()), d_off);
    SL_HIP(hipGetLastError());
    return 0;
}

// spec v2 on the groups `ids` of the caller's list; rows land in *d_rows at off[k] for ids[k] (processing order:
// the offsets are not cumulative in k), widths in width[k].
static int msa2_core(const int64_t* grp_off, const int32_t* grp, const std::vector<int64_t>& ids, const uint8_t* d_seq,
                     const std::vector<int64_t>& rel, double match, double mismatch, double gap_extension, double gap_opening,
                     int bandwidth, std::vector<int32_t>& width, std::vector<long long>& off, uint8_t** d_rows,
                     const std::function<int()>* overlap, const CodeSpec& code, hipStream_t s) {
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
        if (hipMalloc(&np, want) != hipSuccess) return fail("sarlacc_amd: cannot allocate %zu bytes of device memory for the alignment rows", want);
        if (rows_ws.ptr) {
            if (used) SL_HIP(hipMemcpyAsync(np, rows_ws.ptr, used, hipMemcpyDeviceToDevice, s));
            SL_HIP(hipStreamSynchronize(s));
            SL_HIP(hipFree(rows_ws.ptr));
        }
        rows_ws.ptr = np;
        rows_ws.cap = want;
        return 0;
    };
    const long long map_budget = 6LL << 30, ent_budget = 8LL << 30, job_budget = 3000000;
    double cells = 0, pairs = 0;
    bool first = true;
    long long used = 0;
    // Pass 0: every group with the fast profile capacity.  Pass 1: the groups whose profiles outgrew it (several
    // unrelated reads in one cluster), with profiles as wide as the sum of the read lengths.
    std::vector<size_t> todo(ids.size());
    std::iota(todo.begin(), todo.end(), size_t(0));
    for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
        const bool exact_w = pass == 1;
        // processing order: by decreasing group size, so that a batch holds groups with about the same number of joins
        std::stable_sort(todo.begin(), todo.end(), [&](size_t x, size_t y) {
            return grp_off[ids[x] + 1] - grp_off[ids[x]] > grp_off[ids[y] + 1] - grp_off[ids[y]];
        });
        std::vector<size_t> again;
        size_t q0 = 0;
        while (q0 < todo.size()) {
            M2Batch B;
            long long map_b = 0, ent_b = 0, jobs_b = 0;
            size_t q1 = q0;
            while (q1 < todo.size()) {
                const int64_t g = ids[todo[q1]];
                const long long n = grp_off[g + 1] - grp_off[g];
                long long sum = 0, mx = 0;
                for (long long a = 0; a < n; ++a) { const long long len = rel[grp[grp_off[g] + a]] - rel[grp[grp_off[g] + a] - 1]; sum += len; mx = std::max(mx, len); }
                const long long wc = exact_w ? sum : std::min(sum, M2_FASTW(mx));
                const long long mb = 2 * (n - 1) * sum, eb = wc * M2_CAP * 12, jb = n * (n - 1) / 2;
                if (q1 > q0 && (map_b + mb > map_budget || ent_b + eb > ent_budget || jobs_b + jb > job_budget)) break;
                map_b += mb; ent_b += eb; jobs_b += jb;
                B.ids.push_back(g);
                B.slot.push_back(todo[q1]);   // (already by decreasing size: the groups with a join left in round r are a prefix)
                ++q1;
            }
            SL_TRY(m2_plan(B, grp_off, grp, rel.data(), exact_w));
            SL_TRY(m2_run_batch(B, "m2", d_seq, match, mismatch, gap_extension, gap_opening, bandwidth, false, first ? overlap : nullptr, &cells, s));
            first = false;
            pairs += static_cast<double>(B.jobs.size());
            long long need = used;
            std::vector<long long> boff(B.groups.size(), 0);
            std::vector<int32_t> bw = B.width;
            for (size_t q = 0; q < B.groups.size(); ++q) {
                if (B.ovf[q]) {   // profile capacity exceeded: next pass
                    if (exact_w) return fail("sarlacc_amd: an alignment wider than 65535 columns is beyond spec v2 (select spec 1 with sarlacc_set_msa_spec)");
                    again.push_back(B.slot[q]);
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
            SL_TRY(m2_write_batch(B, "m2", boff, rows_ws.ptr, code, s));
            SL_HIP(hipStreamSynchronize(s));   // the batch's host vectors and workspaces are reused by the next one
            used = need;
            q0 = q1;
        }
        todo.swap(again);
    }
    c.counts["msa_pairs"] = (c.counts.count("msa_pairs") ? c.counts["msa_pairs"] : 0.0) + pairs;
    c.counts["msa_cells"] = (c.counts.count("msa_cells") ? c.counts["msa_cells"] : 0.0) + cells;
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
        if (n <= M2_MAXN && M2_FASTW(mx) <= 65535) v2.push_back(g); else v1.push_back(g);
    }
    SL_TRY(ensure_device());
    Context& c = ctx();
    hipStream_t s = nullptr;
    c.stage_reset("msa_pairwise");
    c.stage_reset("msa_merge");
    c.counts["msa_pairs"] = 0;
    c.counts["msa_cells"] = 0;
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
    SL_TRY(msa2_core(grp_off, grp, v2, d_seq, rel, match, mismatch, gap_extension, gap_opening, bandwidth, w2, o2, &d_rows2, overlap, res->code, s));
    c.counts["msa_v1_fallback"] = static_cast<double>(v1.size());
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
    return 0;
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" int sarlacc_set_msa_spec(int spec) {
    if (spec != 0 && spec != 1 && spec != 2) return fail("sarlacc_amd: MSA spec must be 1 (centre-star) or 2 (consistency-based progressive), 0 = default");
    sarlacc::g_msa_spec = spec;
    return 0;
}
