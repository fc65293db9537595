// umi.hip -- masked-Levenshtein neighbour search and greedy UMI clustering on gfx950.
//
// Replaces, for the umiGroup stage of the reference:
//   sorted_trie            /root/reference/src/sorted_trie.cpp:107-278 (thresholded search)
//   compute_lev_masked     src/compute_lev_masked.cpp:13-64            (dense distances)
//   cluster_umis           src/cluster_umis.cpp:7-112                  (greedy clustering)
//   umi_group              src/umi_group.cpp:14-116                    (per pre-group driver)
//
// Design (DESIGN.md "UMI stage"):
//   * The trie returns exactly { j : d2(i,j) <= 2*limit } listed in the order
//     A<C<G<T<N, shorter prefix first, ties by input index (SURVEY App.B Q11).  We
//     get the same lists from an all-pairs tile kernel over the UMIs sorted in that
//     order: 2-bit packed bases + N bit-mask per UMI, one thread per (row, tile),
//     columns broadcast from LDS, a composition / length lower bound rejecting most
//     pairs, then an exact banded DP in registers (band = limit, costs x2 as in
//     src/sorted_trie.cpp:13-21).  Only the upper triangle is evaluated; directed
//     adjacency is a radix sort of (row << 32 | column-rank) keys.
//   * The greedy clustering is sequential by definition; it is evaluated exactly,
//     in parallel rounds: a node whose (remaining, index) key is the maximum within
//     two hops cannot be affected by any earlier pick, so all such nodes are picked
//     in the same round.  The pick sequence of the reference is recovered by sorting
//     the picks by their key (descending), solos first (SURVEY section 0, App.B Q12).
#include <cstdlib>
#include <cstring>

#include "common.hpp"

#include <chrono>

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <limits>
#include <string>
#include <vector>

namespace sarlacc {

constexpr int UMI_MAXLEN = 32;        // one 64-bit word of 2-bit codes: the fast path (all filters, queued DP)
constexpr int UMI_LONG_WORDS = 4;     // strings of 33..128 bases: the same search on 4-word codes (k_umi_pairs_long)
constexpr int UMI_LONG_MAX = 32 * UMI_LONG_WORDS;
constexpr int UMI_XL_WORDS = 32;      // strings of 129..1024 bases: as many words as the longest string needs, read from HBM (k_umi_pairs_long<K, true>)
constexpr int UMI_XL_MAX = 32 * UMI_XL_WORDS;
// meta word of a string: length | number of N << 12 (12 bits each)
__host__ __device__ __forceinline__ int umi_len(uint32_t meta) { return static_cast<int>(meta & 0xfffu); }
__host__ __device__ __forceinline__ int umi_nn(uint32_t meta) { return static_cast<int>((meta >> 12) & 0xfffu); }
constexpr uint32_t UMI_META_NONE = 0xffffffu;   // a padding column: length 4095, never within any limit of a real string
constexpr int UMI_KEY_BASES = 21;     // bases per 64-bit sort key (3 bits each)
constexpr int UMI_LONG_KEYS = (UMI_LONG_MAX + UMI_KEY_BASES - 1) / UMI_KEY_BASES;
constexpr int TILE = 256;
constexpr int INF_D = 1 << 20;

// ---------------------------------------------------------------------------
// encoding

struct UmiArrays {
    unsigned long long* code;   // 2 bits per base (N stored as 0); word w of string s at code[w * stride + s]
    uint32_t* nmask;            // bit i set: base i is N; same layout
    uint32_t* comp;             // counts of A,C,G,T, one byte each
    uint32_t* meta;             // len | nN << 12 (umi_len, umi_nn)
    long long stride;           // strings per word plane (one plane on the fast path)
};

// members: optional 1-based ids selecting the strings of one pre-group.
__global__ void k_umi_encode(const uint8_t* chars, const int64_t* off, const int32_t* members, int n,
                             UmiArrays U, unsigned long long* key_hi, unsigned long long* key_lo, int* idx,
                             const uint8_t* skip /* optional: elements that are never compared (pre-groups of one read pass
                                                    through unchecked, src/umi_group.cpp:39-42): encoded as empty strings */,
                             int* bad /* [0]: min local index with unsupported char, [1]: min index too long,
                                         [2]: max(-length) of the too long ones, i.e. minus the longest */) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const long long id = members ? static_cast<long long>(members[s]) - 1 : s;
    const long long o = off[id];
    const int len = (skip && skip[s]) ? 0 : static_cast<int>(min(off[id + 1] - o, static_cast<int64_t>(1 << 30)));
    idx[s] = s;
    if (len > UMI_MAXLEN) {
        atomicMin(&bad[1], s);
        atomicMin(&bad[2], -len);
        U.code[s] = 0; U.nmask[s] = 0; U.comp[s] = 0; U.meta[s] = 0; key_hi[s] = 0; key_lo[s] = 0;
        return;
    }
    unsigned long long code = 0, khi = 0, klo = 0;
    uint32_t nmask = 0, comp = 0;
    for (int i = 0; i < len; ++i) {
        const uint8_t c = chars[o + i];
        unsigned v;  // trie child order A,C,G,T,N (src/sorted_trie.cpp:10)
        switch (c) {
            case 'A': v = 0; break;
            case 'C': v = 1; break;
            case 'G': v = 2; break;
            case 'T': v = 3; break;
            case 'N': v = 4; break;
            default: v = 4; atomicMin(&bad[0], s); break;
        }
        if (v == 4) nmask |= 1u << i;
        else { code |= static_cast<unsigned long long>(v) << (2 * i); comp += 1u << (8 * v); }
        const unsigned long long k = v + 1;  // 0 = past the end, so prefixes sort first
        if (i < 21) khi |= k << (3 * (20 - i));
        else klo |= k << (3 * (20 - (i - 21)));
    }
    U.code[s] = code; U.nmask[s] = nmask; U.comp[s] = comp;
    U.meta[s] = static_cast<uint32_t>(len) | (static_cast<uint32_t>(__popc(nmask)) << 12);
    key_hi[s] = khi; key_lo[s] = klo;
}

// The same for strings of up to UMI_LONG_MAX bases: UMI_LONG_WORDS code / mask words, one sort key per 21 bases.
__global__ void k_umi_encode_long(const uint8_t* chars, const int64_t* off, const int32_t* members, int n,
                                  UmiArrays U, unsigned long long* keys /* [nkeys][n] */, int* idx,
                                  const uint8_t* skip, int* bad, int words, int nkeys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const long long id = members ? static_cast<long long>(members[s]) - 1 : s;
    const long long o = off[id];
    const int len = (skip && skip[s]) ? 0 : static_cast<int>(off[id + 1] - o);   // <= 32 words: checked by the caller
    idx[s] = s;
    uint32_t comp = 0;
    int nN = 0;
    for (int w = 0; w < words; ++w) {
        unsigned long long code = 0;
        uint32_t nmask = 0;
        const int hi = min(len - 32 * w, 32);
        for (int i = 0; i < hi; ++i) {
            const uint8_t c = chars[o + 32 * w + i];
            unsigned v;
            switch (c) {
                case 'A': v = 0; break;
                case 'C': v = 1; break;
                case 'G': v = 2; break;
                case 'T': v = 3; break;
                case 'N': v = 4; break;
                default: v = 4; atomicMin(&bad[0], s); break;
            }
            if (v == 4) { nmask |= 1u << i; ++nN; }
            else { code |= static_cast<unsigned long long>(v) << (2 * i); comp += 1u << (8 * v); }
        }
        U.code[w * U.stride + s] = code;
        U.nmask[w * U.stride + s] = nmask;
    }
    for (int k = 0; k < nkeys; ++k) {
        unsigned long long key = 0;
        const int hi = min(len - UMI_KEY_BASES * k, UMI_KEY_BASES);
        for (int i = 0; i < hi; ++i) {
            const uint8_t c = chars[o + UMI_KEY_BASES * k + i];
            const unsigned long long v = c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 5;
            key |= v << (3 * (UMI_KEY_BASES - 1 - i));
        }
        keys[static_cast<long long>(k) * n + s] = key;
    }
    U.comp[s] = words > UMI_LONG_WORDS ? 0u : comp;   // byte counters: at most 128 per letter (beyond 4 words the composition bound is not used)
    U.meta[s] = static_cast<uint32_t>(len) | (static_cast<uint32_t>(nN) << 12);
}

__global__ void k_gather_u64(const unsigned long long* src, const int* perm, unsigned long long* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void k_gather_gid(const int* gid, const int* perm, unsigned long long* key, int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = gid[perm[i]];
    if (key) key[i] = static_cast<unsigned long long>(g);
    if (out) out[i] = g;
}

__global__ void k_gather_umi(UmiArrays src, const int* perm, UmiArrays dst, int n, int words) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = perm[i];
    for (int w = 0; w < words; ++w) {
        dst.code[w * dst.stride + i] = src.code[w * src.stride + p];
        dst.nmask[w * dst.stride + i] = src.nmask[w * src.stride + p];
    }
    dst.comp[i] = src.comp[p]; dst.meta[i] = src.meta[p];
}

// ---------------------------------------------------------------------------
// distance

// Doubled masked Levenshtein distance restricted to the band |i-j| <= K; returns
// INF_D as soon as every cell of a row exceeds lim2 (src/sorted_trie.cpp:13-21 costs).
template <int K>
__device__ __forceinline__ int banded_lev2(unsigned long long ca, uint32_t na, int la,
                                           unsigned long long cb, uint32_t nb, int lb, int lim2) {
    constexpr int BW = 2 * K + 1;
    int v[BW];
#pragma unroll
    for (int d = 0; d < BW; ++d) {
        const int i = d - K;
        v[d] = (i >= 0 && i <= la) ? 2 * i : INF_D;
    }
    for (int j = 1; j <= lb; ++j) {
        const unsigned cbj = static_cast<unsigned>(cb >> (2 * (j - 1))) & 3u;
        const unsigned nbj = (nb >> (j - 1)) & 1u;
        int rowmin = INF_D, left = INF_D;
#pragma unroll
        for (int d = 0; d < BW; ++d) {
            const int i = j + d - K;
            int best = INF_D;
            if (i >= 0 && i <= la) {
                if (i == 0) {
                    best = 2 * j;
                } else {
                    const unsigned nai = (na >> (i - 1)) & 1u;
                    const unsigned cai = static_cast<unsigned>(ca >> (2 * (i - 1))) & 3u;
                    const int sub = (nai | nbj) ? 1 : (cai == cbj ? 0 : 2);
                    best = v[d] + sub;
                    if (d + 1 < BW) best = min(best, v[d + 1] + 2);
                    best = min(best, left + 2);
                }
            }
            v[d] = best;
            left = best;
            rowmin = min(rowmin, best);
        }
        if (rowmin > lim2) return INF_D;
    }
    const int dd = la - lb + K;
    int res = INF_D;
#pragma unroll
    for (int d = 0; d < BW; ++d) res = (d == dd) ? v[d] : res;
    return res;
}

struct TileInfo;
struct PairArgs {
    UmiArrays U;                    // in (pre-group, trie) order
    const int* gid;                 // pre-group of every element in that order (nullptr: one group)
    int n;
    int lim2;
    unsigned long long* edges;      // (rank_i << 32 | rank_j), rank_i < rank_j
    unsigned long long* count;
    unsigned long long cap;
    int tile_lo;                    // first row tile of this launch (row tiles shard across GPUs)
    const uint32_t* tile_list;      // optional: the (row tile << 16 | column tile) pairs to search, one per block
    const TileInfo* sub_info;       // optional: common prefixes of the 64-element blocks (4 per tile)
    unsigned list_stride;           // block b searches tile_list[b * list_stride] (1; larger: a sample of the list)
    int special_lreq;               // >= 0: only pairs with a member that holds an N or is shorter than this (the rest
                                    // comes from the split-key search); -1: every pair
};

// ---------------------------------------------------------------------------
// Tile-level prefilter.  Inside a pre-group the elements are in trie (lexicographic) order, so
// the 256 strings of a tile share the common prefix of its first and last element.  A pair
// (s in row tile, t in column tile) within `L` edits aligns s[0..m) with some t[0..m'),
// |m - m'| <= L, at a cost <= L; with m <= |P_R| and m + L <= |P_C| both prefixes are known from
// the tiles alone, so whole tile pairs are discarded when no such m' exists -- exactly, because
// only pairs that cannot be neighbours are skipped.  Tiles that hold an N (a masked base costs
// half an edit) or span two pre-groups carry no prefix and are never discarded.
struct TileInfo {
    unsigned long long pcode;   // common prefix, 2 bits per base from bit 0
    int plen;                   // its length; -1: no information
    int special;                // some string of the tile holds an N or is shorter than `lreq`
};

template <int BLK>
__global__ void __launch_bounds__(BLK) k_tile_info(UmiArrays U, const int* gid, int n, int lreq, TileInfo* info) {
    const int t0 = blockIdx.x * BLK, t1 = min(t0 + BLK, n) - 1;
    const int i = t0 + threadIdx.x;
    const uint32_t nm = i < n ? U.nmask[i] : 0u;
    const int anyN = __syncthreads_or(nm != 0u);
    const int anyShort = __syncthreads_or(i < n && umi_len(U.meta[i]) < lreq);
    if (threadIdx.x != 0) return;
    TileInfo ti{0ull, -1, (anyN || anyShort) ? 1 : 0};
    if (!anyN && (!gid || gid[t0] == gid[t1])) {
        const unsigned long long a = U.code[t0], b = U.code[t1];
        const int la = umi_len(U.meta[t0]), lb = umi_len(U.meta[t1]);
        const unsigned long long x = a ^ b;
        int cp = x ? (__builtin_ctzll(x) >> 1) : 32;
        cp = min(cp, min(la, lb));
        ti.plen = cp;
        ti.pcode = cp >= 32 ? a : (a & ((1ull << (2 * cp)) - 1ull));
    }
    info[blockIdx.x] = ti;
}

// min over m' in [m - L, m + L] of the edit distance between x[0..m) and y[0..m') (unit costs),
// y known to at least m + L bases; > L is reported as L + 1.
template <int L>
__device__ __forceinline__ int prefix_dist(unsigned long long x, int m, unsigned long long y) {
    constexpr int BW = 2 * L + 1;
    int v[BW];   // v[d]: D[i][i + d - L]
#pragma unroll
    for (int d = 0; d < BW; ++d) v[d] = (d >= L) ? d - L : (L + 1);   // row 0: D[0][j] = j
    for (int i = 1; i <= m; ++i) {
        const unsigned xi = static_cast<unsigned>(x >> (2 * (i - 1))) & 3u;
        int left = L + 1;
#pragma unroll
        for (int d = 0; d < BW; ++d) {
            const int j = i + d - L;
            int best = L + 1;
            if (j == 0) best = i;
            else if (j > 0) {
                const unsigned yj = static_cast<unsigned>(y >> (2 * (j - 1))) & 3u;
                best = v[d] + (xi == yj ? 0 : 1);                 // D[i-1][j-1]
                if (d + 1 < BW) best = min(best, v[d + 1] + 1);   // D[i-1][j]
                best = min(best, left + 1);                        // D[i][j-1]
            }
            best = min(best, L + 1);
            v[d] = best;
            left = best;
        }
    }
    int res = L + 1;
#pragma unroll
    for (int d = 0; d < BW; ++d) res = min(res, v[d]);
    return res;
}

template <int L>
__global__ void k_tile_pairs(const TileInfo* info, int nt, int tile_lo, int tile_hi, int special_only, uint32_t* list, unsigned int* count) {
    const int bj = blockIdx.x * blockDim.x + threadIdx.x;
    const int bi = blockIdx.y + tile_lo;
    if (bi >= tile_hi || bj >= nt || bj < bi) return;
    bool keep = true;
    if (special_only && !info[bi].special && !info[bj].special) keep = false;
    if (keep && bj != bi) {
        const TileInfo R = info[bi], C = info[bj];
        if (R.plen >= 0 && C.plen >= 0) {
            // either orientation may prove that the tiles hold no neighbours
            const int m1 = min(R.plen, C.plen - L), m2 = min(C.plen, R.plen - L);
            if (m1 > L && prefix_dist<L>(R.pcode, m1, C.pcode) > L) keep = false;
            if (keep && m2 > L && prefix_dist<L>(C.pcode, m2, R.pcode) > L) keep = false;
        }
    }
    if (keep) list[atomicAdd(count, 1u)] = (static_cast<uint32_t>(bi) << 16) | static_cast<uint32_t>(bj);
}

// Shifted-Hamming lower bound for N-free pairs: a position of `a` that differs from b at every
// shift -K..K cannot be matched by any alignment within the band, so it costs a substitution or
// an indel (2 each).  More than `limit` such positions => d2 > 2*limit.
template <int K>
__device__ __forceinline__ bool shd_reject(unsigned long long ca, int la, unsigned long long cb, int lb, int limit) {
    const unsigned long long EVEN = 0x5555555555555555ull;
    const unsigned long long amask = la >= 32 ? ~0ull : ((1ull << (2 * la)) - 1ull);
    unsigned long long all = EVEN & amask;
#pragma unroll
    for (int s = -K; s <= K; ++s) {
        const unsigned long long xb = s >= 0 ? (cb >> (2 * s)) : (cb << (-2 * s));
        const unsigned long long diff = ca ^ xb;
        unsigned long long m = (diff | (diff >> 1)) & EVEN;
        // positions whose partner p+s falls outside b count as mismatches
        const int hi = lb - s;  // p < hi
        unsigned long long valid = hi >= 32 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << (2 * hi)) - 1ull));
        if (s < 0) valid &= ~((1ull << (-2 * s)) - 1ull);
        m |= ~valid;
        all &= m;
    }
    return __popcll(all) > limit;
}

template <int K>
__global__ void __launch_bounds__(TILE) k_umi_pairs(const PairArgs A) {
    int bi = blockIdx.x + A.tile_lo, bj = blockIdx.y;
    if (A.tile_list) { const uint32_t e = A.tile_list[static_cast<size_t>(blockIdx.x) * A.list_stride]; bi = static_cast<int>(e >> 16); bj = static_cast<int>(e & 0xffffu); }
    if (bj < bi) return;
    // column tile (c*) and row tile (r*) both live in LDS: survivors of the cheap filters are
    // queued per wave and evaluated 64 at a time, so the exact DP always runs on full waves
    __shared__ unsigned long long c_code[TILE], r_code[TILE];
    __shared__ uint32_t c_nmask[TILE], c_meta[TILE], r_nmask[TILE], r_meta[TILE];
    __shared__ uint4 c_key[TILE];  // {pre-group, meta, composition, N mask}: one broadcast read per column
    __shared__ uint32_t s_q1[TILE / 64][128], s_q2[TILE / 64][128];
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    if (A.gid && bj > bi) {
        // elements are sorted by pre-group: the tiles share no group unless the first group of
        // the column tile is still open at the end of the row tile
        const int row_last = min(bi * TILE + TILE, A.n) - 1;
        if (A.gid[bj * TILE] > A.gid[row_last]) return;
    }
    const int jcol = bj * TILE + t;
    if (jcol < A.n) {
        c_code[t] = A.U.code[jcol]; c_nmask[t] = A.U.nmask[jcol]; c_meta[t] = A.U.meta[jcol];
        c_key[t] = make_uint4(A.gid ? static_cast<uint32_t>(A.gid[jcol]) : 0u, A.U.meta[jcol], A.U.comp[jcol], A.U.nmask[jcol]);
    } else {
        c_code[t] = 0; c_nmask[t] = 0; c_meta[t] = UMI_META_NONE;  // never matches
        c_key[t] = make_uint4(0xffffffffu, UMI_META_NONE, 0u, 0u);
    }
    const int i = bi * TILE + t;
    const bool row_on = i < A.n;
    const unsigned long long ca = row_on ? A.U.code[i] : 0ull;
    const uint32_t na = row_on ? A.U.nmask[i] : 0u, compa = row_on ? A.U.comp[i] : 0u, ma = row_on ? A.U.meta[i] : UMI_META_NONE;
    r_code[t] = ca; r_nmask[t] = na; r_meta[t] = ma;
    __syncthreads();
    const int la = umi_len(ma), nNa = umi_nn(ma);
    const int gi = (row_on && A.gid) ? A.gid[i] : (row_on ? 0 : -2);
    const int limit = A.lim2 / 2;
    const bool row_special = A.special_lreq < 0 || na != 0u || la < A.special_lreq;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* const q1 = s_q1[wv];   // pairs that passed the length / composition bounds
    uint32_t* const q2 = s_q2[wv];   // ... and the shifted-Hamming bound: exact DP pending
    int n1 = 0, n2 = 0;              // wave-uniform fill levels

    // Two-level compaction: each filter runs on full waves of candidates, so a rare survivor
    // never drags 63 idle lanes through the next, more expensive stage.
    auto run_dp = [&](int count) {
        if (lane < count) {
            const uint32_t e = q2[lane];
            const int ti = e >> 8, jj = e & 0xff;
            const uint32_t mra = r_meta[ti], mcb = c_meta[jj];
            const int d = banded_lev2<K>(r_code[ti], r_nmask[ti], umi_len(mra), c_code[jj], c_nmask[jj], umi_len(mcb), A.lim2);
            if (d <= A.lim2) {
                const unsigned long long slot = atomicAdd(A.count, 1ull);
                if (slot < A.cap)
                    A.edges[slot] = (static_cast<unsigned long long>(bi * TILE + ti) << 32) | static_cast<unsigned>(bj * TILE + jj);
            }
        }
    };
    auto push2 = [&](bool keep, uint32_t e) {
        const unsigned long long m = __ballot(keep);
        if (m) {
            if (keep) q2[n2 + __popcll(m & lt)] = e;
            n2 += __popcll(m);
            if (n2 >= 64) {
                run_dp(64);
                const uint32_t moved = (64 + lane < n2) ? q2[64 + lane] : 0u;
                if (64 + lane < n2) q2[lane] = moved;
                n2 -= 64;
            }
        }
    };
    auto run_shd = [&](int count) {
        bool keep = false;
        uint32_t e = 0;
        if (lane < count) {
            e = q1[lane];
            const int ti = e >> 8, jj = e & 0xff;
            keep = true;
            if (K <= 8 && (r_nmask[ti] | c_nmask[jj]) == 0u)
                keep = !shd_reject<(K <= 8 ? K : 0)>(r_code[ti], umi_len(r_meta[ti]), c_code[jj], umi_len(c_meta[jj]), limit);
        }
        push2(keep, e);
    };

    const int jn = min(TILE, A.n - bj * TILE);
    // Sub-tile prefilter: the 64 rows of this wave and each 64-column block of the tile have
    // longer common prefixes than the 256-element tiles; a block whose prefix cannot align with
    // the wave's within `limit` edits holds no neighbour of these rows (same argument as k_tile_pairs).
    unsigned sub_ok = 0xfu;
    if (K >= 1 && K <= 5 && A.sub_info && bi != bj) {
        bool ok = true;
        if (lane < 4) {
            const int rb = bi * 4 + wv, cb = bj * 4 + lane;
            if (rb * 64 < A.n && cb * 64 < A.n) {
                const TileInfo R = A.sub_info[rb], C = A.sub_info[cb];
                if (R.plen >= 0 && C.plen >= 0) {
                    constexpr int L = (K >= 1 && K <= 5) ? K : 1;
                    const int m1 = min(R.plen, C.plen - L), m2 = min(C.plen, R.plen - L);
                    if (m1 > L && prefix_dist<L>(R.pcode, m1, C.pcode) > L) ok = false;
                    if (ok && m2 > L && prefix_dist<L>(C.pcode, m2, R.pcode) > L) ok = false;
                }
            }
        }
        sub_ok = static_cast<unsigned>(__ballot(ok)) & 0xfu;
    }
    bool row_ok = true;   // this row can have neighbours in the current 64-column block
    for (int jj = 0; jj < jn; ++jj) {
        if (!((sub_ok >> (jj >> 6)) & 1u)) { jj |= 63; continue; }   // skip the whole 64-column block
        if (K >= 1 && K <= 3 && A.sub_info && bi != bj && (jj & 63) == 0) {
            // the row's own string is fully known: the block's whole prefix has to align with its
            // first plen +- limit bases within `limit` edits
            constexpr int L = (K >= 1 && K <= 3) ? K : 1;
            const TileInfo C = A.sub_info[bj * 4 + (jj >> 6)];
            const int m2 = min(C.plen, la - L);
            row_ok = !(row_on && na == 0u && C.plen >= 0 && m2 > L && prefix_dist<L>(C.pcode, m2, ca) > L);
            if (!__ballot(row_ok)) { jj |= 63; continue; }
        }
        const uint4 ck = c_key[jj];
        bool pass = row_ok && row_on && static_cast<int>(ck.x) == gi && (bi != bj || jj > t);
        {
            const int lb = umi_len(ck.y), nNb = umi_nn(ck.y);
            const int dl = la > lb ? la - lb : lb - la;
            // composition lower bound: every edit costs >= 1 and moves the 5-letter composition by
            // <= 2 (<= its cost when no N is involved)
            const int l1 = static_cast<int>(__builtin_amdgcn_sad_u8(compa, ck.z, 0u)) + (nNa > nNb ? nNa - nNb : nNb - nNa);
            const bool anyN = (na | ck.w) != 0u;
            pass = pass && 2 * dl <= A.lim2 && l1 <= (anyN ? 2 * A.lim2 : A.lim2);
            pass = pass && (row_special || ck.w != 0u || lb < A.special_lreq);
        }
        const unsigned long long mask = __ballot(pass);
        if (mask) {
            if (pass) q1[n1 + __popcll(mask & lt)] = (static_cast<uint32_t>(t) << 8) | static_cast<uint32_t>(jj);
            n1 += __popcll(mask);
            if (n1 >= 64) {
                run_shd(64);
                const uint32_t moved = (64 + lane < n1) ? q1[64 + lane] : 0u;
                if (64 + lane < n1) q1[lane] = moved;
                n1 -= 64;
            }
        }
    }
    run_shd(n1);
    run_dp(n2);
}

// ---------------------------------------------------------------------------
// Strings of 33..UMI_LONG_MAX bases (4-word codes).  Same distance, same tiling, the same exact
// length / composition bounds; the prefix and shifted-Hamming filters of the one-word path are not
// carried over (they are only filters: the result is the same set of pairs).

// base i (0-based) of a 4-word string whose word planes lie `plane` elements apart
struct LongStr {
    const unsigned long long* code;
    const uint32_t* nmask;
    int plane;
    __device__ __forceinline__ unsigned base(int i) const { return static_cast<unsigned>(code[(i >> 5) * plane] >> (2 * (i & 31))) & 3u; }
    __device__ __forceinline__ unsigned isn(int i) const { return (nmask[(i >> 5) * plane] >> (i & 31)) & 1u; }
};

// banded_lev2 with the band in registers (K <= 16) on LongStr operands
template <int K>
__device__ __forceinline__ int banded_lev2_long(const LongStr a, int la, const LongStr b, int lb, int lim2) {
    constexpr int BW = 2 * K + 1;
    int v[BW];
#pragma unroll
    for (int d = 0; d < BW; ++d) {
        const int i = d - K;
        v[d] = (i >= 0 && i <= la) ? 2 * i : INF_D;
    }
    for (int j = 1; j <= lb; ++j) {
        const unsigned cbj = b.base(j - 1), nbj = b.isn(j - 1);
        int rowmin = INF_D, left = INF_D;
#pragma unroll
        for (int d = 0; d < BW; ++d) {
            const int i = j + d - K;
            int best = INF_D;
            if (i >= 0 && i <= la) {
                if (i == 0) {
                    best = 2 * j;
                } else {
                    const int sub = (a.isn(i - 1) | nbj) ? 1 : (a.base(i - 1) == cbj ? 0 : 2);
                    best = v[d] + sub;
                    if (d + 1 < BW) best = min(best, v[d + 1] + 2);
                    best = min(best, left + 2);
                }
            }
            v[d] = best;
            left = best;
            rowmin = min(rowmin, best);
        }
        if (rowmin > lim2) return INF_D;
    }
    const int dd = la - lb + K;
    int res = INF_D;
#pragma unroll
    for (int d = 0; d < BW; ++d) res = (d == dd) ? v[d] : res;
    return res;
}

// full (unbanded) doubled masked Levenshtein distance, one row in thread-private memory; gives up with
// INF_D once a whole row exceeds lim2.  For thresholds beyond 16 and for the dense distances.
template <int MAXL>
__device__ __forceinline__ int full_lev2_long(const LongStr a, int la, const LongStr b, int lb, int lim2) {
    int row[MAXL + 1];
    for (int i = 0; i <= la; ++i) row[i] = 2 * i;
    for (int j = 1; j <= lb; ++j) {
        const unsigned cbj = b.base(j - 1), nbj = b.isn(j - 1);
        int diag = row[0];
        row[0] = 2 * j;
        int rowmin = row[0];
        for (int i = 1; i <= la; ++i) {
            const int sub = (a.isn(i - 1) | nbj) ? 1 : (a.base(i - 1) == cbj ? 0 : 2);
            const int best = min(diag + sub, min(row[i] + 2, row[i - 1] + 2));
            diag = row[i];
            row[i] = best;
            rowmin = min(rowmin, best);
        }
        if (rowmin > lim2) return INF_D;
    }
    return row[la];
}

// K: band held in registers; K < 0: full DP.  XL: strings of more than UMI_LONG_MAX bases -- as many words as the longest
// needs (up to UMI_XL_WORDS), read where they lie in HBM instead of from a staged tile (the planes of a tile's 256 strings
// are 2 KB runs each; the words a band touches stay in L1 / L2), no composition bound (its byte counters stop at 255).
template <int K, bool XL>
__global__ void __launch_bounds__(TILE) k_umi_pairs_long(const PairArgs A) {
    const int bi = blockIdx.x + A.tile_lo, bj = blockIdx.y;
    if (bj < bi) return;
    constexpr int SW = XL ? 1 : UMI_LONG_WORDS;   // staged words per string
    __shared__ unsigned long long c_code[SW * TILE], r_code[SW * TILE];
    __shared__ uint32_t c_nmask[SW * TILE], r_nmask[SW * TILE];
    __shared__ uint4 c_key[TILE];  // {pre-group, meta, composition, any N}
    const int t = threadIdx.x;
    if (A.gid && bj > bi) {
        const int row_last = min(bi * TILE + TILE, A.n) - 1;
        if (A.gid[bj * TILE] > A.gid[row_last]) return;
    }
    const int jcol = bj * TILE + t, i = bi * TILE + t;
    const bool row_on = i < A.n;
    uint32_t anyN_col = 0, na = 0;
    if (XL) {   // only "some base is N" is needed up front
        anyN_col = (jcol < A.n && umi_nn(A.U.meta[jcol]) > 0) ? 1u : 0u;
        na = (row_on && umi_nn(A.U.meta[i]) > 0) ? 1u : 0u;
    } else {
        for (int w = 0; w < UMI_LONG_WORDS; ++w) {
            const bool on = jcol < A.n;
            c_code[w * TILE + t] = on ? A.U.code[w * A.U.stride + jcol] : 0ull;
            const uint32_t m = on ? A.U.nmask[w * A.U.stride + jcol] : 0u;
            c_nmask[w * TILE + t] = m;
            anyN_col |= m;
            r_code[w * TILE + t] = row_on ? A.U.code[w * A.U.stride + i] : 0ull;
            const uint32_t mr = row_on ? A.U.nmask[w * A.U.stride + i] : 0u;
            r_nmask[w * TILE + t] = mr;
            na |= mr;
        }
    }
    c_key[t] = jcol < A.n ? make_uint4(A.gid ? static_cast<uint32_t>(A.gid[jcol]) : 0u, A.U.meta[jcol], A.U.comp[jcol], anyN_col)
                          : make_uint4(0xffffffffu, UMI_META_NONE, 0u, 0u);
    __syncthreads();
    const uint32_t compa = row_on ? A.U.comp[i] : 0u, ma = row_on ? A.U.meta[i] : UMI_META_NONE;
    const int la = umi_len(ma), nNa = umi_nn(ma);
    const int gi = (row_on && A.gid) ? A.gid[i] : (row_on ? 0 : -2);
    const int gstride = static_cast<int>(A.U.stride);
    const LongStr sa = XL ? LongStr{A.U.code + (row_on ? i : 0), A.U.nmask + (row_on ? i : 0), gstride} : LongStr{r_code + t, r_nmask + t, TILE};
    const int jn = min(TILE, A.n - bj * TILE);
    for (int jj = 0; jj < jn; ++jj) {
        const uint4 ck = c_key[jj];
        bool pass = row_on && static_cast<int>(ck.x) == gi && (bi != bj || jj > t);
        const int lb = umi_len(ck.y), nNb = umi_nn(ck.y);
        const int dl = la > lb ? la - lb : lb - la;
        // the composition bound of k_umi_pairs; the byte counters hold up to 128 per letter, sad_u8 is exact
        const int l1 = static_cast<int>(__builtin_amdgcn_sad_u8(compa, ck.z, 0u)) + (nNa > nNb ? nNa - nNb : nNb - nNa);
        const bool anyN = (na | ck.w) != 0u;
        pass = pass && 2 * dl <= A.lim2 && l1 <= (anyN ? 2 * A.lim2 : A.lim2);
        if (!pass) continue;
        const LongStr sb = XL ? LongStr{A.U.code + bj * TILE + jj, A.U.nmask + bj * TILE + jj, gstride} : LongStr{c_code + jj, c_nmask + jj, TILE};
        int d;
        if constexpr (K >= 0) d = banded_lev2_long<(K >= 0 ? K : 0)>(sa, la, sb, lb, A.lim2);
        else d = full_lev2_long<(XL ? UMI_XL_MAX : UMI_LONG_MAX)>(sa, la, sb, lb, A.lim2);
        if (d <= A.lim2) {
            const unsigned long long slot = atomicAdd(A.count, 1ull);
            if (slot < A.cap)
                A.edges[slot] = (static_cast<unsigned long long>(bi * TILE + t) << 32) | static_cast<unsigned>(bj * TILE + jj);
        }
    }
}

// ---------------------------------------------------------------------------
// Split-key neighbour search for thresholds 1 to 3 on large sets.
//
// The all-tile-pairs search above looks at every pair: at threshold 3 on 12-base UMIs its length, composition and
// shifted-Hamming bounds pass most of them on to the exact DP.  The search below enumerates candidates instead.
// Let lev(a, b) <= k for N-free a, b, and fix h, s with h + s <= |a|.  An optimal alignment sends a[0..h) to a prefix
// b1 of b and the rest of a to the rest of b; the costs of the two parts sum to <= k.  So with k1 + k2 = k - 1
//     P(a,b): some prefix of b is within k1 edits of the first h bases of a,  or
//     S(a,b): some suffix of b is within k2 edits of the last s bases of a
// (the last s bases of a lie inside the second part, and the part of an alignment that covers them costs no more than
// the whole; threshold 1: k1 = k2 = 0, one of the two keys matches exactly).  With k1, k2 <= 1 the strings b that satisfy P(a, .) are those that start with one of the <= 8h + 5
// one-edit variants of a[0..h) -- a union of contiguous ranges of the set in trie order; S(a, .) the same in the order
// of the reversed strings.  Rows that share their first h bases share the ranges, so the work items are (row group,
// 256 candidate columns); a lane holds one column as the pattern of a bit-vector edit distance (Myers 1999 / Hyyro
// 2003, global variant) and the rows of the group stream through as the text, from scalar registers.
// Every pair {a, b} is reported once, from its lower-ranked member a: by the prefix scan if P(a, b), else by the
// suffix scan (which evaluates P(a, b) on the first h + 1 bases to leave those pairs to the prefix scan).  Only a
// needs h + s <= |a|, so the rows are scanned by length class, each with keys of half its length (at most 8 bases),
// against columns of any length.
// Strings with an N (a masked base costs half an edit: the argument above does not hold) or shorter than 8 bases are
// "special": their pairs come from the tile kernel restricted to pairs with a special member.
// The result is the same set of pairs as the tile search (tests: both against the oracle and against each other).

struct SkElem {
    uint32_t plo, phi;   // bit planes of the 2-bit codes, base i at bit i; the scan order's own orientation
    uint32_t meta;       // len | special << 6 | (first 9 bases of the FORWARD string, 2 bits each) << 8
    uint32_t rank;       // rank in trie order
};

struct SkOrder {
    SkElem* el;                  // [n] in scan order
    unsigned long long* okey;    // [n] 3 bits per base (A..T = 1..4, N = 5, past the end = 0), first 21 bases, in scan order
    int* gid;                    // [n] pre-group in scan order (nullptr: one group)
    int n;
};

__device__ __forceinline__ uint32_t sk_even_bits(unsigned long long x) {   // bits 0, 2, 4, ... of x -> bits 0, 1, 2, ...
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return static_cast<uint32_t>(x);
}

__device__ __forceinline__ unsigned long long sk_order_key(uint32_t plo, uint32_t phi, uint32_t nmask, int len) {
    unsigned long long k = 0;
    const int m = min(len, 21);
    for (int i = 0; i < m; ++i) {
        const unsigned long long d = ((nmask >> i) & 1u) ? 5ull : 1ull + ((plo >> i) & 1u) + 2ull * ((phi >> i) & 1u);
        k |= d << (3 * (20 - i));
    }
    return k;
}

__global__ void __launch_bounds__(256) k_sk_lenhist(UmiArrays U, int n, unsigned int* hist /* [34]: lengths 0..32, [33] strings with an N */) {
    __shared__ unsigned int s_h[34];
    if (threadIdx.x < 34) s_h[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        atomicAdd(&s_h[min(umi_len(U.meta[i]), 32)], 1u);
        if (U.nmask[i]) atomicAdd(&s_h[33], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 34 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// Elements in trie order (REV = false) or, per trie rank, the reversed string with its sort key (REV = true).
template <bool REV>
__global__ void k_sk_elems(UmiArrays U, int n, int lreq, SkElem* el, unsigned long long* okey, int* val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long code = U.code[i];
    const uint32_t nm = U.nmask[i];
    const int len = umi_len(U.meta[i]);
    uint32_t plo = sk_even_bits(code), phi = sk_even_bits(code >> 1), nmo = nm;
    if (REV && len > 0) { plo = __brev(plo) >> (32 - len); phi = __brev(phi) >> (32 - len); nmo = __brev(nm) >> (32 - len); }
    const uint32_t special = (nm != 0u || len < lreq) ? 1u : 0u;
    el[i] = SkElem{plo, phi, static_cast<uint32_t>(len) | (special << 6) | (static_cast<uint32_t>(code & 0x3ffffull) << 8), static_cast<uint32_t>(i)};
    okey[i] = sk_order_key(plo, phi, nmo, len);
    if (REV) val[i] = i;
}

__global__ void k_sk_permute(const SkElem* el, const int* val, const int* gid, int n, SkElem* out, int* gid_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = val[i];
    out[i] = el[r];
    if (gid_out) gid_out[i] = gid[r];
}

// Row groups: maximal runs of the scan order that share pre-group and first h bases.
__global__ void k_sk_group_flags(SkOrder O, int h, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > O.n) return;
    if (i == O.n) { flag[i] = 0; return; }
    bool f = i == 0;
    if (!f) {
        const int sh = 3 * (21 - h);
        f = (O.okey[i] >> sh) != (O.okey[i - 1] >> sh) || (O.gid && O.gid[i] != O.gid[i - 1]);
    }
    flag[i] = f ? 1 : 0;
}

__global__ void k_sk_group_starts(const int* flag, const long long* pos, int n, int* start) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { start[pos[n]] = n; return; }
    if (flag[i]) start[pos[i]] = i;
}

constexpr int SK_MAXR = 72;    // ranges per row group: 1 + 3h + h + 4(h + 1) <= 69 for h <= 8
constexpr int SK_ROWS = 128;   // rows per work item
constexpr int SK_COLS = 256;   // candidate columns per work item (one per thread)

// first index of the scan order whose (pre-group, key) is >= (g, k) [upper = false] or > (g, k) [upper = true]
__device__ __forceinline__ int sk_bound(const SkOrder& O, int g, unsigned long long k, bool upper) {
    int lo = 0, hi = O.n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int gm = O.gid ? O.gid[mid] : 0;
        const unsigned long long km = O.okey[mid];
        const bool before = gm != g ? gm < g : (upper ? km <= k : km < k);
        if (before) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The candidate columns of every row group: ranges of the scan order that start with a variant (<= k1 edits) of the
// group's first h bases, merged, as (start, candidates before it); `clip`: columns from the group's own start on
// (trie order: a pair is reported from its lower-ranked member).
__global__ void k_sk_ranges(SkOrder O, const int* rg_start, int nrg, int h, int k1, int clip,
                            int2* ranges, int* nranges, int* ctotal, long long* items) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > nrg) return;
    if (g == nrg) { items[g] = 0; return; }
    const int r0 = rg_start[g], r1 = rg_start[g + 1];
    const unsigned long long key = O.okey[r0];
    const int grp = O.gid ? O.gid[r0] : 0;
    int d[10];
    bool plain = true;   // first h bases present and N-free
    for (int i = 0; i < h; ++i) { d[i] = static_cast<int>((key >> (3 * (20 - i))) & 7ull); plain = plain && d[i] >= 1 && d[i] <= 4; }
    int lo[SK_MAXR], hi[SK_MAXR], nr = 0;
    auto add = [&](const int* v, int m) {
        unsigned long long k = 0;
        for (int i = 0; i < m; ++i) k |= static_cast<unsigned long long>(v[i]) << (3 * (20 - i));
        const unsigned long long fill = (1ull << (3 * (21 - m))) - 1ull;
        int a = sk_bound(O, grp, k, false);
        const int b = sk_bound(O, grp, k | fill, true);
        if (clip) a = max(a, r0);
        if (a >= b) return;
        // insert by start
        int p = nr++;
        while (p > 0 && lo[p - 1] > a) { lo[p] = lo[p - 1]; hi[p] = hi[p - 1]; --p; }
        lo[p] = a; hi[p] = b;
    };
    if (plain) {
        int v[10];
        for (int i = 0; i < h; ++i) v[i] = d[i];
        add(v, h);
        if (k1 >= 1) {
            for (int p = 0; p < h; ++p) {           // substitutions
                for (int c = 1; c <= 4; ++c) if (c != d[p]) { v[p] = c; add(v, h); }
                v[p] = d[p];
            }
            for (int p = 0; p < h; ++p) {           // one base of the h missing in b
                if (p > 0 && d[p] == d[p - 1]) continue;   // the same string as deleting p - 1
                int m = 0;
                for (int i = 0; i < h; ++i) if (i != p) v[m++] = d[i];
                add(v, h - 1);
            }
            for (int p = 0; p <= h; ++p)            // one more base in b
                for (int c = 1; c <= 4; ++c) {
                    if (p < h && c == d[p]) continue;      // the same string as inserting after the run
                    int m = 0;
                    for (int i = 0; i < p; ++i) v[m++] = d[i];
                    v[m++] = c;
                    for (int i = p; i < h; ++i) v[m++] = d[i];
                    add(v, h + 1);
                }
        }
    }
    // merge overlapping ranges
    int out = 0, total = 0;
    int2* R = ranges + static_cast<long long>(g) * SK_MAXR;
    int ca = 0, cb = 0;
    for (int i = 0; i < nr; ++i) {
        if (i == 0) { ca = lo[0]; cb = hi[0]; continue; }
        if (lo[i] <= cb) { cb = max(cb, hi[i]); continue; }
        R[out++] = make_int2(ca, total); total += cb - ca;
        ca = lo[i]; cb = hi[i];
    }
    if (nr) { R[out++] = make_int2(ca, total); total += cb - ca; }
    nranges[g] = out;
    ctotal[g] = total;
    items[g] = static_cast<long long>((r1 - r0 + SK_ROWS - 1) / SK_ROWS) * ((total + SK_COLS - 1) / SK_COLS);
}

// P(x, y): some prefix of y (ly bases long) within k1 (0 or 1) edits of the first h bases of x; x, y: 2-bit codes of the
// first 9 bases (zero beyond the end) -- exactly "y starts with one of the variants k_sk_ranges lists for x".
__device__ __forceinline__ bool sk_prefix_within(uint32_t x, uint32_t y, int ly, int h, int k1) {
    auto mism = [](uint32_t u) { return (u | (u >> 1)) & 0x55555555u; };
    const uint32_t mh = (1u << (2 * h)) - 1u;
    const uint32_t d0 = mism(x ^ y) & mh;
    if (ly >= h && __popc(d0) <= k1) return true;
    if (k1 == 0 || ly < h - 1) return false;
    const int f = d0 ? (__builtin_ctz(d0) >> 1) : h;             // first mismatch
    const uint32_t d1 = mism((x >> 2) ^ y) & (mh >> 2);          // x[p + 1] against y[p], p < h - 1
    const uint32_t d2 = mism(x ^ (y >> 2)) & mh;                 // x[p] against y[p + 1], p < h
    if ((d1 >> (2 * min(f, h - 1))) == 0u) return true;          // x without its base p is a prefix of y
    return ly >= h + 1 && (d2 >> (2 * min(f, h))) == 0u;         // x with one base inserted at p is a prefix of y
}

struct SkScanArgs {
    const SkElem* el;
    const int* rg_start;
    int nrg;
    const int2* ranges;
    const int* nranges;
    const int* ctotal;
    const long long* item_off;     // [nrg + 1] exclusive
    unsigned item_stride;          // block b works on item b * item_stride (1; larger: a sample)
    int limit, h, k1;              // threshold; the prefix split (the suffix scan's check of P)
    int len_lo, len_hi;            // rows of this launch: lengths len_lo..len_hi (their keys are h and s bases long)
    uint32_t row_lo, row_hi;       // trie ranks of the rows this launch reports (row tiles shard across GPUs)
    unsigned long long* edges;
    unsigned long long* count;
    unsigned long long cap;
};

template <bool SUFFIX>
__global__ void __launch_bounds__(SK_COLS) k_sk_scan(const SkScanArgs A) {
    __shared__ int2 s_rng[SK_MAXR];
    __shared__ unsigned long long s_q[SK_COLS / 64][128];
    __shared__ SkElem s_rows[SK_ROWS];
    const int t = threadIdx.x, lane = t & 63;
    const long long item = static_cast<long long>(blockIdx.x) * A.item_stride;
    int g = 0;
    {
        int lo = 0, hi = A.nrg;   // item_off[lo] <= item < item_off[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.item_off[mid] <= item) lo = mid; else hi = mid; }
        g = lo;
    }
    const int r0g = A.rg_start[g], r1g = A.rg_start[g + 1];
    const int C = A.ctotal[g], ncc = (C + SK_COLS - 1) / SK_COLS;
    const long long local = item - A.item_off[g];
    const int rc = static_cast<int>(local / ncc), cc = static_cast<int>(local % ncc);
    const int r0 = r0g + rc * SK_ROWS, r1 = min(r0 + SK_ROWS, r1g);
    const int nr = A.nranges[g];
    if (t < nr) s_rng[t] = A.ranges[static_cast<long long>(g) * SK_MAXR + t];
    if (t < r1 - r0) s_rows[t] = A.el[r0 + t];
    __syncthreads();
    const int vc = cc * SK_COLS + t;
    int col = -1;
    if (vc < C) {
        int a = 0, b = nr;
        while (b - a > 1) { const int m = (a + b) >> 1; if (s_rng[m].y <= vc) a = m; else b = m; }
        col = s_rng[a].x + (vc - s_rng[a].y);
    }
    SkElem e{0u, 0u, 1u | (1u << 6), 0u};
    if (col >= 0) e = A.el[col];
    const int lb = e.meta & 63;
    const bool valid = col >= 0 && !((e.meta >> 6) & 1u);
    const int sh = 32 - (valid ? lb : 1);
    const uint32_t pl = e.plo << sh, ph = e.phi << sh, pat = ~0u << sh, low = (2u << sh) - 1u;
    const uint32_t cfwd = e.meta >> 8;
    unsigned long long* const q = s_q[t >> 6];
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nq = 0;
    auto flush = [&](int count) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.count, static_cast<unsigned long long>(count));
        base = (static_cast<unsigned long long>(__shfl(static_cast<int>(base >> 32), 0)) << 32) | static_cast<unsigned>(__shfl(static_cast<int>(base), 0));
        if (lane < count && base + lane < A.cap) A.edges[base + lane] = q[lane];
    };
    for (int r = r0; r < r1; ++r) {
        const SkElem R = s_rows[r - r0];   // the same address in every lane: one broadcast read
        const uint32_t rmeta = __builtin_amdgcn_readfirstlane(R.meta);
        const uint32_t rrank = __builtin_amdgcn_readfirstlane(R.rank);
        const int la = rmeta & 63;
        if (((rmeta >> 6) & 1u) || la < A.len_lo || la > A.len_hi) continue;
        if (rrank < A.row_lo || rrank >= A.row_hi) continue;
        const uint32_t tlo = __builtin_amdgcn_readfirstlane(R.plo), thi = __builtin_amdgcn_readfirstlane(R.phi);
        uint32_t Pv = pat, Mv = 0u;
        for (int j = 0; j < la; ++j) {
            const uint32_t m0 = 0u - ((tlo >> j) & 1u), m1 = 0u - ((thi >> j) & 1u);
            const uint32_t Eq = ~((pl ^ m0) | (ph ^ m1)) & pat;
            const uint32_t Xv = Eq | Mv;
            const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint32_t Ph = Mv | ~(Xh | Pv);
            uint32_t Mh = Pv & Xh;
            Ph = (Ph << 1) | low;
            Mh <<= 1;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
        }
        // D[lb][la] = D[0][la] + the vertical deltas down the last column (row 0 of a global alignment holds j)
        const int d = la + __popc(Pv & pat) - __popc(Mv & pat);
        bool hit = valid && d <= A.limit && e.rank > rrank;   // a pair is reported from its lower-ranked member
        if (SUFFIX) hit = hit && !sk_prefix_within(rmeta >> 8, cfwd, lb, A.h, A.k1);
        const unsigned long long ball = __ballot(hit);
        if (ball) {
            if (hit) q[nq + __popcll(ball & lt)] = (static_cast<unsigned long long>(rrank) << 32) | e.rank;
            nq += __popcll(ball);
            if (nq >= 64) {
                flush(64);
                const unsigned long long moved = (64 + lane < nq) ? q[64 + lane] : 0ull;
                if (64 + lane < nq) q[lane] = moved;
                nq -= 64;
            }
        }
    }
    if (nq) flush(nq);
}

// The same scan with TWO candidate columns per lane, for the rows of up to 15 - limit bases (12-base UMIs; a string of 16
// and more bases is no neighbour of such a row whatever it holds, so those columns are not valid here):
// the two patterns sit top-aligned in the halves of one 32-bit word, so every operation of the recurrence advances both.
// What crosses from the lower half into the upper one lands on bits below the upper pattern: the carry of the addition
// (bit 16 holds no pattern bit as long as the upper string has at most 15 bases, and nothing is added there, so it goes no
// further), the top bit of Ph << 1 (overwritten by `low`, the row-0 deltas) and the top bit of Mh << 1 (masked out).
// Work items, candidate ranges and the pairs reported are those of k_sk_scan; half the threads per item.
template <bool SUFFIX>
__global__ void __launch_bounds__(SK_COLS / 2) k_sk_scan_pk(const SkScanArgs A) {
    __shared__ int2 s_rng[SK_MAXR];
    __shared__ unsigned long long s_q[SK_COLS / 128][192];
    __shared__ SkElem s_rows[SK_ROWS];
    const int t = threadIdx.x, lane = t & 63;
    const long long item = static_cast<long long>(blockIdx.x) * A.item_stride;
    int g = 0;
    {
        int lo = 0, hi = A.nrg;   // item_off[lo] <= item < item_off[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.item_off[mid] <= item) lo = mid; else hi = mid; }
        g = lo;
    }
    const int r0g = A.rg_start[g], r1g = A.rg_start[g + 1];
    const int C = A.ctotal[g], ncc = (C + SK_COLS - 1) / SK_COLS;
    const long long local = item - A.item_off[g];
    const int rc = static_cast<int>(local / ncc), cc = static_cast<int>(local % ncc);
    const int r0 = r0g + rc * SK_ROWS, r1 = min(r0 + SK_ROWS, r1g);
    const int nr = A.nranges[g];
    if (t < nr) s_rng[t] = A.ranges[static_cast<long long>(g) * SK_MAXR + t];
    for (int q = t; q < r1 - r0; q += SK_COLS / 2) s_rows[q] = A.el[r0 + q];
    __syncthreads();
    auto column = [&](int vc) -> int {
        if (vc >= C) return -1;
        int a = 0, b = nr;
        while (b - a > 1) { const int m = (a + b) >> 1; if (s_rng[m].y <= vc) a = m; else b = m; }
        return s_rng[a].x + (vc - s_rng[a].y);
    };
    const int col0 = column(cc * SK_COLS + t), col1 = column(cc * SK_COLS + SK_COLS / 2 + t);
    const SkElem none{0u, 0u, 1u | (1u << 6), 0u};
    const SkElem e0 = col0 >= 0 ? A.el[col0] : none, e1 = col1 >= 0 ? A.el[col1] : none;
    const int lb0 = e0.meta & 63, lb1 = e1.meta & 63;
    const bool valid0 = col0 >= 0 && !((e0.meta >> 6) & 1u) && lb0 < 16, valid1 = col1 >= 0 && !((e1.meta >> 6) & 1u) && lb1 < 16;
    const int sh0 = 16 - (valid0 ? lb0 : 1), sh1 = 16 - (valid1 ? lb1 : 1);
    const uint32_t F = 0xFFFFu;
    const uint32_t pl = ((e0.plo << sh0) & F) | (((e1.plo << sh1) & F) << 16), ph = ((e0.phi << sh0) & F) | (((e1.phi << sh1) & F) << 16);
    const uint32_t pat = ((F << sh0) & F) | (((F << sh1) & F) << 16);
    const uint32_t low = ((2u << sh0) - 1u) | (((2u << sh1) - 1u) << 16), keep = ~low;
    const uint32_t cfwd0 = e0.meta >> 8, cfwd1 = e1.meta >> 8;
    unsigned long long* const q = s_q[t >> 6];
    const unsigned long long lt = (1ull << lane) - 1ull;
    int nq = 0;
    auto flush = [&](int count) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.count, static_cast<unsigned long long>(count));
        base = (static_cast<unsigned long long>(__shfl(static_cast<int>(base >> 32), 0)) << 32) | static_cast<unsigned>(__shfl(static_cast<int>(base), 0));
        if (lane < count && base + lane < A.cap) A.edges[base + lane] = q[lane];
    };
    for (int r = r0; r < r1; ++r) {
        const SkElem R = s_rows[r - r0];   // the same address in every lane: one broadcast read
        const uint32_t rmeta = __builtin_amdgcn_readfirstlane(R.meta);
        const uint32_t rrank = __builtin_amdgcn_readfirstlane(R.rank);
        const int la = rmeta & 63;
        if (((rmeta >> 6) & 1u) || la < A.len_lo || la > A.len_hi) continue;
        if (rrank < A.row_lo || rrank >= A.row_hi) continue;
        const uint32_t tlo = __builtin_amdgcn_readfirstlane(R.plo), thi = __builtin_amdgcn_readfirstlane(R.phi);
        uint32_t Pv = pat, Mv = 0u;
        for (int j = 0; j < la; ++j) {
            const uint32_t m0 = 0u - ((tlo >> j) & 1u), m1 = 0u - ((thi >> j) & 1u);
            const uint32_t Eq = ~((pl ^ m0) | (ph ^ m1)) & pat;
            const uint32_t Xv = Eq | Mv;
            const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint32_t Ph = Mv | ~(Xh | Pv);
            uint32_t Mh = Pv & Xh;
            Ph = (Ph << 1) | low;
            Mh = (Mh << 1) & keep;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
        }
        const uint32_t pv = Pv & pat, mv = Mv & pat;
        const int d0 = la + __popc(pv & F) - __popc(mv & F), d1 = la + __popc(pv >> 16) - __popc(mv >> 16);
        bool hit0 = valid0 && d0 <= A.limit && e0.rank > rrank;   // a pair is reported from its lower-ranked member
        bool hit1 = valid1 && d1 <= A.limit && e1.rank > rrank;
        if (SUFFIX) {
            hit0 = hit0 && !sk_prefix_within(rmeta >> 8, cfwd0, lb0, A.h, A.k1);
            hit1 = hit1 && !sk_prefix_within(rmeta >> 8, cfwd1, lb1, A.h, A.k1);
        }
        const unsigned long long ball0 = __ballot(hit0), ball1 = __ballot(hit1);
        if (ball0 | ball1) {
            if (hit0) q[nq + __popcll(ball0 & lt)] = (static_cast<unsigned long long>(rrank) << 32) | e0.rank;
            nq += __popcll(ball0);
            if (hit1) q[nq + __popcll(ball1 & lt)] = (static_cast<unsigned long long>(rrank) << 32) | e1.rank;
            nq += __popcll(ball1);
            while (nq >= 64) {
                flush(64);
                for (int k = 64; k < nq; k += 64) {
                    const unsigned long long moved = (k + lane < nq) ? q[k + lane] : 0ull;
                    if (k + lane < nq) q[k - 64 + lane] = moved;
                }
                nq -= 64;
            }
        }
    }
    if (nq) flush(nq);
}

template <int MAXL>
__global__ void k_lev_dense_long(UmiArrays U, int n, double* out) {
    const long long p = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    const long long npairs = static_cast<long long>(n) * (n - 1) / 2;
    if (p >= npairs) return;
    long long i = static_cast<long long>((2.0 * n - 1 - sqrt((2.0 * n - 1) * (2.0 * n - 1) - 8.0 * p)) / 2);
    auto start = [&](long long r) { return r * (2LL * n - r - 1) / 2; };
    while (i > 0 && start(i) > p) --i;
    while (start(i + 1) <= p) ++i;
    const long long j = i + 1 + (p - start(i));
    const LongStr a{U.code + i, U.nmask + i, static_cast<int>(U.stride)}, b{U.code + j, U.nmask + j, static_cast<int>(U.stride)};
    const int d = full_lev2_long<MAXL>(a, umi_len(U.meta[i]), b, umi_len(U.meta[j]), 8 * MAXL);
    out[p] = static_cast<double>(d) / 2.0;
}

// Dense distances for compute_lev_masked: out in R 'dist' order (i-major lower triangle),
// value = d2 / 2 (multiples of 0.5 are exact in fp64).
__global__ void k_lev_dense(UmiArrays U, int n, double* out) {
    const long long p = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    const long long npairs = static_cast<long long>(n) * (n - 1) / 2;
    if (p >= npairs) return;
    // invert p -> (i, j), i < j, i-major
    long long i = static_cast<long long>((2.0 * n - 1 - sqrt((2.0 * n - 1) * (2.0 * n - 1) - 8.0 * p)) / 2);
    auto start = [&](long long r) { return r * (2LL * n - r - 1) / 2; };
    while (i > 0 && start(i) > p) --i;
    while (start(i + 1) <= p) ++i;
    const long long j = i + 1 + (p - start(i));
    const uint32_t ma = U.meta[i], mb = U.meta[j];
    const int d = banded_lev2<UMI_MAXLEN>(U.code[i], U.nmask[i], umi_len(ma), U.code[j], U.nmask[j], umi_len(mb), 4 * UMI_MAXLEN);
    out[p] = static_cast<double>(d) / 2.0;
}

// ---------------------------------------------------------------------------
// adjacency

// undirected rank pairs -> directed keys (orig_row << 32 | column rank), plus self links
__global__ void k_expand_edges(const unsigned long long* edges, unsigned long long m, const int* perm,
                               unsigned long long* keys) {
    const unsigned long long e = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x;
    if (e >= m) return;
    const unsigned ri = static_cast<unsigned>(edges[e] >> 32), rj = static_cast<unsigned>(edges[e]);
    keys[2 * e] = (static_cast<unsigned long long>(perm[ri]) << 32) | rj;
    keys[2 * e + 1] = (static_cast<unsigned long long>(perm[rj]) << 32) | ri;
}

// single: pre-groups of one read pass through untouched (src/umi_group.cpp:39-42): they always
// get their self link so that the clustering emits them as solos without any check
__global__ void k_self_flags(UmiArrays U, int n, int lim2, const uint8_t* single, const int* perm, int* flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const bool self = umi_nn(U.meta[r]) <= lim2;  // d2(x,x) = #N (App.B Q10)
    flag[r] = (self || (single && single[perm[r]])) ? 1 : 0;
}

__global__ void k_self_keys(const int* flag, const long long* pos, const int* perm, int n, unsigned long long* keys) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && flag[r]) keys[pos[r]] = (static_cast<unsigned long long>(perm[r]) << 32) | static_cast<unsigned>(r);
}

// sorted keys (row << 32 | x) -> row offsets
__global__ void k_row_offsets(const unsigned long long* keys, long long nk, int n, long long* off) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i > nk) return;
    const long long prev = (i == 0) ? -1 : static_cast<long long>(keys[i - 1] >> 32);
    const long long cur = (i == nk) ? n : static_cast<long long>(keys[i] >> 32);
    for (long long r = prev + 1; r <= cur; ++r) off[r] = i;
}

__global__ void k_cols_from_keys(const unsigned long long* keys, long long nk, const int* perm, int* nbr) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i < nk) nbr[i] = perm[static_cast<unsigned>(keys[i])];
}

// (row << 32 | rank) -> (row << 32 | orig col)
__global__ void k_keys_to_orig(const unsigned long long* keys, long long nk, const int* perm, unsigned long long* out) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i < nk) out[i] = (keys[i] & 0xffffffff00000000ull) | static_cast<unsigned>(perm[static_cast<unsigned>(keys[i])]);
}

// keep[i] = 1 if (row, perm2[rank]) of keys2[i] is present in the sorted set s1
__global__ void k_intersect_flags(const unsigned long long* keys2, long long nk2, const int* perm2,
                                  const unsigned long long* s1, long long n1, int* keep) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i >= nk2) return;
    const unsigned long long want = (keys2[i] & 0xffffffff00000000ull) | static_cast<unsigned>(perm2[static_cast<unsigned>(keys2[i])]);
    long long lo = 0, hi = n1;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (s1[mid] < want) lo = mid + 1; else hi = mid;
    }
    keep[i] = (lo < n1 && s1[lo] == want) ? 1 : 0;
}

__global__ void k_compact_keys(const unsigned long long* keys, const int* keep, const long long* pos, long long nk,
                               unsigned long long* out) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i < nk && keep[i]) out[pos[i]] = keys[i];
}

// ---------------------------------------------------------------------------
// greedy clustering in exact parallel rounds

struct ClusterState {
    const long long* off;
    const int* nbr;
    int n;
    int* remaining;
    int* state;                 // 0 live, 1 solo, 2 clustered
    int* mark;                  // round in which the node was clustered
    unsigned long long* key;
    unsigned long long* m1;
    int* seed;                  // 1 if picked as a seed (any round)
    unsigned long long* pickkey;
    int* memb;                  // members of the cluster seeded at v, stored at off[v]..
    int* csize;
    int* err;                   // [0] min index with empty list, [1] min index bad solo, [2] missing self / asymmetric
    int* live;                  // count of live pool nodes this round
};

__global__ void k_cl_init(ClusterState S, int check_sym) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    const long long a = S.off[v], b = S.off[v + 1];
    const int deg = static_cast<int>(b - a);
    S.remaining[v] = deg;
    S.seed[v] = 0; S.csize[v] = 0; S.mark[v] = -1; S.pickkey[v] = 0;
    int st = 0;
    if (deg == 0) { atomicMin(&S.err[0], v); st = 1; }
    else if (deg == 1) {
        if (S.nbr[a] != v) atomicMin(&S.err[1], v);
        st = 1;
    } else if (check_sym) {
        bool self = false;
        for (long long p = a; p < b; ++p) {
            const int w = S.nbr[p];
            if (w == v) self = true;
            else {  // symmetric?
                bool back = false;
                for (long long q = S.off[w]; q < S.off[w + 1] && !back; ++q) back = S.nbr[q] == v;
                if (!back) atomicMin(&S.err[2], v);
            }
        }
        if (!self) atomicMin(&S.err[2], v);
    }
    S.state[v] = st;
}

__global__ void k_cl_keys(ClusterState S) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    unsigned long long k = 0;
    if (S.state[v] == 0 && S.remaining[v] > 0) {
        k = (static_cast<unsigned long long>(S.remaining[v]) << 32) | static_cast<unsigned>(v);
        atomicAdd(S.live, 1);
    }
    S.key[v] = k;
}

__global__ void k_cl_m1(ClusterState S) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    unsigned long long m = 0;
    if (S.state[v] == 0)
        for (long long p = S.off[v]; p < S.off[v + 1]; ++p) m = max(m, S.key[S.nbr[p]]);
    S.m1[v] = m;
}

// A live node is picked when its key is the maximum over everything within two hops
// through live nodes: no earlier pick of the sequential greedy can touch it.
__global__ void k_cl_pick(ClusterState S, int round) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    const unsigned long long k = S.key[v];
    if (k == 0) return;
    unsigned long long m2 = k;
    for (long long p = S.off[v]; p < S.off[v + 1]; ++p) {
        const int w = S.nbr[p];
        if (S.state[w] == 0) m2 = max(m2, S.m1[w]);
    }
    if (m2 != k) return;
    // cluster = still-unused neighbours in list order (src/cluster_umis.cpp:78-91)
    int c = 0;
    const long long a = S.off[v];
    for (long long p = a; p < S.off[v + 1]; ++p) {
        const int w = S.nbr[p];
        if (S.state[w] == 0) { S.memb[a + c] = w; ++c; S.mark[w] = round; }
    }
    S.csize[v] = c;
    S.seed[v] = 1;
    S.pickkey[v] = k;
}

// The same three passes with one wavefront per node, for dense neighbourhoods (threshold 3 on 12-base UMIs: hundreds
// of neighbours per node, thousands for some -- one thread per node would walk them alone): the lanes stride over the
// node's list, so the list is read coalesced and the neighbours' words are gathered 64 at a time.
__device__ __forceinline__ unsigned long long cl_wave_max64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = static_cast<unsigned>(__shfl_xor(static_cast<int>(v), d));
        const unsigned hi = static_cast<unsigned>(__shfl_xor(static_cast<int>(v >> 32), d));
        const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

__global__ void __launch_bounds__(256) k_cl_m1_w(ClusterState S) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= S.n) return;
    unsigned long long m = 0;
    if (S.state[v] == 0)
        for (long long p = S.off[v] + lane; p < S.off[v + 1]; p += 64) m = max(m, S.key[S.nbr[p]]);
    m = cl_wave_max64(m);
    if (lane == 0) S.m1[v] = m;
}

__global__ void __launch_bounds__(256) k_cl_pick_w(ClusterState S, int round) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= S.n) return;
    const unsigned long long k = S.key[v];
    if (k == 0) return;
    const long long a = S.off[v], b = S.off[v + 1];
    unsigned long long m2 = k;
    for (long long p = a + lane; p < b; p += 64) {
        const int w = S.nbr[p];
        if (S.state[w] == 0) m2 = max(m2, S.m1[w]);
    }
    m2 = cl_wave_max64(m2);
    if (m2 != k) return;
    // cluster = still-unused neighbours in list order (src/cluster_umis.cpp:78-91)
    int c = 0;
    for (long long p0 = a; p0 < b; p0 += 64) {
        const long long p = p0 + lane;
        const int w = p < b ? S.nbr[p] : -1;
        const bool live = w >= 0 && S.state[w] == 0;
        const unsigned long long ball = __ballot(live);
        if (live) {
            S.memb[a + c + __popcll(ball & ((1ull << lane) - 1ull))] = w;
            S.mark[w] = round;
        }
        c += __popcll(ball);
    }
    if (lane == 0) { S.csize[v] = c; S.seed[v] = 1; S.pickkey[v] = k; }
}

__global__ void __launch_bounds__(256) k_cl_decrement_w(ClusterState S, int round) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= S.n || S.mark[v] != round) return;
    for (long long p = S.off[v] + lane; p < S.off[v + 1]; p += 64) {
        const int x = S.nbr[p];
        if (S.state[x] == 0) atomicSub(&S.remaining[x], 1);
    }
}

// Dense graphs (threshold 3 on 12-base UMIs: the 2-hop ball of a node covers a large share of the graph) yield only a
// few picks per round, all of them among the nodes with the largest keys, while the two passes above walk every live
// list.  A round can instead be decided on a candidate set C = {live v : remaining[v] >= t}: C is closed upwards in key
// order, so a candidate is a 2-hop maximum of the whole graph exactly when no other CANDIDATE with a larger key lies
// within two hops -- hop1[w] = max key over candidates adjacent to w (lists are symmetric), and v is picked iff the
// maximum of hop1 over its live neighbours is its own key.  Picks outside C are left for a later round (every pick is
// valid on its own), so the clusters are those of the full rounds; the work per round is |C| lists instead of all.
struct ClusterTop {
    int* maxrem;                // largest remaining of a live node, this round
    int* cand;                  // candidate list
    int* counts;                // [0] candidates, [1] picks, [2] 1 when the list was cut off at `cap`, [3] nodes clustered this round
    unsigned long long* hop1;
    int cap;
    int* marked;                // the nodes clustered this round (two picks of a round share no neighbour: no node twice)
    int* ctl;                   // the rounds' control block, see k_cl_control
};
// Control block of the candidate-set rounds.  The decisions between two rounds -- is anything left, did the last candidate
// list overflow, how wide is the next one -- need three counters of the round before; taken on the host they cost one
// read-back per round (190 us per round with its six launches, 330 rounds at threshold 3 on 10^6 12-base UMIs).  k_cl_control
// takes them on the device, every kernel of a round looks at the mode first, and the host enqueues CL_GROUP rounds at a time.
enum { CTL_MODE, CTL_DELTA, CTL_WAS_TOP, CTL_ROUNDS, CTL_MAXREM, CTL_N = 8 };
enum { CL_DONE = 0, CL_TOP = 1, CL_WANTS_FULL = 3 };   // CL_WANTS_FULL: the list was cut off -- the host runs one round over every list
constexpr int CL_GROUP = 8;

__global__ void __launch_bounds__(1024) k_cl_keys_top(ClusterState S, ClusterTop T) {
    __shared__ int s_live[16], s_max[16];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    int rem = 0;
    if (v < S.n && S.state[v] == 0 && S.remaining[v] > 0) rem = S.remaining[v];
    if (v < S.n) {
        S.key[v] = rem ? ((static_cast<unsigned long long>(rem) << 32) | static_cast<unsigned>(v)) : 0ull;
        if (T.ctl) T.hop1[v] = 0ull;   // (device-controlled rounds: the round's maxima start here instead of in a memset)
    }
    const unsigned long long live = __ballot(rem > 0);
    int m = rem;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) { s_live[threadIdx.x >> 6] = __popcll(live); s_max[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nl = 0, mm = 0;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) { nl += s_live[w]; mm = max(mm, s_max[w]); }
        if (nl) { atomicAdd(S.live, nl); atomicMax(T.maxrem, mm); }
    }
}

// One thread between the key pass and the candidate pass of a round: the host loop's decisions (see ClusterTop).
__global__ void k_cl_control(ClusterState S, ClusterTop T) {
    int* const c = T.ctl;
    const int live = *S.live;
    c[CTL_MAXREM] = *T.maxrem;
    *S.live = 0; *T.maxrem = 0;
    if (c[CTL_MODE] != CL_TOP) return;   // finished, or waiting for the host's round over every list
    if (live == 0) { c[CTL_MODE] = CL_DONE; return; }
    int delta = c[CTL_DELTA];
    if (c[CTL_WAS_TOP]) {
        // the candidate list of the previous round: cut off -> a round over every list and the window shrinks;
        // a productive list (a quarter or more of it picked) may be hiding picks just below it -> widen
        const long long nc = T.counts[0], np = T.counts[1];
        if (T.counts[2]) { c[CTL_DELTA] = delta / 2; c[CTL_MODE] = CL_WANTS_FULL; c[CTL_WAS_TOP] = 0; return; }
        if (4 * np >= nc) delta = 2 * delta + 1;
        else if (64 * np < nc && nc > 256) delta /= 2;
    }
    c[CTL_DELTA] = delta; c[CTL_WAS_TOP] = 1; c[CTL_ROUNDS] += 1;
    T.counts[0] = T.counts[1] = T.counts[2] = T.counts[3] = 0;
}

__global__ void k_cl_collect(ClusterState S, ClusterTop T, int delta) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (T.ctl && T.ctl[CTL_MODE] != CL_TOP) return;
    const int t = T.ctl ? max(1, T.ctl[CTL_MAXREM] - T.ctl[CTL_DELTA]) : max(1, *T.maxrem - delta);
    const bool in = v < S.n && static_cast<int>(S.key[v] >> 32) >= t;
    const unsigned long long ball = __ballot(in);
    if (!ball) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&T.counts[0], __popcll(ball));
    base = __shfl(base, 0);
    if (in) {
        const int slot = base + __popcll(ball & ((1ull << lane) - 1ull));
        if (slot < T.cap) T.cand[slot] = v; else atomicExch(&T.counts[2], 1);
    }
}

__global__ void __launch_bounds__(256) k_cl_mark_top(ClusterState S, ClusterTop T) {
    if (T.ctl && T.ctl[CTL_MODE] != CL_TOP) return;
    if (T.counts[2]) return;
    const int nc = T.counts[0], lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nc; c += gridDim.x * 4) {
        const int v = T.cand[c];
        const unsigned long long k = S.key[v];
        for (long long p = S.off[v] + lane; p < S.off[v + 1]; p += 64) {
            const int w = S.nbr[p];
            if (S.state[w] == 0) atomicMax(&T.hop1[w], k);
        }
    }
}

__global__ void __launch_bounds__(256) k_cl_pick_top(ClusterState S, ClusterTop T, int round) {
    if (T.ctl && T.ctl[CTL_MODE] != CL_TOP) return;
    if (T.counts[2]) return;
    const int nc = T.counts[0], lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nc; c += gridDim.x * 4) {
        const int v = T.cand[c];
        const unsigned long long k = S.key[v];
        const long long a = S.off[v], b = S.off[v + 1];
        unsigned long long m2 = k;
        for (long long p = a + lane; p < b; p += 64) {
            const int w = S.nbr[p];
            if (S.state[w] == 0) m2 = max(m2, T.hop1[w]);
        }
        m2 = cl_wave_max64(m2);
        if (m2 != k) continue;
        // cluster = still-unused neighbours in list order (src/cluster_umis.cpp:78-91)
        int cnt = 0;
        for (long long p0 = a; p0 < b; p0 += 64) {
            const long long p = p0 + lane;
            const int w = p < b ? S.nbr[p] : -1;
            const bool live = w >= 0 && S.state[w] == 0;
            const unsigned long long ball = __ballot(live);
            int lbase = 0;
            if (T.marked && ball) {
                if (lane == 0) lbase = atomicAdd(&T.counts[3], __popcll(ball));
                lbase = __shfl(lbase, 0);
            }
            if (live) {
                const int before = __popcll(ball & ((1ull << lane) - 1ull));
                S.memb[a + cnt + before] = w;
                S.mark[w] = round;
                if (T.marked) T.marked[lbase + before] = w;
            }
            cnt += __popcll(ball);
        }
        if (lane == 0) { S.csize[v] = cnt; S.seed[v] = 1; S.pickkey[v] = k; atomicAdd(&T.counts[1], 1); }
    }
}

// commit and decrement of a device-controlled round, over the list of the nodes it clustered instead of over all nodes
__global__ void __launch_bounds__(256) k_cl_commit_list(ClusterState S, ClusterTop T) {
    if (T.ctl[CTL_MODE] != CL_TOP) return;
    const int nm = T.counts[3];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nm; q += gridDim.x * blockDim.x) {
        const int w = T.marked[q];
        S.state[w] = 2; S.remaining[w] = 0;
    }
}

__global__ void __launch_bounds__(256) k_cl_decrement_list(ClusterState S, ClusterTop T) {
    if (T.ctl[CTL_MODE] != CL_TOP) return;
    const int nm = T.counts[3], lane = threadIdx.x & 63;
    for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nm; q += gridDim.x * 4) {
        const int v = T.marked[q];
        for (long long p = S.off[v] + lane; p < S.off[v + 1]; p += 64) {
            const int x = S.nbr[p];
            if (S.state[x] == 0) atomicSub(&S.remaining[x], 1);
        }
    }
}

// state flips happen in a separate pass so that k_cl_pick sees a consistent snapshot
__global__ void k_cl_commit(ClusterState S, int round) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    if (S.mark[v] == round) { S.state[v] = 2; S.remaining[v] = 0; }
}

__global__ void k_cl_decrement(ClusterState S, int round) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n || S.mark[v] != round) return;
    for (long long p = S.off[v]; p < S.off[v + 1]; ++p) {
        const int x = S.nbr[p];
        if (S.state[x] == 0) atomicSub(&S.remaining[x], 1);
    }
}

// output assembly
__global__ void k_cl_flags(ClusterState S, int* is_solo, int* is_seed) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= S.n) return;
    is_solo[v] = (S.state[v] == 1) ? 1 : 0;
    is_seed[v] = S.seed[v];
}

// Clusters in one list: solos get key = index (top bit clear), picks key = ~pickkey (top bit
// set): an ascending sort lists solos by index, then picks by (remaining, index) descending.
__global__ void k_cl_list(const int* is_solo, const long long* spos, const int* is_seed, const long long* kpos,
                          long long nsolo, const unsigned long long* pickkey, int n, unsigned long long* sortkey, int* val) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (is_solo[v]) { sortkey[spos[v]] = static_cast<unsigned long long>(v); val[spos[v]] = v; }
    if (is_seed[v]) { sortkey[nsolo + kpos[v]] = ~pickkey[v]; val[nsolo + kpos[v]] = v; }
}

__global__ void k_cl_gidkey(const int* gid, const int* val, long long n, unsigned long long* key) {
    const long long c = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (c < n) key[c] = static_cast<unsigned long long>(gid[val[c]]);
}

__global__ void k_cl_sizes(ClusterState S, const int* order, long long nclu, int* sizes) {
    const long long c = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (c < nclu) sizes[c] = S.seed[order[c]] ? S.csize[order[c]] : 1;
}

__global__ void k_cl_write(ClusterState S, const int* order, long long nclu, const long long* coff,
                           const int32_t* members /* optional 1-based ids */, int32_t* out) {
    const long long c = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (c >= nclu) return;
    const int v = order[c];
    const long long o = coff[c];
    if (!S.seed[v]) { out[o] = members ? members[v] : v + 1; return; }
    const long long a = S.off[v];
    for (int k = 0; k < S.csize[v]; ++k) {
        const int w = S.memb[a + k];
        out[o + k] = members ? members[w] : w + 1;
    }
}

// ---------------------------------------------------------------------------
// host orchestration

static inline unsigned nblk(long long n, int bs) { return static_cast<unsigned>((n + bs - 1) / bs); }

static int sort_keys_u64(const char* tag, unsigned long long* in, unsigned long long* out, size_t n, int bits, hipStream_t s) {
    size_t tmp = 0;
    SL_HIP(rocprim::radix_sort_keys(nullptr, tmp, in, out, n, 0, bits, s));
    void* d_tmp;
    SL_TRY(ctx().buffer((std::string(tag) + ".sorttmp").c_str(), tmp ? tmp : 16, &d_tmp));
    SL_HIP(rocprim::radix_sort_keys(d_tmp, tmp, in, out, n, 0, bits, s));
    return 0;
}

static int sort_pairs_u64_i32(const char* tag, unsigned long long* kin, unsigned long long* kout, int* vin, int* vout,
                              size_t n, int bits, hipStream_t s) {
    size_t tmp = 0;
    SL_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, s));
    void* d_tmp;
    SL_TRY(ctx().buffer((std::string(tag) + ".sorttmp").c_str(), tmp ? tmp : 16, &d_tmp));
    SL_HIP(rocprim::radix_sort_pairs(d_tmp, tmp, kin, kout, vin, vout, n, 0, bits, s));
    return 0;
}

static int exclusive_scan_i32(const char* tag, const int* in, long long* out, size_t n, hipStream_t s) {
    size_t tmp = 0;
    SL_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, 0ll, n, rocprim::plus<long long>(), s));
    void* d_tmp;
    SL_TRY(ctx().buffer((std::string(tag) + ".scantmp").c_str(), tmp ? tmp : 16, &d_tmp));
    SL_HIP(rocprim::exclusive_scan(d_tmp, tmp, in, out, 0ll, n, rocprim::plus<long long>(), s));
    return 0;
}

static int ceil_log2(unsigned long long x) {
    int b = 1;
    while (b < 64 && (1ull << b) < x) ++b;
    return b;
}

static int alloc_umi(const std::string& p, size_t n, UmiArrays* U, int words = 1) {
    U->stride = static_cast<long long>(n);
    SL_TRY(scratch((p + ".code").c_str(), n * static_cast<size_t>(words), &U->code));
    SL_TRY(scratch((p + ".nmask").c_str(), n * static_cast<size_t>(words), &U->nmask));
    SL_TRY(scratch((p + ".comp").c_str(), n, &U->comp));
    SL_TRY(scratch((p + ".meta").c_str(), n, &U->meta));
    return 0;
}

struct SortedUmis {
    UmiArrays U;   // in (pre-group, trie) order
    int* perm;     // rank -> local index
    int* gid;      // pre-group per rank (nullptr: a single group)
    int n;
    int words;     // 1: every string has at most 32 bases; UMI_LONG_WORDS up to 128 bases; beyond, what the longest string needs
    int ngroups;   // pre-groups (1 when gid is nullptr)
    int nskip = 0;       // elements that are never compared (pre-groups of one read: encoded as empty strings)
    int max_group = 0;   // size of the largest pre-group (0: unknown, the whole set)
};

// Encode one set of UMIs (optionally the members of a pre-group) and order it like the trie.
static int encode_and_rank(const std::string& p, const uint8_t* d_chars, const int64_t* d_off, const int32_t* d_members,
                           const int* d_gid, int ngroups, int n, SortedUmis* out, hipStream_t s, const uint8_t* d_skip = nullptr,
                           int nskip = 0, int max_group = 0) {
    out->nskip = d_skip ? nskip : 0;
    out->max_group = max_group > 0 ? max_group : n;
    UmiArrays raw;
    SL_TRY(alloc_umi(p + ".raw", n, &raw));
    SL_TRY(alloc_umi(p + ".srt", n, &out->U));
    unsigned long long *khi, *klo, *k2;
    int *idx, *idx2, *bad;
    SL_TRY(scratch((p + ".khi").c_str(), n, &khi));
    SL_TRY(scratch((p + ".klo").c_str(), n, &klo));
    SL_TRY(scratch((p + ".k2").c_str(), n, &k2));
    SL_TRY(scratch((p + ".idx").c_str(), n, &idx));
    SL_TRY(scratch((p + ".idx2").c_str(), n, &idx2));
    SL_TRY(scratch((p + ".bad").c_str(), 3, &bad));
    const int init[3] = {std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), 0};
    SL_HIP(hipMemcpyAsync(bad, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_umi_encode, dim3(nblk(n, 256)), dim3(256), 0, s, d_chars, d_off, d_members, n, raw, khi, klo, idx, d_skip, bad);
    SL_HIP(hipGetLastError());
    int hbad[3];
    SL_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    out->words = 1;
    if (hbad[1] != init[1]) {
        // some string has more than 32 bases: the whole call runs on 4-word codes
        const int maxlen = -hbad[2];
        if (maxlen > UMI_XL_MAX) return fail("sarlacc_amd: UMI longer than %d bases is not supported", UMI_XL_MAX);
        // 33..128 bases: 4 words per string, tiles staged in LDS; beyond: as many words as the longest string needs
        out->words = maxlen <= UMI_LONG_MAX ? UMI_LONG_WORDS : (maxlen + 31) / 32;
        const int nkeys = (maxlen + UMI_KEY_BASES - 1) / UMI_KEY_BASES;
        SL_TRY(alloc_umi(p + ".rawL", n, &raw, out->words));
        SL_TRY(alloc_umi(p + ".srtL", n, &out->U, out->words));
        unsigned long long* keys;
        SL_TRY(scratch((p + ".keysL").c_str(), static_cast<size_t>(n) * nkeys, &keys));
        SL_HIP(hipMemcpyAsync(bad, init, sizeof init, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_umi_encode_long, dim3(nblk(n, 256)), dim3(256), 0, s, d_chars, d_off, d_members, n, raw, keys, idx, d_skip, bad, out->words, nkeys);
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        if (hbad[0] != init[0])
            return fail("sarlacc_amd: UMI contains a character outside ACGTN (the reference silently drops such strings)");
        // stable sorts, least-significant key first (only the keys some string reaches)
        int *from = idx, *to = idx2;
        for (int k = nkeys - 1; k >= 0; --k) {
            hipLaunchKernelGGL(k_gather_u64, dim3(nblk(n, 256)), dim3(256), 0, s, keys + static_cast<size_t>(k) * n, from, klo, n);
            SL_TRY(sort_pairs_u64_i32(p.c_str(), klo, k2, from, to, n, 63, s));
            std::swap(from, to);
        }
        if (from != idx) SL_HIP(hipMemcpyAsync(idx, from, sizeof(int) * static_cast<size_t>(n), hipMemcpyDeviceToDevice, s));
    } else {
        if (hbad[0] != init[0])
            return fail("sarlacc_amd: UMI contains a character outside ACGTN (the reference silently drops such strings)");
        // least-significant key first; both sorts are stable, ties keep the input order
        SL_TRY(sort_pairs_u64_i32(p.c_str(), klo, k2, idx, idx2, n, 64, s));
        hipLaunchKernelGGL(k_gather_u64, dim3(nblk(n, 256)), dim3(256), 0, s, khi, idx2, klo, n);
        SL_TRY(sort_pairs_u64_i32(p.c_str(), klo, k2, idx2, idx, n, 64, s));
    }
    out->gid = nullptr;
    out->ngroups = 1;
    if (d_gid && ngroups > 1) {  // most significant key: the pre-group
        out->ngroups = ngroups;
        int* gsorted;
        SL_TRY(scratch((p + ".gid").c_str(), n, &gsorted));
        hipLaunchKernelGGL(k_gather_gid, dim3(nblk(n, 256)), dim3(256), 0, s, d_gid, idx, klo, static_cast<int*>(nullptr), n);
        SL_TRY(sort_pairs_u64_i32(p.c_str(), klo, k2, idx, idx2, n, ceil_log2(static_cast<unsigned long long>(ngroups) + 1), s));
        SL_HIP(hipMemcpyAsync(idx, idx2, sizeof(int) * static_cast<size_t>(n), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_gather_gid, dim3(nblk(n, 256)), dim3(256), 0, s, d_gid, idx, static_cast<unsigned long long*>(nullptr), gsorted, n);
        out->gid = gsorted;
    }
    hipLaunchKernelGGL(k_gather_umi, dim3(nblk(n, 256)), dim3(256), 0, s, raw, idx, out->U, n, out->words);
    SL_HIP(hipGetLastError());
    out->perm = idx;
    out->n = n;
    return 0;
}

static int exclusive_scan_i64(const char* tag, const long long* in, long long* out, size_t n, hipStream_t s) {
    size_t tmp = 0;
    SL_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, 0ll, n, rocprim::plus<long long>(), s));
    void* d_tmp;
    SL_TRY(ctx().buffer((std::string(tag) + ".scantmp").c_str(), tmp ? tmp : 16, &d_tmp));
    SL_HIP(rocprim::exclusive_scan(d_tmp, tmp, in, out, 0ll, n, rocprim::plus<long long>(), s));
    return 0;
}

// ---- split-key search: host side ----
constexpr int SK_MIN_N = 32768;          // below this (or with pre-groups averaging under half of it) the all-tile-pairs search is quick enough

constexpr int SK_MIN_LEN = 8;            // shorter strings are "special" (keys under 4 bases select too much)
constexpr int SK_MAX_KEY = 8;            // bases per key at most (9 bases of the forward string travel with every element)

struct SkClass { int len_lo, len_hi, h, s; };

struct SkPlan {
    std::vector<SkClass> classes;   // row lengths present in the set, with the key lengths they scan with
    int k1 = 0, k2 = 0;             // edits allowed in the prefix / suffix key
    long long nspecial = 0;
};

struct SkScan {          // one scan order with row groups for one key length, ready to launch
    SkOrder O{};
    int key = 0;
    int* rg_start = nullptr;
    int nrg = 0;
    int2* ranges = nullptr;
    int *nranges = nullptr, *ctotal = nullptr;
    long long* item_off = nullptr;
    long long nitems = 0;
};

// The lengths present decide the scans: one class per length 8..15 and one for 16 and longer.
static int sk_plan(const std::string& p, const SortedUmis& S, int limit, SkPlan* plan, hipStream_t s) {
    unsigned int* d_hist;
    SL_TRY(scratch((p + ".sk.hist").c_str(), 34, &d_hist));
    SL_HIP(hipMemsetAsync(d_hist, 0, 34 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(k_sk_lenhist, dim3(std::min(nblk(S.n, 256), 1024u)), dim3(256), 0, s, S.U, S.n, d_hist);
    unsigned int hist[34];
    SL_HIP(hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    plan->classes.clear();
    for (int L = SK_MIN_LEN; L <= 2 * SK_MAX_KEY; ++L) {
        long long rows = hist[L];
        if (L == 2 * SK_MAX_KEY) for (int M = L + 1; M <= 32; ++M) rows += hist[M];
        if (!rows) continue;
        const int h = std::min(L / 2, SK_MAX_KEY);
        plan->classes.push_back(SkClass{L, L == 2 * SK_MAX_KEY ? 32 : L, h, std::min(L - h, SK_MAX_KEY)});
    }
    plan->k1 = limit >= 2 ? 1 : 0;   // limit 1: both keys have to match exactly; 2: the suffix key; 3: one edit in either
    plan->k2 = limit - 1 - plan->k1;
    long long shorter = 0;
    for (int L = 0; L < SK_MIN_LEN; ++L) shorter += hist[L];
    plan->nspecial = shorter + hist[33];   // an upper bound (short strings with an N count twice)
    return 0;
}

static int sk_prepare(const std::string& p, const SkOrder& O, int h, int k1, bool clip, SkScan* out, hipStream_t s) {
    const int n = O.n;
    int* d_flag; long long* d_pos;
    SL_TRY(scratch((p + ".flag").c_str(), static_cast<size_t>(n) + 1, &d_flag));
    SL_TRY(scratch((p + ".pos").c_str(), static_cast<size_t>(n) + 1, &d_pos));
    hipLaunchKernelGGL(k_sk_group_flags, dim3(nblk(n + 1, 256)), dim3(256), 0, s, O, h, d_flag);
    SL_TRY(exclusive_scan_i32(p.c_str(), d_flag, d_pos, static_cast<size_t>(n) + 1, s));
    long long nrg = 0;
    SL_HIP(hipMemcpyAsync(&nrg, d_pos + n, sizeof nrg, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    out->O = O;
    out->key = h;
    out->nrg = static_cast<int>(nrg);
    SL_TRY(scratch((p + ".start").c_str(), static_cast<size_t>(nrg) + 1, &out->rg_start));
    SL_TRY(scratch((p + ".ranges").c_str(), static_cast<size_t>(nrg) * SK_MAXR + 1, &out->ranges));
    SL_TRY(scratch((p + ".nranges").c_str(), static_cast<size_t>(nrg) + 1, &out->nranges));
    SL_TRY(scratch((p + ".ctotal").c_str(), static_cast<size_t>(nrg) + 1, &out->ctotal));
    long long* d_items;
    SL_TRY(scratch((p + ".items").c_str(), static_cast<size_t>(nrg) + 1, &d_items));
    SL_TRY(scratch((p + ".itemoff").c_str(), static_cast<size_t>(nrg) + 1, &out->item_off));
    hipLaunchKernelGGL(k_sk_group_starts, dim3(nblk(n + 1, 256)), dim3(256), 0, s, d_flag, d_pos, n, out->rg_start);
    hipLaunchKernelGGL(k_sk_ranges, dim3(nblk(nrg + 1, 64)), dim3(64), 0, s, O, out->rg_start, out->nrg, h, k1, clip ? 1 : 0,
                       out->ranges, out->nranges, out->ctotal, d_items);
    SL_HIP(hipGetLastError());
    SL_TRY(exclusive_scan_i64(p.c_str(), d_items, out->item_off, static_cast<size_t>(nrg) + 1, s));
    SL_HIP(hipMemcpyAsync(&out->nitems, out->item_off + nrg, sizeof(long long), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return 0;
}

// Both scan orders with their row groups and candidate ranges.
static int sk_build(const std::string& p, const SortedUmis& S, const SkPlan& plan, std::vector<SkScan>* fwd, std::vector<SkScan>* rev, hipStream_t s) {
    const int n = S.n;
    SkOrder X{}, Y{};
    X.n = Y.n = n;
    X.gid = S.gid;
    SL_TRY(scratch((p + ".sk.elx").c_str(), static_cast<size_t>(n), &X.el));
    SL_TRY(scratch((p + ".sk.keyx").c_str(), static_cast<size_t>(n), &X.okey));
    hipLaunchKernelGGL(k_sk_elems<false>, dim3(nblk(n, 256)), dim3(256), 0, s, S.U, n, SK_MIN_LEN, X.el, X.okey, static_cast<int*>(nullptr));
    // reversed strings: sort by key (then by pre-group, stably), carry the trie rank
    SkElem* d_tmp; unsigned long long *d_rk, *d_rk2; int *d_val, *d_val2;
    SL_TRY(scratch((p + ".sk.eltmp").c_str(), static_cast<size_t>(n), &d_tmp));
    SL_TRY(scratch((p + ".sk.rk").c_str(), static_cast<size_t>(n), &d_rk));
    SL_TRY(scratch((p + ".sk.rk2").c_str(), static_cast<size_t>(n), &d_rk2));
    SL_TRY(scratch((p + ".sk.val").c_str(), static_cast<size_t>(n), &d_val));
    SL_TRY(scratch((p + ".sk.val2").c_str(), static_cast<size_t>(n), &d_val2));
    SL_TRY(scratch((p + ".sk.ely").c_str(), static_cast<size_t>(n), &Y.el));
    hipLaunchKernelGGL(k_sk_elems<true>, dim3(nblk(n, 256)), dim3(256), 0, s, S.U, n, SK_MIN_LEN, d_tmp, d_rk, d_val);
    SL_HIP(hipGetLastError());
    SL_TRY(sort_pairs_u64_i32(p.c_str(), d_rk, d_rk2, d_val, d_val2, static_cast<size_t>(n), 63, s));
    Y.okey = d_rk2;
    int* order = d_val2;
    if (S.gid) {
        unsigned long long *d_gk, *d_gk2;
        SL_TRY(scratch((p + ".sk.gk").c_str(), static_cast<size_t>(n), &d_gk));
        SL_TRY(scratch((p + ".sk.gk2").c_str(), static_cast<size_t>(n), &d_gk2));
        hipLaunchKernelGGL(k_cl_gidkey, dim3(nblk(n, 256)), dim3(256), 0, s, S.gid, d_val2, static_cast<long long>(n), d_gk);
        SL_TRY(sort_pairs_u64_i32(p.c_str(), d_gk, d_gk2, d_val2, d_val, static_cast<size_t>(n), ceil_log2(static_cast<unsigned long long>(S.ngroups) + 1), s));
        order = d_val;
        // keys in the final order
        hipLaunchKernelGGL(k_sk_elems<true>, dim3(nblk(n, 256)), dim3(256), 0, s, S.U, n, SK_MIN_LEN, d_tmp, d_rk, d_val2);
        hipLaunchKernelGGL(k_gather_u64, dim3(nblk(n, 256)), dim3(256), 0, s, d_rk, order, d_rk2, n);
        SL_TRY(scratch((p + ".sk.gidy").c_str(), static_cast<size_t>(n), &Y.gid));
    }
    hipLaunchKernelGGL(k_sk_permute, dim3(nblk(n, 256)), dim3(256), 0, s, d_tmp, order, S.gid, n, Y.el, Y.gid);
    SL_HIP(hipGetLastError());
    // row groups and ranges once per key length in use; the prefix scan starts at the row group itself (columns
    // ranked below the row report the pair themselves)
    const bool clip = true;
    fwd->clear(); rev->clear();
    for (const SkClass& c : plan.classes) {
        bool have = false;
        for (const SkScan& q : *fwd) have = have || q.key == c.h;
        if (!have) { fwd->emplace_back(); SL_TRY(sk_prepare(p + ".skx" + std::to_string(c.h), X, c.h, plan.k1, clip, &fwd->back(), s)); }
        have = false;
        for (const SkScan& q : *rev) have = have || q.key == c.s;
        if (!have) { rev->emplace_back(); SL_TRY(sk_prepare(p + ".sky" + std::to_string(c.s), Y, c.s, plan.k2, false, &rev->back(), s)); }
    }
    return 0;
}

static long long sk_launch(const std::vector<SkScan>& F, const std::vector<SkScan>& R, const SkPlan& plan, int limit, uint32_t row_lo, uint32_t row_hi,
                           unsigned stride, unsigned long long* edges, unsigned long long* count, unsigned long long cap, hipStream_t s) {
    long long items = 0;
    ctx().counts["umi_scan_two_columns"] = 0;   // launches of k_sk_scan_pk
    for (const SkClass& c : plan.classes)
        for (int pass = 0; pass < 2; ++pass) {
            const SkScan* Q = nullptr;
            for (const SkScan& q : pass ? R : F) if (q.key == (pass ? c.s : c.h)) Q = &q;
            if (!Q) continue;
            const long long blocks = (Q->nitems + stride - 1) / stride;
            if (blocks <= 0) continue;
            items += Q->nitems;
            SkScanArgs a{Q->O.el, Q->rg_start, Q->nrg, Q->ranges, Q->nranges, Q->ctotal, Q->item_off, stride, limit, c.h, plan.k1,
                         c.len_lo, c.len_hi, row_lo, row_hi, edges, count, cap};
            // two candidate columns per lane for the rows no string of 16 and more bases can be a neighbour of (the lengths of
            // neighbours differ by `limit` at most; such columns are simply not valid there)
            const bool two = c.len_hi + limit < 16 && !option(OPT_UMI_SCAN_SINGLE);
            if (two) ctx().counts["umi_scan_two_columns"] += 1;
            if (two) {
                if (pass) hipLaunchKernelGGL(k_sk_scan_pk<true>, dim3(static_cast<unsigned>(blocks)), dim3(SK_COLS / 2), 0, s, a);
                else hipLaunchKernelGGL(k_sk_scan_pk<false>, dim3(static_cast<unsigned>(blocks)), dim3(SK_COLS / 2), 0, s, a);
            } else if (pass) hipLaunchKernelGGL(k_sk_scan<true>, dim3(static_cast<unsigned>(blocks)), dim3(SK_COLS), 0, s, a);
            else hipLaunchKernelGGL(k_sk_scan<false>, dim3(static_cast<unsigned>(blocks)), dim3(SK_COLS), 0, s, a);
        }
    return items;
}

template <int K>
static void launch_pairs(const PairArgs& a, int tile_hi, unsigned ntiles_listed, hipStream_t s) {
    const unsigned nt = nblk(a.n, TILE);
    if (tile_hi <= a.tile_lo) return;
    if (a.tile_list) {
        if (ntiles_listed) hipLaunchKernelGGL(k_umi_pairs<K>, dim3(ntiles_listed), dim3(TILE), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(k_umi_pairs<K>, dim3(static_cast<unsigned>(tile_hi - a.tile_lo), nt), dim3(TILE), 0, s, a);
}

template <int L>
static void launch_tile_pairs(const TileInfo* info, int nt, int tile_lo, int tile_hi, int special_only, uint32_t* list, unsigned int* count, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_pairs<L>, dim3(nblk(nt, 256), static_cast<unsigned>(tile_hi - tile_lo)), dim3(256), 0, s, info, nt,
                       tile_lo, tile_hi, special_only, list, count);
}

struct DirectedKeys {
    unsigned long long* keys;  // sorted (orig_row << 32 | column rank)
    long long nk;
};

// All neighbour pairs within `limit`, as sorted directed keys (self links included).
// Undirected neighbour pairs (rank_i << 32 | rank_j) of the row tiles [tile_lo, tile_hi)
// (tile_hi < 0: all tiles).  The buffer stays on the device: *d_edges_out, *m_out.
// number of pairs the last sarlacc_dev_umi_pairs_shard of this thread left in the workspace (-1: none)
static thread_local long long g_shard_pairs = -1;

static int pair_edges(const std::string& p, const SortedUmis& S, int limit, int tile_lo, int tile_hi,
                      unsigned long long** d_edges_out, unsigned long long* m_out, hipStream_t s) {
    Context& c = ctx();
    const int n = S.n;
    const int nt = static_cast<int>(nblk(n, TILE));
    g_shard_pairs = -1;   // (the buffer a sarlacc_dev_umi_pairs_shard call left its pairs in is about to be reused)
    if (tile_hi < 0) { tile_lo = 0; tile_hi = nt; }
    tile_lo = std::max(0, std::min(tile_lo, nt));
    tile_hi = std::max(tile_lo, std::min(tile_hi, nt));
    const int lim2 = 2 * limit;
    unsigned long long* d_count;
    SL_TRY(scratch((p + ".ecount").c_str(), 1, &d_count));
    unsigned long long cap = std::max<unsigned long long>(1u << 20, 32ull * n);
    unsigned long long m = 0;
    unsigned long long* d_edges = nullptr;
    if (n > static_cast<long long>(65535) * TILE) return fail("sarlacc_amd: more than %d UMIs in one call", 65535 * TILE);
    {   // make sure a buffer exists even when nothing is launched
        void* pe;
        SL_TRY(c.buffer((p + ".edges").c_str(), cap * sizeof(unsigned long long), &pe));
        d_edges = static_cast<unsigned long long*>(pe);
    }
    // Thresholds 1 to 3 on a large set: candidates from the split keys (see k_sk_scan); the tile kernel then only
    // looks at the pairs with a "special" member, if there are any.
    SkPlan plan;
    std::vector<SkScan> fwd, rev;
    const int min_n = option(OPT_UMI_SPLIT_MIN) > 0 ? option(OPT_UMI_SPLIT_MIN) : SK_MIN_N;
    // (the largest pre-group decides, not the average: a set with one large pre-group among thousands of small ones has the
    // quadratic tile search to lose; the reads that sit alone in their pre-group are encoded as empty strings, take no part
    // in any comparison and are not "special")
    bool split = S.words == 1 && limit >= 1 && limit <= 3 && n >= min_n && !option(OPT_UMI_TILE_SEARCH) && S.max_group >= min_n / 2;
    if (split) {
        SL_TRY(sk_plan(p, S, limit, &plan, s));
        plan.nspecial = std::max<long long>(0, plan.nspecial - S.nskip);
        split = !plan.classes.empty() && 4 * plan.nspecial <= n - S.nskip;
    }
    if (split) SL_TRY(sk_build(p, S, plan, &fwd, &rev, s));
    const int special_lreq = split ? SK_MIN_LEN : -1;
    long long split_items = 0;
    const bool tiles = !split || plan.nspecial > 0;
    c.counts["umi_split_search"] = split ? 1 : 0;
    c.counts["umi_split_classes"] = split ? static_cast<double>(plan.classes.size()) : 0;
    c.counts["umi_split_special"] = split ? static_cast<double>(plan.nspecial) : 0;
    // tile pairs that can hold neighbours (see k_tile_pairs); worth it from a few dozen tiles on
    const uint32_t* d_list = nullptr;
    const TileInfo* d_subinfo = nullptr;
    unsigned int nlisted = 0;
    const long long ntp = static_cast<long long>(tile_hi - tile_lo) * nt;
    if (tiles && S.words == 1 && limit >= 0 && limit <= 5 && nt >= 16 && ntp <= (1ll << 31)) {
        TileInfo* d_info; uint32_t* d_l; unsigned int* d_lc;
        SL_TRY(scratch((p + ".tinfo").c_str(), static_cast<size_t>(nt), &d_info));
        // the list is bounded by the upper triangle of the launch
        const size_t max_list = static_cast<size_t>(tile_hi - tile_lo) * static_cast<size_t>(nt);
        SL_TRY(scratch((p + ".tlist").c_str(), max_list, &d_l));
        SL_TRY(scratch((p + ".tcount").c_str(), 1, &d_lc));
        SL_HIP(hipMemsetAsync(d_lc, 0, sizeof(unsigned int), s));
        hipLaunchKernelGGL(k_tile_info<TILE>, dim3(static_cast<unsigned>(nt)), dim3(TILE), 0, s, S.U, S.gid, n, std::max(special_lreq, 0), d_info);
        // prefixes of the 64-element blocks for the sub-tile filter inside the pair kernel
        TileInfo* d_sub;
        const unsigned nsub = nblk(n, 64);
        SL_TRY(scratch((p + ".tsub").c_str(), static_cast<size_t>(nsub), &d_sub));
        hipLaunchKernelGGL(k_tile_info<64>, dim3(nsub), dim3(64), 0, s, S.U, S.gid, n, 0, d_sub);
        d_subinfo = d_sub;
        const int so = split ? 1 : 0;
        switch (limit) {
            case 0: launch_tile_pairs<0>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
            case 1: launch_tile_pairs<1>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
            case 2: launch_tile_pairs<2>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
            case 3: launch_tile_pairs<3>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
            case 4: launch_tile_pairs<4>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
            default: launch_tile_pairs<5>(d_info, nt, tile_lo, tile_hi, so, d_l, d_lc, s); break;
        }
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpyAsync(&nlisted, d_lc, sizeof nlisted, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        d_list = d_l;
    }
    const uint32_t row_lo = static_cast<uint32_t>(std::min<long long>(static_cast<long long>(tile_lo) * TILE, n));
    const uint32_t row_hi = static_cast<uint32_t>(std::min<long long>(static_cast<long long>(tile_hi) * TILE, n));
    // every search kernel of one pass over the set; `stride` > 1: a sample (every stride-th tile pair / work item)
    auto launch_all = [&](unsigned stride, unsigned long long capacity) {
        if (split && row_hi > row_lo) split_items = sk_launch(fwd, rev, plan, limit, row_lo, row_hi, stride, d_edges, d_count, capacity, s);
        if (!tiles) return;
        PairArgs a{S.U, S.gid, n, lim2, d_edges, d_count, capacity, tile_lo, d_list, d_subinfo, stride, special_lreq};
        const unsigned listed = d_list ? nlisted / stride : 0u;
        if (stride > 1 && !d_list) return;   // a sample needs the list
        const int K = std::min(limit, UMI_MAXLEN);
        if (S.words > 1) {
            if (tile_hi > tile_lo) {
                const dim3 grid(static_cast<unsigned>(tile_hi - tile_lo), static_cast<unsigned>(nt));
#define UMI_LONG_LAUNCH(KK)                                                                                  \
    {                                                                                                        \
        if (S.words > UMI_LONG_WORDS) hipLaunchKernelGGL((k_umi_pairs_long<KK, true>), grid, dim3(TILE), 0, s, a);   \
        else hipLaunchKernelGGL((k_umi_pairs_long<KK, false>), grid, dim3(TILE), 0, s, a);                   \
    }
                if (limit <= 0) UMI_LONG_LAUNCH(0)
                else if (limit == 1) UMI_LONG_LAUNCH(1)
                else if (limit == 2) UMI_LONG_LAUNCH(2)
                else if (limit == 3) UMI_LONG_LAUNCH(3)
                else if (limit <= 5) UMI_LONG_LAUNCH(5)
                else if (limit <= 8) UMI_LONG_LAUNCH(8)
                else if (limit <= 16) UMI_LONG_LAUNCH(16)
                else UMI_LONG_LAUNCH(-1)
#undef UMI_LONG_LAUNCH
            }
        }
        else if (K <= 0) launch_pairs<0>(a, tile_hi, listed, s);
        else if (K == 1) launch_pairs<1>(a, tile_hi, listed, s);
        else if (K == 2) launch_pairs<2>(a, tile_hi, listed, s);
        else if (K == 3) launch_pairs<3>(a, tile_hi, listed, s);
        else if (K == 4) launch_pairs<4>(a, tile_hi, listed, s);
        else if (K == 5) launch_pairs<5>(a, tile_hi, listed, s);
        else if (K <= 8) launch_pairs<8>(a, tile_hi, listed, s);
        else if (K <= 16) launch_pairs<16>(a, tile_hi, listed, s);
        else launch_pairs<UMI_MAXLEN>(a, tile_hi, listed, s);
    };
    // Capacity of the pair buffer: 32 per element covers thresholds 1 and 2; dense neighbourhoods (threshold 3 on 12-base
    // UMIs: hundreds of neighbours each) would overflow it and cost a second full search, so the density is first
    // estimated from a sample: every 32nd listed tile pair (k_tile_pairs appends them in no particular order) and
    // every 32nd work item of the split-key scans.
    const bool sample_tiles = d_list && nlisted >= 2048;
    long long all_items = 0;
    for (const SkScan& q : fwd) all_items += q.nitems;
    for (const SkScan& q : rev) all_items += q.nitems;
    const bool sample_items = split && all_items >= 2048;
    if (limit >= 0 && (sample_tiles || sample_items) && (!tiles || d_list)) {
        const unsigned stride = 32;
        SL_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
        launch_all(stride, cap);
        SL_HIP(hipGetLastError());
        unsigned long long ms = 0;
        SL_HIP(hipMemcpyAsync(&ms, d_count, sizeof ms, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        const double est = static_cast<double>(ms) * stride;
        cap = std::max<unsigned long long>(cap, static_cast<unsigned long long>(est * 1.25) + (1u << 20));
        ctx().counts["umi_pairs_estimated"] = est;
    }
    for (int attempt = 0; attempt < 2 && limit >= 0; ++attempt) {
        void* pe;
        SL_TRY(c.buffer((p + ".edges").c_str(), cap * sizeof(unsigned long long), &pe));
        d_edges = static_cast<unsigned long long*>(pe);
        SL_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
        SL_HIP(hipEventRecord(c.ev_start, s));
        c.stage_reset("umi_pairs");
        SL_TRY(c.stage_begin("umi_pairs", s));
        launch_all(1u, cap);
        SL_HIP(hipGetLastError());
        SL_HIP(hipEventRecord(c.ev_stop, s));
        SL_TRY(c.stage_end("umi_pairs", s));
        c.timed = true;
        SL_HIP(hipMemcpyAsync(&m, d_count, sizeof m, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        ctx().counts["umi_pair_attempts"] = attempt + 1;
        if (m <= cap) break;
        cap = m;  // the kernels kept counting: second attempt has the exact size
    }
    c.counts["umi_split_items"] = static_cast<double>(split_items);
    *d_edges_out = d_edges;
    *m_out = m;
    return 0;
}

// Directed, sorted adjacency keys (self links included) from undirected rank pairs.
static int keys_from_edges(const std::string& p, const SortedUmis& S, int limit, const uint8_t* d_single,
                           const unsigned long long* d_edges, unsigned long long m, DirectedKeys* out, hipStream_t s) {
    const int n = S.n;
    const int lim2 = 2 * limit;
    int* d_flag; long long* d_pos;
    SL_TRY(scratch((p + ".sflag").c_str(), static_cast<size_t>(n) + 1, &d_flag));
    SL_TRY(scratch((p + ".spos").c_str(), static_cast<size_t>(n) + 1, &d_pos));
    hipLaunchKernelGGL(k_self_flags, dim3(nblk(n, 256)), dim3(256), 0, s, S.U, n, lim2, d_single, S.perm, d_flag);
    SL_HIP(hipMemsetAsync(d_flag + n, 0, sizeof(int), s));
    SL_TRY(exclusive_scan_i32(p.c_str(), d_flag, d_pos, static_cast<size_t>(n) + 1, s));
    long long nself = 0;
    SL_HIP(hipMemcpyAsync(&nself, d_pos + n, sizeof nself, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    const long long nk = 2 * static_cast<long long>(m) + nself;
    // The neighbour lists are explicit (as the reference's are, src/umi_group.cpp:59-103): 20 bytes per link while they are
    // sorted.  Measured up to 2.8e9 links (7e5 12-base UMIs at threshold 4: 4 000 neighbours each); beyond 2^32 the
    // call stops here instead of running out of memory half way.
    if (nk > 0xFFFFFFFFll)
        return fail("sarlacc_amd: %lld neighbour links in one call (at most 4294967295): the threshold joins most of the set -- "
                    "lower it or split the reads into pre-groups", nk);
    unsigned long long *d_k0, *d_k1;
    SL_TRY(scratch((p + ".k0").c_str(), static_cast<size_t>(nk), &d_k0));
    SL_TRY(scratch((p + ".k1").c_str(), static_cast<size_t>(nk), &d_k1));
    if (m) hipLaunchKernelGGL(k_expand_edges, dim3(nblk(static_cast<long long>(m), 256)), dim3(256), 0, s, d_edges, m, S.perm, d_k0 + nself);
    hipLaunchKernelGGL(k_self_keys, dim3(nblk(n, 256)), dim3(256), 0, s, d_flag, d_pos, S.perm, n, d_k0);
    SL_HIP(hipGetLastError());
    if (nk) SL_TRY(sort_keys_u64(p.c_str(), d_k0, d_k1, static_cast<size_t>(nk), 32 + ceil_log2(static_cast<unsigned long long>(n) + 1), s));
    out->keys = d_k1;
    out->nk = nk;
    return 0;
}

// All neighbour pairs within `limit`, as sorted directed keys (self links included).
static int neighbour_keys(const std::string& p, const SortedUmis& S, int limit, const uint8_t* d_single, DirectedKeys* out, hipStream_t s) {
    if (limit < 0) limit = -1;  // nothing can match a negative limit
    unsigned long long* d_edges;
    unsigned long long m;
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    SL_TRY(pair_edges(p, S, limit, 0, -1, &d_edges, &m, s));
    ctx().counts["umi_pair_search_s"] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
    return keys_from_edges(p, S, limit, d_single, d_edges, m, out, s);
}

struct DevAdj {
    long long* off;  // [n+1]
    int* nbr;        // local 0-based ids, trie order
    long long nnz;
};

static int adjacency_from_keys(const std::string& p, const unsigned long long* keys, long long nk, const int* perm, int n,
                               DevAdj* adj, hipStream_t s) {
    SL_TRY(scratch((p + ".off").c_str(), static_cast<size_t>(n) + 1, &adj->off));
    SL_TRY(scratch((p + ".nbr").c_str(), static_cast<size_t>(nk), &adj->nbr));
    hipLaunchKernelGGL(k_row_offsets, dim3(nblk(nk + 1, 256)), dim3(256), 0, s, keys, nk, n, adj->off);
    if (nk) hipLaunchKernelGGL(k_cols_from_keys, dim3(nblk(nk, 256)), dim3(256), 0, s, keys, nk, perm, adj->nbr);
    SL_HIP(hipGetLastError());
    adj->nnz = nk;
    return 0;
}

// Neighbour lists for one group: UMI1 only, or UMI1 n UMI2 listed in UMI2's order
// (src/umi_group.cpp:59-103).
static int group_adjacency(const uint8_t* d_c1, const int64_t* d_o1, const uint8_t* d_c2, const int64_t* d_o2,
                           const int32_t* d_members, const int* d_gid, const uint8_t* d_single, int ngroups, int n,
                           int limit1, int limit2, DevAdj* adj, hipStream_t s, int nsingle = 0, int max_group = 0) {
    SortedUmis S1;
    DirectedKeys K1;
    auto now = [&] { (void)hipStreamSynchronize(s); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    SL_TRY(encode_and_rank("u1", d_c1, d_o1, d_members, d_gid, ngroups, n, &S1, s, d_single, nsingle, max_group));
    const double t1 = now();
    SL_TRY(neighbour_keys("u1", S1, limit1, d_single, &K1, s));
    const double t2 = now();
    ctx().counts["umi_encode_sort_s"] = t1 - t0;
    ctx().counts["umi_search_and_key_sort_s"] = t2 - t1;
    if (!d_c2) return adjacency_from_keys("adj", K1.keys, K1.nk, S1.perm, n, adj, s);

    // membership set of UMI1 links keyed by original column id
    unsigned long long *d_s1a, *d_s1;
    SL_TRY(scratch("u1.set0", static_cast<size_t>(K1.nk), &d_s1a));
    SL_TRY(scratch("u1.set1", static_cast<size_t>(K1.nk), &d_s1));
    if (K1.nk) {
        hipLaunchKernelGGL(k_keys_to_orig, dim3(nblk(K1.nk, 256)), dim3(256), 0, s, K1.keys, K1.nk, S1.perm, d_s1a);
        SL_TRY(sort_keys_u64("u1", d_s1a, d_s1, static_cast<size_t>(K1.nk), 64, s));
    }
    SortedUmis S2;
    DirectedKeys K2;
    SL_TRY(encode_and_rank("u2", d_c2, d_o2, d_members, d_gid, ngroups, n, &S2, s, d_single, nsingle, max_group));
    SL_TRY(neighbour_keys("u2", S2, limit2, d_single, &K2, s));
    int* d_keep; long long* d_pos;
    SL_TRY(scratch("u2.keep", static_cast<size_t>(K2.nk) + 1, &d_keep));
    SL_TRY(scratch("u2.kpos", static_cast<size_t>(K2.nk) + 1, &d_pos));
    if (K2.nk) hipLaunchKernelGGL(k_intersect_flags, dim3(nblk(K2.nk, 256)), dim3(256), 0, s, K2.keys, K2.nk, S2.perm, d_s1, K1.nk, d_keep);
    SL_HIP(hipMemsetAsync(d_keep + K2.nk, 0, sizeof(int), s));
    SL_TRY(exclusive_scan_i32("u2", d_keep, d_pos, static_cast<size_t>(K2.nk) + 1, s));
    long long nkeep = 0;
    SL_HIP(hipMemcpyAsync(&nkeep, d_pos + K2.nk, sizeof nkeep, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    unsigned long long* d_kk;
    SL_TRY(scratch("u2.kept", static_cast<size_t>(nkeep), &d_kk));
    if (K2.nk) hipLaunchKernelGGL(k_compact_keys, dim3(nblk(K2.nk, 256)), dim3(256), 0, s, K2.keys, d_keep, d_pos, K2.nk, d_kk);
    SL_HIP(hipGetLastError());
    return adjacency_from_keys("adj", d_kk, nkeep, S2.perm, n, adj, s);
}

struct ClusterResult {
    long long nclu = 0;
    long long* d_coff = nullptr;  // [nclu+1]
    int32_t* d_out = nullptr;     // member ids (1-based; mapped through members when given)
    long long total = 0;
};

// Greedy clustering of a device CSR graph.  `require_symmetric` is the documented
// precondition of this implementation (the reference's results on asymmetric input
// are an accident of its update order; umi_group always produces symmetric lists).
static int cluster_dev(const DevAdj& adj, int n, const int32_t* d_members, const int* d_gid, int ngroups, bool check_sym,
                       ClusterResult* res, hipStream_t s) {
    ClusterState S{};
    S.off = adj.off; S.nbr = adj.nbr; S.n = n;
    const size_t nn = static_cast<size_t>(n) + 1;
    SL_TRY(scratch("cl.remaining", nn, &S.remaining));
    SL_TRY(scratch("cl.state", nn, &S.state));
    SL_TRY(scratch("cl.mark", nn, &S.mark));
    SL_TRY(scratch("cl.key", nn, &S.key));
    SL_TRY(scratch("cl.m1", nn, &S.m1));
    SL_TRY(scratch("cl.seed", nn, &S.seed));
    SL_TRY(scratch("cl.pickkey", nn, &S.pickkey));
    SL_TRY(scratch("cl.memb", static_cast<size_t>(adj.nnz) + 1, &S.memb));
    SL_TRY(scratch("cl.csize", nn, &S.csize));
    SL_TRY(scratch("cl.err", 3, &S.err));
    SL_TRY(scratch("cl.live", 1, &S.live));
    const int big = std::numeric_limits<int>::max();
    const int init[3] = {big, big, big};
    SL_HIP(hipMemcpyAsync(S.err, init, sizeof init, hipMemcpyHostToDevice, s));
    const dim3 g(nblk(n, 256)), b(256), gw(nblk(n, 4));
    // one wavefront per node from 24 links per node on average (one thread per node below that)
    const bool dense = adj.nnz >= 24LL * n;
    ctx().counts["umi_links"] = static_cast<double>(adj.nnz);
    ctx().counts["umi_cluster_candidate_rounds"] = 0;
    ctx().counts["umi_cluster_full_rounds"] = 0;
    hipLaunchKernelGGL(k_cl_init, g, b, 0, s, S, check_sym ? 1 : 0);
    int herr[3];
    SL_HIP(hipMemcpyAsync(herr, S.err, sizeof herr, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    // first error in index order, as the reference's loop would meet it (src/cluster_umis.cpp:21-40)
    if (herr[0] != big || herr[1] != big) {
        if (herr[0] < herr[1]) return fail("zero length read group");
        return fail("single-read groups should contain only the read itself");
    }
    if (herr[2] != big)
        return fail("sarlacc_amd: neighbour lists must be symmetric and contain the read itself (list %d is not)", herr[2] + 1);

    ClusterTop T{};
    if (dense) {
        SL_TRY(scratch("cl.maxrem", 1, &T.maxrem));
        SL_TRY(scratch("cl.counts", 4, &T.counts));
        SL_TRY(scratch("cl.hop1", nn, &T.hop1));
        T.cap = std::max(1024, n / 8);
        SL_TRY(scratch("cl.cand", static_cast<size_t>(T.cap), &T.cand));
        SL_HIP(hipMemsetAsync(T.counts, 0, 4 * sizeof(int), s));
    }
    const bool top_rounds = dense && !option(OPT_UMI_FULL_ROUNDS);
    long long top_rounds_run = 0, full_rounds_run = 0;
    int rounds_total = 0;
    if (top_rounds) {
        // Candidate-set rounds under the device's control (k_cl_control): CL_GROUP rounds per read-back.  A round whose list
        // was cut off parks the control block (CL_WANTS_FULL; the rounds still queued behind it do nothing) and the host
        // runs that round over every list, as before.
        SL_TRY(scratch("cl.marked", nn, &T.marked));
        SL_TRY(scratch("cl.ctl", CTL_N, &T.ctl));
        int hctl[CTL_N] = {CL_TOP, 0, 0, 0, 0, 0, 0, 0};
        SL_HIP(hipMemcpyAsync(T.ctl, hctl, sizeof hctl, hipMemcpyHostToDevice, s));
        SL_HIP(hipMemsetAsync(S.live, 0, sizeof(int), s));
        SL_HIP(hipMemsetAsync(T.maxrem, 0, sizeof(int), s));
        int round = 0;
        for (;;) {
            for (int q = 0; q < CL_GROUP; ++q, ++round) {
                hipLaunchKernelGGL(k_cl_keys_top, dim3(nblk(n, 1024)), dim3(1024), 0, s, S, T);
                hipLaunchKernelGGL(k_cl_control, dim3(1), dim3(1), 0, s, S, T);
                hipLaunchKernelGGL(k_cl_collect, g, b, 0, s, S, T, 0);
                hipLaunchKernelGGL(k_cl_mark_top, dim3(1024), b, 0, s, S, T);
                hipLaunchKernelGGL(k_cl_pick_top, dim3(1024), b, 0, s, S, T, round);
                hipLaunchKernelGGL(k_cl_commit_list, dim3(64), b, 0, s, S, T);
                hipLaunchKernelGGL(k_cl_decrement_list, dim3(1024), b, 0, s, S, T);
            }
            SL_HIP(hipGetLastError());
            SL_HIP(hipMemcpyAsync(hctl, T.ctl, sizeof hctl, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
            if (hctl[CTL_MODE] == CL_DONE) break;
            if (hctl[CTL_MODE] == CL_WANTS_FULL) {   // (the keys are those of the round that asked: nothing changed since)
                hipLaunchKernelGGL(k_cl_m1_w, gw, b, 0, s, S);
                hipLaunchKernelGGL(k_cl_pick_w, gw, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_commit, g, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_decrement_w, gw, b, 0, s, S, round);
                ++round; ++full_rounds_run;
                const int back[3] = {CL_TOP, hctl[CTL_DELTA], 0};
                SL_HIP(hipMemcpyAsync(T.ctl, back, sizeof back, hipMemcpyHostToDevice, s));
                SL_HIP(hipStreamSynchronize(s));   // (`back` is on this frame)
            }
            if (hctl[CTL_ROUNDS] + full_rounds_run > 4LL * n + 16) return fail("sarlacc_amd: clustering did not converge");
        }
        top_rounds_run = hctl[CTL_ROUNDS];
        rounds_total = static_cast<int>(top_rounds_run + full_rounds_run);
    } else {
        for (int round = 0;; ++round) {
            SL_HIP(hipMemsetAsync(S.live, 0, sizeof(int), s));
            hipLaunchKernelGGL(k_cl_keys, g, b, 0, s, S);
            int live = 0;
            SL_HIP(hipMemcpyAsync(&live, S.live, sizeof live, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
            if (live == 0) break;
            if (dense) {
                hipLaunchKernelGGL(k_cl_m1_w, gw, b, 0, s, S);
                hipLaunchKernelGGL(k_cl_pick_w, gw, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_commit, g, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_decrement_w, gw, b, 0, s, S, round);
            } else {
                hipLaunchKernelGGL(k_cl_m1, g, b, 0, s, S);
                hipLaunchKernelGGL(k_cl_pick, g, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_commit, g, b, 0, s, S, round);
                hipLaunchKernelGGL(k_cl_decrement, g, b, 0, s, S, round);
            }
            SL_HIP(hipGetLastError());
            if (round > 4 * n + 16) return fail("sarlacc_amd: clustering did not converge");
            ++full_rounds_run;
            rounds_total = round + 1;
        }
    }
    ctx().counts["umi_cluster_rounds"] = rounds_total;
    ctx().counts["umi_cluster_candidate_rounds"] = static_cast<double>(top_rounds_run);
    ctx().counts["umi_cluster_full_rounds"] = static_cast<double>(full_rounds_run);

    // ---- output order: solos by index, then picks by key descending ----
    int *d_issolo, *d_isseed, *d_order, *d_val, *d_val2, *d_sizes;
    long long *d_spos, *d_kpos;
    unsigned long long *d_sk, *d_sk2;
    SL_TRY(scratch("cl.issolo", nn, &d_issolo));
    SL_TRY(scratch("cl.isseed", nn, &d_isseed));
    SL_TRY(scratch("cl.spos", nn, &d_spos));
    SL_TRY(scratch("cl.kpos", nn, &d_kpos));
    SL_TRY(scratch("cl.order", nn, &d_order));
    SL_TRY(scratch("cl.val", nn, &d_val));
    SL_TRY(scratch("cl.val2", nn, &d_val2));
    SL_TRY(scratch("cl.sk", nn, &d_sk));
    SL_TRY(scratch("cl.sk2", nn, &d_sk2));
    SL_TRY(scratch("cl.sizes", nn, &d_sizes));
    hipLaunchKernelGGL(k_cl_flags, g, b, 0, s, S, d_issolo, d_isseed);
    SL_HIP(hipMemsetAsync(d_issolo + n, 0, sizeof(int), s));
    SL_HIP(hipMemsetAsync(d_isseed + n, 0, sizeof(int), s));
    SL_TRY(exclusive_scan_i32("cl", d_issolo, d_spos, nn, s));
    SL_TRY(exclusive_scan_i32("cl", d_isseed, d_kpos, nn, s));
    long long nsolo = 0, nseed = 0;
    SL_HIP(hipMemcpyAsync(&nsolo, d_spos + n, sizeof nsolo, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(&nseed, d_kpos + n, sizeof nseed, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    const long long nclu = nsolo + nseed;
    if (nclu) {
        hipLaunchKernelGGL(k_cl_list, g, b, 0, s, d_issolo, d_spos, d_isseed, d_kpos, nsolo, S.pickkey, n, d_sk, d_val);
        SL_TRY(sort_pairs_u64_i32("cl", d_sk, d_sk2, d_val, d_val2, static_cast<size_t>(nclu), 64, s));
        if (d_gid && ngroups > 1) {  // stable: keeps the in-group order, groups in input order
            hipLaunchKernelGGL(k_cl_gidkey, dim3(nblk(nclu, 256)), b, 0, s, d_gid, d_val2, nclu, d_sk);
            SL_TRY(sort_pairs_u64_i32("cl", d_sk, d_sk2, d_val2, d_val, static_cast<size_t>(nclu),
                                      ceil_log2(static_cast<unsigned long long>(ngroups) + 1), s));
            SL_HIP(hipMemcpyAsync(d_order, d_val, sizeof(int) * static_cast<size_t>(nclu), hipMemcpyDeviceToDevice, s));
        } else {
            SL_HIP(hipMemcpyAsync(d_order, d_val2, sizeof(int) * static_cast<size_t>(nclu), hipMemcpyDeviceToDevice, s));
        }
    }
    long long* d_coff;
    int32_t* d_out;
    SL_TRY(scratch("cl.coff", static_cast<size_t>(nclu) + 2, &d_coff));
    SL_TRY(scratch("cl.out", nn, &d_out));
    if (nclu) {
        hipLaunchKernelGGL(k_cl_sizes, dim3(nblk(nclu, 256)), b, 0, s, S, d_order, nclu, d_sizes);
        SL_HIP(hipMemsetAsync(d_sizes + nclu, 0, sizeof(int), s));
        SL_TRY(exclusive_scan_i32("cl", d_sizes, d_coff, static_cast<size_t>(nclu) + 1, s));
        hipLaunchKernelGGL(k_cl_write, dim3(nblk(nclu, 256)), b, 0, s, S, d_order, nclu, d_coff, d_members, d_out);
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpyAsync(&res->total, d_coff + nclu, sizeof(long long), hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
    } else {
        const long long zero = 0;
        SL_HIP(hipMemcpyAsync(d_coff, &zero, sizeof zero, hipMemcpyHostToDevice, s));
        res->total = 0;
    }
    res->nclu = nclu;
    res->d_coff = d_coff;
    res->d_out = d_out;
    return 0;
}

// pairs handed in from outside (the exchange of a tile-sharded search): i < j < n, rank_i << 32 | rank_j
__global__ void k_check_pairs(const unsigned long long* __restrict__ pairs, long long m, unsigned long long n, int* __restrict__ bad) {
    const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const unsigned long long a = pairs[e] >> 32, b = pairs[e] & 0xffffffffull;
    if (a >= n || b >= n || a >= b) *bad = 1;
}

static int upload_strings(const char* tag, const char* chars, const int64_t* off, int64_t n, uint8_t** d_chars,
                          int64_t** d_off, hipStream_t s) {
    const int64_t base = n ? off[0] : 0;
    const int64_t total = n ? off[n] - base : 0;
    std::vector<int64_t> rel(static_cast<size_t>(n) + 1);
    for (int64_t i = 0; i <= n; ++i) rel[i] = (n ? off[i] : 0) - base;
    SL_TRY(upload((std::string(tag) + ".chars").c_str(), reinterpret_cast<const uint8_t*>(chars) + base, static_cast<size_t>(total), d_chars, s));
    SL_TRY(upload((std::string(tag) + ".off").c_str(), rel.data(), rel.size(), d_off, s));
    return 0;
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" {

int sarlacc_compute_lev_masked(const char* seq, const int64_t* off, int64_t n, double* out) {
    if (n < 0) return fail("sarlacc_amd: negative number of sequences");
    if (n < 2) return 0;
    if (n > 60000) return fail("sarlacc_amd: compute_lev_masked is dense (n^2/2 doubles); n = %lld is too large", static_cast<long long>(n));
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t* d_c; int64_t* d_o;
    SL_TRY(upload_strings("lev", seq, off, n, &d_c, &d_o, s));
    // no ordering needed: encode in input order
    UmiArrays U;
    SL_TRY(alloc_umi("lev.raw", n, &U));
    unsigned long long *khi, *klo; int *idx, *bad;
    SL_TRY(scratch("lev.khi", n, &khi));
    SL_TRY(scratch("lev.klo", n, &klo));
    SL_TRY(scratch("lev.idx", n, &idx));
    SL_TRY(scratch("lev.bad", 3, &bad));
    const int init[3] = {std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), 0};
    SL_HIP(hipMemcpyAsync(bad, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_umi_encode, dim3(nblk(n, 256)), dim3(256), 0, s, d_c, d_o, static_cast<const int32_t*>(nullptr), static_cast<int>(n), U, khi, klo, idx, static_cast<const uint8_t*>(nullptr), bad);
    int hbad[3];
    SL_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    const bool is_long = hbad[1] != init[1];
    bool xl = false;
    if (is_long) {
        const int maxlen = -hbad[2];
        if (maxlen > UMI_XL_MAX) return fail("sarlacc_amd: sequence longer than %d bases is not supported", UMI_XL_MAX);
        xl = maxlen > UMI_LONG_MAX;
        const int words = xl ? (maxlen + 31) / 32 : UMI_LONG_WORDS, nkeys = (maxlen + UMI_KEY_BASES - 1) / UMI_KEY_BASES;
        SL_TRY(alloc_umi("lev.rawL", n, &U, words));
        unsigned long long* keys;
        SL_TRY(scratch("lev.keysL", static_cast<size_t>(n) * nkeys, &keys));
        SL_HIP(hipMemcpyAsync(bad, init, sizeof init, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_umi_encode_long, dim3(nblk(n, 256)), dim3(256), 0, s, d_c, d_o, static_cast<const int32_t*>(nullptr), static_cast<int>(n), U, keys, idx, static_cast<const uint8_t*>(nullptr), bad, words, nkeys);
        SL_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
    }
    // characters outside ACGTN behave as ordinary distinct letters in the reference
    // (src/compute_lev_masked.cpp:51); only ACGTN is supported here
    if (hbad[0] != init[0]) return fail("sarlacc_amd: sequence contains a character outside ACGTN");
    const long long npairs = n * (n - 1) / 2;
    double* d_out;
    SL_TRY(scratch("lev.out", static_cast<size_t>(npairs), &d_out));
    if (is_long && xl) hipLaunchKernelGGL(k_lev_dense_long<UMI_XL_MAX>, dim3(nblk(npairs, 128)), dim3(128), 0, s, U, static_cast<int>(n), d_out);
    else if (is_long) hipLaunchKernelGGL(k_lev_dense_long<UMI_LONG_MAX>, dim3(nblk(npairs, 128)), dim3(128), 0, s, U, static_cast<int>(n), d_out);
    else hipLaunchKernelGGL(k_lev_dense, dim3(nblk(npairs, 128)), dim3(128), 0, s, U, static_cast<int>(n), d_out);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpy(out, d_out, sizeof(double) * static_cast<size_t>(npairs), hipMemcpyDeviceToHost));
    return 0;
}

int sarlacc_fast_levdist_test(const char* seq, const int64_t* off, int64_t n, int limit, int64_t* nbr_off,
                              int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need) {
    if (n < 0) return fail("sarlacc_amd: negative number of sequences");
    *nbr_need = 0;
    nbr_off[0] = 0;
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t* d_c; int64_t* d_o;
    SL_TRY(upload_strings("lv", seq, off, n, &d_c, &d_o, s));
    DevAdj adj;
    SL_TRY(group_adjacency(d_c, d_o, nullptr, nullptr, nullptr, nullptr, nullptr, 1, static_cast<int>(n), limit, limit, &adj, s));
    std::vector<long long> hoff(static_cast<size_t>(n) + 1);
    SL_HIP(hipMemcpy(hoff.data(), adj.off, sizeof(long long) * hoff.size(), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i <= n; ++i) nbr_off[i] = hoff[i];
    *nbr_need = adj.nnz;
    if (!nbr || nbr_cap < adj.nnz) return 0;  // sizing call
    std::vector<int> h(static_cast<size_t>(adj.nnz));
    if (adj.nnz) SL_HIP(hipMemcpy(h.data(), adj.nbr, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    for (long long i = 0; i < adj.nnz; ++i) nbr[i] = h[i] + 1;
    return 0;
}

int sarlacc_cluster_umis_test(const int64_t* link_off, const int32_t* links, int64_t n, int64_t* nclusters,
                              int64_t* clu_off, int32_t* clu) {
    if (n < 0) return fail("sarlacc_amd: negative number of lists");
    *nclusters = 0;
    clu_off[0] = 0;
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    const int64_t nnz = link_off[n] - link_off[0];
    std::vector<long long> hoff(static_cast<size_t>(n) + 1);
    std::vector<int> hn(static_cast<size_t>(nnz) + 1);
    for (int64_t i = 0; i <= n; ++i) hoff[i] = link_off[i] - link_off[0];
    for (int64_t i = 0; i < nnz; ++i) {
        const int32_t v = links[link_off[0] + i];
        if (v < 1 || v > n) return fail("sarlacc_amd: link %d outside 1..%lld", v, static_cast<long long>(n));
        hn[i] = v - 1;
    }
    DevAdj adj;
    SL_TRY(upload("adj.off", hoff.data(), hoff.size(), &adj.off, s));
    SL_TRY(upload("adj.nbr", hn.data(), hn.size(), &adj.nbr, s));
    adj.nnz = nnz;
    ClusterResult res;
    SL_TRY(cluster_dev(adj, static_cast<int>(n), nullptr, nullptr, 1, true, &res, s));
    std::vector<long long> co(static_cast<size_t>(res.nclu) + 1);
    SL_HIP(hipMemcpy(co.data(), res.d_coff, sizeof(long long) * co.size(), hipMemcpyDeviceToHost));
    if (res.total) SL_HIP(hipMemcpy(clu, res.d_out, sizeof(int32_t) * static_cast<size_t>(res.total), hipMemcpyDeviceToHost));
    for (long long c = 0; c <= res.nclu; ++c) clu_off[c] = co[c];
    *nclusters = res.nclu;
    return 0;
}

int sarlacc_umi_group(const char* umi1, const int64_t* off1, const char* umi2, const int64_t* off2, int64_t n,
                      int thresh1, int thresh2, const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                      int64_t* nclusters, int64_t* clu_off, int32_t* clu) {
    if (n < 0 || ngroups < 0) return fail("sarlacc_amd: negative sizes");
    *nclusters = 0;
    clu_off[0] = 0;
    if (ngroups == 0) return 0;
    const int64_t total = grp_off[ngroups] - grp_off[0];
    for (int64_t i = 0; i < total; ++i) {
        const int32_t v = grp[grp_off[0] + i];
        if (v < 1 || v > n) return fail("sarlacc_amd: pre-group index %d outside 1..%lld", v, static_cast<long long>(n));
    }
    SL_TRY(ensure_device());
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now();
    hipStream_t s = nullptr;
    uint8_t *d_c1, *d_c2 = nullptr;
    int64_t *d_o1, *d_o2 = nullptr;
    SL_TRY(upload_strings("g1", umi1, off1, n, &d_c1, &d_o1, s));
    if (umi2) SL_TRY(upload_strings("g2", umi2, off2, n, &d_c2, &d_o2, s));
    // all pre-groups go through the kernels together: elements = flattened member list,
    // pairs are only formed inside a pre-group, clusters come back group by group
    if (total > std::numeric_limits<int>::max() - 1024) return fail("sarlacc_amd: more than 2^31 pre-group members");
    const int N = static_cast<int>(total);
    std::vector<int> gid(static_cast<size_t>(N) + 1);
    std::vector<uint8_t> single(static_cast<size_t>(N) + 1, 0);
    int64_t nsingle = 0, max_group = 0;
    for (int64_t g = 0; g < ngroups; ++g) {
        const int64_t a = grp_off[g] - grp_off[0], b = grp_off[g + 1] - grp_off[0];
        for (int64_t i = a; i < b; ++i) { gid[i] = static_cast<int>(g); single[i] = (b - a == 1); }
        nsingle += (b - a == 1) ? 1 : 0;
        max_group = std::max<int64_t>(max_group, b - a);
    }
    int64_t nc = 0;
    if (N > 0) {
        int32_t* d_grp; int* d_gid; uint8_t* d_single;
        SL_TRY(upload("g.members", grp + grp_off[0], static_cast<size_t>(N), &d_grp, s));
        SL_TRY(upload("g.gid", gid.data(), static_cast<size_t>(N), &d_gid, s));
        SL_TRY(upload("g.single", single.data(), static_cast<size_t>(N), &d_single, s));
        DevAdj adj;
        const double t0 = now();
        SL_TRY(group_adjacency(d_c1, d_o1, d_c2, d_o2, d_grp, d_gid, d_single, static_cast<int>(ngroups), N, thresh1, thresh2, &adj, s,
                               static_cast<int>(nsingle), static_cast<int>(max_group)));
        SL_HIP(hipStreamSynchronize(s));
        const double t1 = now();
        ClusterResult res;
        SL_TRY(cluster_dev(adj, N, d_grp, d_gid, static_cast<int>(ngroups), false, &res, s));
        const double t2 = now();
        // where a call's time goes (sarlacc_stage_count): seconds of the neighbour search incl. sorts / of the clustering
        ctx().counts["umi_adjacency_s"] = t1 - t0;
        ctx().counts["umi_cluster_s"] = t2 - t1;
        ctx().counts["umi_links"] = static_cast<double>(adj.nnz);
        std::vector<long long> co(static_cast<size_t>(res.nclu) + 1);
        SL_HIP(hipMemcpy(co.data(), res.d_coff, sizeof(long long) * co.size(), hipMemcpyDeviceToHost));
        if (res.total) SL_HIP(hipMemcpy(clu, res.d_out, sizeof(int32_t) * static_cast<size_t>(res.total), hipMemcpyDeviceToHost));
        for (long long c = 0; c <= res.nclu; ++c) clu_off[c] = co[c];
        nc = res.nclu;
        ctx().counts["umi_tables_in_s"] = t0 - t_in;    // strings and pre-group tables to the device
        ctx().counts["umi_clusters_out_s"] = now() - t2;
    }
    *nclusters = nc;
    return 0;
}

// Row tiles of the all-pairs matrix owned by shard `index` of `count`: boundaries balance the
// triangular work (row tile b meets nt - b column tiles).
static void shard_tiles(int nt, int index, int count, int* lo, int* hi) {
    const double total = 0.5 * nt * (nt + 1.0);
    auto bound = [&](int k) -> int {  // smallest b with work(0..b) >= k/count of the total
        if (k <= 0) return 0;
        if (k >= count) return nt;
        const double want = total * k / count;
        int b = 0;
        double acc = 0;
        while (b < nt && acc < want) { acc += nt - b; ++b; }
        return b;
    };
    *lo = bound(index);
    *hi = bound(index + 1);
}

int sarlacc_umi_pairs_shard(const char* umi, const int64_t* off, int64_t n, int limit, int shard_index,
                            int shard_count, uint64_t* pairs, int64_t cap, int64_t* npairs) {
    if (n < 0 || shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail("sarlacc_amd: bad shard request");
    *npairs = 0;
    if (n == 0 || limit < 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t* d_c; int64_t* d_o;
    SL_TRY(upload_strings("ps", umi, off, n, &d_c, &d_o, s));
    SortedUmis S;
    SL_TRY(encode_and_rank("u1", d_c, d_o, nullptr, nullptr, 1, static_cast<int>(n), &S, s));
    int lo, hi;
    shard_tiles(static_cast<int>(nblk(n, TILE)), shard_index, shard_count, &lo, &hi);
    unsigned long long* d_edges;
    unsigned long long m;
    SL_TRY(pair_edges("u1", S, limit, lo, hi, &d_edges, &m, s));
    *npairs = static_cast<int64_t>(m);
    if (!pairs || cap < static_cast<int64_t>(m)) return 0;  // sizing call
    if (m) SL_HIP(hipMemcpy(pairs, d_edges, sizeof(uint64_t) * m, hipMemcpyDeviceToHost));
    return 0;
}

// clustering of ONE pre-group from its neighbour pairs, which are already on the device (validated there)
static int group_from_device_pairs(const char* umi, const int64_t* off, int64_t n, int limit, const unsigned long long* d_edges,
                                   int64_t npairs, int64_t* nclusters, int64_t* clu_off, int32_t* clu, hipStream_t s) {
    uint8_t* d_c; int64_t* d_o;
    SL_TRY(upload_strings("ps", umi, off, n, &d_c, &d_o, s));
    SortedUmis S;
    SL_TRY(encode_and_rank("u1", d_c, d_o, nullptr, nullptr, 1, static_cast<int>(n), &S, s));
    int* d_bad;
    SL_TRY(scratch("ps.badpair", 1, &d_bad));
    SL_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    if (npairs) hipLaunchKernelGGL(k_check_pairs, dim3(nblk(npairs, 256)), dim3(256), 0, s, d_edges, static_cast<long long>(npairs), static_cast<unsigned long long>(n), d_bad);
    int bad = 0;
    SL_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (bad) return fail("sarlacc_amd: malformed neighbour pair");
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    DirectedKeys K;
    SL_TRY(keys_from_edges("u1", S, limit < 0 ? -1 : limit, nullptr, d_edges, static_cast<unsigned long long>(npairs), &K, s));
    DevAdj adj;
    SL_TRY(adjacency_from_keys("adj", K.keys, K.nk, S.perm, static_cast<int>(n), &adj, s));
    SL_HIP(hipStreamSynchronize(s));
    const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    ClusterResult res;
    SL_TRY(cluster_dev(adj, static_cast<int>(n), nullptr, nullptr, 1, false, &res, s));
    ctx().counts["umi_adjacency_s"] = t1 - t0;
    ctx().counts["umi_cluster_s"] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t1;
    std::vector<long long> co(static_cast<size_t>(res.nclu) + 1);
    SL_HIP(hipMemcpy(co.data(), res.d_coff, sizeof(long long) * co.size(), hipMemcpyDeviceToHost));
    if (res.total) SL_HIP(hipMemcpy(clu, res.d_out, sizeof(int32_t) * static_cast<size_t>(res.total), hipMemcpyDeviceToHost));
    for (long long c = 0; c <= res.nclu; ++c) clu_off[c] = co[c];
    *nclusters = res.nclu;
    return 0;
}

int sarlacc_umi_group_from_pairs(const char* umi, const int64_t* off, int64_t n, int limit, const uint64_t* pairs,
                                 int64_t npairs, int64_t* nclusters, int64_t* clu_off, int32_t* clu) {
    if (n < 0 || npairs < 0) return fail("sarlacc_amd: negative sizes");
    *nclusters = 0;
    clu_off[0] = 0;
    if (n == 0) return 0;
    if (n == 1) {  // a pre-group of one read passes through (src/umi_group.cpp:39-42)
        clu[0] = 1; clu_off[1] = 1; *nclusters = 1;
        return 0;
    }
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    unsigned long long* d_edges;
    SL_TRY(upload("u1.edges_in", reinterpret_cast<const unsigned long long*>(pairs), static_cast<size_t>(npairs), &d_edges, s));
    return group_from_device_pairs(umi, off, n, limit, d_edges, npairs, nclusters, clu_off, clu, s);
}

// ---- the pair exchange with the pairs kept in HBM (one giant pre-group over several GPUs: 10^8 pairs at 8 x 10^6 reads) ----
int sarlacc_dev_umi_pairs_shard(const char* umi, const int64_t* off, int64_t n, int limit, int shard_index,
                                int shard_count, int64_t* npairs) {
    if (n < 0 || shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail("sarlacc_amd: bad shard request");
    *npairs = 0;
    g_shard_pairs = -1;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    if (n == 0 || limit < 0) { g_shard_pairs = 0; return 0; }
    uint8_t* d_c; int64_t* d_o;
    SL_TRY(upload_strings("ps", umi, off, n, &d_c, &d_o, s));
    SortedUmis S;
    SL_TRY(encode_and_rank("u1", d_c, d_o, nullptr, nullptr, 1, static_cast<int>(n), &S, s));
    int lo, hi;
    shard_tiles(static_cast<int>(nblk(n, TILE)), shard_index, shard_count, &lo, &hi);
    unsigned long long* d_edges;
    unsigned long long m;
    SL_TRY(pair_edges("u1", S, limit, lo, hi, &d_edges, &m, s));
    g_shard_pairs = static_cast<long long>(m);   // (they stay in the workspace buffer "u1.edges" until the next search or release)
    *npairs = static_cast<int64_t>(m);
    return 0;
}

int sarlacc_dev_umi_pairs_fetch(uint64_t* d_pairs, int64_t cap) {
    if (g_shard_pairs < 0) return fail("sarlacc_amd: no neighbour pairs to fetch (sarlacc_dev_umi_pairs_shard must be the call before)");
    if (cap < g_shard_pairs) return fail("sarlacc_amd: pair buffer too small (%lld needed)", g_shard_pairs);
    if (g_shard_pairs == 0) return 0;
    auto it = ctx().ws.find("u1.edges");
    if (it == ctx().ws.end() || !it->second.ptr || it->second.cap < sizeof(uint64_t) * static_cast<size_t>(g_shard_pairs)) {
        g_shard_pairs = -1;
        return fail("sarlacc_amd: the neighbour pairs of the last shard search are gone (workspace released)");
    }
    if (!d_pairs) return fail("sarlacc_amd: null pair buffer");
    SL_HIP(hipMemcpy(d_pairs, it->second.ptr, sizeof(uint64_t) * static_cast<size_t>(g_shard_pairs), hipMemcpyDeviceToDevice));
    return 0;
}

int sarlacc_dev_umi_group_from_pairs(const char* umi, const int64_t* off, int64_t n, int limit, const uint64_t* d_pairs,
                                     int64_t npairs, int64_t* nclusters, int64_t* clu_off, int32_t* clu) {
    if (n < 0 || npairs < 0) return fail("sarlacc_amd: negative sizes");
    if (npairs && !d_pairs) return fail("sarlacc_amd: null pair buffer");
    *nclusters = 0;
    clu_off[0] = 0;
    if (n == 0) return 0;
    if (n == 1) {  // a pre-group of one read passes through (src/umi_group.cpp:39-42)
        clu[0] = 1; clu_off[1] = 1; *nclusters = 1;
        return 0;
    }
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    SL_HIP(hipDeviceSynchronize());   // the caller's collective wrote d_pairs on a stream of its own
    return group_from_device_pairs(umi, off, n, limit, reinterpret_cast<const unsigned long long*>(d_pairs), npairs, nclusters, clu_off, clu, s);
}
}
