// align.hip -- quality-weighted read-vs-reference affine-gap DP on gfx950.
//
// Replaces the CPU loop of the reference's reference_align class
// (/root/reference/src/reference_align.cpp:54-181 fill, :231-351 backtrack +
// interval lookup, :353-389 strings) and its three .Call wrappers
// (src/adaptor_align.cpp, src/barcode_align.cpp, src/general_align.cpp).
//
// Mapping (see DESIGN.md "DP kernel"):
//   * one wavefront = up to NGMAX alignments side by side; inside one alignment a
//     lane owns K consecutive reference (adaptor) columns, lanes are skewed by one
//     read row (anti-diagonal wavefront);
//   * per-column state (vertical jump score, previous score of the column) lives in
//     VGPRs; per-row state (score of the column to the left, horizontal jump score,
//     "left cell was a horizontal gap" flag) travels to the next lane with DPP moves
//     (row_shr:1 for 16-lane groups, whose leaders get their column-0 inputs from the
//     DPP fill operand; wave_shr:1 otherwise) -- no LDS on the recurrence path;
//   * read bases + qualities are staged 64 rows at a time into an LDS ring, the
//     5 x navail fp64 cost tables sit in LDS;
//   * traceback needs 4 bits per cell (move + "jump continued" flags; the jump lengths
//     the reference stores are rebuilt during the walk).  adaptor_align (MODE 3) does not
//     produce them in the fill at all: it snapshots the lane state every SNAP_P steps and
//     afterwards recomputes, with codes, only the window above each alignment's landing
//     row (tracked online; it replaces the walk up the last column), which the group's
//     leader lane then walks.  MODE 1/2 stream the codes of every cell with nontemporal
//     stores to a per-wave tile in HBM (general_align's strings; gapopen < 0).
//
// All arithmetic is fp64 add/sub/compare in the reference's order, compiled with
// -ffp-contract=off, so scores are bit-identical to the CPU path.
#include "common.hpp"

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace sarlacc {

constexpr int NGMAX = 4;   // alignments processed side by side in one wavefront (groups of 16 lanes or more)
constexpr int NGMAX2 = 8;  // ... with two 8-lane alignments interleaved in every 16-lane DPP row (ROWF 2)
constexpr int NWAVES = 4;  // wavefronts per workgroup: they share one copy of the cost table in LDS
constexpr int RING = 128;        // staged read positions per alignment (2 x 64)
constexpr int RING_MIRROR = 8;   // the first entries again behind the ring: a block of up to 8 steps reads base + 2u without wrapping
constexpr int RING_SLOT = 256;   // uint16 entries reserved per alignment (512 B: a ring address is base | offset)
constexpr int MAX_REF = 1024;

struct AlignArgs {
    const uint8_t* seq;       // ASCII bases, or 2-bit packed bases when nmask != nullptr
    const uint8_t* nmask;     // packed input only: 1 bit per base, set where the base is not A/C/G/T
    const uint8_t* qual;
    const int64_t* off;
    long long n;
    int R, W, ngroups, local;
    int qoffset, navail;
    double GO, GE;
    const double* tables;     // cost table rows of navail doubles, laid out by build_cost_rows
    int tab_doubles;          // size of `tables`
    int row_bytes;            // navail * sizeof(double)
    const double* rowzero;    // [R+1] scores of DP row 0
    const uint32_t* colbase;  // [R+1] byte offset (inside `tables`) of the rows a column reads
    const uint8_t* refchars;  // [R]
    double* scores;
    int32_t* starts;
    int32_t* ends;
    const int32_t* sec_s;
    const int32_t* sec_e;
    int nsec;
    int32_t* sec_so;
    int32_t* sec_wo;
    void* dirs;
    unsigned long long dirs_per_wave;  // elements per wavefront
    int snap_head, snap_win;           // MODE 3: see SNAP_P
    int* badqual;                      // min index of a read holding a quality below the offset
    int read_base;                     // index of this launch's first read in the caller's batch (chunked host calls)
    long long sec_stride;              // reads per section row of sec_so / sec_wo (the caller's whole batch)
    uint8_t* aln_ref;                  // MODE 2: reversed gapped strings, stride L+R per read
    uint8_t* aln_qry;
    int32_t* aln_len;
    int32_t* edits;
    int wide_maxlen;                   // k_align_wide: the longest read of the launch (sizes of its code tiles and boundary arrays)
    double* wide_bnd;                  // k_align_wide with several strips: per workgroup [2][3][wide_maxlen + 2] row states at a strip's last column
    int wide_band;                     // k_align_wide_q with traceback: codes only for the steps that can hold cells within this many rows of the
                                       // main diagonal (0: codes of every cell); a walk that leaves them puts its read on the redo list
    int* wide_redo;                    // [0] number of reads on the list, [1 ...] their indices (written by the banded launch)
    const int* wide_list;              // the redo launch: the reads to align (count read from the device: wide_list[-1]); null: reads 0 .. n - 1
};

// Traceback code of one cell, 4 bits -- the raw outcomes of the cell's four comparisons:
//   bit 0  H > V            (horizontal beats vertical)
//   bit 1  M > max(H, V)    (diagonal move; otherwise bit 0 picks horizontal / vertical)
//   bit 2  the horizontal jump was continued here (left_jump_point kept)
//   bit 3  the vertical jump was continued here (up_jump_point kept)
// The reference stores jump LENGTHS (src/reference_align.cpp:139,154,170-177); they are
// recovered during the walk: a horizontal step at column c is 1 + (number of consecutive
// columns c, c-1, ... whose bit 2 is set), a vertical step at row i is 1 + (number of
// consecutive rows i, i-1, ... whose bit 3 is set) -- exactly 1 + pos - left_jump_point and
// 1 + i - up_jump_point.  A lane packs the codes of STEPS consecutive steps x K columns
// into one word, first cell in the top nibble: 0.5 B per cell instead of the 2-4 B of a
// stored length.
// Lane masks (one bit per lane) live in SGPR pairs.  The selects and the traceback-bit packing
// below take them as scalar operands, so all flag logic (and / andn2 / nor) runs on the scalar
// unit and the vector unit sees exactly one instruction per select and one per packed bit.
using mask_t = unsigned long long;
typedef const double __attribute__((address_space(3))) lds_cdouble;
typedef const uint16_t __attribute__((address_space(3))) lds_cu16;

__device__ __forceinline__ int sel32(int if0, int if1, mask_t m) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}
// pk = 2 * pk + bit: shifts the word left and drops the lane's mask bit in at the bottom
__device__ __forceinline__ uint32_t push_bit(uint32_t pk, mask_t m) {
    uint32_t r;
    mask_t carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(pk), "s"(m));
    return r;
}
__device__ __forceinline__ int hi32(double v) { return static_cast<int>(__double_as_longlong(v) >> 32); }
__device__ __forceinline__ int lo32(double v) { return static_cast<int>(__double_as_longlong(v)); }
__device__ __forceinline__ double mk64(int hi, int lo) {
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// Hand a value to the lane that owns the next columns of the same alignment.  ROWF 1: alignments are 16 lanes wide
// and start on DPP row boundaries (row_shr:1); the first lane of every row -- the leader of an alignment -- has no
// source lane and keeps `keep` (its DP column 0 value: pass the destination itself and the constant set before the
// loop stays there for free).  ROWF 2: two 8-lane alignments share a DPP row, one in the even and one in the odd
// lanes, and the hand-over is row_shr:2 -- lanes 0 and 1 of the row are the two leaders and again have no source.
// Twice the columns per lane at the same width in columns: half the cross-lane moves per cell.  ROWF 0: across
// the whole wave (wave_shr:1), leaders overwrite what arrives.
template <int ROWF>
__device__ __forceinline__ int lane_shr1(int v, int keep) {
    if (ROWF == 1) return __builtin_amdgcn_update_dpp(keep, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    if (ROWF == 2) return __builtin_amdgcn_update_dpp(keep, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
template <int ROWF>
__device__ __forceinline__ double lane_shr1(double v, double keep) {
    return mk64(lane_shr1<ROWF>(hi32(v), hi32(keep)), lane_shr1<ROWF>(lo32(v), lo32(keep)));
}

template <int K>
struct TbSteps { static constexpr int value = (8 / K) > 0 ? 8 / K : 1; };
template <int K>
struct TbStore { using type = uint32_t; };
template <>
struct TbStore<16> { using type = unsigned long long; };

template <bool B>
struct Flag { static constexpr bool value = B; };
template <int V>
struct Int { static constexpr int value = V; };

// MODE 3 (adaptor_align, local, gapopen >= 0): no traceback stream at all.  The fill keeps only
// scores, the landing row of column R and, every SNAP_P steps, a snapshot of the complete lane
// state (column scores, jump scores, the row state in flight between lanes).  Afterwards each
// alignment restores the snapshot just above its landing row and recomputes a window of at
// most SNAP_WIN steps with traceback codes; the leader walks the window and, should the path
// leave it through the top, the next window up is recomputed.  A snapshot is the exact state
// of the recurrence, so the recomputed cells are the cells of the first pass, bit for bit.
constexpr int SNAP_P = 64;      // steps between snapshots (multiple of the 64-row staging block)
// rows above the row a walk asks for that its window must cover: an alignment of R columns
// rarely spans more than R + R/4 rows, and a longer one only costs another window
static inline int snap_head(int R) { return R + R / 4 + 4; }
// steps per window: SNAP_P - 1 + head + W lanes + 1, rounded up to the 8-step store granule
static inline int snap_win(int R, int W) { return (SNAP_P + snap_head(R) + W + 7) / 8 * 8; }

// MODE 0: scores only.  MODE 1: scores + reference->read map (adaptor_align), codes streamed.
// MODE 2: scores + gapped strings + edit distance (general_align).
// MODE 3: as MODE 1 by snapshots + windowed recompute (LOCAL, !PENSEL only; see SNAP_P).
// LOCAL: free leading read bases + free vertical gaps in the last column (adaptor mode).
// ROWF: 0 alignments of any width, wave-wide shifts; 1 alignments are 16 lanes wide and start on DPP row
// boundaries; 2 alignments are 8 lanes wide, two interleaved per DPP row (see lane_shr1).
// KLAST: index (inside its lane) of reference column R when known at compile time, else -1.
// PENSEL: select the gap penalty of every step explicitly (only needed when gapopen < 0).
//
// Penalty selection.  The reference charges a gap step "extension" instead of "open +
// extension" when the cell it leaves was itself reached by a gap of the same kind
// (src/reference_align.cpp:125-158).  In that case the cell's score IS the running jump
// score (best == H == lj, bit for bit), so the reference's candidate `left - GE` equals the
// jump continuation `lj - GE` exactly, the comparison `continuation > candidate` is false and
// the maximum is the continuation.  With gapopen >= 0 the candidate `left - GO` is <= that
// continuation, so max(continuation, left - GO) is the same double: the kernel always
// subtracts the opening penalty and selects nothing.  Only the "jump continued" flag differs
// (true instead of false), and the true flag is raw && !(previous cell's move is the same gap
// kind) -- the traceback applies that from the neighbour's code, which it reads anyway.
template <int K, int MODE, bool LOCAL, int ROWF, int KLAST, bool PENSEL>
// (six wavefronts per SIMD for the snapshot mode; five with eight alignments per wavefront: four columns per lane need the registers --
// at six the recompute code spilled -- 2 182 -> 2 213 GCUPS on the same box)
__global__ void __launch_bounds__(64 * NWAVES, (MODE == 3 && ROWF != 2) ? 6 : 5) k_align(const AlignArgs A) {
    constexpr bool ROW16 = ROWF != 0;                      // leaders keep their column-0 inputs through the DPP fill operand
    constexpr int NG = ROWF == 2 ? NGMAX2 : NGMAX;         // alignments per wavefront at most
    // uint16 entries per alignment's ring slot: 512 B slots let a ring address be base | offset; with eight
    // alignments per wave the slots are packed (ring + mirror) and the address is an add
    constexpr int SLOT = ROWF == 2 ? RING + RING_MIRROR : RING_SLOT;
    constexpr int UNR = TbSteps<K>::value;
    constexpr int CELLS = UNR * K;
    using Word = typename TbStore<K>::type;
    constexpr bool ADDC = sizeof(Word) == 4;
    // LDS: read rings first (256 B per alignment, so a ring address is base | offset), then the
    // cost table shared by the waves of the workgroup, then the per-wave reference->read maps
    extern __shared__ __align__(256) unsigned char smem[];
    static_assert(RING * sizeof(uint16_t) == 256 && RING + RING_MIRROR <= RING_SLOT, "ring addressing assumes 256 B of ring inside a 512 B slot");
    constexpr int RING_BYTES = NWAVES * NG * SLOT * static_cast<int>(sizeof(uint16_t));
    static_assert(RING_BYTES % 8 == 0 && (SLOT * sizeof(uint16_t)) % 8 == 0, "table and 8-byte ring stores stay aligned");
    double* const s_tab = reinterpret_cast<double*>(smem + RING_BYTES);
    // LDS byte address of smem (256-aligned), for hand-built LDS addresses
    const int lds0 = static_cast<int>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)smem));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // tell the compiler it is wave-uniform
    const int W = A.W, R = A.R;
    // lane <-> (alignment g of the wave, lane j inside it)
    auto lane_of = [&](int gg, int jj) -> int { return ROWF == 2 ? ((gg >> 1) << 4) + (jj << 1) + (gg & 1) : gg * W + jj; };
    const int g = ROWF == 2 ? (((lane >> 4) << 1) | (lane & 1)) : lane / W;
    const int j = ROWF == 2 ? ((lane & 15) >> 1) : lane - g * W;
    const bool lane_on = g < A.ngroups;
    const bool leader = (j == 0);
    const int c0 = j * K + 1;
    const double NEG_INF = -__builtin_huge_val();
    const double GO = A.GO, GE = A.GE;
    const int GOhi = hi32(GO), GOlo = lo32(GO), GEhi = hi32(GE), GElo = lo32(GE);

    for (int x = threadIdx.x; x < A.tab_doubles; x += 64 * NWAVES) s_tab[x] = A.tables[x];
    uint16_t* const s_ring = reinterpret_cast<uint16_t*>(smem) + wave * NG * SLOT;
    int32_t* const s_map = reinterpret_cast<int32_t*>(s_tab + A.tab_doubles) + wave * A.ngroups * (R + 1);

    double vgo[K], vge[K], rz[K];
    int colbase[K];  // LDS byte address of the table rows this column reads (see build_tables)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = c0 + k;
        // columns past R (last lane when K does not divide R) reuse column R's tables: they
        // compute garbage that nothing reads (their outputs only feed a leader or an idle lane)
        const int cc = c <= R ? c : R;
        colbase[k] = lds0 + RING_BYTES + static_cast<int>(A.colbase[cc]);
        // only column R is "last"; with KLAST known it sits at k == KLAST, every other k takes the penalties from SGPRs
        const bool last = LOCAL && cc == R && (KLAST < 0 || k == KLAST);
        vgo[k] = last ? 0.0 : GO;
        vge[k] = last ? 0.0 : GE;
        rz[k] = A.rowzero[cc];
    }
    const double rz_left = A.rowzero[c0 - 1 <= R ? c0 - 1 : R];
    const int jlast = (R - 1) / K, klast = KLAST >= 0 ? KLAST : (R - 1) % K;
    const long long gwave = static_cast<long long>(blockIdx.x) * NWAVES + wave;
    const long long nwaves = static_cast<long long>(gridDim.x) * NWAVES;
    Word* const scr = static_cast<Word*>(A.dirs) + static_cast<size_t>(gwave) * A.dirs_per_wave;
    // MODE 3: the tile holds the codes of one window, the snapshots follow it
    constexpr int NSV = 2 * K + 3;  // doubles per lane in a snapshot
    double* const snap = reinterpret_cast<double*>(scr + static_cast<size_t>(A.snap_win / UNR) * 64);
    const int ring_g = lds0 + (wave * NG + g) * static_cast<int>(SLOT * sizeof(uint16_t));  // byte address of this alignment's ring
    __syncthreads();

    const long long nitems = (A.n + A.ngroups - 1) / A.ngroups;
    for (long long item = gwave; item < nitems; item += nwaves) {
        const long long read = item * A.ngroups + g;
        const bool valid = lane_on && read < A.n;
        long long start = 0;
        int L = 0;
        if (valid) {
            start = A.off[read];
            L = static_cast<int>(A.off[read + 1] - start);
        }
        // wave-uniform description of the (up to NG) reads of this work item, kept in SGPRs
        long long gstart[NG];
        int glen[NG];
        int Lmax = 0, Lmin = 0x7fffffff;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            const int src = gg < A.ngroups ? lane_of(gg, 0) : 0;
            const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(start), src));
            const int hi = __builtin_amdgcn_readlane(static_cast<int>(start >> 32), src);
            const int len = __builtin_amdgcn_readlane(L, src);
            gstart[gg] = (static_cast<long long>(hi) << 32) | lo;
            glen[gg] = gg < A.ngroups ? len : 0;
            Lmax = max(Lmax, glen[gg]);
            // reads past the end of the batch (last work item) compute garbage nobody stores
            if (gg < A.ngroups && item * A.ngroups + gg < A.n) Lmin = min(Lmin, glen[gg]);
        }

        // Read staging, one refill (64 positions per alignment) ahead of use: lane -> (alignment sg = lane / 16, four
        // consecutive positions), so one pass of the wave stages four alignments (two passes for eight).  A staged entry
        // is the byte offset of (base code, quality) inside a block of table rows; base codes 0-3 = ACGT, 4 = anything else.
        constexpr int NPASS = NG / 4;
        const int sq = (lane & 15) * 4;
        int sgs[NPASS], slens[NPASS], stoffs[NPASS];   // per pass: the alignment this lane stages, its length, its window start (toff[sg])
        long long sstarts[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            sgs[ps] = (lane >> 4) + 4 * ps;
            const int sgl = sgs[ps] < A.ngroups ? lane_of(sgs[ps], 0) : 0;
            const int slen_any = __shfl(L, sgl);   // every lane takes part: a source lane switched off would read as 0
            slens[ps] = sgs[ps] < A.ngroups ? slen_any : 0;
            sstarts[ps] = (static_cast<long long>(__shfl(static_cast<int>(start >> 32), sgl)) << 32) |
                          static_cast<unsigned>(__shfl(static_cast<int>(start), sgl));
            stoffs[ps] = 0;
        }
        struct Pf { uint32_t q, b; };   // four qualities (one per byte); four 2-bit base codes | four exception bits << 8
        auto fetch4 = [&](int slen, long long sstart, int r0) -> Pf {
            const int r = r0 + sq;
            const int nv = min(max(slen - r, 0), 4);
            Pf v{0u, 0u};
            if (nv == 0) return v;
            const long long idx = sstart + r;
            if (nv == 4) {
                __builtin_memcpy(&v.q, A.qual + idx, 4);   // every byte read lies inside the read (no over-read of caller memory)
                if (A.nmask) {  // 2-bit packed bases + exception mask (sarlacc_dev_pack_reads)
                    const uint32_t two = (A.seq[idx >> 2] | (static_cast<uint32_t>(A.seq[(idx + 3) >> 2]) << 8)) >> ((idx & 3) * 2);
                    const uint32_t exc = (A.nmask[idx >> 3] | (static_cast<uint32_t>(A.nmask[(idx + 3) >> 3]) << 8)) >> (idx & 7);
                    v.b = (two & 0xffu) | ((exc & 0xfu) << 8);
                } else {
                    uint32_t s4;
                    __builtin_memcpy(&s4, A.seq + idx, 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t b = (s4 >> (8 * e)) & 0xffu;
                        const uint32_t x = (b >> 1) & 3u, code = x ^ (x >> 1);         // A C G T -> 0 1 2 3
                        const uint32_t exc = ((0x54474341u >> (8 * code)) & 0xffu) != b;   // anything else
                        v.b |= (code << (2 * e)) | (exc << (8 + e));
                    }
                }
                return v;
            }
            for (int e = 0; e < nv; ++e) {   // the last positions of a read
                v.q |= static_cast<uint32_t>(A.qual[idx + e]) << (8 * e);
                uint32_t code, exc;
                if (A.nmask) {
                    code = (A.seq[(idx + e) >> 2] >> (((idx + e) & 3) * 2)) & 3u;
                    exc = (A.nmask[(idx + e) >> 3] >> ((idx + e) & 7)) & 1u;
                } else {
                    const uint32_t b = A.seq[idx + e];
                    code = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u;
                    exc = !(b == 'A' || b == 'C' || b == 'G' || b == 'T');
                }
                v.b |= (code << (2 * e)) | (exc << (8 + e));
            }
            return v;
        };
        auto stage4 = [&](int sg, int slen, int r0, Pf v) {
            const int r = r0 + sq;
            uint32_t ent[4];
            bool bad = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int qi = static_cast<int>(static_cast<signed char>((v.q >> (8 * e)) & 0xffu)) - A.qoffset;
                bad = bad || (r + e < slen && qi < 0);
                qi = qi < 0 ? 0 : (qi >= A.navail ? A.navail - 1 : qi);
                const uint32_t code = ((v.b >> (8 + e)) & 1u) ? 4u : ((v.b >> (2 * e)) & 3u);
                ent[e] = (r + e < slen) ? code * static_cast<uint32_t>(A.row_bytes) + static_cast<uint32_t>(qi << 3) : 0u;
            }
            if (bad) atomicMin(A.badqual, A.read_base + static_cast<int>(item * A.ngroups + sg));
            const uint2 w = make_uint2(ent[0] | (ent[1] << 16), ent[2] | (ent[3] << 16));
            uint16_t* const slot = s_ring + sg * SLOT + (r & (RING - 1));   // r is a multiple of 4: 8-byte aligned
            *reinterpret_cast<uint2*>(slot) = w;
            if ((r & (RING - 1)) < RING_MIRROR) *reinterpret_cast<uint2*>(slot + RING) = w;
        };
        Pf pf[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) pf[ps] = fetch4(slens[ps], sstarts[ps], 0);
        int toff[NG];  // first step of the window being recomputed (MODE 3), per alignment; 0 in the fill
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) toff[gg] = 0;

        // per-column state: score of the previous row, vertical jump score (and, PENSEL only,
        // the penalty the next vertical step pays)
        double S[K], UJ[K];
        int vp_hi[K], vp_lo[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { S[k] = rz[k]; UJ[k] = NEG_INF; vp_hi[k] = hi32(vgo[k]); vp_lo[k] = lo32(vgo[k]); }
        // per-row state arriving from the lane to the left: score, horizontal jump score (PENSEL:
        // and the penalty a horizontal step from that cell pays).  Leaders keep column-0 constants.
        double s_in = 0.0, lj_in = NEG_INF, diag_prev = rz_left;
        int ph_in = GOhi, pl_in = GOlo;
        // Row position as x2 = 2 * (i - 1), i = t - j: doubles as the ring byte offset.
        int x2 = -2 * j - 2;
        int x2max = valid ? 2 * L - 2 : -2;
        // Landing row of the upward walk the traceback performs in column R (tracked by the lane
        // owning that column, in x2 units): land[i] = i if the move at (i,R) is not vertical, else
        // the landing row of the cell the vertical jump leads to.  Without it the walk up the last
        // column (free vertical gaps => ~L single steps in local mode) costs ~L dependent loads.
        int land_prev = -2, land_up = -2;
        int vnl = 0;  // the move at (i-1, R) was a vertical gap
        // MODE 3 tracks the landing row directly: in local mode the vertical gaps of column R are
        // free, every row of a vertical run resets the jump (its candidate S[i-1] equals the
        // running jump score, and the comparison is strict), so the walk up column R stops at the
        // last row whose move is not vertical, i.e. the last row with best > V.
        int land_x2 = -2;

        // Steps [t_begin, t_end).  GUARD = false is the steady state: every lane of every
        // alignment of the wave is inside its read, nothing is predicated and all flags are
        // lane masks in SGPRs.
        // TR: 0 no traceback, 1 codes streamed to the per-wave tile (MODE 1/2), 2 landing row +
        // snapshots (MODE 3 fill), 3 codes of a recomputed window (MODE 3; t counts from toff[])
        auto run = [&](auto guard_tag, auto trace_tag, int t_begin, int t_end) {
            constexpr bool GUARD = decltype(guard_tag)::value;
            constexpr int TR = decltype(trace_tag)::value;
            constexpr bool CODES = TR == 1 || TR == 3;
            mask_t m_vnl = GUARD ? 0 : __builtin_amdgcn_ballot_w64(vnl != 0);
            for (int t0 = t_begin; t0 < t_end; t0 += UNR) {
                if ((t0 & 63) == 0) {
                    int t0s = t0;   // opaque copy: keeps the staging addresses out of the step loop's induction variables
                    asm volatile("" : "+s"(t0s));
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        stage4(sgs[ps], slens[ps], stoffs[ps] + t0s, pf[ps]);
                        pf[ps] = fetch4(slens[ps], sstarts[ps], stoffs[ps] + t0s + 64);
                    }
                    if (TR == 2 && (t0 & (SNAP_P - 1)) == 0) {
                        // complete lane state before step t0
                        double* sp = snap + static_cast<size_t>(t0 / SNAP_P) * (NSV * 64) + lane;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            __builtin_nontemporal_store(S[k], sp + (2 * k) * 64);
                            __builtin_nontemporal_store(UJ[k], sp + (2 * k + 1) * 64);
                        }
                        __builtin_nontemporal_store(s_in, sp + (2 * K) * 64);
                        __builtin_nontemporal_store(lj_in, sp + (2 * K + 1) * 64);
                        __builtin_nontemporal_store(diag_prev, sp + (2 * K + 2) * 64);
                    }
                }
                Word pk = 0;
                double v_first = 0.0;   // TR 2: vertical candidate of column R at the block's first step
                // ring address of the block's first row; the following rows sit behind it (mirror: no wrap inside a block)
                const uint32_t ring_blk = static_cast<uint32_t>(ROWF == 2 ? ring_g + (x2 & 0xff) : (ring_g | (x2 & 0xff)));
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (ROWF == 0) {
                        // column 0 of the DP (src/reference_align.cpp:63-78): what a leader lane consumes
                        const int im1 = x2 >> 1;
                        const double col0 = (LOCAL || im1 < 0) ? 0.0 : (-GO - GE * static_cast<double>(im1));
                        if (leader) { s_in = col0; lj_in = NEG_INF; ph_in = GOhi; pl_in = GOlo; }
                    }
                    double left = s_in, lj = lj_in;
                    int ph = ph_in, pl = pl_in;
                    const double s_two_back = diag_prev;  // what s_in held two steps ago
                    bool act = true;
                    if (GUARD) act = x2 >= 0 && x2 <= x2max;
                    if (act) {
                        const int rd = *reinterpret_cast<lds_cu16*>(ring_blk + 2u * u);
                        double diag = diag_prev;
                        diag_prev = s_in;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            // Every "if (a > b) {x = a} else {a = b}" pair of the reference leaves
                            // max(a, b) in both; no NaN or -0 can occur on this path, so fmax is the
                            // same value bit for bit (src/reference_align.cpp:125-158).
                            const double hcand = left - (PENSEL ? mk64(ph, pl) : GO);
                            const double ljm = lj - GE;
                            const bool b_hj = ljm > hcand;
                            const double H = fmax(ljm, hcand);
                            lj = H;
                            const double vcand = S[k] - (PENSEL ? mk64(vp_hi[k], vp_lo[k]) : vgo[k]);
                            const double ujm = UJ[k] - vge[k];
                            const bool b_vj = ujm > vcand;
                            const double V = fmax(ujm, vcand);
                            UJ[k] = V;
                            const double w = *reinterpret_cast<lds_cdouble*>(static_cast<uint32_t>(colbase[k] + rd));
                            const double M = diag + w;
                            diag = S[k];
                            // (:164-174): M only if greater than both, else H only if greater than V
                            const bool b_hv = H > V;
                            const double G = fmax(H, V);
                            const bool b_tm = M > G;
                            const double best = fmax(M, G);
                            S[k] = best;
                            left = best;
                            const bool is_last = KLAST >= 0 ? (k == KLAST) : (k == klast);
                            if (!GUARD) {
                                const mask_t m_hj = __builtin_amdgcn_ballot_w64(b_hj), m_vj = __builtin_amdgcn_ballot_w64(b_vj);
                                const mask_t m_hv = __builtin_amdgcn_ballot_w64(b_hv), m_tm = __builtin_amdgcn_ballot_w64(b_tm);
                                const mask_t m_vn = ~(m_hv | m_tm);  // the move here is a vertical gap
                                if (PENSEL) {
                                    const mask_t m_hp = m_hv & ~m_tm;  // the move here is a horizontal gap
                                    ph = sel32(GOhi, GEhi, m_hp);
                                    pl = sel32(GOlo, GElo, m_hp);
                                    vp_hi[k] = sel32(hi32(vgo[k]), hi32(vge[k]), m_vn);
                                    vp_lo[k] = sel32(lo32(vgo[k]), lo32(vge[k]), m_vn);
                                }
                                if (TR == 2 && is_last) {
                                    // Column R's score never decreases down the rows (free vertical gaps), so
                                    // "some row of this block improved it" is one comparison per block: the
                                    // score after the block against the running maximum before it.  The
                                    // block's last row is recorded; the walk steps up the <= UNR - 1 vertical
                                    // moves to the exact landing row through the recomputed codes.
                                    if (u == 0) v_first = V;
                                    if (u == UNR - 1) land_x2 = sel32(land_x2, x2, __builtin_amdgcn_ballot_w64(best > v_first));
                                }
                                if (CODES) {
                                    if (ADDC) {
                                        pk = push_bit(push_bit(push_bit(push_bit(static_cast<uint32_t>(pk), m_vj), m_hj), m_tm), m_hv);
                                    } else {
                                        const unsigned raw = (b_hv ? 1u : 0u) | (b_tm ? 2u : 0u) | (b_hj ? 4u : 0u) | (b_vj ? 8u : 0u);
                                        pk |= static_cast<Word>(raw) << (4 * (CELLS - 1 - (u * K + k)));
                                    }
                                    if (TR == 1 && is_last) {
                                        const mask_t m_cont = m_vj & ~m_vnl;  // the vertical jump really continued
                                        land_up = sel32(land_prev, land_up, m_cont);  // continued: same landing row
                                        land_prev = sel32(land_up, x2, ~m_vn);
                                        m_vnl = m_vn;
                                    }
                                }
                            } else {
                                const bool hp = b_hv && !b_tm, vn = !b_hv && !b_tm;
                                if (PENSEL) {
                                    ph = hp ? GEhi : GOhi;
                                    pl = hp ? GElo : GOlo;
                                    vp_hi[k] = vn ? hi32(vge[k]) : hi32(vgo[k]);
                                    vp_lo[k] = vn ? lo32(vge[k]) : lo32(vgo[k]);
                                }
                                if (TR == 2) {
                                    if (is_last && best > V) land_x2 = x2;
                                }
                                if (CODES) {
                                    const unsigned raw = (b_hv ? 1u : 0u) | (b_tm ? 2u : 0u) | (b_hj ? 4u : 0u) | (b_vj ? 8u : 0u);
                                    pk |= static_cast<Word>(raw) << (4 * (CELLS - 1 - (u * K + k)));
                                    if (TR == 1 && is_last) {
                                        const bool cont = b_vj && vnl == 0;
                                        land_up = cont ? land_up : land_prev;
                                        land_prev = vn ? land_up : x2;
                                        vnl = vn ? 1 : 0;
                                    }
                                }
                            }
                        }
                    }
                    // hand the row state to the next lane; with 16-lane groups the leaders have no
                    // source lane and keep their column-0 values
                    if (ROW16 && !LOCAL) {
                        const int i = (x2 >> 1) + 1;  // global mode: column 0 changes with the row
                        const double col0_next = (i < 0) ? 0.0 : (-GO - GE * static_cast<double>(i));
                        s_in = lane_shr1<ROWF>(left, col0_next);
                    } else if (ROW16 && !GUARD) {
                        // s_in itself stays live as the next step's diagonal; the register that
                        // held it two steps ago is free and, on leader lanes, holds the same 0.0
                        s_in = lane_shr1<ROWF>(left, s_two_back);
                    } else {
                        s_in = lane_shr1<ROWF>(left, s_in);
                    }
                    lj_in = lane_shr1<ROWF>(lj, lj_in);
                    if (PENSEL) {
                        ph_in = lane_shr1<ROWF>(ph, ph_in);
                        pl_in = lane_shr1<ROWF>(pl, pl_in);
                    }
                    x2 += 2;
                }
                if (TR == 1) __builtin_nontemporal_store(pk, scr + static_cast<size_t>(t0 / UNR) * 64 + lane);
                if (TR == 3) scr[static_cast<size_t>(t0 / UNR) * 64 + lane] = pk;
            }
            if (!GUARD) vnl = sel32(0, 1, m_vnl);
        };

        const int nsteps = ((Lmax + W + UNR - 1) / UNR) * UNR;
        int t_a = ((W + UNR - 1) / UNR) * UNR;                 // every lane has entered its read
        int t_b = Lmin == 0x7fffffff ? 0 : ((Lmin + 1) / UNR) * UNR;  // first block leaving the shortest read
        t_a = min(t_a, nsteps);
        t_b = min(max(t_b, t_a), nsteps);
        constexpr int TR_FILL = MODE == 3 ? 2 : (MODE >= 1 ? 1 : 0);
        run(Flag<true>{}, Int<TR_FILL>{}, 0, t_a);
        run(Flag<false>{}, Int<TR_FILL>{}, t_a, t_b);
        run(Flag<true>{}, Int<TR_FILL>{}, t_b, nsteps);

        if (valid && j == jlast) {
            double sc = S[0];
#pragma unroll
            for (int k = 1; k < K; ++k) sc = (k == klast) ? S[k] : sc;
            A.scores[read] = sc;
        }

        if (MODE == 3) {
            int32_t* map = s_map + g * (R + 1);
            const Word* const wtile = scr;
            // walk state of the group's leader: position, and the jump chain being measured
            int row = (__shfl(land_x2, lane_of(g, jlast)) >> 1) + 1, c = R;
            int phase = 0, chain_n = 0, chain_at = 0;
            unsigned cur = 0;
            int want = row;         // lowest row the next window has to hold
            int pending = valid ? 1 : 0;
            // every round moves at least one alignment to an earlier snapshot, so the number of rounds
            // is bounded; the explicit bound guarantees that the wave leaves the loop whatever the codes say
            int rounds_left = (Lmax + W) / SNAP_P + 8;
            while (__builtin_amdgcn_ballot_w64(pending != 0)) {
                if (--rounds_left < 0) {
                    if (lane == 0) atomicExch(A.badqual + 1, 1);
                    break;
                }
                // ---- recompute the window that holds row `want` (per alignment) ----
                const int ts = pending ? (max(want - A.snap_head, 0) / SNAP_P) * SNAP_P : 0;
                // steps [0, nwin) of the window; [t_wa, t_wb) of them have every lane of every
                // pending alignment inside its read (the others compute garbage nobody reads)
                int nwin = 0, t_wa = 0, t_wb = 0x7fffffff;
#pragma unroll
                for (int gg = 0; gg < NG; ++gg) {
                    const int src = gg < A.ngroups ? lane_of(gg, 0) : 0;
                    toff[gg] = __builtin_amdgcn_readlane(ts, src);
                    const int pend = __builtin_amdgcn_readlane(pending, src);
                    const int need = __builtin_amdgcn_readlane(want + W + 1 - ts, src);
                    if (gg < A.ngroups && pend) {
                        nwin = max(nwin, need);
                        if (toff[gg] < W) t_wa = W;                 // lanes still entering the read
                        t_wb = min(t_wb, glen[gg] + 1 - toff[gg]);  // first step leaving it
                    }
                }
                nwin = min(((nwin + UNR - 1) / UNR) * UNR, A.snap_win);
                t_wa = min(((t_wa + UNR - 1) / UNR) * UNR, nwin);
                t_wb = min(max((t_wb / UNR) * UNR, t_wa), nwin);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const double* sp = snap + static_cast<size_t>(ts / SNAP_P) * (NSV * 64) + lane;
#pragma unroll
                for (int k = 0; k < K; ++k) { S[k] = sp[(2 * k) * 64]; UJ[k] = sp[(2 * k + 1) * 64]; }
                s_in = sp[(2 * K) * 64];
                lj_in = sp[(2 * K + 1) * 64];
                diag_prev = sp[(2 * K + 2) * 64];
                x2 = 2 * (ts - j - 1);
                x2max = pending ? 2 * L - 2 : -2;
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int q = lane >> 4;   // sgs[ps] = q + 4 * ps
                    stoffs[ps] = q == 0 ? toff[4 * ps] : q == 1 ? toff[4 * ps + 1] : q == 2 ? toff[4 * ps + 2] : toff[4 * ps + 3];
                    if (stoffs[ps] >= 64) stage4(sgs[ps], slens[ps], stoffs[ps] - 64, fetch4(slens[ps], sstarts[ps], stoffs[ps] - 64));
                    pf[ps] = fetch4(slens[ps], sstarts[ps], stoffs[ps]);
                }
                run(Flag<true>{}, Int<3>{}, 0, t_wa);
                run(Flag<false>{}, Int<3>{}, t_wa, t_wb);
                run(Flag<true>{}, Int<3>{}, t_wb, nwin);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

                // ---- leaders walk (src/reference_align.cpp:231-278) until done or out of window ----
                if (pending && leader) {
                    // raw compare bits of cell (rr, cc), see nibble() below; false: above the window
                    auto code_at = [&](int cc, int rr, unsigned& out) -> bool {
                        const int jj = (cc - 1) / K, kk = (cc - 1) % K;
                        const int tt = rr + jj - ts;
                        if (tt < 0) return false;
                        const Word w = wtile[static_cast<size_t>(tt / UNR) * 64 + lane_of(g, jj)];
                        out = static_cast<unsigned>(w >> (4 * (CELLS - 1 - ((tt % UNR) * K + kk)))) & 15u;
                        return true;
                    };
                    bool stalled = false;
                    while (c > 0 && !stalled) {
                        if (phase == 0) {
                            // runs of diagonal moves (most of the path) in a loop of their own
                            while (c > 0 && row > 0) {
                                const int jj = (c - 1) / K, kk = (c - 1) % K;
                                const int tt = row + jj - ts;
                                if (tt < 0) break;
                                const Word w = wtile[static_cast<unsigned>(tt / UNR) * 64u + static_cast<unsigned>(lane_of(g, jj))];
                                if (!((w >> (4 * (CELLS - 1 - ((tt % UNR) * K + kk)))) & 2u)) break;
                                map[c] = row * 2 + 1;
                                --row; --c;
                            }
                            if (c <= 0) break;
                            if (row <= 0) { map[c] = (row + 1) * 2; --c; continue; }  // D[c][0] = 1
                            unsigned nb;
                            if (!code_at(c, row, nb)) { want = row; stalled = true; break; }
                            if (nb & 2u) { map[c] = row * 2 + 1; --row; --c; continue; }
                            cur = nb;
                            chain_n = 0;
                            phase = (nb & 1u) ? 1 : 2;
                            chain_at = (nb & 1u) ? c : row;
                        }
                        if (phase == 1) {  // horizontal jump: count the columns it continued through
                            while ((cur & 4u) && chain_at > 1) {
                                unsigned prev;
                                if (!code_at(chain_at - 1, row, prev)) { want = row; stalled = true; break; }
                                if ((prev & 3u) == 1u) break;
                                ++chain_n; --chain_at; cur = prev;
                            }
                            if (stalled) break;
                            for (int x = 0; x <= chain_n; ++x) { map[c] = (row + 1) * 2; --c; }
                            phase = 0;
                        } else {           // vertical jump: count the rows it continued through
                            while ((cur & 8u) && chain_at > 1) {
                                unsigned prev;
                                if (!code_at(c, chain_at - 1, prev)) { want = chain_at - 1; stalled = true; break; }
                                if ((prev & 3u) == 0u) break;
                                ++chain_n; --chain_at; cur = prev;
                            }
                            if (stalled) break;
                            row -= 1 + chain_n;  // up moves leave the map untouched (:286)
                            phase = 0;
                        }
                    }
                    if (!stalled) {
                        pending = 0;
                        // (src/reference_align.cpp:307-351), size_t wrap kept via unsigned
                        auto interval = [&](int a, int b, bool gaps, unsigned& s, unsigned& e) {
                            if (!gaps) {
                                s = map[a + 1] >> 1;
                                e = (map[b] >> 1) + (map[b] & 1);
                            } else {
                                s = (a == 0) ? 1u : static_cast<unsigned>((map[a] >> 1) + (map[a] & 1));
                                e = (b + 1 == R + 1) ? static_cast<unsigned>(L + 1) : static_cast<unsigned>(map[b + 1] >> 1);
                            }
                            s -= 1;
                            e -= 1;
                        };
                        unsigned s, e;
                        interval(0, R, false, s, e);
                        const bool nonempty = s < e;
                        A.starts[read] = nonempty ? static_cast<int32_t>(s + 1) : 0;
                        A.ends[read] = nonempty ? static_cast<int32_t>(e) : 0;
                        for (int x = 0; x < A.nsec; ++x) {
                            interval(A.sec_s[x], A.sec_e[x], true, s, e);
                            A.sec_so[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(s + 1);
                            A.sec_wo[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(e - s);
                        }
                    }
                }
                pending = __shfl(pending, lane_of(g, 0));
                want = __shfl(want, lane_of(g, 0));
            }
        }

        if (MODE == 1 || MODE == 2) {
            // make this wave's traceback stores visible to its leader lanes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int land = (__shfl(land_prev, lane_of(g, jlast)) >> 1) + 1;  // where the walk up column R ends
            if (valid && leader) {
                // raw compare bits of cell (row, c): 1 H>V, 2 M>max(H,V), 4 horizontal jump
                // continuation beat the step from the left, 8 same for the vertical jump
                auto nibble = [&](int c, int row) -> unsigned {  // 1 <= c <= R, 1 <= row <= L
                    const int jj = (c - 1) / K, kk = (c - 1) % K;
                    const int tt = row + jj;
                    const Word w = scr[static_cast<size_t>(tt / UNR) * 64 + lane_of(g, jj)];
                    return static_cast<unsigned>(w >> (4 * (CELLS - 1 - ((tt % UNR) * K + kk)))) & 15u;
                };
                // direction value the reference would have stored at (row, c); a jump continues
                // through a cell only if the cell it came from was not itself entered by the
                // same kind of gap (see "Penalty selection" above)
                auto loadD = [&](int c, int row) -> int {
                    if (row <= 0) return 1;  // D[c][0] = 1 (src/reference_align.cpp:118)
                    unsigned nb = nibble(c, row);
                    if (nb & 2u) return 0;
                    int n = 0;
                    if (nb & 1u) {
                        int x = c;
                        while ((nb & 4u) && x > 1) {
                            const unsigned prev = nibble(x - 1, row);
                            if ((prev & 3u) == 1u) break;
                            ++n; --x; nb = prev;
                        }
                        return 1 + n;
                    }
                    int y = row;
                    while ((nb & 8u) && y > 1) {
                        const unsigned prev = nibble(c, y - 1);
                        if ((prev & 3u) == 0u) break;
                        ++n; --y; nb = prev;
                    }
                    return -(1 + n);
                };
                int row = L, c = R;
                if (MODE == 1) {
                    int32_t* map = s_map + g * (R + 1);
                    row = land;  // up moves leave the map untouched (src/reference_align.cpp:286)
                    while (c > 0) {
                        int d = loadD(c, row);
                        while (row > 0 && d < 0) { row += d; d = loadD(c, row); }
                        if (d == 0) { map[c] = row * 2 + 1; --row; --c; }
                        else { for (int x = 0; x < d; ++x) { map[c] = (row + 1) * 2; --c; } }
                    }
                    // (src/reference_align.cpp:307-351), size_t wrap kept via unsigned
                    auto interval = [&](int a, int b, bool gaps, unsigned& s, unsigned& e) {
                        if (!gaps) {
                            s = map[a + 1] >> 1;
                            e = (map[b] >> 1) + (map[b] & 1);
                        } else {
                            s = (a == 0) ? 1u : static_cast<unsigned>((map[a] >> 1) + (map[a] & 1));
                            e = (b + 1 == R + 1) ? static_cast<unsigned>(L + 1) : static_cast<unsigned>(map[b + 1] >> 1);
                        }
                        s -= 1;
                        e -= 1;
                    };
                    unsigned s, e;
                    interval(0, R, false, s, e);
                    const bool nonempty = s < e;
                    A.starts[read] = nonempty ? static_cast<int32_t>(s + 1) : 0;
                    A.ends[read] = nonempty ? static_cast<int32_t>(e) : 0;
                    for (int x = 0; x < A.nsec; ++x) {
                        interval(A.sec_s[x], A.sec_e[x], true, s, e);
                        A.sec_so[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(s + 1);
                        A.sec_wo[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(e - s);
                    }
                } else {
                    // gapped strings, emitted from the end (src/reference_align.cpp:353-389)
                    const long long base = start + read * static_cast<long long>(R);
                    uint8_t* oref = A.aln_ref + base;
                    uint8_t* oqry = A.aln_qry + base;
                    const uint8_t* sq = A.seq + start;
                    int m = 0, ed = 0;
                    for (; row > land; --row) { oref[m] = '-'; oqry[m] = sq[row - 1]; ++m; ++ed; }
                    while (c > 0) {
                        int d = loadD(c, row);
                        while (row > 0 && d < 0) {
                            for (int x = 0; x < -d; ++x) { oref[m] = '-'; oqry[m] = sq[row - 1]; ++m; ++ed; --row; }
                            d = loadD(c, row);
                        }
                        if (d == 0) {
                            const uint8_t rc = A.refchars[c - 1], qc = sq[row - 1];
                            oref[m] = rc; oqry[m] = qc; ++m;
                            ed += rc != qc;
                            --row; --c;
                        } else {
                            for (int x = 0; x < d; ++x) { oref[m] = A.refchars[c - 1]; oqry[m] = '-'; ++m; ++ed; --c; }
                        }
                    }
                    while (row > 0) { oref[m] = '-'; oqry[m] = sq[row - 1]; ++m; ++ed; --row; }
                    A.aln_len[read] = m;
                    A.edits[read] = ed;
                }
            }
        }
    }
}

// Empty reference: the DP has only column 0 (src/reference_align.cpp:63-78,:104).
__global__ void k_align_emptyref(const int64_t* off, long long n, int local, double GO, double GE,
                                 double* scores, int32_t* starts, int32_t* ends, int nsec,
                                 int32_t* sec_so, int32_t* sec_wo, int32_t* aln_len, int32_t* edits,
                                 const uint8_t* seq, uint8_t* aln_ref, uint8_t* aln_qry) {
    const long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (r >= n) return;
    const int L = static_cast<int>(off[r + 1] - off[r]);
    scores[r] = (local || L < 1) ? 0.0 : (-GO - GE * static_cast<double>(L - 1));
    if (starts) { starts[r] = 0; ends[r] = 0; }
    for (int x = 0; x < nsec; ++x) { sec_so[x * n + r] = 1; sec_wo[x * n + r] = 0; }
    if (aln_len) {
        // every read base sits opposite a gap; strings are stored reversed
        for (int m = 0; m < L; ++m) { aln_ref[off[r] + m] = '-'; aln_qry[off[r] + m] = seq[off[r] + L - 1 - m]; }
        aln_len[r] = L;
        edits[r] = L;
    }
}

// ---------------------------------------------------------------------------
// References of more than MAX_REF columns (qualityAlign against a transcript or a genomic
// region, src/general_align.cpp:12-16; src/reference_align.cpp:7-13 takes any length).
// k_align keeps one alignment inside a wavefront (at most 64 lanes x 16 columns); here ONE
// WORKGROUP holds the alignment: thread t owns K consecutive columns, threads are skewed by one
// read row exactly as k_align's lanes are, and the row state (score of the column to the left,
// "that cell was a horizontal gap", running horizontal jump score) goes to the next thread
// through a double-buffered LDS slot, one barrier per step.  The recurrence is the reference's
// statement for statement (explicit penalty selects, strict >, src/reference_align.cpp:108-181),
// fp64, -ffp-contract=off.  Traceback: 4 bits per cell -- move (0 diagonal, 1 horizontal,
// 2 vertical) + "horizontal jump continued here" + "vertical jump continued here" -- one word per
// thread and step, stored by step so that a step's words are contiguous; the jump LENGTHS the
// reference stores (:131,147) are 1 + the run of continued flags that ends in the cell, and the
// walk (thread 0) counts them.  Read positions are staged WIDE_CH rows at a time into an LDS
// ring as (base code * row bytes + quality * 8), the byte offset inside a column's cost rows.
constexpr int WIDE_CH = 512;      // rows staged per refill
constexpr int WIDE_RING = 2048;   // ring entries (> WIDE_CH + 1024 threads)
constexpr int WIDE_MAXT = 1024;

template <int K>
struct WideWord { using type = uint32_t; };

// The walk of one alignment of k_align_wide / k_align_wide_q, run by ONE wavefront (lane = 0 .. 63) after the codes of every
// strip are in `dirs` (word of thread tt at step s: dirs[strip * strip_words + s * Rw + tt], column k of the thread in nibble K - 1 - k).
// k_align_wide_q's band of steps that carry codes, per wavefront w of an alignment of L rows against R columns (K = 8 columns per
// lane): the cells within `band` rows of the main diagonal i = c L / R sit, for the 512 columns of wavefront w, at its steps
// [lo, hi] (lane l is at row sg - l + 1).  The fill switches its codes on one step before lo (the flags a cell's codes take from the
// row above are then in place at lo); the walk may read the codes of a cell iff its step lies in [lo, hi] or in the first 63
// steps of its wavefront, which always carry codes.
struct WideBand {
    int band, L, R;
    __device__ __forceinline__ int lo(int w) const { return static_cast<int>(512ll * w * L / R) - band - 2; }
    __device__ __forceinline__ int hi(int w) const { return static_cast<int>((512ll * (w + 1) * L + R - 1) / R) + band + 64; }
    __device__ __forceinline__ bool stored(int c, int row) const {   // 1 <= c <= R, 1 <= row <= L
        if (band <= 0) return true;
        const int t = (c - 1) >> 3, w = t >> 6, sg = row + (t & 63) - 1;
        return sg < 63 || (sg >= lo(w) && sg <= hi(w));
    }
};

// (`wb`: the band of stored codes, null = every cell; a walk that needs a cell outside it stops and returns false -- nothing it wrote counts)
template <int K, int MODE, bool STRIPS>
__device__ __forceinline__ bool wide_walk(const AlignArgs& A, typename WideWord<K>::type* dirs, long long strip_words, int CS, long long Rw,
                                          int L, int R, long long read, long long start, int lane, const WideBand* wb = nullptr) {
    using Word = typename WideWord<K>::type;
    bool left_band = false;
    auto ok = [&](int c, int row) -> bool { return !wb || wb->stored(c, row); };
    // The walk, on the first wavefront: the path of a global alignment of like sequences is mostly diagonal, so lane m
    // looks at the cell m steps up the diagonal from (row, c), a ballot gives the length of the diagonal run and its
    // moves are written side by side -- one memory round trip per run instead of one per cell; gaps are taken one
    // run at a time (every lane follows, the outputs are spread over the lanes).
    auto nibble = [&](int c, int row) -> unsigned {   // 1 <= c <= R, 1 <= row <= L
        const int sc = STRIPS ? (c - 1) / CS : 0, cc = (c - 1) - sc * CS;   // strip, column inside it
        const int tt = cc / K, kk = cc % K;
        const Word w = dirs[sc * strip_words + static_cast<long long>(row + tt) * Rw + tt];
        return static_cast<unsigned>(w >> (4 * (K - 1 - kk))) & 15u;
    };
    // direction value the reference stores at (row, c): 0 diagonal, +length horizontal, -length vertical
    auto loadD = [&](int c, int row) -> int {
        if (row <= 0) return 1;   // D[c][0] = 1 (:118)
        if (!ok(c, row)) { left_band = true; return 0; }
        const unsigned nb = nibble(c, row);
        if ((nb & 3u) == 0u) return 0;
        int len = 1;
        if ((nb & 3u) == 1u) {
            unsigned f = nb;
            for (int x = c; (f & 4u) && x > 1;) {
                ++len; --x;
                if (!ok(x, row)) { left_band = true; break; }
                f = nibble(x, row);
            }
            return len;
        }
        unsigned f = nb;
        for (int y = row; (f & 8u) && y > 1;) {
            ++len; --y;
            if (!ok(c, y)) { left_band = true; break; }
            f = nibble(c, y);
        }
        return -len;
    };
    auto diag_run = [&](int c, int row) -> int {   // diagonal moves from (row, c) on, at most 64 (a cell without codes ends the run: the walk meets it next)
        const int rr = row - lane, cc = c - lane;
        const bool stop = !(rr >= 1 && cc >= 1) || !ok(cc, rr) || (nibble(cc, rr) & 3u) != 0u;
        const unsigned long long nd = __ballot(stop);
        return nd ? static_cast<int>(__builtin_ctzll(nd)) : 64;
    };
    int row = L, c = R;
    if (MODE == 1) {
        const unsigned long long map_words = (static_cast<unsigned long long>(R + 1) * 4 + sizeof(Word) - 1) / sizeof(Word);
        int32_t* const map = reinterpret_cast<int32_t*>(dirs + (A.dirs_per_wave - map_words));   // (behind the codes)
        while (c > 0) {
            const int run = diag_run(c, row);
            if (run > 0) {
                if (lane < run) map[c - lane] = (row - lane) * 2 + 1;
                row -= run; c -= run;
                continue;
            }
            const int d = loadD(c, row);
            if (left_band) return false;
            if (d < 0) { row += d; continue; }   // up moves leave the map untouched (:286)
            for (int x = lane; x < d && c - x > 0; x += 64) map[c - x] = (row + 1) * 2;
            c -= min(d, c);
        }
        __threadfence();
        if (lane == 0) {
            auto interval = [&](int a, int b, bool gaps, unsigned& st, unsigned& en) {   // (:307-351), size_t wrap kept via unsigned
                if (!gaps) {
                    st = map[a + 1] >> 1;
                    en = (map[b] >> 1) + (map[b] & 1);
                } else {
                    st = (a == 0) ? 1u : static_cast<unsigned>((map[a] >> 1) + (map[a] & 1));
                    en = (b + 1 == R + 1) ? static_cast<unsigned>(L + 1) : static_cast<unsigned>(map[b + 1] >> 1);
                }
                st -= 1;
                en -= 1;
            };
            unsigned st, en;
            interval(0, R, false, st, en);
            const bool nonempty = st < en;
            A.starts[read] = nonempty ? static_cast<int32_t>(st + 1) : 0;
            A.ends[read] = nonempty ? static_cast<int32_t>(en) : 0;
            for (int x = 0; x < A.nsec; ++x) {
                interval(A.sec_s[x], A.sec_e[x], true, st, en);
                A.sec_so[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(st + 1);
                A.sec_wo[static_cast<long long>(x) * A.sec_stride + read] = static_cast<int32_t>(en - st);
            }
        }
    } else {   // gapped strings, emitted from the end (:353-389)
        const long long base = start + read * static_cast<long long>(R);
        uint8_t* const oref = A.aln_ref + base;
        uint8_t* const oqry = A.aln_qry + base;
        const uint8_t* const sq = A.seq + start;
        int m = 0, ed = 0;
        while (c > 0) {
            const int run = diag_run(c, row);
            if (run > 0) {
                bool diff = false;
                if (lane < run) {
                    const uint8_t rc = A.refchars[c - 1 - lane], qc = sq[row - 1 - lane];
                    oref[m + lane] = rc; oqry[m + lane] = qc;
                    diff = rc != qc;
                }
                ed += static_cast<int>(__popcll(__ballot(diff)));
                m += run; row -= run; c -= run;
                continue;
            }
            const int d = loadD(c, row);
            if (left_band) return false;
            if (d < 0) {   // read bases opposite a gap
                for (int x = lane; x < -d; x += 64) { oref[m + x] = '-'; oqry[m + x] = sq[row - 1 - x]; }
                m -= d; ed -= d; row += d;
            } else {       // reference characters opposite a gap
                const int dd = min(d, c);
                for (int x = lane; x < dd; x += 64) { oref[m + x] = A.refchars[c - 1 - x]; oqry[m + x] = '-'; }
                m += dd; ed += dd; c -= dd;
            }
        }
        for (int x = lane; x < row; x += 64) { oref[m + x] = '-'; oqry[m + x] = sq[row - 1 - x]; }
        m += max(row, 0); ed += max(row, 0);
        if (lane == 0) { A.aln_len[read] = m; A.edits[read] = ed; }
    }
    return true;
}

// MODE 0 scores, 1 scores + map + sections, 2 scores + strings + edit distance.  PENSEL: the reference's penalty selects spelled
// out (needed when gapopen < 0).  Otherwise the "Penalty selection" argument of k_align applies cell for cell: a gap step out of
// a cell that was itself entered by the same kind of gap finds the running jump score equal to that cell's score, bit for bit,
// so with open >= extension max(jump - extension, score - open) is the double the reference computes either way, and its
// "jump continued" flag is the raw comparison AND NOT "that cell was entered by this kind of gap" -- both at hand here, so
// the codes, and the walk, are the same in both instantiations.
template <int K, int MODE, bool PENSEL, bool STRIPS>
__global__ void __launch_bounds__(WIDE_MAXT) k_align_wide(const AlignArgs A) {
    using Word = typename WideWord<K>::type;
    extern __shared__ __align__(16) unsigned char w_smem[];
    const int T = static_cast<int>(blockDim.x);
    const int t = static_cast<int>(threadIdx.x);
    const int R = A.R;
    // References of more than T K columns go through in STRIPS of T K columns, one after the other for every alignment: the last
    // column of a strip leaves its row state (score, running horizontal jump score, "was a horizontal gap") in a boundary array
    // in the workgroup's scratch, row by row, and the first thread of the next strip takes its left inputs from there (staged
    // WIDE_CH rows at a time like the read) instead of from DP column 0.  Codes: one tile per strip.
    const int CS = T * K;                                             // columns per strip
    const int nstrips = STRIPS ? (R + CS - 1) / CS : 1;               // (STRIPS = false: the instantiation for references of one strip, without the boundary code)
    double* const s_tab = reinterpret_cast<double*>(w_smem);
    double* const h_s = s_tab + A.tab_doubles;                        // [2][T] score handed to the next thread
    double* const h_lj = h_s + 2 * T;                                 // [2][T] running horizontal jump score
    double* const b_s = h_lj + 2 * T;                                 // [WIDE_CH] boundary rows staged for thread 0: score ...
    double* const b_lj = b_s + WIDE_CH;                               // ... jump score ...
    int* const h_fl = reinterpret_cast<int*>(b_lj + WIDE_CH);         // [2][T] the cell was a horizontal gap
    int* const b_fl = h_fl + 2 * T;                                   // [WIDE_CH] ... and flag
    uint16_t* const s_ring = reinterpret_cast<uint16_t*>(b_fl + WIDE_CH);   // [WIDE_RING]
    for (int e = t; e < A.tab_doubles; e += T) s_tab[e] = A.tables[e];
    const double NEG_INF = -__builtin_huge_val();
    const double GO = A.GO, GE = A.GE;
    const bool local = A.local != 0;
    const long long Rw = T;                        // words per step
    const long long strip_words = (static_cast<long long>(A.wide_maxlen) + T + 1) * Rw;
    Word* const dirs = MODE ? reinterpret_cast<Word*>(A.dirs) + static_cast<size_t>(blockIdx.x) * A.dirs_per_wave : nullptr;
    // boundary arrays of the workgroup (two sets, strips alternate): [2][3][longest read + 2] doubles
    const size_t bstride = static_cast<size_t>(A.wide_maxlen) + 2;
    double* const bnd = A.wide_bnd + static_cast<size_t>(blockIdx.x) * 6 * bstride;
    double Sc[K], Dg[K], UJ[K];
    for (long long read = blockIdx.x; read < A.n; read += gridDim.x) {
        const long long start = A.off[read];
        const int L = static_cast<int>(A.off[read + 1] - start);
        for (int st = 0; st < nstrips; ++st) {
        __syncthreads();
        const int col0 = st * CS;                                     // columns col0 + 1 .. col0 + Rs
        const int Rs = min(CS, R - col0);
        const int ncolv = max(0, min(K, Rs - t * K));                 // this thread's columns inside the strip
        const int tR = (Rs - 1) / K;
        const bool bnd_in = STRIPS && st > 0, bnd_out = STRIPS && st + 1 < nstrips;
        const double* const bi_s = bnd + static_cast<size_t>((st + 1) & 1) * 3 * bstride;   // written by the strip before
        double* const bo_s = bnd + static_cast<size_t>(st & 1) * 3 * bstride;
        Word* const sdirs = MODE ? dirs + st * strip_words : nullptr;
        int cb[K];
        unsigned upneg = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {   // DP row 0 of the thread's columns and of the column to their left
            const int col = min(col0 + t * K + k + 1, R);
            cb[k] = static_cast<int>(A.colbase[col]);
            Sc[k] = A.rowzero[col]; Dg[k] = A.rowzero[col - 1]; UJ[k] = NEG_INF;
        }
        const int nsteps = L > 0 ? L + tR : 0;   // thread tR finishes row L at step L + tR
        for (int s = 1; s <= nsteps; ++s) {
            if ((s - 1) % WIDE_CH == 0) {   // rows s .. s + WIDE_CH - 1 into the ring (thread 0 needs row s now)
                for (int e = t; e < WIDE_CH; e += T) {
                    const int row = s + e;
                    uint32_t ent = 0;
                    if (row <= L) {
                        const long long idx = start + row - 1;
                        int qi = static_cast<int>(static_cast<signed char>(A.qual[idx])) - A.qoffset;
                        if (qi < 0) atomicMin(A.badqual, A.read_base + static_cast<int>(read));
                        qi = qi < 0 ? 0 : (qi >= A.navail ? A.navail - 1 : qi);
                        uint32_t code;
                        if (A.nmask) {
                            code = ((A.nmask[idx >> 3] >> (idx & 7)) & 1u) ? 4u : ((A.seq[idx >> 2] >> ((idx & 3) * 2)) & 3u);
                        } else {
                            const uint32_t b = A.seq[idx];
                            code = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u;
                        }
                        ent = code * static_cast<uint32_t>(A.row_bytes) + static_cast<uint32_t>(qi << 3);
                        if (bnd_in) {   // the row state the strip before left in its last column
                            b_s[e] = bi_s[row]; b_lj[e] = bi_s[bstride + row];
                            b_fl[e] = static_cast<int>(reinterpret_cast<const long long*>(bi_s + 2 * bstride)[row]);
                        }
                    }
                    s_ring[row & (WIDE_RING - 1)] = static_cast<uint16_t>(ent);
                }
                __syncthreads();
            }
            const int i = s - t;
            const int par = s & 1;
            if (i >= 1 && i <= L && ncolv > 0) {
                double ls, lj;
                bool lpos;
                if (t == 0) {
                    if (bnd_in) {   // (thread 0 is at row s: entry (s - 1) % WIDE_CH of the staged boundary rows)
                        const int e = (i - 1) % WIDE_CH;
                        ls = b_s[e]; lj = b_lj[e]; lpos = b_fl[e] != 0;
                    } else {        // DP column 0 (src/reference_align.cpp:63-78)
                        ls = local ? 0.0 : (-GO - GE * static_cast<double>(i - 1));
                        lj = NEG_INF;
                        lpos = false;
                    }
                } else {
                    ls = h_s[(par ^ 1) * T + t - 1];
                    lj = h_lj[(par ^ 1) * T + t - 1];
                    lpos = h_fl[(par ^ 1) * T + t - 1] != 0;
                }
                const int ent = s_ring[i & (WIDE_RING - 1)];
                Word w = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k < ncolv) {
                        const bool lastcol = local && (col0 + t * K + k + 1 == R);   // free vertical gaps in the last column (:93)
                        const double vgo = lastcol ? 0.0 : GO, vge = lastcol ? 0.0 : GE;
                        const bool upn = ((upneg >> k) & 1u) != 0u;
                        double horiz, vert;
                        bool hc, vc;
                        if (PENSEL) {
                            horiz = ls - (lpos ? GE : GO);
                            lj -= GE;
                            hc = lj > horiz;
                            if (hc) horiz = lj; else lj = horiz;
                            vert = Sc[k] - (upn ? vge : vgo);
                            double uj = UJ[k] - vge;
                            vc = uj > vert;
                            if (vc) vert = uj; else uj = vert;
                            UJ[k] = uj;
                        } else {
                            const double hopen = ls - GO, vopen = Sc[k] - vgo;
                            lj -= GE;
                            hc = lj > hopen && !lpos;
                            horiz = fmax(lj, hopen);
                            lj = horiz;
                            const double uj = UJ[k] - vge;
                            vc = uj > vopen && !upn;
                            vert = fmax(uj, vopen);
                            UJ[k] = vert;
                        }
                        const double cost = *reinterpret_cast<const double*>(reinterpret_cast<const unsigned char*>(s_tab) + cb[k] + ent);
                        const double match = Dg[k] + cost;
                        Dg[k] = ls;
                        // (:164-177) the diagonal if it beats both gaps, else the horizontal gap if it beats the vertical one; the score is
                        // the maximum of the three either way (no NaN, no -0 on this path: the selected operand bit for bit)
                        const double hv = fmax(horiz, vert);
                        const double cur = fmax(match, hv);
                        const unsigned kind = match > hv ? 0u : (horiz > vert ? 1u : 2u);
                        Sc[k] = cur;
                        upneg = (upneg & ~(1u << k)) | ((kind == 2u ? 1u : 0u) << k);
                        ls = cur;
                        lpos = kind == 1u;
                        if (MODE) w |= static_cast<Word>(kind | (hc ? 4u : 0u) | (vc ? 8u : 0u)) << (4 * (K - 1 - k));   // (column k in nibble K - 1 - k: what k_align_wide_q's bit pushes leave)
                    }
                }
                h_s[par * T + t] = ls;
                h_lj[par * T + t] = lj;
                h_fl[par * T + t] = lpos ? 1 : 0;
                if (bnd_out && t == tR) {   // the strip's last column: its row state for the next strip
                    bo_s[i] = ls; bo_s[bstride + i] = lj;
                    reinterpret_cast<long long*>(bo_s + 2 * bstride)[i] = lpos ? 1 : 0;
                }
                if (MODE) __builtin_nontemporal_store(w, &sdirs[static_cast<long long>(s) * Rw + t]);
            }
            __syncthreads();
        }
        if (bnd_out) __threadfence();
        if (st + 1 < nstrips) continue;
        const int kR = (Rs - 1) % K;
        if (t == tR) {
            double sc = Sc[0];
#pragma unroll
            for (int k = 1; k < K; ++k) sc = (k == kR) ? Sc[k] : sc;
            if (L == 0) sc = A.rowzero[R];   // no rows: row 0 of the last column (:115-118)
            A.scores[read] = sc;
        }
        if (MODE) {
            __threadfence();
            __syncthreads();
            if (t < 64) (void)wide_walk<K, MODE, STRIPS>(A, dirs, strip_words, CS, Rw, L, R, read, start, t);
        }
        }   // (strips)
    }
}

// ---------------------------------------------------------------------------
// k_align_wide_q: the same alignment without a barrier per step (global mode, gapopen >= 0, one strip: the qualityAlign shape).
// k_align_wide hands the row state from thread to thread through LDS with one workgroup barrier per step; measured at 2 kb x
// 2 kb the vector units idle half of the time -- every step is a round trip LDS write -> barrier -> LDS read in front of a
// chain of 8 dependent cells.  Here a WAVEFRONT is the unit: inside it the row state goes to the next lane with DPP moves
// (wave_shr:1, lane 0 keeps the fill operand), exactly as in k_align, and only between consecutive wavefronts does it go
// through LDS -- a queue of WQ_B rows per wavefront boundary, filled by lane 63 of the producer and read by lane 0 of the
// consumer, which runs 64 rows behind by construction.  The two look at each other's progress counters once per WQ_SYNC
// steps (consumer: "are my next WQ_SYNC rows there", producer: "are the slots of my next WQ_SYNC rows free") and otherwise
// run freely; every wait is bounded and sets the abort flag, which every wavefront sees at its next look, so the grid drains
// whatever happens.  Read rows are staged per wavefront, 64 at a time one refill ahead, in a 128-entry ring (k_align's
// scheme).  All flags are lane masks in scalar registers, inactive lanes (ramp-up and ramp-down of the skew) are switched
// off by EXEC alone.  Codes, tile layout and walk are k_align_wide's.
constexpr int WQ_B = 128;       // rows of a queue (power of two; producer and consumer are 64 rows apart + WQ_SYNC of slack each way)
constexpr int WQ_SYNC = 16;     // steps between two looks at the neighbours' progress
constexpr int WQ_SPINS = 1 << 22;
constexpr int WIDE_BAND = 96;   // rows either side of the main diagonal whose cells carry codes in the first launch (reads with 1 % indel events drift by a few dozen)

__device__ __forceinline__ int wq_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wq_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait until *p >= need; false: gave up (another wavefront did, or the bound was reached) -- the caller leaves its loop
__device__ __forceinline__ bool wq_wait(const int* p, int need, int* abort_flag) {
    for (int spins = 0; wq_load(p) < need; ++spins) {
        if (wq_load(abort_flag) != 0 || spins > WQ_SPINS) { wq_store(abort_flag, 1); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}
__device__ __forceinline__ double dpp_wave_shr1(double v, double lane0) {
    const int hi = __builtin_amdgcn_update_dpp(hi32(lane0), hi32(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const int lo = __builtin_amdgcn_update_dpp(lo32(lane0), lo32(v), 0x138, 0xf, 0xf, false);
    return mk64(hi, lo);
}

// v_max_f64 as the instruction itself: fmax() of a value that arrives through a loop-carried register makes the compiler
// canonicalise it first (one more v_max_f64 x, x, x per use), which costs what carrying the value saves; what follows a
// raw maximum in the chain of a cell is raw too.  (No NaN reaches these: maximum of two numbers, bit for bit the greater.)
__device__ __forceinline__ double max_f64_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int K, int MODE>
__global__ void __launch_bounds__(WIDE_MAXT) k_align_wide_q(const AlignArgs A) {
    using Word = typename WideWord<K>::type;
    extern __shared__ __align__(16) unsigned char w_smem[];
    const int T = static_cast<int>(blockDim.x);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int NWV = T >> 6;
    // Which 512 columns a wavefront takes is rotated from workgroup to workgroup: hardware wavefront h of every workgroup sits on
    // SIMD h mod 4, and with the codes confined to a band the wavefronts near the diagonal have more to issue than the others --
    // without the rotation the four SIMDs of a CU take turns at being the busy one.  `wv` and `t` are the LOGICAL indices (the
    // wavefront's position along the reference, the thread's position among the column owners) everywhere below.
    const int hw = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int wv = (hw + static_cast<int>((blockIdx.x * 2654435761u) >> 16)) % NWV;
    const int t = wv * 64 + lane;
    const int R = A.R;
    double* const s_tab = reinterpret_cast<double*>(w_smem);
    double* const q_s = s_tab + A.tab_doubles;                          // [NWV][WQ_B] score of a wavefront's last column, by row
    double* const q_lj = q_s + NWV * WQ_B;                              // ... its running horizontal jump score
    int* const q_fl = reinterpret_cast<int*>(q_lj + NWV * WQ_B);        // ... "that cell was a horizontal gap"
    int* const s_prog = q_fl + NWV * WQ_B;                              // [NWV] rows a wavefront's lane 63 has written
    int* const s_cons = s_prog + NWV;                                   // [NWV] rows a wavefront's lane 0 has taken
    int* const s_abort = s_cons + NWV;                                  // [1] (+ padding)
    uint16_t* const s_ring = reinterpret_cast<uint16_t*>(s_abort + 2) + wv * 128;   // [NWV][128] staged read rows of this wavefront
    for (int e = t; e < A.tab_doubles; e += T) s_tab[e] = A.tables[e];
    const double NEG_INF = -__builtin_huge_val();
    const double GO = A.GO, GE = A.GE;
    Word* const dirs = MODE ? reinterpret_cast<Word*>(A.dirs) + static_cast<size_t>(blockIdx.x) * A.dirs_per_wave : nullptr;
    const int tR = (R - 1) / K, kR = (R - 1) % K;
    // (threads beyond column R -- the spare columns of thread tR, the padding of the last wavefront -- compute on column R's
    // tables values nobody reads)
    const bool has_next = wv + 1 < NWV;
    int cb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cb[k] = static_cast<int>(A.colbase[min(t * K + k + 1, R)]);
    const unsigned char* const tabb = reinterpret_cast<const unsigned char*>(s_tab);
    // Per column: Sg = score of the row above MINUS the opening penalty (what the vertical candidate of this row and the
    // horizontal candidate of the next column both are: one subtraction per cell serves both), Dg = score of the column to the
    // left in the row above (this row's diagonal), UJ = running vertical jump score.
    double Sg[K], Dg[K], UJ[K];
    // the reads of this launch: 0 .. n - 1, or (the launch behind a banded one) the reads whose walks left the band
    const long long nwork = A.wide_list ? static_cast<long long>(A.wide_list[-1]) : A.n;
    for (long long item = blockIdx.x; item < nwork; item += gridDim.x) {
        const long long read = A.wide_list ? static_cast<long long>(A.wide_list[item]) : item;
        const long long start = A.off[read];
        const int L = static_cast<int>(A.off[read + 1] - start);
        __syncthreads();   // (the walk of the alignment before is done with the tile; the tables are in place)
        if (lane == 0) { s_prog[wv] = 0; s_cons[wv] = 0; }
        if (t == 0) *s_abort = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {   // DP row 0 of the thread's columns and of the column to their left
            const int col = min(t * K + k + 1, R);
            Sg[k] = A.rowzero[col] - GO; Dg[k] = A.rowzero[col - 1]; UJ[k] = NEG_INF;
        }
        double ls_out = 0.0, lj_out = NEG_INF;
        double final_score = A.rowzero[R];   // (no rows: row 0 of the last column, :115-118)
        // A read row is requested one refill (64 steps) before it is staged: the loads (quality byte, base byte or its two packed
        // bytes) are issued and left in flight, the entry -- byte offset inside a column's cost rows -- is made from them when the
        // ring takes it, so no step waits for memory.
        struct RawRow { uint32_t q, b, m; };
        auto request_row = [&](int row) -> RawRow {
            RawRow r{0u, 0u, 0u};
            if (row > L) return r;
            const long long idx = start + row - 1;
            r.q = A.qual[idx];
            if (A.nmask) { r.m = A.nmask[idx >> 3]; r.b = A.seq[idx >> 2]; }
            else r.b = A.seq[idx];
            return r;
        };
        auto entry_of = [&](const RawRow& r, int row) -> uint32_t {
            if (row > L) return 0u;
            const long long idx = start + row - 1;
            int qi = static_cast<int>(static_cast<signed char>(r.q)) - A.qoffset;
            if (qi < 0) atomicMin(A.badqual, A.read_base + static_cast<int>(read));
            qi = qi < 0 ? 0 : (qi >= A.navail ? A.navail - 1 : qi);
            uint32_t code;
            if (A.nmask) code = ((r.m >> (idx & 7)) & 1u) ? 4u : ((r.b >> ((idx & 3) * 2)) & 3u);
            else code = r.b == 'A' ? 0u : r.b == 'C' ? 1u : r.b == 'G' ? 2u : r.b == 'T' ? 3u : 4u;
            return code * static_cast<uint32_t>(A.row_bytes) + static_cast<uint32_t>(qi << 3);
        };
        RawRow pf = request_row(1 + lane);
        const int nsteps = L > 0 ? L + 63 : 0;   // lane 63 finishes row L at step L + 62
        // Codes for the cells near the main diagonal only (wide_band > 0): the steps [c_lo, c_hi] of this wavefront.  Outside them
        // a step still takes the two comparisons that say which move a cell made -- the flags the next row, the next column and the
        // next wavefront take from it stay right everywhere -- but not the two jump comparisons, the four bit pushes and the store.
        const WideBand wband{MODE ? A.wide_band : 0, L, R};
        const bool banded = MODE != 0 && A.wide_band > 0;
        const int c_lo = banded ? wband.lo(wv) : 0, c_hi = banded ? wband.hi(wv) : 0x7fffffff;
        bool aborted = false;
        // what every step starts with: the ring refill and, every WQ_SYNC steps, the look at the neighbours; false: give up
        auto step_head = [&](int sg) -> bool {
            if ((sg & 63) == 0) {   // rows sg + 1 .. sg + 64 into the ring (lane 0 needs row sg + 1 now); the next 64 requested
                s_ring[(sg + lane) & 127] = static_cast<uint16_t>(entry_of(pf, sg + 1 + lane));
                pf = request_row(sg + 65 + lane);
            }
            if ((sg & (WQ_SYNC - 1)) == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (lane 63's queue entries of the steps before, in front of the counters)
                int ok = 1;
                if (lane == 0) {
                    // what this wavefront has taken and written so far -- published before it waits for anybody
                    wq_store(&s_cons[wv], min(sg, L));
                    wq_store(&s_prog[wv], min(max(sg - 63, 0), L));
                    if (wv > 0) ok = wq_wait(&s_prog[wv - 1], min(L, sg + WQ_SYNC), s_abort) ? 1 : 0;                    // my next rows are there
                    if (ok && has_next) ok = wq_wait(&s_cons[wv + 1], min(L, sg - 47) - WQ_B, s_abort) ? 1 : 0;         // the slots of my next rows are free
                    if (wq_load(s_abort) != 0) ok = 0;
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (!ok) return false;
            }
            return true;
        };
        // the row state of the column to lane 0's left at step sg (row sg + 1): DP column 0 (first wavefront,
        // src/reference_align.cpp:63-78, global mode) or the queue of the wavefront before
        auto lane0_in = [&](int sg, double& l0, double& j0, int& f0) {
            if (wv == 0) {
                l0 = -GO - GE * static_cast<double>(sg);   // -GO - GE (i - 1)
                j0 = NEG_INF;
                f0 = 0;
            } else {
                const int e = (wv - 1) * WQ_B + (sg & (WQ_B - 1));   // row sg + 1 sits in slot (row - 1) mod WQ_B
                l0 = q_s[e]; j0 = q_lj[e];
                f0 = MODE ? q_fl[e] : 0;
            }
        };

        // ---- steps 0 .. 62: the lanes enter the read one by one.  Per-lane flags, the cells under `if (active)` (a lane that has
        // not entered keeps DP row 0 in its columns; also every step of a read of fewer than 64 rows' start) ----
        unsigned upneg = 0;      // bit k: the move of column k's cell in the row above was a vertical gap (row 0: no)
        int lpos_out = 0;        // my last column's cell was a horizontal gap, of the step before
        int sg = 0;
        for (; sg < min(nsteps, 63); ++sg) {
            if (!step_head(sg)) { aborted = true; break; }
            const int i = sg - lane + 1;
            const bool active = i >= 1 && i <= L;
            double l0, j0;
            int f0;
            lane0_in(sg, l0, j0, f0);
            double ls = dpp_wave_shr1(ls_out, l0);
            double lj = dpp_wave_shr1(lj_out, j0);
            bool lpos = __builtin_amdgcn_update_dpp(f0, lpos_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false) != 0;
            if (active) {
                const int ent = s_ring[(sg - lane) & 127];
                uint32_t w = 0;
                double hopen = ls - GO;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double vopen = Sg[k];
                    const double ljm = lj - GE;
                    const bool hc = ljm > hopen && !lpos;
                    const double horiz = fmax(ljm, hopen);
                    lj = horiz;
                    const double ujm = UJ[k] - GE;
                    const bool vc = ujm > vopen && ((upneg >> k) & 1u) == 0u;
                    const double vert = max_f64_raw(ujm, vopen);
                    UJ[k] = vert;
                    const double match = Dg[k] + *reinterpret_cast<const double*>(tabb + cb[k] + ent);
                    Dg[k] = ls;
                    const double hv = max_f64_raw(horiz, vert);
                    const double cur = max_f64_raw(match, hv);
                    const unsigned kind = match > hv ? 0u : (horiz > vert ? 1u : 2u);
                    upneg = (upneg & ~(1u << k)) | ((kind == 2u ? 1u : 0u) << k);
                    lpos = kind == 1u;
                    if (MODE) w |= static_cast<uint32_t>(kind | (hc ? 4u : 0u) | (vc ? 8u : 0u)) << (4 * (K - 1 - k));
                    ls = cur;
                    hopen = cur - GO;
                    Sg[k] = hopen;
                    if (t == tR && i == L && k == kR) final_score = cur;   // (a read of fewer than 64 rows can end here)
                }
                ls_out = ls;
                lj_out = lj;
                lpos_out = lpos ? 1 : 0;
                if (MODE) __builtin_nontemporal_store(static_cast<Word>(w), &dirs[static_cast<long long>(i + t) * T + t]);
                if (has_next && lane == 63) {
                    const int e = wv * WQ_B + ((i - 1) & (WQ_B - 1));
                    q_s[e] = ls; q_lj[e] = lj;
                    if (MODE) q_fl[e] = lpos ? 1 : 0;
                }
            }
        }

        // ---- steps 63 .. L + 62: every lane has entered.  Nothing in the cells is predicated and every flag is a wave-uniform
        // lane mask in scalar registers.  A lane past row L computes all the same; what it computes is never used: codes and
        // queue entries are stored for rows 1 .. L only, the score is taken at row L, and nobody takes the outputs of a lane
        // that is not at a row of the read. ----
        mask_t m_up[K];
#pragma unroll
        for (int k = 0; k < K; ++k) m_up[k] = MODE ? __builtin_amdgcn_ballot_w64(((upneg >> k) & 1u) != 0u) : 0;
        mask_t m_lp_out = MODE ? __builtin_amdgcn_ballot_w64(lpos_out != 0) : 0;
        // (three loops -- before, inside and behind the steps that carry codes -- rather than one loop with both versions of the cells:
        // with both in one loop the allocator spilled in the cells, 65 -> 102 ms)
        auto main_steps = [&](auto coded_tag, int s_end) {
          constexpr bool CODED = decltype(coded_tag)::value;
          for (; !aborted && sg < s_end; ++sg) {
            if (!step_head(sg)) { aborted = true; break; }
            const int i = sg - lane + 1;
            const bool active = i <= L;
            double l0, j0;
            int f0;
            lane0_in(sg, l0, j0, f0);
            double ls = dpp_wave_shr1(ls_out, l0);
            double lj = dpp_wave_shr1(lj_out, j0);
            mask_t m_lp = (m_lp_out << 1) | (MODE ? (__builtin_amdgcn_ballot_w64(f0 != 0) & 1ull) : 0ull);
            const int ent = s_ring[(sg - lane) & 127];   // (a lane past the read finds the 0 the refill staged for rows beyond L: a valid table address)
            uint32_t w = 0;
            double cost[K], curs[K];
#pragma unroll
            for (int k = 0; k < K; ++k) cost[k] = *reinterpret_cast<const double*>(tabb + cb[k] + ent);   // (requested back to back, ahead of the chain of cells)
            {
                double hopen = ls - GO;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    // (src/reference_align.cpp:125-177 without the penalty selects: see k_align's "Penalty selection")
                    const double vopen = Sg[k];
                    const double ljm = lj - GE;
                    const mask_t m_hcr = CODED ? __builtin_amdgcn_ballot_w64(ljm > hopen) : 0;
                    const double horiz = fmax(ljm, hopen);
                    lj = horiz;
                    const double ujm = UJ[k] - GE;
                    const mask_t m_vcr = CODED ? __builtin_amdgcn_ballot_w64(ujm > vopen) : 0;
                    const double vert = max_f64_raw(ujm, vopen);
                    UJ[k] = vert;
                    const double match = Dg[k] + cost[k];
                    Dg[k] = ls;
                    const double hv = max_f64_raw(horiz, vert);
                    const double cur = max_f64_raw(match, hv);
                    if (MODE) {
                        const mask_t m_tm = __builtin_amdgcn_ballot_w64(match > hv), m_hv = __builtin_amdgcn_ballot_w64(horiz > vert);
                        const mask_t m_k1 = m_hv & ~m_tm, m_k2 = ~(m_hv | m_tm);   // horizontal gap / vertical gap (neither: diagonal)
                        if (CODED) {
                            const mask_t m_hc = m_hcr & ~m_lp, m_vc = m_vcr & ~m_up[k];   // the jumps really continued
                            w = push_bit(push_bit(push_bit(push_bit(w, m_vc), m_hc), m_k2), m_k1);   // column k ends up in nibble K - 1 - k
                        }
                        m_lp = m_k1;
                        m_up[k] = m_k2;
                    }
                    curs[k] = cur;
                    ls = cur;
                    hopen = cur - GO;   // the next column's horizontal candidate and, one step on, this column's vertical one
                    Sg[k] = hopen;
                }
            }
            ls_out = ls;
            lj_out = lj;
            if (MODE) m_lp_out = m_lp;
            if (sg == L - 1 + (tR & 63) && wv == NWV - 1) {   // thread tR is at row L: the score (one step of one wavefront)
                double sc = curs[0];
#pragma unroll
                for (int k = 1; k < K; ++k) sc = (k == kR) ? curs[k] : sc;
                if (t == tR) final_score = sc;
            }
            if (active) {
                // (word of thread t at its step i + t = sg + 1 + 64 wv, the same for every lane: a scalar row address + the lane's offset)
                if (MODE && CODED) __builtin_nontemporal_store(static_cast<Word>(w), dirs + static_cast<long long>(sg + 1 + 64 * wv) * T + t);
                if (has_next && lane == 63) {   // my last column's row state for the wavefront after me
                    const int e = wv * WQ_B + ((i - 1) & (WQ_B - 1));
                    q_s[e] = ls; q_lj[e] = lj;
                    if (MODE) q_fl[e] = static_cast<int>((m_lp >> 63) & 1ull);
                }
            }
          }
        };
        if (MODE != 0) main_steps(Flag<false>{}, min(nsteps, c_lo));
        main_steps(Flag<MODE != 0>{}, c_hi < nsteps - 1 ? c_hi + 1 : nsteps);
        if (MODE != 0) main_steps(Flag<false>{}, nsteps);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { wq_store(&s_cons[wv], L); wq_store(&s_prog[wv], aborted ? 0 : L); }
        if (aborted && lane == 0) atomicExch(A.badqual + 1, 1);
        if (t == tR) A.scores[read] = final_score;
        if (MODE) {
            __threadfence();
            __syncthreads();
            if (t < 64 && wq_load(s_abort) == 0) {
                const bool done = wide_walk<K, MODE, false>(A, dirs, 0, T * K, T, L, R, read, start, t, banded ? &wband : nullptr);
                if (!done && t == 0) A.wide_redo[1 + atomicAdd(A.wide_redo, 1)] = static_cast<int>(read);   // again, with the codes of every cell
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side

// Per-column lookup info: which fp64 table a column reads, decided by the
// reference character alone (src/reference_align.cpp:184-212, SURVEY App.B Q2).
static int column_info(char r, uint32_t* info) {
    uint32_t code = 7, tmatch, tmis;
    switch (r) {
        case 'A': code = 0; tmatch = 0; tmis = 1; break;
        case 'C': code = 1; tmatch = 0; tmis = 1; break;
        case 'G': code = 2; tmatch = 0; tmis = 1; break;
        case 'T': code = 3; tmatch = 0; tmis = 1; break;
        case 'M': case 'R': case 'W': case 'S': case 'Y': case 'K': tmatch = tmis = 2; break;
        case 'V': case 'H': case 'D': case 'B': tmatch = tmis = 3; break;
        case 'N': tmatch = tmis = 4; break;
        default: return 1;
    }
    *info = code | (tmatch << 8) | (tmis << 16);
    return 0;
}

// (src/reference_align.cpp:21-52) -- built on the host with the host libm so the
// device never evaluates a transcendental on the scoring path.
static void build_tables(const double* errors, int n, std::vector<double>& tab) {
    tab.assign(static_cast<size_t>(5) * n, 0.0);
    const double four = 4.0, ratio = four / (four - 1.0);
    auto odds = [&](double g, double e) { return std::log(g * (1 - e) * four + (1 - g) * e * ratio) / M_LN2; };
    for (int q = 0; q < n; ++q) {
        const double e = errors[q];
        tab[0 * n + q] = odds(1.0 / 1.0, e);          // exact, match      (mode 1 match)
        tab[1 * n + q] = odds(1 - 1.0 / 1.0, e);      // exact, mismatch   (mode 1 mismatch)
        tab[2 * n + q] = odds(1 - 1.0 / 2.0, e);      // 2-fold            (mode 2 mismatch)
        tab[3 * n + q] = odds(1.0 / 3.0, e);          // 3-fold            (mode 3 match)
        tab[4 * n + q] = odds(1.0 / 4.0, e);          // N                 (mode 4 match)
    }
}

// Device layout of the cost table: rows of n doubles, addressed as
//     colbase[column] + (read base code * n + quality) * 8          (one add per cell)
// with read base codes 0-3 = ACGT, 4 = anything else.
//   * columns holding A/C/G/T share eight rows "mis mis mis MATCH mis mis mis mis"; the column
//     of reference base r starts 3 - r rows in, so code == r lands on the match row and every
//     other code (including 4) on a mismatch row;
//   * each ambiguity class present in the reference (2-fold, 3-fold, N) gets five identical
//     rows: its score does not depend on the read base (src/reference_align.cpp:184-212).
static void build_cost_rows(const std::vector<double>& tab, int n, const uint32_t* colinfo, int R,
                            std::vector<double>& rows, std::vector<uint32_t>& colbase) {
    rows.clear();
    auto append = [&](int t) { rows.insert(rows.end(), tab.begin() + static_cast<size_t>(t) * n, tab.begin() + static_cast<size_t>(t + 1) * n); };
    for (int r = 0; r < 8; ++r) append(r == 3 ? 0 : 1);
    uint32_t class_base[5] = {0, 0, 0, 0, 0};
    colbase.assign(static_cast<size_t>(R) + 1, 0);
    const uint32_t row_bytes = static_cast<uint32_t>(n * sizeof(double));
    for (int col = 1; col <= R; ++col) {
        const uint32_t code = colinfo[col] & 0xff, tmatch = (colinfo[col] >> 8) & 0xff;
        if (code < 4) { colbase[col] = (3 - code) * row_bytes; continue; }
        if (!class_base[tmatch]) {
            class_base[tmatch] = static_cast<uint32_t>(rows.size() * sizeof(double));
            for (int r = 0; r < 5; ++r) append(static_cast<int>(tmatch));
        }
        colbase[col] = class_base[tmatch];
    }
    colbase[0] = colbase[R ? 1 : 0];
}

struct Shape { int K, W, ngroups, rowf; };   // rowf: see k_align's ROWF

// Columns per lane / lanes per alignment / alignments per wave for a reference
// of R columns: maximise busy lanes, prefer more columns per lane on ties (fewer
// cross-lane moves per cell).
static Shape pick_shape(int R) {
    Shape best{1, 64, 1, 0};
    double best_u = -1;
    for (int K : {1, 2, 4, 8, 16}) {
        int W = (R + K - 1) / K;
        if (W > 64) continue;
        // groups of up to 16 lanes are padded to one DPP row: NGMAX = 4 of them fill the wave
        // and the leaders get their column-0 inputs for free (lane_shr1<1>)
        if (W <= 16) W = 16;
        const int ng = std::min(64 / W, NGMAX);
        const double u = static_cast<double>(ng) * R / (64.0 * K);
        if (u > best_u + 1e-9 || (u > best_u - 1e-9 && K > best.K && K <= 2)) { best_u = u; best = {K, W, ng, W == 16 ? 1 : 0}; }
    }
    // references of up to 32 columns: eight alignments of 8 lanes, two interleaved per DPP row (lane_shr1<2>) -- the
    // same share of busy lanes as four alignments of 16 lanes with half the columns per lane, and half the
    // cross-lane moves per cell
    for (int K : {2, 4}) {
        if ((R + K - 1) / K > 8) continue;
        const double u = static_cast<double>(NGMAX2) * R / (64.0 * K);
        if (u > best_u - 1e-9) { best_u = u; best = {K, 8, NGMAX2, 2}; break; }
    }
    return best;
}

template <int K, int ROW16, int KLAST, bool PENSEL>
static int launch_mode(int mode, bool local, const AlignArgs& a, int grid, size_t lds, hipStream_t s) {
    // adaptor_align is always local, general_align always global; score-only comes in both
    if (mode == 0 && local) hipLaunchKernelGGL((k_align<K, 0, true, ROW16, KLAST, PENSEL>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else if (mode == 0) hipLaunchKernelGGL((k_align<K, 0, false, ROW16, KLAST, PENSEL>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else if (mode == 1 && local) hipLaunchKernelGGL((k_align<K, 1, true, ROW16, KLAST, PENSEL>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else if (mode == 3 && local && !PENSEL) hipLaunchKernelGGL((k_align<K, 3, true, ROW16, KLAST, false>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else if (mode == 2 && !local) hipLaunchKernelGGL((k_align<K, 2, false, ROW16, KLAST, PENSEL>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else return fail("sarlacc_amd: unsupported alignment mode");
    SL_HIP(hipGetLastError());
    return 0;
}

template <int K, int ROW16, int KLAST>
static int launch_pen(bool pensel, int mode, bool local, const AlignArgs& a, int grid, size_t lds, hipStream_t s) {
    return pensel ? launch_mode<K, ROW16, KLAST, true>(mode, local, a, grid, lds, s)
                  : launch_mode<K, ROW16, KLAST, false>(mode, local, a, grid, lds, s);
}

template <int K, int KLAST>
static int launch_il(int mode, const AlignArgs& a, int grid, size_t lds, hipStream_t s) {
    if (mode == 3) hipLaunchKernelGGL((k_align<K, 3, true, 2, KLAST, false>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    else hipLaunchKernelGGL((k_align<K, 0, true, 2, KLAST, false>), dim3(grid), dim3(64 * NWAVES), lds, s, a);
    SL_HIP(hipGetLastError());
    return 0;
}

static int launch_k(int K, int rowf, int R, bool pensel, int mode, bool local, const AlignArgs& a, int grid, size_t lds,
                    hipStream_t s) {
    const int klast = (R - 1) % K;
    if (rowf == 2) {
        // interleaved alignments: local mode without penalty selects only (adaptor_align by snapshots, score-only)
        if (!local || pensel || !(mode == 3 || mode == 0)) return fail("sarlacc_amd: interleaved alignments serve local modes 0 and 3 only");
        const int key = K * 4 + klast;
        switch (key) {
            case 2 * 4 + 0: return launch_il<2, 0>(mode, a, grid, lds, s);
            case 2 * 4 + 1: return launch_il<2, 1>(mode, a, grid, lds, s);
            case 4 * 4 + 0: return launch_il<4, 0>(mode, a, grid, lds, s);
            case 4 * 4 + 1: return launch_il<4, 1>(mode, a, grid, lds, s);
            case 4 * 4 + 2: return launch_il<4, 2>(mode, a, grid, lds, s);
            case 4 * 4 + 3: return launch_il<4, 3>(mode, a, grid, lds, s);
        }
        return fail("sarlacc_amd: unsupported columns-per-lane %d for interleaved alignments", K);
    }
    const bool row16 = rowf == 1;
    if (K == 1) return row16 ? launch_pen<1, 1, 0>(pensel, mode, local, a, grid, lds, s) : launch_pen<1, 0, 0>(pensel, mode, local, a, grid, lds, s);
    if (K == 2) {
        if (row16) return klast == 0 ? launch_pen<2, 1, 0>(pensel, mode, local, a, grid, lds, s) : launch_pen<2, 1, 1>(pensel, mode, local, a, grid, lds, s);
        return klast == 0 ? launch_pen<2, 0, 0>(pensel, mode, local, a, grid, lds, s) : launch_pen<2, 0, 1>(pensel, mode, local, a, grid, lds, s);
    }
    switch (K) {
        case 4: return row16 ? launch_pen<4, 1, -1>(pensel, mode, local, a, grid, lds, s) : launch_pen<4, 0, -1>(pensel, mode, local, a, grid, lds, s);
        case 8: return row16 ? launch_pen<8, 1, -1>(pensel, mode, local, a, grid, lds, s) : launch_pen<8, 0, -1>(pensel, mode, local, a, grid, lds, s);
        case 16: return row16 ? launch_pen<16, 1, -1>(pensel, mode, local, a, grid, lds, s) : launch_pen<16, 0, -1>(pensel, mode, local, a, grid, lds, s);
    }
    return fail("sarlacc_amd: unsupported columns-per-lane %d", K);
}

struct AlignOut {
    double* d_scores = nullptr;
    int32_t* d_starts = nullptr;
    int32_t* d_ends = nullptr;
    int32_t* d_sec_so = nullptr;
    int32_t* d_sec_wo = nullptr;
    // mode 2
    uint8_t* d_aln_ref = nullptr;
    uint8_t* d_aln_qry = nullptr;
    int32_t* d_aln_len = nullptr;
    int32_t* d_edits = nullptr;
};

// References beyond MAX_REF columns: one workgroup per alignment (k_align_wide).  `a` is complete except for the scratch.
static int launch_wide(AlignArgs& a, int R, int kernel_mode, int32_t max_len, long long n, hipStream_t stream) {
    Context& c = ctx();
    constexpr int K = 8;   // (12 or 16 columns per thread at 2 kb: 3 x slower -- the unrolled columns spill at the 128 registers a 1 024-thread bound leaves)
    if (R > (1 << 20)) return fail("sarlacc_amd: reference longer than %d columns is not supported", 1 << 20);
    if (kernel_mode == 2 && a.nmask) return fail("sarlacc_amd: alignment strings need ASCII reads");
    // up to 8 192 columns one strip of as many threads as the columns need; beyond, strips of 1 024 threads x 8 columns
    const int T = R <= K * WIDE_MAXT ? ((R + K - 1) / K + 63) / 64 * 64 : WIDE_MAXT;
    const int nstrips = (R + T * K - 1) / (T * K);
    const size_t word = 4;
    size_t per_wg = 0;   // words: codes of every step of every strip, then the reference -> read map
    if (kernel_mode) per_wg = static_cast<size_t>(nstrips) * (static_cast<size_t>(max_len) + T + 1) * T + ((static_cast<size_t>(R) + 1) * 4 + word - 1) / word;
    long long grid = std::min<long long>(n, static_cast<long long>(c.num_cu) * std::max(1, 2048 / T));
    if (kernel_mode) {
        const size_t budget = static_cast<size_t>(16) << 30;
        // (4 bits per cell of the whole matrix per alignment in flight: a 2^20-column reference against 100-kb reads would be 50 GB)
        if (per_wg * word > budget)
            return fail("sarlacc_amd: the traceback of one alignment of %d reference columns against reads of up to %d bases needs %.1f GB "
                        "of scratch (at most 16): ask for scores only (barcode_align) or align shorter pieces", R, max_len,
                        static_cast<double>(per_wg * word) / 1073741824.0);
        grid = std::min(grid, std::max<long long>(1, static_cast<long long>(budget / (per_wg * word))));
    }
    void* d_dirs = nullptr;
    if (kernel_mode) SL_TRY(c.buffer("align.dirs", static_cast<size_t>(grid) * per_wg * word, &d_dirs));
    double* d_bnd = nullptr;
    SL_TRY(scratch("align.bnd", nstrips > 1 ? static_cast<size_t>(grid) * 6 * (static_cast<size_t>(max_len) + 2) : 8, &d_bnd));
    a.dirs = d_dirs;
    a.dirs_per_wave = per_wg;
    a.wide_maxlen = max_len;
    a.wide_bnd = d_bnd;
    const size_t lds = sizeof(double) * a.tab_doubles + static_cast<size_t>(T) * (2 * 8 + 2 * 8 + 2 * 4) + static_cast<size_t>(WIDE_CH) * (8 + 8 + 4) +
                       WIDE_RING * sizeof(uint16_t);
    if (lds > 150 * 1024) return fail("sarlacc_amd: alignment tables do not fit in LDS");
    const dim3 g(static_cast<unsigned>(grid)), b(static_cast<unsigned>(T));
    const bool pensel = !(a.GO >= a.GE) || option(OPT_ALIGN_PENSEL) != 0;   // (gapopen < 0, or the tests' switch)
    // global mode, no penalty selects, one strip: the kernel without a barrier per step (k_align_wide_q); align_wide_barrier = 1: the A/B
    if (!pensel && !a.local && nstrips == 1 && !option(OPT_ALIGN_WIDE_BARRIER)) {
        const int NWV = T / 64;
        const size_t ldq = sizeof(double) * a.tab_doubles + static_cast<size_t>(NWV) * WQ_B * (8 + 8 + 4) + static_cast<size_t>(2 * NWV + 2) * 4 +
                           static_cast<size_t>(NWV) * 128 * sizeof(uint16_t);
        if (ldq > 150 * 1024) return fail("sarlacc_amd: alignment tables do not fit in LDS");
#define WIDEQ_LAUNCH(MM)                                                                                                          \
    {                                                                                                                              \
        static thread_local size_t lds_set = 0;   /* (per instantiation: the attribute is raised once, not at every launch) */      \
        if (ldq > 48 * 1024 && ldq > lds_set) {                                                                                    \
            SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_align_wide_q<K, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       static_cast<int>(ldq)));                                                                    \
            lds_set = ldq;                                                                                                         \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_align_wide_q<K, MM>), g, b, ldq, stream, a);                                                         \
    }
        // With traceback: codes only for the steps that can hold cells within WIDE_BAND rows of the main diagonal (40 % of the steps
        // at 2 kb x 2 kb), then a second launch, with the codes of every cell, over the reads whose walks left them (their count
        // stays on the device: no wait in between; none for reads of the reference's molecule).  align_wide_band = -1: no band.
        const int band = kernel_mode == 0 || option(OPT_ALIGN_WIDE_BAND) < 0 ? 0 : (option(OPT_ALIGN_WIDE_BAND) > 0 ? option(OPT_ALIGN_WIDE_BAND) : WIDE_BAND);
        int* d_redo = nullptr;
        if (band > 0) {
            SL_TRY(scratch("align.redo", static_cast<size_t>(n) + 2, &d_redo));
            SL_HIP(hipMemsetAsync(d_redo, 0, sizeof(int), stream));
        }
        a.wide_band = band; a.wide_redo = d_redo; a.wide_list = nullptr;
        if (kernel_mode == 0) WIDEQ_LAUNCH(0) else if (kernel_mode == 1) WIDEQ_LAUNCH(1) else WIDEQ_LAUNCH(2)
        if (band > 0) {
            a.wide_band = 0; a.wide_list = d_redo + 1;
            if (kernel_mode == 1) WIDEQ_LAUNCH(1) else WIDEQ_LAUNCH(2)
        }
#undef WIDEQ_LAUNCH
        SL_HIP(hipGetLastError());
        return 0;
    }
#define WIDE_LAUNCH2(MM, PP, SS)                                                                                                   \
    {                                                                                                                              \
        static thread_local size_t lds_set = 0;   /* (per instantiation: the attribute is raised once, not at every launch) */      \
        if (lds > 48 * 1024 && lds > lds_set) {                                                                                    \
            SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_align_wide<K, MM, PP, SS>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       static_cast<int>(lds)));                                                                    \
            lds_set = lds;                                                                                                         \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_align_wide<K, MM, PP, SS>), g, b, lds, stream, a);                                                   \
    }
#define WIDE_LAUNCH(MM, PP) { if (nstrips > 1) WIDE_LAUNCH2(MM, PP, true) else WIDE_LAUNCH2(MM, PP, false) }
    if (pensel) { if (kernel_mode == 0) WIDE_LAUNCH(0, true) else if (kernel_mode == 1) WIDE_LAUNCH(1, true) else WIDE_LAUNCH(2, true) }
    else { if (kernel_mode == 0) WIDE_LAUNCH(0, false) else if (kernel_mode == 1) WIDE_LAUNCH(1, false) else WIDE_LAUNCH(2, false) }
#undef WIDE_LAUNCH
#undef WIDE_LAUNCH2
    SL_HIP(hipGetLastError());
    return 0;
}

// kernel_mode: 0 scores, 1 map, 2 strings.  Returns in *bad_qual_read the smallest
// index of a read with a quality character below the encoding offset (or INT_MAX).
// A host call may hand its batch over in chunks so that the upload of chunk k+1 overlaps the
// kernel of chunk k: every chunk is one run_align on a slice of the same device arrays.
struct ChunkOpts {
    int64_t sec_stride = 0;   // 0: n (stand-alone launch)
    int read_base = 0;        // index of the slice's first read in the whole batch
    bool init_bad = true;     // reset the bad-quality flag (first chunk only)
    bool finish = true;       // read the flag back and wait for the stream (last chunk only)
};

static int run_align(const uint8_t* d_seq, const uint8_t* d_nmask, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                     int32_t max_len, const double* enc_errors, const char* enc_names, int enc_n,
                     double gapopen, double gapext, const char* ref, int R, bool local, int kernel_mode,
                     const int32_t* sec_starts, const int32_t* sec_ends, int nsec, const AlignOut& out,
                     hipStream_t stream, int* bad_qual_read, const ChunkOpts& co = ChunkOpts()) {
    Context& c = ctx();
    if (co.init_bad) *bad_qual_read = std::numeric_limits<int>::max();
    if (n <= 0) return 0;
    const bool wide = R > MAX_REF;   // one workgroup per alignment instead of (a part of) one wavefront
    if (enc_n > 256) return fail("sarlacc_amd: encoding vector longer than 256 entries");
    if (n > std::numeric_limits<int>::max() - 8) return fail("sarlacc_amd: more than 2^31 reads in one call");
    const double GO = gapopen + gapext, GE = gapext;  // (src/reference_align.cpp:8)

    if (R == 0) {
        const int bs = 256;
        hipLaunchKernelGGL(k_align_emptyref, dim3(static_cast<unsigned>((n + bs - 1) / bs)), dim3(bs), 0, stream,
                           d_off, static_cast<long long>(n), local ? 1 : 0, GO, GE, out.d_scores, out.d_starts,
                           out.d_ends, nsec, out.d_sec_so, out.d_sec_wo, out.d_aln_len, out.d_edits, d_seq,
                           out.d_aln_ref, out.d_aln_qry);
        SL_HIP(hipGetLastError());
        return 0;
    }

    // ---- small per-call tables ----
    std::vector<double> tab;
    build_tables(enc_errors, enc_n, tab);
    std::vector<double> rowzero(R + 1);
    rowzero[0] = 0.0;
    for (int col = 1; col <= R; ++col) rowzero[col] = rowzero[col - 1] - (col == 1 ? GO : GE);  // (:115-118)
    std::vector<uint32_t> colinfo(R + 1, 0);
    for (int col = 1; col <= R; ++col)
        if (column_info(ref[col - 1], &colinfo[col])) colinfo[col] = 7u | (4u << 8) | (4u << 16);  // caller reports the error

    std::vector<double> rows;
    std::vector<uint32_t> colbase;
    build_cost_rows(tab, enc_n, colinfo.data(), R, rows, colbase);
    if (4u * enc_n * sizeof(double) + enc_n * sizeof(double) > 0xffffu) return fail("sarlacc_amd: encoding vector too long for the staged read format");

    AlignArgs a{};
    double* d_tab; double* d_rz; uint32_t* d_cb; uint8_t* d_ref; int32_t* d_ss = nullptr; int32_t* d_se = nullptr; int* d_bad;
    if (co.init_bad) {
        SL_TRY(upload("align.tab", rows.data(), rows.size(), &d_tab, stream));
        SL_TRY(upload("align.rz", rowzero.data(), rowzero.size(), &d_rz, stream));
        SL_TRY(upload("align.cb", colbase.data(), colbase.size(), &d_cb, stream));
        SL_TRY(upload("align.ref", reinterpret_cast<const uint8_t*>(ref), static_cast<size_t>(R), &d_ref, stream));
        if (nsec) {
            SL_TRY(upload("align.ss", sec_starts, static_cast<size_t>(nsec), &d_ss, stream));
            SL_TRY(upload("align.se", sec_ends, static_cast<size_t>(nsec), &d_se, stream));
        }
        const int sentinel[2] = {std::numeric_limits<int>::max(), 0};   // [0] bad-quality read, [1] walk exceeded its bound
        SL_TRY(upload("align.bad", sentinel, 2, &d_bad, stream));
        // a chunk that returns without waiting for its kernel must not leave copies from this
        // frame's vectors in flight (nothing else is queued on the stream yet)
        if (!co.finish) SL_HIP(hipStreamSynchronize(stream));
    } else {   // later chunks of the same call: the tables of the first chunk are still in place
        SL_TRY(scratch("align.tab", rows.size(), &d_tab));
        SL_TRY(scratch("align.rz", rowzero.size(), &d_rz));
        SL_TRY(scratch("align.cb", colbase.size(), &d_cb));
        SL_TRY(scratch("align.ref", static_cast<size_t>(R), &d_ref));
        if (nsec) {
            SL_TRY(scratch("align.ss", static_cast<size_t>(nsec), &d_ss));
            SL_TRY(scratch("align.se", static_cast<size_t>(nsec), &d_se));
        }
        SL_TRY(scratch("align.bad", 2, &d_bad));
    }

    // gapopen >= 0 (GO >= GE): no penalty selects on the device, see k_align
    bool pensel = !(GO >= GE);
    pensel = pensel || option(OPT_ALIGN_PENSEL) != 0;  // testing: force the general path
    Shape sh = wide ? Shape{1, 64, 1, 0} : pick_shape(R);
    // interleaved alignments exist for the local modes without penalty selects (adaptor_align by snapshots, score-only);
    // align_interleave = -1: the A/B without them
    const bool il_mode = local && !pensel && (kernel_mode == 0 || kernel_mode == 1);
    if (sh.rowf == 2 && (!il_mode || option(OPT_ALIGN_INTERLEAVE) < 0)) {
        const int K = R > 16 ? 2 : 1;
        sh = {K, 16, NGMAX, 1};
    }
    if (const int K = option(OPT_ALIGN_K)) {  // tuning override
        const int W = (R + K - 1) / K;
        if ((K == 1 || K == 2 || K == 4 || K == 8 || K == 16) && W <= 64) {
            const int Wp = W <= 16 ? 16 : W;
            sh = {K, Wp, std::min(64 / Wp, NGMAX), Wp == 16 ? 1 : 0};
            if (option(OPT_ALIGN_INTERLEAVE) > 0 && il_mode && W <= 8 && (K == 2 || K == 4)) sh = {K, 8, NGMAX2, 2};
        }
    }
    const long long nitems = (n + sh.ngroups - 1) / sh.ngroups;
    // traceback tile of one resident wave: one word per lane per TbSteps<K> steps (4 bits per cell)
    const int tb_steps = std::max(1, 8 / sh.K);
    const size_t word_bytes = sh.K == 16 ? 8 : 4;
    size_t per_wave_elems = kernel_mode ? ((static_cast<size_t>(max_len) + sh.W + 16) / tb_steps + 2) * 64 : 0;
    // adaptor_align without penalty selects: snapshots + windowed recompute instead of the code stream
    if (kernel_mode == 1 && local && !pensel) {
        kernel_mode = 3;
        const size_t nsnap = (static_cast<size_t>(max_len) + sh.W + 16) / SNAP_P + 2;
        per_wave_elems = static_cast<size_t>(snap_win(R, sh.W) / tb_steps) * 64 +
                         nsnap * (2 * sh.K + 3) * 64 * (sizeof(double) / word_bytes);
    }
    // Far more workgroups than fit at once: each wave then owns only a few work items and the
    // hardware hands out workgroups as CUs free up, which balances the load much better than an
    // exactly resident grid with a static stride (1.74 -> 2.07 TCUPS at 1M x 2kb; flat from 128 to
    // 384 waves per CU).  The per-wave scratch tiles scale with the grid and are capped below.
    int waves_per_cu = 128;
    if (option(OPT_ALIGN_WAVES_PER_CU) > 0) waves_per_cu = option(OPT_ALIGN_WAVES_PER_CU);
    // workgroups of NWAVES wavefronts; every wavefront owns a traceback tile
    long long grid = std::min<long long>((nitems + NWAVES - 1) / NWAVES,
                                         (static_cast<long long>(c.num_cu) * waves_per_cu + NWAVES - 1) / NWAVES);
    if (kernel_mode) {
        const size_t budget = static_cast<size_t>(6) << 30;
        const long long fit = std::max<long long>(1, static_cast<long long>(budget / std::max<size_t>(1, per_wave_elems * word_bytes * NWAVES)));
        grid = std::min(grid, fit);
    }
    void* d_dirs = nullptr;
    if (kernel_mode && !wide) SL_TRY(c.buffer("align.dirs", static_cast<size_t>(grid) * NWAVES * per_wave_elems * word_bytes, &d_dirs));

    a.seq = d_seq; a.nmask = d_nmask; a.qual = d_qual; a.off = d_off; a.n = n;
    a.R = R; a.W = sh.W; a.ngroups = sh.ngroups; a.local = local ? 1 : 0;
    a.qoffset = static_cast<int>(enc_names[0]); a.navail = enc_n;
    a.GO = GO; a.GE = GE;
    a.tables = d_tab; a.tab_doubles = static_cast<int>(rows.size()); a.row_bytes = static_cast<int>(enc_n * sizeof(double));
    a.rowzero = d_rz; a.colbase = d_cb; a.refchars = d_ref;
    a.scores = out.d_scores; a.starts = out.d_starts; a.ends = out.d_ends;
    a.sec_s = d_ss; a.sec_e = d_se; a.nsec = nsec; a.sec_so = out.d_sec_so; a.sec_wo = out.d_sec_wo;
    a.dirs = d_dirs; a.dirs_per_wave = per_wave_elems; a.badqual = d_bad;
    a.read_base = co.read_base; a.sec_stride = co.sec_stride ? co.sec_stride : n;
    a.snap_head = snap_head(R); a.snap_win = snap_win(R, sh.W);
    a.aln_ref = out.d_aln_ref; a.aln_qry = out.d_aln_qry; a.aln_len = out.d_aln_len; a.edits = out.d_edits;

    const size_t lds = sizeof(uint16_t) * NWAVES * (sh.rowf == 2 ? NGMAX2 * (RING + RING_MIRROR) : NGMAX * RING_SLOT) + sizeof(double) * rows.size() +
                       sizeof(int32_t) * NWAVES * sh.ngroups * (R + 1) + 16;
    if (!wide && lds > 64 * 1024) return fail("sarlacc_amd: alignment tables do not fit in LDS");
    SL_HIP(hipEventRecord(c.ev_start, stream));
    if (wide) SL_TRY(launch_wide(a, R, kernel_mode == 3 ? 1 : kernel_mode, max_len, n, stream));
    else SL_TRY(launch_k(sh.K, sh.rowf, R, pensel, kernel_mode, local, a, static_cast<int>(grid), lds, stream));
    SL_HIP(hipEventRecord(c.ev_stop, stream));
    c.timed = true;

    if (!co.finish) return 0;
    int flags[2] = {0, 0};
    SL_HIP(hipMemcpyAsync(flags, d_bad, sizeof flags, hipMemcpyDeviceToHost, stream));
    SL_HIP(hipStreamSynchronize(stream));
    *bad_qual_read = flags[0];
    if (flags[1]) return fail("sarlacc_amd: internal error: an alignment traceback exceeded its bound");
    return 0;
}

// Error that the reference would raise first while looping over the reads
// (length mismatch is tested before each alignment, src/adaptor_align.cpp:51-53;
// inside an alignment column 1 is evaluated first: its reference character, then
// the qualities of every row, then the remaining reference characters).
static int first_error(int64_t n, const int64_t* seq_off, const int64_t* qual_off, const char* ref, int R,
                       int bad_qual_read) {
    int64_t len_bad = -1;
    if (qual_off)
        for (int64_t i = 0; i < n; ++i)
            if (seq_off[i + 1] - seq_off[i] != qual_off[i + 1] - qual_off[i]) { len_bad = i; break; }
    int first_bad_col = -1;
    uint32_t tmp;
    for (int col = 0; col < R; ++col)
        if (column_info(ref[col], &tmp)) { first_bad_col = col; break; }
    int64_t first_nonempty = -1;
    if (first_bad_col >= 0)
        for (int64_t i = 0; i < n; ++i)
            if (seq_off[i + 1] - seq_off[i] > 0) { first_nonempty = i; break; }

    const int64_t INF = std::numeric_limits<int64_t>::max();
    const int64_t e_len = len_bad >= 0 ? len_bad : INF;
    const int64_t e_qual = (R > 0 && bad_qual_read != std::numeric_limits<int>::max()) ? bad_qual_read : INF;
    const int64_t e_ref = first_nonempty >= 0 ? first_nonempty : INF;
    const int64_t first = std::min(e_len, std::min(e_qual, e_ref));
    if (first == INF) return 0;
    if (first == e_len) return fail("sequence and quality strings should have the same length");
    if (first == e_ref && first_bad_col == 0) return fail("unrecognized base in reference sequence");
    if (first == e_qual) return fail("quality cannot be lower than smallest encoded value");
    return fail("unrecognized base in reference sequence");
}

struct HostBatch {
    uint8_t* d_seq = nullptr;
    uint8_t* d_qual = nullptr;
    int64_t* d_off = nullptr;
    int32_t max_len = 0;
    int64_t len_bad = -1;
};

// Uploads a batch given host string sets.  Reads whose quality string has a
// different length make the whole call fail later (first_error), so qualities
// are copied with the sequence offsets only when every length agrees.
static int upload_batch(const char* seq, const int64_t* seq_off, const char* qual, const int64_t* qual_off,
                        int64_t n, HostBatch* hb, hipStream_t s, bool defer_data = false) {
    int64_t mx = 0;
    bool same = true;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t L = seq_off[i + 1] - seq_off[i];
        mx = std::max(mx, L);
        if (L != qual_off[i + 1] - qual_off[i]) { same = false; if (hb->len_bad < 0) hb->len_bad = i; }
    }
    if (mx > std::numeric_limits<int32_t>::max() / 2) return fail("sarlacc_amd: read longer than 2^30 bases");
    hb->max_len = static_cast<int32_t>(mx);
    if (!same) return 0;
    const int64_t base = n ? seq_off[0] : 0;
    const int64_t total = n ? seq_off[n] - base : 0;
    std::vector<int64_t> rel(static_cast<size_t>(n) + 1);
    for (int64_t i = 0; i <= n; ++i) rel[i] = (n ? seq_off[i] : 0) - base;
    if (defer_data) {   // the caller copies the bases and qualities chunk by chunk
        SL_TRY(scratch("batch.seq", static_cast<size_t>(total), &hb->d_seq));
        SL_TRY(scratch("batch.qual", static_cast<size_t>(total), &hb->d_qual));
    } else {
        SL_TRY(upload("batch.seq", reinterpret_cast<const uint8_t*>(seq) + base, static_cast<size_t>(total), &hb->d_seq, s));
        SL_TRY(upload("batch.qual", reinterpret_cast<const uint8_t*>(qual) + (n ? qual_off[0] : 0), static_cast<size_t>(total), &hb->d_qual, s));
    }
    SL_TRY(upload("batch.off", rel.data(), rel.size(), &hb->d_off, s));
    if (defer_data) SL_HIP(hipStreamSynchronize(s));   // `rel` goes out of scope
    return 0;
}

static int check_sections(const int32_t* ss, const int32_t* se, int nsec, int R) {
    for (int x = 0; x < nsec; ++x)
        if (ss[x] < 0 || ss[x] > R || se[x] < 0 || se[x] > R)
            return fail("sarlacc_amd: section bounds outside the adaptor (the reference reads out of range here)");
    return 0;
}

static int host_align(const char* seq, const int64_t* seq_off, const char* qual, const int64_t* qual_off, int64_t n,
                      const double* enc_errors, const char* enc_names, int enc_n, double gapopen, double gapext,
                      const char* ref, int R, bool local, int kernel_mode,
                      const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                      double* scores, int32_t* starts, int32_t* ends, int32_t* sec_so, int32_t* sec_wo,
                      int32_t* edits, char* aln_ref, char* aln_qry, int64_t* aln_off, int64_t aln_cap) {
    SL_TRY(check_encoding(enc_errors, enc_names, enc_n));
    if (n < 0) return fail("sarlacc_amd: negative number of sequences");
    if (kernel_mode == 1) SL_TRY(check_sections(sec_starts, sec_ends, nsec, R));
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    if (n == 0) { if (aln_off) aln_off[0] = 0; return 0; }

    // Large score / map calls go to the device in chunks: the bases and qualities of chunk k+1
    // cross PCIe on a stream of their own while chunk k is aligned.
    const int64_t total_bytes = seq_off[n] - seq_off[0];
    int64_t nchunks = 1;
    if (kernel_mode != 2 && R > 0) {
        if (total_bytes >= (static_cast<int64_t>(512) << 20)) nchunks = std::min<int64_t>(8, total_bytes / (static_cast<int64_t>(256) << 20));
        if (option(OPT_ALIGN_CHUNKS) > 0) nchunks = option(OPT_ALIGN_CHUNKS);   // testing
        nchunks = std::max<int64_t>(1, std::min<int64_t>(nchunks, n));
    }
    HostBatch hb;
    SL_TRY(upload_batch(seq, seq_off, qual, qual_off, n, &hb, s, nchunks > 1));
    if (hb.len_bad >= 0) {
        // reads before the offending one could still raise an earlier error, but
        // only a bad reference character is detectable without the qualities
        return first_error(n, seq_off, qual_off, ref, R, std::numeric_limits<int>::max());
    }
    AlignOut out;
    const size_t nn = static_cast<size_t>(n);
    SL_TRY(scratch("out.scores", nn, &out.d_scores));
    if (kernel_mode == 1) {
        SL_TRY(scratch("out.starts", nn, &out.d_starts));
        SL_TRY(scratch("out.ends", nn, &out.d_ends));
        SL_TRY(scratch("out.sso", nn * std::max(nsec, 1), &out.d_sec_so));
        SL_TRY(scratch("out.swo", nn * std::max(nsec, 1), &out.d_sec_wo));
    }
    const int64_t total = seq_off[n] - seq_off[0];
    const size_t aln_bytes = static_cast<size_t>(total + n * R);
    if (kernel_mode == 2) {
        SL_TRY(scratch("out.aref", aln_bytes, &out.d_aln_ref));
        SL_TRY(scratch("out.aqry", aln_bytes, &out.d_aln_qry));
        SL_TRY(scratch("out.alen", nn, &out.d_aln_len));
        SL_TRY(scratch("out.edits", nn, &out.d_edits));
    }
    int bad = 0;
    if (nchunks == 1) {
        SL_TRY(run_align(hb.d_seq, nullptr, hb.d_qual, hb.d_off, n, hb.max_len, enc_errors, enc_names, enc_n, gapopen, gapext,
                         ref, R, local, kernel_mode, sec_starts, sec_ends, nsec, out, s, &bad));
    } else {
        hipStream_t copy_stream = nullptr;
        SL_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        std::vector<hipEvent_t> ready(static_cast<size_t>(nchunks), nullptr);
        const int64_t sbase = seq_off[0], qbase = qual_off[0];
        auto bound = [&](int64_t k) { return n * k / nchunks; };
        auto send = [&](int64_t k) -> int {   // bases and qualities of the reads of chunk k
            const int64_t lo = seq_off[bound(k)] - sbase, hi = seq_off[bound(k + 1)] - sbase;
            if (hi > lo) {
                SL_HIP(hipMemcpyAsync(hb.d_seq + lo, seq + sbase + lo, static_cast<size_t>(hi - lo), hipMemcpyHostToDevice, copy_stream));
                SL_HIP(hipMemcpyAsync(hb.d_qual + lo, qual + qbase + lo, static_cast<size_t>(hi - lo), hipMemcpyHostToDevice, copy_stream));
            }
            SL_HIP(hipEventCreateWithFlags(&ready[k], hipEventDisableTiming));
            SL_HIP(hipEventRecord(ready[k], copy_stream));
            return 0;
        };
        int rc = send(0);
        for (int64_t k = 0; k < nchunks && !rc; ++k) {
            const int64_t lo = bound(k), hi = bound(k + 1);
            rc = hipStreamWaitEvent(s, ready[k], 0) == hipSuccess ? 0 : fail("HIP error waiting for a chunk upload");
            if (rc) break;
            AlignOut co_out = out;
            co_out.d_scores = out.d_scores + lo;
            if (out.d_starts) { co_out.d_starts = out.d_starts + lo; co_out.d_ends = out.d_ends + lo; }
            if (out.d_sec_so) { co_out.d_sec_so = out.d_sec_so + lo; co_out.d_sec_wo = out.d_sec_wo + lo; }
            ChunkOpts co;
            co.sec_stride = n; co.read_base = static_cast<int>(lo); co.init_bad = (k == 0); co.finish = (k + 1 == nchunks);
            rc = run_align(hb.d_seq, nullptr, hb.d_qual, hb.d_off + lo, hi - lo, hb.max_len, enc_errors, enc_names, enc_n, gapopen,
                           gapext, ref, R, local, kernel_mode, sec_starts, sec_ends, nsec, co_out, s, &bad, co);
            if (!rc && k + 1 < nchunks) rc = send(k + 1);   // travels while chunk k is being aligned
        }
        (void)hipStreamSynchronize(copy_stream);
        (void)hipStreamSynchronize(s);
        for (hipEvent_t e : ready) if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(copy_stream);
        if (rc) return rc;
    }
    SL_TRY(first_error(n, seq_off, nullptr, ref, R, bad));

    SL_HIP(hipMemcpy(scores, out.d_scores, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (kernel_mode == 1) {
        SL_HIP(hipMemcpy(starts, out.d_starts, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(ends, out.d_ends, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (nsec) {
            SL_HIP(hipMemcpy(sec_so, out.d_sec_so, nn * nsec * sizeof(int32_t), hipMemcpyDeviceToHost));
            SL_HIP(hipMemcpy(sec_wo, out.d_sec_wo, nn * nsec * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
    }
    if (kernel_mode == 2) {
        SL_HIP(hipMemcpy(edits, out.d_edits, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (aln_ref) {
            std::vector<uint8_t> hr(aln_bytes), hq(aln_bytes);
            std::vector<int32_t> hl(nn);
            SL_HIP(hipMemcpy(hr.data(), out.d_aln_ref, aln_bytes, hipMemcpyDeviceToHost));
            SL_HIP(hipMemcpy(hq.data(), out.d_aln_qry, aln_bytes, hipMemcpyDeviceToHost));
            SL_HIP(hipMemcpy(hl.data(), out.d_aln_len, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
            int64_t used = 0;
            aln_off[0] = 0;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t base = (seq_off[i] - seq_off[0]) + i * static_cast<int64_t>(R);
                const int32_t m = hl[i];
                if (used + m > aln_cap) return fail("sarlacc_amd: alignment string buffer too small");
                for (int32_t x = 0; x < m; ++x) {  // the device wrote them end-first
                    aln_ref[used + x] = static_cast<char>(hr[base + m - 1 - x]);
                    aln_qry[used + x] = static_cast<char>(hq[base + m - 1 - x]);
                }
                used += m;
                aln_off[i + 1] = used;
            }
        }
    }
    return 0;
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" {

int sarlacc_adaptor_align(const char* seq, const int64_t* seq_off, const char* qual, const int64_t* qual_off,
                          int64_t n, const double* enc_errors, const char* enc_names, int enc_n, double gapopen,
                          double gapext, const char* adaptor, int adaptor_len, const int32_t* sec_starts,
                          const int32_t* sec_ends, int nsec, double* scores, int32_t* starts, int32_t* ends,
                          int32_t* sec_start_out, int32_t* sec_width_out) {
    return host_align(seq, seq_off, qual, qual_off, n, enc_errors, enc_names, enc_n, gapopen, gapext, adaptor,
                      adaptor_len, true, 1, sec_starts, sec_ends, nsec, scores, starts, ends, sec_start_out,
                      sec_width_out, nullptr, nullptr, nullptr, nullptr, 0);
}

int sarlacc_adaptor_align_score_only(const char* seq, const int64_t* seq_off, const char* qual,
                                     const int64_t* qual_off, int64_t n, const double* enc_errors,
                                     const char* enc_names, int enc_n, double gapopen, double gapext,
                                     const char* adaptor, int adaptor_len, double* scores) {
    return host_align(seq, seq_off, qual, qual_off, n, enc_errors, enc_names, enc_n, gapopen, gapext, adaptor,
                      adaptor_len, true, 0, nullptr, nullptr, 0, scores, nullptr, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, 0);
}

int sarlacc_barcode_align(const char* seq, const int64_t* seq_off, const char* qual, const int64_t* qual_off,
                          int64_t n, const double* enc_errors, const char* enc_names, int enc_n, double gapopen,
                          double gapext, const char* reference, int reference_len, double* scores) {
    return host_align(seq, seq_off, qual, qual_off, n, enc_errors, enc_names, enc_n, gapopen, gapext, reference,
                      reference_len, false, 0, nullptr, nullptr, 0, scores, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, nullptr, 0);
}

int sarlacc_general_align(const char* seq, const int64_t* seq_off, const char* qual, const int64_t* qual_off,
                          int64_t n, const double* enc_errors, const char* enc_names, int enc_n, double gapopen,
                          double gapext, const char* reference, int reference_len, int edit_only, double* scores,
                          int32_t* edits, char* aln_ref, char* aln_query, int64_t* aln_off, int64_t aln_cap) {
    return host_align(seq, seq_off, qual, qual_off, n, enc_errors, enc_names, enc_n, gapopen, gapext, reference,
                      reference_len, false, 2, nullptr, nullptr, 0, scores, nullptr, nullptr, nullptr, nullptr, edits,
                      edit_only ? nullptr : aln_ref, edit_only ? nullptr : aln_query, aln_off, aln_cap);
}

static int dev_align_impl(const uint8_t* d_seq, const uint8_t* d_nmask, const uint8_t* d_qual, const int64_t* d_off,
                          int64_t n, int32_t max_len, const double* enc_errors, const char* enc_names, int enc_n,
                          double gapopen, double gapext, const char* reference, int reference_len, int mode,
                          const int32_t* sec_starts, const int32_t* sec_ends, int nsec, double* d_scores,
                          int32_t* d_starts, int32_t* d_ends, int32_t* d_sec_start_out, int32_t* d_sec_width_out,
                          void* stream) {
    SL_TRY(check_encoding(enc_errors, enc_names, enc_n));
    SL_TRY(ensure_device());
    const bool trace = d_starts != nullptr;
    if (trace && mode != 0) return fail("sarlacc_amd: positions are only defined for the local (adaptor) mode");
    if (trace) SL_TRY(check_sections(sec_starts, sec_ends, nsec, reference_len));
    AlignOut out;
    out.d_scores = d_scores;
    out.d_starts = d_starts;
    out.d_ends = d_ends;
    out.d_sec_so = d_sec_start_out;
    out.d_sec_wo = d_sec_width_out;
    int bad = 0;
    SL_TRY(run_align(d_seq, d_nmask, d_qual, d_off, n, max_len, enc_errors, enc_names, enc_n, gapopen, gapext, reference,
                     reference_len, mode == 0, trace ? 1 : 0, sec_starts, sec_ends, trace ? nsec : 0, out,
                     static_cast<hipStream_t>(stream), &bad));
    if (reference_len > 0 && bad != std::numeric_limits<int>::max())
        return fail("quality cannot be lower than smallest encoded value");
    uint32_t tmp;
    for (int col = 0; col < reference_len; ++col)
        if (column_info(reference[col], &tmp) && max_len > 0) return fail("unrecognized base in reference sequence");
    return 0;
}

// 8 bases per thread: 2 bytes of 2-bit codes + 1 byte of exception bits
__global__ void k_pack_reads(const uint8_t* seq, long long total, uint8_t* packed, uint8_t* nmask) {
    const long long o = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) * 8;
    if (o >= total) return;
    uint32_t bits = 0, exc = 0;
    for (int k = 0; k < 8; ++k) {
        const uint8_t c = (o + k < total) ? seq[o + k] : static_cast<uint8_t>('A');
        uint32_t v;
        switch (c) {
            case 'A': v = 0; break;
            case 'C': v = 1; break;
            case 'G': v = 2; break;
            case 'T': v = 3; break;
            default: v = 0; exc |= 1u << k; break;
        }
        bits |= v << (2 * k);
    }
    packed[o / 4] = static_cast<uint8_t>(bits);
    packed[o / 4 + 1] = static_cast<uint8_t>(bits >> 8);  // the buffer has one spare byte
    nmask[o / 8] = static_cast<uint8_t>(exc);
}

int sarlacc_dev_pack_reads(const uint8_t* d_seq, int64_t total, uint8_t* d_packed, uint8_t* d_nmask, void* stream) {
    if (total < 0) return fail("sarlacc_amd: negative size");
    if (total == 0) return 0;
    SL_TRY(ensure_device());
    const long long threads = (total + 7) / 8;
    hipLaunchKernelGGL(k_pack_reads, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_seq, static_cast<long long>(total), d_packed, d_nmask);
    SL_HIP(hipGetLastError());
    return 0;
}

int sarlacc_dev_align(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                      int32_t max_len, const double* enc_errors, const char* enc_names, int enc_n, double gapopen,
                      double gapext, const char* reference, int reference_len, int mode, const int32_t* sec_starts,
                      const int32_t* sec_ends, int nsec, double* d_scores, int32_t* d_starts, int32_t* d_ends,
                      int32_t* d_sec_start_out, int32_t* d_sec_width_out, void* stream) {
    return dev_align_impl(d_seq, nullptr, d_qual, d_off, n, max_len, enc_errors, enc_names, enc_n, gapopen, gapext,
                          reference, reference_len, mode, sec_starts, sec_ends, nsec, d_scores, d_starts, d_ends,
                          d_sec_start_out, d_sec_width_out, stream);
}

int sarlacc_dev_align_packed(const uint8_t* d_packed, const uint8_t* d_nmask, const uint8_t* d_qual,
                             const int64_t* d_off, int64_t n, int32_t max_len, const double* enc_errors,
                             const char* enc_names, int enc_n, double gapopen, double gapext, const char* reference,
                             int reference_len, int mode, const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                             double* d_scores, int32_t* d_starts, int32_t* d_ends, int32_t* d_sec_start_out,
                             int32_t* d_sec_width_out, void* stream) {
    if (!d_nmask) return fail("sarlacc_amd: packed alignment needs the exception mask of sarlacc_dev_pack_reads");
    return dev_align_impl(d_packed, d_nmask, d_qual, d_off, n, max_len, enc_errors, enc_names, enc_n, gapopen, gapext,
                          reference, reference_len, mode, sec_starts, sec_ends, nsec, d_scores, d_starts, d_ends,
                          d_sec_start_out, d_sec_width_out, stream);
}
}
