// msa.hip -- per-group multiple sequence alignment ("MSA spec v1") on gfx950.
//
// Stands in for the reference's quick_msa (/root/reference/src/quick_msa.cpp:15-80),
// which hands each group to SeqAn's T-Coffee.  SeqAn is a third-party dependency
// that is neither in the reference tree nor pinned by any reference test, so the
// arithmetic here follows our own written specification (DESIGN.md "MSA spec v1",
// checker: oracle/msa.c) and keeps only the documented contract of quick_msa:
// integer simple scores, banded global pairwise alignment, one gapped row per read in
// group order, equal widths, '-' for gaps, non-ACGT shown as N, singletons verbatim.
//
// Kernels:
//   k_msa_pairwise_ad  (default) one wavefront per (read, centre) pair: banded Gotoh
//                   scheduled along anti-diagonals of the band (t = 2i + x), no scan,
//                   every lane busy on every step; see the comment above the kernel.
//   k_msa_pairwise  (SARLACC_MSA_SCAN=1, kept for A/B runs) the same recurrences row by
//                   row, lanes own C consecutive band cells; the horizontal gap chain is
//                   a max-plus prefix scan across the wave (DPP row_shr / row_bcast);
//                   4 traceback bits per cell stream to a per-wave HBM tile (one
//                   coalesced dword per lane per row) and are walked back through an
//                   LDS window of 32 rows.
//   k_msa_width     per group: column budget (max insertions before each centre base).
//   k_msa_write     per group: emits the gapped rows.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <limits>
#include <vector>

namespace sarlacc {

constexpr int MSA_NEG = -(1 << 28);
constexpr int TB_ROWS = 32;
template <bool B>
struct Flag2 { static constexpr bool value = B; };

struct MsaJob {
    long long read_off;   // into seq
    long long ctr_off;
    long long out_off;    // into ins[] (lc+1 entries) and aln[] (lc entries, same base)
    int lr, lc;
};

struct MsaArgs {
    const uint8_t* seq;
    const MsaJob* jobs;
    const int* order;       // optional: the njobs job indices this launch works on
    int njobs;
    int ma, mm, go, ge, bw;
    uint16_t* ins;          // per pair: insertions before each centre position
    uint8_t* aln;           // per pair: 1 if the centre base is matched to a read base
    void* tb;               // per-wave traceback tile
    unsigned long long tb_per_wave;  // in tile words
    int* stuck;             // set when a traceback exceeds its step bound (cannot happen with consistent codes;
                            // the bound is what guarantees that every wave leaves the walk)
};

__device__ __forceinline__ uint8_t dna5_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
    }
    return 4;
}

template <int CTRL>
__device__ __forceinline__ int dpp_int(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;

// inclusive max-scan over the 64 lanes
__device__ __forceinline__ int wave_scan_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return v;
}

template <int C>
struct TbWord { using type = uint32_t; };
template <>
struct TbWord<16> { using type = unsigned long long; };

template <int C>
__global__ void __launch_bounds__(64) k_msa_pairwise(const MsaArgs A) {
    using Word = typename TbWord<C>::type;
    extern __shared__ __align__(16) unsigned char smem[];
    Word* s_tb = reinterpret_cast<Word*>(smem);
    uint8_t* s_ctr = reinterpret_cast<uint8_t*>(s_tb + TB_ROWS * 64);

    const int lane = threadIdx.x;
    const int ma = A.ma, mm = A.mm, go = A.go, ge = A.ge;
    const int step = (go <= ge) ? ge : go;      // slope of the horizontal gap chain
    const int fadd = go - step;
    Word* const tile = static_cast<Word*>(A.tb) + static_cast<size_t>(blockIdx.x) * A.tb_per_wave;

    for (int jobn = blockIdx.x; jobn < A.njobs; jobn += gridDim.x) {
        const MsaJob J = A.jobs[A.order ? A.order[jobn] : jobn];
        const int lr = J.lr, lc = J.lc;
        const int dlo = min(0, lc - lr) - A.bw;
        const int dhi = max(0, lc - lr) + A.bw;
        const int B = dhi - dlo + 1;
        const uint8_t* rd = A.seq + J.read_off;
        const uint8_t* ct = A.seq + J.ctr_off;
        for (int p = lane; p < lc; p += 64) s_ctr[p] = dna5_code(ct[p]);
        __syncthreads();
        auto ctr_at = [&](int idx) -> int { return (idx >= 0 && idx < lc) ? s_ctr[idx] : 0xff; };

        int Hp[C], Ep[C], cw[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            Hp[k] = MSA_NEG;
            Ep[k] = MSA_NEG;
            cw[k] = ctr_at(dlo + lane * C + k - 1);  // centre base of column j = dlo + x at row 0
        }

        // read bases: lane l holds base (chunk start + l); the next 64 are already in flight, so
        // no row waits on a global load
        int rchunk = (lane < lr) ? dna5_code(rd[lane]) : 4;
        int rnext = (64 + lane < lr) ? dna5_code(rd[64 + lane]) : 4;
        for (int i = 0; i <= lr; ++i) {
            if (i > 1 && ((i - 1) & 63) == 0) {
                rchunk = rnext;
                rnext = (i - 1 + 64 + lane < lr) ? dna5_code(rd[i - 1 + 64 + lane]) : 4;
            }
            const int rc = (i > 0) ? __builtin_amdgcn_readlane(rchunk, (i - 1) & 63) : 0xfe;
            const int upH_r = dpp_int<DPP_WAVE_SHL1>(MSA_NEG, Hp[0]);
            const int upE_r = dpp_int<DPP_WAVE_SHL1>(MSA_NEG, Ep[0]);
            int hq[C], ev[C], dv[C], lp[C];
            unsigned eo = 0;
            int run = MSA_NEG;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int x = lane * C + k;
                const int j = i + dlo + x;
                const bool valid = x < B && j >= 0 && j <= lc;
                const int uH = (k + 1 < C) ? Hp[k + 1] : (lane < 63 ? upH_r : MSA_NEG);
                const int uE = (k + 1 < C) ? Ep[k + 1] : (lane < 63 ? upE_r : MSA_NEG);
                const int eop = uH + go, eex = uE + ge;
                int e = max(eop, eex);
                if (eop >= eex) eo |= 1u << k;
                int d = Hp[k] + (rc == cw[k] ? ma : mm);
                if (i == 0) { e = MSA_NEG; d = (j == 0) ? 0 : MSA_NEG; }
                if (!valid) { e = MSA_NEG; d = MSA_NEG; }
                e = max(e, MSA_NEG);
                d = max(d, MSA_NEG);
                ev[k] = e;
                dv[k] = d;
                hq[k] = max(d, e);
                run = max(run, hq[k] - step * x);
                lp[k] = run;
            }
            const int incl = wave_scan_max(run);
            const int excl = dpp_int<DPP_WAVE_SHR1>(MSA_NEG, incl);
            int Hn[C], Fn[C];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int x = lane * C + k;
                const int j = i + dlo + x;
                const bool valid = x < B && j >= 0 && j <= lc;
                const int pprev = (k > 0) ? max(excl, lp[k - 1]) : excl;
                int f = step * x + pprev + fadd;
                if (!valid || j < 1 || x < 1) f = MSA_NEG;
                f = max(f, MSA_NEG);
                Fn[k] = f;
                Hn[k] = valid ? max(hq[k], f) : MSA_NEG;
            }
            const int lH = dpp_int<DPP_WAVE_SHR1>(MSA_NEG, Hn[C - 1]);
            const int lF = dpp_int<DPP_WAVE_SHR1>(MSA_NEG, Fn[C - 1]);
            Word bits = 0;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int pH = (k > 0) ? Hn[k - 1] : lH;
                const int pF = (k > 0) ? Fn[k - 1] : lF;
                const unsigned fo = (pH + go >= pF + ge) ? 1u : 0u;
                const int d = dv[k], e = ev[k], f = Fn[k];
                unsigned hd;
                if (d >= e && d >= f) hd = 0;
                else if (e >= f) hd = 1;
                else hd = 2;
                const unsigned t = hd | (((eo >> k) & 1u) << 2) | (fo << 3);
                bits |= static_cast<Word>(t) << (4 * k);
            }
            tile[static_cast<size_t>(i) * 64 + lane] = bits;
#pragma unroll
            for (int k = 0; k < C; ++k) { Hp[k] = Hn[k]; Ep[k] = ev[k]; }
            // slide the centre window by one column
            const int nxt = dpp_int<DPP_WAVE_SHL1>(0xff, cw[0]);
#pragma unroll
            for (int k = 0; k + 1 < C; ++k) cw[k] = cw[k + 1];
            cw[C - 1] = (lane < 63) ? nxt : ctr_at(i + 1 + dlo + 64 * C - 2);
        }

        // ---- traceback through an LDS window (wave-uniform walk) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint16_t* ins = A.ins + J.out_off;
        uint8_t* aln = A.aln + J.out_off;
        int i = lr, j = lc, state = 0, cnt = 0;
        int cb = lr + 1;  // first row held in the window (none yet)
        int walk_budget = 2 * (lr + lc) + 64;   // every iteration consumes a row, a column or changes state once
        while ((i > 0 || j > 0) && --walk_budget >= 0) {
            if (i < cb) {
                cb = max(0, i - (TB_ROWS - 1));
                __syncthreads();
                for (int r = 0; r < TB_ROWS; ++r)
                    if (cb + r <= lr) s_tb[r * 64 + lane] = tile[static_cast<size_t>(cb + r) * 64 + lane];
                __syncthreads();
            }
            const int x = j - i - dlo;
            if (state == 0) {
                // A run of diagonal moves stays on one band diagonal x: the rows of the window
                // are inspected 64 at a time (lane l looks at row i-l), one step per run instead
                // of one per base -- ~94 % of the moves of same-molecule reads are diagonal.
                const int reach = min(min(i, j), i - cb + 1);       // cells (i-l, j-l), l < reach
                unsigned tl = 1;                                    // out of reach counts as "not diagonal"
                if (lane < reach)
                    tl = static_cast<unsigned>(s_tb[(i - lane - cb) * 64 + x / C] >> (4 * (x % C))) & 3u;
                const unsigned long long nd = __ballot(tl != 0);
                const int run = nd ? static_cast<int>(__builtin_ctzll(nd)) : 64;
                if (run > 0) {
                    if (lane < run) { ins[j - lane] = (lane == 0) ? static_cast<uint16_t>(cnt) : static_cast<uint16_t>(0); aln[j - lane - 1] = 1; }
                    cnt = 0; i -= run; j -= run;
                    continue;
                }
                if (reach <= 0) {
                    // i == 0 or j == 0: only gap moves remain (the window always holds row i here)
                }
            }
            const unsigned t = static_cast<unsigned>(s_tb[(i - cb) * 64 + x / C] >> (4 * (x % C))) & 15u;
            if (state == 0) {
                state = t & 3;
                continue;
            }
            if (state == 1) {               // read base inserted before centre position j
                ++cnt;
                state = (t & 4) ? 0 : 1;
                --i;
            } else {                        // centre base j-1 opposite a gap
                if (lane == 0) { ins[j] = static_cast<uint16_t>(cnt); aln[j - 1] = 0; }
                cnt = 0;
                state = (t & 8) ? 0 : 2;
                --j;
            }
        }
        if (lane == 0) ins[0] = static_cast<uint16_t>(cnt);
        if (walk_budget < 0 && lane == 0) atomicExch(A.stuck, 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_msa_pairwise_ad: the same banded Gotoh, scheduled along anti-diagonals of the BAND.
//
// Band coordinates: cell (i, x), x = j - i - dlo in [0, B).  Its inputs are
//   diagonal (i-1, x)      vertical (i-1, x+1)      horizontal (i, x-1)
// so with the time step  t = 2 i + x  every input was produced at step t-1 or t-2: no prefix
// scan over the row is needed (the row-by-row kernel above spends most of its instructions
// on that scan and on per-cell validity tests).  Lane l owns the C consecutive diagonals
// x = C l + k; at step t it updates its cells with k = t (mod 2) -- C/2 cells per lane per step,
// every lane busy on every step -- reading the neighbouring diagonals k-1 / k+1 from its own
// registers, or from the adjacent lane with one DPP shift (H and F from the left on even steps,
// H and E from the right on odd steps).  Two steps ("sub-block") advance every diagonal by one
// row.  Read and centre codes sit in LDS; a lane needs C/2 + 1 consecutive read bases and
// C/2 + 2 consecutive centre bases per sub-block.
//
// Traceback: 4 bits per cell as before; a sub-block yields C nibbles per lane, packed into
// words of 8 (C = 4: two sub-blocks per dword) or C nibbles and stored coalesced.  The walk
// descends in t, so it runs through an LDS window of the last MSA_WIN word rows; runs of
// diagonal moves stay on one band diagonal and are consumed up to 64 rows per step.
constexpr int MSA_WIN = 16;  // word rows of traceback codes held in LDS during the walk

// pk = 2 * pk + (this lane's bit of the SGPR mask m)
__device__ __forceinline__ uint32_t msa_push_bit(uint32_t pk, unsigned long long m) {
    uint32_t r;
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(pk), "s"(m));
    return r;
}

template <int C>
struct AdWord { using type = uint32_t; static constexpr int SPW = 8 / C; };
template <>
struct AdWord<16> { using type = unsigned long long; static constexpr int SPW = 1; };

template <int C>
__global__ void __launch_bounds__(64) k_msa_pairwise_ad(const MsaArgs A) {
    using Word = typename AdWord<C>::type;
    constexpr int SPW = AdWord<C>::SPW;   // sub-blocks per stored word
    constexpr int H2 = C / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    Word* const s_tb = reinterpret_cast<Word*>(smem);
    uint8_t* const s_ct = reinterpret_cast<uint8_t*>(s_tb + MSA_WIN * 64);   // centre codes, 2 bytes of padding in front
    const int lane = threadIdx.x;
    const int ma = A.ma, mm = A.mm, go = A.go, ge = A.ge;
    Word* const tile = static_cast<Word*>(A.tb) + static_cast<size_t>(blockIdx.x) * A.tb_per_wave;

    for (int jobn = blockIdx.x; jobn < A.njobs; jobn += gridDim.x) {
        const MsaJob J = A.jobs[A.order ? A.order[jobn] : jobn];
        const int lr = J.lr, lc = J.lc;
        const int dlo = min(0, lc - lr) - A.bw;
        const int dhi = max(0, lc - lr) + A.bw;
        const int B = dhi - dlo + 1;
        const uint8_t* rd = A.seq + J.read_off;
        const uint8_t* ct = A.seq + J.ctr_off;
        uint8_t* const s_rd = s_ct + ((lc + 8 + 3) & ~3);   // read codes after the centre codes
        __syncthreads();
        for (int p = lane; p < lc; p += 64) s_ct[4 + p] = dna5_code(ct[p]);
        for (int p = lane; p < lr; p += 64) s_rd[4 + p] = dna5_code(rd[p]);
        if (lane < 4) { s_ct[lane] = 0xf0; s_rd[lane] = 0xf1; s_ct[4 + lc + lane] = 0xf0; s_rd[4 + lr + lane] = 0xf1; }
        __syncthreads();

        int Hc[C], Ec[C], Fc[C];
        bool kvalid[C];
        // Band edge: the vertical input of diagonal B - 1 lies outside the band.  Instead of
        // masking the cells beyond the band after every update, the gap penalties a cell adds to
        // its vertical input are per (lane, k) values that sink the candidate below MSA_NEG.
        int gou[C], geu[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            Hc[k] = MSA_NEG; Ec[k] = MSA_NEG; Fc[k] = MSA_NEG;
            kvalid[k] = lane * C + k < B;
            gou[k] = (lane * C + k + 1 < B) ? go : MSA_NEG;
            geu[k] = (lane * C + k + 1 < B) ? ge : MSA_NEG;
        }

        // One sub-block: steps t0 (even k) and t0 + 1 (odd k), t0 even.  Row and column of cell k:
        //   i = t0/2 - (C/2) l - (k >> 1),   j = i + dlo + C l + k
        // Traceback nibble of a cell (raw outcomes, pushed most significant first):
        //   bit 3 F opened (H_left + go >= F_left + ge)   bit 2 E opened   bit 1 e >= f   bit 0 d >= max(e, f)
        // Cells are pushed in the order (parity, h); 8 cells fill one 32-bit chunk `pk`.
        // Values arriving from the neighbouring lanes.  They are also the DPP destinations: lane 0
        // (63) has no source lane for the shift, keeps what the register held, and that is the
        // MSA_NEG it was initialised with -- no re-initialisation per step.
        int xlH = MSA_NEG, xlF = MSA_NEG, xrH = MSA_NEG, xrE = MSA_NEG;
        auto subblock = [&](auto guard_tag, int t0, uint32_t& pk, uint32_t& pk_hi) {
            constexpr bool GUARD = decltype(guard_tag)::value;
            const int ib = (t0 >> 1) - H2 * lane;            // row of cells k = 0, 1
            const int jb = ib + dlo + C * lane;              // column of cell k = 0
            // codes: read bases of rows ib - H2 + 1 .. ib  (s_rd[4 + i - 1]), centre bases of
            // columns jb .. jb + H2 (s_ct[4 + j - 1]); out-of-range indices are clamped, the
            // cells that would use them are invalid and discarded
            int rc[H2], cc[H2 + 1];
#pragma unroll
            for (int h = 0; h < H2; ++h) {
                int idx = ib - h - 1;
                if (GUARD) idx = min(max(idx, -4), lr + 3);
                rc[h] = s_rd[4 + idx];
            }
#pragma unroll
            for (int h = 0; h <= H2; ++h) {
                int idx = jb + h - 1;
                if (GUARD) idx = min(max(idx, -4), lc + 3);
                cc[h] = s_ct[4 + idx];
            }
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                // neighbours across the lane boundary (values of the previous step)
                if (par == 0) { xlH = dpp_int<DPP_WAVE_SHR1>(xlH, Hc[C - 1]); xlF = dpp_int<DPP_WAVE_SHR1>(xlF, Fc[C - 1]); }
                else { xrH = dpp_int<DPP_WAVE_SHL1>(xrH, Hc[0]); xrE = dpp_int<DPP_WAVE_SHL1>(xrE, Ec[0]); }
                int nH[H2], nE[H2], nF[H2];
#pragma unroll
                for (int h = 0; h < H2; ++h) {
                    const int k = 2 * h + par;
                    const int uH = (k + 1 < C) ? Hc[k + 1] : xrH, uE = (k + 1 < C) ? Ec[k + 1] : xrE;
                    const int lH = (k > 0) ? Hc[k - 1] : xlH, lF = (k > 0) ? Fc[k - 1] : xlF;
                    const int eop = uH + gou[k], eex = uE + geu[k];
                    // Interior cells are reachable inside the band, so their scores are finite; the
                    // "minus infinity" inputs from outside the band are re-derived from constants every
                    // step and cannot drift, hence no clamping in the unguarded path.
                    int e = max(eop, eex);
                    const bool eo = eop >= eex;
                    // cell k: row ib - h, column jb + h + par
                    int d = Hc[k] + (rc[h] == cc[h + par] ? ma : mm);
                    const int fop = lH + go, fex = lF + ge;
                    int f = max(fop, fex);
                    if (GUARD) { e = max(e, MSA_NEG); d = max(d, MSA_NEG); f = max(f, MSA_NEG); }
                    const bool fo = fop >= fex;
                    bool valid = true;
                    if (GUARD) {
                        const int i = ib - h, j = jb + h + par;
                        valid = kvalid[k] && i >= 0 && i <= lr && j >= 0 && j <= lc;
                        if (i == 0) { e = MSA_NEG; d = (j == 0) ? 0 : MSA_NEG; }
                        if (j < 1) f = MSA_NEG;
                        if (!valid) { e = MSA_NEG; d = MSA_NEG; f = MSA_NEG; }
                    }
                    const int m = max(e, f);
                    const int hv = max(d, m);
                    const unsigned long long m_fo = __builtin_amdgcn_ballot_w64(fo), m_eo = __builtin_amdgcn_ballot_w64(eo);
                    const unsigned long long m_ef = __builtin_amdgcn_ballot_w64(e >= f), m_dm = __builtin_amdgcn_ballot_w64(d >= m);
                    pk = msa_push_bit(msa_push_bit(msa_push_bit(msa_push_bit(pk, m_fo), m_eo), m_ef), m_dm);
                    if (C == 16 && par == 0 && h == H2 - 1) { pk_hi = pk; pk = 0; }   // 8 cells done: first chunk of a 64-bit word
                    nH[h] = (GUARD && !valid) ? MSA_NEG : hv;
                    nE[h] = e;
                    nF[h] = f;
                }
#pragma unroll
                for (int h = 0; h < H2; ++h) { Hc[2 * h + par] = nH[h]; Ec[2 * h + par] = nE[h]; Fc[2 * h + par] = nF[h]; }
            }
        };

        // steps 0 .. 2 lr + B - 1, in word blocks of 2 SPW steps
        const int tsteps = 2 * lr + B;
        const int nwords = (tsteps + 2 * SPW - 1) / (2 * SPW);
        // sub-blocks [sb_lo, sb_hi) have every cell with x < B inside the matrix (i >= 1, 1 <= j <= lc, i <= lr)
        int sb_lo = (max(B + 1, 2 - 2 * dlo) + 1) / 2 + 1;
        int sb_hi = min(2 * lr, 2 * (lc - dlo) - B) / 2 - 1;
        int w_lo = (sb_lo + SPW - 1) / SPW, w_hi = sb_hi / SPW;
        w_lo = min(w_lo, nwords);
        w_hi = min(max(w_hi, w_lo), nwords);
        auto words = [&](auto guard_tag, int wb, int we) {
            for (int w = wb; w < we; ++w) {
                uint32_t pk = 0, pk_hi = 0;
#pragma unroll
                for (int sbk = 0; sbk < SPW; ++sbk) subblock(guard_tag, 2 * (w * SPW + sbk), pk, pk_hi);
                Word out;
                if (C == 16) out = static_cast<Word>((static_cast<unsigned long long>(pk_hi) << 32) | pk);
                else out = static_cast<Word>(pk);
                tile[static_cast<size_t>(w) * 64 + lane] = out;
            }
        };
        words(Flag2<true>{}, 0, w_lo);
        words(Flag2<false>{}, w_lo, w_hi);
        words(Flag2<true>{}, w_hi, nwords);

        // ---- traceback through an LDS window (wave-uniform walk) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        uint16_t* ins = A.ins + J.out_off;
        uint8_t* aln = A.aln + J.out_off;
        int i = lr, j = lc, state = 0, cnt = 0;
        // The LDS window holds MSA_WL lanes x MSA_WR word rows of codes around the path (the path
        // stays near one band diagonal, so a narrow strip covers many more rows than full rows would).
        constexpr int MSA_WL = 8, MSA_WR = MSA_WIN * 64 / MSA_WL;
        int wlo = nwords, llo = 0;   // first word row / first lane held in the window (none yet)
        // code of cell (ii, xx): sub-block ws = ii + (xx + C l ... ) -- t = 2 ii + xx, ws = t >> 1
        auto locate = [&](int ii, int xx, int& wrow, int& shift, int& ln) {
            const int t = 2 * ii + xx;
            const int ws = t >> 1;
            const int k = xx % C;
            ln = xx / C;
            wrow = ws / SPW;
            // cells are pushed in the order (sub-block, parity, h), the first one ends up on top
            const int q = (ws % SPW) * C + (k & 1) * H2 + (k >> 1);
            shift = 4 * (SPW * C - 1 - q);
        };
        int walk_budget = 2 * (lr + lc) + 64;   // every iteration consumes a row, a column or changes state once
        while ((i > 0 || j > 0) && --walk_budget >= 0) {
            const int x = j - i - dlo;
            int wrow, shift, ln;
            locate(i, x, wrow, shift, ln);
            if (wrow < wlo || ln < llo || ln >= llo + MSA_WL) {
                wlo = max(0, wrow - (MSA_WR - 1));
                llo = min(max(ln - MSA_WL / 2, 0), 64 - MSA_WL);
                __syncthreads();
                for (int idx = lane; idx < MSA_WR * MSA_WL; idx += 64) {
                    const int r = idx / MSA_WL, cl = idx % MSA_WL;
                    if (wlo + r < nwords) s_tb[idx] = tile[static_cast<size_t>(wlo + r) * 64 + llo + cl];
                }
                __syncthreads();
            }
            if (state == 0) {
                // run of diagonal moves: cell (i - m, j - m) keeps x; lane m inspects it
                const int reach = min(min(i, j), ((2 * i + x) >> 1) - wlo * SPW + 1);
                unsigned tl = 1;
                if (lane < reach) {
                    int wr, sh, l2;
                    locate(i - lane, x, wr, sh, l2);
                    tl = (static_cast<unsigned>(s_tb[(wr - wlo) * MSA_WL + (l2 - llo)] >> sh) & 1u) ^ 1u;   // bit 0: diagonal
                }
                const unsigned long long nd = __ballot(tl != 0);
                const int run = nd ? static_cast<int>(__builtin_ctzll(nd)) : 64;
                if (run > 0) {
                    if (lane < run) { ins[j - lane] = (lane == 0) ? static_cast<uint16_t>(cnt) : static_cast<uint16_t>(0); aln[j - lane - 1] = 1; }
                    cnt = 0; i -= run; j -= run;
                    continue;
                }
            }
            const unsigned t = static_cast<unsigned>(s_tb[(wrow - wlo) * MSA_WL + (ln - llo)] >> shift) & 15u;
            if (state == 0) {
                state = (t & 1u) ? 0 : ((t & 2u) ? 1 : 2);   // diagonal, else vertical if e >= f, else horizontal
                continue;
            }
            if (state == 1) {               // read base inserted before centre position j
                ++cnt;
                state = (t & 4) ? 0 : 1;
                --i;
            } else {                        // centre base j-1 opposite a gap
                if (lane == 0) { ins[j] = static_cast<uint16_t>(cnt); aln[j - 1] = 0; }
                cnt = 0;
                state = (t & 8) ? 0 : 2;
                --j;
            }
        }
        if (lane == 0) ins[0] = static_cast<uint16_t>(cnt);
        if (walk_budget < 0 && lane == 0) atomicExch(A.stuck, 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
struct MsaGroup {
    long long first_job;   // jobs of this group are first_job .. first_job + nreads - 2 (centre has none)
    long long read0;       // index of the group's first entry in the flattened member list
    int nreads;
    int centre;            // position of the centre inside the group
    int lc;
};

struct MergeArgs {
    const uint8_t* seq;
    const int64_t* seq_off;
    const int32_t* members;   // flattened 1-based read ids
    const MsaGroup* groups;
    const MsaJob* jobs;
    long long ngroups;
    const uint16_t* ins;
    const uint8_t* aln;
    uint16_t* maxins;          // per group: lc+1 entries at mi_off[g]
    const long long* mi_off;
    int32_t* width;            // per group
    const long long* out_off;  // per group start in out
    uint8_t* out;
};

// job index of read position r (inside its group), -1 for the centre
__device__ __forceinline__ long long job_of(const MsaGroup& G, int r) {
    if (r == G.centre) return -1;
    return G.first_job + (r < G.centre ? r : r - 1);
}

__global__ void k_msa_width(MergeArgs A) {
    const long long g = blockIdx.x;
    const MsaGroup G = A.groups[g];
    __shared__ long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    if (G.nreads < 2) {
        if (threadIdx.x == 0) {
            int w = 0;
            if (G.nreads == 1) { const long long id = A.members[G.read0] - 1; w = static_cast<int>(A.seq_off[id + 1] - A.seq_off[id]); }
            A.width[g] = w;
        }
        return;
    }
    long long local = 0;
    for (int p = threadIdx.x; p <= G.lc; p += blockDim.x) {
        int m = 0;
        for (int r = 0; r < G.nreads; ++r) {
            const long long jb = job_of(G, r);
            if (jb >= 0) m = max(m, static_cast<int>(A.ins[A.jobs[jb].out_off + p]));
        }
        A.maxins[A.mi_off[g] + p] = static_cast<uint16_t>(m);
        local += m;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(&s_sum), static_cast<unsigned long long>(local));
    __syncthreads();
    if (threadIdx.x == 0) A.width[g] = static_cast<int32_t>(G.lc + s_sum);
}

// One block per (group, read): columns come from prefix sums over the centre positions.
__global__ void k_msa_write(MergeArgs A, const long long* row_group, const int* row_pos, long long nrows) {
    const long long row = blockIdx.x;
    if (row >= nrows) return;
    const long long g = row_group[row];
    const int r = row_pos[row];
    const MsaGroup G = A.groups[g];
    const long long id = A.members[G.read0 + r] - 1;
    const uint8_t* src = A.seq + A.seq_off[id];
    const int W = A.width[g];
    uint8_t* dst = A.out + A.out_off[g] + static_cast<long long>(r) * W;
    if (G.nreads == 1) {  // verbatim (src/quick_msa.cpp:46-50)
        for (int p = threadIdx.x; p < W; p += blockDim.x) dst[p] = src[p];
        return;
    }
    const long long jb = job_of(G, r);
    const uint16_t* ins = jb >= 0 ? A.ins + A.jobs[jb].out_off : nullptr;
    const uint8_t* aln = jb >= 0 ? A.aln + A.jobs[jb].out_off : nullptr;
    const uint16_t* mi = A.maxins + A.mi_off[g];
    // serial chunked prefix: each thread owns a contiguous slice of centre positions
    const int lc = G.lc;
    const int per = (lc + 1 + blockDim.x - 1) / blockDim.x;
    const int p0 = min(static_cast<int>(threadIdx.x) * per, lc + 1), p1 = min(p0 + per, lc + 1);
    __shared__ long long s_col[1024], s_rp[1024];
    long long ccol = 0, crp = 0;
    for (int p = p0; p < p1; ++p) {
        ccol += mi[p] + (p < lc ? 1 : 0);
        crp += (ins ? ins[p] : 0) + ((p < lc) ? (aln ? aln[p] : 1) : 0);
    }
    s_col[threadIdx.x] = ccol;
    s_rp[threadIdx.x] = crp;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long a = 0, b = 0;
        for (unsigned t = 0; t < blockDim.x; ++t) {
            const long long x = s_col[t], y = s_rp[t];
            s_col[t] = a; s_rp[t] = b;
            a += x; b += y;
        }
    }
    __syncthreads();
    long long col = s_col[threadIdx.x], rp = s_rp[threadIdx.x];
    for (int p = p0; p < p1; ++p) {
        const int k = ins ? ins[p] : 0;
        const int m = mi[p];
        for (int x = 0; x < k; ++x) dst[col++] = "ACGTN"[dna5_code(src[rp++])];
        for (int x = k; x < m; ++x) dst[col++] = '-';
        if (p < lc) {
            const bool matched = aln ? aln[p] != 0 : true;
            dst[col++] = matched ? "ACGTN"[dna5_code(src[rp++])] : '-';
        }
    }
}

// ---------------------------------------------------------------------------
template <int C>
static int launch_pairwise(const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(k_msa_pairwise<C>, dim3(grid), dim3(64), lds, s, a);
    SL_HIP(hipGetLastError());
    return 0;
}

template <int C>
static int launch_pairwise_ad(const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    // long reads stage more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
    if (lds > 48 * 1024)
        SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msa_pairwise_ad<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
    hipLaunchKernelGGL(k_msa_pairwise_ad<C>, dim3(grid), dim3(64), lds, s, a);
    SL_HIP(hipGetLastError());
    return 0;
}

// The whole MSA stage with the gapped rows left in HBM (res->d_out): shared by sarlacc_quick_msa,
// which copies them back, and sarlacc_msa_consensus, which votes on them where they are.
// out_cap: < 0 no limit; otherwise the rows are only written when they fit (sizing protocol of
// sarlacc_quick_msa: widths and offsets are always filled in).
int msa_run(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq, const int64_t* seq_off,
            int64_t nseq, double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
            bool want_rows, int64_t out_cap, MsaResult* res, const std::function<int()>* overlap,
            const uint8_t* d_seq_resident) {
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
    const bool timing = std::getenv("SARLACC_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tm0 = now();
    // ---- host-side job list: centre = lower median by (length, position) ----
    std::vector<MsaGroup> groups(static_cast<size_t>(ngroups));
    std::vector<MsaJob> jobs;
    std::vector<long long> mi_off(static_cast<size_t>(ngroups) + 1, 0), row_group;
    std::vector<int> row_pos;
    long long pair_out = 0;
    int max_lr = 0, max_lc = 0, max_band = 1;
    std::vector<std::pair<int64_t, int>> order;
    for (int64_t g = 0; g < ngroups; ++g) {
        const int32_t* mem = grp + grp_off[g];
        const int m = static_cast<int>(grp_off[g + 1] - grp_off[g]);
        MsaGroup G{};
        G.first_job = static_cast<long long>(jobs.size());
        G.read0 = grp_off[g] - grp_off[0];
        G.nreads = m;
        G.centre = 0;
        G.lc = 0;
        auto len_of = [&](int r) { return seq_off[mem[r]] - seq_off[mem[r] - 1]; };
        if (m >= 2) {
            order.clear();
            for (int r = 0; r < m; ++r) order.emplace_back(len_of(r), r);
            std::sort(order.begin(), order.end());
            G.centre = order[(m - 1) / 2].second;
            const int64_t lc = len_of(G.centre);
            if (lc > 60000) return fail("sarlacc_amd: reads longer than 60000 bases are not supported by the MSA stage");
            G.lc = static_cast<int>(lc);
            for (int r = 0; r < m; ++r) {
                if (r == G.centre) continue;
                const int64_t lr = len_of(r);
                if (lr > 60000) return fail("sarlacc_amd: reads longer than 60000 bases are not supported by the MSA stage");
                MsaJob J{};
                J.read_off = seq_off[mem[r] - 1] - seq_off[0];
                J.ctr_off = seq_off[mem[G.centre] - 1] - seq_off[0];
                J.out_off = pair_out;
                J.lr = static_cast<int>(lr);
                J.lc = G.lc;
                pair_out += G.lc + 1;
                jobs.push_back(J);
                max_lr = std::max(max_lr, J.lr);
                const long long band = std::llabs(static_cast<long long>(G.lc) - J.lr) + 2LL * bandwidth + 1;
                if (band > 1024) return fail("sarlacc_amd: alignment band of %lld diagonals exceeds 1024 (length difference + 2*bandwidth + 1)", band);
                max_band = std::max(max_band, static_cast<int>(band));
            }
            max_lc = std::max(max_lc, G.lc);
        }
        mi_off[g + 1] = mi_off[g] + (m >= 2 ? G.lc + 1 : 0);
        for (int r = 0; r < m; ++r) { row_group.push_back(g); row_pos.push_back(r); }
        groups[g] = G;
    }
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    Context& c = ctx();
    const double tm1 = now();
    const int64_t total = nseq ? seq_off[nseq] - seq_off[0] : 0;
    std::vector<int64_t> rel(static_cast<size_t>(nseq) + 1);
    for (int64_t i = 0; i <= nseq; ++i) rel[i] = (nseq ? seq_off[i] : 0) - (nseq ? seq_off[0] : 0);

    uint8_t* d_seq; int64_t* d_soff; int32_t* d_mem; MsaGroup* d_groups; MsaJob* d_jobs; long long* d_mioff;
    if (d_seq_resident) d_seq = const_cast<uint8_t*>(d_seq_resident);
    else SL_TRY(upload("msa.seq", reinterpret_cast<const uint8_t*>(seq) + (nseq ? seq_off[0] : 0), static_cast<size_t>(total), &d_seq, s));
    SL_TRY(upload("msa.soff", rel.data(), rel.size(), &d_soff, s));
    SL_TRY(upload("msa.mem", grp + grp_off[0], static_cast<size_t>(nmemb), &d_mem, s));
    SL_TRY(upload("msa.groups", groups.data(), groups.size(), &d_groups, s));
    SL_TRY(upload("msa.jobs", jobs.data(), jobs.size(), &d_jobs, s));
    SL_TRY(upload("msa.mioff", mi_off.data(), mi_off.size(), &d_mioff, s));

    uint16_t* d_ins; uint8_t* d_aln; uint16_t* d_maxins; int32_t* d_width;
    SL_TRY(scratch("msa.ins", static_cast<size_t>(pair_out) + 1, &d_ins));
    SL_TRY(scratch("msa.aln", static_cast<size_t>(pair_out) + 1, &d_aln));
    SL_TRY(scratch("msa.maxins", static_cast<size_t>(mi_off[ngroups]) + 1, &d_maxins));
    SL_TRY(scratch("msa.width", static_cast<size_t>(ngroups), &d_width));

    if (!jobs.empty()) {
        const bool scan_kernel = std::getenv("SARLACC_MSA_SCAN") != nullptr;   // the row-by-row kernel, kept for A/B runs
        // jobs by band class: 4, 8 or 16 diagonals per lane (bands up to 256 / 512 / 1024); one launch per
        // class, so the common narrow bands are not dragged to the widest job's shape
        std::vector<int> order[3];
        int cls_lr[3] = {0, 0, 0}, cls_lc[3] = {0, 0, 0}, cls_band[3] = {1, 1, 1};
        for (size_t q = 0; q < jobs.size(); ++q) {
            const int band = std::abs(jobs[q].lc - jobs[q].lr) + 2 * bandwidth + 1;
            const int cls = scan_kernel ? (max_band <= 256 ? 0 : (max_band <= 512 ? 1 : 2)) : (band <= 256 ? 0 : (band <= 512 ? 1 : 2));
            order[cls].push_back(static_cast<int>(q));
            cls_lr[cls] = std::max(cls_lr[cls], jobs[q].lr);
            cls_lc[cls] = std::max(cls_lc[cls], jobs[q].lc);
            cls_band[cls] = std::max(cls_band[cls], band);
        }
        MsaArgs a{};
        a.seq = d_seq; a.jobs = d_jobs;
        a.ma = static_cast<int>(match); a.mm = static_cast<int>(mismatch);
        // SeqAn's Score(match, mismatch, gap_extend, gap_open): gap of length k = open + (k-1)*extend
        a.go = static_cast<int>(gap_opening); a.ge = static_cast<int>(gap_extension);
        a.bw = bandwidth; a.ins = d_ins; a.aln = d_aln;
        int* d_stuck;
        SL_TRY(scratch("msa.stuck", 1, &d_stuck));
        SL_HIP(hipMemsetAsync(d_stuck, 0, sizeof(int), s));
        a.stuck = d_stuck;
        double cells = 0;
        for (const MsaJob& J : jobs) cells += static_cast<double>(J.lr) * (std::abs(J.lc - J.lr) + 2 * bandwidth + 1);
        c.counts["msa_pairs"] = static_cast<double>(jobs.size());
        c.counts["msa_cells"] = cells;
        SL_HIP(hipEventRecord(c.ev_start, s));
        SL_TRY(c.stage_begin("msa_pairwise", s));
        for (int cls = 0; cls < 3; ++cls) {
            if (order[cls].empty()) continue;
            const int C = 4 << cls;
            const size_t word = C == 16 ? 8 : 4;
            // traceback tile of one resident wave: row-by-row kernel one word row per read row; anti-diagonal
            // kernel one word row per 2 * SPW steps of the 2 lr + B steps
            const size_t spw = C == 4 ? 2 : 1;
            const size_t per_wave = scan_kernel ? (static_cast<size_t>(cls_lr[cls]) + 2) * 64
                                                : ((2 * static_cast<size_t>(cls_lr[cls]) + cls_band[cls]) / (2 * spw) + 2) * 64;
            const size_t lds = scan_kernel ? TB_ROWS * 64 * word + static_cast<size_t>(cls_lc[cls]) + 64
                                           : MSA_WIN * 64 * word + static_cast<size_t>(cls_lc[cls]) + static_cast<size_t>(cls_lr[cls]) + 32;
            if (lds > 160 * 1024) return fail("sarlacc_amd: reads of %d bases do not fit the MSA kernel's LDS staging", std::max(cls_lr[cls], cls_lc[cls]));
            const long long by_lds = std::max<long long>(1, static_cast<long long>((160 * 1024) / lds));
            long long grid = std::min<long long>(static_cast<long long>(order[cls].size()),
                                                 static_cast<long long>(c.num_cu) * (scan_kernel ? std::min<long long>(C == 4 ? 16 : 8, by_lds) : 128));
            // many more single-wave workgroups than fit at once (a wave then aligns only a few pairs and the
            // hardware balances the load); their traceback tiles are the price, capped at 24 GB of HBM
            const size_t budget = static_cast<size_t>(24) << 30;
            grid = std::min<long long>(grid, std::max<long long>(1, static_cast<long long>(budget / (per_wave * word))));
            void* d_tb; int* d_order;
            const char* tb_name[3] = {"msa.tb0", "msa.tb1", "msa.tb2"};
            const char* ord_name[3] = {"msa.ord0", "msa.ord1", "msa.ord2"};
            SL_TRY(c.buffer(tb_name[cls], static_cast<size_t>(grid) * per_wave * word, &d_tb));
            SL_TRY(upload(ord_name[cls], order[cls].data(), order[cls].size(), &d_order, s));
            a.order = d_order; a.njobs = static_cast<int>(order[cls].size());
            a.tb = d_tb; a.tb_per_wave = per_wave;
            if (scan_kernel) {
                if (C == 4) SL_TRY(launch_pairwise<4>(a, static_cast<int>(grid), lds, s));
                else if (C == 8) SL_TRY(launch_pairwise<8>(a, static_cast<int>(grid), lds, s));
                else SL_TRY(launch_pairwise<16>(a, static_cast<int>(grid), lds, s));
            } else {
                if (C == 4) SL_TRY(launch_pairwise_ad<4>(a, static_cast<int>(grid), lds, s));
                else if (C == 8) SL_TRY(launch_pairwise_ad<8>(a, static_cast<int>(grid), lds, s));
                else SL_TRY(launch_pairwise_ad<16>(a, static_cast<int>(grid), lds, s));
            }
        }
        SL_HIP(hipEventRecord(c.ev_stop, s));
        SL_TRY(c.stage_end("msa_pairwise", s));
        c.timed = true;
    }
    const double tm2 = now();
    if (overlap) SL_TRY((*overlap)());
    const double tm3 = now();
    MergeArgs m{};
    m.seq = d_seq; m.seq_off = d_soff; m.members = d_mem; m.groups = d_groups; m.jobs = d_jobs; m.ngroups = ngroups;
    m.ins = d_ins; m.aln = d_aln; m.maxins = d_maxins; m.mi_off = d_mioff; m.width = d_width;
    SL_TRY(c.stage_begin("msa_merge", s));
    hipLaunchKernelGGL(k_msa_width, dim3(static_cast<unsigned>(ngroups)), dim3(256), 0, s, m);
    SL_HIP(hipGetLastError());
    std::vector<int32_t> width(static_cast<size_t>(ngroups));
    SL_HIP(hipMemcpy(width.data(), d_width, sizeof(int32_t) * width.size(), hipMemcpyDeviceToHost));
    if (!jobs.empty()) {
        int* d_stuck; int stuck = 0;
        SL_TRY(scratch("msa.stuck", 1, &d_stuck));
        SL_HIP(hipMemcpy(&stuck, d_stuck, sizeof stuck, hipMemcpyDeviceToHost));
        if (stuck) return fail("sarlacc_amd: internal error: an MSA traceback exceeded its step bound");
    }
    std::vector<long long> ooff(static_cast<size_t>(ngroups) + 1, 0);
    for (int64_t g = 0; g < ngroups; ++g) {
        width_out[g] = width[g];
        ooff[g + 1] = ooff[g] + static_cast<long long>(width[g]) * groups[g].nreads;
        out_off[g + 1] = ooff[g + 1];
    }
    res->d_members = d_mem;
    if (!want_rows) return 0;  // sizing call
    if (out_cap >= 0 && out_cap < ooff[ngroups]) return fail("sarlacc_amd: MSA output buffer too small (%lld needed)", ooff[ngroups]);
    if (ooff[ngroups] == 0) return 0;
    long long* d_ooff; long long* d_rg; int* d_rp; uint8_t* d_out;
    SL_TRY(upload("msa.ooff", ooff.data(), ooff.size(), &d_ooff, s));
    SL_TRY(upload("msa.rg", row_group.data(), row_group.size(), &d_rg, s));
    SL_TRY(upload("msa.rp", row_pos.data(), row_pos.size(), &d_rp, s));
    SL_TRY(scratch("msa.out", static_cast<size_t>(ooff[ngroups]), &d_out));
    m.out_off = d_ooff; m.out = d_out;
    const long long nrows = static_cast<long long>(row_group.size());
    hipLaunchKernelGGL(k_msa_write, dim3(static_cast<unsigned>(nrows)), dim3(256), 0, s, m, d_rg, d_rp, nrows);
    SL_HIP(hipGetLastError());
    SL_TRY(c.stage_end("msa_merge", s));
    res->d_out = d_out;
    if (timing) {
        SL_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "msa_run: job list %.3f s | uploads + launches %.3f s | overlap hook %.3f s | kernels + row write %.3f s\n",
                tm1 - tm0, tm2 - tm1, tm3 - tm2, now() - tm3);
    }
    return 0;
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" int sarlacc_quick_msa(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq,
                                 const int64_t* seq_off, int64_t nseq, double match, double mismatch,
                                 double gap_extension, double gap_opening, int bandwidth, int32_t* width_out,
                                 int64_t* out_off, char* out, int64_t out_cap) {
    if (ngroups < 0 || nseq < 0) return fail("sarlacc_amd: negative sizes");
    out_off[0] = 0;
    if (ngroups == 0) return 0;
    MsaResult res;
    res.width.assign(static_cast<size_t>(ngroups), 0);
    res.out_off.assign(static_cast<size_t>(ngroups) + 1, 0);
    const int rc = msa_run(grp_off, grp, ngroups, seq, seq_off, nseq, match, mismatch, gap_extension, gap_opening, bandwidth,
                           out != nullptr, out_cap, &res);
    std::copy(res.width.begin(), res.width.end(), width_out);
    std::copy(res.out_off.begin(), res.out_off.end(), out_off);
    if (rc) return rc;
    if (res.d_out) SL_HIP(hipMemcpy(out, res.d_out, static_cast<size_t>(out_off[ngroups]), hipMemcpyDeviceToHost));
    return 0;
}
