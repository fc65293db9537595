// msa_pairwise.hip -- banded global Gotoh of one read against one centre per wavefront (gfx950).
//
// The pairwise stage of quick_msa (/root/reference/src/quick_msa.cpp:25-35: Score<int, Simple>,
// global, banded) for both MSA drivers (msa.hip spec v1, msa2.hip spec v2).  Recurrences, band and
// tie rules are DESIGN.md section 5's: E = max(H_up + open, E_up + ext), F likewise from the left,
// H = max(diag, E, F); ties: diagonal >= vertical >= horizontal, open >= extend; band on the
// diagonals j - i in [min(0, lc - lr) - bw, max(0, lc - lr) + bw].
//
// Scheduling (both kernels): band coordinates (i, x), x = j - i - dlo.  The inputs of a cell are
//   diagonal (i-1, x)      vertical (i-1, x+1)      horizontal (i, x-1)
// so with the time step  t = 2 i + x  every input was produced at step t-1 or t-2: no prefix scan
// over the row.  Lane l owns the C consecutive diagonals x = C l + k; at step t it updates its cells
// with k = t (mod 2) -- C/2 cells per lane per step, every lane busy on every step -- reading the
// neighbouring diagonals k-1 / k+1 from its own registers, or from the adjacent lane with one DPP
// shift (H and F from the left on even steps, H and E from the right on odd steps).
//
//   k_msa_pairwise_pk<C>  (default) scores as packed 16-bit pairs: the two cells a lane updates in
//       one step sit in the halves of one VGPR and every add / max is a v_pk_*_i16, so the
//       recurrence costs half the instructions; traceback flags come from equality of the packed
//       max with its first operand (no v_cmp -> SGPR -> v_addc chains).  Scores are kept in a cost
//       domain (match 0, everything else <= 0, see cost_domain()) relative to a base that is moved
//       every 128 steps, so 16 bits suffice for any read length; the penalties only have to satisfy
//       the spread bound checked in pk_range_ok().
//   k_msa_pairwise_ad<C>  the same schedule on 32-bit scores, for penalty sets outside that bound
//       (and SARLACC_MSA_INT32=1 for A/B runs).
// Both write 4 traceback bits per cell into a per-wave HBM tile (coalesced dwords) and walk it back
// through an LDS window; runs of diagonal moves stay on one band diagonal and are consumed up to 64
// rows per step.
#include <cstdlib>
#include <cstring>

#include "msa_common.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace sarlacc {

constexpr int MSA_NEG = -(1 << 28);
constexpr int PK_SPREAD_MAX = 11000;  // largest score spread inside one anti-diagonal the packed kernel accepts
constexpr int PK_REBASE_ROWS = 32;    // word rows (of 4 steps) between two moves of the score base
constexpr int MSA_WIN = 8;            // tile rows x 64 lanes of traceback codes held in LDS during the walk
template <bool B>
struct Flag2 { static constexpr bool value = B; };

template <int CTRL>
__device__ __forceinline__ int dpp_int(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;

__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(MSA_NEG, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------------------
// Traceback walk shared by both kernels (wave-uniform state machine).
//   Tile layout: groups of RG tile rows (64 words each) hold the codes of TSTEP consecutive steps;
//   locate(i, x) -> (row group, row inside the group, bit shift of the 4-bit code, lane).
//   Code bits after the optional inversion: bit 0 d >= max(e, f); bit 1 e >= f; bit 2 E opened;
//   bit 3 F opened.
//   OUT 0 (spec v1): ins[p] = read bases inserted before centre position p, aln[p] = centre base p matched.
//   OUT 1 (spec v2): position maps of both directions (0xFFFF = opposite a gap) + (equal, aligned) counts.
//   RDREV: the read codes in LDS are stored last base first (packed kernel).
template <int OUT, bool INV, bool RDREV, int TSTEP, int RG, typename Word, typename Locate>
__device__ __forceinline__ void msa_walk(const MsaArgs& A, const MsaJob& J, int jobidx, int dlo, const Word* tile, int ngrp,
                                         Word* s_tb, const uint8_t* s_rd, const uint8_t* s_ct, Locate locate) {
    const int lane = threadIdx.x;
    const int lr = J.lr, lc = J.lc;
    constexpr int WL = 8, WR = MSA_WIN * 64 / WL, WG = WR / RG;
    uint16_t* ins = nullptr; uint8_t* aln = nullptr; uint16_t* mapA = nullptr; uint16_t* mapB = nullptr;
    if (OUT == 0) { ins = A.ins + J.out_off; aln = A.aln + J.out_off; }
    else { mapA = A.map + J.out_off; mapB = A.map + J.out2_off; }
    int i = lr, j = lc, state = 0, cnt = 0, nequal = 0, ndiag = 0;
    int glo = ngrp, llo = 0;   // first row group / first lane held in the window (none yet)
    int walk_budget = 2 * (lr + lc) + 64;   // every iteration consumes a row, a column or changes state once
    while ((i > 0 || j > 0) && --walk_budget >= 0) {
        const int x = j - i - dlo;
        int grp, rin, shift, ln;
        locate(i, x, grp, rin, shift, ln);
        if (grp < glo || ln < llo || ln >= llo + WL) {
            glo = max(0, grp - (WG - 1));
            llo = min(max(ln - WL / 2, 0), 64 - WL);
            __syncthreads();
            for (int idx = lane; idx < WG * RG * WL; idx += 64) {
                const int r = idx / WL, cl = idx % WL;
                if (glo * RG + r < ngrp * RG) s_tb[idx] = tile[static_cast<size_t>(glo * RG + r) * 64 + llo + cl];
            }
            __syncthreads();
        }
        if (state == 0) {
            // run of diagonal moves: cell (i - m, j - m) keeps x; lane m inspects it
            const int reach = min(min(i, j), ((2 * i + x - glo * TSTEP) >> 1) + 1);
            unsigned tl = 1;
            if (lane < reach) {
                int g2, r2, sh2, l2;
                locate(i - lane, x, g2, r2, sh2, l2);
                unsigned c = static_cast<unsigned>(s_tb[((g2 - glo) * RG + r2) * WL + (l2 - llo)] >> sh2) & 1u;
                if (INV) c ^= 1u;
                tl = c ^ 1u;   // bit 0: diagonal
            }
            const unsigned long long nd = __ballot(tl != 0);
            const int run = nd ? static_cast<int>(__builtin_ctzll(nd)) : 64;
            if (run > 0) {
                if (OUT == 0) {
                    if (lane < run) { ins[j - lane] = (lane == 0) ? static_cast<uint16_t>(cnt) : static_cast<uint16_t>(0); aln[j - lane - 1] = 1; }
                } else {
                    bool eq = false;
                    if (lane < run) {
                        mapA[j - 1 - lane] = static_cast<uint16_t>(i - 1 - lane);
                        mapB[i - 1 - lane] = static_cast<uint16_t>(j - 1 - lane);
                        eq = s_rd[4 + (RDREV ? lr - i + lane : i - 1 - lane)] == s_ct[4 + j - 1 - lane];
                    }
                    nequal += __popcll(__ballot(eq));
                    ndiag += run;
                }
                cnt = 0; i -= run; j -= run;
                continue;
            }
        }
        // the walk's state is the same in every lane: telling the compiler so keeps it in SGPRs / on the scalar unit
        unsigned t = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(s_tb[((grp - glo) * RG + rin) * WL + (ln - llo)] >> shift))) & 15u;
        if (INV) t ^= 15u;
        if (state == 0) {
            state = (t & 1u) ? 0 : ((t & 2u) ? 1 : 2);   // diagonal, else vertical if e >= f, else horizontal
            continue;
        }
        if (state == 1) {               // read base i-1 inserted before centre position j
            if (OUT == 0) ++cnt;
            else if (lane == 0) mapB[i - 1] = 0xFFFF;
            state = (t & 4) ? 0 : 1;
            --i;
        } else {                        // centre base j-1 opposite a gap
            if (lane == 0) {
                if (OUT == 0) { ins[j] = static_cast<uint16_t>(cnt); aln[j - 1] = 0; }
                else mapA[j - 1] = 0xFFFF;
            }
            cnt = 0;
            state = (t & 8) ? 0 : 2;
            --j;
        }
    }
    if (lane == 0) {
        if (OUT == 0) ins[0] = static_cast<uint16_t>(cnt);
        else A.stats[jobidx] = make_int2(nequal, ndiag);
        if (walk_budget < 0) atomicExch(A.stuck, 1);
    }
}

// The diagonal alignment of a pair whose length difference alone exceeds the band cap (msa_pair_bandwidth() < 0).
template <int OUT, bool RDREV>
__device__ __forceinline__ void msa_diagonal_pair(const MsaArgs& A, const MsaJob& J, int jobidx, const uint8_t* s_rd, const uint8_t* s_ct) {
    const int lane = threadIdx.x;
    const int lr = J.lr, lc = J.lc, k = min(lr, lc);
    if (OUT == 0) {
        uint16_t* ins = A.ins + J.out_off;
        uint8_t* aln = A.aln + J.out_off;
        for (int p = lane; p <= lc; p += 64) {
            ins[p] = (p == lc && lr > lc) ? static_cast<uint16_t>(lr - lc) : static_cast<uint16_t>(0);
            if (p < lc) aln[p] = p < k ? 1 : 0;
        }
    } else {
        uint16_t* mapA = A.map + J.out_off;
        uint16_t* mapB = A.map + J.out2_off;
        for (int p = lane; p < lc; p += 64) mapA[p] = p < k ? static_cast<uint16_t>(p) : static_cast<uint16_t>(0xFFFF);
        for (int p = lane; p < lr; p += 64) mapB[p] = p < k ? static_cast<uint16_t>(p) : static_cast<uint16_t>(0xFFFF);
        int nequal = 0;
        for (int p0 = 0; p0 < k; p0 += 64) {
            const int p = p0 + lane;
            const bool eq = p < k && s_rd[4 + (RDREV ? lr - 1 - p : p)] == s_ct[4 + p];
            nequal += __popcll(__ballot(eq));
        }
        if (lane == 0) A.stats[jobidx] = make_int2(nequal, k);
    }
}

// stages the Dna5 codes (<< SHIFT) of centre and read into LDS (4 bytes of padding on either side)
// pad_c / pad_r (0 or 1) shift the two arrays by one byte (the packed kernel makes its read addresses even).
template <bool RDREV, int SHIFT>
__device__ __forceinline__ void stage_codes(const MsaArgs& A, const MsaJob& J, uint8_t*& s_ct, uint8_t*& s_rd, int pad_c = 0, int pad_r = 0) {
    const int lane = threadIdx.x;
    const int lr = J.lr, lc = J.lc;
    const uint8_t* rd = A.seq + J.read_off;
    const uint8_t* ct = A.seq + J.ctr_off;
    s_rd = s_ct + ((lc + 12 + 3) & ~3) + pad_r;   // read codes after the centre codes
    s_ct += pad_c;
    __syncthreads();
    for (int p = lane; p < lc; p += 64) s_ct[4 + p] = dna5_code(ct[p]) << SHIFT;
    for (int p = lane; p < lr; p += 64) s_rd[4 + (RDREV ? lr - 1 - p : p)] = dna5_code(rd[p]) << SHIFT;
    if (lane < 4) { s_ct[lane] = 0xf0; s_rd[lane] = 0xf1; s_ct[4 + lc + lane] = 0xf0; s_rd[4 + lr + lane] = 0xf1; }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// 32-bit kernel.  Traceback nibble of a cell (raw outcomes, pushed most significant first):
//   bit 3 F opened (H_left + go >= F_left + ge)   bit 2 E opened   bit 1 e >= f   bit 0 d >= max(e, f)
// A sub-block (2 steps) yields C nibbles per lane, packed into words of 8 (C = 4: two sub-blocks per
// dword) or C nibbles and stored coalesced.

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

template <int C, int OUT>
__global__ void __launch_bounds__(64) k_msa_pairwise_ad(const MsaArgs A) {
    using Word = typename AdWord<C>::type;
    constexpr int SPW = AdWord<C>::SPW;   // sub-blocks per stored word
    constexpr int H2 = C / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    Word* const s_tb = reinterpret_cast<Word*>(smem);
    uint8_t* const s_ct0 = reinterpret_cast<uint8_t*>(s_tb + MSA_WIN * 64);   // centre codes, padding in front
    const int lane = threadIdx.x;
    const int ma = A.ma, mm = A.mm, go = A.go, ge = A.ge;
    Word* const tile = static_cast<Word*>(A.tb) + static_cast<size_t>(blockIdx.x) * A.tb_per_wave;

    for (int jobn = blockIdx.x; jobn < A.njobs; jobn += gridDim.x) {
        const int jobidx = A.order ? A.order[jobn] : jobn;
        const MsaJob J = A.jobs[jobidx];
        const int lr = J.lr, lc = J.lc;
        const int bw = msa_pair_bandwidth(A.bw, lr, lc);
        const int dlo = min(0, lc - lr) - bw;
        const int dhi = max(0, lc - lr) + bw;
        const int B = dhi - dlo + 1;
        uint8_t* s_ct = s_ct0;
        uint8_t* s_rd;
        stage_codes<false, 0>(A, J, s_ct, s_rd);
        if (bw < 0) {   // uniform over the wave
            msa_diagonal_pair<OUT, false>(A, J, jobidx, s_rd, s_ct);
            continue;
        }

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

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // code of cell (ii, xx): t = 2 ii + xx, sub-block ws = t >> 1
        auto locate = [&](int ii, int xx, int& grp, int& rin, int& shift, int& ln) {
            const int t = 2 * ii + xx;
            const int ws = t >> 1;
            const int k = xx % C;
            ln = xx / C;
            grp = ws / SPW;
            rin = 0;
            // cells are pushed in the order (sub-block, parity, h), the first one ends up on top
            const int q = (ws % SPW) * C + (k & 1) * H2 + (k >> 1);
            shift = 4 * (SPW * C - 1 - q);
        };
        msa_walk<OUT, false, false, 2 * SPW, 1, Word>(A, J, jobidx, dlo, tile, nwords, s_tb, s_rd, s_ct, locate);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Packed 16-bit kernel.
//
// Cost domain.  For a global alignment the number of aligned pairs is (lr + lc - gap characters) / 2,
// so with k = 2 (k = 1 when match == 0) the scores  match' = 0, mismatch' = k (mismatch - match),
// open' = k open - match, extend' = k extend - match  (open = cost of the first gap character)
// give every alignment the score k S - match (lr + lc): the same optimum, and -- since the two
// candidates of every max in the recurrence end at the same cell, i.e. have consumed the same number
// of characters -- the same outcome of every comparison, ties included.  All steps are then <= 0 and
// the kernel works on their negations: unsigned COSTS, min instead of max, "infinity" = PK_INF.
// Measured on MI355X (tools/ubench_valu.hip): v_add_u32 / v_xor_b32 issue in ~2.7 cycles per wave64,
// every VOP3 / packed / DPP instruction in ~4.2 -- so the additions are plain 32-bit adds on both
// halves at once (no carry can cross: every sum stays below 2^16, see pk_range_ok()), only the
// minima and the flag normalisation are packed instructions.
//
// Registers (per lane, M = C / 4 VGPRs each): the cells of one parity, two per register --
//   even cells k = 2h: Hev[h >> 1], half h & 1;  odd cells k = 2h + 1: Hod[h >> 1], half h & 1.
// Even step: the vertical neighbours (k + 1) are the odd registers as they are; the horizontal
// neighbours (k - 1) are the odd registers shifted by one half (v_alignbit with the previous
// register, or with the left lane's last register delivered by DPP).  Odd step: mirrored.
// Flags: with e = min(eop, eex), "E opened" (eop <= eex) is e == eop; likewise f == fop, m == e for
// "e at least as good as f" and h == d for the diagonal; min(x ^ y, 1) turns each into one bit per half
// and a packed mad shifts it into the lane's code accumulator: after 4 steps each half holds 4 codes,
// one dword per register is stored.  Stored codes are the complements of the 32-bit kernel's
// (1 = "not equal").
// Mismatch cost: the code bytes (stored as code << 4) are widened into the HIGH byte of each half, so
// the xor of two different codes is >= 4096 and min(xor, mismatch cost) is the cost itself.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr unsigned PK_INF = 0x5000;   // per half

__device__ __forceinline__ unsigned pk_splat(unsigned v) { return (v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// 1 per half where a != b.  Inline asm: written as min(a ^ b, 1) in C the compiler recognises a packed
// "not equal", has no packed compare to select and falls back to two v_cmp + v_cndmask + v_perm per flag.
__device__ __forceinline__ unsigned pk_ne(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a ^ b));
    return r;
}
// per half: K * a + b
template <int K>
__device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_mad_u16 %0, %1, %3, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "n"(K));
    return r;
}
// bytes B, B + 1 of the 8 code bytes (hi : lo) as the HIGH bytes of the two halves of a register
template <int B>
__device__ __forceinline__ unsigned pk_widen(unsigned hi, unsigned lo) {
    return __builtin_amdgcn_perm(hi, lo, 0x000c000cu | (static_cast<unsigned>(B) << 8) | (static_cast<unsigned>(B + 1) << 24));
}
// NW dwords of code bytes from LDS at an EVEN address, as aligned 16-bit reads.  Misaligned LDS reads are
// legal but slow on gfx950: with one misaligned ds_read_b32 per lane per word the LDS array (SQ_LDS_IDX_ACTIVE
// 45 cycles per read), not the VALU, set the pace of this kernel.  Inline asm keeps the halves separate loads
// (fused they would be a 2-byte aligned ds_read_b32 again); the compiler does not track the counters of loads
// issued from inline asm, so lds_words_ready() waits and hands the registers back THROUGH the wait statement --
// nothing that uses them can be scheduled above it.
struct LdsHalves { unsigned lo, hi; };
template <int NW>
__device__ __forceinline__ void lds_issue_words(const uint8_t* p, LdsHalves (&h)[NW]) {
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    lds_u16* q16 = (lds_u16*)p;
#pragma unroll
    for (int q = 0; q < NW; ++q)
        asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2" : "=&v"(h[q].lo), "=&v"(h[q].hi) : "v"(q16 + 2 * q) : "memory");
}
template <int NW>
__device__ __forceinline__ void lds_words_ready(LdsHalves (&a)[NW], LdsHalves (&b)[NW], unsigned (&wa)[NW], unsigned (&wb)[NW]) {
    if constexpr (NW == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(b[0].lo), "+v"(b[0].hi));
    else if constexpr (NW == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(b[1].lo), "+v"(b[1].hi));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(b[1].lo), "+v"(b[1].hi),
                                              "+v"(a[2].lo), "+v"(a[2].hi), "+v"(b[2].lo), "+v"(b[2].hi));
#pragma unroll
    for (int q = 0; q < NW; ++q) { wa[q] = a[q].lo | (a[q].hi << 16); wb[q] = b[q].lo | (b[q].hi << 16); }
}
// (hi : lo) >> 16, i.e. halves (lo.hi, hi.lo)
__device__ __forceinline__ unsigned pk_shift_in(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
__device__ __forceinline__ unsigned pk_half(unsigned v, int hf) { return (v >> (16 * hf)) & 0xffffu; }

__device__ __forceinline__ int wave_min(int v) {
    constexpr int BIG = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// LIN: the first gap character costs no more than a further one (open' <= extend' in the cost domain -- what the
// reference's DEFAULT call gives: R/multiReadAlign.R:47 hands (-gapOpening, -gapExtension) = (-5, -1) to parameters that
// src/quick_msa.cpp:26-31 reads as (extend, open), so the aligner sees open -1, extend -5).  H <= E and H <= F in every
// cell, so  H(up) + open <= E(up) + extend  always: E = H(up) + open, F = H(left) + open, the "opened" flags are constant
// (ties prefer open by the spec), and the recurrence needs neither the E / F registers and their two lane exchanges per
// step nor half of the traceback flags -- the same cells, codes and walk with 17 instead of 29 instructions per step.
template <int C, int OUT, bool LIN>
__global__ void __launch_bounds__(64) k_msa_pairwise_pk(const MsaArgs A) {
    constexpr int H2 = C / 2, M = C / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* const s_tb = reinterpret_cast<uint32_t*>(smem);
    uint8_t* const s_ct0 = reinterpret_cast<uint8_t*>(s_tb + MSA_WIN * 64);
    const int lane = threadIdx.x;
    // costs (A.mm .. A.ge arrive transformed and negated by the host, see cost_domain())
    const unsigned mmc = A.mm, goc = A.go, gec = A.ge;
    const unsigned GO = pk_splat(goc), GE = pk_splat(gec), MM = pk_splat(mmc);
    uint32_t* const tile = static_cast<uint32_t*>(A.tb) + static_cast<size_t>(blockIdx.x) * A.tb_per_wave;

    for (int jobn = blockIdx.x; jobn < A.njobs; jobn += gridDim.x) {
        const int jobidx = A.order ? A.order[jobn] : jobn;
        const MsaJob J = A.jobs[jobidx];
        const int lr = J.lr, lc = J.lc;
        const int bw = msa_pair_bandwidth(A.bw, lr, lc);
        const int dlo = min(0, lc - lr) - bw;
        const int dhi = max(0, lc - lr) + bw;
        const int B = dhi - dlo + 1;
        // code arrays placed so that every lane's code-word address is even (see lds_load_words):
        //   centre words start at s_ct + 3 + 2 w + dlo + (C/2) lane, read words at s_rd + 3 + lr - 2 w + (C/2) lane
        uint8_t* s_ct = s_ct0;
        uint8_t* s_rd;
        stage_codes<true, 4>(A, J, s_ct, s_rd, (dlo + 1) & 1, (lr + 1) & 1);
        if (bw < 0) {   // uniform over the wave
            msa_diagonal_pair<OUT, true>(A, J, jobidx, s_rd, s_ct);
            continue;
        }

        unsigned Hev[M], Hod[M], Eev[M], Eod[M], Fev[M], Fod[M];
        unsigned gouEv[M], geuEv[M], gouOd[M], geuOd[M];   // vertical costs; PK_INF sinks the candidate at the band edge
        unsigned vmEv[M], vmOd[M];                         // 0xffff per half where the cell's diagonal is inside the band
        unsigned acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            Hev[m] = Hod[m] = Eev[m] = Eod[m] = Fev[m] = Fod[m] = pk_splat(PK_INF);
            acc[m] = 0;
            gouEv[m] = geuEv[m] = gouOd[m] = geuOd[m] = vmEv[m] = vmOd[m] = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int xe = lane * C + 2 * (2 * m + hf), xo = xe + 1;
                gouEv[m] |= (xe + 1 < B ? goc : PK_INF) << (16 * hf);
                geuEv[m] |= (xe + 1 < B ? gec : PK_INF) << (16 * hf);
                gouOd[m] |= (xo + 1 < B ? goc : PK_INF) << (16 * hf);
                geuOd[m] |= (xo + 1 < B ? gec : PK_INF) << (16 * hf);
                vmEv[m] |= (xe < B ? 0xffffu : 0u) << (16 * hf);
                vmOd[m] |= (xo < B ? 0xffffu : 0u) << (16 * hf);
            }
        }
        // values arriving from the neighbouring lanes; as DPP destinations lane 0 (63) keeps the PK_INF pair
        int xlH = static_cast<int>(pk_splat(PK_INF)), xlF = xlH, xrH = xlH, xrE = xlH;

        // ---- one step (one parity) on the packed state ----
        auto step_pk = [&](auto par_tag, const unsigned (&pen)[M], int inject) {
            constexpr int par = decltype(par_tag)::value;
            if (par == 0) { xlH = dpp_int<DPP_WAVE_SHR1>(xlH, static_cast<int>(Hod[M - 1])); if (!LIN) xlF = dpp_int<DPP_WAVE_SHR1>(xlF, static_cast<int>(Fod[M - 1])); }
            else { xrH = dpp_int<DPP_WAVE_SHL1>(xrH, static_cast<int>(Hev[0])); if (!LIN) xrE = dpp_int<DPP_WAVE_SHL1>(xrE, static_cast<int>(Eev[0])); }
            unsigned nH[M], nE[M], nF[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                unsigned upH, upE = 0, lfH, lfF = 0, same;
                if (par == 0) {
                    upH = Hod[m];
                    lfH = pk_shift_in(Hod[m], m > 0 ? Hod[m > 0 ? m - 1 : 0] : static_cast<unsigned>(xlH));
                    if (!LIN) { upE = Eod[m]; lfF = pk_shift_in(Fod[m], m > 0 ? Fod[m > 0 ? m - 1 : 0] : static_cast<unsigned>(xlF)); }
                    same = Hev[m];
                } else {
                    lfH = Hev[m];
                    upH = pk_shift_in(m + 1 < M ? Hev[m + 1 < M ? m + 1 : 0] : static_cast<unsigned>(xrH), Hev[m]);
                    if (!LIN) { lfF = Fev[m]; upE = pk_shift_in(m + 1 < M ? Eev[m + 1 < M ? m + 1 : 0] : static_cast<unsigned>(xrE), Eev[m]); }
                    same = Hod[m];
                }
                const unsigned eop = upH + (par == 0 ? gouEv[m] : gouOd[m]);
                const unsigned fop = lfH + GO;
                unsigned e = eop, f = fop;
                if (!LIN) {
                    e = pk_min(eop, upE + (par == 0 ? geuEv[m] : geuOd[m]));
                    f = pk_min(fop, lfF + GE);
                }
                const unsigned d = same + pen[m];
                const unsigned mn = pk_min(e, f);
                const unsigned hv = pk_min(d, mn);
                // code = 8 [f != fop] + 4 [e != eop] + 2 [mn != e] + [hv != d], shifted into the accumulator (LIN: e is eop, f is fop)
                if (LIN) acc[m] = pk_mad<16>(acc[m], pk_mad<2>(pk_ne(mn, e), pk_ne(hv, d)));
                else acc[m] = pk_mad<16>(acc[m], pk_mad<4>(pk_mad<2>(pk_ne(f, fop), pk_ne(e, eop)), pk_mad<2>(pk_ne(mn, e), pk_ne(hv, d))));
                nH[m] = hv; nE[m] = e; nF[m] = f;
            }
            // first row of the job only: the step that computes cell (0, 0) sets H(0, 0) = 0 in its lane
            // (inject = 2 (register index) + half, -1 otherwise)
            if (inject >= 0) {
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        if (inject == 2 * m + hf) nH[m] &= ~(0xffffu << (16 * hf));
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (par == 0) { Hev[m] = nH[m]; if (!LIN) { Eev[m] = nE[m]; Fev[m] = nF[m]; } }
                else { Hod[m] = nH[m]; if (!LIN) { Eod[m] = nE[m]; Fod[m] = nF[m]; } }
            }
        };

        // Code words of one word row (2 sub-blocks): rw = read codes from s_rd + 4 + lr - ib2 (ib2 = row of cells
        // k = 0, 1 in the SECOND sub-block; the first one starts one byte higher), cw = centre codes from
        // s_ct + 4 + jb1 - 1 (jb1 = column of cell k = 0 in the FIRST sub-block).  H2 + 1 and H2 + 2 bytes are used.
        // One sub-block: steps t0 (even cells) and t0 + 1 (odd cells), t0 even.  Row and column of cell k:
        //   i = t0/2 - (C/2) l - (k >> 1),   j = i + dlo + C l + k
        constexpr int NW = (H2 + 2 + 3) / 4;
        // bytes Bx, Bx + 1 of a code word array as (high bytes of the) halves
        auto widen = [&](auto b_tag, const unsigned (&w)[NW]) -> unsigned {
            constexpr int Bx = decltype(b_tag)::value;
            constexpr int q = Bx / 4, r = Bx % 4;
            return pk_widen<r>(q + 1 < NW ? w[q + 1 < NW ? q + 1 : q] : 0u, w[q]);
        };
        auto subblock_pk = [&](auto sb_tag, const unsigned (&rw)[NW], const unsigned (&cw)[NW], int inj_even, int inj_odd) {
            constexpr int SB = decltype(sb_tag)::value;   // 0: first sub-block of the word row, 1: second
            unsigned pen0[M], pen1[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                // cells h = 2m, 2m + 1: read byte (1 - SB) + h, centre byte SB + h (even step) / SB + h + 1 (odd step)
                unsigned r2 = 0, c2 = 0, c3 = 0;
                if constexpr (M > 0) if (m == 0) { r2 = widen(std::integral_constant<int, 1 - SB>{}, rw); c2 = widen(std::integral_constant<int, SB>{}, cw); c3 = widen(std::integral_constant<int, SB + 1>{}, cw); }
                if constexpr (M > 1) if (m == 1) { r2 = widen(std::integral_constant<int, 3 - SB>{}, rw); c2 = widen(std::integral_constant<int, SB + 2>{}, cw); c3 = widen(std::integral_constant<int, SB + 3>{}, cw); }
                if constexpr (M > 2) if (m == 2) { r2 = widen(std::integral_constant<int, 5 - SB>{}, rw); c2 = widen(std::integral_constant<int, SB + 4>{}, cw); c3 = widen(std::integral_constant<int, SB + 5>{}, cw); }
                if constexpr (M > 3) if (m == 3) { r2 = widen(std::integral_constant<int, 7 - SB>{}, rw); c2 = widen(std::integral_constant<int, SB + 6>{}, cw); c3 = widen(std::integral_constant<int, SB + 7>{}, cw); }
                pen0[m] = pk_min(r2 ^ c2, MM);   // 0 where the codes agree, the mismatch cost where they differ
                pen1[m] = pk_min(r2 ^ c3, MM);
            }
            step_pk(std::integral_constant<int, 0>{}, pen0, inj_even);
            step_pk(std::integral_constant<int, 1>{}, pen1, inj_odd);
        };

        // Steps 0 .. 2 lr + B - 1, in word rows of 4 steps (2 sub-blocks).  No boundary tests anywhere: every
        // cell starts at "infinity", a cell outside the matrix only ever sees such neighbours (its inputs
        // have a smaller row or column) and so stays there, and no cell inside the matrix depends on one with a
        // larger row or column.  The single exception is H(0, 0) = 0, set by the step that computes that cell
        // (t = x0 = -dlo); the word rows before it hold nothing the walk can reach and are skipped.
        const int tsteps = 2 * lr + B;
        const int nwr = (tsteps + 3) / 4;
        const int x0 = -dlo;
        const int w0 = x0 >> 2;
        const int inj_code = (lane == x0 / C) ? (((x0 % C) >> 1) >> 1) * 2 + (((x0 % C) >> 1) & 1) : -1;   // register index, half
        auto code_ptrs = [&](int w, const uint8_t*& rp, const uint8_t*& cp) {
            rp = s_rd + 4 + lr - (2 * w + 1 - H2 * lane);              // read codes of the second sub-block's row ib2 ..
            cp = s_ct + 4 + (2 * w - H2 * lane) + dlo + C * lane - 1;   // centre codes from the first sub-block's jb1 - 1
        };
        auto rebase = [&]() {
            // move the cost base: the best H of the wave goes back to 0 (comparisons are between cells of
            // the same neighbourhood and do not see a common offset).  E, F >= H in every cell and H >= the
            // minimum, so the 32-bit subtraction borrows nothing across the halves.
            int best = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if (pk_half(vmEv[m], hf)) best = min(best, static_cast<int>(pk_half(Hev[m], hf)));
                    if (pk_half(vmOd[m], hf)) best = min(best, static_cast<int>(pk_half(Hod[m], hf)));
                }
            best = wave_min(best);
            const unsigned delta = pk_splat(best < static_cast<int>(PK_INF) ? static_cast<unsigned>(best) : 0u);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const unsigned de = delta & vmEv[m], dd = delta & vmOd[m];
                Hev[m] -= de; Hod[m] -= dd;
                if (!LIN) { Eev[m] -= de; Fev[m] -= de; Eod[m] -= dd; Fod[m] -= dd; }
            }
        };
        // one word row on the code halves (rh, ch) issued one row earlier; those of row w + 1 are issued into (rn, cn)
        auto row = [&](int w, LdsHalves (&rh)[NW], LdsHalves (&ch)[NW], LdsHalves (&rn)[NW], LdsHalves (&cn)[NW],
                       int inj0, int inj1, int inj2, int inj3) {
            unsigned rw[NW], cw[NW];
            lds_words_ready<NW>(rh, ch, rw, cw);
            const uint8_t *rp, *cp;
            code_ptrs(w + 1, rp, cp);
            lds_issue_words<NW>(rp, rn);
            lds_issue_words<NW>(cp, cn);
            subblock_pk(std::integral_constant<int, 0>{}, rw, cw, inj0, inj1);
            subblock_pk(std::integral_constant<int, 1>{}, rw, cw, inj2, inj3);
#pragma unroll
            for (int m = 0; m < M; ++m) tile[static_cast<size_t>(w * M + m) * 64 + lane] = acc[m];
        };
        LdsHalves ra[NW], ca[NW], rb[NW], cb[NW];   // code words in flight, double buffered
        {
            const uint8_t *rp, *cp;
            code_ptrs(w0, rp, cp);
            lds_issue_words<NW>(rp, ra);
            lds_issue_words<NW>(cp, ca);
        }
        {
            const int s0 = x0 & 3;   // step of row w0 that computes cell (0, 0)
            row(w0, ra, ca, rb, cb, s0 == 0 ? inj_code : -1, s0 == 1 ? inj_code : -1, s0 == 2 ? inj_code : -1, s0 == 3 ? inj_code : -1);
        }
        int w = w0 + 1;
        for (int blk = 0; w + 1 < nwr; w += 2, ++blk) {
            if ((blk % (PK_REBASE_ROWS / 2)) == PK_REBASE_ROWS / 2 - 1) rebase();
            row(w, rb, cb, ra, ca, -1, -1, -1, -1);
            row(w + 1, ra, ca, rb, cb, -1, -1, -1, -1);
        }
        if (w < nwr) row(w, rb, cb, ra, ca, -1, -1, -1, -1);
        {   // the halves issued by the last row are still in flight: drain them before the walk uses the LDS counters
            unsigned t0[NW], t1[NW];
            if ((nwr - w0) & 1) lds_words_ready<NW>(rb, cb, t0, t1); else lds_words_ready<NW>(ra, ca, t0, t1);
        }

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // code of cell (ii, xx): step t = 2 ii + xx, word row t >> 2, register (k >> 1) >> 1, half (k >> 1) & 1,
        // code t & 3 of the half counted from the top
        auto locate = [&](int ii, int xx, int& grp, int& rin, int& shift, int& ln) {
            const int t = 2 * ii + xx;
            const int k = xx % C, h = k >> 1;
            ln = xx / C;
            grp = t >> 2;
            rin = h >> 1;
            shift = 16 * (h & 1) + 4 * (3 - (t & 3));
        };
        msa_walk<OUT, true, true, 4, M, uint32_t>(A, J, jobidx, dlo, tile, nwr, s_tb, s_rd, s_ct, locate);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Bit-vector kernel for the unit-cost linear regime (the reference's DEFAULT scores: match 0, mismatch -1 and, after the
// swapped reading of src/quick_msa.cpp:26-31, every gap character -1): the banded DP is then plain Levenshtein distance
// inside the band, and one 32-bit operation advances 32 cells (Myers 1999, in Hyyro's 2003 form).
//   * ONE PAIR PER LANE, 64 pairs per wavefront, no cross-lane traffic.  A lane walks the columns j = 1 .. lc of its
//     pair with the band's rows i = j - dhi .. j - dlo as a vector of vertical deltas (Pv: +1, Mv: -1) that slides down
//     one row per column; bit 32 NW - 1 is always the band's bottom row, so the base entering the band and its
//     vertical delta are inserted at a fixed bit, and the band's top (bit OFF = 32 NW - B) is a per-lane mask.
//   * Cells outside the band are "infinity" in the specification (oracle/msa.c orc_msa_pairwise).  A cell below the
//     bottom edge is emulated by vertical delta +1, one above the top edge by horizontal delta +1: either makes the move
//     from it cost diagonal + 2 >= diagonal + mismatch, so it never wins and never ties its way into the traceback --
//     the values and decisions inside the band are those of the banded DP (tools model + tests against the oracle).
//     Rows above row 0 are given D[i][j] = j - i (vertical delta -1, horizontal +1), which the recurrence preserves
//     and which leaves D[0][j] = j in row 0.
//   * Traceback: per cell "the diagonal is optimal" (Dg = Eq | ~D0) and "the step from above is" (the new Pv); the walk
//     takes the diagonal if it can, else up if it can, else left -- the spec's order (d >= e, f; then e >= f).  Two bits
//     per cell, 2 NW words per column and lane, one contiguous record per column in the lane's own part of the tile:
//     the walk, per lane as well (~lr + lc dependent steps), then needs one 64-byte line per step and reads its lines
//     in address order (records interleaved over the lanes made every step two random 64-byte reads from HBM: 0.70 s
//     for 4.5 M pairs, all of it the walk).
//   * PHASE 1 / 2: the two halves as kernels of their own over the batches [batch0, batch1) of a chunk, every batch with its
//     own tile -- all wavefronts of the first compute (instruction latency hidden by each other instead of sharing the SIMDs
//     with waiting walks), the second runs at several times the residency (no LDS, few registers), which is what a walk
//     that waits for memory needs.  PHASE 0: both in one kernel, one tile per resident wavefront (small job lists).
//   * NS < NW: only the words CORE0 .. CORE0 + NS - 1 of every record are kept.  The diagonals from (0, 0) to (lr, lc) sit at
//     bits 32 NW - 1 - bandwidth - |lc - lr| .. 32 NW - 1 - bandwidth whatever the pair (the band is anchored at its bottom
//     bit), so for the usual bandwidths one fixed window of four words holds every path that drifts less than ~60 diagonals
//     -- all alignments of reads of one molecule -- at half the records.  A walk that needs another word stops and marks its
//     pair (move count BV_REDO); the marked pairs are collected and run again with whole records (k_bv_collect_redo).
constexpr uint32_t BV_REDO = 0xFFFFFFFFu;
constexpr int BV_CORE0 = 3, BV_CORE_WORDS = 4;   // NW = 8, bandwidths 96 .. 127: bits 96 .. 223 around the main diagonals (bits 128 - |lc - lr| .. 159)

template <int NW, int PHASE, int NS>
__global__ void __launch_bounds__(64) k_msa_pairwise_bv(const MsaArgs A) {
    constexpr int CORE0 = NS < NW ? BV_CORE0 : 0;
    constexpr int BV_LSTRIDE = 2 * NS * 4 + 4;   // dwords per pair in the LDS block (padded: the pairs' pieces spread over the banks)
    __shared__ __align__(16) uint32_t s_rec[PHASE == 2 ? 4 : 64 * BV_LSTRIDE];
    const int lane = threadIdx.x;
    for (int batch = A.batch0 + blockIdx.x; batch < A.batch1; batch += gridDim.x) {
        uint4* const tile = static_cast<uint4*>(A.tb) + static_cast<size_t>(PHASE == 0 ? blockIdx.x : batch - A.batch0) * (A.tb_per_wave / 4);
        const int jobn = batch * 64 + lane;
        bool on = jobn < A.njobs;
        const int jobidx = on ? (A.order ? A.order[jobn] : jobn) : 0;
        MsaJob J{};
        if (on) J = A.jobs[jobidx];
        if (A.skip_wide && on && msa_pair_band(A.bw, J.lr, J.lc) > 256) on = false;   // a job of a wider band class: the packed kernel's
        const int lr = on ? J.lr : 0, lc = on ? J.lc : 0;
        const int bw = msa_pair_bandwidth(A.bw, lr, lc);
        uint16_t* const mapA = A.map + J.out_off;
        uint16_t* const mapB = A.map + J.out2_off;
        const uint8_t* const rd = A.seq + J.read_off;
        const uint8_t* const ct = A.seq + J.ctr_off;
        auto code_of = [](uint32_t b) -> uint32_t {   // dna5_code without branches
            const uint32_t u = b & 0xdfu, x = (u >> 1) & 3u, c = x ^ (x >> 1);
            return ((0x54474341u >> (8 * c)) & 0xffu) == u ? c : 4u;
        };
        if (PHASE != 2 && on && bw < 0) {   // the diagonal alignment (msa_diagonal_pair), per lane
            const int k = min(lr, lc);
            if (A.map) {
                int nequal = 0;
                for (int p = 0; p < lc; ++p) mapA[p] = p < k ? static_cast<uint16_t>(p) : static_cast<uint16_t>(0xFFFF);
                for (int p = 0; p < lr; ++p) mapB[p] = p < k ? static_cast<uint16_t>(p) : static_cast<uint16_t>(0xFFFF);
                for (int p = 0; p < k; ++p) nequal += code_of(rd[p]) == code_of(ct[p]);
                A.stats[jobidx] = make_int2(nequal, k);
            } else {   // spec v1's outputs: insertions before every centre position, centre bases matched
                uint16_t* const ins = A.ins + J.out_off;
                uint8_t* const aln = A.aln + J.out_off;
                for (int p = 0; p <= lc; ++p) ins[p] = (p == lc && lr > lc) ? static_cast<uint16_t>(lr - lc) : static_cast<uint16_t>(0);
                for (int p = 0; p < lc; ++p) aln[p] = p < k ? 1 : 0;
            }
        }
        const bool dp = on && bw >= 0;
        const int dlo = min(0, lc - lr) - bw, dhi = max(0, lc - lr) + bw;
        const int B = dp ? dhi - dlo + 1 : 1;
        const int OFF = 32 * NW - B;       // bit of the band's top row
        const int lcd = dp ? lc : 0;       // columns this lane computes
        if (PHASE != 2) {
        // column 0: rows i <= 0 carry Mv, rows 1 .. -dlo carry Pv; planes hold code 7 (equal to nothing) outside 1 .. lr
        uint32_t Pv[NW], Mv[NW], R0[NW], R1[NW], R2[NW];
        uint32_t inb[NW], top[NW];   // bits of the band (>= OFF); the bit of its top row, where the edge's horizontal delta +1 enters
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            Pv[w] = Mv[w] = 0u; R0[w] = R1[w] = R2[w] = ~0u;
            const int n = OFF - 32 * w;   // bits of this word below the band
            inb[w] = n <= 0 ? ~0u : (n >= 32 ? 0u : ~((1u << n) - 1u));
            top[w] = (n >= 0 && n < 32) ? (1u << n) : 0u;
        }
        // the distance itself, for the count of equal pairs: D moves by 1 - D0 from a cell to the next one on its diagonal, and
        // the diagonal of the final cell keeps its bit; summed over the columns from |lc - lr| (rows above row 0 add nothing)
        uint32_t fin[NW];
        {
            const int pbit = OFF + dhi - (lc - lr);
#pragma unroll
            for (int w = 0; w < NW; ++w) fin[w] = (dp && (pbit >> 5) == w) ? (1u << (pbit & 31)) : 0u;
        }
        int dist = lc > lr ? lc - lr : lr - lc;
        if (dp) {
            for (int b = 0; b < B; ++b) {
                const int i = b - dhi, bit = OFF + b;
                uint32_t c = 7u;
                if (i >= 1 && i <= lr) c = code_of(rd[i - 1]);
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    if ((bit >> 5) == w) {
                        const uint32_t m = 1u << (bit & 31);
                        if (i <= 0) Mv[w] |= m; else Pv[w] |= m;
                        if (!(c & 1u)) R0[w] &= ~m;
                        if (!(c & 2u)) R1[w] &= ~m;
                        if (!(c & 4u)) R2[w] &= ~m;
                    }
            }
        }
        // bases of columns 1 .. 4 / 5 .. 8 / 9 .. 12 (see the fetch inside the loop)
        uint32_t rbuf = 0u, cbuf = 0u, rnext = 0u, cnext = 0u, rnext2 = 0u, cnext2 = 0u;
        if (dp) {
            auto grab = [&](int j1, uint32_t& rb, uint32_t& cb) {
                const int i1 = j1 - dlo;
                for (int e = 0; e < 4; ++e) {
                    if (i1 + e >= 1 && i1 + e <= lr) rb |= static_cast<uint32_t>(rd[i1 + e - 1]) << (8 * e);
                    if (j1 + e <= lc) cb |= static_cast<uint32_t>(ct[j1 + e - 1]) << (8 * e);
                }
            };
            grab(1, rnext, cnext);
            grab(5, rnext2, cnext2);
        }
        int lcw = lcd;   // columns of the longest pair of the wave
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lcw = max(lcw, __shfl_xor(lcw, d));

        for (int j = 1; j <= lcw; ++j) {
            const bool act = j <= lcd;
            // the base entering the band at the bottom (row i = j - dlo) and the column's base, four columns per fetch, fetched
            // EIGHT columns ahead: a wait for a load also waits for every store issued before it, so the fetch has to be
            // older than the record stores it would otherwise sit behind
            const int inew = j - dlo;
            if ((j & 3) == 1) {
                rbuf = rnext; cbuf = cnext; rnext = rnext2; cnext = cnext2;
                rnext2 = cnext2 = 0u;
                const int j2 = j + 8, i2 = j2 - dlo;
                if (j2 <= lcd) {
                    if (i2 + 3 <= lr) __builtin_memcpy(&rnext2, rd + i2 - 1, 4);
                    else for (int e = 0; e < 4; ++e) if (i2 + e <= lr) rnext2 |= static_cast<uint32_t>(rd[i2 + e - 1]) << (8 * e);
                    if (j2 + 3 <= lc) __builtin_memcpy(&cnext2, ct + j2 - 1, 4);
                    else for (int e = 0; e < 4; ++e) if (j2 + e <= lc) cnext2 |= static_cast<uint32_t>(ct[j2 + e - 1]) << (8 * e);
                }
            }
            const int e4 = (j - 1) & 3;
            uint32_t cn = 7u, cc = 7u;
            if (act) {
                if (inew <= lr) cn = code_of((rbuf >> (8 * e4)) & 0xffu);
                cc = code_of((cbuf >> (8 * e4)) & 0xffu);
            }
            const uint32_t m0 = 0u - (cc & 1u), m1 = 0u - ((cc >> 1) & 1u), m2 = 0u - ((cc >> 2) & 1u);
            // slide down one row; bottom row: vertical delta +1 (its left neighbour lies outside the band)
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const bool last = w == NW - 1;
                Pv[w] = last ? ((Pv[w] >> 1) | 0x80000000u) : __builtin_amdgcn_alignbit(Pv[w + (last ? 0 : 1)], Pv[w], 1);
                Mv[w] = last ? (Mv[w] >> 1) : __builtin_amdgcn_alignbit(Mv[w + (last ? 0 : 1)], Mv[w], 1);
                R0[w] = last ? ((R0[w] >> 1) | ((cn & 1u) << 31)) : __builtin_amdgcn_alignbit(R0[w + (last ? 0 : 1)], R0[w], 1);
                R1[w] = last ? ((R1[w] >> 1) | (((cn >> 1) & 1u) << 31)) : __builtin_amdgcn_alignbit(R1[w + (last ? 0 : 1)], R1[w], 1);
                R2[w] = last ? ((R2[w] >> 1) | (((cn >> 2) & 1u) << 31)) : __builtin_amdgcn_alignbit(R2[w + (last ? 0 : 1)], R2[w], 1);
            }
            uint32_t Dg[NW], Up[NW];
            uint32_t carry = 0u, ph_in = 0u, mh_in = 0u, grew = 0u;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t Eq = ~((R0[w] ^ m0) | (R1[w] ^ m1) | (R2[w] ^ m2));
                const uint32_t band_pv = Pv[w] & inb[w];   // rows that slid out through the top edge no longer count
                const uint32_t addend = Eq & band_pv;
                const unsigned long long sum = static_cast<unsigned long long>(addend) + band_pv + carry;
                carry = static_cast<uint32_t>(sum >> 32);
                const uint32_t Xh = (static_cast<uint32_t>(sum) ^ band_pv) | Eq;
                const uint32_t Xv = Eq | Mv[w];
                const uint32_t Ph = Mv[w] | ~(Xh | band_pv);
                const uint32_t Mh = band_pv & Xh;
                Dg[w] = Eq | ~(Xh | Mv[w]);
                grew |= ~(Xh | Mv[w]) & fin[w];   // D0 clear on the final cell's diagonal
                const uint32_t Phs = (Ph << 1) | ph_in | top[w];   // Mh is zero below the band, so Mhs needs nothing at the edge
                const uint32_t Mhs = (Mh << 1) | mh_in;
                ph_in = Ph >> 31; mh_in = Mh >> 31;
                Pv[w] = Mhs | ~(Xv | Phs);
                Mv[w] = Phs & Xv;
                Up[w] = Pv[w];
            }
            if (act && grew) ++dist;
            // record of the column into the lane's LDS block: piece 2 w = Dg[w] of the block's four columns, 2 w + 1 = Up[w]
            {
                const int c4 = (j - 1) & 3;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w >= CORE0 && w < CORE0 + NS) {
                        s_rec[lane * BV_LSTRIDE + (2 * (w - CORE0)) * 4 + c4] = Dg[w];
                        s_rec[lane * BV_LSTRIDE + (2 * (w - CORE0) + 1) * 4 + c4] = Up[w];
                    }
                }
            }
            if ((j & 3) == 0 || j == lcw) {
                // four columns of all 64 pairs -> one block of the tile, [pair][piece] with 16 bytes per piece: every store
                // instruction writes 1 KB of consecutive addresses, and a pair's record (2 NW pieces) is contiguous for the walk
                __syncthreads();
                const size_t blk = static_cast<size_t>((j - 1) >> 2) * (2 * NS * 64 / 4) * 4;   // uint4 index of the block
#pragma unroll
                for (int sidx = 0; sidx < 2 * NS * 64 / 4 / 16; ++sidx) {
                    const int lin = sidx * 64 + lane;                // 16-byte unit inside the block
                    const int pl = lin / (2 * NS), piece = lin % (2 * NS);
                    const uint4 v = *reinterpret_cast<const uint4*>(&s_rec[pl * BV_LSTRIDE + piece * 4]);
                    tile[blk + lin] = v;
                }
                __syncthreads();
            }
        }
        if (dp && A.stats) A.stats[jobidx] = make_int2(dist, 0);   // for k_msa_moves_expand: equal pairs = diagonal moves - (distance - gaps)
        }
        if (PHASE == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (PHASE == 1) continue;

        // ---- the walk, per lane: decisions only.  The moves go to the pair's move string (2 bits each: 0 diagonal, 1 up,
        // 2 left; 16 per word, first move in the low bits), k_msa_moves_expand turns them into the position maps -- the walk
        // then has one store per 16 moves, and no load of it waits behind scattered 2-byte stores (loads and stores share
        // one counter on this ISA).  A step group fetches the records of FOUR columns at the current word of the band (along a
        // diagonal the bit stays where it is) and takes up to four diagonal moves and the one move that ends the run: one
        // memory round trip per ~3 moves instead of one per move. ----
        {
            int i = dp ? lr : 0, j = dp ? lc : 0;
            uint32_t* const mv = A.moves + static_cast<size_t>(jobidx) * A.moves_stride;
            uint32_t macc = 0u;
            int nmoves = 0;
            bool redo = false;
            auto push = [&](uint32_t m) {
                macc |= m << (2 * (nmoves & 15));
                ++nmoves;
                if ((nmoves & 15) == 0) { mv[1 + (nmoves >> 4) - 1] = macc; macc = 0u; }
            };
            int budget = lr + lc + 2;
            while (__ballot(i > 0 || j > 0)) {
                if (!(i > 0 || j > 0)) continue;
                if (--budget < 0) { atomicExch(A.stuck, 1); i = j = 0; continue; }
                if (i == 0) { push(2u); --j; continue; }
                if (j == 0) { push(1u); --i; continue; }
                const int bit = OFF + i - (j - dhi);
                if (bit < OFF || bit >= 32 * NW) { atomicExch(A.stuck, 1); i = j = 0; continue; }   // cannot happen: the codes keep the path inside the band
                const int w = bit >> 5, sh = bit & 31;
                // the pair's record of the four-column block that holds column j: pieces 2 w (Dg) and 2 w + 1 (Up)
                const int c4 = (j - 1) & 3;
                if (NS < NW && (w < A.core_lo || w > A.core_hi)) { redo = true; i = j = 0; continue; }   // outside the kept words
                const uint4* const rec = tile + (static_cast<size_t>((j - 1) >> 2) * 64 + lane) * (2 * NS);
                const uint4 dq = rec[2 * (w - CORE0)], uq = rec[2 * (w - CORE0) + 1];
                auto pick = [](const uint4& q, int c) -> uint32_t { return c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w)); };
                bool run = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (run && k <= c4 && i > 0 && j > 0) {
                        const uint32_t dgw = pick(dq, c4 - k), upw = pick(uq, c4 - k);
                        if ((dgw >> sh) & 1u) { push(0u); --i; --j; }
                        else {
                            if ((upw >> sh) & 1u) { push(1u); --i; } else { push(2u); --j; }
                            run = false;
                        }
                    }
                }
            }
            if (dp) {
                if (nmoves & 15) mv[1 + (nmoves >> 4)] = macc;
                mv[0] = redo ? BV_REDO : static_cast<uint32_t>(nmoves);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
}

// The move strings of k_msa_pairwise_bv -> position maps and (equal, aligned) counts, one wavefront per pair: 64 moves per
// step, their positions from prefix counts over the lanes (ballots), so the map entries of a step are written side by side.
// The whole move string sits in registers first (word k of it in lane k % 64), and four steps' worth of positions are
// worked out before their bases are fetched, so a step does not wait for the memory of the one before.
// the jobs whose walk left the kept words of their records (see NS above), in job order
__global__ void k_bv_collect_redo(const MsaArgs A, int* list, int* count) {
    const int jobn = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    int jobidx = 0;
    if (jobn < A.njobs) {
        jobidx = A.order ? A.order[jobn] : jobn;
        const MsaJob J = A.jobs[jobidx];
        hit = msa_pair_bandwidth(A.bw, J.lr, J.lc) >= 0 && !(A.skip_wide && msa_pair_band(A.bw, J.lr, J.lc) > 256) &&
              A.moves[static_cast<size_t>(jobidx) * A.moves_stride] == BV_REDO;
    }
    const unsigned long long m = __ballot(hit);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0);
    if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = jobidx;
}

__global__ void __launch_bounds__(256) k_msa_moves_expand(const MsaArgs A) {
    const int lane = threadIdx.x & 63;
    const int jobn = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (jobn >= A.njobs) return;
    const int jobidx = A.order ? A.order[jobn] : jobn;
    const MsaJob J = A.jobs[jobidx];
    if (msa_pair_bandwidth(A.bw, J.lr, J.lc) < 0) return;   // the diagonal alignment: written by the pairwise kernel itself
    if (A.skip_wide && msa_pair_band(A.bw, J.lr, J.lc) > 256) return;
    const bool maps = A.map != nullptr;   // spec v2's position maps + counts; otherwise spec v1's insertion counts + matched flags
    uint16_t* const mapA = A.map + J.out_off;
    uint16_t* const mapB = A.map + J.out2_off;
    uint16_t* const ins = A.ins + J.out_off;
    uint8_t* const aln = A.aln + J.out_off;
    int ups = 0;   // moves "up" (read bases opposite a gap) since the last move that consumed a centre column
    const uint32_t* const mv = A.moves + static_cast<size_t>(jobidx) * A.moves_stride;
    const int nmoves = static_cast<int>(mv[0]);
    const int nwords = (nmoves + 15) >> 4;
    constexpr int MAXW = 16;   // 64 lanes x 16 words x 16 moves = 16 384 moves and more: longer strings read their words from memory
    uint32_t wreg[MAXW];
#pragma unroll
    for (int q = 0; q < MAXW; ++q) wreg[q] = (q * 64 + lane < nwords) ? mv[1 + q * 64 + lane] : 0u;
    int i = J.lr, j = J.lc, ndiag = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int m0 = 0; m0 < nmoves; m0 += 256) {
        uint32_t tq[4]; int iq[4], jq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + 64 * u + lane;
            const int wi = m >> 4;                       // word of the move: register wi / 64 of lane wi % 64
            uint32_t word = 0u;
            const int q = (m0 + 64 * u) >> 10;           // the same for every lane of the step (64 moves span 4 words of one register index)
#pragma unroll
            for (int qq = 0; qq < MAXW; ++qq) if (qq == q) word = static_cast<uint32_t>(__shfl(static_cast<int>(wreg[qq]), wi & 63));
            if (q >= MAXW && m < nmoves) word = mv[1 + wi];
            const uint32_t tm = m < nmoves ? (word >> (2 * (m & 15))) & 3u : 3u;
            const unsigned long long bi = __ballot(tm == 0u || tm == 1u), bj = __ballot(tm == 0u || tm == 2u);
            tq[u] = tm; iq[u] = i - __popcll(bi & lt); jq[u] = j - __popcll(bj & lt);   // position before this lane's move
            i -= __popcll(bi); j -= __popcll(bj);
            ndiag += __popcll(__ballot(tm == 0u));
        }
        if (maps) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (tq[u] == 0u) {
                    mapA[jq[u] - 1] = static_cast<uint16_t>(iq[u] - 1);
                    mapB[iq[u] - 1] = static_cast<uint16_t>(jq[u] - 1);
                } else if (tq[u] == 1u) mapB[iq[u] - 1] = 0xFFFF;
                else if (tq[u] == 2u) mapA[jq[u] - 1] = 0xFFFF;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // the read bases inserted before centre position j are the "up" moves met (walking back from the end) since the
                // previous column-consuming move: the lanes between that move and this one, plus what earlier steps left over
                const bool colmove = tq[u] == 0u || tq[u] == 2u;
                const unsigned long long mc = __ballot(colmove), mvld = __ballot(tq[u] != 3u);
                if (colmove) {
                    const unsigned long long below = mc & lt;
                    const int since = below ? lane - (63 - __builtin_clzll(below)) - 1 : lane + ups;
                    ins[jq[u]] = static_cast<uint16_t>(since);
                    aln[jq[u] - 1] = tq[u] == 0u ? 1 : 0;
                }
                const int nv = __popcll(mvld);
                ups = mc ? nv - 1 - (63 - __builtin_clzll(mc)) : ups + nv;
            }
        }
    }
    if (!maps) {
        if (lane == 0) {
            ins[0] = static_cast<uint16_t>(ups);
            if (i != 0 || j != 0) atomicExch(A.stuck, 1);
        }
        return;
    }
    // every move costs 1 except a diagonal move over equal bases: mismatches = distance - gaps
    const int dist = A.stats[jobidx].x;
    const int nequal = ndiag - (dist - (nmoves - ndiag));
    if (lane == 0) {
        A.stats[jobidx] = make_int2(nequal, ndiag);
        if (i != 0 || j != 0) atomicExch(A.stuck, 1);
    }
}

// ---------------------------------------------------------------------------------------------
// host side

// scores -> non-negative costs of the packed kernel; false when they do not fit it
static bool cost_domain(int ma, int mm, int go, int ge, int* mmc, int* goc, int* gec) {
    const int k = ma != 0 ? 2 : 1;
    const long long m2 = static_cast<long long>(k) * (mm - ma), g2 = static_cast<long long>(k) * go - (k == 2 ? ma : 0),
                    e2 = static_cast<long long>(k) * ge - (k == 2 ? ma : 0);
    if (m2 > 0 || g2 > 0 || e2 > 0 || m2 < -4000 || g2 < -4000 || e2 < -4000) return false;
    *mmc = static_cast<int>(-m2); *goc = static_cast<int>(-g2); *gec = static_cast<int>(-e2);
    return true;
}

// The packed kernel adds both halves of a register with one 32-bit addition, so no 16-bit sum may reach
// 2^16.  Finite costs stay below `spread` (a cell is reached from the best cell of its step through a gap
// of at most `band` characters plus as many mismatches, and the base moves every 4 * PK_REBASE_ROWS steps);
// the cells that are still "infinite" at the start drift up by at most one penalty per step until their
// diagonal enters the matrix (<= band + 8 steps).  The largest sum is infinity + drift + infinity (band
// edge) + one penalty; with PK_INF = 0x5000 it fits when spread and drift stay below PK_SPREAD_MAX.
static bool pk_range_ok(int mmc, int goc, int gec, int band) {
    const long long step = std::max(std::max(mmc, goc), gec);
    const long long spread = goc + (static_cast<long long>(gec) + mmc) * (band + 2) + step * (4 * PK_REBASE_ROWS + 8);
    const long long drift = step * (band + 8);
    return spread < PK_SPREAD_MAX && drift < PK_SPREAD_MAX;
}

template <int C, int OUT>
static int launch_ad(const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    // long reads stage more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
    if (lds > 48 * 1024)
        SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msa_pairwise_ad<C, OUT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
    hipLaunchKernelGGL((k_msa_pairwise_ad<C, OUT>), dim3(grid), dim3(64), lds, s, a);
    SL_HIP(hipGetLastError());
    return 0;
}

template <int C, int OUT, bool LIN>
static int launch_pk2(const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024)
        SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msa_pairwise_pk<C, OUT, LIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(lds)));
    hipLaunchKernelGGL((k_msa_pairwise_pk<C, OUT, LIN>), dim3(grid), dim3(64), lds, s, a);
    SL_HIP(hipGetLastError());
    return 0;
}
template <int C, int OUT>
static int launch_pk(const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    // (costs as the kernel sees them: a.go = first gap character, a.ge = every further one)
    return a.go <= a.ge && !option(OPT_MSA_AFFINE) ? launch_pk2<C, OUT, true>(a, grid, lds, s) : launch_pk2<C, OUT, false>(a, grid, lds, s);
}

template <int OUT>
static int launch_class(bool packed, int C, const MsaArgs& a, int grid, size_t lds, hipStream_t s) {
    if (packed) {
        if (C == 4) return launch_pk<4, OUT>(a, grid, lds, s);
        if (C == 8) return launch_pk<8, OUT>(a, grid, lds, s);
        return launch_pk<16, OUT>(a, grid, lds, s);
    }
    if (C == 4) return launch_ad<4, OUT>(a, grid, lds, s);
    if (C == 8) return launch_ad<8, OUT>(a, grid, lds, s);
    return launch_ad<16, OUT>(a, grid, lds, s);
}

int msa_pairwise_launch(const MsaJob* jobs, size_t njobs, const MsaJob* d_jobs, const uint8_t* d_seq, double match,
                        double mismatch, double gap_extension, double gap_opening, int bandwidth, int out_mode,
                        uint16_t* d_ins, uint8_t* d_aln, uint16_t* d_map, int2* d_stats, hipStream_t s, const MsaJobSummary* summary, bool reset_stuck) {
    if (njobs == 0) return 0;
    if (!jobs && !(summary && summary->wide_listed)) return fail("msa_pairwise_launch: no host copy of the jobs and no summary with the wide jobs listed");
    Context& c = ctx();
    // jobs by band class: 4, 8 or 16 diagonals per lane (bands up to 256 / 512 / 1024); one launch per
    // class, so the common narrow bands are not dragged to the widest job's shape
    std::vector<int> order[3];
    MsaJobSummary own;
    if (!summary) {
        for (size_t q = 0; q < njobs; ++q) own.add(bandwidth, jobs[q].lr, jobs[q].lc);
        summary = &own;
    }
    const size_t* cls_n = summary->n;
    const int *cls_lr = summary->lr, *cls_lc = summary->lc, *cls_band = summary->band;
    // the usual case -- every pair in one class -- needs no index list: the launch takes the jobs as they are
    const bool one_class = cls_n[0] == njobs || cls_n[1] == njobs || cls_n[2] == njobs;
    // The planner lists the few jobs of the wide classes itself (msa2.hip): the bit-vector kernels then take the whole job list
    // and skip those lanes, and no 4-million-entry index list is built or uploaded for class 0.
    const int cost_ok_ma = static_cast<int>(match), cost_ok_mm = static_cast<int>(mismatch), cost_ok_go = static_cast<int>(gap_opening),
              cost_ok_ge = static_cast<int>(gap_extension);
    int pm = 0, pg = 0, pe = 0;
    const bool bv_class0 = !option(OPT_MSA_INT32) && option(OPT_MSA_BITVECTOR) >= 0 &&
                           cost_domain(cost_ok_ma, cost_ok_mm, cost_ok_go, cost_ok_ge, &pm, &pg, &pe) && pg <= pe && pm == pg && pg > 0 &&
                           pk_range_ok(pm, pg, pe, cls_band[0]);
    const bool implicit0 = !one_class && summary->wide_listed && bv_class0;
    if (implicit0) {
        order[1] = summary->wide[0];
        order[2] = summary->wide[1];
        std::sort(order[1].begin(), order[1].end());
        std::sort(order[2].begin(), order[2].end());
    } else if (!one_class && summary->wide_listed) {   // (other scorings: class 0 is what the two lists leave)
        order[1] = summary->wide[0];
        order[2] = summary->wide[1];
        std::sort(order[1].begin(), order[1].end());
        std::sort(order[2].begin(), order[2].end());
        order[0].reserve(cls_n[0]);
        size_t p1 = 0, p2 = 0;
        for (size_t q = 0; q < njobs; ++q) {
            if (p1 < order[1].size() && order[1][p1] == static_cast<int>(q)) { ++p1; continue; }
            if (p2 < order[2].size() && order[2][p2] == static_cast<int>(q)) { ++p2; continue; }
            order[0].push_back(static_cast<int>(q));
        }
    } else if (!one_class) {
        for (int k = 0; k < 3; ++k) order[k].reserve(cls_n[k]);
        for (size_t q = 0; q < njobs; ++q) {
            const int band = msa_pair_band(bandwidth, jobs[q].lr, jobs[q].lc);
            order[band <= 256 ? 0 : (band <= 512 ? 1 : 2)].push_back(static_cast<int>(q));
        }
    }
    MsaArgs a{};
    a.seq = d_seq; a.jobs = d_jobs;
    // SeqAn's Score(match, mismatch, gap_extend, gap_open): gap of length k = open + (k-1)*extend
    const int ma = static_cast<int>(match), mm = static_cast<int>(mismatch), go = static_cast<int>(gap_opening),
              ge = static_cast<int>(gap_extension);
    a.bw = bandwidth; a.ins = d_ins; a.aln = d_aln; a.map = d_map; a.stats = d_stats;
    int* d_stuck;
    SL_TRY(scratch("msa.stuck", 1, &d_stuck));
    if (reset_stuck) SL_HIP(hipMemsetAsync(d_stuck, 0, sizeof(int), s));
    a.stuck = d_stuck;
    int mmc = 0, goc = 0, gec = 0;
    const bool domain_ok = cost_domain(ma, mm, go, ge, &mmc, &goc, &gec) && !option(OPT_MSA_INT32);
    for (int cls = 0; cls < 3; ++cls) {
        if (cls_n[cls] == 0) continue;
        const int C = 4 << cls;
        const bool packed = domain_ok && pk_range_ok(mmc, goc, gec, cls_band[cls]);
        if (packed) { a.ma = 0; a.mm = mmc; a.go = goc; a.ge = gec; }
        else { a.ma = ma; a.mm = mm; a.go = go; a.ge = ge; }
        // traceback tile of one resident wave, 4 bits per cell: packed kernel C/4 dwords per lane per 4 steps,
        // 32-bit kernel one (C = 16: 64-bit) word per lane per 2 SPW steps
        // unit-cost linear regime (the default scores), maps + stats, bands of up to 256 diagonals: the bit-vector kernels
        if (cls == 0 && bv_class0) {
            const int NWb = cls_band[cls] <= 128 ? 4 : 8;
            // partial records (the four words around the main diagonals) for the usual bandwidths on large lists; whole records otherwise
            const int core_opt = option(OPT_MSA_BITVECTOR_CORE);
            const bool core = NWb == 8 && bandwidth >= 96 && bandwidth <= 127 && core_opt >= 0;
            const int NSb = core ? BV_CORE_WORDS : NWb;
            const size_t per_full = (static_cast<size_t>(cls_lc[cls]) + 8) * 2 * NWb * 64;   // 32-bit words: 2 NW per column and pair, in blocks of four columns
            const size_t per_wave = (static_cast<size_t>(cls_lc[cls]) + 8) * 2 * NSb * 64;
            const long long nbatch = ((implicit0 ? static_cast<long long>(njobs) : static_cast<long long>(cls_n[cls])) + 63) / 64;
            int* d_order = nullptr;
            if (!one_class && !implicit0) SL_TRY(upload("msa.ord0", order[cls].data(), order[cls].size(), &d_order, s));
            a.order = d_order; a.njobs = implicit0 ? static_cast<int>(njobs) : static_cast<int>(cls_n[cls]);
            a.skip_wide = implicit0 ? 1 : 0;
            a.tb_per_wave = per_wave;
            a.core_lo = BV_CORE0; a.core_hi = BV_CORE0 + BV_CORE_WORDS - 1;
            if (core_opt == 1) a.core_lo = a.core_hi = BV_CORE0 + 1;   // tests: one word only, most walks leave it and take the second run
            // move strings: one word of length + 2 bits per move, at the job's index
            a.moves_stride = static_cast<unsigned>((cls_lr[cls] + cls_lc[cls] + 15) / 16 + 2);
            SL_TRY(scratch("msa.moves", njobs * static_cast<size_t>(a.moves_stride), &a.moves));
            // Large lists: fill and walk as kernels of their own over chunks of batches, every batch of a chunk with its own
            // tile (up to 48 GB of records, a third of the free memory at most); otherwise both phases in one kernel.
            size_t free_b = 0, total_b = 0;
            SL_HIP(hipMemGetInfo(&free_b, &total_b));
            size_t have = 0;
            { auto it = c.ws.find("msa.tb0"); if (it != c.ws.end()) have = it->second.cap; }
            const size_t tile_gb = option(OPT_MSA_BITVECTOR_TILE_GB) > 0 ? static_cast<size_t>(option(OPT_MSA_BITVECTOR_TILE_GB)) : 48;
            const size_t tile_budget = std::min<size_t>(tile_gb << 30, std::max<size_t>(have, (free_b + have) / 3));
            const long long chunk = static_cast<long long>(tile_budget / (per_wave * 4));
            const bool split = option(OPT_MSA_BITVECTOR) != 2 && nbatch >= 4LL * c.num_cu && chunk >= 8LL * c.num_cu;
            void* d_tb;
            auto whole_records = [&](long long nb, const char* name) -> int {   // both phases in one kernel, one tile per resident wavefront
                long long grid = std::min<long long>(nb, static_cast<long long>(c.num_cu) * 16);
                const size_t budget = static_cast<size_t>(24) << 30;
                grid = std::min<long long>(grid, std::max<long long>(1, static_cast<long long>(budget / (per_full * 4))));
                SL_TRY(c.buffer(name, static_cast<size_t>(grid) * per_full * 4, &d_tb));
                a.tb = d_tb; a.tb_per_wave = per_full; a.batch0 = 0; a.batch1 = static_cast<int>(nb);
                if (NWb == 4) hipLaunchKernelGGL((k_msa_pairwise_bv<4, 0, 4>), dim3(static_cast<unsigned>(grid)), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((k_msa_pairwise_bv<8, 0, 8>), dim3(static_cast<unsigned>(grid)), dim3(64), 0, s, a);
                return 0;
            };
            long long redone = 0;
            // fill and walk kernels over chunks of `nb` batches; partial = only the core words of the records
            auto chunks = [&](long long nb, bool partial) -> int {
                const size_t pw = partial ? per_wave : per_full;
                const long long per = std::max<long long>(1, std::min(static_cast<long long>(tile_budget / (pw * 4)), nb));
                SL_TRY(c.buffer("msa.tb0", static_cast<size_t>(per) * pw * 4, &d_tb));
                a.tb = d_tb; a.tb_per_wave = pw;
                for (long long b0 = 0; b0 < nb; b0 += per) {
                    a.batch0 = static_cast<int>(b0); a.batch1 = static_cast<int>(std::min(nb, b0 + per));
                    const unsigned g = static_cast<unsigned>(a.batch1 - a.batch0);
                    if (NWb == 4) {
                        hipLaunchKernelGGL((k_msa_pairwise_bv<4, 1, 4>), dim3(g), dim3(64), 0, s, a);
                        hipLaunchKernelGGL((k_msa_pairwise_bv<4, 2, 4>), dim3(g), dim3(64), 0, s, a);
                    } else if (partial) {
                        hipLaunchKernelGGL((k_msa_pairwise_bv<8, 1, BV_CORE_WORDS>), dim3(g), dim3(64), 0, s, a);
                        hipLaunchKernelGGL((k_msa_pairwise_bv<8, 2, BV_CORE_WORDS>), dim3(g), dim3(64), 0, s, a);
                    } else {
                        hipLaunchKernelGGL((k_msa_pairwise_bv<8, 1, 8>), dim3(g), dim3(64), 0, s, a);
                        hipLaunchKernelGGL((k_msa_pairwise_bv<8, 2, 8>), dim3(g), dim3(64), 0, s, a);
                    }
                }
                return 0;
            };
            if (split) {
                SL_TRY(chunks(nbatch, core));
                if (core) {
                    // the pairs whose path left the kept words: collected, counted (one wait for the stream) and run again with whole records
                    int *d_list, *d_cnt;
                    SL_TRY(scratch("msa.redo", static_cast<size_t>(a.njobs) + 1, &d_list));
                    SL_TRY(scratch("msa.redon", 1, &d_cnt));
                    SL_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
                    hipLaunchKernelGGL(k_bv_collect_redo, dim3(static_cast<unsigned>((static_cast<size_t>(a.njobs) + 255) / 256)), dim3(256), 0, s, a, d_list, d_cnt);
                    int nredo = 0;
                    SL_HIP(hipMemcpyAsync(&nredo, d_cnt, sizeof nredo, hipMemcpyDeviceToHost, s));
                    SL_HIP(hipStreamSynchronize(s));
                    redone = nredo;
                    if (nredo > 0) {
                        const MsaArgs keep = a;
                        a.order = d_list; a.njobs = nredo; a.skip_wide = 0;
                        const long long nb2 = (static_cast<long long>(nredo) + 63) / 64;
                        if (nb2 >= 4LL * c.num_cu) SL_TRY(chunks(nb2, false)); else SL_TRY(whole_records(nb2, "msa.tb0"));
                        a.order = keep.order; a.njobs = keep.njobs; a.skip_wide = keep.skip_wide;
                    }
                }
            } else {
                SL_TRY(whole_records(nbatch, "msa.tb0"));
            }
            ctx().counts["msa_bitvector_redone"] += static_cast<double>(redone);
            ctx().counts["msa_bitvector_core"] = (split && core) ? 1 : 0;
            hipLaunchKernelGGL(k_msa_moves_expand, dim3(static_cast<unsigned>((static_cast<size_t>(a.njobs) + 3) / 4)), dim3(256), 0, s, a);
            a.skip_wide = 0;
            SL_HIP(hipGetLastError());
            ctx().counts["msa_pairs_bitvector"] += static_cast<double>(cls_n[cls]);
            // traceback records: 2 NW words per centre column and pair, written once
            ctx().counts["msa_bitvector_tile_bytes"] += summary->cols[cls] * 2 * ((split && core) ? BV_CORE_WORDS : NWb) * 4;
            ctx().counts["msa_bitvector_words"] = NWb;
            ctx().counts["msa_bitvector_split"] = split ? 1 : 0;
            continue;
        }
        const size_t steps = 2 * static_cast<size_t>(cls_lr[cls]) + cls_band[cls];
        const size_t word = (!packed && C == 16) ? 8 : 4;
        const size_t spw = C == 4 ? 2 : 1;
        const size_t per_wave = packed ? ((steps + 3) / 4 + 1) * (C / 4) * 64 : (steps / (2 * spw) + 2) * 64;
        size_t lds = MSA_WIN * 64 * word + static_cast<size_t>(cls_lc[cls]) + static_cast<size_t>(cls_lr[cls]) + 48;
        if (lds > 160 * 1024) return fail("sarlacc_amd: reads of %d bases do not fit the MSA kernel's LDS staging", std::max(cls_lr[cls], cls_lc[cls]));
        // many more single-wave workgroups than fit at once (a wave then aligns only a few pairs and the
        // hardware balances the load); their traceback tiles are the price, capped at 24 GB of HBM
        long long grid = std::min<long long>(static_cast<long long>(cls_n[cls]), static_cast<long long>(c.num_cu) * 128);
        const size_t budget = static_cast<size_t>(24) << 30;
        grid = std::min<long long>(grid, std::max<long long>(1, static_cast<long long>(budget / (per_wave * word))));
        void* d_tb; int* d_order = nullptr;
        const char* tb_name[3] = {"msa.tb0", "msa.tb1", "msa.tb2"};
        const char* ord_name[3] = {"msa.ord0", "msa.ord1", "msa.ord2"};
        SL_TRY(c.buffer(tb_name[cls], static_cast<size_t>(grid) * per_wave * word, &d_tb));
        if (!one_class) SL_TRY(upload(ord_name[cls], order[cls].data(), order[cls].size(), &d_order, s));
        a.order = d_order; a.njobs = static_cast<int>(cls_n[cls]);
        a.tb = d_tb; a.tb_per_wave = per_wave;
        if (out_mode == 0) SL_TRY(launch_class<0>(packed, C, a, static_cast<int>(grid), lds, s));
        else SL_TRY(launch_class<1>(packed, C, a, static_cast<int>(grid), lds, s));
    }
    return 0;
}

}  // namespace sarlacc
