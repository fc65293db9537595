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
//   msa_pairwise.hip  one wavefront per (read, centre) pair: banded Gotoh scheduled along
//                   anti-diagonals of the band, packed 16-bit scores (32-bit fallback).
//   k_msa_width     per group: column budget (max insertions before each centre base).
//   k_msa_write     per group: emits the gapped rows.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "msa_common.hpp"

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <limits>
#include <vector>

namespace sarlacc {

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
    // rows as vote codes instead of characters (CodeSpec, common.hpp): out16 != nullptr
    uint16_t* out16;
    const uint8_t* qual;       // laid out like seq
    int qoffset, navail;
    int* bad;
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

__device__ __forceinline__ int wave_scan(int x) {   // inclusive prefix sum over the wavefront (DPP)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return x;
}

constexpr int WRITE_THREADS = 256;
// Character tables of the row writer: 0 aligned rows as characters (upper-cased bases, N for the rest),
// 1 aligned rows as vote slots, 2 verbatim rows as vote slots, 3 verbatim rows as characters (identity).
struct WriteMap {
    uint8_t t[4][256];
    constexpr WriteMap() : t{} {
        for (int c = 0; c < 256; ++c) {
            const int u = (c >= 'a' && c <= 'z') ? c - 32 : c;
            const bool base = u == 'A' || u == 'C' || u == 'G' || u == 'T';
            const int up = base ? u : 'N';
            t[0][c] = static_cast<uint8_t>(up);
            t[1][c] = static_cast<uint8_t>(up == 'A' ? 3 : up == 'C' ? 2 : up == 'T' ? 1 : up == 'G' ? 0 : 5);
            t[2][c] = static_cast<uint8_t>(c == 'N' ? 5 : c == 'A' ? 3 : c == 'C' ? 2 : c == 'T' ? 1 : c == 'G' ? 0 : 4);
            t[3][c] = static_cast<uint8_t>(c);
        }
    }
};
__device__ const WriteMap k_write_map{};

// One block per (group, read): columns come from prefix sums over the centre positions.  CODES: the cells are
// 16-bit vote codes (msa_common.hpp) instead of characters.  The per-character work (upper-casing, N for anything
// that is not a base, the vote slot) goes through a 256-entry LDS table built once per block.
template <bool CODES>
__global__ void k_msa_write(MergeArgs A, const long long* row_group, const int* row_pos, long long nrows) {
    const long long row = blockIdx.x;
    if (row >= nrows) return;
    const long long g = row_group[row];
    const int r = row_pos[row];
    const MsaGroup G = A.groups[g];
    const long long id = A.members[G.read0 + r] - 1;
    const uint8_t* src = A.seq + A.seq_off[id];
    const int W = A.width[g];
    uint8_t* dst = CODES ? nullptr : A.out + A.out_off[g] + static_cast<long long>(r) * W;
    uint16_t* dst16 = CODES ? A.out16 + A.out_off[g] + static_cast<long long>(r) * W : nullptr;
    const uint8_t* ql = CODES ? A.qual + A.seq_off[id] : nullptr;
    const unsigned zero = code_zero_index(A.navail);
    const uint16_t gapcode = static_cast<uint16_t>(CODE_GAPBIT | zero);
    const bool verbatim = G.nreads == 1;   // src/quick_msa.cpp:46-50
    // s_map[c]: characters -> the row's character; codes -> the vote slot, 5 = N (the all-zero table row)
    __shared__ uint8_t s_map[256];
    s_map[threadIdx.x] = k_write_map.t[CODES ? (verbatim ? 2 : 1) : (verbatim ? 3 : 0)][threadIdx.x];
    __syncthreads();
    bool badq = false;
    // cell `c` of the row holds read position `rp`, or a gap
    auto put = [&](unsigned c, unsigned rp) {   // unsigned: 32-bit offsets from the scalar row bases
        const unsigned v = s_map[src[rp]];
        if (CODES) {
            int qi = static_cast<int>(static_cast<signed char>(ql[rp])) - A.qoffset;
            badq |= qi < 0 && v != 5;
            qi = min(max(qi, 0), A.navail - 1);
            dst16[c] = static_cast<uint16_t>(v == 5 ? zero : static_cast<unsigned>(qi) * CODE_STRIP + v);
        } else {
            dst[c] = static_cast<uint8_t>(v);
        }
    };
    auto gap = [&](unsigned c) {
        if (CODES) dst16[c] = gapcode;
        else dst[c] = '-';
    };
    if (verbatim) {
        for (int p = threadIdx.x; p < W; p += blockDim.x) put(p, p);
        if (badq) atomicMin(A.bad, static_cast<int>(row));
        return;
    }
    const long long jb = job_of(G, r);
    const bool other = jb >= 0;            // not the centre itself: has an alignment to the centre
    const uint16_t* mi = A.maxins + A.mi_off[g];
    const uint16_t* ins = other ? A.ins + A.jobs[jb].out_off : mi;   // (centre: loaded and ignored)
    const uint8_t* aln = other ? A.aln + A.jobs[jb].out_off : reinterpret_cast<const uint8_t*>(mi);
    // Tiles of WRITE_THREADS centre positions, one per thread: a block-wide exclusive scan of (columns taken,
    // read bases taken) gives every position its first cell, so neighbouring threads write neighbouring cells.
    const int lc = G.lc;
    const int wv = threadIdx.x >> 6;
    constexpr int NWV = WRITE_THREADS / 64;
    __shared__ int s_wc[2][NWV], s_wr[2][NWV];
    int base_col = 0, base_rp = 0, flip = 0;
    // clamped addresses: the loads stay unconditional, and the next tile's are issued before this tile's barrier
    int kraw, mraw, araw;
    auto fetch = [&](int p) {
        const unsigned pc = static_cast<unsigned>(min(p, lc));
        kraw = ins[pc]; mraw = mi[pc]; araw = aln[pc];   // aln has lc + 1 entries too (the last is unused)
    };
    fetch(threadIdx.x);
    for (int tb = 0; tb <= lc; tb += WRITE_THREADS, flip ^= 1) {
        const int p = tb + static_cast<int>(threadIdx.x);
        const bool live = p <= lc, cell = p < lc;
        const int kcur = kraw, mcur = mraw, acur = araw;
        fetch(p + WRITE_THREADS);
        const int k = live && other ? kcur : 0;
        const int m = live ? mcur : 0;
        const bool matched = cell && (other ? acur != 0 : true);
        const int vcol = m + (cell ? 1 : 0), vrp = k + (matched ? 1 : 0);
        int icol = wave_scan(vcol), irp = wave_scan(vrp);
        asm volatile("" : "+v"(icol), "+v"(irp));   // keeps the scans as fused DPP adds (no re-derivation from their steps)
        if ((threadIdx.x & 63) == 63) { s_wc[flip][wv] = icol; s_wr[flip][wv] = irp; }
        __syncthreads();   // alternating buffers: one barrier per tile
        int col = base_col + icol - vcol, rp = base_rp + irp - vrp;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const int a = s_wc[flip][w], b = s_wr[flip][w];
            if (w < wv) { col += a; rp += b; }
            base_col += a; base_rp += b;
        }
        if (m) {   // rare: insertion columns before this centre position
            for (int x = 0; x < k; ++x) { put(col, rp); ++col; ++rp; }
            for (int x = k; x < m; ++x) { gap(col); ++col; }
        }
        if (matched) put(col, rp);
        else if (cell) gap(col);
    }
    if (badq) atomicMin(A.bad, static_cast<int>(row));
}

// ---------------------------------------------------------------------------
// The whole MSA stage with the gapped rows left in HBM (res->d_out): shared by sarlacc_quick_msa,
// which copies them back, and sarlacc_msa_consensus, which votes on them where they are.
// out_cap: < 0 no limit; otherwise the rows are only written when they fit (sizing protocol of
// sarlacc_quick_msa: widths and offsets are always filled in).
int msa1_run(const int64_t* grp_off, const int32_t* grp, int64_t ngroups, const char* seq, const int64_t* seq_off,
            int64_t nseq, double match, double mismatch, double gap_extension, double gap_opening, int bandwidth,
            bool want_rows, int64_t out_cap, MsaResult* res, const std::function<int()>* overlap,
            const uint8_t* d_seq_resident, bool accumulate) {
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
                max_band = std::max(max_band, msa_pair_band(bandwidth, J.lr, G.lc));
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
        double cells = 0;
        for (const MsaJob& J : jobs) cells += static_cast<double>(J.lr) * msa_pair_band(bandwidth, J.lr, J.lc);
        // accumulate: part of a spec v2 call (msa2.hip), which has reset the timers and counters itself
        c.counts["msa_pairs"] = (accumulate ? c.counts["msa_pairs"] : 0.0) + static_cast<double>(jobs.size());
        c.counts["msa_cells"] = (accumulate ? c.counts["msa_cells"] : 0.0) + cells;
        SL_HIP(hipEventRecord(c.ev_start, s));
        if (!accumulate) c.stage_reset("msa_pairwise");
        SL_TRY(c.stage_begin("msa_pairwise", s));
        SL_TRY(msa_pairwise_launch(jobs, d_jobs, d_seq, match, mismatch, gap_extension, gap_opening, bandwidth, 0, d_ins, d_aln,
                                   nullptr, nullptr, s));
        SL_HIP(hipEventRecord(c.ev_stop, s));
        SL_TRY(c.stage_end("msa_pairwise", s));
        c.timed = true;
    }
    if (overlap) SL_TRY((*overlap)());
    MergeArgs m{};
    m.seq = d_seq; m.seq_off = d_soff; m.members = d_mem; m.groups = d_groups; m.jobs = d_jobs; m.ngroups = ngroups;
    m.ins = d_ins; m.aln = d_aln; m.maxins = d_maxins; m.mi_off = d_mioff; m.width = d_width;
    if (!accumulate) c.stage_reset("msa_merge");
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
    m.out_off = d_ooff;
    if (res->code.want) {
        // rows as vote codes: the qualities may still be on their way (upload hook on a stream of its own)
        uint16_t* d_out16;
        SL_TRY(scratch("msa.out16", static_cast<size_t>(ooff[ngroups]) + 4, &d_out16));   // (+4: the vote kernel reads four codes at a time)
        if (res->code.ready) SL_HIP(hipStreamWaitEvent(s, res->code.ready, 0));
        m.out = nullptr; m.out16 = d_out16; m.qual = *res->code.qual; m.qoffset = res->code.qoffset; m.navail = res->code.navail;
        m.bad = res->code.d_bad;
        res->d_codes = d_out16;
        d_out = nullptr;
    } else {
        SL_TRY(scratch("msa.out", static_cast<size_t>(ooff[ngroups]), &d_out));
        m.out = d_out; m.out16 = nullptr;
    }
    const long long nrows = static_cast<long long>(row_group.size());
    if (m.out16) hipLaunchKernelGGL(k_msa_write<true>, dim3(static_cast<unsigned>(nrows)), dim3(WRITE_THREADS), 0, s, m, d_rg, d_rp, nrows);
    else hipLaunchKernelGGL(k_msa_write<false>, dim3(static_cast<unsigned>(nrows)), dim3(WRITE_THREADS), 0, s, m, d_rg, d_rp, nrows);
    SL_HIP(hipGetLastError());
    SL_TRY(c.stage_end("msa_merge", s));
    res->d_out = d_out;
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
