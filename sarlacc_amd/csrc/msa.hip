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
    const bool codes = A.out16 != nullptr;
    uint8_t* dst = codes ? nullptr : A.out + A.out_off[g] + static_cast<long long>(r) * W;
    uint16_t* dst16 = codes ? A.out16 + A.out_off[g] + static_cast<long long>(r) * W : nullptr;
    const uint8_t* ql = codes ? A.qual + A.seq_off[id] : nullptr;
    const uint16_t gapcode = static_cast<uint16_t>(CODE_GAPBIT | code_zero_index(A.navail));
    bool badq = false;
    // cell `c` of the row holds read position `rp` (character ch), or a gap
#define MSA_PUT(c, ch, rp) { if (codes) dst16[c] = vote_code(ch, ql[rp], A.qoffset, A.navail, badq); else dst[c] = ch; }
#define MSA_GAP(c) { if (codes) dst16[c] = gapcode; else dst[c] = '-'; }
    if (G.nreads == 1) {  // verbatim (src/quick_msa.cpp:46-50)
        for (int p = threadIdx.x; p < W; p += blockDim.x) MSA_PUT(p, src[p], p)
        if (badq) atomicMin(A.bad, static_cast<int>(row));
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
        for (int x = 0; x < k; ++x) { MSA_PUT(col, "ACGTN"[dna5_code(src[rp])], rp) ++col; ++rp; }
        for (int x = k; x < m; ++x) { MSA_GAP(col) ++col; }
        if (p < lc) {
            const bool matched = aln ? aln[p] != 0 : true;
            if (matched) { MSA_PUT(col, "ACGTN"[dna5_code(src[rp])], rp) ++rp; }
            else MSA_GAP(col)
            ++col;
        }
    }
    if (badq) atomicMin(A.bad, static_cast<int>(row));
#undef MSA_PUT
#undef MSA_GAP
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
    const double tm2 = now();
    if (overlap) SL_TRY((*overlap)());
    const double tm3 = now();
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
