// profile.hip -- the reference's alignment-profiling routines on gfx950:
//   find_homopolymers   /root/reference/src/homopolymer.cpp:87-134
//   match_homopolymers  /root/reference/src/homopolymer.cpp:141-209
//   find_errors         /root/reference/src/find_errors.cpp:9-121
// Byte scans of gapped alignment strings with variable-length results.  ONE WAVEFRONT PER STRING, 64 characters per
// step, read side by side: what the reference's walker object carries from character to character becomes lane
// masks.  A run (a maximal stretch of equal non-gap characters, gaps inside it notwithstanding) STARTS at a non-gap
// character that differs from the non-gap character before it -- one ballot of the non-gap lanes, one lane permutation
// for "the non-gap character before me", one ballot of the starts; the run that ends where the next one starts has its
// length, position and neighbours as population counts and leading-one searches over those masks.  Only the run that is
// still open at the end of a step is carried into the next one (a handful of scalars).  Every routine is a counting
// pass, an exclusive scan of the counts (rocPRIM, plumbing) and a writing pass, so the lists come out in the reference's
// order (alignment by alignment, left to right).  Not on the hot path.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <string>
#include <vector>

namespace sarlacc {

constexpr int PF_WAVES = 4;   // strings per workgroup

__device__ __forceinline__ int pf_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long pf_below(int lane) { return (1ull << lane) - 1ull; }
// highest set bit of m below `lane`, -1 if none
__device__ __forceinline__ int pf_prev(unsigned long long m, int lane) {
    const unsigned long long b = m & pf_below(lane);
    return b ? 63 - __builtin_clzll(b) : -1;
}

// The runs of one string, as the wave meets their ends.  emit(lane_is_emitting, start index, index of the next run's
// first character (or the string length), position in the ungapped string, bases in the run, base, index after the
// previous non-gap character (the run's start extended over the gaps before it), index after the run's last base):
// called once per step and once for the run that is open at the end, with every lane taking part.
template <typename Emit>
__device__ __forceinline__ void pf_runs(const uint8_t* s, long long len, Emit emit) {
    const int lane = pf_lane();
    // the open run: its base (0: none yet), first index, ungapped position, bases so far, index after the non-gap character
    // before it; plus the ungapped characters and the index after the last non-gap character seen so far
    int cbase = 0;
    long long cstart = 0, cpos = 0, clen = 0, cfar = 0, ung = 0, lastng = 0;
    for (long long x0 = 0; x0 < len; x0 += 64) {
        const long long x = x0 + lane;
        const int c = x < len ? s[x] : '-';
        const bool ng = c != '-';
        const unsigned long long m_ng = __ballot(ng);
        const int pl = pf_prev(m_ng, lane);
        const int pc_lane = __shfl(c, pl < 0 ? 0 : pl);
        const int prevc = pl < 0 ? cbase : pc_lane;
        const bool st = ng && prevc != c;
        const unsigned long long m_st = __ballot(st);
        // a start at lane t closes the run before it: the one that started at the previous start of this step, or the open one
        const int u = pf_prev(m_st, lane);
        const long long u_pos = ung + __popcll(m_ng & pf_below(u < 0 ? 0 : u));
        const long long u_len = __popcll(m_ng & pf_below(lane) & ~pf_below(u < 0 ? 0 : u));
        const int u_base = __shfl(c, u < 0 ? 0 : u);
        const int u_pl = __shfl(pl, u < 0 ? 0 : u);                      // the non-gap lane before the run's first base
        const long long u_far = u_pl < 0 ? lastng : x0 + u_pl + 1;
        const bool from_open = u < 0;
        const bool fire = st && (!from_open || cbase != 0);
        const long long e_start = from_open ? cstart : x0 + u;
        const long long e_pos = from_open ? cpos : u_pos;
        const long long e_len = from_open ? clen + __popcll(m_ng & pf_below(lane)) : u_len;
        const int e_base = from_open ? cbase : u_base;
        const long long e_far = from_open ? cfar : u_far;
        const long long e_right = pl < 0 ? lastng : x0 + pl + 1;          // index after the last base before this start
        emit(fire, e_start, x, e_pos, e_len, e_base, e_far, e_right);
        // what stays open
        if (m_st) {
            const int t = 63 - __builtin_clzll(m_st);
            const int t_pl = __shfl(pl, t);
            cbase = __shfl(c, t);
            cstart = x0 + t;
            cpos = ung + __popcll(m_ng & pf_below(t));
            clen = __popcll(m_ng & ~pf_below(t));
            cfar = t_pl < 0 ? lastng : x0 + t_pl + 1;
        } else {
            clen += __popcll(m_ng);
        }
        ung += __popcll(m_ng);
        if (m_ng) lastng = x0 + (63 - __builtin_clzll(m_ng)) + 1;
    }
    emit(lane == 0 && cbase != 0, cstart, len, cpos, clen, cbase, cfar, lastng);
}

// WRITE = false: counts per string; WRITE = true: entries at the scanned offsets
template <bool WRITE>
__global__ void __launch_bounds__(64 * PF_WAVES) k_homopolymers(const uint8_t* seq, const int64_t* off, long long n, long long* count, int32_t* idx,
                                                                int32_t* pos, int32_t* size, uint8_t* base) {
    const long long i = blockIdx.x * static_cast<long long>(PF_WAVES) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = pf_lane();
    long long k = WRITE ? count[i] : 0;
    pf_runs(seq + off[i], off[i + 1] - off[i], [&](bool fire, long long, long long, long long rpos, long long rlen, int rbase, long long, long long) {
        const bool out = fire && rlen > 1;
        const unsigned long long m = __ballot(out);
        if (WRITE && out) {
            const long long o = k + __popcll(m & pf_below(lane));
            idx[o] = static_cast<int32_t>(i); pos[o] = static_cast<int32_t>(rpos + 1); size[o] = static_cast<int32_t>(rlen); base[o] = static_cast<uint8_t>(rbase);
        }
        k += __popcll(m);
    });
    if (!WRITE && lane == 0) count[i] = k;
}

template <bool WRITE>
__global__ void __launch_bounds__(64 * PF_WAVES) k_match_homopolymers(const uint8_t* ref, const int64_t* ref_off, const uint8_t* read, const int64_t* read_off,
                                                                      long long n, long long* count, int32_t* idx, int32_t* pos, int32_t* rlen) {
    const long long i = blockIdx.x * static_cast<long long>(PF_WAVES) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = pf_lane();
    const uint8_t* const rd = read + read_off[i];
    long long k = WRITE ? count[i] : 0;
    pf_runs(ref + ref_off[i], ref_off[i + 1] - ref_off[i], [&](bool fire, long long left, long long far_right, long long rpos, long long rl, int rbase, long long far_left, long long right) {
        const bool out = fire && rl > 1;
        const unsigned long long m = __ballot(out);
        if (WRITE && out) {
            // the longest run of the same base in the read that overlaps the reference run proper [left, right); the read is
            // examined over the reference run extended by the gaps on either side, [far_left, far_right) -- a few characters,
            // walked by the lane that holds the run
            long long best = 0, qs = 0, qe = 0, ql = 0;
            int qb = 0;
            for (long long x = far_left; x <= far_right; ++x) {
                const int c = x < far_right ? rd[x] : 0;     // 0 closes the last run
                if (c == '-') continue;
                if (c != qb) {
                    if (qb != 0 && qb == rbase && right > qs && left < qe && ql > best) best = ql;
                    qb = c; qs = x; ql = 0;
                }
                ++ql; qe = x + 1;
            }
            const long long o = k + __popcll(m & pf_below(lane));
            idx[o] = static_cast<int32_t>(i); pos[o] = static_cast<int32_t>(rpos + 1); rlen[o] = static_cast<int32_t>(best);
        }
        k += __popcll(m);
    });
    if (!WRITE && lane == 0) count[i] = k;
}

// first_bad: minimum of (alignment << 34 | position << 2 | kind), kind 1 = reference longer than the first one,
// 2 = unknown read character -- the error the reference's loop would meet first
template <bool WRITE>
__global__ void __launch_bounds__(64 * PF_WAVES) k_find_errors(const uint8_t* ref, const int64_t* ref_off, const uint8_t* read, const int64_t* read_off, long long n,
                                                               long long standard_len, int* to_a, int* to_c, int* to_g, int* to_t, int* del, long long* count,
                                                               int32_t* ins_pos, int32_t* ins_len, unsigned long long* first_bad) {
    const long long i = blockIdx.x * static_cast<long long>(PF_WAVES) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = pf_lane();
    const uint8_t* const rf = ref + ref_off[i];
    const uint8_t* const rd = read + read_off[i];
    const long long len = ref_off[i + 1] - ref_off[i];
    long long ung = 0, gaprun = 0, k = WRITE ? count[i] : 0;   // reference bases so far; gaps at the end of what was seen
    for (long long x0 = 0; x0 < len; x0 += 64) {
        const long long x = x0 + lane;
        const int rc = x < len ? rf[x] : 'A';      // (past the end: "a base", closes nothing by itself -- see `in`)
        const bool in = x < len;
        const bool ng = in && rc != '-';
        const unsigned long long m_ng = __ballot(ng), m_in = __ballot(in);
        const long long tp = ung + __popcll(m_ng & pf_below(lane));
        if (!WRITE && ng) {
            // (the reference stops at its first error; the call fails then, so what is counted beyond it does not matter)
            int kind = 0;
            if (tp >= standard_len) kind = 1;
            else {
                switch (rd[x]) {
                    case '-': atomicAdd(&del[tp], 1); break;
                    case 'A': atomicAdd(&to_a[tp], 1); break;
                    case 'C': atomicAdd(&to_c[tp], 1); break;
                    case 'G': atomicAdd(&to_g[tp], 1); break;
                    case 'T': atomicAdd(&to_t[tp], 1); break;
                    default: kind = 2; break;
                }
            }
            if (kind) atomicMin(first_bad, (static_cast<unsigned long long>(i) << 34) | (static_cast<unsigned long long>(x) << 2) | static_cast<unsigned>(kind));
        }
        // a run of gaps in the reference (an insertion) is listed where it ends: at the next reference base
        const int pl = pf_prev(m_ng, lane);
        const long long glen = pl < 0 ? gaprun + lane : lane - pl - 1;
        const bool out = ng && glen > 0;
        const unsigned long long m = __ballot(out);
        if (WRITE && out) {
            const long long o = k + __popcll(m & pf_below(lane));
            ins_pos[o] = static_cast<int32_t>(tp); ins_len[o] = static_cast<int32_t>(glen);
        }
        k += __popcll(m);
        const int nin = __popcll(m_in);
        if (m_ng) gaprun = nin - (63 - __builtin_clzll(m_ng)) - 1; else gaprun += nin;
        ung += __popcll(m_ng);
    }
    if (gaprun > 0) {   // gaps at the very end
        if (WRITE && lane == 0) { ins_pos[k] = static_cast<int32_t>(ung); ins_len[k] = static_cast<int32_t>(gaprun); }
        ++k;
    }
    if (!WRITE && lane == 0) count[i] = k;
}

static int scan_counts(const char* tag, long long* d_count, size_t n, long long* total, hipStream_t s) {
    // exclusive scan in place over n + 1 entries (the last one, zeroed, receives the total)
    size_t tmp = 0;
    SL_HIP(hipMemsetAsync(d_count + n, 0, sizeof(long long), s));
    SL_HIP(rocprim::exclusive_scan(nullptr, tmp, d_count, d_count, 0ll, n + 1, rocprim::plus<long long>(), s));
    void* d_tmp;
    SL_TRY(ctx().buffer((std::string(tag) + ".scantmp").c_str(), tmp ? tmp : 16, &d_tmp));
    SL_HIP(rocprim::exclusive_scan(d_tmp, tmp, d_count, d_count, 0ll, n + 1, rocprim::plus<long long>(), s));
    SL_HIP(hipMemcpyAsync(total, d_count + n, sizeof(long long), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return 0;
}

static int upload_set(const std::string& tag, const char* chars, const int64_t* off, int64_t n, uint8_t** d_chars, int64_t** d_off,
                      hipStream_t s) {
    const int64_t base = n ? off[0] : 0;
    const int64_t total = n ? off[n] - base : 0;
    std::vector<int64_t> rel(static_cast<size_t>(n) + 1);
    for (int64_t i = 0; i <= n; ++i) rel[i] = (n ? off[i] : 0) - base;
    SL_TRY(upload((tag + ".chars").c_str(), reinterpret_cast<const uint8_t*>(chars) + base, static_cast<size_t>(total), d_chars, s));
    SL_TRY(upload((tag + ".off").c_str(), rel.data(), rel.size(), d_off, s));
    return 0;
}

static inline unsigned blocks_for(long long n) { return static_cast<unsigned>(std::max<long long>(1, (n + PF_WAVES - 1) / PF_WAVES)); }

}  // namespace sarlacc

using namespace sarlacc;

extern "C" {

int sarlacc_find_homopolymers(const char* seq, const int64_t* off, int64_t n, int32_t* idx, int32_t* pos, int32_t* size, char* base,
                              int64_t cap, int64_t* count) {
    if (n < 0) return fail("sarlacc_amd: negative number of sequences");
    *count = 0;
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t* d_c; int64_t* d_o; long long* d_cnt;
    SL_TRY(upload_set("hp", seq, off, n, &d_c, &d_o, s));
    SL_TRY(scratch("hp.count", static_cast<size_t>(n) + 1, &d_cnt));
    hipLaunchKernelGGL(k_homopolymers<false>, dim3(blocks_for(n)), dim3(64 * PF_WAVES), 0, s, d_c, d_o, static_cast<long long>(n), d_cnt, nullptr, nullptr, nullptr, nullptr);
    SL_HIP(hipGetLastError());
    long long total = 0;
    SL_TRY(scan_counts("hp", d_cnt, static_cast<size_t>(n), &total, s));
    *count = total;
    if (total == 0 || cap < total || !idx) return 0;   // sizing call
    int32_t *d_idx, *d_pos, *d_size; uint8_t* d_base;
    SL_TRY(scratch("hp.idx", static_cast<size_t>(total), &d_idx));
    SL_TRY(scratch("hp.pos", static_cast<size_t>(total), &d_pos));
    SL_TRY(scratch("hp.size", static_cast<size_t>(total), &d_size));
    SL_TRY(scratch("hp.base", static_cast<size_t>(total), &d_base));
    hipLaunchKernelGGL(k_homopolymers<true>, dim3(blocks_for(n)), dim3(64 * PF_WAVES), 0, s, d_c, d_o, static_cast<long long>(n), d_cnt, d_idx, d_pos, d_size, d_base);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpy(idx, d_idx, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(pos, d_pos, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(size, d_size, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(base, d_base, static_cast<size_t>(total), hipMemcpyDeviceToHost));
    return 0;
}

int sarlacc_match_homopolymers(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                               int64_t nread, int32_t* idx, int32_t* pos, int32_t* rlen, int64_t cap, int64_t* count) {
    if (nref < 0 || nread < 0) return fail("sarlacc_amd: negative number of alignments");
    *count = 0;
    if (nref != nread) return fail("lengths of alignment vectors should match up");
    for (int64_t i = 0; i < nref; ++i)
        if (ref_off[i + 1] - ref_off[i] != read_off[i + 1] - read_off[i]) return fail("read and reference alignment strings should have equal length");
    if (nref == 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t *d_r, *d_q; int64_t *d_ro, *d_qo; long long* d_cnt;
    SL_TRY(upload_set("mh.ref", ref, ref_off, nref, &d_r, &d_ro, s));
    SL_TRY(upload_set("mh.read", read, read_off, nread, &d_q, &d_qo, s));
    SL_TRY(scratch("mh.count", static_cast<size_t>(nref) + 1, &d_cnt));
    hipLaunchKernelGGL(k_match_homopolymers<false>, dim3(blocks_for(nref)), dim3(64 * PF_WAVES), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(nref), d_cnt, nullptr, nullptr, nullptr);
    SL_HIP(hipGetLastError());
    long long total = 0;
    SL_TRY(scan_counts("mh", d_cnt, static_cast<size_t>(nref), &total, s));
    *count = total;
    if (total == 0 || cap < total || !idx) return 0;
    int32_t *d_idx, *d_pos, *d_len;
    SL_TRY(scratch("mh.idx", static_cast<size_t>(total), &d_idx));
    SL_TRY(scratch("mh.pos", static_cast<size_t>(total), &d_pos));
    SL_TRY(scratch("mh.len", static_cast<size_t>(total), &d_len));
    hipLaunchKernelGGL(k_match_homopolymers<true>, dim3(blocks_for(nref)), dim3(64 * PF_WAVES), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(nref), d_cnt, d_idx, d_pos, d_len);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpy(idx, d_idx, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(pos, d_pos, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(rlen, d_len, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    return 0;
}

int sarlacc_find_errors(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off, int64_t nread,
                        int64_t* standard_len, char* bases, int32_t* to_a, int32_t* to_c, int32_t* to_g, int32_t* to_t,
                        int32_t* deletions, int64_t cap_bases, int32_t* ins_pos, int32_t* ins_len, int64_t cap_ins, int64_t* nins) {
    if (nref < 0 || nread < 0) return fail("sarlacc_amd: negative number of alignments");
    *standard_len = 0; *nins = 0;
    if (nref != nread) return fail("lengths of alignment vectors should match up");
    // the reference fixes the base sequence from the first alignment (src/find_errors.cpp:19-40)
    int64_t sl = 0;
    if (nref) {
        for (int64_t x = ref_off[0]; x < ref_off[1]; ++x)
            if (ref[x] != '-') { if (sl < cap_bases && bases) bases[sl] = ref[x]; ++sl; }
    }
    *standard_len = sl;
    if (sl > cap_bases || (sl && !to_a)) return 0;   // sizing call: *standard_len says what is needed
    // first alignment, in order, whose two strings differ in length: the reference meets it before scanning that pair
    int64_t bad_len = -1;
    for (int64_t i = 0; i < nref && bad_len < 0; ++i)
        if (ref_off[i + 1] - ref_off[i] != read_off[i + 1] - read_off[i]) bad_len = i;
    const int64_t neval = bad_len >= 0 ? bad_len : nref;
    if (neval == 0) {
        if (bad_len >= 0) return fail("read and reference alignment strings should have equal length");
        return 0;
    }
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    uint8_t *d_r, *d_q; int64_t *d_ro, *d_qo; long long* d_cnt; int* d_cols; unsigned long long* d_bad;
    SL_TRY(upload_set("fe.ref", ref, ref_off, neval, &d_r, &d_ro, s));
    SL_TRY(upload_set("fe.read", read, read_off, neval, &d_q, &d_qo, s));
    SL_TRY(scratch("fe.count", static_cast<size_t>(neval) + 1, &d_cnt));
    SL_TRY(scratch("fe.cols", 5 * static_cast<size_t>(sl) + 1, &d_cols));
    SL_TRY(scratch("fe.bad", 1, &d_bad));
    SL_HIP(hipMemsetAsync(d_cols, 0, sizeof(int) * (5 * static_cast<size_t>(sl) + 1), s));
    SL_HIP(hipMemsetAsync(d_bad, 0xff, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_find_errors<false>, dim3(blocks_for(neval)), dim3(64 * PF_WAVES), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(neval), static_cast<long long>(sl),
                       d_cols, d_cols + sl, d_cols + 2 * sl, d_cols + 3 * sl, d_cols + 4 * sl, d_cnt, nullptr, nullptr, d_bad);
    SL_HIP(hipGetLastError());
    long long total = 0;
    SL_TRY(scan_counts("fe", d_cnt, static_cast<size_t>(neval), &total, s));
    unsigned long long bad = ~0ull;
    SL_HIP(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    if (bad != ~0ull) {
        const int64_t i = static_cast<int64_t>(bad >> 34), cur = static_cast<int64_t>((bad >> 2) & 0xffffffffull);
        if ((bad & 3ull) == 1ull) return fail("reference sequence should be the same for all alignments");
        return fail("unknown character '%c' in alignment string", read[read_off[i] + cur]);
    }
    if (bad_len >= 0) return fail("read and reference alignment strings should have equal length");
    *nins = total;
    SL_HIP(hipMemcpy(to_a, d_cols, sizeof(int) * static_cast<size_t>(sl), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(to_c, d_cols + sl, sizeof(int) * static_cast<size_t>(sl), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(to_g, d_cols + 2 * sl, sizeof(int) * static_cast<size_t>(sl), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(to_t, d_cols + 3 * sl, sizeof(int) * static_cast<size_t>(sl), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(deletions, d_cols + 4 * sl, sizeof(int) * static_cast<size_t>(sl), hipMemcpyDeviceToHost));
    if (total == 0 || cap_ins < total || !ins_pos) return 0;   // (insertion lists: sizing)
    int32_t *d_ip, *d_il;
    SL_TRY(scratch("fe.ipos", static_cast<size_t>(total), &d_ip));
    SL_TRY(scratch("fe.ilen", static_cast<size_t>(total), &d_il));
    hipLaunchKernelGGL(k_find_errors<true>, dim3(blocks_for(neval)), dim3(64 * PF_WAVES), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(neval), static_cast<long long>(sl),
                       nullptr, nullptr, nullptr, nullptr, nullptr, d_cnt, d_ip, d_il, d_bad);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpy(ins_pos, d_ip, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(ins_len, d_il, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    return 0;
}
}
