// profile.hip -- the reference's alignment-profiling routines on gfx950:
//   find_homopolymers   /root/reference/src/homopolymer.cpp:87-134
//   match_homopolymers  /root/reference/src/homopolymer.cpp:141-209
//   find_errors         /root/reference/src/find_errors.cpp:9-121
// Byte scans of gapped alignment strings with variable-length results.  One thread walks one string (the runs of
// a string are a serial recurrence: a run ends at the next non-gap character that differs from it); every routine
// is a counting pass, an exclusive scan of the counts (rocPRIM, plumbing) and a writing pass, so the lists come
// out in the reference's order (alignment by alignment, left to right).  Not on the hot path: HBM-bound, each
// byte read twice, read through the thread's own cache lines.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <string>
#include <vector>

namespace sarlacc {

// A run: a maximal stretch of equal non-gap characters, gap characters inside or after it notwithstanding.
struct RunWalker {
    const uint8_t* p;
    long long len, start, next, gaps_before, gaps_upto;
    uint8_t base;
    __device__ void init(const uint8_t* s, long long n) {
        p = s; len = n; start = 0; gaps_before = 0; gaps_upto = 0; base = 0;
        long long i = 0;
        while (i < n && s[i] == '-') { ++i; ++gaps_upto; }
        next = i;
    }
    __device__ bool done() const { return next == len; }
    __device__ void advance() {
        start = next;
        gaps_before = gaps_upto;
        base = p[start];
        long long i = start + 1;
        while (i < len) {
            const uint8_t c = p[i];
            if (c != '-' && c != base) break;
            if (c == '-') ++gaps_upto;
            ++i;
        }
        next = i;
    }
    __device__ long long pos() const { return start - gaps_before; }               // start in the ungapped string
    __device__ long long length() const { return (next - gaps_upto) - pos(); }     // bases in the run
    __device__ long long start_with_gaps() const {
        long long q = start;
        while (q > 0 && p[q - 1] == '-') --q;
        return q;
    }
    __device__ long long end() const {                                              // without the trailing gaps
        long long q = next;
        while (q > start && p[q - 1] == '-') --q;
        return q;
    }
};

// WRITE = false: counts per string; WRITE = true: entries at the scanned offsets
template <bool WRITE>
__global__ void k_homopolymers(const uint8_t* seq, const int64_t* off, long long n, long long* count, int32_t* idx, int32_t* pos,
                               int32_t* size, uint8_t* base) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    RunWalker r;
    r.init(seq + off[i], off[i + 1] - off[i]);
    long long k = WRITE ? count[i] : 0;
    while (!r.done()) {
        r.advance();
        const long long L = r.length();
        if (L == 1) continue;
        if (WRITE) { idx[k] = static_cast<int32_t>(i); pos[k] = static_cast<int32_t>(r.pos() + 1); size[k] = static_cast<int32_t>(L); base[k] = r.base; }
        ++k;
    }
    if (!WRITE) count[i] = k;
}

template <bool WRITE>
__global__ void k_match_homopolymers(const uint8_t* ref, const int64_t* ref_off, const uint8_t* read, const int64_t* read_off,
                                     long long n, long long* count, int32_t* idx, int32_t* pos, int32_t* rlen) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint8_t* rf = ref + ref_off[i];
    const uint8_t* rd = read + read_off[i];
    RunWalker r;
    r.init(rf, ref_off[i + 1] - ref_off[i]);
    long long k = WRITE ? count[i] : 0;
    while (!r.done()) {
        r.advance();
        if (r.length() == 1) continue;
        if (WRITE) {
            // the longest run of the same base in the read that overlaps the reference run proper; the read is
            // examined over the reference run extended by the gaps on either side
            const long long far_left = r.start_with_gaps(), far_right = r.next, left = r.start, right = r.end();
            RunWalker q;
            q.init(rd + far_left, far_right - far_left);
            long long best = 0;
            while (!q.done()) {
                q.advance();
                if (right > q.start + far_left && left < q.end() + far_left) {
                    const long long L = q.length();
                    if (L > best && q.base == r.base) best = L;
                }
            }
            idx[k] = static_cast<int32_t>(i); pos[k] = static_cast<int32_t>(r.pos() + 1); rlen[k] = static_cast<int32_t>(best);
        }
        ++k;
    }
    if (!WRITE) count[i] = k;
}

// first_bad: minimum of (alignment << 34 | position << 2 | kind), kind 1 = reference longer than the first one,
// 2 = unknown read character -- the error the reference's loop would meet first
template <bool WRITE>
__global__ void k_find_errors(const uint8_t* ref, const int64_t* ref_off, const uint8_t* read, const int64_t* read_off, long long n,
                              long long standard_len, int* to_a, int* to_c, int* to_g, int* to_t, int* del, long long* count,
                              int32_t* ins_pos, int32_t* ins_len, unsigned long long* first_bad) {
    const long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint8_t* rf = ref + ref_off[i];
    const uint8_t* rd = read + read_off[i];
    const long long len = ref_off[i + 1] - ref_off[i];
    long long cur = 0, gaps = 0, k = WRITE ? count[i] : 0;
    while (cur < len) {
        if (rf[cur] != '-') {
            if (!WRITE) {
                const long long tp = cur - gaps;
                int kind = 0;
                if (tp >= standard_len) kind = 1;
                else {
                    switch (rd[cur]) {
                        case '-': atomicAdd(&del[tp], 1); break;
                        case 'A': atomicAdd(&to_a[tp], 1); break;
                        case 'C': atomicAdd(&to_c[tp], 1); break;
                        case 'G': atomicAdd(&to_g[tp], 1); break;
                        case 'T': atomicAdd(&to_t[tp], 1); break;
                        default: kind = 2; break;
                    }
                }
                if (kind) {
                    atomicMin(first_bad, (static_cast<unsigned long long>(i) << 34) | (static_cast<unsigned long long>(cur) << 2) | static_cast<unsigned>(kind));
                    break;   // the reference stops here; what follows in this string is never counted
                }
            }
            ++cur;
        } else {
            const long long first = cur;
            while (cur < len && rf[cur] == '-') { ++cur; ++gaps; }
            if (WRITE) { ins_pos[k] = static_cast<int32_t>(cur - gaps); ins_len[k] = static_cast<int32_t>(cur - first); }
            ++k;
        }
    }
    if (!WRITE) count[i] = k;
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

static inline unsigned blocks_for(long long n) { return static_cast<unsigned>(std::max<long long>(1, (n + 127) / 128)); }

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
    hipLaunchKernelGGL(k_homopolymers<false>, dim3(blocks_for(n)), dim3(128), 0, s, d_c, d_o, static_cast<long long>(n), d_cnt, nullptr, nullptr, nullptr, nullptr);
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
    hipLaunchKernelGGL(k_homopolymers<true>, dim3(blocks_for(n)), dim3(128), 0, s, d_c, d_o, static_cast<long long>(n), d_cnt, d_idx, d_pos, d_size, d_base);
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
    hipLaunchKernelGGL(k_match_homopolymers<false>, dim3(blocks_for(nref)), dim3(128), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(nref), d_cnt, nullptr, nullptr, nullptr);
    SL_HIP(hipGetLastError());
    long long total = 0;
    SL_TRY(scan_counts("mh", d_cnt, static_cast<size_t>(nref), &total, s));
    *count = total;
    if (total == 0 || cap < total || !idx) return 0;
    int32_t *d_idx, *d_pos, *d_len;
    SL_TRY(scratch("mh.idx", static_cast<size_t>(total), &d_idx));
    SL_TRY(scratch("mh.pos", static_cast<size_t>(total), &d_pos));
    SL_TRY(scratch("mh.len", static_cast<size_t>(total), &d_len));
    hipLaunchKernelGGL(k_match_homopolymers<true>, dim3(blocks_for(nref)), dim3(128), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(nref), d_cnt, d_idx, d_pos, d_len);
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
    hipLaunchKernelGGL(k_find_errors<false>, dim3(blocks_for(neval)), dim3(128), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(neval), static_cast<long long>(sl),
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
    hipLaunchKernelGGL(k_find_errors<true>, dim3(blocks_for(neval)), dim3(128), 0, s, d_r, d_ro, d_q, d_qo, static_cast<long long>(neval), static_cast<long long>(sl),
                       nullptr, nullptr, nullptr, nullptr, nullptr, d_cnt, d_ip, d_il, d_bad);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpy(ins_pos, d_ip, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    SL_HIP(hipMemcpy(ins_len, d_il, sizeof(int32_t) * static_cast<size_t>(total), hipMemcpyDeviceToHost));
    return 0;
}
}
