// mask.hip -- quality masking of bases (umiGroup / expectedDist pre-step).
// Replaces mask_bad_bases (/root/reference/src/mask_bad_bases.cpp:10-52): a base
// becomes 'N' where the error probability of its quality character is strictly
// greater than the threshold.  One thread per base, error table in LDS.
#include "common.hpp"

#include "../../include/sarlacc_amd.h"

#include <limits>
#include <algorithm>
#include <vector>

namespace sarlacc {

__global__ void k_mask(const uint8_t* seq, const uint8_t* qual, long long total, const double* errors, int navail,
                       int qoffset, double threshold, uint8_t* out, unsigned long long* first_bad) {
    extern __shared__ double s_err[];
    for (int x = threadIdx.x; x < navail; x += blockDim.x) s_err[x] = errors[x];
    __syncthreads();
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total; i += stride) {
        int qi = static_cast<int>(static_cast<signed char>(qual[i])) - qoffset;
        if (qi < 0) { atomicMin(first_bad, static_cast<unsigned long long>(i)); qi = 0; }
        // the reference's bound test is 'i > size' (src/quality_encoding.cpp:43, UB at
        // i == size); clamp to the last entry as every reachable Phred input does
        if (qi >= navail) qi = navail - 1;
        out[i] = (s_err[qi] > threshold) ? static_cast<uint8_t>('N') : seq[i];
    }
}

// unmask_alignment: one wavefront per row, 64 columns per step; the ungapped position of a
// column is a running count of non-gap characters (ballot + popcount).  status: 1 = a masked
// base lies past the end of the original, 2 = lengths differ.
__global__ void __launch_bounds__(64) k_unmask(const uint8_t* aln, const int64_t* aln_off, const uint8_t* orig,
                                               const int64_t* orig_off, long long nrows, uint8_t* out, int8_t* status) {
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (long long r = blockIdx.x; r < nrows; r += gridDim.x) {
        const long long a0 = aln_off[r], W = aln_off[r + 1] - a0;
        const long long o0 = orig_off[r], L = orig_off[r + 1] - o0;
        long long nominal = 0;
        bool too_long = false;
        for (long long c0 = 0; c0 < W; c0 += 64) {
            const long long col = c0 + lane;
            uint8_t c = col < W ? aln[a0 + col] : static_cast<uint8_t>('-');
            const unsigned long long nongap = __ballot(c != '-');
            const long long p = nominal + __popcll(nongap & below);
            if (c == 'N' || c == 'n') {
                if (p >= L) too_long = true;
                else c = orig[o0 + p];
            }
            if (col < W) out[a0 + col] = c;
            nominal += __popcll(nongap);
        }
        const bool any_long = __ballot(too_long) != 0;
        if (lane == 0) status[r] = any_long ? 1 : (nominal != L ? 2 : 0);
    }
}

}  // namespace sarlacc

using namespace sarlacc;

extern "C" int sarlacc_unmask_alignment(const char* aln, const int64_t* aln_off, int64_t naln, const char* orig,
                                        const int64_t* orig_off, int64_t norig, char* out) {
    if (naln < 0 || norig < 0) return fail("sarlacc_amd: negative number of sequences");
    if (naln != norig) return fail("alignment and original sequences should have the same number of entries");
    for (int64_t i = 1; i < naln; ++i)   // check_alignment_width (src/DNA_input.cpp:90-104)
        if (aln_off[i + 1] - aln_off[i] != aln_off[1] - aln_off[0]) return fail("alignment strings should have the same length");
    if (naln == 0) return 0;
    SL_TRY(ensure_device());
    hipStream_t s = nullptr;
    const int64_t abase = aln_off[0], obase = orig_off[0];
    const int64_t atotal = aln_off[naln] - abase, ototal = orig_off[naln] - obase;
    std::vector<int64_t> arel(static_cast<size_t>(naln) + 1), orel(static_cast<size_t>(naln) + 1);
    for (int64_t i = 0; i <= naln; ++i) { arel[i] = aln_off[i] - abase; orel[i] = orig_off[i] - obase; }
    uint8_t *d_a, *d_o, *d_out; int64_t *d_aoff, *d_ooff; int8_t* d_status;
    SL_TRY(upload("unmask.aln", reinterpret_cast<const uint8_t*>(aln) + abase, static_cast<size_t>(atotal), &d_a, s));
    SL_TRY(upload("unmask.orig", reinterpret_cast<const uint8_t*>(orig) + obase, static_cast<size_t>(ototal), &d_o, s));
    SL_TRY(upload("unmask.aoff", arel.data(), arel.size(), &d_aoff, s));
    SL_TRY(upload("unmask.ooff", orel.data(), orel.size(), &d_ooff, s));
    SL_TRY(scratch("unmask.out", static_cast<size_t>(atotal), &d_out));
    SL_TRY(scratch("unmask.status", static_cast<size_t>(naln), &d_status));
    const int grid = static_cast<int>(std::min<int64_t>(naln, static_cast<int64_t>(ctx().num_cu) * 32));
    hipLaunchKernelGGL(k_unmask, dim3(grid), dim3(64), 0, s, d_a, d_aoff, d_o, d_ooff, static_cast<long long>(naln), d_out, d_status);
    SL_HIP(hipGetLastError());
    std::vector<int8_t> status(static_cast<size_t>(naln));
    SL_HIP(hipMemcpy(status.data(), d_status, status.size(), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < naln; ++i) {   // first failing row in the reference's order
        if (status[i] == 1) return fail("sequence in alignment string is longer than the original");
        if (status[i] == 2) return fail("original sequence and that in the alignment string have different lengths");
    }
    if (atotal) SL_HIP(hipMemcpy(out + abase, d_out, static_cast<size_t>(atotal), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sarlacc_mask_bad_bases(const char* seq, const int64_t* seq_off, const char* qual,
                                      const int64_t* qual_off, int64_t n, const double* enc_errors,
                                      const char* enc_names, int enc_n, double threshold, char* out) {
    if (n < 0) return fail("sarlacc_amd: negative number of sequences");
    SL_TRY(check_encoding(enc_errors, enc_names, enc_n));
    if (n == 0) return 0;
    // first length mismatch in read order (src/mask_bad_bases.cpp:34-36)
    int64_t len_bad = -1;
    for (int64_t i = 0; i < n; ++i)
        if (seq_off[i + 1] - seq_off[i] != qual_off[i + 1] - qual_off[i]) { len_bad = i; break; }
    const int64_t n_eval = len_bad >= 0 ? len_bad : n;
    const int64_t total = seq_off[n_eval] - seq_off[0];
    unsigned long long bad = ~0ull;
    if (total > 0) {
        SL_TRY(ensure_device());
        hipStream_t s = nullptr;
        uint8_t *d_s, *d_q, *d_o;
        double* d_e;
        unsigned long long* d_bad;
        SL_TRY(upload("mask.seq", reinterpret_cast<const uint8_t*>(seq) + seq_off[0], static_cast<size_t>(total), &d_s, s));
        SL_TRY(upload("mask.qual", reinterpret_cast<const uint8_t*>(qual) + qual_off[0], static_cast<size_t>(total), &d_q, s));
        SL_TRY(upload("mask.err", enc_errors, static_cast<size_t>(enc_n), &d_e, s));
        SL_TRY(scratch("mask.out", static_cast<size_t>(total), &d_o));
        SL_TRY(upload("mask.bad", &bad, 1, &d_bad, s));
        const int bs = 256;
        const int grid = static_cast<int>(std::min<long long>((total + bs - 1) / bs, static_cast<long long>(ctx().num_cu) * 16));
        hipLaunchKernelGGL(k_mask, dim3(grid), dim3(bs), sizeof(double) * enc_n, s, d_s, d_q, static_cast<long long>(total),
                           d_e, enc_n, static_cast<int>(enc_names[0]), threshold, d_o, d_bad);
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
        if (bad == ~0ull) SL_HIP(hipMemcpy(out, d_o, static_cast<size_t>(total), hipMemcpyDeviceToHost));
    }
    if (bad != ~0ull) return fail("quality cannot be lower than smallest encoded value");
    if (len_bad >= 0) return fail("sequence and quality strings should have the same length");
    return 0;
}
