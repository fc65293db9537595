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

}  // namespace sarlacc

using namespace sarlacc;

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
