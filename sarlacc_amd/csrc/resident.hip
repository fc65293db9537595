// resident.hip -- device-resident read batches for callers that reuse the same reads many
// times: the scrambled-control grid search of tuneAlignment (/root/reference/R/tuneAlignment.R:30-72,
// up to 7 x 5 parameter pairs x 4 orientations x {real, scrambled}) and getAdaptorThresholds
// (R/getAdaptorThresholds.R:35-48,105-128).  Reads are uploaded once; the front/back windows
// (.get_front_and_back, R/adaptorAlign.R:86-95) and the per-read shuffles (.scramble_input,
// R/getAdaptorThresholds.R:68-92) are built on the device, and every grid point is one
// score-only launch of the DP kernel on resident data.
#include "common.hpp"

#include <map>
#include <mutex>
#include <vector>

#include "../../include/sarlacc_amd.h"

namespace sarlacc {

__device__ __forceinline__ uint8_t complement_base(uint8_t c) {
    switch (c) {  // Biostrings reverseComplement on DNA incl. IUPAC codes
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
        case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H';
    }
    return c;  // W, S, N, '-' map to themselves
}

// which = 0: first min(tol, L) bases.  which = 1: reverse complement of the last min(tol, L)
// bases, qualities reversed.  woff are the offsets of the output windows.
__global__ void k_windows(const uint8_t* seq, const uint8_t* qual, const int64_t* off, long long n,
                          const int64_t* woff, int which, uint8_t* oseq, uint8_t* oqual) {
    const long long r = blockIdx.x;
    if (r >= n) return;
    const long long s = off[r], L = off[r + 1] - s;
    const long long w0 = woff[r], w = woff[r + 1] - w0;
    for (long long p = threadIdx.x; p < w; p += blockDim.x) {
        if (which == 0) {
            oseq[w0 + p] = seq[s + p];
            oqual[w0 + p] = qual[s + p];
        } else {
            const long long src = s + L - 1 - p;
            oseq[w0 + p] = complement_base(seq[src]);
            oqual[w0 + p] = qual[src];
        }
    }
}

// realizeReads (R/realizeReads.R:28-43): output r is read idx[r], reverse-complemented when
// rev[r] (qualities reversed), then cut to the 1-based inclusive range [tstart[r], tend[r]] of
// the oriented read; ooff are the offsets of the outputs.
__global__ void k_realize(const uint8_t* seq, const uint8_t* qual, const int64_t* off, const int64_t* idx,
                          const uint8_t* rev, const int32_t* tstart, long long nout, const int64_t* ooff,
                          uint8_t* oseq, uint8_t* oqual) {
    const long long r = blockIdx.x;
    if (r >= nout) return;
    const long long src = idx[r];
    const long long s = off[src], L = off[src + 1] - s;
    const long long o0 = ooff[r], w = ooff[r + 1] - o0;
    const long long q0 = tstart[r] - 1;
    const bool flip = rev[r] != 0;
    for (long long p = threadIdx.x; p < w; p += blockDim.x) {
        const long long q = q0 + p;   // position in the oriented read
        if (flip) {
            oseq[o0 + p] = complement_base(seq[s + L - 1 - q]);
            oqual[o0 + p] = qual[s + L - 1 - q];
        } else {
            oseq[o0 + p] = seq[s + q];
            oqual[o0 + p] = qual[s + q];
        }
    }
}

// Sub-sequences of resident reads (XVector::subseq as .align_and_extract uses it, R/adaptorAlign.R:160-174): output r = `width[r]`
// bases from the 1-based position `start[r]` of read r of batch A, or of batch B where from_b[r] != 0 (the strand choice of
// adaptorAlign picks, per read, the alignment on the front or on the back window).  One thread per read: the pieces are a
// dozen bases long.  bad: smallest r whose range leaves its read.
__global__ void k_subseq(const uint8_t* seq_a, const int64_t* off_a, const uint8_t* seq_b, const int64_t* off_b, const uint8_t* from_b,
                         const int32_t* start, const int64_t* ooff, long long n, uint8_t* out, int* bad) {
    const long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (r >= n) return;
    const bool b = from_b && from_b[r];
    const uint8_t* seq = b ? seq_b : seq_a;
    const int64_t* off = b ? off_b : off_a;
    const long long s = off[r], L = off[r + 1] - s, o0 = ooff[r], w = ooff[r + 1] - o0, q0 = static_cast<long long>(start[r]) - 1;
    if (w <= 0) return;
    if (q0 < 0 || q0 + w > L) { atomicMin(bad, static_cast<int>(r)); return; }
    for (long long p = 0; p < w; ++p) out[o0 + p] = seq[s + q0 + p];
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Fisher-Yates shuffle of every read (bases and qualities with the same permutation).
// Stream per read: splitmix64 seeded with seed ^ (read + 1) * 0xD1342543DE82EF95; the swap
// partner of position k is ((next >> 32) * (k + 1)) >> 32.  oracle/align.c:orc_scramble
// restates it for parity tests.
__global__ void k_scramble(const uint8_t* seq, const uint8_t* qual, const int64_t* off, long long n,
                           unsigned long long seed, uint8_t* oseq, uint8_t* oqual) {
    const long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (r >= n) return;
    const long long s = off[r], L = off[r + 1] - s;
    for (long long p = 0; p < L; ++p) { oseq[s + p] = seq[s + p]; oqual[s + p] = qual[s + p]; }
    unsigned long long st = seed ^ (static_cast<unsigned long long>(r + 1) * 0xD1342543DE82EF95ull);
    for (long long k = L - 1; k > 0; --k) {
        const unsigned long long rnd = splitmix64(st) >> 32;
        const long long jx = static_cast<long long>((rnd * static_cast<unsigned long long>(k + 1)) >> 32);
        const uint8_t a = oseq[s + k], b = oqual[s + k];
        oseq[s + k] = oseq[s + jx]; oqual[s + k] = oqual[s + jx];
        oseq[s + jx] = a; oqual[s + jx] = b;
    }
}

// .resolve_strand and the row selection of .align_AA_internal (R/adaptorAlign.R:112-122, :190-207) on result blocks of
// sarlacc_dev_align that are still in HBM.  A block of n alignments with S sections: double scores[n] | int32 starts[n] |
// int32 ends[n] | int32 section starts[max(S, 1)][n] | int32 section widths[max(S, 1)][n].  Read r is reversed iff
// max(cs, 0) + max(ce, 0) < max(rs, 0) + max(re, 0) (strict, fp64 as the reference's R arithmetic); out1 takes row r of rs or
// cs, out2 of re or ce.
__device__ __forceinline__ void choose_row(const uint8_t* cur, const uint8_t* rc, uint8_t* out, bool rev, long long n, int S, long long r) {
    const uint8_t* src = rev ? rc : cur;
    reinterpret_cast<double*>(out)[r] = reinterpret_cast<const double*>(src)[r];
    const int32_t* si = reinterpret_cast<const int32_t*>(src + 8 * n);
    int32_t* oi = reinterpret_cast<int32_t*>(out + 8 * n);
    const int rows = 2 + 2 * (S > 0 ? S : 1);
    for (int k = 0; k < rows; ++k) oi[k * n + r] = si[k * n + r];
}
__global__ void k_choose_strand(const uint8_t* __restrict__ cs, const uint8_t* __restrict__ ce, const uint8_t* __restrict__ rs,
                                const uint8_t* __restrict__ re, long long n, int S1, int S2, uint8_t* __restrict__ out1,
                                uint8_t* __restrict__ out2, uint8_t* __restrict__ rev_out) {
    const long long r = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double fs = fmax(reinterpret_cast<const double*>(cs)[r], 0.0) + fmax(reinterpret_cast<const double*>(ce)[r], 0.0);
    const double bs = fmax(reinterpret_cast<const double*>(rs)[r], 0.0) + fmax(reinterpret_cast<const double*>(re)[r], 0.0);
    const bool rev = fs < bs;
    rev_out[r] = rev ? 1 : 0;
    choose_row(cs, rs, out1, rev, n, S1, r);
    choose_row(ce, re, out2, rev, n, S2, r);
}

}  // namespace sarlacc

using namespace sarlacc;

// ---- page-locked host blocks for results ----
// A device -> host copy into ordinary (pageable) memory goes through the runtime's staging buffer at the speed of a host memcpy
// (the 360 MB of consensus strings of a 10^6-read pass: 35 ms; the four result blocks of adaptorAlign: 20 ms); into page-locked
// memory it is one DMA transfer.  Page-locking costs far more than the copy it saves, so the blocks are pooled: sizes in
// powers of two from 1 MB, a freed block waits for the next request of its size (sarlacc_host_release frees the pool).
namespace {
struct HostPool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> idle;   // by block size
    std::map<void*, size_t> live;
    size_t idle_bytes = 0;
};
HostPool& host_pool() { static HostPool p; return p; }
}  // namespace

// Device blocks of the resident read batches (windows, realized reads ...) are pooled like the page-locked host blocks: hipMalloc
// + hipFree of the four 250-MB window arrays of a 10^6-read adaptorAlign cost 3 ms on one box of the pool and 25-35 ms on
// another -- more than the kernels that fill them.  Sizes in powers of two from 1 MB (smaller ones go to hipMalloc directly),
// at most 4 GB idle per device; sarlacc_release_workspace empties the pool.
namespace {
struct DevPool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> idle;   // by (device, block size)
    std::map<void*, std::pair<int, size_t>> live;               // pooled blocks handed out
    std::map<int, size_t> idle_bytes;
};
DevPool& dev_pool() { static DevPool p; return p; }
}  // namespace

extern "C" {

int sarlacc_dev_malloc(void** p, int64_t bytes) {
    if (!p || bytes < 0) return fail("sarlacc_amd: bad allocation request");
    SL_TRY(ensure_device());
    if (bytes < (1 << 20)) { SL_HIP(hipMalloc(p, static_cast<size_t>(bytes > 0 ? bytes : 1))); return 0; }
    size_t size = size_t(1) << 20;
    while (size < static_cast<size_t>(bytes)) size <<= 1;
    int dev = 0;
    SL_HIP(hipGetDevice(&dev));
    DevPool& P = dev_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    std::vector<void*>& idle = P.idle[{dev, size}];
    if (!idle.empty()) { *p = idle.back(); idle.pop_back(); P.idle_bytes[dev] -= size; }
    else if (hipMalloc(p, size) != hipSuccess) {
        (void)hipGetLastError();
        // (make room once: the idle blocks of this device go back, then the exact size is asked for)
        for (auto& kv : P.idle)
            if (kv.first.first == dev) { for (void* q : kv.second) (void)hipFree(q); kv.second.clear(); }
        P.idle_bytes[dev] = 0;
        SL_HIP(hipMalloc(p, static_cast<size_t>(bytes)));
        return 0;   // (not pooled: its size is not a pool size)
    }
    P.live[*p] = {dev, size};
    return 0;
}

int sarlacc_dev_free(void* p) {
    if (!p) return 0;
    DevPool& P = dev_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) { SL_HIP(hipFree(p)); return 0; }   // a small block, or one allocated outside the pool sizes
    const int dev = it->second.first;
    const size_t size = it->second.second;
    P.live.erase(it);
    if (P.idle_bytes[dev] + size > (size_t(4) << 30)) { SL_HIP(hipFree(p)); return 0; }
    // (the block may still be read by kernels queued on the null stream; the next user's work is queued behind them there)
    P.idle[{dev, size}].push_back(p);
    P.idle_bytes[dev] += size;
    return 0;
}

int sarlacc_dev_pool_release(void) {
    DevPool& P = dev_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& kv : P.idle) { for (void* q : kv.second) (void)hipFree(q); kv.second.clear(); }
    P.idle_bytes.clear();
    return 0;
}

int sarlacc_dev_upload(void* d, const void* h, int64_t bytes) {
    SL_TRY(ensure_device());
    if (bytes > 0) SL_HIP(hipMemcpy(d, h, static_cast<size_t>(bytes), hipMemcpyHostToDevice));
    return 0;
}

int sarlacc_dev_download(void* h, const void* d, int64_t bytes) {
    SL_TRY(ensure_device());
    if (bytes > 0) SL_HIP(hipMemcpy(h, d, static_cast<size_t>(bytes), hipMemcpyDeviceToHost));
    return 0;
}

int sarlacc_host_alloc(void** p, int64_t bytes) {
    if (!p || bytes < 0) return fail("sarlacc_amd: bad allocation request");
    SL_TRY(ensure_device());
    size_t size = size_t(1) << 20;
    while (size < static_cast<size_t>(bytes)) size <<= 1;
    HostPool& P = host_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    std::vector<void*>& idle = P.idle[size];
    if (!idle.empty()) { *p = idle.back(); idle.pop_back(); P.idle_bytes -= size; }
    else if (hipHostMalloc(p, size, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        // (make room once: every idle block goes back to the system, then the request is tried again)
        for (auto& kv : P.idle) { for (void* q : kv.second) (void)hipHostFree(q); kv.second.clear(); }
        P.idle_bytes = 0;
        SL_HIP(hipHostMalloc(p, size, hipHostMallocDefault));
    }
    P.live[*p] = size;
    return 0;
}

int sarlacc_host_free(void* p) {
    if (!p) return 0;
    HostPool& P = host_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) return fail("sarlacc_amd: sarlacc_host_free of a block that sarlacc_host_alloc did not hand out");
    const size_t size = it->second;
    P.live.erase(it);
    // (at most 1.5 GB wait in the pool -- the two result blocks of a 10^6-read pass --; beyond that the block goes back to the
    // system: page-locked memory cannot be swapped, and on an 8-GPU node every rank has a pool of its own)
    if (P.idle_bytes + size > (size_t(3) << 29)) { SL_HIP(hipHostFree(p)); return 0; }
    P.idle[size].push_back(p);
    P.idle_bytes += size;
    return 0;
}

int sarlacc_host_release(void) {
    HostPool& P = host_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    for (auto& kv : P.idle) { for (void* q : kv.second) (void)hipHostFree(q); kv.second.clear(); }
    P.idle_bytes = 0;
    return 0;
}

int sarlacc_dev_realize(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, const int64_t* d_idx,
                        const uint8_t* d_rev, const int32_t* d_tstart, int64_t n_out, const int64_t* d_ooff,
                        uint8_t* d_oseq, uint8_t* d_oqual, void* stream) {
    if (n_out < 0) return fail("sarlacc_amd: bad realize request");
    if (n_out == 0) return 0;
    SL_TRY(ensure_device());
    hipLaunchKernelGGL(k_realize, dim3(static_cast<unsigned>(n_out)), dim3(128), 0, static_cast<hipStream_t>(stream), d_seq,
                       d_qual, d_off, d_idx, d_rev, d_tstart, static_cast<long long>(n_out), d_ooff, d_oseq, d_oqual);
    SL_HIP(hipGetLastError());
    return 0;
}

int sarlacc_dev_windows(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                        const int64_t* d_woff, int which, uint8_t* d_oseq, uint8_t* d_oqual, void* stream) {
    if (n < 0 || (which != 0 && which != 1)) return fail("sarlacc_amd: bad window request");
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipLaunchKernelGGL(k_windows, dim3(static_cast<unsigned>(n)), dim3(64), 0, static_cast<hipStream_t>(stream), d_seq,
                       d_qual, d_off, static_cast<long long>(n), d_woff, which, d_oseq, d_oqual);
    SL_HIP(hipGetLastError());
    return 0;
}

int sarlacc_dev_subseq(const uint8_t* d_seq_a, const int64_t* d_off_a, const uint8_t* d_seq_b, const int64_t* d_off_b,
                       const uint8_t* from_b, const int32_t* start, const int32_t* width, int64_t n, char* out_chars,
                       int64_t out_cap, int64_t* out_off, void* stream) {
    if (n < 0) return fail("sarlacc_amd: negative number of reads");
    out_off[0] = 0;
    for (int64_t r = 0; r < n; ++r) out_off[r + 1] = out_off[r] + (width[r] > 0 ? width[r] : 0);
    if (n == 0 || out_off[n] == 0) return 0;
    if (out_off[n] > out_cap) return fail("sarlacc_amd: sub-sequence buffer too small (%lld needed)", static_cast<long long>(out_off[n]));
    if (from_b && (!d_seq_b || !d_off_b)) return fail("sarlacc_amd: a second batch is selected but not given");
    SL_TRY(ensure_device());
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* d_sel = nullptr; int32_t* d_start; int64_t* d_ooff; uint8_t* d_out; int* d_bad;
    if (from_b) SL_TRY(upload("subseq.sel", from_b, static_cast<size_t>(n), &d_sel, s));
    SL_TRY(upload("subseq.start", start, static_cast<size_t>(n), &d_start, s));
    SL_TRY(upload("subseq.ooff", out_off, static_cast<size_t>(n) + 1, &d_ooff, s));
    SL_TRY(scratch("subseq.out", static_cast<size_t>(out_off[n]), &d_out));
    SL_TRY(scratch("subseq.bad", 1, &d_bad));
    const int big = 0x7fffffff;
    SL_HIP(hipMemcpyAsync(d_bad, &big, sizeof big, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_subseq, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_seq_a, d_off_a, d_seq_b, d_off_b, d_sel, d_start,
                       d_ooff, static_cast<long long>(n), d_out, d_bad);
    SL_HIP(hipGetLastError());
    int bad = big;
    SL_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(out_chars, d_out, static_cast<size_t>(out_off[n]), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (bad != big) return fail("sarlacc_amd: sub-sequence of read %d lies outside the read", bad + 1);
    return 0;
}

int sarlacc_dev_choose_strand(const void* d_cs, const void* d_ce, const void* d_rs, const void* d_re, int64_t n, int nsec1, int nsec2,
                              void* d_out1, void* d_out2, uint8_t* d_rev, void* stream) {
    if (n < 0 || nsec1 < 0 || nsec2 < 0) return fail("sarlacc_amd: bad strand choice request");
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipLaunchKernelGGL(k_choose_strand, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t*>(d_cs), static_cast<const uint8_t*>(d_ce), static_cast<const uint8_t*>(d_rs),
                       static_cast<const uint8_t*>(d_re), static_cast<long long>(n), nsec1, nsec2, static_cast<uint8_t*>(d_out1),
                       static_cast<uint8_t*>(d_out2), d_rev);
    SL_HIP(hipGetLastError());
    return 0;
}

int sarlacc_dev_scramble(const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_off, int64_t n,
                         uint64_t seed, uint8_t* d_oseq, uint8_t* d_oqual, void* stream) {
    if (n < 0) return fail("sarlacc_amd: negative number of reads");
    if (n == 0) return 0;
    SL_TRY(ensure_device());
    hipLaunchKernelGGL(k_scramble, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), d_seq, d_qual, d_off, static_cast<long long>(n),
                       static_cast<unsigned long long>(seed), d_oseq, d_oqual);
    SL_HIP(hipGetLastError());
    return 0;
}
}
