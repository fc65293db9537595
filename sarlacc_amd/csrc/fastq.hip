// fastq.hip -- FASTQ text -> resident read batch, on gfx950.
//
// The reference streams the file through ShortRead::FastqStreamer and converts every chunk
// to a QualityScaledDNAStringSet on the host (/root/reference/R/adaptorAlign.R:26-37,
// .FASTQ2QSDS :104-110; again in R/realizeReads.R:15-26).  Here the raw text is copied to
// HBM once and parsed there; sequences, qualities, offsets and read names come out in the
// flat layout every other kernel of the library consumes, without a host-side pass over
// the bases.
//
//   k_fq_count    newlines per 8-KB tile                           (reads the text once)
//   (rocPRIM exclusive scan of the tile counts)
//   k_fq_lines    start offset of every line                       (reads the text again)
//   k_fq_records  per 4-line record: checks '@' / '+' / equal lengths, strips CR, lengths
//   (rocPRIM exclusive scans of the sequence and name lengths)
//   k_fq_copy     sequence (upper-cased), quality and name bytes -> contiguous arrays
//   k_fq_nth_newline  end of the k-th record of a block of text (chunked streaming, `number` of adaptorAlign)
//
// Streaming byte work, HBM-bound: ~2 reads + 1 write of the text.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

#include "../../include/sarlacc_amd.h"

#include <algorithm>
#include <cstdlib>
#include <limits>

namespace sarlacc {

constexpr int FQ_TILE = 8192;      // bytes per block in the line passes
constexpr int FQ_THREADS = 256;
constexpr int FQ_PER_THREAD = FQ_TILE / FQ_THREADS;  // 32 bytes: two 16-byte loads

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// number of '\n' among the 4 bytes of w
__device__ __forceinline__ int newlines4(uint32_t w) {
    const uint32_t x = w ^ 0x0a0a0a0au;  // zero byte where '\n'
    int c = 0;
    c += (x & 0x000000ffu) == 0;
    c += (x & 0x0000ff00u) == 0;
    c += (x & 0x00ff0000u) == 0;
    c += (x & 0xff000000u) == 0;
    return c;
}

__device__ __forceinline__ void load32(const uint8_t* text, long long pos, long long nbytes, uint32_t (&w)[8]) {
    if (pos + FQ_PER_THREAD <= nbytes && (reinterpret_cast<uintptr_t>(text + pos) & 15) == 0) {
        const uint4 a = *reinterpret_cast<const uint4*>(text + pos);
        const uint4 b = *reinterpret_cast<const uint4*>(text + pos + 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long long p = pos + 4 * k + b;
                if (p < nbytes) v |= static_cast<uint32_t>(text[p]) << (8 * b);
            }
            w[k] = v;
        }
    }
}

__global__ void __launch_bounds__(FQ_THREADS) k_fq_count(const uint8_t* text, long long nbytes, long long* tile_count) {
    const long long pos = static_cast<long long>(blockIdx.x) * FQ_TILE + threadIdx.x * FQ_PER_THREAD;
    int c = 0;
    if (pos < nbytes) {
        uint32_t w[8];
        load32(text, pos, nbytes, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) c += newlines4(w[k]);
    }
    __shared__ int s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_sum;
}

// line_start[k + 1] = position after the k-th newline (0-based); line_start[0] = 0 is set by the host
__global__ void __launch_bounds__(FQ_THREADS) k_fq_lines(const uint8_t* text, long long nbytes, const long long* tile_base,
                                                         long long* line_start) {
    const long long pos = static_cast<long long>(blockIdx.x) * FQ_TILE + threadIdx.x * FQ_PER_THREAD;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    if (pos < nbytes) {
        load32(text, pos, nbytes, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) c += newlines4(w[k]);
    }
    // exclusive scan of the per-thread counts over the block
    __shared__ int s_wave[FQ_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = incl - c;
    for (int k = 0; k < wave; ++k) before += s_wave[k];
    if (c == 0) return;
    long long rank = tile_base[blockIdx.x] + before;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (((w[k] >> (8 * b)) & 0xffu) == 0x0au && pos + 4 * k + b < nbytes) {
                line_start[rank + 1] = pos + 4 * k + b + 1;
                ++rank;
            }
        }
    }
}

// position just after newline number `target` (1-based): only the tile that holds it does any work
__global__ void __launch_bounds__(FQ_THREADS) k_fq_nth_newline(const uint8_t* text, long long nbytes, const long long* tile_base,
                                                               long long target, long long* out) {
    const long long before_tile = tile_base[blockIdx.x];
    if (target <= before_tile || target > tile_base[blockIdx.x + 1]) return;   // uniform per block
    const long long pos = static_cast<long long>(blockIdx.x) * FQ_TILE + threadIdx.x * FQ_PER_THREAD;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    if (pos < nbytes) {
        load32(text, pos, nbytes, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) c += newlines4(w[k]);
    }
    __shared__ int s_wave[FQ_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long long rank = before_tile + incl - c;
    for (int k = 0; k < wave; ++k) rank += s_wave[k];
    if (target <= rank || target > rank + c) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (((w[k] >> (8 * b)) & 0xffu) == 0x0au && pos + 4 * k + b < nbytes) {
                if (++rank == target) *out = pos + 4 * k + b + 1;
            }
        }
    }
}

struct FqRec {
    long long seq_pos, qual_pos, name_pos;
};

// status codes written through atomicMin on (record << 3 | code)
enum { FQ_BAD_HEADER = 1, FQ_BAD_PLUS = 2, FQ_BAD_LENGTH = 3 };

__global__ void k_fq_records(const uint8_t* text, const long long* line_start, long long nrec, FqRec* rec,
                             long long* seq_len, long long* name_len, unsigned long long* first_bad) {
    const long long r = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
    if (r >= nrec) return;
    long long b[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        b[k] = line_start[4 * r + k];
        e[k] = line_start[4 * r + k + 1] - 1;               // the newline (or the end of the text)
        if (e[k] > b[k] && text[e[k] - 1] == '\r') --e[k];  // CRLF files
    }
    int bad = 0;
    if (e[0] <= b[0] || text[b[0]] != '@') bad = FQ_BAD_HEADER;
    else if (e[2] <= b[2] || text[b[2]] != '+') bad = FQ_BAD_PLUS;
    else if (e[1] - b[1] != e[3] - b[3]) bad = FQ_BAD_LENGTH;
    if (bad) atomicMin(first_bad, (static_cast<unsigned long long>(r) << 3) | static_cast<unsigned>(bad));
    rec[r].seq_pos = b[1];
    rec[r].qual_pos = b[3];
    rec[r].name_pos = b[0] + 1;
    seq_len[r] = bad ? 0 : e[1] - b[1];
    name_len[r] = bad ? 0 : e[0] - b[0] - 1;
}

// a-z -> A-Z on four packed bytes
__device__ __forceinline__ uint32_t upper4(uint32_t w) {
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t c = (w >> (8 * b)) & 0xffu;
        if (c - 'a' < 26u) c -= 32u;
        out |= c << (8 * b);
    }
    return out;
}

template <bool UPPER>
__device__ __forceinline__ void copy_bytes(const uint8_t* src, uint8_t* dst, long long len, int tid, int nthreads) {
    const long long words = len >> 2;
    for (long long p = tid; p < words; p += nthreads) {
        uint32_t v = *reinterpret_cast<const u32_unaligned*>(src + 4 * p);
        if (UPPER) v = upper4(v);
        *reinterpret_cast<u32_unaligned*>(dst + 4 * p) = v;
    }
    for (long long p = 4 * words + tid; p < len; p += nthreads) {
        uint8_t c = src[p];
        if (UPPER && static_cast<unsigned>(c - 'a') < 26u) c -= 32;
        dst[p] = c;
    }
}

__global__ void __launch_bounds__(128) k_fq_copy(const uint8_t* text, const FqRec* rec, const int64_t* off,
                                                 const int64_t* name_off, long long nrec, uint8_t* seq, uint8_t* qual,
                                                 uint8_t* names) {
    for (long long r = blockIdx.x; r < nrec; r += gridDim.x) {
        const FqRec R = rec[r];
        const long long o = off[r], len = off[r + 1] - o;
        copy_bytes<true>(text + R.seq_pos, seq + o, len, threadIdx.x, blockDim.x);
        copy_bytes<false>(text + R.qual_pos, qual + o, len, threadIdx.x, blockDim.x);
        if (names) {
            const long long no = name_off[r];
            copy_bytes<false>(text + R.name_pos, names + no, name_off[r + 1] - no, threadIdx.x, blockDim.x);
        }
    }
}

template <typename T>
static int exclusive_scan_i64(const char* tag, const T* d_in, int64_t* d_out, size_t n, hipStream_t s) {
    size_t tmp = 0;
    SL_HIP(rocprim::exclusive_scan(nullptr, tmp, d_in, d_out, static_cast<int64_t>(0), n, rocprim::plus<int64_t>(), s));
    void* d_tmp;
    SL_TRY(ctx().buffer(tag, tmp ? tmp : 1, &d_tmp));
    SL_HIP(rocprim::exclusive_scan(d_tmp, tmp, d_in, d_out, static_cast<int64_t>(0), n, rocprim::plus<int64_t>(), s));
    return 0;
}

// state of the last sarlacc_dev_fastq_index call on this thread
struct FqIndex {
    const uint8_t* text = nullptr;
    int64_t nbytes = 0, nrec = 0, total_bases = 0, total_name = 0;
};
static thread_local FqIndex g_fq;

}  // namespace sarlacc

using namespace sarlacc;

extern "C" {

int sarlacc_dev_fastq_index(const uint8_t* d_text, int64_t nbytes, int64_t* n_records, int64_t* total_bases,
                            int64_t* total_name_bytes, void* stream) {
    SL_TRY(ensure_device());
    if (nbytes < 0) return fail("sarlacc_amd: negative FASTQ size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    g_fq = FqIndex{};
    *n_records = 0; *total_bases = 0; *total_name_bytes = 0;
    // trailing blank lines do not start a record: drop them from the text (host look at the tail)
    while (nbytes > 0) {
        const int64_t k = std::min<int64_t>(nbytes, 4096);
        std::vector<uint8_t> tail(static_cast<size_t>(k));
        SL_HIP(hipMemcpyAsync(tail.data(), d_text + nbytes - k, static_cast<size_t>(k), hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        int64_t cut = k;
        while (cut > 0 && (tail[cut - 1] == '\n' || tail[cut - 1] == '\r')) --cut;
        nbytes -= k - cut;
        if (cut > 0) break;
    }
    if (nbytes == 0) {
        g_fq.text = d_text;  // an empty batch is a valid result
        return 0;
    }

    const long long ntiles = (nbytes + FQ_TILE - 1) / FQ_TILE;
    long long* d_count; long long* d_base;
    SL_TRY(scratch("fq.count", static_cast<size_t>(ntiles) + 1, &d_count));
    SL_TRY(scratch("fq.base", static_cast<size_t>(ntiles) + 1, &d_base));
    SL_HIP(hipMemsetAsync(d_count + ntiles, 0, sizeof(long long), s));
    hipLaunchKernelGGL(k_fq_count, dim3(static_cast<unsigned>(ntiles)), dim3(FQ_THREADS), 0, s, d_text, static_cast<long long>(nbytes), d_count);
    SL_HIP(hipGetLastError());
    SL_TRY(exclusive_scan_i64("fq.scan", d_count, reinterpret_cast<int64_t*>(d_base), static_cast<size_t>(ntiles) + 1, s));
    long long newlines = 0;
    SL_HIP(hipMemcpyAsync(&newlines, d_base + ntiles, sizeof newlines, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    // the text no longer ends in a newline, so there is one more line than newlines.  A last
    // record whose quality line is empty lost that line with the trailing newlines: it is put
    // back as an empty virtual line (the record check still compares the two lengths).
    const long long nlines = newlines + 1;
    if (nlines % 4 != 0 && nlines % 4 != 3) return fail("FASTQ text ends inside a record (%lld lines)", nlines);
    const long long nrec = (nlines + 1) / 4;

    long long* d_lines;
    SL_TRY(scratch("fq.lines", static_cast<size_t>(nlines) + 2, &d_lines));
    const long long zero = 0, ends[2] = {nbytes + 1, nbytes + 2};
    SL_HIP(hipMemcpyAsync(d_lines, &zero, sizeof zero, hipMemcpyHostToDevice, s));
    SL_HIP(hipMemcpyAsync(d_lines + nlines, ends, sizeof ends, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fq_lines, dim3(static_cast<unsigned>(ntiles)), dim3(FQ_THREADS), 0, s, d_text, static_cast<long long>(nbytes), d_base, d_lines);
    SL_HIP(hipGetLastError());

    FqRec* d_rec; long long* d_slen; long long* d_nlen; int64_t* d_off; int64_t* d_noff; unsigned long long* d_bad;
    SL_TRY(scratch("fq.rec", static_cast<size_t>(nrec), &d_rec));
    SL_TRY(scratch("fq.slen", static_cast<size_t>(nrec) + 1, &d_slen));
    SL_TRY(scratch("fq.nlen", static_cast<size_t>(nrec) + 1, &d_nlen));
    SL_TRY(scratch("fq.off", static_cast<size_t>(nrec) + 1, &d_off));
    SL_TRY(scratch("fq.noff", static_cast<size_t>(nrec) + 1, &d_noff));
    SL_TRY(scratch("fq.bad", 1, &d_bad));
    SL_HIP(hipMemsetAsync(d_bad, 0xff, sizeof(unsigned long long), s));
    SL_HIP(hipMemsetAsync(d_slen + nrec, 0, sizeof(long long), s));
    SL_HIP(hipMemsetAsync(d_nlen + nrec, 0, sizeof(long long), s));
    hipLaunchKernelGGL(k_fq_records, dim3(static_cast<unsigned>((nrec + 255) / 256)), dim3(256), 0, s, d_text, d_lines, nrec,
                       d_rec, d_slen, d_nlen, d_bad);
    SL_HIP(hipGetLastError());
    SL_TRY(exclusive_scan_i64("fq.scan", d_slen, d_off, static_cast<size_t>(nrec) + 1, s));
    SL_TRY(exclusive_scan_i64("fq.scan", d_nlen, d_noff, static_cast<size_t>(nrec) + 1, s));
    unsigned long long bad = 0;
    int64_t tb = 0, tn = 0;
    SL_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(&tb, d_off + nrec, sizeof tb, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(&tn, d_noff + nrec, sizeof tn, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (bad != ~0ull) {
        const long long r = static_cast<long long>(bad >> 3);
        switch (bad & 7u) {
            case FQ_BAD_HEADER: return fail("FASTQ record %lld does not start with '@'", r + 1);
            case FQ_BAD_PLUS: return fail("FASTQ record %lld has no '+' line", r + 1);
            default: return fail("FASTQ record %lld: sequence and quality lengths differ", r + 1);
        }
    }
    g_fq.text = d_text; g_fq.nbytes = nbytes; g_fq.nrec = nrec; g_fq.total_bases = tb; g_fq.total_name = tn;
    *n_records = nrec; *total_bases = tb; *total_name_bytes = tn;
    return 0;
}

int sarlacc_dev_fastq_split(const uint8_t* d_text, int64_t nbytes, int64_t max_records, int64_t* n_records,
                            int64_t* consumed_bytes, void* stream) {
    SL_TRY(ensure_device());
    if (nbytes < 0 || max_records < 0) return fail("sarlacc_amd: negative FASTQ size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    *n_records = 0; *consumed_bytes = 0;
    if (nbytes == 0 || max_records == 0) return 0;
    const long long ntiles = (nbytes + FQ_TILE - 1) / FQ_TILE;
    long long* d_count; long long* d_base; long long* d_pos;
    SL_TRY(scratch("fq.count", static_cast<size_t>(ntiles) + 1, &d_count));
    SL_TRY(scratch("fq.base", static_cast<size_t>(ntiles) + 1, &d_base));
    SL_TRY(scratch("fq.nth", 1, &d_pos));
    SL_HIP(hipMemsetAsync(d_count + ntiles, 0, sizeof(long long), s));
    hipLaunchKernelGGL(k_fq_count, dim3(static_cast<unsigned>(ntiles)), dim3(FQ_THREADS), 0, s, d_text, static_cast<long long>(nbytes), d_count);
    SL_HIP(hipGetLastError());
    SL_TRY(exclusive_scan_i64("fq.scan", d_count, reinterpret_cast<int64_t*>(d_base), static_cast<size_t>(ntiles) + 1, s));
    long long newlines = 0;
    SL_HIP(hipMemcpyAsync(&newlines, d_base + ntiles, sizeof newlines, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    const long long k = std::min<long long>(newlines / 4, max_records);   // records whose four lines all end in a newline
    if (k == 0) return 0;
    long long pos = -1;
    SL_HIP(hipMemsetAsync(d_pos, 0xff, sizeof(long long), s));
    hipLaunchKernelGGL(k_fq_nth_newline, dim3(static_cast<unsigned>(ntiles)), dim3(FQ_THREADS), 0, s, d_text, static_cast<long long>(nbytes),
                       d_base, 4 * k, d_pos);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpyAsync(&pos, d_pos, sizeof pos, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (pos <= 0 || pos > nbytes) return fail("sarlacc_amd: internal error locating the end of FASTQ record %lld", k);
    *n_records = k; *consumed_bytes = pos;
    return 0;
}

int sarlacc_dev_fastq_extract(const uint8_t* d_text, uint8_t* d_seq, uint8_t* d_qual, int64_t* d_off, uint8_t* d_names,
                              int64_t* d_name_off, void* stream) {
    SL_TRY(ensure_device());
    if (d_text != g_fq.text) return fail("sarlacc_amd: sarlacc_dev_fastq_extract without a matching sarlacc_dev_fastq_index");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nrec = g_fq.nrec;
    FqRec* d_rec; int64_t* off; int64_t* noff;
    SL_TRY(scratch("fq.rec", static_cast<size_t>(std::max<int64_t>(nrec, 1)), &d_rec));
    SL_TRY(scratch("fq.off", static_cast<size_t>(nrec) + 1, &off));
    SL_TRY(scratch("fq.noff", static_cast<size_t>(nrec) + 1, &noff));
    if (nrec == 0) {
        const int64_t zero = 0;
        SL_HIP(hipMemcpyAsync(d_off, &zero, sizeof zero, hipMemcpyHostToDevice, s));
        if (d_name_off) SL_HIP(hipMemcpyAsync(d_name_off, &zero, sizeof zero, hipMemcpyHostToDevice, s));
        SL_HIP(hipStreamSynchronize(s));
        return 0;
    }
    SL_HIP(hipMemcpyAsync(d_off, off, sizeof(int64_t) * (static_cast<size_t>(nrec) + 1), hipMemcpyDeviceToDevice, s));
    if (d_name_off) SL_HIP(hipMemcpyAsync(d_name_off, noff, sizeof(int64_t) * (static_cast<size_t>(nrec) + 1), hipMemcpyDeviceToDevice, s));
    Context& c = ctx();
    // one workgroup per record (up to 1024 per CU): dynamic dispatch beats a resident grid with a stride
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(nrec, static_cast<int64_t>(c.num_cu) * 1024));
    SL_HIP(hipEventRecord(c.ev_start, s));
    hipLaunchKernelGGL(k_fq_copy, dim3(grid), dim3(128), 0, s, d_text, d_rec, off, noff, static_cast<long long>(nrec), d_seq, d_qual,
                       d_names);
    SL_HIP(hipGetLastError());
    SL_HIP(hipEventRecord(c.ev_stop, s));
    c.timed = true;
    SL_HIP(hipStreamSynchronize(s));
    return 0;
}
}
