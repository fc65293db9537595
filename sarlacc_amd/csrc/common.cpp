// common.cpp -- error reporting, device context and workspace cache.
#include "common.hpp"

#include <algorithm>
#include <vector>

#include <cstdlib>

#include "../../include/sarlacc_amd.h"

namespace sarlacc {

std::string& last_error() {
    static thread_local std::string msg;
    return msg;
}

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return 1;
}

static const char* const kOptNames[OPT_N] = {"msa_spec", "msa2_general_rows", "msa2_chain_hbm", "msa2_waves_per_cu", "msa2_single_wave", "msa2_batches", "align_pensel", "align_chunks",
                                             "align_k", "align_waves_per_cu", "consensus_chars", "consensus_generic", "msa_int32", "msa_affine", "umi_full_rounds", "umi_tile_search", "msa_bitvector", "msa_bitvector_core", "msa_bitvector_tile_gb", "align_interleave", "umi_split_min", "msa2_tight_profiles", "umi_scan_single", "align_wide_barrier", "msa2_budget_gb", "msa2_max_columns", "align_wide_band", "msa2_simple_extend", "msa2_wide_extend"};
static int* option_values() {
    static int values[OPT_N];
    static const bool parsed = [] {
        for (int k = 0; k < OPT_N; ++k) {
            std::string env = "SARLACC_";
            for (const char* p = kOptNames[k]; *p; ++p) env += static_cast<char>(*p >= 'a' && *p <= 'z' ? *p - 32 : *p);
            const char* e = std::getenv(env.c_str());
            values[k] = e ? std::atoi(e) : 0;
        }
        return true;
    }();
    (void)parsed;
    return values;
}
int option(Opt o) { return option_values()[o]; }
int set_option(const char* name, int value) {
    for (int k = 0; k < OPT_N; ++k)
        if (name && std::strcmp(name, kOptNames[k]) == 0) { option_values()[k] = value; return 0; }
    return fail("sarlacc_amd: unknown option '%s'", name ? name : "(null)");
}

Context& ctx() {
    static thread_local Context c;
    return c;
}

int ensure_device() {
    Context& c = ctx();
    if (c.ready) {
        SL_HIP(hipSetDevice(c.device));
        return 0;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail("sarlacc_amd: no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorName(e));
    if (c.device >= count) return fail("sarlacc_amd: device %d out of range (%d visible)", c.device, count);
    SL_HIP(hipSetDevice(c.device));
    hipDeviceProp_t prop;
    SL_HIP(hipGetDeviceProperties(&prop, c.device));
    c.num_cu = prop.multiProcessorCount;
    SL_HIP(hipEventCreate(&c.ev_start));
    SL_HIP(hipEventCreate(&c.ev_stop));
    c.ready = true;
    return 0;
}

int Context::buffer(const char* name, size_t bytes, void** out) {
    Workspace& w = ws[name];
    if (w.cap < bytes) {
        if (w.ptr) SL_HIP(hipFree(w.ptr));
        w.ptr = nullptr;
        w.cap = 0;
        // a quarter of slack so that a slowly growing request does not reallocate every call -- but at most 256 MB of it: on the
        // 48-GB tile of traceback records and the 30-GB map sets of a large MSA call a quarter was 30 GB of HBM nobody used
        size_t want = bytes + std::min<size_t>(bytes / 4, size_t(256) << 20) + 256;
        hipError_t e = hipMalloc(&w.ptr, want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&w.ptr, want);
        }
        if (e != hipSuccess) return fail("sarlacc_amd: cannot allocate %zu bytes of device memory for '%s'", bytes, name);
        w.cap = want;
    }
    *out = w.ptr;
    return 0;
}

void Context::stage_reset(const char* name) {
    StageTimer& t = stages[name];
    t.used = 0;
    t.open = false;
}

int Context::stage_begin(const char* name, hipStream_t s) {
    StageTimer& t = stages[name];
    if (t.used == t.segs.size()) {
        hipEvent_t a, b;
        SL_HIP(hipEventCreate(&a));
        SL_HIP(hipEventCreate(&b));
        t.segs.emplace_back(a, b);
    }
    SL_HIP(hipEventRecord(t.segs[t.used].first, s));
    t.open = true;
    return 0;
}

int Context::stage_end(const char* name, hipStream_t s) {
    StageTimer& t = stages[name];
    if (!t.open) return fail("sarlacc_amd: stage timer '%s' was never started", name);
    SL_HIP(hipEventRecord(t.segs[t.used].second, s));
    ++t.used;
    t.open = false;
    return 0;
}

void Context::release() {
    for (auto& kv : ws)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    ws.clear();
}

int check_encoding(const double* errors, const char* names, int n) {
    if (n <= 0 || !errors || !names) return fail("encoding vector must be non-empty and named");
    for (int i = 1; i < n; ++i) {
        // `curval != last + 1` compares a char with an int (src/quality_encoding.cpp:21): with the reference's signed char (x86) a
        // table that runs past byte 127 -- Biostrings' PhredQuality to Q 99 is '!' .. byte 132 -- is rejected by the reference
        // itself, which is why the tables of sarlacc_amd/encoding.py stop at '~'
        if (static_cast<int>(static_cast<signed char>(names[i])) != static_cast<int>(static_cast<signed char>(names[i - 1])) + 1)
            return fail("names of encoding vector should increase consecutively");
        if (errors[i] > errors[i - 1]) return fail("error probabilities should decrease");
    }
    return 0;
}

}  // namespace sarlacc

extern "C" {

const char* sarlacc_last_error(void) { return sarlacc::last_error().c_str(); }

int sarlacc_version(void) { return 100; }

int sarlacc_set_option(const char* name, int value) { return sarlacc::set_option(name, value); }

int sarlacc_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int sarlacc_set_device(int device) {
    sarlacc::Context& c = sarlacc::ctx();
    if (c.ready && c.device != device) {
        c.release();
        for (auto& kv : c.stages)   // events belong to the device they were created on
            for (auto& seg : kv.second.segs) {
                (void)hipEventDestroy(seg.first);
                (void)hipEventDestroy(seg.second);
            }
        c.stages.clear();
        c.ready = false;
    }
    c.device = device;
    return sarlacc::ensure_device();
}

void sarlacc_release_workspace(void) {
    sarlacc::ctx().release();
    (void)sarlacc_host_release();   // the idle page-locked result blocks as well
    (void)sarlacc_dev_pool_release();   // ... and the idle device blocks of sarlacc_dev_malloc
}

int64_t sarlacc_release_umi_workspace(void) {
    // every buffer umi.hip asks for carries one of these prefixes (encode_and_rank / pair_edges: "u1.", "u2."; the uploads of
    // the entry points: "g1.", "g2.", "g.", "ps", "lv", "lev"; adjacency "adj."; clustering "cl.")
    static const char* const prefixes[] = {"u1.", "u2.", "g1.", "g2.", "g.", "ps.", "ps", "lv.", "lv", "lev.", "lev", "adj.", "cl."};
    sarlacc::Context& c = sarlacc::ctx();
    int64_t freed = 0;
    for (auto it = c.ws.begin(); it != c.ws.end();) {
        bool mine = false;
        for (const char* pf : prefixes) {
            const size_t n = std::strlen(pf);
            const bool whole = pf[n - 1] != '.';   // names without a dot must match entirely
            if (whole ? it->first == pf : it->first.compare(0, n, pf) == 0) { mine = true; break; }
        }
        if (mine) {
            if (it->second.ptr) { (void)hipFree(it->second.ptr); freed += static_cast<int64_t>(it->second.cap); }
            it = c.ws.erase(it);
        } else {
            ++it;
        }
    }
    return freed;
}

int64_t sarlacc_workspace_report(char* buf, int64_t cap) {
    // "name bytes" lines of the cached device buffers of this thread, largest first; returns their total
    sarlacc::Context& c = sarlacc::ctx();
    std::vector<std::pair<size_t, std::string>> v;
    int64_t total = 0;
    for (const auto& kv : c.ws) { v.emplace_back(kv.second.cap, kv.first); total += static_cast<int64_t>(kv.second.cap); }
    std::sort(v.begin(), v.end(), [](const std::pair<size_t, std::string>& a, const std::pair<size_t, std::string>& b) { return a.first > b.first; });
    std::string out;
    for (const auto& e : v) out += e.second + " " + std::to_string(e.first) + "\n";
    if (buf && cap > 0) {
        const size_t nbytes = std::min<size_t>(static_cast<size_t>(cap) - 1, out.size());
        std::memcpy(buf, out.data(), nbytes);
        buf[nbytes] = 0;
    }
    return total;
}

double sarlacc_stage_ms(const char* name) {
    sarlacc::Context& c = sarlacc::ctx();
    if (!c.ready || !name) return -1.0;
    auto it = c.stages.find(name);
    if (it == c.stages.end() || it->second.used == 0) return -1.0;
    double total = 0;
    for (size_t k = 0; k < it->second.used; ++k) {
        if (hipEventSynchronize(it->second.segs[k].second) != hipSuccess) return -1.0;
        float ms = 0;
        if (hipEventElapsedTime(&ms, it->second.segs[k].first, it->second.segs[k].second) != hipSuccess) return -1.0;
        total += ms;
    }
    return total;
}

double sarlacc_stage_count(const char* name) {
    sarlacc::Context& c = sarlacc::ctx();
    if (!name) return -1.0;
    auto it = c.counts.find(name);
    return it == c.counts.end() ? -1.0 : it->second;
}

double sarlacc_last_kernel_ms(void) {
    sarlacc::Context& c = sarlacc::ctx();
    if (!c.ready || !c.timed) return -1.0;
    if (hipEventSynchronize(c.ev_stop) != hipSuccess) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c.ev_start, c.ev_stop) != hipSuccess) return -1.0;
    return ms;
}
}
