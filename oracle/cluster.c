/*
 * cluster.c -- oracle restatement of sarlacc's greedy UMI clustering and umi_group.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows (file:line relative to /root/reference):
 *   src/cluster_umis.cpp:7-112   greedy max-neighbour clustering
 *   src/umi_group.cpp:14-116     per pre-group trie search -> (UMI1 n UMI2) -> cluster -> remap
 *   R/umiGroup.R:21-22           flattening of the per-group lists
 *
 * Semantics restated (SURVEY.md section 8 a11 / App.B Q12):
 *   1. nodes whose list has exactly one entry are emitted first, in index order
 *      (the entry must be the node itself; an empty list is an error);
 *   2. then repeatedly: take the node with the largest 'remaining' count, ties to
 *      the LARGEST index; its cluster is its still-unused neighbours in list
 *      order; each newly used neighbour zeroes its own count and decrements the
 *      count of everything in its list.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);
int orc_trie_neighbours(const char* const* strs, const int32_t* lens, int64_t n, int limit,
                        int64_t* nbr_off, int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need);

/* Emits the cluster seeded at 'seed'; returns its size. */
static int64_t take_cluster(const int64_t* off, const int32_t* links, int64_t* remaining,
                            int64_t seed, int32_t* dst) {
    int64_t m = 0;
    for (int64_t p = off[seed]; p < off[seed + 1]; ++p) {
        const int32_t v = links[p];
        if (remaining[v] == 0) continue;
        dst[m++] = v;
        remaining[v] = 0;
        for (int64_t q = off[v]; q < off[v + 1]; ++q) {
            const int32_t w = links[q];
            if (remaining[w] > 0) --remaining[w];
        }
    }
    return m;
}

static int emit_solos(const int64_t* off, const int32_t* links, int64_t n, int64_t* remaining,
                      int64_t* nclu, int64_t* clu_off, int32_t* clu) {
    *nclu = 0;
    clu_off[0] = 0;
    for (int64_t a = 0; a < n; ++a) {
        const int64_t len = off[a + 1] - off[a];
        remaining[a] = len;
        if (len == 0) return orc_fail("zero length read group");
        if (len == 1) {
            if (links[off[a]] != a) return orc_fail("single-read groups should contain only the read itself");
            clu[clu_off[*nclu]] = (int32_t)a;
            clu_off[*nclu + 1] = clu_off[*nclu] + 1;
            ++*nclu;
        }
    }
    return 0;
}

/* Literal O(#clusters * n) form. */
int orc_cluster_umis(const int64_t* off, const int32_t* links, int64_t n,
                     int64_t* nclu, int64_t* clu_off, int32_t* clu) {
    int64_t* remaining = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    if (emit_solos(off, links, n, remaining, nclu, clu_off, clu)) { free(remaining); return 1; }
    /* solos stay in the candidate pool of the reference only until they are
     * stripped as "remaining==0"/never: they were swapped in front of 'left'
     * (src/cluster_umis.cpp:42-43), i.e. excluded from the greedy phase.  Note
     * their counts are NOT zeroed; a solo lists only itself so nothing else can
     * reference it in a symmetric graph. */
    char* in_pool = (char*)malloc((size_t)(n ? n : 1));
    for (int64_t a = 0; a < n; ++a) in_pool[a] = (off[a + 1] - off[a]) > 1;

    for (;;) {
        int64_t best = -1;
        for (int64_t a = 0; a < n; ++a) {
            if (!in_pool[a] || remaining[a] == 0) continue;
            if (best < 0 || remaining[a] >= remaining[best]) best = a; /* ties -> larger index */
        }
        if (best < 0) break;
        in_pool[best] = 0;
        const int64_t m = take_cluster(off, links, remaining, best, clu + clu_off[*nclu]);
        clu_off[*nclu + 1] = clu_off[*nclu] + m;
        ++*nclu;
    }
    free(in_pool);
    free(remaining);
    return 0;
}

/* Same picks through a lazy max-heap keyed by (remaining, index). */
typedef struct { int64_t cnt; int64_t idx; } hkey;
static int hless(hkey a, hkey b) { return a.cnt < b.cnt || (a.cnt == b.cnt && a.idx < b.idx); }
static void hpush(hkey* h, int64_t* n, hkey k) {
    int64_t i = (*n)++;
    h[i] = k;
    while (i > 0) {
        int64_t p = (i - 1) / 2;
        if (!hless(h[p], h[i])) break;
        hkey t = h[p]; h[p] = h[i]; h[i] = t;
        i = p;
    }
}
static hkey hpop(hkey* h, int64_t* n) {
    hkey top = h[0];
    h[0] = h[--(*n)];
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hless(h[m], h[l])) m = l;
        if (r < *n && hless(h[m], h[r])) m = r;
        if (m == i) break;
        hkey t = h[m]; h[m] = h[i]; h[i] = t;
        i = m;
    }
    return top;
}

int orc_cluster_umis_fast(const int64_t* off, const int32_t* links, int64_t n,
                          int64_t* nclu, int64_t* clu_off, int32_t* clu) {
    int64_t* remaining = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    if (emit_solos(off, links, n, remaining, nclu, clu_off, clu)) { free(remaining); return 1; }
    /* every decrement re-pushes, so the heap can hold n + E entries */
    const int64_t E = off[n];
    hkey* heap = (hkey*)malloc(sizeof(hkey) * (size_t)(n + E + 1));
    int64_t hn = 0;
    char* in_pool = (char*)malloc((size_t)(n ? n : 1));
    for (int64_t a = 0; a < n; ++a) {
        in_pool[a] = (off[a + 1] - off[a]) > 1;
        if (in_pool[a]) { hkey k = {remaining[a], a}; hpush(heap, &hn, k); }
    }
    while (hn > 0) {
        const hkey top = hpop(heap, &hn);
        if (!in_pool[top.idx] || remaining[top.idx] != top.cnt || top.cnt == 0) continue; /* stale */
        in_pool[top.idx] = 0;
        int32_t* dst = clu + clu_off[*nclu];
        int64_t m = 0;
        for (int64_t p = off[top.idx]; p < off[top.idx + 1]; ++p) {
            const int32_t v = links[p];
            if (remaining[v] == 0) continue;
            dst[m++] = v;
            remaining[v] = 0;
            for (int64_t q = off[v]; q < off[v + 1]; ++q) {
                const int32_t w = links[q];
                if (remaining[w] > 0) {
                    --remaining[w];
                    if (in_pool[w] && remaining[w] > 0) { hkey k = {remaining[w], w}; hpush(heap, &hn, k); }
                }
            }
        }
        clu_off[*nclu + 1] = clu_off[*nclu] + m;
        ++*nclu;
    }
    free(in_pool);
    free(heap);
    free(remaining);
    return 0;
}

/* ------------------------------------------------------------------ */
static int cmp_i32(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

static int neighbours_grow(const char* const* strs, const int32_t* lens, int64_t n, int limit,
                           int64_t* off, int32_t** buf, int64_t* cap) {
    int64_t need = 0;
    int rc = orc_trie_neighbours(strs, lens, n, limit, off, *buf, *cap, &need);
    if (rc == 2) {
        *cap = need;
        *buf = (int32_t*)realloc(*buf, sizeof(int32_t) * (size_t)(need ? need : 1));
        rc = orc_trie_neighbours(strs, lens, n, limit, off, *buf, *cap, &need);
    }
    return rc;
}

int orc_umi_group(const char* umi1, const int64_t* off1,
                  const char* umi2, const int64_t* off2,
                  int64_t n, int thresh1, int thresh2,
                  const int64_t* grp_off, const int32_t* grp, int64_t ngroups,
                  int fast_cluster,
                  int64_t* nclusters, int64_t* clu_off, int32_t* clu) {
    (void)n;
    int64_t maxg = 0;
    for (int64_t g = 0; g < ngroups; ++g)
        if (grp_off[g + 1] - grp_off[g] > maxg) maxg = grp_off[g + 1] - grp_off[g];
    const size_t m1 = (size_t)(maxg ? maxg : 1);
    const char** strs = (const char**)malloc(sizeof(char*) * m1);
    int32_t* lens = (int32_t*)malloc(sizeof(int32_t) * m1);
    int64_t* o1 = (int64_t*)malloc(sizeof(int64_t) * (m1 + 1));
    int64_t* o2 = (int64_t*)malloc(sizeof(int64_t) * (m1 + 1));
    int64_t* oi = (int64_t*)malloc(sizeof(int64_t) * (m1 + 1));
    int64_t* loc_off = (int64_t*)malloc(sizeof(int64_t) * (m1 + 1));
    int32_t* loc = (int32_t*)malloc(sizeof(int32_t) * m1);
    int64_t cap1 = 16 * (int64_t)m1, cap2 = 16 * (int64_t)m1;
    int32_t* n1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap1);
    int32_t* n2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap2);
    int32_t* ni = NULL;
    int64_t capi = 0;
    int rc = 0;

    *nclusters = 0;
    clu_off[0] = 0;
    for (int64_t g = 0; g < ngroups && !rc; ++g) {
        const int32_t* members = grp + grp_off[g];
        const int64_t N = grp_off[g + 1] - grp_off[g];
        if (N == 1) { /* passthrough (src/umi_group.cpp:39-42) */
            clu[clu_off[*nclusters]] = members[0];
            clu_off[*nclusters + 1] = clu_off[*nclusters] + 1;
            ++*nclusters;
            continue;
        }
        for (int64_t s = 0; s < N; ++s) {
            const int64_t id = members[s] - 1;
            strs[s] = umi1 + off1[id];
            lens[s] = (int32_t)(off1[id + 1] - off1[id]);
        }
        if ((rc = neighbours_grow(strs, lens, N, thresh1, o1, &n1, &cap1))) break;
        const int64_t* use_off = o1;
        const int32_t* use = n1;
        if (umi2) {
            for (int64_t s = 0; s < N; ++s) {
                const int64_t id = members[s] - 1;
                strs[s] = umi2 + off2[id];
                lens[s] = (int32_t)(off2[id + 1] - off2[id]);
            }
            if ((rc = neighbours_grow(strs, lens, N, thresh2, o2, &n2, &cap2))) break;
            /* intersection listed in UMI2 order, membership via sorted UMI1 list
             * (src/umi_group.cpp:72,:85-102) */
            if (o2[N] > capi) { capi = o2[N]; ni = (int32_t*)realloc(ni, sizeof(int32_t) * (size_t)(capi ? capi : 1)); }
            oi[0] = 0;
            for (int64_t s = 0; s < N; ++s) {
                int32_t* a = n1 + o1[s];
                const int64_t na = o1[s + 1] - o1[s];
                qsort(a, (size_t)na, sizeof(int32_t), cmp_i32);
                int64_t w = oi[s];
                for (int64_t p = o2[s]; p < o2[s + 1]; ++p)
                    if (bsearch(&n2[p], a, (size_t)na, sizeof(int32_t), cmp_i32)) ni[w++] = n2[p];
                oi[s + 1] = w;
            }
            use_off = oi;
            use = ni;
        }
        int64_t nloc = 0;
        rc = fast_cluster ? orc_cluster_umis_fast(use_off, use, N, &nloc, loc_off, loc)
                          : orc_cluster_umis(use_off, use, N, &nloc, loc_off, loc);
        if (rc) break;
        for (int64_t c = 0; c < nloc; ++c) {
            int64_t w = clu_off[*nclusters];
            for (int64_t p = loc_off[c]; p < loc_off[c + 1]; ++p) clu[w++] = members[loc[p]];
            clu_off[*nclusters + 1] = w;
            ++*nclusters;
        }
    }
    free(strs); free(lens); free(o1); free(o2); free(oi); free(loc_off); free(loc);
    free(n1); free(n2); free(ni);
    return rc;
}
