/*
 * msa2.c -- CPU statement of "MSA spec v2" (DESIGN.md section 5): the consistency-based
 * progressive alignment the reference obtains from SeqAn (T-Coffee;
 * /root/reference/src/quick_msa.cpp:25-35 configures it, :61-67 runs globalMsaAlignment),
 * restated at base resolution.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED.  SeqAn (Bioconductor RSeqAn, no version pin, /root/reference/DESCRIPTION:20) is
 * not in the reference tree or in this image and the reference has no test of
 * quick_msa / multiReadAlign, so nothing here can be checked against the reference's output.  The
 * pipeline below is SeqAn's published one for fewer than ~30 sequences; where SeqAn's exact choice is
 * unknown or cannot be reproduced, the choice made is written down as part of the spec:
 *
 *   1. all pairs a < b: banded global Gotoh, rows = sequence b, columns = sequence a, with spec v1's
 *      recurrences, band and tie rules (orc_msa_pairwise).                    [SeqAn: appendSegmentMatches,
 *      GlobalPairwiseLibrary + Banded]
 *   2. distance d(a,b) = 1 - (aligned pairs with equal bases) / (alignment columns).
 *                                                                              [SeqAn: getAlignmentStatistics]
 *   3. guide tree: neighbour joining on d (Saitou-Nei Q criterion, first minimum in (i, j) order wins,
 *      the joined node takes the lower slot); the last three nodes are joined lowest two first.
 *                                                                              [SeqAn: njTree, MsaOptions::build = 0]
 *   4. primary library at base resolution: every aligned pair (a,p)-(b,q) is an edge of weight
 *      w0 = max(1, score(a_p, b_q)).   [SeqAn: buildAlignmentGraph(..., ReScore): segment score, raised to 1
 *      when not positive; SeqAn refines matches to common segments, here every segment is one base]
 *   5. triplet extension with a bounded library: for an ordered pair of reads (a, b) and a position p of a the
 *      extended library holds at most FOUR partner positions of b --
 *        slot 0: the direct partner q0 = the position aligned to p in the pairwise alignment of a and b (if any),
 *                weight w0(a_p, b_q0) + the sum of min(w0(a_p, c_r), w0(c_r, b_q0)) over the third reads c whose
 *                pairwise alignments link p - r - q0;
 *        slots 1-3: the first MSA2_LIBRARY (3) OTHER positions of b that a triplet p - r - q names, third reads c in
 *                ascending order, each with the sum of its triplets' weights;
 *      triplets that name a further position are ignored.                     [SeqAn: tripletLibraryExtension keeps every
 *      position; bounded library: own rule -- it makes the library a function of (a, p, b) alone, of fixed size, which is
 *      what lets the extension run once per group, outside the progressive merging.  Rounds 2-4 kept every position and
 *      bounded only the ROW; the bound can be moved or the old enumeration switched back for comparison,
 *      orc_msa2_set_library, tools/msa2_rules.py.]
 *      A row (column i of the first child) sums the library over its members: a ascending, b ascending, slots 0 to 3,
 *      weight onto the partner's column.  Bounded rows: a candidate whose partner column is not among the first
 *      MSA2_ROWCAP (16) distinct columns seen in that order is ignored.       [own rule]
 *      Noise filter: of a row's entries those lighter than half the heaviest are dropped -- the partner
 *      columns reached only through reads unrelated to the rest of the cluster.   [own rule]
 *   6. progressive alignment along the tree: two profiles (lists of columns) are merged by the heaviest
 *      common subsequence of their columns, weight(col_i, col_j) = sum of W over the members, no gap
 *      penalties; among equally heavy chains the one built from the earliest matches (row-major) wins;
 *      unmatched columns of the first child precede those of the second between two matched columns.
 *                                                                              [SeqAn: progressiveAlignment,
 *                                                                               heaviestCommonSubsequence]
 *   7. output: one gapped row per read in group order, '-' for gaps.
 */
#include "oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);

#define MSA2_ROWCAP 16

/* The two own rules of step 5 can be switched off, and their effect counted, to bound what they change
 * (tests/test_oracle_msa2_rules.py, tools/msa2_rules.py).  Process-wide; the defaults are the spec. */
static int g_nocap = 0, g_nofilter = 0;
#define MSA2_XMAX 63
#define MSA2_LIBRARY 3
static int g_library = MSA2_LIBRARY;   /* other partner positions kept per (a, p, b) beside the direct one: 3 by the spec; -1: the unbounded library of rounds 2-4 */
enum { ST_JOINS, ST_ROWS, ST_ROWS_WITH_CAND, ST_ROWS_CAPPED, ST_CAND_IGNORED, ST_ENT_BEFORE_FILTER, ST_ENT_FILTERED,
       ST_ROWS_FILTERED, ST_ENT_KEPT, ST_TRIPLES, ST_TRIPLES_2Q, ST_TRIPLES_3Q, ST_TRIPLES_GAPDIRECT, ST_CANDIDATES,
       ST_ROWS_MULTI, ST_MAX_ROW_ENTRIES, ST_LIB_IGNORED, ST_N };
static int64_t g_stats[ST_N];
#define STAT_ADD(k, v) __atomic_fetch_add(&g_stats[k], (int64_t)(v), __ATOMIC_RELAXED)

void orc_msa2_set_rules(int nocap, int nofilter) { g_nocap = nocap; g_nofilter = nofilter; }
void orc_msa2_set_library(int others) { g_library = others > MSA2_XMAX ? MSA2_XMAX : others; }
int orc_msa2_stats(int64_t* out, int64_t cap, int reset) {
    for (int k = 0; k < ST_N && k < cap; ++k) out[k] = __atomic_load_n(&g_stats[k], __ATOMIC_RELAXED);
    if (reset) for (int k = 0; k < ST_N; ++k) __atomic_store_n(&g_stats[k], 0, __ATOMIC_RELAXED);
    return ST_N;
}

static char dna5(char c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': return c;
        case 'a': return 'A'; case 'c': return 'C'; case 'g': return 'G'; case 't': return 'T';
    }
    return 'N';
}

typedef struct {
    int64_t n;
    char* s;            /* Dna5 copies, concatenated */
    int64_t* off;       /* n + 1, relative */
    int32_t** map;      /* map[a * n + b][p] = position of b aligned to position p of a, or -1 (a != b) */
    double* dist;       /* n * n */
} lib_t;

#define LEN(L, r) ((L)->off[(r) + 1] - (L)->off[r])
#define SEQ(L, r) ((L)->s + (L)->off[r])

static void lib_free(lib_t* L) {
    if (L->map) {
        for (int64_t k = 0; k < L->n * L->n; ++k) free(L->map[k]);
        free(L->map);
    }
    free(L->s); free(L->off); free(L->dist);
}

/* steps 1 and 2 */
static int lib_build(lib_t* L, const char* seq, const int64_t* off, int64_t n, int ma, int mm, int go, int ge, int bw) {
    memset(L, 0, sizeof *L);
    L->n = n;
    const int64_t total = off[n] - off[0];
    L->s = (char*)malloc((size_t)(total ? total : 1));
    L->off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    L->map = (int32_t**)calloc((size_t)(n * n), sizeof(int32_t*));
    L->dist = (double*)calloc((size_t)(n * n), sizeof(double));
    if (!L->s || !L->off || !L->map || !L->dist) return orc_fail("out of memory in msa2 oracle");
    for (int64_t k = 0; k < total; ++k) L->s[k] = dna5(seq[off[0] + k]);
    for (int64_t r = 0; r <= n; ++r) L->off[r] = off[r] - off[0];
    int rc = 0;
    for (int64_t a = 0; a < n && !rc; ++a)
        for (int64_t b = a + 1; b < n && !rc; ++b) {
            const int64_t la = LEN(L, a), lb = LEN(L, b);
            int32_t* ins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(la + 1));
            int64_t* alg = (int64_t*)malloc(sizeof(int64_t) * (size_t)(la ? la : 1));
            int32_t* mab = (int32_t*)malloc(sizeof(int32_t) * (size_t)(la ? la : 1));
            int32_t* mba = (int32_t*)malloc(sizeof(int32_t) * (size_t)(lb ? lb : 1));
            if (!ins || !alg || !mab || !mba) { free(ins); free(alg); free(mab); free(mba); return orc_fail("out of memory in msa2 oracle"); }
            /* rows = b, columns = a */
            rc = orc_msa_pairwise(SEQ(L, b), lb, SEQ(L, a), la, ma, mm, go, ge, bw, ins, alg);
            int64_t nequal = 0, ndiag = 0;
            for (int64_t q = 0; q < lb; ++q) mba[q] = -1;
            for (int64_t p = 0; p < la; ++p) {
                mab[p] = (int32_t)alg[p];
                if (alg[p] >= 0) {
                    mba[alg[p]] = (int32_t)p;
                    ++ndiag;
                    if (SEQ(L, a)[p] == SEQ(L, b)[alg[p]]) ++nequal;
                }
            }
            const int64_t alen = la + lb - ndiag;
            const double d = alen > 0 ? 1.0 - (double)nequal / (double)alen : 0.0;
            L->dist[a * n + b] = L->dist[b * n + a] = d;
            L->map[a * n + b] = mab;
            L->map[b * n + a] = mba;
            free(ins); free(alg);
        }
    return rc;
}

/* step 3: joins[2k], joins[2k+1] = node ids merged by join k (first child, second child) */
static void nj_tree(const double* dist, int64_t n, int32_t* joins) {
    double* D = (double*)malloc(sizeof(double) * (size_t)(n * n));
    double* R = (double*)malloc(sizeof(double) * (size_t)n);
    int* active = (int*)malloc(sizeof(int) * (size_t)n);
    int32_t* node = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(D, dist, sizeof(double) * (size_t)(n * n));
    for (int64_t i = 0; i < n; ++i) { active[i] = 1; node[i] = (int32_t)i; }
    int64_t r = n, nj = 0;
    while (r > 3) {
        for (int64_t i = 0; i < n; ++i) {
            if (!active[i]) continue;
            double sum = 0.0;
            for (int64_t k = 0; k < n; ++k)
                if (active[k] && k != i) sum = sum + D[i * n + k];
            R[i] = sum;
        }
        int64_t bi = -1, bj = -1;
        double best = 0.0;
        const double rm2 = (double)(r - 2);
        for (int64_t i = 0; i < n; ++i) {
            if (!active[i]) continue;
            for (int64_t j = i + 1; j < n; ++j) {
                if (!active[j]) continue;
                const double q = (rm2 * D[i * n + j] - R[i]) - R[j];
                if (bi < 0 || q < best) { best = q; bi = i; bj = j; }
            }
        }
        joins[2 * nj] = node[bi]; joins[2 * nj + 1] = node[bj];
        const double dij = D[bi * n + bj];
        for (int64_t k = 0; k < n; ++k) {
            if (!active[k] || k == bi || k == bj) continue;
            const double v = ((D[bi * n + k] + D[bj * n + k]) - dij) * 0.5;
            D[bi * n + k] = D[k * n + bi] = v;
        }
        active[bj] = 0;
        node[bi] = (int32_t)(n + nj);
        ++nj; --r;
    }
    /* the remaining two or three nodes: lowest slots first */
    int64_t l[3], c = 0;
    for (int64_t i = 0; i < n && c < 3; ++i) if (active[i]) l[c++] = i;
    if (c >= 2) {
        joins[2 * nj] = node[l[0]]; joins[2 * nj + 1] = node[l[1]];
        node[l[0]] = (int32_t)(n + nj);
        ++nj;
    }
    if (c == 3) {
        joins[2 * nj] = node[l[0]]; joins[2 * nj + 1] = node[l[2]];
        ++nj;
    }
    free(D); free(R); free(active); free(node);
}

static int w0(char x, char y, int ma, int mm) {
    const int s = (x == y) ? ma : mm;
    return s > 1 ? s : 1;
}

typedef struct {
    int64_t nmem;
    int32_t* mem;       /* member sequences (indices inside the group) */
    int64_t ncols;
} prof_t;

int orc_msa2_tree(const char* seq, const int64_t* off, int64_t n, int ma, int mm, int go, int ge, int bw,
                  int32_t* joins, double* dist) {
    if (n < 2) return 0;
    lib_t L;
    int rc = lib_build(&L, seq, off, n, ma, mm, go, ge, bw);
    if (!rc) {
        nj_tree(L.dist, n, joins);
        if (dist) memcpy(dist, L.dist, sizeof(double) * (size_t)(n * n));
    }
    lib_free(&L);
    return rc;
}

int orc_msa2_group(const char* seq, const int64_t* off, int64_t n, int ma, int mm, int go, int ge, int bw,
                   char* out, int64_t cap, int64_t* width) {
    *width = 0;
    if (n == 0) return 0;
    if (n == 1) { /* verbatim (quick_msa.cpp:46-50) */
        const int64_t Ln = off[1] - off[0];
        if (Ln > cap) return orc_fail("msa output buffer too small");
        memcpy(out, seq + off[0], (size_t)Ln);
        *width = Ln;
        return 0;
    }
    lib_t L;
    int rc = lib_build(&L, seq, off, n, ma, mm, go, ge, bw);
    if (rc) { lib_free(&L); return rc; }
    int32_t* joins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * (n - 1)));
    nj_tree(L.dist, n, joins);

    const int64_t total = L.off[n];
    /* col[off[a] + p] = column of position p of sequence a in the profile that currently holds a */
    int64_t* col = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total ? total : 1));
    prof_t* prof = (prof_t*)calloc((size_t)(2 * n - 1), sizeof(prof_t));
    for (int64_t a = 0; a < n; ++a) {
        prof[a].nmem = 1;
        prof[a].mem = (int32_t*)malloc(sizeof(int32_t));
        prof[a].mem[0] = (int32_t)a;
        prof[a].ncols = LEN(&L, a);
        for (int64_t p = 0; p < LEN(&L, a); ++p) col[L.off[a] + p] = p;
    }
    for (int64_t k = 0; k + 1 < n && !rc; ++k) {
        prof_t* A = &prof[joins[2 * k]];
        prof_t* B = &prof[joins[2 * k + 1]];
        prof_t* P = &prof[n + k];
        const int64_t nA = A->ncols, nB = B->ncols;
        /* position of member a at column i of A (or -1) */
        int64_t* posA = (int64_t*)malloc(sizeof(int64_t) * (size_t)(A->nmem * (nA ? nA : 1)));
        for (int64_t x = 0; x < A->nmem * nA; ++x) posA[x] = -1;
        for (int64_t u = 0; u < A->nmem; ++u) {
            const int64_t a = A->mem[u];
            for (int64_t p = 0; p < LEN(&L, a); ++p) posA[u * nA + col[L.off[a] + p]] = p;
        }
        /* matches, row-major; per row at most MSA2_ROWCAP partner columns, in order of first appearance */
        int64_t mcap = 1024, nm = 0;
        int64_t* mi = (int64_t*)malloc(sizeof(int64_t) * (size_t)mcap);
        int64_t* mj = (int64_t*)malloc(sizeof(int64_t) * (size_t)mcap);
        int64_t* mw = (int64_t*)malloc(sizeof(int64_t) * (size_t)mcap);
        uint64_t inA = 0, inB = 0;   /* member sets (spec v2 takes groups of up to 64 reads) */
        for (int64_t u = 0; u < A->nmem; ++u) inA |= (uint64_t)1 << A->mem[u];
        for (int64_t v = 0; v < B->nmem; ++v) inB |= (uint64_t)1 << B->mem[v];
        int64_t* idxA = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);   /* member slot of sequence a in posA */
        for (int64_t u = 0; u < A->nmem; ++u) idxA[A->mem[u]] = u;
        /* a row's list: MSA2_ROWCAP entries by the spec; every candidate column when the cap is switched off */
        const int64_t rowcap = g_nocap ? A->nmem * B->nmem * (n + 1) + 1 : MSA2_ROWCAP;
        int64_t* lj = (int64_t*)malloc(sizeof(int64_t) * (size_t)rowcap);
        int64_t* lw = (int64_t*)malloc(sizeof(int64_t) * (size_t)rowcap);
        STAT_ADD(ST_JOINS, 1);
        STAT_ADD(ST_ROWS, nA);
        for (int64_t i = 0; i < nA; ++i) {
            int64_t cnt = 0, ignored = 0, ncand = 0;
#define ADD(J, Wt) do { int64_t k_; ++ncand; for (k_ = 0; k_ < cnt; ++k_) if (lj[k_] == (J)) { lw[k_] += (Wt); break; } \
                        if (k_ == cnt) { if (cnt < rowcap) { lj[cnt] = (J); lw[cnt] = (Wt); ++cnt; } else ++ignored; } } while (0)
            for (int64_t a = 0; a < n; ++a) {
                if (!((inA >> a) & 1u)) continue;
                const int64_t p = posA[idxA[a] * nA + i];
                if (p < 0) continue;
                const char xa = SEQ(&L, a)[p];
                if (g_library < 0) {   /* the enumeration of rounds 2-4 (kept for tools/msa2_rules.py): unbounded library */
                    for (int64_t b = 0; b < n; ++b) {          /* direct edges */
                        if (!((inB >> b) & 1u)) continue;
                        const int64_t q = L.map[a * n + b][p];
                        if (q < 0) continue;
                        ADD(col[L.off[b] + q], w0(xa, SEQ(&L, b)[q], ma, mm));
                    }
                    for (int64_t c = 0; c < n; ++c) {          /* through sequence c */
                        if (c == a) continue;
                        const int64_t r = L.map[a * n + c][p];
                        if (r < 0) continue;
                        const int w1 = w0(xa, SEQ(&L, c)[r], ma, mm);
                        for (int64_t b = 0; b < n; ++b) {
                            if (!((inB >> b) & 1u) || b == c) continue;
                            const int64_t q = L.map[c * n + b][r];
                            if (q < 0) continue;
                            const int w2 = w0(SEQ(&L, c)[r], SEQ(&L, b)[q], ma, mm);
                            ADD(col[L.off[b] + q], w1 < w2 ? w1 : w2);
                        }
                    }
                    continue;
                }
                for (int64_t b = 0; b < n; ++b) {
                    if (!((inB >> b) & 1u)) continue;
                    /* the extended library of (a, p) towards b: slot 0 = the direct partner, slots 1 .. = the first other
                     * positions of b named by a triplet, in order of c */
                    int64_t eq[1 + MSA2_XMAX], ew[1 + MSA2_XMAX];
                    int64_t ne = 1, ndist = 0;
                    eq[0] = L.map[a * n + b][p];
                    ew[0] = eq[0] >= 0 ? w0(xa, SEQ(&L, b)[eq[0]], ma, mm) : 0;
                    for (int64_t c = 0; c < n; ++c) {
                        if (c == a || c == b) continue;
                        const int64_t r = L.map[a * n + c][p];
                        if (r < 0) continue;
                        const int64_t q = L.map[c * n + b][r];
                        if (q < 0) continue;
                        const int w1 = w0(xa, SEQ(&L, c)[r], ma, mm);
                        const int w2 = w0(SEQ(&L, c)[r], SEQ(&L, b)[q], ma, mm);
                        const int w = w1 < w2 ? w1 : w2;
                        int64_t k;
                        for (k = 0; k < ne; ++k) if (eq[k] == q) { ew[k] += w; break; }
                        if (k == ne) {
                            ++ndist;
                            if (ne < 1 + g_library) { eq[ne] = q; ew[ne] = w; ++ne; }
                            else STAT_ADD(ST_LIB_IGNORED, 1);
                        }
                    }
                    STAT_ADD(ST_TRIPLES, 1);
                    if (ndist + (eq[0] >= 0) >= 2) STAT_ADD(ST_TRIPLES_2Q, 1);
                    if (ndist + (eq[0] >= 0) >= 3) STAT_ADD(ST_TRIPLES_3Q, 1);
                    if (eq[0] < 0 && ndist) STAT_ADD(ST_TRIPLES_GAPDIRECT, 1);
                    for (int64_t k = 0; k < ne; ++k)
                        if (eq[k] >= 0) ADD(col[L.off[b] + eq[k]], ew[k]);
                }
            }
#undef ADD
            STAT_ADD(ST_CANDIDATES, ncand);
            if (cnt) STAT_ADD(ST_ROWS_WITH_CAND, 1);
            if (ignored) { STAT_ADD(ST_ROWS_CAPPED, 1); STAT_ADD(ST_CAND_IGNORED, ignored); }
            STAT_ADD(ST_ENT_BEFORE_FILTER, cnt);
            /* noise filter */
            if (!g_nofilter) {
                int64_t wmax = 0;
                int64_t kept = 0;
                for (int64_t x = 0; x < cnt; ++x) if (lw[x] > wmax) wmax = lw[x];
                for (int64_t x = 0; x < cnt; ++x)
                    if (2 * lw[x] >= wmax) { lj[kept] = lj[x]; lw[kept] = lw[x]; ++kept; }
                if (kept < cnt) { STAT_ADD(ST_ROWS_FILTERED, 1); STAT_ADD(ST_ENT_FILTERED, cnt - kept); }
                cnt = kept;
            }
            STAT_ADD(ST_ENT_KEPT, cnt);
            if (cnt > 1) STAT_ADD(ST_ROWS_MULTI, 1);
            {
                int64_t cur = __atomic_load_n(&g_stats[ST_MAX_ROW_ENTRIES], __ATOMIC_RELAXED);
                while (cnt > cur && !__atomic_compare_exchange_n(&g_stats[ST_MAX_ROW_ENTRIES], &cur, cnt, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
            /* by column */
            for (int64_t x = 1; x < cnt; ++x) {
                const int64_t tj = lj[x], tw = lw[x];
                int64_t y = x - 1;
                while (y >= 0 && lj[y] > tj) { lj[y + 1] = lj[y]; lw[y + 1] = lw[y]; --y; }
                lj[y + 1] = tj; lw[y + 1] = tw;
            }
            for (int64_t x = 0; x < cnt; ++x) {
                if (nm == mcap) {
                    mcap *= 2;
                    mi = (int64_t*)realloc(mi, sizeof(int64_t) * (size_t)mcap);
                    mj = (int64_t*)realloc(mj, sizeof(int64_t) * (size_t)mcap);
                    mw = (int64_t*)realloc(mw, sizeof(int64_t) * (size_t)mcap);
                }
                mi[nm] = i; mj[nm] = lj[x]; mw[nm] = lw[x];
                ++nm;
            }
        }
        free(idxA); free(lj); free(lw);
        /* heaviest chain: f(m) = w(m) + best f over matches with smaller row and smaller column;
         * "best" = larger f, then smaller match index.  bestAt[j] = best match ending in column j among the
         * processed rows; a row's matches all look at the state before the row. */
        int64_t* f = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nm ? nm : 1));
        int64_t* pred = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nm ? nm : 1));
        int64_t* bestAt = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nB ? nB : 1));
        for (int64_t j = 0; j < nB; ++j) bestAt[j] = -1;
#define BETTER(x, y) ((y) < 0 || ((x) >= 0 && (f[x] > f[y] || (f[x] == f[y] && (x) < (y)))))
        for (int64_t m0 = 0; m0 < nm;) {
            int64_t m1 = m0;
            while (m1 < nm && mi[m1] == mi[m0]) ++m1;
            for (int64_t m = m0; m < m1; ++m) {
                int64_t b = -1;
                for (int64_t j = 0; j < mj[m]; ++j)
                    if (bestAt[j] >= 0 && BETTER(bestAt[j], b)) b = bestAt[j];
                pred[m] = b;
                f[m] = mw[m] + (b >= 0 ? f[b] : 0);
            }
            for (int64_t m = m0; m < m1; ++m)
                if (BETTER(m, bestAt[mj[m]])) bestAt[mj[m]] = m;
            m0 = m1;
        }
        int64_t tail = -1;
        for (int64_t j = 0; j < nB; ++j)
            if (bestAt[j] >= 0 && BETTER(bestAt[j], tail)) tail = bestAt[j];
#undef BETTER
        if (getenv("ORC_MSA2_DEBUG")) {
            fprintf(stderr, "ORC round %d: join %d %d nA %d nB %d\n", (int)k, joins[2 * k], joins[2 * k + 1], (int)nA, (int)nB);
            int64_t* pdbg = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nA ? nA : 1));
            for (int64_t i = 0; i < nA; ++i) pdbg[i] = -1;
            for (int64_t m = tail; m >= 0; m = pred[m]) pdbg[mi[m]] = mj[m];
            int64_t m = 0;
            for (int64_t i = 0; i < nA; ++i) {
                fprintf(stderr, "  row %d part %d :", (int)i, (int)pdbg[i]);
                for (; m < nm && mi[m] == i; ++m) fprintf(stderr, " (%d w %d f %d)", (int)mj[m], (int)mw[m], (int)f[m]);
                fprintf(stderr, "\n");
            }
            free(pdbg);
        }
        /* matched column pairs, ascending */
        int64_t* pa = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nA ? nA : 1));   /* partner column in B of column i of A, or -1 */
        for (int64_t i = 0; i < nA; ++i) pa[i] = -1;
        for (int64_t m = tail; m >= 0; m = pred[m]) pa[mi[m]] = mj[m];
        /* new column numbers: unmatched columns of A, then of B, then the merged column */
        int64_t* ncA = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nA ? nA : 1));
        int64_t* ncB = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nB ? nB : 1));
        int64_t nextB = 0, cnum = 0;
        for (int64_t i = 0; i < nA; ++i) {
            if (pa[i] < 0) { ncA[i] = cnum++; continue; }
            /* all unmatched A columns before i were numbered; now B's columns before its partner */
            while (nextB < pa[i]) ncB[nextB++] = cnum++;
            ncA[i] = cnum; ncB[nextB++] = cnum; ++cnum;
        }
        while (nextB < nB) ncB[nextB++] = cnum++;
        /* (in the tail after the last matched pair too: A's columns were numbered inside the loop) */
        P->ncols = cnum;
        P->nmem = A->nmem + B->nmem;
        P->mem = (int32_t*)malloc(sizeof(int32_t) * (size_t)P->nmem);
        memcpy(P->mem, A->mem, sizeof(int32_t) * (size_t)A->nmem);
        memcpy(P->mem + A->nmem, B->mem, sizeof(int32_t) * (size_t)B->nmem);
        for (int64_t u = 0; u < A->nmem; ++u) {
            const int64_t a = A->mem[u];
            for (int64_t p = 0; p < LEN(&L, a); ++p) col[L.off[a] + p] = ncA[col[L.off[a] + p]];
        }
        for (int64_t v = 0; v < B->nmem; ++v) {
            const int64_t b = B->mem[v];
            for (int64_t q = 0; q < LEN(&L, b); ++q) col[L.off[b] + q] = ncB[col[L.off[b] + q]];
        }
        free(posA); free(mi); free(mj); free(mw); free(f); free(pred); free(bestAt); free(pa); free(ncA); free(ncB);
    }
    if (!rc) {
        const int64_t Wd = prof[2 * n - 2].ncols;
        if (Wd * n > cap) rc = orc_fail("msa output buffer too small");
        else {
            memset(out, '-', (size_t)(Wd * n));
            for (int64_t a = 0; a < n; ++a)
                for (int64_t p = 0; p < LEN(&L, a); ++p) out[a * Wd + col[L.off[a] + p]] = SEQ(&L, a)[p];
            *width = Wd;
        }
    }
    for (int64_t x = 0; x < 2 * n - 1; ++x) free(prof[x].mem);
    free(prof); free(col); free(joins);
    lib_free(&L);
    return rc;
}
