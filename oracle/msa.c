/*
 * msa.c -- CPU statement of "MSA spec v1" (DESIGN.md), the per-group multiple
 * alignment used in place of the reference's call into SeqAn's T-Coffee
 * (/root/reference/src/quick_msa.cpp:15-80).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED: SeqAn (Bioconductor RSeqAn, no version pin in
 * /root/reference/DESCRIPTION:20) is not in the reference tree or this image and
 * the reference has no test of quick_msa/multiReadAlign.  This file therefore
 * does NOT restate SeqAn; it defines the deterministic algorithm our HIP kernels
 * implement, and is the checker for those kernels only.  What is kept from the
 * reference is the documented contract (quick_msa.cpp:25-35,:44-50,:69-75):
 * integer simple scores, global banded pairwise alignment, one gapped row per
 * read in group order, equal widths, '-' gaps, non-ACGT shown as N, singleton
 * groups verbatim.
 *
 * Spec v1 (centre-star):
 *   centre  = lower median of the reads ordered by (length, index);
 *   pairwise = banded global Gotoh, gap of length k scores open + (k-1)*ext
 *              (SeqAn's convention), band on diagonals j-i in
 *              [min(0,lc-lr)-bw, max(0,lc-lr)+bw];
 *   ties     = H: diagonal >= vertical >= horizontal; gap states: open >= extend;
 *   merge    = insertions before centre position p are left-justified in
 *              max-over-reads columns; "once a gap, always a gap".
 */
#include "oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);

#define NEG (INT_MIN / 4)

static char dna5(char c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': return c;
        case 'a': return 'A'; case 'c': return 'C'; case 'g': return 'G'; case 't': return 'T';
    }
    return 'N';
}

typedef struct { int64_t len; int64_t idx; } lenidx;
static int cmp_lenidx(const void* a, const void* b) {
    const lenidx* x = (const lenidx*)a;
    const lenidx* y = (const lenidx*)b;
    if (x->len != y->len) return (x->len > y->len) - (x->len < y->len);
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* Pairwise alignment of read r (rows) against centre c (columns).
 * Output: for every centre position p in [0,lc]: ins_cnt[p] read chars inserted
 * before p; aligned[p] (p<lc) = read position matched to centre p or -1.
 * Insertions are consecutive read positions ending just before the next
 * matched/after position, so only counts are needed plus the walk order. */
int orc_msa_pairwise(const char* r, int64_t lr, const char* c, int64_t lc,
                    int ma, int mm, int go, int ge, int bw,
                    int32_t* ins_cnt, int64_t* aligned) {
    /* Band cap (MSA spec, both versions): a pair is aligned inside at most ORC_MSA_MAXBAND diagonals.
     * Where |lc - lr| + 2 bw + 1 exceeds that, the bandwidth of THIS pair shrinks to the largest that
     * fits; where even |lc - lr| + 1 does not fit (reads differing by >= 1024 bases), the pair gets the
     * diagonal alignment: position p of the read opposite position p of the centre, the tail of the
     * longer sequence unaligned.  (SeqAn's banded alignment is undefined for a band that does not
     * contain both corners; the reference has no rule to follow here.) */
    {
        const int64_t dl = lc > lr ? lc - lr : lr - lc;
        if (dl + 2 * (int64_t)bw + 1 > ORC_MSA_MAXBAND) bw = (int)((ORC_MSA_MAXBAND - 1 - dl) >= 0 ? (ORC_MSA_MAXBAND - 1 - dl) / 2 : -1);
        if (bw < 0) {
            const int64_t k = lr < lc ? lr : lc;
            for (int64_t p = 0; p <= lc; ++p) ins_cnt[p] = 0;
            for (int64_t p = 0; p < lc; ++p) aligned[p] = p < k ? p : -1;
            if (lr > lc) ins_cnt[lc] = (int32_t)(lr - lc);
            return 0;
        }
    }
    const int64_t dlo = (lc - lr < 0 ? lc - lr : 0) - bw;
    const int64_t dhi = (lc - lr > 0 ? lc - lr : 0) + bw;
    const int64_t B = dhi - dlo + 1; /* band width */
    /* row-major band storage: cell (i,j) at i*B + (j-i-dlo) */
    const size_t cells = (size_t)(lr + 1) * (size_t)B;
    int* H = (int*)malloc(sizeof(int) * cells);
    int* E = (int*)malloc(sizeof(int) * cells);
    int* F = (int*)malloc(sizeof(int) * cells);
    unsigned char* T = (unsigned char*)malloc(cells);
    if (!H || !E || !F || !T) { free(H); free(E); free(F); free(T); return orc_fail("out of memory in msa oracle"); }
#define AT(i, j) ((size_t)(i) * (size_t)B + (size_t)((j) - (i) - dlo))
#define INB(i, j) ((j) - (i) >= dlo && (j) - (i) <= dhi && (j) >= 0 && (j) <= lc)
    for (int64_t i = 0; i <= lr; ++i) {
        int64_t jlo = i + dlo, jhi = i + dhi;
        if (jlo < 0) jlo = 0;
        if (jhi > lc) jhi = lc;
        for (int64_t j = jlo; j <= jhi; ++j) {
            int e = NEG, f = NEG, d = NEG;
            unsigned char t = 0;
            if (i > 0 && INB(i - 1, j)) {
                const int o = H[AT(i - 1, j)] + go, x = E[AT(i - 1, j)] + ge;
                if (o >= x) { e = o; t |= 4; } else e = x;
            }
            if (j > 0 && INB(i, j - 1)) {
                const int o = H[AT(i, j - 1)] + go, x = F[AT(i, j - 1)] + ge;
                if (o >= x) { f = o; t |= 8; } else f = x;
            }
            if (i > 0 && j > 0 && INB(i - 1, j - 1))
                d = H[AT(i - 1, j - 1)] + (r[i - 1] == c[j - 1] ? ma : mm);
            int h;
            if (i == 0 && j == 0) { h = 0; }
            else if (d >= e && d >= f) { h = d; }
            else if (e >= f) { h = e; t |= 1; }
            else { h = f; t |= 2; }
            if (e < NEG) e = NEG;
            if (f < NEG) f = NEG;
            if (h < NEG) h = NEG;
            H[AT(i, j)] = h; E[AT(i, j)] = e; F[AT(i, j)] = f; T[AT(i, j)] = t;
        }
    }
    /* traceback */
    for (int64_t p = 0; p <= lc; ++p) ins_cnt[p] = 0;
    for (int64_t p = 0; p < lc; ++p) aligned[p] = -1;
    int64_t i = lr, j = lc;
    int state = 0; /* 0 = H, 1 = E (vertical), 2 = F (horizontal) */
    while (i > 0 || j > 0) {
        const unsigned char t = T[AT(i, j)];
        if (state == 0) {
            state = t & 3;
            if (state == 0) { aligned[j - 1] = i - 1; --i; --j; }
            continue;
        }
        if (state == 1) { /* read char i-1 inserted before centre position j */
            ++ins_cnt[j];
            state = (t & 4) ? 0 : 1;
            --i;
        } else {          /* centre char j-1 opposite a gap */
            state = (t & 8) ? 0 : 2;
            --j;
        }
    }
#undef AT
#undef INB
    free(H); free(E); free(F); free(T);
    return 0;
}

int orc_msa_group(const char* seq, const int64_t* off, int64_t m,
                  int match, int mismatch, int gapopen, int gapext, int bandwidth,
                  char* out, int64_t cap, int64_t* width) {
    *width = 0;
    if (m == 0) return 0;
    if (m == 1) { /* verbatim (quick_msa.cpp:46-50) */
        const int64_t L = off[1] - off[0];
        if (L > cap) return orc_fail("msa output buffer too small");
        memcpy(out, seq + off[0], (size_t)L);
        *width = L;
        return 0;
    }
    /* Dna5 copies */
    const int64_t total = off[m] - off[0];
    char* s = (char*)malloc((size_t)(total ? total : 1));
    for (int64_t k = 0; k < total; ++k) s[k] = dna5(seq[off[0] + k]);
#define RD(r) (s + (off[r] - off[0]))
#define LEN(r) (off[(r) + 1] - off[r])

    lenidx* order = (lenidx*)malloc(sizeof(lenidx) * (size_t)m);
    for (int64_t r = 0; r < m; ++r) { order[r].len = LEN(r); order[r].idx = r; }
    qsort(order, (size_t)m, sizeof(lenidx), cmp_lenidx);
    const int64_t ctr = order[(m - 1) / 2].idx;
    free(order);
    const int64_t lc = LEN(ctr);

    int32_t* ins = (int32_t*)calloc((size_t)m * (size_t)(lc + 1), sizeof(int32_t));
    int64_t* alg = (int64_t*)malloc(sizeof(int64_t) * (size_t)m * (size_t)(lc ? lc : 1));
    int32_t* maxins = (int32_t*)calloc((size_t)(lc + 1), sizeof(int32_t));
    int rc = 0;
    for (int64_t r = 0; r < m && !rc; ++r) {
        if (r == ctr) {
            for (int64_t p = 0; p < lc; ++p) alg[r * lc + p] = p;
            continue;
        }
        rc = orc_msa_pairwise(RD(r), LEN(r), RD(ctr), lc, match, mismatch, gapopen, gapext, bandwidth,
                      ins + r * (lc + 1), alg + r * (lc ? lc : 1));
        for (int64_t p = 0; p <= lc; ++p)
            if (ins[r * (lc + 1) + p] > maxins[p]) maxins[p] = ins[r * (lc + 1) + p];
    }
    int64_t W = lc;
    for (int64_t p = 0; p <= lc; ++p) W += maxins[p];
    if (!rc && W * m > cap) rc = orc_fail("msa output buffer too small");
    if (!rc) {
        for (int64_t r = 0; r < m; ++r) {
            char* row = out + r * W;
            int64_t w = 0, rp = 0; /* rp = next unread position of read r */
            for (int64_t p = 0; p <= lc; ++p) {
                const int32_t k = ins[r * (lc + 1) + p];
                for (int32_t x = 0; x < k; ++x) row[w++] = RD(r)[rp++];
                for (int32_t x = k; x < maxins[p]; ++x) row[w++] = '-';
                if (p < lc) {
                    if (alg[r * lc + p] >= 0) row[w++] = RD(r)[rp++];
                    else row[w++] = '-';
                }
            }
        }
        *width = W;
    }
#undef RD
#undef LEN
    free(ins); free(alg); free(maxins); free(s);
    return rc;
}
