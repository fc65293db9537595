/*
 * consensus.c -- oracle restatement of sarlacc's per-column consensus vote.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows (file:line relative to /root/reference):
 *   src/create_consensus.cpp:18-32    log-error -> Phred+33 character
 *   src/create_consensus.cpp:61-135   count-based vote
 *   src/create_consensus.cpp:178-272  quality-weighted vote with log-sum-exp error
 *   src/DNA_input.cpp:90-104          all rows must have equal width
 *
 * R::log1pexp is R's nmath routine (src/nmath/plogis.c in R >= 3.x): it is not
 * part of the reference tree; restated from its published definition
 *   x <= 18 -> log1p(exp(x));  18 < x <= 33.3 -> x + exp(-x);  else x.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static const double MAX_ERR = 0.99999999, MIN_ERR = 0.00000001;

static double r_log1pexp(double x) {
    if (x <= 18.) return log1p(exp(x));
    if (x > 33.3) return x;
    return x + exp(-x);
}

void orc_errors_to_string(const double* lerr, int64_t n, char* out) {
    for (int64_t i = 0; i < n; ++i) {
        double q = round(-10 * lerr[i] / log(10));
        if (q > 93.0) q = 93.0;
        const int a = (int)q;
        out[i] = (char)(a + 33);
    }
    out[n] = '\0';
}

static int alignment_width(const int64_t* off, int64_t nrows, int64_t* width) {
    *width = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t w = off[r + 1] - off[r];
        if (r == 0) *width = w;
        else if (w != *width) return orc_fail("alignment strings should have the same length");
    }
    return 0;
}

static int first_max4(const double* s) {
    int best = 0;
    for (int b = 1; b < 4; ++b) if (s[b] > s[best]) best = b;
    return best;
}

int orc_consensus_basic(const char* aln, const int64_t* off, int64_t nrows,
                        double mincov, double pseudo,
                        char* cons, double* lerr, int64_t* conlen) {
    int64_t W;
    if (alignment_width(off, nrows, &W)) return 1;
    double* score = (double*)calloc((size_t)(W ? W : 1) * 4, sizeof(double));
    int* present = (int*)calloc((size_t)(W ? W : 1), sizeof(int));
    const double pseudo_num = pseudo / 4;
    int rc = 0;

    for (int64_t r = 0; r < nrows && !rc; ++r) {
        const char* row = aln + off[r];
        for (int64_t i = 0; i < W; ++i) {
            const char c = row[i];
            if (c == '-') continue;
            ++present[i];
            if (c == 'N') continue;
            int b = -1;
            switch (c) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; }
            if (b < 0) {
                char msg[96];
                snprintf(msg, sizeof msg, "unknown character '%c' in alignment string", c);
                rc = orc_fail(msg);
                break;
            }
            score[i * 4 + b] += 1;
        }
    }
    int64_t k = 0;
    for (int64_t i = 0; i < W && !rc; ++i) {
        if (present[i] < (double)nrows * mincov) continue;
        const double* s = score + i * 4;
        const int best = first_max4(s);
        cons[k] = BASES[best];
        /* the reference sums with an int accumulator (std::accumulate(..., 0));
         * counts are whole numbers so the truncation is exact */
        int total_i = 0;
        for (int b = 0; b < 4; ++b) total_i = (int)(total_i + s[b]);
        const double total = total_i;
        const double p = (s[best] + pseudo_num) / (total + pseudo);
        lerr[k] = log1p(-p);
        ++k;
    }
    cons[k] = '\0';
    *conlen = k;
    free(score);
    free(present);
    return rc;
}

static int cmp_dbl(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

int orc_consensus_quality(const char* aln, const int64_t* off, int64_t nrows,
                          const char* qual, const int64_t* qoff, int64_t nquals,
                          double mincov,
                          const double* errors, const char* names, int nenc,
                          char* cons, double* lerr, int64_t* conlen) {
    if (orc_check_encoding(errors, names, nenc)) return 1;
    const int offset = (int)names[0];
    int64_t W;
    if (alignment_width(off, nrows, &W)) return 1;
    if (nquals != nrows) return orc_fail("alignments and qualities have different numbers of entries");

    double* score = (double*)calloc((size_t)(W ? W : 1) * 4, sizeof(double));
    int* present = (int*)calloc((size_t)(W ? W : 1), sizeof(int));
    int rc = 0;

    for (int64_t r = 0; r < nrows && !rc; ++r) {
        const char* row = aln + off[r];
        const char* q = qual + qoff[r];
        const int64_t qlen = qoff[r + 1] - qoff[r];
        int64_t pos = 0;
        for (int64_t i = 0; i < W; ++i) {
            const char c = row[i];
            if (c == '-') continue;
            ++present[i];
            if (pos >= qlen) { rc = orc_fail("quality vector is shorter than the alignment sequence"); break; }
            if (c == 'N') { ++pos; continue; }
            if ((int)q[pos] < offset) { rc = orc_fail("quality cannot be lower than smallest encoded value"); break; }
            int qi = (int)q[pos] - offset;
            if (qi >= nenc) qi = nenc - 1;
            double e = errors[qi];
            if (e > MAX_ERR) e = MAX_ERR;
            else if (e < MIN_ERR) e = MIN_ERR;
            const double right = log1p(-e);
            const double wrong = log(e / 3);
            ++pos;
            for (int b = 0; b < 4; ++b) score[i * 4 + b] += (c == BASES[b]) ? right : wrong;
        }
        if (!rc && pos != qlen) rc = orc_fail("quality vector is longer than the alignment sequence");
    }

    int64_t k = 0;
    for (int64_t i = 0; i < W && !rc; ++i) {
        if (present[i] < (double)nrows * mincov) continue;
        double s[4];
        memcpy(s, score + i * 4, sizeof s);
        cons[k] = BASES[first_max4(s)];
        qsort(s, 4, sizeof(double), cmp_dbl);
        double denom = s[0], err3 = 0;
        for (int b = 1; b < 4; ++b) {
            denom += r_log1pexp(s[b] - denom);
            if (b == 2) err3 = denom;
        }
        lerr[k] = err3 - denom;
        ++k;
    }
    cons[k] = '\0';
    *conlen = k;
    free(score);
    free(present);
    return rc;
}
