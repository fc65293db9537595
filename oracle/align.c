/*
 * align.c -- oracle restatement of sarlacc's quality-weighted affine-gap DP.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows (file:line relative to /root/reference):
 *   src/quality_encoding.cpp:5-47     encoding validation, to_error
 *   src/reference_align.cpp:21-52     cost tables
 *   src/reference_align.cpp:54-181    column-major DP with jump-length directions
 *   src/reference_align.cpp:184-225   per-cell cost lookup
 *   src/reference_align.cpp:231-278   backtrack
 *   src/reference_align.cpp:280-351   reference->query map and interval lookup
 *   src/reference_align.cpp:353-389   gapped strings
 *   src/adaptor_align.cpp, barcode_align.cpp, general_align.cpp  batch loops
 *   src/mask_bad_bases.cpp:10-52
 *
 * Written from the algorithm description in SURVEY.md Appendix A; the operation
 * order of every fp64 add/sub/compare is kept so that results are bit-identical.
 * Compile with -ffp-contract=off.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
int orc_fail(const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return 1;
}

/* ------------------------------------------------------------------ */
/* quality encoding: names must be consecutive single chars, errors non-increasing
 * (src/quality_encoding.cpp:5-33).  Multi-character names cannot be expressed
 * through this flat interface; the host layer rejects them before calling. */
int orc_check_encoding(const double* errors, const char* names, int n) {
    if (n <= 0) return orc_fail("encoding vector must be non-empty and named");
    for (int i = 1; i < n; ++i) {
        /* `curval != last + 1` compares a char with an int (src/quality_encoding.cpp:21): on a platform whose char is signed
         * (x86) a table that runs past byte 127 -- Biostrings' PhredQuality to Q 99 is '!' .. byte 132 -- is rejected here */
        if ((int)(signed char)names[i] != (int)(signed char)names[i - 1] + 1)
            return orc_fail("names of encoding vector should increase consecutively");
        if (errors[i] > errors[i - 1])
            return orc_fail("error probabilities should decrease");
    }
    return 0;
}

/* (src/quality_encoding.cpp:39-47).  The reference's upper bound test is
 * 'i > size' (one past the end is UB there, SURVEY App.B Q6); we clamp at
 * size-1, which is what every reachable Phred input does anyway. */
static int to_error(const double* errors, int offset, int n, char q, double* out) {
    if ((int)q < offset) return orc_fail("quality cannot be lower than smallest encoded value");
    int i = (int)q - offset;
    if (i >= n) i = n - 1;
    *out = errors[i];
    return 0;
}

/* (src/reference_align.cpp:21-52) log2 of the per-quality (mis)match odds for
 * degeneracy 1..4. */
int orc_cost_tables(const double* errors, int n, double* match, double* mismatch) {
    const double four = 4.0;
    const double ratio = four / (four - 1.0);
    for (int m = 0; m < 4; ++m) {
        const double g = 1.0 / (m + 1.0);
        const double g1 = 1 - g;
        for (int j = 0; j < n; ++j) {
            const double e = errors[j];
            match[m * n + j] = log(g * (1 - e) * four + g1 * e * ratio) / M_LN2;
            mismatch[m * n + j] = log(g1 * (1 - e) * four + g * e * ratio) / M_LN2;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
typedef struct {
    int R;
    const char* ref;
    double GO, GE;        /* internal: open = go+ge (src/reference_align.cpp:8) */
    int offset, navail;
    double* match;        /* [4][navail] */
    double* mismatch;
} aligner;

static int aligner_init(aligner* A, const char* ref, int R, const double* errors,
                        const char* names, int nenc, double go, double ge) {
    if (orc_check_encoding(errors, names, nenc)) return 1;
    A->R = R;
    A->ref = ref;
    A->GO = go + ge;
    A->GE = ge;
    A->offset = (int)names[0];
    A->navail = nenc;
    A->match = (double*)malloc(sizeof(double) * 8 * (size_t)nenc);
    A->mismatch = A->match + 4 * (size_t)nenc;
    return orc_cost_tables(errors, nenc, A->match, A->mismatch);
}
static void aligner_free(aligner* A) { free(A->match); }

/* (src/reference_align.cpp:184-225).  Ambiguity classes are decided on the
 * reference character alone (SURVEY App.B Q2): 2-fold codes always take the
 * mismatch table of mode 2, 3-fold codes the match table of mode 3, N the match
 * table of mode 4. */
static int cell_cost(const aligner* A, char r, char obs, char q, double* out) {
    int mode, matched;
    switch (r) {
        case 'A': case 'C': case 'G': case 'T': mode = 1; matched = (r == obs); break;
        case 'M': case 'R': case 'W': case 'S': case 'Y': case 'K': mode = 2; matched = 0; break;
        case 'V': case 'H': case 'D': case 'B': mode = 3; matched = 1; break;
        case 'N': mode = 4; matched = 1; break;
        default: return orc_fail("unrecognized base in reference sequence");
    }
    if ((int)q < A->offset) return orc_fail("quality cannot be lower than smallest encoded value");
    int loc = (int)q - A->offset;
    if (loc >= A->navail) loc = A->navail - 1;
    *out = (matched ? A->match : A->mismatch)[(mode - 1) * A->navail + loc];
    return 0;
}

/* Work buffers for one alignment. */
typedef struct {
    size_t cap_rows, cap_dirs;
    double* S;      /* scores[i]            */
    double* LJ;     /* left_jump_scores[i]  */
    int64_t* LP;    /* left_jump_points[i]  */
    int32_t* D;     /* directions, column-major */
} workspace;

static void ws_reserve(workspace* W, size_t rows, size_t cols) {
    if (rows > W->cap_rows) {
        W->S = (double*)realloc(W->S, rows * sizeof(double));
        W->LJ = (double*)realloc(W->LJ, rows * sizeof(double));
        W->LP = (int64_t*)realloc(W->LP, rows * sizeof(int64_t));
        W->cap_rows = rows;
    }
    if (rows * cols > W->cap_dirs) {
        W->D = (int32_t*)realloc(W->D, rows * cols * sizeof(int32_t));
        W->cap_dirs = rows * cols;
    }
}
static void ws_free(workspace* W) { free(W->S); free(W->LJ); free(W->LP); free(W->D); }

/* One DP fill (src/reference_align.cpp:54-181). */
static int dp_fill(const aligner* A, workspace* W, const char* seq, const char* qual,
                   int L, int local, double* score) {
    const int R = A->R;
    const size_t nrows = (size_t)L + 1;
    ws_reserve(W, nrows, (size_t)R + 1);
    double* S = W->S;
    double* LJ = W->LJ;
    int64_t* LP = W->LP;
    int32_t* D = W->D;
    const double GO = A->GO, GE = A->GE;

    /* column 0 */
    for (size_t i = 0; i < nrows; ++i) { D[i] = -1; LJ[i] = -INFINITY; LP[i] = 0; }
    if (local) {
        for (size_t i = 0; i < nrows; ++i) S[i] = 0;
    } else {
        S[0] = 0;
        for (size_t i = 1; i < nrows; ++i) S[i] = -GO - GE * (double)(i - 1);
    }

    for (int c = 1; c <= R; ++c) {
        const int pos = c - 1;
        const char r = A->ref[pos];
        const int last = local && c == R;
        const double VGO = last ? 0 : GO, VGE = last ? 0 : GE;
        const int32_t* Dp = D + (size_t)(c - 1) * nrows;
        int32_t* Dc = D + (size_t)c * nrows;

        double lag = S[0];
        S[0] -= (Dp[0] > 0 ? GE : GO);
        Dc[0] = 1;
        double UJ = -INFINITY;
        int64_t UP = 0;

        for (int i = 1; i <= L; ++i) {
            /* horizontal candidate, with the remembered sub-optimal opening */
            double H = S[i] - (Dp[i] > 0 ? GE : GO);
            LJ[i] -= GE;
            int64_t hstep = 1;
            if (LJ[i] > H) { hstep = 1 + pos - LP[i]; H = LJ[i]; }
            else { LJ[i] = H; LP[i] = pos; }

            /* vertical candidate (uses the already updated S[i-1] of this column) */
            double V = S[i - 1] - (Dc[i - 1] < 0 ? VGE : VGO);
            UJ -= VGE;
            int64_t vstep = 1;
            if (UJ > V) { vstep = 1 + i - UP; V = UJ; }
            else { UJ = V; UP = i; }

            double w = 0;
            if (cell_cost(A, r, seq[i - 1], qual[i - 1], &w)) return 1;
            const double M = lag + w;
            lag = S[i];

            if (M > H && M > V) { S[i] = M; Dc[i] = 0; }
            else if (H > V)     { S[i] = H; Dc[i] = (int32_t)hstep; }
            else                { S[i] = V; Dc[i] = -(int32_t)vstep; }
        }
    }
    *score = S[L];
    return 0;
}

/* Backtrack (src/reference_align.cpp:231-278) expressed as a visitor over moves. */
enum { MV_UP = 0, MV_DIAG = 1, MV_LEFT = 2 };
typedef void (*move_fn)(void* ctx, int kind, int col, int row);

static void backtrack(const workspace* W, int R, int L, move_fn fn, void* ctx) {
    const size_t nrows = (size_t)L + 1;
    int row = L;
    int c = R;
    while (c > 0) {
        const int32_t* Dc = W->D + (size_t)c * nrows;
        while (row > 0 && Dc[row] < 0) {
            int k = -Dc[row];
            while (k-- > 0) { fn(ctx, MV_UP, c, row); --row; }
        }
        const int32_t d = Dc[row];
        if (d == 0) {
            fn(ctx, MV_DIAG, c, row);
            --row;
            --c;
        } else {
            for (int k = 0; k < d; ++k) { fn(ctx, MV_LEFT, c, row); --c; }
        }
    }
    while (row > 0) { fn(ctx, MV_UP, 0, row); --row; }
}

/* fill_map (src/reference_align.cpp:280-305) */
typedef struct { int* is_diag; int64_t* pos; } mapctx;
static void map_move(void* p, int kind, int col, int row) {
    mapctx* m = (mapctx*)p;
    if (kind == MV_DIAG) { m->is_diag[col] = 1; m->pos[col] = row; }
    else if (kind == MV_LEFT) { m->is_diag[col] = 0; m->pos[col] = row + 1; }
}

/* querymap::operator() (src/reference_align.cpp:307-351); size_t arithmetic
 * there wraps, so do the same in uint64 and let callers compare as unsigned. */
static void map_interval(const mapctx* m, int R, int nrows, int a, int b, int with_gaps,
                         uint64_t* start, uint64_t* end) {
    if (R + 1 <= 1) { *start = 0; *end = 0; return; }
    uint64_t s, e;
    if (!with_gaps) {
        s = (uint64_t)m->pos[a + 1];
        e = (uint64_t)m->pos[b] + (m->is_diag[b] ? 1 : 0);
    } else {
        if (a == 0) s = 1;
        else s = (uint64_t)m->pos[a] + (m->is_diag[a] ? 1 : 0);
        if (b + 1 == R + 1) e = (uint64_t)nrows;
        else e = (uint64_t)m->pos[b + 1];
    }
    *start = s - 1;
    *end = e - 1;
}

/* ------------------------------------------------------------------ */
int orc_align_one(const char* ref, int R, const char* seq, const char* qual, int L,
                  const double* errors, const char* names, int nenc,
                  double gapopen, double gapext, int local,
                  double* score, int32_t* dirs) {
    aligner A;
    if (aligner_init(&A, ref, R, errors, names, nenc, gapopen, gapext)) return 1;
    workspace W = {0};
    int rc = dp_fill(&A, &W, seq, qual, L, local, score);
    if (!rc && dirs) memcpy(dirs, W.D, sizeof(int32_t) * (size_t)(L + 1) * (size_t)(R + 1));
    ws_free(&W);
    aligner_free(&A);
    return rc;
}

static int check_lengths(const int64_t* so, const int64_t* qo, int64_t i) {
    if (so[i + 1] - so[i] != qo[i + 1] - qo[i])
        return orc_fail("sequence and quality strings should have the same length");
    return 0;
}

int orc_adaptor_align(const char* seq, const int64_t* seq_off,
                      const char* qual, const int64_t* qual_off, int64_t n,
                      const double* errors, const char* names, int nenc,
                      double gapopen, double gapext,
                      const char* adaptor, int R,
                      const int32_t* sec_starts, const int32_t* sec_ends, int nsec,
                      double* scores, int32_t* starts, int32_t* ends,
                      int32_t* sec_start_out, int32_t* sec_width_out) {
    aligner A;
    if (aligner_init(&A, adaptor, R, errors, names, nenc, gapopen, gapext)) return 1;
    workspace W = {0};
    mapctx m;
    m.is_diag = (int*)calloc((size_t)R + 1, sizeof(int));
    m.pos = (int64_t*)calloc((size_t)R + 1, sizeof(int64_t));
    int rc = 0;
    for (int64_t i = 0; i < n && !rc; ++i) {
        if ((rc = check_lengths(seq_off, qual_off, i))) break;
        const int L = (int)(seq_off[i + 1] - seq_off[i]);
        if ((rc = dp_fill(&A, &W, seq + seq_off[i], qual + qual_off[i], L, 1, &scores[i]))) break;
        backtrack(&W, R, L, map_move, &m);

        uint64_t s, e;
        map_interval(&m, R, L + 1, 0, R, 0, &s, &e);
        starts[i] = 0;
        ends[i] = 0;
        if (s < e) { starts[i] = (int32_t)(s + 1); ends[i] = (int32_t)e; }
        for (int k = 0; k < nsec; ++k) {
            map_interval(&m, R, L + 1, sec_starts[k], sec_ends[k], 1, &s, &e);
            sec_start_out[(int64_t)k * n + i] = (int32_t)(s + 1);
            sec_width_out[(int64_t)k * n + i] = (int32_t)(e - s);
        }
    }
    free(m.is_diag);
    free(m.pos);
    ws_free(&W);
    aligner_free(&A);
    return rc;
}

int orc_align_scores(const char* seq, const int64_t* seq_off,
                     const char* qual, const int64_t* qual_off, int64_t n,
                     const double* errors, const char* names, int nenc,
                     double gapopen, double gapext,
                     const char* ref, int R, int local, double* scores) {
    aligner A;
    if (aligner_init(&A, ref, R, errors, names, nenc, gapopen, gapext)) return 1;
    workspace W = {0};
    int rc = 0;
    for (int64_t i = 0; i < n && !rc; ++i) {
        if ((rc = check_lengths(seq_off, qual_off, i))) break;
        const int L = (int)(seq_off[i + 1] - seq_off[i]);
        rc = dp_fill(&A, &W, seq + seq_off[i], qual + qual_off[i], L, local, &scores[i]);
    }
    ws_free(&W);
    aligner_free(&A);
    return rc;
}

/* fill_strings (src/reference_align.cpp:353-389): moves are emitted from the
 * end of the alignment, so collect reversed and flip. */
typedef struct { char* r; char* q; int64_t n; const char* ref; const char* seq; } strctx;
static void str_move(void* p, int kind, int col, int row) {
    strctx* s = (strctx*)p;
    s->r[s->n] = (kind == MV_UP) ? '-' : s->ref[col - 1];
    s->q[s->n] = (kind == MV_LEFT) ? '-' : s->seq[row - 1];
    ++s->n;
}

int orc_general_align(const char* seq, const int64_t* seq_off,
                      const char* qual, const int64_t* qual_off, int64_t n,
                      const double* errors, const char* names, int nenc,
                      double gapopen, double gapext,
                      const char* ref, int R,
                      double* scores, int32_t* edits,
                      char* aln_ref, char* aln_query, int64_t* aln_off, int64_t aln_cap) {
    aligner A;
    if (aligner_init(&A, ref, R, errors, names, nenc, gapopen, gapext)) return 1;
    workspace W = {0};
    int rc = 0;
    int64_t used = 0;
    char* tr = NULL;
    char* tq = NULL;
    size_t tcap = 0;
    if (aln_off) aln_off[0] = 0;
    for (int64_t i = 0; i < n && !rc; ++i) {
        if ((rc = check_lengths(seq_off, qual_off, i))) break;
        const int L = (int)(seq_off[i + 1] - seq_off[i]);
        if ((rc = dp_fill(&A, &W, seq + seq_off[i], qual + qual_off[i], L, 0, &scores[i]))) break;
        if ((size_t)(L + R) > tcap) {
            tcap = (size_t)(L + R) * 2 + 16;
            tr = (char*)realloc(tr, tcap);
            tq = (char*)realloc(tq, tcap);
        }
        strctx s = {tr, tq, 0, ref, seq + seq_off[i]};
        backtrack(&W, R, L, str_move, &s);
        int32_t ed = 0;
        for (int64_t j = 0; j < s.n; ++j) ed += (tr[j] != tq[j]);
        edits[i] = ed;
        if (aln_ref) {
            if (used + s.n > aln_cap) { rc = orc_fail("alignment string buffer too small"); break; }
            for (int64_t j = 0; j < s.n; ++j) {
                aln_ref[used + j] = tr[s.n - 1 - j];
                aln_query[used + j] = tq[s.n - 1 - j];
            }
            used += s.n;
            aln_off[i + 1] = used;
        }
    }
    free(tr);
    free(tq);
    ws_free(&W);
    aligner_free(&A);
    return rc;
}

/* (src/mask_bad_bases.cpp:10-52): strict '>' against the threshold. */
int orc_mask_bad_bases(const char* seq, const int64_t* seq_off,
                       const char* qual, const int64_t* qual_off, int64_t n,
                       const double* errors, const char* names, int nenc,
                       double threshold, char* out) {
    if (orc_check_encoding(errors, names, nenc)) return 1;
    const int offset = (int)names[0];
    for (int64_t i = 0; i < n; ++i) {
        if (check_lengths(seq_off, qual_off, i)) return 1;
        const int64_t L = seq_off[i + 1] - seq_off[i];
        for (int64_t j = 0; j < L; ++j) {
            double e = 0;
            if (to_error(errors, offset, nenc, qual[qual_off[i] + j], &e)) return 1;
            out[seq_off[i] + j] = (e > threshold) ? 'N' : seq[seq_off[i] + j];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* (reference src/unmask_alignment.cpp:12-59): copy each gapped row, replacing every 'N'/'n' by
 * the base at the same ungapped position of the original sequence.  Errors in the reference's
 * order: entry counts, row widths (check_alignment_width, src/DNA_input.cpp:90-104), then row by
 * row "longer than the original" (at the first masked base past the end) or "different lengths". */
int orc_unmask_alignment(const char* aln, const int64_t* aln_off, int64_t naln, const char* orig,
                         const int64_t* orig_off, int64_t norig, char* out) {
    if (naln != norig) return orc_fail("alignment and original sequences should have the same number of entries");
    for (int64_t i = 1; i < naln; ++i)
        if (aln_off[i + 1] - aln_off[i] != aln_off[1] - aln_off[0]) return orc_fail("alignment strings should have the same length");
    for (int64_t i = 0; i < naln; ++i) {
        const int64_t W = aln_off[i + 1] - aln_off[i], L = orig_off[i + 1] - orig_off[i];
        int64_t pos_nominal = 0;
        for (int64_t p = 0; p < W; ++p) {
            char c = aln[aln_off[i] + p];
            if (c != '-') {
                if (c == 'N' || c == 'n') {
                    if (pos_nominal >= L) return orc_fail("sequence in alignment string is longer than the original");
                    c = orig[orig_off[i] + pos_nominal];
                }
                ++pos_nominal;
            }
            out[aln_off[i] + p] = c;
        }
        if (pos_nominal != L) return orc_fail("original sequence and that in the alignment string have different lengths");
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Per-read shuffle used by the scrambled-control callers.  The reference shuffles with R's
 * sample() (R/getAdaptorThresholds.R:68-92), which cannot be reproduced without R; this
 * restates OUR generator (sarlacc_amd/csrc/resident.hip:k_scramble) so the device shuffle can
 * be checked byte for byte: splitmix64 stream per read seeded with
 * seed ^ (read+1)*0xD1342543DE82EF95, Fisher-Yates from the end, partner of k =
 * ((next>>32)*(k+1))>>32. */
static uint64_t splitmix64(uint64_t* x) {
    *x += 0x9E3779B97F4A7C15ull;
    uint64_t z = *x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int orc_scramble(const char* seq, const char* qual, const int64_t* off, int64_t n, uint64_t seed,
                 char* oseq, char* oqual) {
    for (int64_t r = 0; r < n; ++r) {
        const int64_t s = off[r], L = off[r + 1] - s;
        memcpy(oseq + s, seq + s, (size_t)L);
        memcpy(oqual + s, qual + s, (size_t)L);
        uint64_t st = seed ^ ((uint64_t)(r + 1) * 0xD1342543DE82EF95ull);
        for (int64_t k = L - 1; k > 0; --k) {
            const uint64_t rnd = splitmix64(&st) >> 32;
            const int64_t j = (int64_t)((rnd * (uint64_t)(k + 1)) >> 32);
            char a = oseq[s + k], b = oqual[s + k];
            oseq[s + k] = oseq[s + j]; oqual[s + k] = oqual[s + j];
            oseq[s + j] = a; oqual[s + j] = b;
        }
    }
    return 0;
}
