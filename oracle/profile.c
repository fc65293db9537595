/*
 * profile.c -- CPU restatement of the reference's alignment-profiling routines:
 *   find_homopolymers   /root/reference/src/homopolymer.cpp:87-134
 *   match_homopolymers  /root/reference/src/homopolymer.cpp:141-209
 *   find_errors         /root/reference/src/find_errors.cpp:9-121
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Runs.  A run is a maximal stretch of equal non-gap characters of a gapped string, gap characters inside or
 * after it notwithstanding (homopolymer.cpp:6-79): it starts at the first character that differs from the run
 * before it (gaps skipped), and everything up to the next differing non-gap character belongs to it.
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);

typedef struct {
    const char* p;
    int64_t len;
    int64_t start;      /* index of the run's first character in the gapped string */
    int64_t next;       /* index of the first character of the next run (or len) */
    int64_t gaps_before;/* gap characters before `start` */
    int64_t gaps_upto;  /* gap characters before `next` */
    char base;
} run_t;

/* positions the walker before the first run: `next` = first non-gap character */
static void run_init(run_t* r, const char* p, int64_t len) {
    r->p = p; r->len = len; r->start = 0; r->gaps_before = 0; r->gaps_upto = 0; r->base = 0;
    int64_t i = 0;
    while (i < len && p[i] == '-') { ++i; ++r->gaps_upto; }
    r->next = i;
}
static int run_done(const run_t* r) { return r->next == r->len; }
static void run_advance(run_t* r) {
    r->start = r->next;
    r->gaps_before = r->gaps_upto;
    r->base = r->p[r->start];
    int64_t i = r->start + 1;
    while (i < r->len) {
        const char c = r->p[i];
        if (c != '-' && c != r->base) break;
        if (c == '-') ++r->gaps_upto;
        ++i;
    }
    r->next = i;
}
static int64_t run_pos(const run_t* r) { return r->start - r->gaps_before; }                        /* ungapped start */
static int64_t run_length(const run_t* r) { return (r->next - r->gaps_upto) - run_pos(r); }         /* bases in it */
/* extent in the gapped string: [start, end) without the gaps on either side, [start_g, next) with them */
static int64_t run_start_with_gaps(const run_t* r) {
    int64_t q = r->start;
    while (q > 0 && r->p[q - 1] == '-') --q;
    return q;
}
static int64_t run_end(const run_t* r) {
    int64_t q = r->next;
    while (q > r->start && r->p[q - 1] == '-') --q;
    return q;
}

int orc_find_homopolymers(const char* seq, const int64_t* off, int64_t n, int32_t* idx, int32_t* pos, int32_t* size,
                          char* base, int64_t cap, int64_t* count) {
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        run_t r;
        run_init(&r, seq + off[i], off[i + 1] - off[i]);
        while (!run_done(&r)) {
            run_advance(&r);
            const int64_t L = run_length(&r);
            if (L == 1) continue;
            if (k < cap) { idx[k] = (int32_t)i; pos[k] = (int32_t)(run_pos(&r) + 1); size[k] = (int32_t)L; base[k] = r.base; }
            ++k;
        }
    }
    *count = k;
    return 0;
}

int orc_match_homopolymers(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                           int64_t nread, int32_t* idx, int32_t* pos, int32_t* rlen, int64_t cap, int64_t* count) {
    *count = 0;
    if (nref != nread) return orc_fail("lengths of alignment vectors should match up");
    int64_t k = 0;
    for (int64_t i = 0; i < nref; ++i) {
        const char* rf = ref + ref_off[i];
        const char* rd = read + read_off[i];
        const int64_t len = ref_off[i + 1] - ref_off[i];
        if (read_off[i + 1] - read_off[i] != len) return orc_fail("read and reference alignment strings should have equal length");
        if (len == 0) continue;
        run_t r;
        run_init(&r, rf, len);
        while (!run_done(&r)) {
            run_advance(&r);
            if (run_length(&r) == 1) continue;
            /* the longest run of the same base in the read that overlaps the reference run proper; the read is
             * examined over the reference run extended by the gaps on either side */
            const int64_t far_left = run_start_with_gaps(&r), far_right = r.next;
            const int64_t left = r.start, right = run_end(&r);
            run_t q;
            run_init(&q, rd + far_left, far_right - far_left);
            int64_t best = 0;
            while (!run_done(&q)) {
                run_advance(&q);
                if (right > q.start + far_left && left < run_end(&q) + far_left) {
                    const int64_t L = run_length(&q);
                    if (L > best && q.base == r.base) best = L;
                }
            }
            if (k < cap) { idx[k] = (int32_t)i; pos[k] = (int32_t)(run_pos(&r) + 1); rlen[k] = (int32_t)best; }
            ++k;
        }
    }
    *count = k;
    return 0;
}

int orc_find_errors(const char* ref, const int64_t* ref_off, int64_t nref, const char* read, const int64_t* read_off,
                    int64_t nread, int64_t* standard_len, char* bases, int32_t* to_a, int32_t* to_c, int32_t* to_g,
                    int32_t* to_t, int32_t* deletions, int64_t cap_bases, int32_t* ins_pos, int32_t* ins_len, int64_t cap_ins,
                    int64_t* nins) {
    *standard_len = 0; *nins = 0;
    if (nref != nread) return orc_fail("lengths of alignment vectors should match up");
    int64_t sl = 0;
    if (nref) {
        const char* s = ref + ref_off[0];
        const int64_t len = ref_off[1] - ref_off[0];
        for (int64_t x = 0; x < len; ++x)
            if (s[x] != '-') { if (sl < cap_bases) bases[sl] = s[x]; ++sl; }
    }
    *standard_len = sl;
    if (sl > cap_bases) return 2;   /* sizing */
    for (int64_t x = 0; x < sl; ++x) to_a[x] = to_c[x] = to_g[x] = to_t[x] = deletions[x] = 0;
    int64_t k = 0;
    for (int64_t i = 0; i < nref; ++i) {
        const char* rf = ref + ref_off[i];
        const char* rd = read + read_off[i];
        const int64_t len = ref_off[i + 1] - ref_off[i];
        if (read_off[i + 1] - read_off[i] != len) return orc_fail("read and reference alignment strings should have equal length");
        int64_t cur = 0, gaps = 0;
        while (cur < len) {
            if (rf[cur] != '-') {
                const int64_t tp = cur - gaps;
                if (tp >= sl) return orc_fail("reference sequence should be the same for all alignments");
                switch (rd[cur]) {
                    case '-': ++deletions[tp]; break;
                    case 'A': ++to_a[tp]; break;
                    case 'C': ++to_c[tp]; break;
                    case 'G': ++to_g[tp]; break;
                    case 'T': ++to_t[tp]; break;
                    default: {
                        char msg[64];
                        snprintf(msg, sizeof msg, "unknown character '%c' in alignment string", rd[cur]);
                        return orc_fail(msg);
                    }
                }
                ++cur;
            } else {   /* an insertion: its length, filed under the position of the next reference base */
                const int64_t first = cur;
                while (cur < len && rf[cur] == '-') { ++cur; ++gaps; }
                if (k < cap_ins) { ins_pos[k] = (int32_t)(cur - gaps); ins_len[k] = (int32_t)(cur - first); }
                ++k;
            }
        }
    }
    *nins = k;
    return 0;
}
