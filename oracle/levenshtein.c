/*
 * levenshtein.c -- oracle restatement of sarlacc's masked Levenshtein machinery.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows (file:line relative to /root/reference):
 *   src/compute_lev_masked.cpp:13-64  dense lower triangle, costs 0 / 0.5 (N) / 1
 *   src/sorted_trie.cpp:13-21         integer costs x2: match 0, N-involved 1, else 2
 *   src/sorted_trie.cpp:39-71         trie insertion (A,C,G,T,N children; other chars dropped)
 *   src/sorted_trie.cpp:107-226       DFS with one DP row per node and row-wise pruning
 *   src/sorted_trie.cpp:245-278       find()
 *
 * The reference additionally caches DP rows between consecutive (sorted)
 * queries (:117-119,:246-257); that is a pure optimisation (its own tests
 * assert sorted == unsorted processing, tests/testthat/test-levenshtein.R:61)
 * and is not restated: every query runs the DFS from the root.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

int orc_fail(const char* msg);

/* ------------------------------------------------------------------ */
static double min3(double a, double b, double c) {
    double m = a < b ? a : b;
    return m < c ? m : c;
}

int orc_compute_lev_masked(const char* seq, const int64_t* off, int64_t n, double* out) {
    int64_t maxlen = 0;
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] - off[i] > maxlen) maxlen = off[i + 1] - off[i];
    double* col = (double*)malloc(sizeof(double) * (size_t)(maxlen + 1));
    double* prev = (double*)malloc(sizeof(double) * (size_t)(maxlen + 1));
    int64_t o = 0;
    for (int64_t i = 0; i < n; ++i) {
        const char* a = seq + off[i];
        const int64_t la = off[i + 1] - off[i];
        for (int64_t j = i + 1; j < n; ++j) {
            const char* b = seq + off[j];
            const int64_t lb = off[j + 1] - off[j];
            for (int64_t x = 0; x <= la; ++x) prev[x] = (double)x;
            for (int64_t y = 0; y < lb; ++y) {
                col[0] = (double)(y + 1);
                for (int64_t x = 0; x < la; ++x) {
                    const double sub = (b[y] == 'N' || a[x] == 'N') ? 0.5 : (b[y] == a[x] ? 0.0 : 1.0);
                    col[x + 1] = min3(prev[x + 1] + 1, col[x] + 1, prev[x] + sub);
                }
                double* t = col; col = prev; prev = t;
            }
            out[o++] = prev[la];
        }
    }
    free(col);
    free(prev);
    return 0;
}

/* ------------------------------------------------------------------ */
#define NCHILD 5
static const char CHILD_BASE[NCHILD] = {'A', 'C', 'G', 'T', 'N'};

typedef struct node {
    int32_t child[NCHILD];   /* index into pool, 0 = absent (root is 0 and never a child) */
    int32_t* idx;
    int32_t nidx, capidx;
} node;

typedef struct {
    node* pool;
    int32_t used, cap;
} trie;

static int32_t trie_new(trie* T) {
    if (T->used == T->cap) {
        T->cap = T->cap ? T->cap * 2 : 1024;
        T->pool = (node*)realloc(T->pool, sizeof(node) * (size_t)T->cap);
    }
    memset(&T->pool[T->used], 0, sizeof(node));
    return T->used++;
}

static int child_slot(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'N': return 4;
    }
    return -1;
}

/* Strings containing any other character are silently not stored
 * (src/sorted_trie.cpp:53-69 has no default branch; SURVEY App.B Q9). */
static void trie_insert(trie* T, const char* s, int64_t len, int32_t index) {
    int32_t cur = 0;
    for (int64_t p = 0; p < len; ++p) {
        const int slot = child_slot(s[p]);
        if (slot < 0) return;
        if (!T->pool[cur].child[slot]) {
            const int32_t fresh = trie_new(T);
            T->pool[cur].child[slot] = fresh;
        }
        cur = T->pool[cur].child[slot];
    }
    node* nd = &T->pool[cur];
    if (nd->nidx == nd->capidx) {
        nd->capidx = nd->capidx ? nd->capidx * 2 : 4;
        nd->idx = (int32_t*)realloc(nd->idx, sizeof(int32_t) * (size_t)nd->capidx);
    }
    nd->idx[nd->nidx++] = index;
}

static void trie_free(trie* T) {
    for (int32_t i = 0; i < T->used; ++i) free(T->pool[i].idx);
    free(T->pool);
}

static int edit_cost(char a, char b) {
    if (a == 'N' || b == 'N') return 1;
    return a == b ? 0 : 2;
}

typedef struct {
    int32_t* out;
    int64_t n, cap;
} sink;

static void sink_push(sink* S, int32_t v) {
    if (S->n < S->cap) S->out[S->n] = v;
    ++S->n;
}

/* rows: (maxdepth+1) rows of (qlen+1) ints; row d belongs to the node at depth d */
static void dfs(const trie* T, int32_t cur, int depth, const char* q, int qlen,
                int limit2, int* rows, sink* S) {
    const node* nd = &T->pool[cur];
    const int* row = rows + (size_t)depth * (size_t)(qlen + 1);
    if (limit2 >= row[qlen])
        for (int32_t k = 0; k < nd->nidx; ++k) sink_push(S, nd->idx[k]);

    int has_child = 0;
    for (int b = 0; b < NCHILD; ++b) has_child |= nd->child[b] != 0;
    if (!has_child) return;

    if (depth > 0) {
        /* prune when no prefix of the query can still come in under the limit
         * (src/sorted_trie.cpp:160-176) */
        int go = (limit2 >= row[qlen] + 2);
        for (int i = qlen - 1; i >= 0 && !go; --i) go = row[i] <= limit2;
        if (!go) return;
    }
    int* next = rows + (size_t)(depth + 1) * (size_t)(qlen + 1);
    for (int b = 0; b < NCHILD; ++b) {
        if (!nd->child[b]) continue;
        const char base = CHILD_BASE[b];
        next[0] = row[0] + 2;
        for (int i = 1; i <= qlen; ++i) {
            int v = row[i] + 2;
            const int ins = next[i - 1] + 2;
            const int sub = row[i - 1] + edit_cost(q[i - 1], base);
            if (ins < v) v = ins;
            if (sub < v) v = sub;
            next[i] = v;
        }
        dfs(T, nd->child[b], depth + 1, q, qlen, limit2, rows, S);
    }
}

/* Neighbour lists of every string within one set, trie order
 * (equivalent of the loop in fast_levdist_test / umi_group). */
int orc_trie_neighbours(const char* const* strs, const int32_t* lens, int64_t n, int limit,
                        int64_t* nbr_off, int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need) {
    trie T = {0};
    trie_new(&T); /* root */
    int maxlen = 0;
    for (int64_t i = 0; i < n; ++i) {
        trie_insert(&T, strs[i], lens[i], (int32_t)i);
        if (lens[i] > maxlen) maxlen = lens[i];
    }
    int* rows = (int*)malloc(sizeof(int) * (size_t)(maxlen + 2) * (size_t)(maxlen + 1));
    sink S = {nbr, 0, nbr_cap};
    nbr_off[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int qlen = lens[i];
        for (int x = 0; x <= qlen; ++x) rows[x] = 2 * x;
        dfs(&T, 0, 0, strs[i], qlen, 2 * limit, rows, &S);
        nbr_off[i + 1] = S.n;
    }
    *nbr_need = S.n;
    free(rows);
    trie_free(&T);
    return (S.n > nbr_cap) ? 2 : 0; /* 2 = buffer too small, *nbr_need holds the size */
}

int orc_fast_levdist(const char* seq, const int64_t* off, int64_t n, int limit,
                     int64_t* nbr_off, int32_t* nbr, int64_t nbr_cap, int64_t* nbr_need) {
    const char** strs = (const char**)malloc(sizeof(char*) * (size_t)(n ? n : 1));
    int32_t* lens = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; ++i) { strs[i] = seq + off[i]; lens[i] = (int32_t)(off[i + 1] - off[i]); }
    int rc = orc_trie_neighbours(strs, lens, n, limit, nbr_off, nbr, nbr_cap, nbr_need);
    free(strs);
    free(lens);
    return rc;
}
