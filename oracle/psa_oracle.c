/*
 * psa_oracle.c -- CPU restatement of the reference's simple.cpp hot path (see psa_oracle.h).
 * TEST INFRASTRUCTURE ONLY; parity PINNED against the compiled reference (tests/golden/).
 *
 * The reference keeps a whole alignment vector in every cell (simple.cpp:164-211).  Here each
 * cell keeps, per table, which predecessor table won (2 bits), and the alignment is rebuilt
 * by walking back: the column pushed at a cell depends only on the winner
 * (simple.cpp:165/169/173 and the six sibling sites), the predecessor cell only on the table
 * being written.
 */
#define _GNU_SOURCE
#include "psa_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { GAP_OPEN = 2 /* h, simple.h:14 */, GAP_EXT = 1 /* g, simple.h:15 */ };

/* simple.cpp:13-16 */
static inline int32_t subst(uint8_t x, uint8_t y) { return x == y ? 2 : 0; }

/* The tie-break shared by every three-way choice (simple.cpp:162/166/170 and siblings):
 * table 1 if t1>=t2 && t1>=t3, else table 2 if t2>=t3, else table 3. */
static inline int pick(int32_t t1, int32_t t2, int32_t t3, int32_t* v) {
    if (t1 >= t2 && t1 >= t3) { *v = t1; return 1; }
    if (t2 >= t3) { *v = t2; return 2; }
    *v = t3;
    return 3;
}

/* simple.cpp:116-119 (row 0), 124-133 (column 0), simple.h:50 (T2 on row 0 is Entry()'s 0). */
void psa_oracle_boundary(int64_t i, int64_t j, int32_t t[3]) {
    if (i == 0 && j == 0) { t[0] = 0; t[1] = -GAP_OPEN; t[2] = -1; return; }
    if (i == 0) { t[0] = -1; t[1] = 0; t[2] = (int32_t)-(GAP_OPEN + GAP_EXT * j); return; }
    t[0] = -1; t[1] = (int32_t)-(GAP_OPEN + GAP_EXT * i); t[2] = -1; /* j == 0 */
}

/* One cell, order T1, T3, T2 as in the reference's j-loop (simple.cpp:158, 175, 194).
 * up* = row i-1, cur* = row i; returns the packed winners w1 | w2<<2 | w3<<4. */
static inline uint8_t cell(uint8_t ai, uint8_t bj, const int32_t* up1, const int32_t* up2,
                           const int32_t* up3, int32_t* cur1, int32_t* cur2, int32_t* cur3,
                           int64_t j) {
    int32_t v;
    int w1 = pick(up1[j - 1], up2[j - 1], up3[j - 1], &v);
    cur1[j] = v + subst(ai, bj);
    int w3 = pick(up1[j] - (GAP_EXT + GAP_OPEN), up2[j] - (GAP_EXT + GAP_OPEN),
                  up3[j] - GAP_EXT, &v);
    cur3[j] = v;
    int w2 = pick(cur1[j - 1] - (GAP_EXT + GAP_OPEN), cur2[j - 1] - GAP_EXT,
                  cur3[j - 1] - (GAP_EXT + GAP_OPEN), &v);
    cur2[j] = v;
    return (uint8_t)(w1 | (w2 << 2) | (w3 << 4));
}

static void init_row0(int32_t* t1, int32_t* t2, int32_t* t3, int64_t m) {
    for (int64_t j = 0; j <= m; j++) {
        int32_t t[3];
        psa_oracle_boundary(0, j, t);
        t1[j] = t[0]; t2[j] = t[1]; t3[j] = t[2];
    }
}

/* Special cases the row loop never reaches (simple.cpp:122 does not execute for n == 0, so the
 * final pick at simple.cpp:246 reads the initial row: T1 = -1, T2 = 0, T3 <= -2 -> score 0). */
static int degenerate(int64_t n, int64_t m, int32_t* score, int32_t triple[3]) {
    if (n != 0) return 0;
    int32_t t[3] = { -1, 0, (int32_t)-(GAP_OPEN + GAP_EXT * m) };
    if (triple) memcpy(triple, t, sizeof t);
    *score = 0;
    return 1;
}

int psa_oracle_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                     int32_t* score_out, int32_t final_triple_out[3]) {
    if (degenerate(n, m, score_out, final_triple_out)) return 0;
    int32_t* rows = (int32_t*)malloc((size_t)(m + 1) * 6 * sizeof(int32_t));
    if (!rows) return -2;
    int32_t *u1 = rows, *u2 = u1 + (m + 1), *u3 = u2 + (m + 1);
    int32_t *c1 = u3 + (m + 1), *c2 = c1 + (m + 1), *c3 = c2 + (m + 1);
    init_row0(u1, u2, u3, m);
    for (int64_t i = 1; i <= n; i++) {
        int32_t t[3];
        psa_oracle_boundary(i, 0, t);
        c1[0] = t[0]; c2[0] = t[1]; c3[0] = t[2];
        for (int64_t j = 1; j <= m; j++) cell(a[i - 1], b[j - 1], u1, u2, u3, c1, c2, c3, j);
        int32_t* s;
        s = u1; u1 = c1; c1 = s;
        s = u2; u2 = c2; c2 = s;
        s = u3; u3 = c3; c3 = s;
    }
    int32_t v;
    pick(u1[m], u2[m], u3[m], &v); /* simple.cpp:246-256 */
    *score_out = v;
    if (final_triple_out) { final_triple_out[0] = u1[m]; final_triple_out[1] = u2[m]; final_triple_out[2] = u3[m]; }
    free(rows);
    return 0;
}

/* Walk back from (n, m, kstar) over the code matrix, then the prefix fix-up of
 * simple.cpp:279-290 with find_last (simple.cpp:24-41: first index != -1, 0 if none). */
static int traceback(const uint8_t* codes, int64_t n, int64_t m, int kstar, int32_t* ai,
                     int32_t* bj, int64_t cap, int64_t* len_out) {
    int64_t i = n, j = m, len = 0;
    int k = kstar;
    const int64_t ld = m + 1;
    while (i > 0 && j > 0) {
        int w = (codes[i * ld + j] >> (2 * (k - 1))) & 3;
        if (len >= cap) return -1;
        ai[len] = (w == 2) ? -1 : (int32_t)(i - 1);
        bj[len] = (w == 3) ? -1 : (int32_t)(j - 1);
        len++;
        if (k == 1) { i--; j--; } else if (k == 2) { j--; } else { i--; }
        k = w;
    }
    /* ai/bj hold the walk backwards; find the first (in forward order) real indices. */
    int64_t fi = 0, fj = 0;
    for (int64_t t = len - 1; t >= 0; t--) if (ai[t] != -1) { fi = ai[t]; break; }
    for (int64_t t = len - 1; t >= 0; t--) if (bj[t] != -1) { fj = bj[t]; break; }
    if (len + fi + fj > cap) return -1;
    /* still in reversed order: append the A-prefix (fi-1 .. 0), then the B-prefix. */
    for (int64_t x = fi; x > 0; x--) { ai[len] = (int32_t)(x - 1); bj[len] = -1; len++; }
    for (int64_t x = fj; x > 0; x--) { ai[len] = -1; bj[len] = (int32_t)(x - 1); len++; }
    for (int64_t x = 0, y = len - 1; x < y; x++, y--) {
        int32_t t = ai[x]; ai[x] = ai[y]; ai[y] = t;
        t = bj[x]; bj[x] = bj[y]; bj[y] = t;
    }
    *len_out = len;
    return 0;
}

int psa_oracle_align(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                     int32_t* score_out, int32_t* ai, int32_t* bj, int64_t cap,
                     int64_t* len_out) {
    *len_out = 0;
    if (degenerate(n, m, score_out, NULL)) return 0;
    uint8_t* codes = (uint8_t*)malloc((size_t)(n + 1) * (size_t)(m + 1));
    int32_t* rows = (int32_t*)malloc((size_t)(m + 1) * 6 * sizeof(int32_t));
    if (!codes || !rows) { free(codes); free(rows); return -2; }
    int32_t *u1 = rows, *u2 = u1 + (m + 1), *u3 = u2 + (m + 1);
    int32_t *c1 = u3 + (m + 1), *c2 = c1 + (m + 1), *c3 = c2 + (m + 1);
    init_row0(u1, u2, u3, m);
    for (int64_t i = 1; i <= n; i++) {
        int32_t t[3];
        psa_oracle_boundary(i, 0, t);
        c1[0] = t[0]; c2[0] = t[1]; c3[0] = t[2];
        uint8_t* crow = codes + (size_t)i * (size_t)(m + 1);
        for (int64_t j = 1; j <= m; j++)
            crow[j] = cell(a[i - 1], b[j - 1], u1, u2, u3, c1, c2, c3, j);
        int32_t* s;
        s = u1; u1 = c1; c1 = s;
        s = u2; u2 = c2; c2 = s;
        s = u3; u3 = c3; c3 = s;
    }
    int32_t v;
    int kstar = pick(u1[m], u2[m], u3[m], &v);
    *score_out = v;
    int rc = traceback(codes, n, m, kstar, ai, bj, cap, len_out);
    free(codes);
    free(rows);
    return rc;
}

int psa_oracle_block_codes(const uint8_t* a, const uint8_t* b, int64_t r0, int64_t r1, int64_t c0,
                           int64_t c1, const int32_t* top[3], const int32_t* left[3],
                           int32_t* bottom[3], int32_t* right[3], uint8_t* codes) {
    const int64_t w = c1 - c0, h = r1 - r0;
    int32_t* rows = (int32_t*)malloc((size_t)(w + 1) * 6 * sizeof(int32_t));
    if (!rows) return -2;
    int32_t *u1 = rows, *u2 = u1 + (w + 1), *u3 = u2 + (w + 1);
    int32_t *k1 = u3 + (w + 1), *k2 = k1 + (w + 1), *k3 = k2 + (w + 1);
    memcpy(u1, top[0], (size_t)(w + 1) * 4);
    memcpy(u2, top[1], (size_t)(w + 1) * 4);
    memcpy(u3, top[2], (size_t)(w + 1) * 4);
    if (right) for (int k = 0; k < 3; k++) if (right[k]) right[k][0] = top[k][w];
    for (int64_t y = 1; y <= h; y++) {
        k1[0] = left[0][y]; k2[0] = left[1][y]; k3[0] = left[2][y];
        for (int64_t x = 1; x <= w; x++) {
            uint8_t cw = cell(a[r0 + y - 1], b[c0 + x - 1], u1, u2, u3, k1, k2, k3, x);
            if (codes) codes[(size_t)(y - 1) * (size_t)w + (size_t)(x - 1)] = cw;
        }
        if (right) {
            if (right[0]) right[0][y] = k1[w];
            if (right[1]) right[1][y] = k2[w];
            if (right[2]) right[2][y] = k3[w];
        }
        int32_t* s;
        s = u1; u1 = k1; k1 = s;
        s = u2; u2 = k2; k2 = s;
        s = u3; u3 = k3; k3 = s;
    }
    if (bottom) {
        if (bottom[0]) memcpy(bottom[0], u1, (size_t)(w + 1) * 4);
        if (bottom[1]) memcpy(bottom[1], u2, (size_t)(w + 1) * 4);
        if (bottom[2]) memcpy(bottom[2], u3, (size_t)(w + 1) * 4);
    }
    free(rows);
    return 0;
}

int psa_oracle_block(const uint8_t* a, const uint8_t* b, int64_t r0, int64_t r1, int64_t c0,
                     int64_t c1, const int32_t* top[3], const int32_t* left[3],
                     int32_t* bottom[3], int32_t* right[3]) {
    return psa_oracle_block_codes(a, b, r0, r1, c0, c1, top, left, bottom, right, NULL);
}

/* ------------------------------------------------------------------------------------------
 * Multi-threaded variant: the reference's own decomposition (simple.cpp:90-101, 261-277).
 * Stripe id owns columns [id*block, id*block+block) (+ remainder for the last one); it may
 * compute row i once stripe id-1 has published its last-column triple of row i.  The
 * single-slot mailbox + condvar of simple.cpp:136-154/215-235 becomes a full boundary column
 * per stripe plus an atomic "rows done" counter, so a stripe never blocks on its consumer.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int p, id;
    const uint8_t *a, *b;
    int64_t n, m;
    uint8_t* codes;             /* (n+1)*(m+1) or NULL (score only) */
    int32_t* edge;              /* p boundaries x 3 x (n+1): edge[(id*3+k)*(n+1)+i] */
    _Atomic int64_t* done;      /* rows published by each stripe */
    int32_t final_triple[3];
} stripe_arg;

static void* stripe_main(void* varg) {
    stripe_arg* s = (stripe_arg*)varg;
    const int64_t n = s->n, M = s->m;
    const int64_t block = M / s->p;
    const int64_t start = (int64_t)s->id * block;
    const int64_t w = (s->id == s->p - 1) ? block + (M % s->p) : block;
    int32_t* rows = (int32_t*)malloc((size_t)(w + 1) * 6 * sizeof(int32_t));
    int32_t *u1 = rows, *u2 = u1 + (w + 1), *u3 = u2 + (w + 1);
    int32_t *c1 = u3 + (w + 1), *c2 = c1 + (w + 1), *c3 = c2 + (w + 1);
    for (int64_t x = 0; x <= w; x++) {
        int32_t t[3];
        psa_oracle_boundary(0, start + x, t);
        u1[x] = t[0]; u2[x] = t[1]; u3[x] = t[2];
    }
    int32_t* out = s->edge + (size_t)s->id * 3 * (size_t)(n + 1);
    const int32_t* in = s->id ? s->edge + (size_t)(s->id - 1) * 3 * (size_t)(n + 1) : NULL;
    const int64_t CH = 32; /* rows per hand-off */
    int64_t avail = s->id ? 0 : n;
    for (int64_t i = 1; i <= n; i++) {
        if (s->id == 0) {
            int32_t t[3];
            psa_oracle_boundary(i, 0, t);
            c1[0] = t[0]; c2[0] = t[1]; c3[0] = t[2];
        } else {
            while (avail < i) {
                avail = atomic_load_explicit(&s->done[s->id - 1], memory_order_acquire);
                if (avail < i) sched_yield();
            }
            c1[0] = in[i]; c2[0] = in[(n + 1) + i]; c3[0] = in[2 * (n + 1) + i];
        }
        const uint8_t ai = s->a[i - 1];
        const uint8_t* bb = s->b + start;
        if (s->codes) {
            uint8_t* crow = s->codes + (size_t)i * (size_t)(M + 1) + start;
            for (int64_t x = 1; x <= w; x++) crow[x] = cell(ai, bb[x - 1], u1, u2, u3, c1, c2, c3, x);
        } else {
            for (int64_t x = 1; x <= w; x++) cell(ai, bb[x - 1], u1, u2, u3, c1, c2, c3, x);
        }
        out[i] = c1[w]; out[(n + 1) + i] = c2[w]; out[2 * (n + 1) + i] = c3[w];
        if (s->id != s->p - 1 && (i % CH == 0 || i == n))
            atomic_store_explicit(&s->done[s->id], i, memory_order_release);
        int32_t* t;
        t = u1; u1 = c1; c1 = t;
        t = u2; u2 = c2; c2 = t;
        t = u3; u3 = c3; c3 = t;
    }
    s->final_triple[0] = u1[w]; s->final_triple[1] = u2[w]; s->final_triple[2] = u3[w];
    free(rows);
    return NULL;
}

static int run_mt(int p, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                  uint8_t* codes, int32_t triple[3]) {
    if (p < 1) p = 1;
    int32_t* edge = (int32_t*)malloc((size_t)p * 3 * (size_t)(n + 1) * sizeof(int32_t));
    _Atomic int64_t* done = (_Atomic int64_t*)calloc((size_t)p, sizeof(_Atomic int64_t));
    stripe_arg* args = (stripe_arg*)calloc((size_t)p, sizeof(stripe_arg));
    pthread_t* th = (pthread_t*)calloc((size_t)p, sizeof(pthread_t));
    if (!edge || !done || !args || !th) { free(edge); free((void*)done); free(args); free(th); return -2; }
    for (int id = 0; id < p; id++) {
        stripe_arg s = { p, id, a, b, n, m, codes, edge, done, { 0, 0, 0 } };
        args[id] = s;
    }
    /* simple.cpp:269-273: p-1 threads, the caller runs the last stripe itself. */
    for (int id = 0; id < p - 1; id++) pthread_create(&th[id], NULL, stripe_main, &args[id]);
    stripe_main(&args[p - 1]);
    for (int id = 0; id < p - 1; id++) pthread_join(th[id], NULL);
    memcpy(triple, args[p - 1].final_triple, sizeof(int32_t) * 3);
    free(edge); free((void*)done); free(args); free(th);
    return 0;
}

int psa_oracle_align_mt(int p, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                        int32_t* score_out, int32_t* ai, int32_t* bj, int64_t cap,
                        int64_t* len_out) {
    *len_out = 0;
    if (degenerate(n, m, score_out, NULL)) return 0;
    uint8_t* codes = (uint8_t*)malloc((size_t)(n + 1) * (size_t)(m + 1));
    if (!codes) return -2;
    int32_t t[3];
    int rc = run_mt(p, a, n, b, m, codes, t);
    if (rc == 0) {
        int32_t v;
        int kstar = pick(t[0], t[1], t[2], &v);
        *score_out = v;
        rc = traceback(codes, n, m, kstar, ai, bj, cap, len_out);
    }
    free(codes);
    return rc;
}

int psa_oracle_score_mt(int p, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                        int32_t* score_out) {
    if (degenerate(n, m, score_out, NULL)) return 0;
    int32_t t[3];
    int rc = run_mt(p, a, n, b, m, NULL, t);
    if (rc == 0) { int32_t v; pick(t[0], t[1], t[2], &v); *score_out = v; }
    return rc;
}

/* read_fasta.cpp:8-23 */
uint8_t* psa_oracle_read_fasta(const char* path, int64_t* len_out) {
    *len_out = 0;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Error opening file: %s\n", path); return NULL; }
    size_t cap = 1 << 16, len = 0;
    uint8_t* seq = (uint8_t*)malloc(cap);
    char* line = NULL;
    size_t lcap = 0;
    ssize_t got;
    while ((got = getline(&line, &lcap, f)) >= 0) {
        if (got > 0 && line[got - 1] == '\n') got--;   /* std::getline drops the '\n' only */
        if (got == 0 || line[0] == '>') continue;      /* read_fasta.cpp:17 */
        if (len + (size_t)got > cap) { while (len + (size_t)got > cap) cap *= 2; seq = (uint8_t*)realloc(seq, cap); }
        memcpy(seq + len, line, (size_t)got);
        len += (size_t)got;
    }
    free(line);
    fclose(f);
    *len_out = (int64_t)len;
    return seq;
}

/* simple.cpp:312 and output_alignement simple.cpp:43-70 (both "Length" lines print al_A's size). */
char* psa_oracle_format(int32_t score, const uint8_t* a, const uint8_t* b, const int32_t* ai,
                        const int32_t* bj, int64_t len, int64_t* out_len) {
    size_t cap = (size_t)len * 2 + 256;
    char* out = (char*)malloc(cap);
    size_t o = (size_t)snprintf(out, cap, "Score: %d\nLength Sequence A : %lld\nLength Sequence B : %lld\nSequence A : ",
                                score, (long long)len, (long long)len);
    for (int64_t k = 0; k < len; k++) out[o++] = ai[k] == -1 ? '-' : (char)a[ai[k]];
    memcpy(out + o, "\nSequence B : ", 14); o += 14;
    for (int64_t k = 0; k < len; k++) out[o++] = bj[k] == -1 ? '-' : (char)b[bj[k]];
    out[o++] = '\n';
    out[o] = 0;
    if (out_len) *out_len = (int64_t)o;
    return out;
}
