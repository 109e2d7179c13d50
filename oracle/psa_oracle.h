/*
 * psa_oracle.h -- CPU restatement of the reference's `simple.cpp` alignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load or run it, and there only
 * as the checker / the reported CPU baseline.  The product path (libpsa_hip.so, ./simple) never
 * links or calls it.
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte against stdout of the real
 * reference (compiled from /root/reference/simple.cpp by oracle/Makefile into oracle/_ref/) on
 * every fixture in tests/golden/ (see tests/golden/make_golden.py and tests/test_oracle.py).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 */
#ifndef PSA_ORACLE_H
#define PSA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Score only (rolling rows, O(m) memory).  simple.cpp:107-133,157-213,244-256. */
int psa_oracle_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                     int32_t* score_out, int32_t final_triple_out[3]);

/*
 * Full alignment, single thread: fill with one byte of winner codes per cell, walk back,
 * apply the prefix fix-up (simple.cpp:261-294).  ai/bj receive the alignment columns
 * (0-based indices, -1 = gap); cap is their capacity (2*(n+m)+2 always suffices).
 * Returns 0, or -1 if cap is too small, -2 on allocation failure.
 */
int psa_oracle_align(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                     int32_t* score_out, int32_t* ai, int32_t* bj, int64_t cap,
                     int64_t* len_out);

/*
 * Same result computed the way the reference partitions the work: p column stripes of B
 * (block = m/p, last stripe takes the remainder, simple.cpp:90-99), one thread per stripe,
 * per-row boundary hand-off left to right (simple.cpp:136-154, 215-235).  Codes replace the
 * per-cell alignment vectors.  Used as the multi-core CPU baseline.
 */
int psa_oracle_align_mt(int p, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                        int32_t* score_out, int32_t* ai, int32_t* bj, int64_t cap,
                        int64_t* len_out);

/* Multi-threaded score-only variant of the above (no code matrix). */
int psa_oracle_score_mt(int p, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                        int32_t* score_out);

/*
 * One rectangular block of the three tables with explicit boundaries, in the plain
 * (unbiased) domain -- the checker for the HIP block entry point.  Rows r0+1..r1 and columns
 * c0+1..c1 of the global matrix are filled.  top[k][x] = T_k(r0, c0+x) for x = 0..c1-c0 and
 * left[k][y] = T_k(r0+y, c0) for y = 0..r1-r0 are inputs; bottom / right are the same
 * lines at row r1 / column c1.  Any pointer among the outputs may be NULL.
 */
int psa_oracle_block(const uint8_t* a, const uint8_t* b, int64_t r0, int64_t r1, int64_t c0,
                     int64_t c1, const int32_t* top[3], const int32_t* left[3],
                     int32_t* bottom[3], int32_t* right[3]);

/* Same, also writing the cell's packed winners w1 | w2<<2 | w3<<4 (destination-indexed, as in
 * psa_oracle_align) to codes[(y-1)*(c1-c0) + (x-1)] for row r0+y, column c0+x. */
int psa_oracle_block_codes(const uint8_t* a, const uint8_t* b, int64_t r0, int64_t r1, int64_t c0,
                           int64_t c1, const int32_t* top[3], const int32_t* left[3],
                           int32_t* bottom[3], int32_t* right[3], uint8_t* codes);

/* Analytic boundary values of the reference (simple.cpp:116-133, simple.h:50). */
void psa_oracle_boundary(int64_t i, int64_t j, int32_t t[3]);

/*
 * The reference's FASTA reader (read_fasta.cpp:8-23): every non-empty line that does not
 * start with '>' is appended verbatim ('\r' kept).  Returns a malloc'd buffer (caller frees)
 * and its length; on open failure prints the reference's message to stderr and returns an
 * empty sequence (*len_out = 0, return value may be NULL).
 */
uint8_t* psa_oracle_read_fasta(const char* path, int64_t* len_out);

/*
 * The reference's stdout (simple.cpp:312-313, 43-70): "Score: ..." plus four lines.
 * Returns a malloc'd NUL-terminated buffer (it may contain arbitrary residue bytes);
 * *out_len receives its length.
 */
char* psa_oracle_format(int32_t score, const uint8_t* a, const uint8_t* b, const int32_t* ai,
                        const int32_t* bj, int64_t len, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
