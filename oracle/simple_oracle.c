/*
 * simple_oracle.c -- command-line front end of the CPU restatement, same argv / stdout
 * contract as the reference's main (simple.cpp:297-313).  TEST INFRASTRUCTURE ONLY.
 * Used to produce restatement-only goldens for inputs the reference cannot run (titin)
 * and as the multi-core CPU baseline binary.
 */
#include "psa_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "Usage: %s <sequence1.fasta> <sequence2.fasta> <p>\n", argv[0]);
        return 1;
    }
    char pa[4096], pb[4096];
    snprintf(pa, sizeof pa, "sequences/%s", argv[1]);
    snprintf(pb, sizeof pb, "sequences/%s", argv[2]);
    int p = atoi(argv[3]);
    int64_t n, m, len = 0, out_len = 0;
    uint8_t* a = psa_oracle_read_fasta(pa, &n);
    uint8_t* b = psa_oracle_read_fasta(pb, &m);
    int64_t cap = 2 * (n + m) + 2;
    int32_t* ai = (int32_t*)malloc((size_t)cap * 4);
    int32_t* bj = (int32_t*)malloc((size_t)cap * 4);
    int32_t score = 0;
    int rc = psa_oracle_align_mt(p, a, n, b, m, &score, ai, bj, cap, &len);
    if (rc) { fprintf(stderr, "oracle failed: %d\n", rc); return 2; }
    char* txt = psa_oracle_format(score, a, b, ai, bj, len, &out_len);
    fwrite(txt, 1, (size_t)out_len, stdout);
    free(txt); free(ai); free(bj); free(a); free(b);
    return 0;
}
