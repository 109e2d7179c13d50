"""Banded (linear-space) alignment against the plain one on the 2^15 pair: equality and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
psa_amd.align(a, b, 1)
t0 = time.perf_counter(); plain = psa_amd.align(a, b, 1); t1 = time.perf_counter()
for band in (4096, 1024, 256):
    psa_amd.align(a, b, 1, band_rows=band)
    t2 = time.perf_counter(); banded = psa_amd.align(a, b, 1, band_rows=band); t3 = time.perf_counter()
    same = plain.score == banded.score and np.array_equal(plain.ai, banded.ai) and np.array_equal(plain.bj, banded.bj)
    print("2^%d: plain %.1f ms, banded(%d rows) %.1f ms, identical: %s, score %d, length %d" % (
        k, (t1 - t0) * 1e3, band, (t3 - t2) * 1e3, same, plain.score, len(plain.ai)), flush=True)
