"""Linear-space alignment at sizes whose codes strain or exceed HBM: 2^18 (plain vs banded, identical?)
and 2^20 (banded only: 825 GB of codes would be needed otherwise)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
ks = [int(x) for x in sys.argv[1:]] or [18, 20]
for k in ks:
    n = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
    t0 = time.perf_counter(); banded = psa_amd.align(a, b, 1, band_rows=0); t1 = time.perf_counter()
    h = hashlib.md5(banded.ai.tobytes() + banded.bj.tobytes()).hexdigest()
    print("2^%d banded: %.2f s, score %d, length %d, md5(columns) %s, %.0f GCUPS end to end" % (
        k, t1 - t0, banded.score, len(banded.ai), h, float(n) * n / (t1 - t0) / 1e9), flush=True)
    if k <= 18:
        t0 = time.perf_counter(); plain = psa_amd.align(a, b, 1); t1 = time.perf_counter()
        same = plain.score == banded.score and np.array_equal(plain.ai, banded.ai) and np.array_equal(plain.bj, banded.bj)
        print("2^%d plain : %.2f s, identical to banded: %s" % (k, t1 - t0, same), flush=True)
    else:
        sc = psa_amd.score(a, b)
        print("2^%d score-only fill agrees: %s (%d)" % (k, sc == banded.score, sc), flush=True)
