"""Diagnostic: per-strip timeline of psa_score(p) in one launch (PSA_PROFILE_STRIPES).  Not part of the product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
for q in (1, p):
    for _ in range(3):
        t0 = time.perf_counter(); sc = psa_amd.score(a, b, q); dt = time.perf_counter() - t0
    print("p =", q, "score", sc, "ms %.3f" % (dt * 1e3))
path = "gpurun_out/stripes_p%d.csv" % p
os.environ["PSA_PROFILE_STRIPES"] = path
print(psa_amd.score(a, b, p))
d = np.loadtxt(path, delimiter=",", skiprows=1)
W = n // p
for j in range(p):
    e = d[d[:, 0] == j]
    start, end = e[:, 2] / 100.0, e[:, 3] / 100.0
    lag = np.diff(start)
    print("stripe %d: first start %.1f us, last end %.1f us; strip duration med %.1f us (min %.1f max %.1f); start lag med %.2f mean %.2f p99 %.2f; wait_in med %d"
          % (j, start[0], end.max(), np.median(end - start), (end - start).min(), (end - start).max(), np.median(lag), lag.mean(), np.percentile(lag, 99), np.median(e[:, 6])))
    if j:
        prev = d[d[:, 0] == j - 1]
        h = start - prev[:, 3] / 100.0
        print("   start(j, r) - end(j-1, r): med %.1f us, min %.1f, max %.1f; at r = 0, 256, 512, 768, 1023:" % (np.median(h), h.min(), h.max()),
              [round(float(h[min(r, len(h) - 1)]), 1) for r in (0, 256, 512, 768, 1023)])
