"""psa_align_many from HOST buffers at the bench's shape: what a caller of the C ABI's run-in-a-loop entry point gets
(engines created inside the first call and kept for the next, residues uploaded, alignments returned in malloc'ed arrays)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import psa_amd
from psa_amd import binding as B
k = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = m = 1 << 16
seqs = [(psa_amd.synth_protein(0xA + 256 * (s % 24), n), psa_amd.synth_protein(0xB0000000 + 256 * (s % 24), m)) for s in range(k)]
first = None
for rep in range(4):
    t0 = time.perf_counter()
    res = psa_amd.align_many(seqs)
    dt = time.perf_counter() - t0
    sc = [r.score for r in res]
    first = first or sc
    assert sc == first
    print("pairs %d  %.1f ms  %.0f GCUPS  (scores %s ...)" % (k, dt * 1e3, k * float(n) * m / dt / 1e9, sc[:3]), flush=True)
# a smaller call is served by the same engines; a single-pair call makes them go; the next call builds them again
t0 = time.perf_counter(); res = psa_amd.align_many(seqs[:7]); dt = time.perf_counter() - t0
assert [r.score for r in res] == first[:7]
print("pairs 7  %.1f ms" % (dt * 1e3), flush=True)
one = psa_amd.align(seqs[0][0], seqs[0][1]); assert one.score == first[0]
t0 = time.perf_counter(); res = psa_amd.align_many(seqs); dt = time.perf_counter() - t0
assert [r.score for r in res] == first
print("pairs %d after a single-pair call  %.1f ms" % (k, dt * 1e3), flush=True)
psa_amd.release_cache()
