import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import psa_amd
import oracle_lib as O   # tests helper
def run(shapes, seed):
    rng = np.random.default_rng(seed)
    al = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    n_cap = max(s[0] for s in shapes); m_cap = max(s[1] for s in shapes)
    eng = psa_amd.Many(len(shapes), n_cap, m_cap, traceback=True)
    pairs = []
    for sl, (n, m) in enumerate(shapes):
        a = al[rng.integers(0, 20, n)]; b = al[rng.integers(0, 20, m)]
        if sl % 3 == 1:   # related pair
            b = a[:m].copy() if m <= n else np.concatenate([a, al[rng.integers(0, 20, m - n)]])
            mut = rng.random(m) < 0.2
            b[mut] = al[rng.integers(0, 20, int(mut.sum()))]
        pairs.append((a, b)); eng.set_pair(sl, a, b)
    t = time.time(); eng.launch(); eng.wait(); dt = time.time() - t
    bad = 0
    for sl, (a, b) in enumerate(pairs):
        r = eng.result(sl, a.size, b.size)
        sc, ai, bj = O.align(a, b, 4)
        ok = r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj)
        print("shapes", shapes[sl], "slot", sl, "score", r.score, sc, "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
    eng.close()
    return bad
bad = 0
def run_fixed(pairs):
    eng = psa_amd.Many(len(pairs), max(a.size for a, b in pairs), max(b.size for a, b in pairs), traceback=True)
    for sl, (a, b) in enumerate(pairs): eng.set_pair(sl, a, b)
    eng.launch(); eng.wait()
    bad = 0
    for sl, (a, b) in enumerate(pairs):
        r = eng.result(sl, a.size, b.size)
        sc, ai, bj = O.align(a, b, 8)
        ok = r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj)
        print("fixed", a.size, b.size, "slot", sl, "score", r.score, sc, "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
    eng.close()
    return bad
al_ = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
rng_ = np.random.default_rng(9)
same = al_[rng_.integers(0, 20, 20000)]
other = al_[rng_.integers(0, 20, 20000)]
poly = np.full(20000, ord("A"), np.uint8)
polyc = np.full(20000, ord("C"), np.uint8)
bad = 0
bad += run_fixed([(same, same.copy()), (same, other), (poly, poly.copy()), (poly, polyc)])
shifted = np.concatenate([other[:3000], same[:17000]])
bad += run_fixed([(same, shifted), (poly, same)])
bad += run([(300, 256), (300, 256)], 1)
bad += run([(1000, 700), (1000, 700), (640, 700)], 2)
bad += run([(4096, 4096)] * 4, 3)
bad += run([(5000, 3000), (5000, 3000), (333, 3000), (333, 3000), (5000, 3000)], 4)
print("bad", bad)
sys.exit(1 if bad else 0)
