"""Latency of one complete alignment from host buffers (psa_align) over sizes 2^8 .. 2^16."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
psa = importlib.import_module("parallel-sequence-alignment_amd")

AL = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
rng = np.random.default_rng(7)
for k in range(8, 17):
    n = 1 << k
    a = AL[rng.integers(0, 20, n)].tobytes()
    b = AL[rng.integers(0, 20, n)].tobytes()
    psa.align(a, b)
    reps = 5 if k < 15 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        r = psa.align(a, b)
    dt = (time.perf_counter() - t0) / reps
    print(f"2^{k}: {dt*1e3:9.3f} ms  {n*n/dt/1e9:8.2f} GCUPS  score {r.score}", flush=True)
