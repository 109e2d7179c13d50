"""Experiment: throughput of independent 2^16 alignments with several of them in flight (one host thread,
stripe object and stream each).  Not part of the product."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd
k = 16
n = m = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, m)
a_dev = torch.from_numpy(a).cuda(); b_dev = torch.from_numpy(b).cuda()
per_thread = int(sys.argv[1]) if len(sys.argv) > 1 else 6
phases = {}


def worker(idx, st, stream, out):
    cap = 2 * (n + m) + 2
    ai = np.zeros(cap, np.int32); bj = np.zeros(cap, np.int32)
    s = stream.cuda_stream
    acc = [0.0] * 5
    for _ in range(per_thread):
        t0 = time.perf_counter()
        st.set_sequences_dev(a_dev.data_ptr(), b_dev.data_ptr(), s)
        t1 = time.perf_counter()
        st.fill_rows(0, n, None, None, s)
        t2 = time.perf_counter()
        t = st.final(s)
        t3 = time.perf_counter()
        kk, sc = psa_amd.pick(*t)
        _, _, _, ln = st.traceback(n, m, kk, ai, bj, 0, s)
        t4 = time.perf_counter()
        fa, fb = psa_amd.finish_alignment(ai[:ln], bj[:ln], n, m)
        t5 = time.perf_counter()
        for q, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[q] += d
        out[idx] = (sc, len(fa))
    phases[idx] = [round(x / per_thread * 1e3, 2) for x in acc]


for T in (3, 4, 5):
    stripes = [psa_amd.Stripe(a, b, 0, m, traceback=True) for _ in range(T)]
    streams = [torch.cuda.Stream() for _ in range(T)]
    out = [None] * T
    for rep in range(2):   # first repetition warms up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i, stripes[i], streams[i], out)) for i in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("in flight %d: %d alignments in %.1f ms -> %.2f ms each, %.0f GCUPS; results %s" % (
        T, T * per_thread, dt * 1e3, dt * 1e3 / (T * per_thread), T * per_thread * float(n) * m / dt / 1e9, out), flush=True)
    print("   per-thread ms [set_sequences, fill launch, final (wait), traceback, finish]:", [phases[i] for i in range(T)], flush=True)
    for s_ in stripes:
        s_.close()
