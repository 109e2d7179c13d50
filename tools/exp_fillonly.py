"""Experiment: throughput of fills alone (score-only or with codes) with several in flight.  Not part of the product."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd
k = 16
n = m = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, m)
per_thread = 8
for codes in (False, True):
    for T in (1, 2, 3, 4, 6):
        stripes = [psa_amd.Stripe(a, b, 0, m, traceback=codes) for _ in range(T)]
        streams = [torch.cuda.Stream() for _ in range(T)]
        def worker(i):
            s = streams[i].cuda_stream
            for _ in range(per_thread):
                stripes[i].fill_rows(0, n, None, None, s)
            stripes[i].final(s)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
            [x.start() for x in th]; [x.join() for x in th]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("codes=%d in flight %d: %.2f ms per fill, %.0f GCUPS" % (codes, T, dt * 1e3 / (T * per_thread), T * per_thread * float(n) * m / dt / 1e9), flush=True)
        for s_ in stripes: s_.close()
