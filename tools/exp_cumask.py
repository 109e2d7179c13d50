"""Experiment: fills on streams restricted to disjoint CU sets (hipExtStreamCreateWithCUMask).  Not part of the product."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits: words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words)
    assert rc == 0, rc
    return st.value
k = 16; n = m = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, m)
torch.cuda.init(); torch.zeros(1).cuda()
per_thread = 8
for codes in (True, False):
  for layout in ("contig", "strided"):
    for P in (2, 4, 8):
        share = 256 // P
        sets = [list(range(i * share, (i + 1) * share)) if layout == "contig" else list(range(i, 256, P)) for i in range(P)]
        streams = [masked_stream(s) for s in sets]
        stripes = [psa_amd.Stripe(a, b, 0, m, traceback=codes) for _ in range(P)]
        def worker(i):
            for _ in range(per_thread):
                stripes[i].fill_rows(0, n, None, None, streams[i])
            stripes[i].final(streams[i])
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
            [x.start() for x in th]; [x.join() for x in th]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("codes=%d %s partitions %d x %d CUs: %.2f ms per fill, %.0f GCUPS, one fill %.2f ms" % (
            codes, layout, P, share, dt * 1e3 / (P * per_thread), P * per_thread * float(n) * m / dt / 1e9, stripes[0].last_fill_ms()), flush=True)
        for s_ in stripes: s_.close()
        for st in streams: hip.hipStreamDestroy(C.c_void_p(st))
