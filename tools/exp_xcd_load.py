"""Experiment: K one-workgroup fills at once, all on XCD 0 (first workgroup of a masked queue lands there), distinct CUs.
Not part of the product."""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psa_amd
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits: words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p(); rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words); assert rc == 0, rc
    return st.value
m = 1 << 16; n = 256
a = psa_amd.synth_protein(0xA0000000 + 16, n); b = psa_amd.synth_protein(0xB0000000 + 16, m)
torch.zeros(1).cuda()
for codes in (True, False):
    stripes = [psa_amd.Stripe(a, b, 0, m, traceback=codes) for _ in range(8)]
    for idxs in ([0], [0, 4], [0, 1, 2, 3], [0, 4, 8, 12], [0, 1, 2, 3, 4, 5, 6, 7], [0, 4, 8, 12, 16, 20, 24, 28]):
        streams = [masked_stream(range(8 * i, 8 * i + 8)) for i in idxs]
        for rep in range(3):
            for s_, st in zip(stripes, streams): s_.fill_rows(0, n, None, None, st)
            for s_, st in zip(stripes, streams): s_.final(st)
        ms = [s_.last_fill_ms() for s_, _ in zip(stripes, streams)]
        print("codes=%d mask indices %s: fill ms %s" % (codes, idxs, [round(x, 3) for x in ms]), flush=True)
        for st in streams: hip.hipStreamDestroy(C.c_void_p(st))
    for s_ in stripes: s_.close()
