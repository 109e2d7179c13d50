"""Experiment: two one-workgroup fills on two single CUs chosen by CU mask: which pairs of CUs slow each other down?
Not part of the product."""
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
codes = (sys.argv[1] != "score") if len(sys.argv) > 1 else True
m = 1 << 16; n = 256
a = psa_amd.synth_protein(0xA0000000 + 16, n); b = psa_amd.synth_protein(0xB0000000 + 16, m)
torch.zeros(1).cuda()
s0 = psa_amd.Stripe(a, b, 0, m, traceback=codes); s1 = psa_amd.Stripe(a, b, 0, m, traceback=codes)
def run_pair(b0, b1):
    st0 = masked_stream(range(8 * b0, 8 * b0 + 8)); st1 = masked_stream(range(8 * b1, 8 * b1 + 8)) if b1 is not None else None
    out = []
    for rep in range(3):
        s0.fill_rows(0, n, None, None, st0)
        if st1: s1.fill_rows(0, n, None, None, st1)
        s0.final(st0)
        if st1: s1.final(st1)
        out.append((s0.last_fill_ms(), s1.last_fill_ms() if st1 else 0.0))
    hip.hipStreamDestroy(C.c_void_p(st0))
    if st1: hip.hipStreamDestroy(C.c_void_p(st1))
    return out[-1]
print("alone on bit 0: %.3f ms" % run_pair(0, None)[0])
print("alone on bit 5: %.3f ms" % run_pair(5, None)[0])
for partner in range(1, 32):
    r = run_pair(0, partner)
    print("CU index (per XCD) 0 + %3d: %.3f ms  %.3f ms" % (partner, r[0], r[1]), flush=True)
