"""Experiment: where does a one-workgroup fill land for a given CU mask index?  Not part of the product."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psa_amd
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits: words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p(); rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words); assert rc == 0, rc
    return st.value
m = 1 << 14; n = 256
a = psa_amd.synth_protein(1, n); b = psa_amd.synth_protein(2, m)
torch.zeros(1).cuda()
s0 = psa_amd.Stripe(a, b, 0, m, traceback=False)
os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/place.csv"
for idx in list(range(0, 32)):
    st = masked_stream(range(8 * idx, 8 * idx + 8))
    where = []
    for rep in range(3):
        s0.fill_rows(0, n, None, None, st); s0.final(st)
        d = np.loadtxt("gpurun_out/place.csv", delimiter=",", skiprows=1)
        hw = int(d[0, 4]); where.append((int(d[0, 3]), (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15))
    print("mask index %2d -> (xcc, se, sh, cu) %s" % (idx, where), flush=True)
    hip.hipStreamDestroy(C.c_void_p(st))
