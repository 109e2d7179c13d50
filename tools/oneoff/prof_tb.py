"""Diagnostic: traceback statistics of the 2^k pair (PSA_TRACE_STATS).  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
st = psa_amd.Stripe(a, b, 0, n, traceback=True)
os.environ["PSA_TRACE_STATS"] = "1"
for rep in range(3):
    st.fill_rows(0, n); t = st.final()
    ai = np.zeros(2 * n + 8, np.int32); bj = np.zeros(2 * n + 8, np.int32)
    r = st.traceback(n, n, psa_amd.pick(*t)[0], ai, bj, 0)
    print("fill %.3f ms traceback %.3f ms" % (st.last_fill_ms(), st.last_traceback_ms()), flush=True)
