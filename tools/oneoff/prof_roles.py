"""Diagnostic: fill time of very short matrices (1, 2, 4, 5, 8 strips) against a long B, to read
the intrinsic per-step cost of each strip role.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
m = 1 << 16


def report(tag, t):
    print("%-46s %.3f ms  %.1f ns/step  %.0f cycles/step@2.37GHz" % (
        tag, t, t * 1e6 / (m + 63), t * 1e6 / (m + 63) * 2.37), flush=True)


b = psa_amd.synth_protein(0xB0000010, m)
for codes in (False, True):
    for rows in (64, 128, 192, 256, 320, 512, 1024):
        a = psa_amd.synth_protein(0xA0000010, rows)
        st = psa_amd.Stripe(a, b, 0, m, traceback=codes)
        st.fill_rows(0, rows); st.final()
        ts = []
        for _ in range(3):
            st.fill_rows(0, rows); st.final(); ts.append(st.last_fill_ms())
        report("codes=%d rows=%4d strips=%2d" % (codes, rows, rows // 64), min(ts))
    # roles in isolation: 4 strips whose last one publishes to HBM, then one strip reading HBM
    a = psa_amd.synth_protein(0xA0000010, 320)
    st = psa_amd.Stripe(a, b, 0, m, traceback=codes)
    t1, t2 = [], []
    for _ in range(3):
        st.fill_rows(0, 256); t1.append(st.last_fill_ms())
        st.fill_rows(256, 320); t2.append(st.last_fill_ms())
        st.final()
    report("codes=%d 4 strips, last one to HBM" % codes, min(t1))
    report("codes=%d 1 strip from HBM (granules ready)" % codes, min(t2))
