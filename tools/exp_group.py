"""Experiment: duration of ONE chained group (4 strips, input granules ready, output to granules) as its own launch.
Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psa_amd
m = 1 << 16; n = 2048
a = psa_amd.synth_protein(0xA0000000 + 16, n); b = psa_amd.synth_protein(0xB0000000 + 16, m)
for codes in (True, False):
    st = psa_amd.Stripe(a, b, 0, m, traceback=codes)
    for rows in (256, 512, 1024):
        for rep in range(2):
            t = []
            r0 = 0
            while r0 < n:
                st.fill_rows(r0, r0 + rows); t.append(st.last_fill_ms()); r0 += rows
            st.final()
        print("codes=%d chunks of %d rows (%d groups per launch): ms per launch %s" % (codes, rows, rows // 256, [round(x, 3) for x in t]), flush=True)
    st.close()
