"""Diagnostic: same fill with and without the per-strip profile.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
m = 1 << 16
b = psa_amd.synth_protein(0xB0000010, m)
for rows in (320, 1024, 65536):
    a = psa_amd.synth_protein(0xA0000010, rows)
    st = psa_amd.Stripe(a, b, 0, m, traceback=False)
    for rep in range(2):
        os.environ.pop("PSA_PROFILE_STRIPS", None)
        st.fill_rows(0, rows); st.final(); t0 = st.last_fill_ms()
        os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips.csv"
        st.fill_rows(0, rows); st.final(); t1 = st.last_fill_ms()
        print("rows=%6d  plain %.3f ms   profiled %.3f ms" % (rows, t0, t1), flush=True)
