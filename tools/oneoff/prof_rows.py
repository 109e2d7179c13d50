"""Diagnostic: end times of the first strips for matrices of growing height against a fixed B.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
m = 1 << 16
codes = len(sys.argv) > 1 and sys.argv[1] == "codes"
b = psa_amd.synth_protein(0xB0000010, m)
os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips.csv"
for rows in (1024, 2048, 8192, 16384, 32768, 65536):
    a = psa_amd.synth_protein(0xA0000010, rows)
    st = psa_amd.Stripe(a, b, 0, m, traceback=codes)
    st.fill_rows(0, rows); st.final()
    st.fill_rows(0, rows); st.final()
    d = np.loadtxt("gpurun_out/strips.csv", delimiter=",", skiprows=1)
    end = d[:, 2] / 100.0
    print("rows=%6d fill %.3f ms; end of strips 0,3,4,7,8: %s ; last %.0f" % (
        rows, st.last_fill_ms(), [round(float(end[i])) for i in (0, 3, 4, 7, 8)], end[-1]), flush=True)
