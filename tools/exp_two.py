"""Experiment: per-strip durations of tiny fills (1..4 strips).  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, psa_amd
m = 1 << 16
for codes in (True, False):
    for n in (64, 128, 192, 256):
        a = psa_amd.synth_protein(0xA0000000 + 16, n); b = psa_amd.synth_protein(0xB0000000 + 16, m)
        st = psa_amd.Stripe(a, b, 0, m, traceback=codes)
        st.fill_rows(0, n); st.final()
        os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/two.csv"
        st.fill_rows(0, n); st.final()
        os.environ.pop("PSA_PROFILE_STRIPS")
        d = np.loadtxt("gpurun_out/two.csv", delimiter=",", skiprows=1, ndmin=2)
        print("codes=%d strips %d: start us %s end us %s wait_in %s" % (codes, n // 64, [round(x / 100.0, 1) for x in d[:, 1]], [round(x / 100.0, 1) for x in d[:, 2]], [int(x) for x in d[:, 5]]), flush=True)
        st.close()
