"""Experiment (run under rocprofv3 --pmc): fills of 1, 2 and 4 strips, three each, with codes.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psa_amd
m = 1 << 16
for n in (64, 128, 256):
    a = psa_amd.synth_protein(0xA0000000 + 16, n); b = psa_amd.synth_protein(0xB0000000 + 16, m)
    st = psa_amd.Stripe(a, b, 0, m, traceback=True)
    for _ in range(3):
        st.fill_rows(0, n); st.final()
    st.close()
