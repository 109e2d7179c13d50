import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psa_amd
k = 16; n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
st = psa_amd.Stripe(a, b, 0, n, traceback=False)
st.fill_rows(0, n); st.final()
os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips.csv"
os.environ["PSA_PROFILE_HANDOFF"] = sys.argv[1]
st.fill_rows(0, n); print(st.final(), st.last_fill_ms())
d = np.loadtxt("gpurun_out/handoff.csv", delimiter=",", skiprows=1)
for b in list(range(2000, 2012)):
    blk, look, got, pre_ok, spins, pub = d[b]
    print("blk %d: published %.2f us, consumer looked %.2f, got %.2f (pre_ok %d spins %d) ; got-published = %.2f us" % (blk, pub/100, look/100, got/100, pre_ok, spins, (got-pub)/100))
