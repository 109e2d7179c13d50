import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psa_amd
k = int(sys.argv[1]); codes = sys.argv[2] != "score"
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
st = psa_amd.Stripe(a, b, 0, n, traceback=codes)
st.fill_rows(0, n); st.final()
os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips.csv"
st.fill_rows(0, n); print(st.final(), st.last_fill_ms())
d = np.loadtxt("gpurun_out/strips.csv", delimiter=",", skiprows=1)
dur_us = (d[:, 2] - d[:, 1]) / 100.0
print("strip: dur_us | cycles per block: input, preload+mask+room, steps, publish+codes")
nb = (n + 63 + 63) // 64 * 4
for s in list(range(0, 10)) + [100, 500, 1000]:
    t0, t1, t2, t3 = d[s, 3], d[s, 4], d[s, 6], d[s, 5]
    print(s, round(dur_us[s], 1), [int(x / nb) for x in (t0, t1, t2, t3)])
