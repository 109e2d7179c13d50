"""Diagnostic: end lag between consecutive strips of a lone score-only fill as a function of the stripe width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
k = 16
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
for W in [int(x) for x in sys.argv[1:]] or [8192, 16384, 32768, 65536]:
    st = psa_amd.Stripe(a, b[:W].copy(), 0, W, traceback=False)
    os.environ.pop("PSA_PROFILE_STRIPS", None)
    st.fill_rows(0, n); st.final()
    st.fill_rows(0, n); st.final(); ms = st.last_fill_ms()
    os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips_w.csv"
    st.fill_rows(0, n); st.final()
    d = np.loadtxt("gpurun_out/strips_w.csv", delimiter=",", skiprows=1)
    end = d[:, 2] / 100.0
    el = np.diff(end)
    print("W %6d: fill %.3f ms; first strip ends %.1f us, last %.1f; end lag by wave-in-wg %s mean %.2f; first 8 lags %s; lags 500..508 %s"
          % (W, ms, end[0], end.max(), [round(float(np.median(el[w::4])), 2) for w in range(4)], el.mean(),
             [round(float(x), 1) for x in el[:8]], [round(float(x), 1) for x in el[500:508]]))
    st.close() if hasattr(st, "close") else None
