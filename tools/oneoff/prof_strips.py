"""Diagnostic: per-strip timeline of one fill (PSA_PROFILE_STRIPS).  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
codes = (sys.argv[2] != "score") if len(sys.argv) > 2 else True
n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
st = psa_amd.Stripe(a, b, 0, n, traceback=codes)
st.fill_rows(0, n); st.final()
os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips.csv"
st.fill_rows(0, n); print(st.final(), st.last_fill_ms())
d = np.loadtxt("gpurun_out/strips.csv", delimiter=",", skiprows=1)
start, end = d[:, 1] / 100.0, d[:, 2] / 100.0   # us
dur = end - start
print("strips", len(d), "total us", end.max())
print("dur us: min %.1f med %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
lag = np.diff(start)
print("start lag us: med %.2f mean %.2f ; by wave-in-wg:" % (np.median(lag), lag.mean()),
      [round(float(np.median(lag[w::4])), 2) for w in range(4)])
endlag = np.diff(end)
print("end lag us: med %.2f mean %.2f ; by wave-in-wg:" % (np.median(endlag), endlag.mean()),
      [round(float(np.median(endlag[w::4])), 2) for w in range(4)])
hw = d[:, 4].astype(np.int64); xcc = d[:, 3].astype(np.int64)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
key = xcc * 1000 + se * 100 + cu
u, c = np.unique(key[::4], return_counts=True)
print("distinct (xcc,se,cu) used by workgroups:", len(u), "max WGs on one CU:", c.max())
print("wait_in spins: med %d max %d; by wave %s ; wait_room: med %d max %d" % (np.median(d[:,5]), d[:,5].max(), [int(np.median(d[w::4,5])) for w in range(4)], np.median(d[:,6]), d[:,6].max()))
print("first 12 strips:", [(int(x), round(float(s_), 1), round(float(e), 1)) for x, s_, e in zip(d[:12, 0], start[:12], end[:12])])
big = np.argsort(endlag)[-12:]
print("largest end lags (strip, us):", [(int(i_) + 1, round(float(endlag[i_]), 1)) for i_ in sorted(big)])
print("end lag percentiles 50/90/99/99.9:", [round(float(np.percentile(endlag, q_)), 2) for q_ in (50, 90, 99, 99.9)])
for i_ in sorted(big)[:4]:
    lo = max(0, i_ - 3)
    print("around strip %d:" % (i_ + 1), [(int(d[x, 0]), round(float(start[x]), 1), round(float(end[x]), 1), int(d[x, 5]), int(xcc[x])) for x in range(lo, min(len(d), i_ + 6))])
