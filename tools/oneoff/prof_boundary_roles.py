"""Diagnostic: end lag between consecutive strips of a lone score-only stripe (W = 16384) by boundary role."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import psa_amd
k = 16; n = 1 << k; W = 16384
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
edge = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
edge2 = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
flags = torch.zeros((n + 63) // 64, dtype=torch.int32, device="cuda")
ones = torch.ones((n + 63) // 64, dtype=torch.int32, device="cuda")
def run(name, c0, fn):
    st = psa_amd.Stripe(a, b, c0, c0 + W, traceback=False)
    os.environ.pop("PSA_PROFILE_STRIPS", None)
    for _ in range(2):
        flags.zero_(); torch.cuda.synchronize(); fn(st); torch.cuda.synchronize()
    ms = st.last_fill_ms()
    os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/strips_w.csv"
    flags.zero_(); torch.cuda.synchronize(); fn(st); torch.cuda.synchronize()
    os.environ.pop("PSA_PROFILE_STRIPS", None)
    d = np.loadtxt("gpurun_out/strips_w.csv", delimiter=",", skiprows=1)
    el = np.diff(d[:, 2] / 100.0)
    print("%-46s fill %.3f ms; end lag by wave-in-wg %s mean %.2f" % (name, ms, [round(float(np.median(el[w::4])), 2) for w in range(4)], el.mean()))
    st.close()
run("no boundaries", 0, lambda st: st.fill_rows(0, n))
run("right boundary, plain", 0, lambda st: st.fill_rows(0, n, None, edge.data_ptr()))
run("left boundary, plain", W, lambda st: st.fill_rows(0, n, edge2.data_ptr(), None))
run("both, plain", W, lambda st: st.fill_rows(0, n, edge2.data_ptr(), edge.data_ptr()))
run("right boundary with flags (flagged)", 0, lambda st: st.fill_flagged(None, None, edge.data_ptr(), flags.data_ptr()))
run("left boundary with flags all set (flagged)", W, lambda st: st.fill_flagged(edge2.data_ptr(), ones.data_ptr(), None, None))
run("no boundaries (flagged entry)", 0, lambda st: st.fill_flagged(None, None, None, None))
