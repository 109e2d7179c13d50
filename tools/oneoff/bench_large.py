"""One pair, larger than the headline size: score-only 2^17 / 2^18 and full alignment 2^17 (codes 12.9 GB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psa_amd
out = {}
for k, codes in ((17, False), (18, False), (17, True)):
    n = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
    st = psa_amd.Stripe(a, b, 0, n, traceback=codes)
    st.fill_rows(0, n); t = st.final()
    st.fill_rows(0, n); t = st.final()
    rec = {"fill_ms": st.last_fill_ms(), "fill_gcups": float(n) * n / (st.last_fill_ms() * 1e-3) / 1e9, "score": int(max(t))}
    if codes:
        ai = np.zeros(2 * n + 8, np.int32); bj = np.zeros(2 * n + 8, np.int32)
        kk, sc = psa_amd.pick(*t)
        _, _, _, ln = st.traceback(n, n, kk, ai, bj, 0)
        rec["traceback_ms"] = st.last_traceback_ms(); rec["alignment_len"] = int(ln)
    st.close()
    out["2p%d_%s" % (k, "codes" if codes else "score_only")] = rec
    print(json.dumps({("2p%d_%s" % (k, "codes" if codes else "score_only")): rec}), flush=True)
