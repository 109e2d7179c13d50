"""A 60-symbol alphabet (mixed-case FASTA with ambiguity codes) against a 20-symbol one on pairs of the same size, one pair
per call, fill with codes.  Until round 3 more than 32 symbols fell to the generic kernel fill_strips_v1 (run this with
PSA_FILL_KERNEL=v1 to see it); now the table kernel takes any byte alphabet."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psa_amd
rng = np.random.default_rng(9)
for k in (13, 14, 15):
    n = 1 << k
    for alpha, name in ((20, "20 symbols"), (60, "60 symbols")):
        a = rng.integers(40, 40 + alpha, n).astype(np.uint8); b = rng.integers(40, 40 + alpha, n).astype(np.uint8)
        st = psa_amd.Stripe(a, b, 0, n, traceback=True)
        ms = []
        for rep in range(3):
            st.fill_rows(0, n); st.final(); ms.append(st.last_fill_ms())
        st.close()
        print(json.dumps({"log2": k, "alphabet": name, "kernel": "generic (PSA_FILL_KERNEL=v1)" if os.environ.get("PSA_FILL_KERNEL") == "v1" else "table", "fill_ms_with_codes": round(min(ms[1:]), 3),
                          "gcups": round(float(n) * n / (min(ms[1:]) * 1e-3) / 1e9, 1)}), flush=True)
