"""One pair per launch at sizes from 2^16 to 2^20: the fill alone (score only; with codes where they fit), the
one-wavefront-per-SIMD kernel against the two-per-SIMD one (PSA_FILL_DENSE_MIN_W picks; the library reads it once, so
every setting is a process of its own: this script re-runs itself).  Usage: bench_single_large.py [k ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PSA_BSL_CHILD"):
    import psa_amd
    k, codes = int(sys.argv[1]), sys.argv[2] == "codes"
    n = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
    st = psa_amd.Stripe(a, b, 0, n, traceback=codes)
    ms = []
    for rep in range(3 if k < 20 else 2):
        st.fill_rows(0, n); t = st.final(); ms.append(st.last_fill_ms())
    st.close()
    best = min(ms[1:])
    print(json.dumps({"log2": k, "codes": codes, "dense_min_w": os.environ.get("PSA_FILL_DENSE_MIN_W"), "fill_ms": best,
                      "gcups": float(n) * n / (best * 1e-3) / 1e9, "score": int(max(t)),
                      "valu_frac": float(n) * n * (11 if codes else 8) / (best * 1e-3) / 1e12 / 39.3216}), flush=True)
else:
    ks = [int(x) for x in sys.argv[1:]] or [17, 18, 19, 20]
    for k in ks:
        for codes in (["score", "codes"] if k <= 18 else ["score"]):
            for minw in ("1", str(1 << 30)):      # dense whenever there are 1 024 workgroups / never
                env = dict(os.environ, PSA_BSL_CHILD="1", PSA_FILL_DENSE_MIN_W=minw)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k), codes], env=env, capture_output=True, text=True, timeout=300)
                print(r.stdout.strip() or r.stderr[-300:], flush=True)
