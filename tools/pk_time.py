"""One engine, S slots of 2^16 x 2^16 pairs, three launches: fill and walk durations of each (HIP events).  With a
timing-only library variant (PSA_LIB_PATH, tools/x_sweep.sh) the results are wrong and the walk's error is printed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, psa_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = m = 1 << 16
eng = psa_amd.Many(S, n, m, traceback=True)
for sl in range(S):
    eng.set_pair(sl, psa_amd.synth_protein(0xA + 256 * sl, n), psa_amd.synth_protein(0xB0000000 + 256 * sl, m))
for rep in range(3):
    eng.launch()
    try:
        eng.wait(); 
    except Exception as e:
        print("wait:", e)
    print("fill ms %.2f walk ms %.2f" % eng.last_ms(), flush=True)
eng.close()
