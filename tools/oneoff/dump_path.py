import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import psa_amd
k = 16; n = 1 << k
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, n)
st = psa_amd.Stripe(a, b, 0, n, traceback=True)
st.fill_rows(0, n); t = st.final()
ai = np.zeros(2 * n + 8, np.int32); bj = np.zeros(2 * n + 8, np.int32)
r = st.traceback(n, n, psa_amd.pick(*t)[0], ai, bj, 0)
ln = r[3]
os.makedirs("gpurun_out/r2", exist_ok=True)
np.savez_compressed("gpurun_out/r2/path16.npz", ai=ai[:ln], bj=bj[:ln], k0=psa_amd.pick(*t)[0])
print("len", ln)
