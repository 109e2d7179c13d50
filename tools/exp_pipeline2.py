"""Experiment: every host thread keeps one fill in flight all the time -- it launches the fill of its next alignment
(second stripe object) before it walks the previous one on a second stream.  Not part of the product."""
import os, sys, threading, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("PSA_BENCH_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, psa_amd
k = 16; n = m = 1 << k
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, m)
a_dev = torch.from_numpy(a).cuda(); b_dev = torch.from_numpy(b).cuda()
stripes = [[psa_amd.Stripe(a, b, 0, m, traceback=True) for _ in range(2)] for _ in range(T)]
sf = [torch.cuda.Stream() for _ in range(T)]; stb = [torch.cuda.Stream() for _ in range(T)]
cap = 2 * (n + m) + 2
def launch(st, s):
    st.set_sequences_dev(a_dev.data_ptr(), b_dev.data_ptr(), s); st.fill_rows(0, n, None, None, s)
def finish(st, sfill, swalk, ai, bj):
    t = st.final(sfill); kk, sc = psa_amd.pick(*t)
    return t, kk, sc
# first use of every stream alone
ai0 = np.zeros(cap, np.int32); bj0 = np.zeros(cap, np.int32)
for i in range(T):
    for st in stripes[i]:
        launch(st, sf[i].cuda_stream); t = st.final(sf[i].cuda_stream); kk, sc = psa_amd.pick(*t)
        st.traceback(n, m, kk, ai0, bj0, 0, stb[i].cuda_stream)
torch.cuda.synchronize()
def run(count):
    lock = threading.Lock(); state = {"next": 0}; res = []
    def take():
        with lock:
            if state["next"] >= count: return False
            state["next"] += 1; return True
    def worker(i):
        ai = np.zeros(cap, np.int32); bj = np.zeros(cap, np.int32)
        time.sleep(0.004 * i)
        cur = 0
        if not take(): return
        launch(stripes[i][cur], sf[i].cuda_stream)
        while True:
            st = stripes[i][cur]
            t = st.final(sf[i].cuda_stream); kk, sc = psa_amd.pick(*t)
            more = take()
            if more: launch(stripes[i][cur ^ 1], sf[i].cuda_stream)
            _, _, _, ln = st.traceback(n, m, kk, ai, bj, 0, stb[i].cuda_stream)
            fa, fb = psa_amd.finish_alignment(ai[:ln], bj[:ln], n, m)
            with lock: res.append((sc, len(fa)))
            if not more: return
            cur ^= 1
    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert len(res) == count and len(set(res)) == 1, res
    return dt
run(8)
for rep in range(3):
    dt = run(steps)
    print("threads %d (one fill in flight each, walk overlapped): %.2f ms per alignment, %.0f GCUPS" % (T, dt * 1e3 / steps, steps * float(n) * m / dt / 1e9), flush=True)
