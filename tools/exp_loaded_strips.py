"""Experiment: per-strip timeline of fills while several are in flight (PSA_PROFILE_STRIPS=path+).  Not part of the product."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd
k = 16
n = m = 1 << k
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
codes = (sys.argv[2] != "score") if len(sys.argv) > 2 else True
a = psa_amd.synth_protein(0xA0000000 + k, n); b = psa_amd.synth_protein(0xB0000000 + k, m)
stripes = [psa_amd.Stripe(a, b, 0, m, traceback=codes) for _ in range(T)]
streams = [torch.cuda.Stream() for _ in range(T)]
os.makedirs("gpurun_out/loaded", exist_ok=True)
for f in glob.glob("gpurun_out/loaded/s*"): os.remove(f)
PROFILE_AT = -1
def worker(i, reps):
    s = streams[i].cuda_stream
    for r in range(reps):
        if i == 0 and r == PROFILE_AT: os.environ["PSA_PROFILE_STRIPS"] = "gpurun_out/loaded/s+"
        stripes[i].fill_rows(0, n, None, None, s)
        if i == 0 and r == PROFILE_AT: os.environ.pop("PSA_PROFILE_STRIPS", None)
    stripes[i].final(s)
def run(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, reps)) for i in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); return time.perf_counter() - t0
run(2)
dt = run(6)
print("unprofiled: %.2f ms per fill, %.0f GCUPS" % (dt * 1e3 / (6 * T), 6 * T * float(n) * m / dt / 1e9))
PROFILE_AT = 5
dt = run(10)
print("profiled: %.2f ms per fill" % (dt * 1e3 / (10 * T)))
for f in sorted(glob.glob("gpurun_out/loaded/s*")):
    d = np.loadtxt(f, delimiter=",", skiprows=1)
    start, end = d[:, 1] / 100.0, d[:, 2] / 100.0
    dur = end - start
    wg_start = start[0::4]; wg_end = end[3::4] if len(end) % 4 == 0 else end[3::4]
    print(os.path.basename(f), "total %.0f us; strip life us: med %.0f p10 %.0f p90 %.0f; wg start lag med %.1f p90 %.1f max %.0f; wait_in med %d" % (
        end.max(), np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90),
        np.median(np.diff(wg_start)), np.percentile(np.diff(wg_start), 90), np.diff(wg_start).max(), np.median(d[:, 5])))
    # workgroups resident at the same time (of this fill)
    ev = sorted([(s_, 1) for s_ in wg_start] + [(e_, -1) for e_ in end[3::4]])
    cur = mx = 0; area = 0.0; last = 0.0
    for t_, dlt in ev:
        area += cur * (t_ - last); last = t_; cur += dlt; mx = max(mx, cur)
    print("   resident workgroups: max %d, mean %.1f" % (mx, area / end.max()))
