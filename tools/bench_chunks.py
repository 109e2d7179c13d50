"""Single-GPU proxy for the multi-GPU stripe pipeline: one stripe filled in row chunks, each
chunk its own launch, consecutive chunks on alternating streams so that chunk c+1 starts while chunk c runs (its
first strip polls the granules chunk c's last strip publishes) -- against the same stripe filled by one launch.
Also psa_score with p = 1 / 4 / 8 column stripes on one GPU (all stripes jobs of ONE launch).  Prints JSON."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd


def chunked(log2n, W, chunk, reps=3):
    n = 1 << log2n
    a = psa_amd.synth_protein(0xA0000000 + log2n, n)
    b = psa_amd.synth_protein(0xB0000000 + log2n, max(W, 1 << log2n))[:W]
    st = psa_amd.Stripe(a, b, 0, W, traceback=False)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s_ in streams:   # first use of each stream before anything is timed
        torch.zeros(1, device="cuda").record_stream(s_)
    out = {}
    for mode in ("one_launch", "chunks_alternating_streams", "chunks_one_stream"):
        ts = []
        for rep in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "one_launch":
                st.fill_rows(0, n, None, None, streams[0].cuda_stream)
            else:
                for c, r0 in enumerate(range(0, n, chunk)):
                    sm = streams[c % 2] if mode == "chunks_alternating_streams" else streams[0]
                    st.fill_rows(r0, min(n, r0 + chunk), None, None, sm.cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            t = st.final(streams[0].cuda_stream)
            out.setdefault("triple", t)
            assert out["triple"] == t, (mode, t, out["triple"])
        out[mode + "_ms"] = min(ts[1:])
    st.close()
    out["chunked_over_unchunked"] = out["chunks_alternating_streams_ms"] / out["one_launch_ms"]
    out["serial_chunks_over_unchunked"] = out["chunks_one_stream_ms"] / out["one_launch_ms"]
    out.update({"log2n": log2n, "W": W, "chunk_rows": chunk})
    return out


def stripes_p(log2n=16, ps=(1, 4, 8)):
    n = 1 << log2n
    a = psa_amd.synth_protein(0xA0000000 + log2n, n)
    b = psa_amd.synth_protein(0xB0000000 + log2n, n)
    out = {}
    for p in ps:
        psa_amd.score(a, b, p)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            sc = psa_amd.score(a, b, p)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["p=%d" % p] = {"ms": min(ts), "score": sc}
    psa_amd.release_cache()
    for p in ps[1:]:
        out["p=%d" % p]["over_p1"] = out["p=%d" % p]["ms"] / out["p=1"]["ms"]
    return out


if __name__ == "__main__":
    res = {"chunked": [chunked(16, 1 << 13, 4096), chunked(18, 1 << 17, 4096)], "psa_score_stripes_2p16": stripes_p()}
    print(json.dumps(res, indent=1))
