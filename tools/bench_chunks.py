"""Single-GPU proxy for the multi-GPU stripe pipeline: one stripe filled in row chunks, each
chunk its own launch, consecutive chunks on alternating streams so that chunk c+1 starts while chunk c runs (its
first strip polls the granules chunk c's last strip publishes) -- against the same stripe filled by one launch.
Also psa_score with p = 1 / 4 / 8 column stripes on one GPU (all stripes jobs of ONE launch).  Prints JSON."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import psa_amd


def chunked(log2n, W, chunk, reps=3, nstreams=2):
    n = 1 << log2n
    a = psa_amd.synth_protein(0xA0000000 + log2n, n)
    b = psa_amd.synth_protein(0xB0000000 + log2n, max(W, 1 << log2n))[:W]
    st = psa_amd.Stripe(a, b, 0, W, traceback=False)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
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
                    sm = streams[c % nstreams] if mode == "chunks_alternating_streams" else streams[0]
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
    out.update({"log2n": log2n, "W": W, "chunk_rows": chunk, "streams": nstreams})
    return out


def flagged(log2n, W, chunk, reps=3):
    """One stripe of the multi-GPU pipeline on one GPU: the fill as ONE launch (psa_stripe_fill_flagged) whose left
    boundary is posted in `chunk`-row pieces (as if a fast neighbour delivered them) and whose right boundary is taken
    in pieces while it runs -- against the same stripe filled by one plain launch with its boundaries whole."""
    n = 1 << log2n
    a = psa_amd.synth_protein(0xA0000000 + log2n, n)
    b = psa_amd.synth_protein(0xB0000000 + log2n, 2 * W)
    head = psa_amd.Stripe(a, b, 0, W, traceback=False)
    edge = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
    head.fill_rows(0, n, None, edge.data_ptr())
    torch.cuda.synchronize()
    head.close()
    st = psa_amd.Stripe(a, b, W, 2 * W, traceback=False)
    strips = (n + 63) // 64
    f = psa_amd.stream_create(-1)
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
    right = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
    packed, outs = [], []
    e3 = edge.view(3, n)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        packed.append(e3[:, r0:r1].contiguous().view(-1))
        outs.append(torch.empty(3 * (r1 - r0), dtype=torch.int32, device="cuda"))
    out = {}
    for mode in ("one_launch", "flagged"):
        ts = []
        for rep in range(reps + 1):
            left = torch.zeros(3 * n, dtype=torch.int32, device="cuda")
            lr = torch.zeros(strips, dtype=torch.int32, device="cuda")
            rd = torch.zeros(strips, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "one_launch":
                st.fill_rows(0, n, edge.data_ptr(), right.data_ptr(), f)
            else:
                st.fill_flagged(left.data_ptr(), lr.data_ptr(), right.data_ptr(), rd.data_ptr(), f)
                for c, r0 in enumerate(range(0, n, chunk)):
                    rows = min(n, r0 + chunk) - r0
                    psa_amd.edge_post(packed[c].data_ptr(), r0, rows, n, left.data_ptr(), lr.data_ptr(), s_in.cuda_stream)
                    st.edge_take(right.data_ptr(), rd.data_ptr(), r0, rows, outs[c].data_ptr(), s_out.cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            t = st.final()
            out.setdefault("triple", t)
            assert out["triple"] == t, (mode, t, out["triple"])
        out[mode + "_ms"] = min(ts[1:])
    assert torch.equal(torch.cat([o.view(3, -1) for o in outs], dim=1).reshape(-1), right)
    st.close()
    psa_amd.stream_destroy(f)
    out["flagged_over_one_launch"] = out["flagged_ms"] / out["one_launch_ms"]
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
    from importlib import import_module
    auto = import_module("parallel-sequence-alignment_amd.distributed").auto_chunk_rows
    # (4096-row chunks of a 2^17-column stripe are 16 workgroups each and two are in flight: the chunk size has to grow
    # with the width, which is what distributed.auto_chunk_rows does -- third case)
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":   # chunk rows x streams on the wide stripe (at most 240 workgroups in flight)
        for rows, ns in ((13312, 4), (14336, 4), (15360, 4), (20480, 3), (19456, 3)):
            c = chunked(18, 1 << 17, rows, reps=2, nstreams=ns)
            print(rows, ns, "%.2f ms, %.3f x one launch (%.2f ms)" % (c["chunks_alternating_streams_ms"], c["chunked_over_unchunked"], c["one_launch_ms"]), flush=True)
        sys.exit(0)
    res = {"chunked": [chunked(16, 1 << 13, 4096), chunked(18, 1 << 17, 4096), chunked(18, 1 << 17, auto(1 << 17))],
           "flagged": [flagged(16, 1 << 13, 4096), flagged(18, 1 << 17, 4096), flagged(18, 1 << 17, 1024)],
           "psa_score_stripes_2p16": stripes_p()}
    print(json.dumps(res, indent=1))
