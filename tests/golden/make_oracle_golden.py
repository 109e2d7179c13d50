#!/usr/bin/env python3
"""Generate the RESTATEMENT-ONLY goldens: results for inputs the real reference cannot run (it keeps an
alignment vector in every cell: about 48 n m bytes and n^3 time), produced by this repository's
own CPU restatement (oracle/psa_oracle.c), which tests/test_oracle.py pins against the reference on
everything the reference can run.

Outputs oracle_only.json (committed, data only), every entry marked "produced_by": "restatement":
  titin pairs           score, alignment length and md5 of the full five-line stdout
  synthetic 2^k x 2^k   score of the mt19937_64 pairs of SURVEY 8d for k in --log2 (default 16 18 20),
                        score-only, the oracle's striped multi-thread variant
  bench_pairs           the 16 DISTINCT 2^16 x 2^16 pairs bench.py and the 16-slot engine test give the slots of a
                        launch (slot s: seeds 0xA0000000+16+0x100*s / 0xB0000000+16+0x100*s; slot 0 is the pair
                        above), and two slots of other sizes (50 000 x 65 536 = prefix of A of slot 1; 65 536 x
                        40 000 = prefix of B of slot 2): score, alignment length and md5 of the columns
                        (int32 ai bytes + int32 bj bytes), full alignment by the oracle (--bench-pairs N)

Usage: python tests/golden/make_oracle_golden.py [--log2 16 18 20] [--threads 8]
       (2^20 x 2^20 is 1.1e12 cells: tens of minutes on 8 cores)
"""
import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from make_golden_shim import synth_protein  # noqa: E402  (the generator's own std::mt19937_64, pure Python)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, nargs="*", default=[16, 18, 20])
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--bench-pairs", type=int, default=0, help="slots of the 2^16 bench / engine test to (re)generate")
    ap.add_argument("--out", default=os.path.join(HERE, "oracle_only.json"))
    args = ap.parse_args()
    out = {"produced_by": "restatement (oracle/psa_oracle.c), NOT the reference binary", "titin": [], "synthetic": []}
    if os.path.exists(args.out):
        out = json.load(open(args.out))
    seqdir = os.path.join(HERE, "sequences")
    done = {(e["a"], e["b"]) for e in out["titin"]}
    for fa, fb in (("titin_homo(50).fasta", "titin_mus.fasta"), ("titin_homo(50).fasta", "titin_rattus(90).fasta"),
                   ("titin_mus.fasta", "titin_rattus(90).fasta")):
        if (fa, fb) in done:
            continue
        a, b = O.read_fasta(os.path.join(seqdir, fa)), O.read_fasta(os.path.join(seqdir, fb))
        t0 = time.time()
        text = O.stdout_for(a, b, args.threads)
        sc = int(text.split(b"\n")[0].split(b": ")[1])
        ln = int(text.split(b"\n")[1].split(b": ")[1])
        out["titin"].append({"a": fa, "b": fb, "n": len(a), "m": len(b), "score": sc, "alignment_len": ln,
                             "stdout_md5": hashlib.md5(text).hexdigest(), "produced_by": "restatement",
                             "seconds": round(time.time() - t0, 1)})
        print(out["titin"][-1], flush=True)
        json.dump(out, open(args.out, "w"), indent=1)
    have = {e["log2"] for e in out["synthetic"]}
    for k in args.log2:
        if k in have:
            continue
        n = 1 << k
        a = np.frombuffer(synth_protein(0xA0000000 + k, n).encode(), np.uint8)
        b = np.frombuffer(synth_protein(0xB0000000 + k, n).encode(), np.uint8)
        t0 = time.time()
        sc = O.score(a, b, args.threads)
        out["synthetic"].append({"log2": k, "n": n, "m": n, "seed_a": "0xA0000000+%d" % k, "seed_b": "0xB0000000+%d" % k,
                                 "score": int(sc), "produced_by": "restatement", "threads": args.threads,
                                 "seconds": round(time.time() - t0, 1)})
        print(out["synthetic"][-1], flush=True)
        json.dump(out, open(args.out, "w"), indent=1)
    k = 16
    n = 1 << k
    out.setdefault("bench_pairs", [])
    have = {(e["slot"], e["n"], e["m"]) for e in out["bench_pairs"]}
    todo = [(s, n, n) for s in range(args.bench_pairs)]
    if args.bench_pairs:
        todo += [(1, 50000, n), (2, n, 40000)]
    for s, nn, mm in todo:
        if (s, nn, mm) in have:
            continue
        sa, sb = 0xA0000000 + k + 0x100 * s, 0xB0000000 + k + 0x100 * s
        a = np.frombuffer(synth_protein(sa, n).encode(), np.uint8)[:nn]
        b = np.frombuffer(synth_protein(sb, n).encode(), np.uint8)[:mm]
        t0 = time.time()
        sc, ai, bj = O.align(a, b, args.threads)
        out["bench_pairs"].append({"slot": s, "n": nn, "m": mm, "seed_a": "0x%X" % sa, "seed_b": "0x%X" % sb, "score": int(sc),
                                   "alignment_len": int(len(ai)),
                                   "columns_md5": hashlib.md5(ai.astype(np.int32).tobytes() + bj.astype(np.int32).tobytes()).hexdigest(),
                                   "produced_by": "restatement", "seconds": round(time.time() - t0, 1)})
        print(out["bench_pairs"][-1], flush=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
