"""CPU suite: the oracle (oracle/psa_oracle.c) against the goldens produced by the real reference.

These pin the restatement: every fixture in tests/golden/ref_*.json is stdout of the compiled
reference (tests/golden/make_golden.py).  No GPU needed.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O


def test_fuzz_goldens_single_thread():
    cases = O.load_golden("ref_fuzz.json")
    assert len(cases) == 300
    for c in cases:
        assert O.stdout_for(c["a_seq"], c["b_seq"]) == O.b64d(c["stdout_b64"])


def test_fuzz_goldens_striped_threads():
    # the reference's own decomposition (p stripes); output must not depend on p
    cases = O.load_golden("ref_fuzz.json")
    for k, c in enumerate(cases[::3]):
        p = [1, 2, 3, 5, 8, 13][k % 6]
        assert O.stdout_for(c["a_seq"], c["b_seq"], p) == O.b64d(c["stdout_b64"])


def test_fasta_pair_goldens(golden_dir):
    for c in O.load_golden("ref_fasta_pairs.json"):
        a = O.read_fasta(os.path.join(golden_dir, "sequences", c["a"]))
        b = O.read_fasta(os.path.join(golden_dir, "sequences", c["b"]))
        assert O.stdout_for(a, b, 4) == O.b64d(c["stdout_b64"]), (c["a"], c["b"])


def test_random_protein_goldens(golden_dir):
    for c in O.load_golden("ref_random.json"):
        a = O.read_fasta(os.path.join(golden_dir, "sequences", c["a"]))
        b = O.read_fasta(os.path.join(golden_dir, "sequences", c["b"]))
        assert O.stdout_for(a, b) == O.b64d(c["stdout_b64"])
        assert O.stdout_for(a, b, 8) == O.b64d(c["stdout_b64"])


def test_format_cases(tmp_path):
    import base64
    for c in O.load_golden("ref_format_cases.json"):
        if "argv" in c:
            continue
        seqs = []
        for key, name in (("a_file_b64", c["a"]), ("b_file_b64", c["b"])):
            path = tmp_path / name
            if c[key] is None:
                if path.exists():
                    path.unlink()
            else:
                path.write_bytes(base64.b64decode(c[key]))
            seqs.append(O.read_fasta(str(path)))
        assert O.stdout_for(seqs[0], seqs[1]) == O.b64d(c["stdout_b64"]), (c["a"], c["b"])
        assert c["rc"] == 0


def test_known_survey_values(golden_dir):
    # SURVEY 8c table
    rd = lambda n: O.read_fasta(os.path.join(golden_dir, "sequences", n))
    out = O.stdout_for(rd("insulin_homo.fasta"), rd("insulin_bovin.fasta"))
    assert hashlib.md5(out).hexdigest() == "7bc3956188dcaa02e82f5213b1ed5e8c"
    assert out.startswith(b"Score: 169\nLength Sequence A : 110\n")
    out = O.stdout_for(rd("Q16739_homo.fasta"), rd("A0A2I3N2W7_baboon.fasta"))
    assert hashlib.md5(out).hexdigest() == "c8a552d9d71d534f4ea143a53413fa90"


def test_score_only_matches_align():
    rng = np.random.default_rng(7)
    for _ in range(50):
        n, m = rng.integers(0, 90, 2)
        a = rng.integers(65, 65 + 3, n).astype(np.uint8)
        b = rng.integers(65, 65 + 3, m).astype(np.uint8)
        s, _, _ = O.align(a, b)
        assert O.score(a, b) == s
        assert O.score(a, b, 3) == s


def test_block_composes():
    # filling the matrix as a 2x2 grid of blocks with explicit boundaries equals the whole
    rng = np.random.default_rng(11)
    n, m = 150, 170
    a = rng.integers(65, 69, n).astype(np.uint8)
    b = rng.integers(65, 69, m).astype(np.uint8)
    top = np.array([[O.boundary(0, j)[k] for j in range(m + 1)] for k in range(3)], np.int32)
    left = np.array([[O.boundary(i, 0)[k] for i in range(n + 1)] for k in range(3)], np.int32)
    bot_full, right_full = O.block(a, b, 0, n, 0, m, top, left)
    r, c = 64, 100
    b00, r00 = O.block(a, b, 0, r, 0, c, top[:, :c + 1], left[:, :r + 1])
    b01, r01 = O.block(a, b, 0, r, c, m, top[:, c:], r00)
    b10, r10 = O.block(a, b, r, n, 0, c, b00, left[:, r:])
    b11, r11 = O.block(a, b, r, n, c, m, b01, r10)
    assert np.array_equal(b11, bot_full[:, c:])
    assert np.array_equal(r11, right_full[:, r:])
    t = bot_full[:, m]
    assert O.score(a, b) == max(t)


@pytest.mark.slow
def test_titin_restatement_md5(golden_dir):
    # restatement-only golden (the reference cannot run this pair: ~58 GB, ~19 h); SURVEY 8c
    rd = lambda n: O.read_fasta(os.path.join(golden_dir, "sequences", n))
    out = O.stdout_for(rd("titin_homo(50).fasta"), rd("titin_mus.fasta"), 8)
    assert hashlib.md5(out).hexdigest() == "65af487c90fa2638ce05f3a6badbaaea"
    assert out.startswith(b"Score: 62103\nLength Sequence A : 35289\n")


def test_neighbour_bounds_behind_the_one_byte_codes():
    """The HIP traceback decodes a cell from the low four bits of D = max(T1,T2,T3) of the cell and of its three
    neighbours (psa_device.hip.h, "winner codes").  That needs, for every interior cell, D - T_k in 0..15 for the
    three values the reference compared there; the recurrence gives 0..8 (horizontal and vertical neighbours of
    D differ by -3..+5, diagonal ones by 0..2 -- the bounds of the Gotoh recurrence under the reference's own
    boundary values).  Checked here on the oracle's tables, row by row through its block function, over small
    alphabets (ties), mutated copies with long gaps and every boundary."""
    rng = np.random.default_rng(2024)
    worst_e, worst_h, worst_v, worst_d = 0, [0, 0], [0, 0], [9, 0]
    for it in range(120):
        n, m = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        alpha = int(rng.integers(1, 6)) if it % 5 else 20
        a = rng.integers(65, 65 + alpha, n).astype(np.uint8)
        if it % 3 == 0:
            b = rng.integers(65, 65 + alpha, m).astype(np.uint8)
        else:   # a copy with substitutions and long gaps
            src = np.concatenate([a, a, a])
            cut = int(rng.integers(0, n))
            b = np.concatenate([src[:cut], src[cut + int(rng.integers(0, 40)):]])[:m]
            if b.size == 0:
                b = a[:1]
            b = np.where(rng.random(b.size) < 0.1, rng.integers(65, 65 + alpha, b.size), b).astype(np.uint8)
            m = int(b.size)
        rows = [np.array([O.boundary(0, j) for j in range(m + 1)], np.int32).T]
        left = np.array([O.boundary(i, 0) for i in range(n + 1)], np.int32).T
        for i in range(n):   # one row at a time: rows[i + 1] = T1,T2,T3 of row i + 1
            bottom, _ = O.block(a, b, i, i + 1, 0, m, rows[-1], left[:, i:i + 2])
            rows.append(bottom)
        T = np.stack(rows, axis=1).astype(np.int64)          # [3][n + 1][m + 1]
        D = T.max(axis=0)
        inner = D[1:, 1:]
        worst_e = max(worst_e, int((inner[None] - T[:, 1:, 1:]).max()))
        dh, dv, dd = inner - D[1:, :-1], inner - D[:-1, 1:], inner - D[:-1, :-1]
        worst_h = [min(worst_h[0], int(dh.min())), max(worst_h[1], int(dh.max()))]
        worst_v = [min(worst_v[0], int(dv.min())), max(worst_v[1], int(dv.max()))]
        worst_d = [min(worst_d[0], int(dd.min())), max(worst_d[1], int(dd.max()))]
    assert worst_e <= 8, worst_e
    assert worst_h[0] >= -3 and worst_h[1] <= 5, worst_h
    assert worst_v[0] >= -3 and worst_v[1] <= 5, worst_v
    assert worst_d[0] >= 0 and worst_d[1] <= 2, worst_d


def test_restatement_only_goldens_are_what_the_generator_writes(golden_dir):
    """tests/golden/oracle_only.json (made by make_oracle_golden.py, restatement-only): well-formed, marked, its titin
    entry is the md5 the GPU test uses, and the entries a CPU run can afford are RE-DERIVED here with the oracle: the
    score of the synthetic 2^16 pair (striped oracle, seconds) and score, length and column md5 of one bench slot of
    another size.  The 2^18 / 2^20 scores are not recomputed (minutes to a quarter of an hour): the GPU tests that
    consume them are 'parity unpinned' beyond the p-independence and banded cross-checks they make themselves."""
    import json
    g = json.load(open(os.path.join(golden_dir, "oracle_only.json")))
    assert "restatement" in g["produced_by"]
    t = [e for e in g["titin"] if e["b"] == "titin_mus.fasta" and e["a"].startswith("titin_homo")][0]
    assert (t["score"], t["alignment_len"], t["stdout_md5"]) == (62103, 35289, "65af487c90fa2638ce05f3a6badbaaea")
    from make_golden_shim import synth_protein
    threads = min(8, os.cpu_count() or 1)
    e16 = [e for e in g["synthetic"] if e["log2"] == 16][0]
    a = np.frombuffer(synth_protein(0xA0000000 + 16, 1 << 16).encode(), np.uint8)
    b = np.frombuffer(synth_protein(0xB0000000 + 16, 1 << 16).encode(), np.uint8)
    assert O.score(a, b, threads) == e16["score"] == 12738
    slot0 = [e for e in g["bench_pairs"] if (e["slot"], e["n"], e["m"]) == (0, 1 << 16, 1 << 16)][0]
    assert slot0["score"] == e16["score"]
    e = [e for e in g["bench_pairs"] if (e["slot"], e["n"], e["m"]) == (2, 1 << 16, 40000)][0]
    a = np.frombuffer(synth_protein(0xA0000000 + 16 + 0x200, 1 << 16).encode(), np.uint8)
    b = np.frombuffer(synth_protein(0xB0000000 + 16 + 0x200, 1 << 16).encode(), np.uint8)[:40000]
    sc, ai, bj = O.align(a, b, threads)
    assert (int(sc), len(ai), hashlib.md5(ai.astype(np.int32).tobytes() + bj.astype(np.int32).tobytes()).hexdigest()) == \
        (e["score"], e["alignment_len"], e["columns_md5"])
    k = 12   # the striped oracle against the single-thread one
    a = synth_protein(0xA0000000 + k, 1 << k).encode()
    b = synth_protein(0xB0000000 + k, 1 << k).encode()
    assert O.score(a, b, 4) == O.score(a, b)
