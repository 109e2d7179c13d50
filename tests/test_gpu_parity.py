"""GPU suite: the HIP path (through the C ABI) against the oracle and the reference goldens.
Bar: bit-exact scores, alignment columns and stdout bytes."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import psa_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_stdout(a, b, p=1):
    res = psa_amd.align(a, b, p)
    return psa_amd.format_result(res, a, b)


def test_wave_shift_smoke():
    # smallest end-to-end case first: a fault here names the kernel, not a big input
    res = psa_amd.align(b"H", b"LC", 1)
    sc, ai, bj = O.align(b"H", b"LC")
    assert res.score == sc and list(res.ai) == list(ai) and list(res.bj) == list(bj)


def test_fuzz_goldens():
    cases = O.load_golden("ref_fuzz.json")
    for k, c in enumerate(cases):
        p = c["p"] if k % 2 else 1
        assert gpu_stdout(c["a_seq"], c["b_seq"], p) == O.b64d(c["stdout_b64"]), k


def test_fasta_pair_goldens(golden_dir):
    for c in O.load_golden("ref_fasta_pairs.json"):
        a = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", c["a"]))
        b = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", c["b"]))
        assert gpu_stdout(a, b, c["p"]) == O.b64d(c["stdout_b64"]), (c["a"], c["b"])


def test_random_protein_goldens(golden_dir):
    for c in O.load_golden("ref_random.json"):
        a = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", c["a"]))
        b = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", c["b"]))
        for p in (1, 3):
            assert gpu_stdout(a, b, p) == O.b64d(c["stdout_b64"])


def test_degenerate_inputs():
    for a, b in ((b"", b""), (b"", b"ACD"), (b"ACD", b""), (b"A", b""), (b"A", b"A"), (b"A", b"C")):
        for p in (1, 4):
            res = psa_amd.align(a, b, p)
            sc, ai, bj = O.align(a, b)
            assert (res.score, list(res.ai), list(res.bj)) == (sc, list(ai), list(bj)), (a, b, p)
            assert psa_amd.score(a, b, p) == sc


@pytest.mark.parametrize("alpha", [2, 4, 20])
def test_random_vs_oracle_tile_edges(alpha):
    rng = np.random.default_rng(100 + alpha)
    sizes = [(1, 1), (1, 200), (200, 1), (63, 63), (64, 64), (65, 65), (64, 1), (65, 127), (128, 129),
             (191, 65), (257, 300), (300, 257), (511, 513), (1000, 37), (37, 1000), (777, 1234)]
    for n, m in sizes:
        a = rng.integers(65, 65 + alpha, n).astype(np.uint8)
        b = rng.integers(65, 65 + alpha, m).astype(np.uint8)
        sc, ai, bj = O.align(a, b)
        for p in (1, 2, 5):
            res = psa_amd.align(a, b, p)
            assert res.score == sc, (n, m, p)
            assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (n, m, p)
        assert psa_amd.score(a, b, 1) == sc


def test_right_boundary_second_pass_widths():
    """Stripes with a right boundary take their values at column W-1 in a second pass over their last one or two
    superblocks (fill_strips_v2_edges, run_capture_pass): stripe widths around the multiples of 64 from 128 up (one or
    two superblocks to repeat, the capturing superblock the head's neighbour), row counts that leave the last strip
    partly empty, with and without winner codes; every stripe count against the oracle (simple.cpp:136-154, 215-235)."""
    rng = np.random.default_rng(777)
    for alpha in (4, 20):
        for W in (128, 129, 191, 192, 193, 255, 256, 257, 320, 383, 385):
            for p, n in ((2, 130), (3, 64), (2, 449)):
                m = W * p + int(rng.integers(0, p))          # (the last stripe takes the remainder, simple.cpp:92-99)
                a = rng.integers(65, 65 + alpha, n).astype(np.uint8)
                b = rng.integers(65, 65 + alpha, m).astype(np.uint8)
                sc, ai, bj = O.align(a, b)
                sc_p, triple_p = psa_amd.score(a, b, p, triple=True)
                sc_1, triple_1 = psa_amd.score(a, b, 1, triple=True)
                assert sc_p == sc and triple_p == triple_1, (alpha, W, p, n)
                res = psa_amd.align(a, b, p)
                assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (alpha, W, p, n)


def test_block_indel_walks():
    """Conserved blocks separated by alternating long insertions in A and in B: the walk alternates diagonal
    runs with horizontal and vertical runs of 100-170 cells, which moves it through the traceback's tile
    window in every direction (tiles revisited through another slot, fetch cursor off the diagonal,
    tiles beside corners)."""
    rng = np.random.default_rng(4242)
    for blocks, lo, hi in ((24, 5, 90), (40, 30, 64), (12, 200, 400)):
        xa, xb = [], []
        for t in range(blocks):
            blk = rng.integers(65, 85, int(rng.integers(lo, hi + 1))).astype(np.uint8)
            xa.append(blk); xb.append(blk.copy())
            ins = rng.integers(65, 85, int(rng.integers(100, 171))).astype(np.uint8)
            (xa if t % 2 == 0 else xb).append(ins)
        a = np.concatenate(xa); b = np.concatenate(xb)
        # a few point mutations inside the blocks
        for v in (a, b):
            idx = rng.integers(0, len(v), len(v) // 50)
            v[idx] = rng.integers(65, 85, len(idx)).astype(np.uint8)
        sc, ai, bj = O.align(a, b)
        for pp in (1, 3):
            res = psa_amd.align(a, b, pp)
            assert res.score == sc, (blocks, pp)
            assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (blocks, pp)
        res = psa_amd.align(a, b, 1, band_rows=512)
        assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), blocks


def test_row_chunks_and_stripes(monkeypatch):
    # several row chunks per stripe (chunk = 128 rows) and several stripes
    monkeypatch.setenv("PSA_CHUNK_ROWS", "128")
    rng = np.random.default_rng(5)
    for n, m, p in ((700, 650, 3), (129, 900, 7), (1024, 64, 2), (1500, 1500, 4)):
        a = rng.integers(65, 68, n).astype(np.uint8)
        b = np.where(rng.random(m) < 0.7, np.resize(a, m), rng.integers(65, 68, m)).astype(np.uint8)
        sc, ai, bj = O.align(a, b)
        res = psa_amd.align(a, b, p)
        assert res.score == sc
        assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (n, m, p)
        assert psa_amd.score(a, b, p) == sc


def test_titin_full_traceback(golden_dir):
    # BASELINE config 3 on one GPU; golden is restatement-only (reference cannot run it), SURVEY 8c
    a = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_homo(50).fasta"))
    b = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_mus.fasta"))
    out = gpu_stdout(a, b, 1)
    assert out.startswith(b"Score: 62103\nLength Sequence A : 35289\n")
    assert hashlib.md5(out).hexdigest() == "65af487c90fa2638ce05f3a6badbaaea"


def test_large_score_only_property():
    # 2^14 random pair: score equals the oracle's rolling-row score (seconds on CPU);
    # split-independence: p = 1 and p = 4 agree
    a = psa_amd.synth_protein(0xA0000000 + 14, 1 << 14)
    b = psa_amd.synth_protein(0xB0000000 + 14, 1 << 14)
    want = O.score(a, b, 8)
    assert psa_amd.score(a, b, 1) == want
    assert psa_amd.score(a, b, 4) == want


def test_huge_pair_score_only_and_linear_space():
    """BASELINE config 4's second size, 2^20 x 2^20 (1.1e12 cells): the score-only fill with p = 1 and p = 3, the
    committed restatement-only golden score (tests/golden/oracle_only.json, made by make_oracle_golden.py) and the
    score of the linear-space alignment all agree; the alignment's columns obey the size-independent properties."""
    import json
    k = 20
    gold = [e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_only.json")))["synthetic"] if e["log2"] == k]
    a = psa_amd.synth_protein(0xA0000000 + k, 1 << k)
    b = psa_amd.synth_protein(0xB0000000 + k, 1 << k)
    s1 = psa_amd.score(a, b, 1)
    assert psa_amd.score(a, b, 3) == s1
    if gold:
        assert s1 == gold[0]["score"]
    psa_amd.release_cache()
    res = psa_amd.align(a, b, 1, band_rows=0)
    assert res.score == s1
    for x in (res.ai, res.bj):
        assert np.all(np.diff(x[x >= 0]) >= 0) and x[x >= 0][-1] == (1 << k) - 1
    assert not np.any((res.ai < 0) & (res.bj < 0))


def test_full_size_pair_vs_oracle():
    # BASELINE config 4 at the size the metric is quoted on (2^16 x 2^16, 4.3 G cells): complete alignment,
    # every column equal to the oracle's (threads of the oracle: its striped variant, a few seconds)
    k = 16
    a = psa_amd.synth_protein(0xA0000000 + k, 1 << k)
    b = psa_amd.synth_protein(0xB0000000 + k, 1 << k)
    sc, ai, bj = O.align(a, b, min(16, os.cpu_count() or 1))
    res = psa_amd.align(a, b, 1)
    assert res.score == sc == 12738
    assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj)
    # size-independent properties: indices never go back, no column is gap/gap (the reference's prefix
    # fix-up, simple.cpp:279-290, may repeat the first aligned residue, so "each residue once" does not hold)
    for x in (res.ai, res.bj):
        assert np.all(np.diff(x[x >= 0]) >= 0) and x[x >= 0][-1] == (1 << k) - 1
    assert not np.any((res.ai < 0) & (res.bj < 0))


def _bench_goldens():
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_only.json")))
    return {(e["slot"], e["n"], e["m"]): e for e in g["bench_pairs"]}


def _bench_pair(slot, n=1 << 16, m=1 << 16):
    k = 16
    a = psa_amd.synth_protein(0xA0000000 + k + 0x100 * slot, 1 << k)[:n]
    b = psa_amd.synth_protein(0xB0000000 + k + 0x100 * slot, 1 << k)[:m]
    return a, b


def _columns_md5(res):
    return hashlib.md5(np.ascontiguousarray(res.ai, np.int32).tobytes() + np.ascontiguousarray(res.bj, np.int32).tobytes()).hexdigest()


def test_dense_launch_at_bench_shape_distinct_pairs(golden_dir):
    """The kernel bench.py times -- fill_strips_v2_dense: ONE launch over 20 pairs of 2^16 x 2^16 (all an argument block holds), 1 024 strips each, two
    workgroups per CU, ring wrap-around -- at the bench's own shape with a DIFFERENT pair in every slot, two slots of
    other sizes (titin; 50 000 x 65 536 / 65 536 x 40 000), two launches back to back on two engines as bench.py
    does.  Every slot: score, alignment length and the md5 of all its columns against the committed restatement-only
    goldens (tests/golden/oracle_only.json, make_oracle_golden.py --bench-pairs: 'parity unpinned' beyond what
    test_oracle.py pins the restatement to); four slots column for column against the oracle run here."""
    gold = _bench_goldens()
    n = m = 1 << 16
    ta = np.frombuffer(psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_homo(50).fasta")), np.uint8)
    tb = np.frombuffer(psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_mus.fasta")), np.uint8)
    S = 20
    engines = [psa_amd.Many(S, n, m, traceback=True) for _ in range(2)]
    layouts = []
    for x in range(2):
        lay = []
        for sl in range(S):
            if sl == S - 2:
                lay.append(("titin", ta, tb))
            elif sl == S - 1:
                key = (1, 50000, m) if x == 0 else (2, n, 40000)
                lay.append((key,) + _bench_pair(*key))
            else:
                slot = (sl + 7 * x) % (S - 2)       # (the second engine holds the pairs in other slots)
                lay.append(((slot, n, m),) + _bench_pair(slot))
        layouts.append(lay)
    for x, eng in enumerate(engines):
        for sl, (_, a, b) in enumerate(layouts[x]):
            eng.set_pair(sl, a, b)
        eng.launch(engines[x ^ 1])
    live = {}
    for x, eng in enumerate(engines):
        eng.wait()
        for sl, (key, a, b) in enumerate(layouts[x]):
            res = eng.result(sl, a.size, b.size)
            if key == "titin":
                out = psa_amd.format_result(res, a, b)
                assert hashlib.md5(out).hexdigest() == "65af487c90fa2638ce05f3a6badbaaea", (x, sl)
                continue
            g = gold[key]
            assert (res.score, len(res.ai), _columns_md5(res)) == (g["score"], g["alignment_len"], g["columns_md5"]), (x, sl, key)
            if (x, sl) in ((0, 0), (0, S - 1), (1, 3), (1, S - 1)):
                live[(x, sl)] = (a, b, res)
    for e in engines:
        e.close()
    for (x, sl), (a, b, res) in live.items():
        sc, ai, bj = O.align(a, b, min(16, os.cpu_count() or 1))
        assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (x, sl)


def test_packed_fill_couples_match_oracle(monkeypatch):
    """fill_strips_pk (psa_fill_pk.hip.h): the engine couples the pairs of a launch that have the same size and fills each
    couple with two pairs per lane, 16-bit halves relative to frames that move with the values.  Cases that move the
    frames hard or not at all -- identical sequences (the score grows by 2 per column), one residue repeated against
    itself and against another, unrelated sequences, a shifted copy --, sizes around the limits of the packed path (256
    columns; rows that are no multiple of 64), launches with a pair left over and with couples of different sizes: every
    column against the oracle, and the same launches again through the int32 kernel (PSA_FILL_PK=0)."""
    al = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    rng = np.random.default_rng(9)
    same = al[rng.integers(0, 20, 12000)]
    other = al[rng.integers(0, 20, 12000)]
    poly, polyc = np.full(12000, ord("A"), np.uint8), np.full(12000, ord("C"), np.uint8)
    shifted = np.concatenate([other[:2000], same[:10000]])

    def rnd(n, m, related):
        a = al[rng.integers(0, 20, n)]
        if not related:
            return a, al[rng.integers(0, 20, m)]
        b = a[:m].copy() if m <= n else np.concatenate([a, al[rng.integers(0, 20, m - n)]])
        mut = rng.random(m) < 0.2
        b[mut] = al[rng.integers(0, 20, int(mut.sum()))]
        return a, b

    launches = [
        [(same, same.copy()), (same, other), (poly, poly.copy()), (poly, polyc)],
        [(same, shifted), (poly, same)],
        [rnd(300, 256, False), rnd(300, 256, True)],
        [rnd(1000, 700, False), rnd(1000, 700, True), rnd(640, 700, False)],
        [rnd(5000, 3000, False), rnd(5000, 3000, True), rnd(333, 3000, False), rnd(333, 3000, True), rnd(5000, 3000, True)],
        [rnd(255, 255, False), rnd(255, 255, True)],          # below the packed path's 256 columns: the int32 kernel
    ]
    want = [[O.align(a, b, 4) for a, b in pairs] for pairs in launches]
    for mode in ("1", "0"):
        monkeypatch.setenv("PSA_FILL_PK", mode)
        for pairs, ref in zip(launches, want):
            eng = psa_amd.Many(len(pairs), max(a.size for a, _ in pairs), max(b.size for _, b in pairs), traceback=True)
            for sl, (a, b) in enumerate(pairs):
                eng.set_pair(sl, a, b)
            eng.launch()
            eng.wait()
            for sl, (a, b) in enumerate(pairs):
                r = eng.result(sl, a.size, b.size)
                sc, ai, bj = ref[sl]
                assert r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj), (mode, a.size, b.size, sl)
            eng.close()
            # score only (the same kernel without the code bytes)
            eng = psa_amd.Many(len(pairs), max(a.size for a, _ in pairs), max(b.size for _, b in pairs), traceback=False)
            for sl, (a, b) in enumerate(pairs):
                eng.set_pair(sl, a, b)
            eng.launch()
            eng.wait()
            for sl, (a, b) in enumerate(pairs):
                assert eng.result(sl, a.size, b.size, want_alignment=False).score == ref[sl][0], (mode, "score only", sl)
            eng.close()


def test_packed_fill_sizes_around_strip_and_superblock_boundaries():
    """The packed fill at row counts around multiples of 64 (strips: lanes without a row, a last strip of one row) and
    column counts around multiples of 64 (superblocks: where the frames move and the capturing tail begins), four pairs per
    launch (unrelated, periodic copies of A, a mutated one): every column against the oracle."""
    rng = np.random.default_rng(5)
    al = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    shapes = [(64, 256), (65, 257), (127, 300), (128, 319), (129, 320), (191, 321), (192, 383), (193, 384), (255, 385),
              (256, 447), (257, 448), (320, 449), (511, 511), (512, 512), (513, 513), (64, 1025), (1000, 256), (2049, 2047)]
    for n, m in shapes:
        pairs = []
        for j in range(4):
            a = al[rng.integers(0, 20, n)]
            b = al[rng.integers(0, 20, m)] if j % 2 == 0 else np.resize(a, m).copy()
            if j == 3:
                mut = rng.random(m) < 0.1
                b[mut] = al[rng.integers(0, 20, int(mut.sum()))]
            pairs.append((a, b))
        eng = psa_amd.Many(4, n, m, traceback=True)
        for sl, (a, b) in enumerate(pairs):
            eng.set_pair(sl, a, b)
        eng.launch()
        eng.wait()
        for sl, (a, b) in enumerate(pairs):
            r = eng.result(sl, n, m)
            sc, ai, bj = O.align(a, b, 2)
            assert r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj), (n, m, sl)
        eng.close()


def test_many_short_launches_on_several_engines_do_not_stall():
    """Five engines, short launches (2 048 x 2 048 pairs) one behind the other: with the walks queued at the launch behind a
    wait for the fill's event on a HIGH-priority stream, the queue scheduler left the fills undispatched for seconds
    (waits of 1-58 s per launch, no error); the walks are now queued by psa_many_wait once the fill has ended.  Bounded
    time for 30 launches, results against the oracle."""
    import time
    n = m = 2048
    S, E = 6, 5
    pairs = [(psa_amd.synth_protein(77 + s, n), psa_amd.synth_protein(7700 + s, m)) for s in range(S)]
    want = [O.align(a, b, 2) for a, b in pairs]
    engines = [psa_amd.Many(S, n, m, traceback=True) for _ in range(E)]
    t0 = time.time()
    for it in range(6):
        for e in engines:
            if it > 0:
                e.wait()
                for sl in range(S):
                    r = e.result(sl, n, m)
                    sc, ai, bj = want[sl]
                    assert r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj), (it, sl)
            for sl, (a, b) in enumerate(pairs):
                e.set_pair(sl, a, b)
            e.launch()
    for e in engines:
        e.wait()
        e.close()
    assert time.time() - t0 < 20.0, "launches stalled: %.1f s for 30 launches of six 2 048 x 2 048 pairs" % (time.time() - t0)


def test_column_stripes_at_metric_sizes(golden_dir):
    """p > 1 at the sizes the metric is quoted on (simple.cpp:90-99, 136-154, 215-235: the stripes of one pair, here
    jobs of ONE launch with flags between them): the titin pair with p = 8 against the committed md5 of its stdout,
    the 2^16 x 2^16 pair with p = 4 and p = 8 column for column against p = 1 and the committed golden."""
    a = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_homo(50).fasta"))
    b = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "titin_mus.fasta"))
    assert hashlib.md5(gpu_stdout(a, b, 8)).hexdigest() == "65af487c90fa2638ce05f3a6badbaaea"
    g = _bench_goldens()[(0, 1 << 16, 1 << 16)]
    a, b = _bench_pair(0)
    ref = psa_amd.align(a, b, 1)
    assert (ref.score, len(ref.ai), _columns_md5(ref)) == (g["score"], g["alignment_len"], g["columns_md5"])
    for p in (4, 8):
        res = psa_amd.align(a, b, p)
        assert res.score == ref.score and np.array_equal(res.ai, ref.ai) and np.array_equal(res.bj, ref.bj), p
        assert psa_amd.score(a, b, p) == ref.score
    psa_amd.release_cache()


def test_cli_byte_exact(golden_dir):
    exe = os.path.join(ROOT, "simple")
    assert os.path.exists(exe), "run `make` first"
    for c in O.load_golden("ref_fasta_pairs.json")[:6]:
        r = subprocess.run([exe, c["a"], c["b"], str(c["p"])], cwd=golden_dir, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout == O.b64d(c["stdout_b64"])
    r = subprocess.run([exe, "nope.fasta", "insulin_homo.fasta", "2"], cwd=golden_dir, capture_output=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith(b"Score: 0\nLength Sequence A : 0\n")
    assert b"Error opening file: sequences/nope.fasta" in r.stderr
    r = subprocess.run([exe, "only_one_arg"], cwd=golden_dir, capture_output=True, timeout=120)
    assert r.returncode == 1 and b"Usage:" in r.stderr and r.stdout == b""


def test_cli_reference_format_cases(tmp_path):
    """The reference's behaviour on FASTA-format and degenerate inputs (CRLF, multi-record, lower case, no header,
    empty / header-only / missing files, p > m, p = 32, usage error), as the real reference produced it
    (ref_format_cases.json), through the GPU ./simple: stdout, exit status and stderr."""
    import base64
    exe = os.path.join(ROOT, "simple")
    seqdir = tmp_path / "sequences"
    seqdir.mkdir()
    cases = O.load_golden("ref_format_cases.json")
    for c in cases:
        for key, name in (("a_file_b64", c.get("a")), ("b_file_b64", c.get("b"))):
            if name and c.get(key) is not None:
                (seqdir / name).write_bytes(base64.b64decode(c[key]))
    for c in cases:
        argv = c["argv"] if "argv" in c else [c["a"], c["b"], str(c["p"])]
        r = subprocess.run([exe] + argv, cwd=tmp_path, capture_output=True, timeout=120)
        assert r.returncode == c["rc"], (argv, r.stderr)
        if "argv" in c:
            assert b"Usage:" in r.stderr and r.stdout == b""
            continue
        assert r.stdout == O.b64d(c["stdout_b64"]), argv
        assert r.stderr == O.b64d(c["stderr_b64"]), argv


def test_cli_timing_sweep(golden_dir, tmp_path):
    """The reference's (commented-out) p = 1..32 sweep, simple.cpp:319-335: same CSV header and rows."""
    exe = os.path.join(ROOT, "simple")
    c = O.load_golden("ref_fasta_pairs.json")[0]
    csv = tmp_path / "timings.csv"
    env = dict(os.environ, PSA_TIMINGS_CSV=str(csv))
    r = subprocess.run([exe, c["a"], c["b"], "1"], cwd=golden_dir, capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    gold = O.b64d(c["stdout_b64"])
    assert r.stdout.startswith(gold)
    score = gold.split(b"\n")[0].split(b": ")[1]
    lines = r.stdout[len(gold):].decode().splitlines()
    assert len(lines) == 33 and all(l.startswith("p=%d, time=" % (k + 1)) and l.endswith("score=" + score.decode())
                                    for k, l in enumerate(lines[:32]))
    rows = csv.read_text().splitlines()
    assert rows[0] == "threads,time_microseconds" and [x.split(",")[0] for x in rows[1:]] == [str(k) for k in range(1, 33)]


def test_batch_scores_match_oracle():
    rng = np.random.default_rng(23)
    seqs = []
    for ln in [0, 1, 2, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 300, 511, 512, 513, 700, 1024, 1100]:
        alpha = [2, 4, 20][ln % 3]
        seqs.append(rng.integers(65, 65 + alpha, ln).astype(np.uint8))
    pairs = [(i, j) for i in range(len(seqs)) for j in range(len(seqs))]
    got = psa_amd.score_batch(seqs, pairs)
    for (i, j), g in zip(pairs, got):
        assert g == O.score(seqs[i], seqs[j]), (len(seqs[i]), len(seqs[j]))
    # many equal-length pairs (the shape of BASELINE config 5)
    prot = [psa_amd.synth_protein(0xC0000000 + k, 512) for k in range(64)]
    pairs = [(i, j) for i in range(64) for j in range(64)]
    got = psa_amd.score_batch(prot, pairs)
    for idx in range(0, len(pairs), 37):
        i, j = pairs[idx]
        assert got[idx] == O.score(prot[i], prot[j])


def test_batch_packed_couples_match_oracle():
    """The batch kernel takes pairs in couples (2c, 2c + 1); a couple whose second sequences have the same length goes
    through the packed 16-bit path (two pairs per lane: psa_batch.hip.h, batch_couple_pk), any other through the int32
    path.  Couples that share a target of every length around the block and strip edges, first sequences of different
    lengths (their last rows in different strips), small alphabets (ties), an odd pair at the end, couples too long for
    16 bits, couples with an empty first sequence: every score against the oracle."""
    rng = np.random.default_rng(2025)
    targets = [1, 2, 15, 16, 17, 63, 64, 65, 95, 96, 97, 111, 112, 113, 127, 128, 129, 200, 511, 512, 513, 1000, 1024]
    seqs, pairs = [], []
    for x, nb in enumerate(targets):
        alpha = [2, 4, 20][x % 3]
        t = len(seqs)
        seqs.append(rng.integers(65, 65 + alpha, nb).astype(np.uint8))
        lens = [1, 64, 65, int(rng.integers(1, 700)), int(rng.integers(1, 700)), 128, 129, 3 * nb + 1]
        for k, na in enumerate(lens):
            q = len(seqs)
            # (some queries are mutated copies of the target: long diagonal runs and real gaps)
            src = np.resize(seqs[t], na) if k % 2 else rng.integers(65, 65 + alpha, na).astype(np.uint8)
            seqs.append(np.where(rng.random(na) < 0.8, src, rng.integers(65, 65 + alpha, na)).astype(np.uint8))
            pairs.append((q, t))
    # equal-length targets that are different sequences, and the same pair twice in one couple
    a512 = [psa_amd.synth_protein(0xC0000000 + k, 512) for k in range(6)]
    base = len(seqs)
    seqs += a512
    pairs += [(base, base + 1), (base + 2, base + 3), (base + 4, base + 4), (base + 4, base + 4), (base + 5, base)]
    # too long for the 16-bit halves (nA + nB > 6000), an empty first sequence in a couple, an odd pair at the end
    big = len(seqs)
    seqs += [rng.integers(65, 69, 5800).astype(np.uint8), rng.integers(65, 69, 600).astype(np.uint8), np.zeros(0, np.uint8)]
    pairs += [(big, big + 1), (big, big + 1), (big + 2, big + 1), (big + 1, big + 1), (big + 1, base)]
    if len(pairs) % 2 == 0:
        pairs.append((base + 1, base + 2))
    assert len(pairs) % 2 == 1
    got = psa_amd.score_batch(seqs, pairs)
    for (i, j), g in zip(pairs, got):
        assert g == O.score(seqs[i], seqs[j]), (len(seqs[i]), len(seqs[j]), int(g))


def test_batch_oversize_second_sequences_among_short_pairs():
    """Pairs whose second sequence exceeds the batch kernel's LDS row buffer (1024) are skipped by the kernel and
    sent through psa_score; the ordinary pairs of the same launch must not be disturbed by them."""
    rng = np.random.default_rng(77)
    seqs = [rng.integers(65, 85, ln).astype(np.uint8) for ln in (1500, 3000, 8000)]
    seqs += [rng.integers(65, 85, int(rng.integers(50, 600))).astype(np.uint8) for _ in range(40)]
    pairs = [(i, j) for i in range(3, 43) for j in range(3, 43, 5)]          # short x short
    pairs += [(i, j) for i in range(3, 43, 3) for j in range(3)]             # short x oversize second sequence
    pairs += [(j, i) for i in range(3, 43, 7) for j in range(3)]             # oversize first sequence
    pairs += [(0, 1), (1, 2), (2, 0)]
    order = rng.permutation(len(pairs))
    pairs = [pairs[k] for k in order]
    got = psa_amd.score_batch(seqs, pairs)
    for (i, j), g in zip(pairs, got):
        assert g == O.score(seqs[i], seqs[j]), (len(seqs[i]), len(seqs[j]))


def test_alignments_in_flight_concurrently():
    """Several whole-pair alignments at once (a host thread each, the library's per-call streams),
    as the bench's throughput mode does: results stay those of the oracle."""
    import threading
    rng = np.random.default_rng(11)
    pairs = []
    for k in range(4):
        n, m = int(rng.integers(3000, 6000)), int(rng.integers(3000, 6000))
        pairs.append((rng.integers(65, 69, n).astype(np.uint8), rng.integers(65, 69, m).astype(np.uint8)))
    want = [O.align(a, b) for a, b in pairs]
    got = [None] * len(pairs)

    def work(i):
        for _ in range(3):
            got[i] = psa_amd.align(pairs[i][0], pairs[i][1], 1)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(pairs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for (sc, ai, bj), res in zip(want, got):
        assert res is not None and res.score == sc
        assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj)


def test_traceback_kernels_agree(monkeypatch):
    """The two-wave walk (default), the single-wave walk and the single-lane walk read the same codes."""
    rng = np.random.default_rng(12)
    a = rng.integers(65, 69, 2500).astype(np.uint8)
    b = rng.integers(65, 69, 2700).astype(np.uint8)
    r1 = psa_amd.align(a, b, 1)
    monkeypatch.setenv("PSA_TRACEBACK_KERNEL", "wave1")    # one wavefront that walks and fetches
    rw = psa_amd.align(a, b, 1)
    rw3 = psa_amd.align(a, b, 3)
    rwb = psa_amd.align(a, b, 1, band_rows=512)
    monkeypatch.setenv("PSA_TRACEBACK_KERNEL", "v1")
    r2 = psa_amd.align(a, b, 1)
    monkeypatch.setenv("PSA_FILL_KERNEL", "v1")
    r3 = psa_amd.align(a, b, 1)
    sc, ai, bj = O.align(a, b)
    for r in (r1, rw, rw3, rwb, r2, r3):
        assert r.score == sc and np.array_equal(r.ai, ai) and np.array_equal(r.bj, bj)


@pytest.mark.parametrize("band", [256, 512, 768])
def test_banded_linear_space_alignment(band):
    """psa_align_banded (codes of one band of rows at a time) against the oracle: sizes that
    are no multiple of the band, alphabets with long runs and long gaps, a pair shorter than a band."""
    rng = np.random.default_rng(100 + band)
    cases = []
    for n, m, alpha in ((700, 650, 4), (513, 1200, 2), (1300, 300, 20), (50, 40, 3), (1000, 1000, 4)):
        cases.append((rng.integers(65, 65 + alpha, n).astype(np.uint8), rng.integers(65, 65 + alpha, m).astype(np.uint8)))
    x = rng.integers(65, 69, 900).astype(np.uint8)
    cases.append((x, np.concatenate([x[:300], x[500:]])))            # one long gap
    cases.append((np.concatenate([x[:200], x[650:]]), x))
    for a, b in cases:
        res = psa_amd.align(a, b, 1, band_rows=band)
        sc, ai, bj = O.align(a, b)
        assert res.score == sc, (a.size, b.size)
        assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (a.size, b.size)


def test_banded_through_env_and_large(monkeypatch):
    """PSA_BAND_ROWS switches psa_align itself; a 6000 x 7000 pair with 1024-row bands equals the plain path."""
    rng = np.random.default_rng(77)
    a = rng.integers(65, 69, 6000).astype(np.uint8)
    b = rng.integers(65, 69, 7000).astype(np.uint8)
    plain = psa_amd.align(a, b, 1)
    monkeypatch.setenv("PSA_BAND_ROWS", "1024")
    banded = psa_amd.align(a, b, 3)
    assert plain.score == banded.score and np.array_equal(plain.ai, banded.ai) and np.array_equal(plain.bj, banded.bj)
    sc, ai, bj = O.align(a, b)
    assert banded.score == sc and np.array_equal(banded.ai, ai) and np.array_equal(banded.bj, bj)


def test_banded_generic_kernels_and_tiny(monkeypatch):
    """Banded mode on a 60-symbol alphabet and degenerate sizes: with the table kernel (a table row per distinct
    residue: any byte alphabet) and with the generic kernel fill_strips_v1 (PSA_FILL_KERNEL=v1), plain and banded."""
    rng = np.random.default_rng(5)
    a = rng.integers(40, 100, 777).astype(np.uint8)      # 60 symbols
    b = rng.integers(40, 100, 555).astype(np.uint8)
    wide = (rng.integers(0, 256, 900).astype(np.uint8), rng.integers(0, 256, 700).astype(np.uint8))   # every byte value
    for force in (None, "v1"):
        if force:
            monkeypatch.setenv("PSA_FILL_KERNEL", force)
        for x, y in ((a, b), (a[:1], b), (a, b[:1]), (a[:300], a[:300]), wide):
            sc, ai, bj = O.align(x, y)
            for band in (256, None):
                res = psa_amd.align(x, y, 1, band_rows=band)
                assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (force, band, x.size, y.size)
            assert psa_amd.score(x, y, 3) == sc


def test_align_many_matches_oracle():
    """psa_align_many (run() in a loop, simple.cpp:311; batches of pairs filled by ONE launch and walked by ONE
    launch): different sizes in one batch, pairs without cells, an alphabet the table kernel does not take."""
    rng = np.random.default_rng(31)
    pairs = []
    for n, m, alpha in ((700, 650, 4), (64, 64, 2), (1, 300, 3), (300, 1, 3), (1500, 2100, 20), (257, 255, 4),
                        (0, 10, 4), (10, 0, 4), (0, 0, 4), (1000, 1000, 4), (333, 4000, 20), (2048, 2048, 20),
                        (129, 4097, 2)):
        pairs.append((rng.integers(65, 65 + alpha, n).astype(np.uint8), rng.integers(65, 65 + alpha, m).astype(np.uint8)))
    pairs.append((rng.integers(40, 100, 500).astype(np.uint8), rng.integers(40, 100, 444).astype(np.uint8)))   # 60 symbols
    x = rng.integers(65, 69, 3000).astype(np.uint8)
    pairs.append((x, np.concatenate([x[:1000], x[1400:]])))
    got = psa_amd.align_many(pairs)
    for (a, b), res in zip(pairs, got):
        sc, ai, bj = O.align(a, b)
        assert res.score == sc, (a.size, b.size)
        assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (a.size, b.size)
    scores = psa_amd.align_many(pairs, alignments=False)
    assert [r.score for r in scores] == [r.score for r in got]


def test_align_many_kept_engines_serve_later_calls():
    """psa_align_many keeps its engines for the thread's next call: a call with fewer and smaller pairs (slots of the
    last call left over), one of the other kind (scores only), one with larger pairs and more batches (new engines),
    a single-pair call in between and psa_release_cache() -- every result against the oracle."""
    rng = np.random.default_rng(47)

    def make(shapes):
        return [(rng.integers(65, 69, n).astype(np.uint8), rng.integers(65, 69, m).astype(np.uint8)) for n, m in shapes]

    def check(pairs, alignments=True):
        got = psa_amd.align_many(pairs, alignments=alignments)
        for (a, b), res in zip(pairs, got):
            sc, ai, bj = O.align(a, b)
            assert res.score == sc, (a.size, b.size)
            if alignments:
                assert np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (a.size, b.size)

    psa_amd.release_cache()
    check(make([(900, 800)] * 12 + [(512, 640)] * 5))           # one batch of 17
    check(make([(300, 280)] * 3))                               # 3 of the 17 slots
    check(make([(640, 512)] * 4), alignments=False)             # engines without codes
    check(make([(900, 800)] * 2))
    one = make([(700, 900)])[0]
    assert psa_amd.align(one[0], one[1]).score == O.align(one[0], one[1])[0]
    check(make([(1200, 1100)] * 45 + [(64, 64)] * 2))           # three batches, larger pairs
    check(make([(256, 256)] * 21))                              # two batches on engines of 16 slots
    psa_amd.release_cache()
    check(make([(300, 280)] * 3))
    psa_amd.release_cache()


def test_many_engine_reuse_and_device_inputs():
    """The engine itself: slots re-bound to pairs of other sizes between launches, residues taken from device memory."""
    import torch
    rng = np.random.default_rng(32)
    eng = psa_amd.Many(3, 3000, 3500)
    for rnd in range(3):
        batch = []
        for sl in range(3 if rnd != 1 else 2):
            n, m = int(rng.integers(1, 3000)), int(rng.integers(1, 3500))
            a = rng.integers(65, 69, n).astype(np.uint8)
            b = np.where(rng.random(m) < 0.6, np.resize(a, m), rng.integers(65, 69, m)).astype(np.uint8)
            batch.append((a, b))
            if rnd == 2:
                ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
                torch.cuda.synchronize()
                eng.set_pair_dev(sl, ta.data_ptr(), n, tb.data_ptr(), m)
                batch[-1] = (a, b, ta, tb)
            else:
                eng.set_pair(sl, a, b)
        if rnd == 1:
            eng.clear_slot(2)
        eng.launch()
        eng.wait()
        for sl, item in enumerate(batch):
            a, b = item[0], item[1]
            res = eng.result(sl, a.size, b.size)
            sc, ai, bj = O.align(a, b)
            assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (rnd, sl)
    # Pairs bound from device memory go into the launch before their alphabet size is back on the host (the launch never
    # waits on the host; the table kernel takes any byte alphabet): 60- and 87-symbol pairs beside an ordinary one in the
    # same launch, twice in a row.
    for rnd in range(2):
        items = []
        for sl, (lo, hi) in enumerate(((40, 100), (65, 69), (33, 120))):
            n, m = int(rng.integers(100, 3000)), int(rng.integers(100, 3500))
            a = rng.integers(lo, hi, n).astype(np.uint8)
            b = np.where(rng.random(m) < 0.5, np.resize(a, m), rng.integers(lo, hi, m)).astype(np.uint8)
            ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            items.append((a, b, ta, tb))
        torch.cuda.synchronize()
        for sl, (a, b, ta, tb) in enumerate(items):
            eng.set_pair_dev(sl, ta.data_ptr(), a.size, tb.data_ptr(), b.size)
        eng.launch()
        eng.wait()
        for sl, (a, b, ta, tb) in enumerate(items):
            res = eng.result(sl, a.size, b.size)
            sc, ai, bj = O.align(a, b)
            assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), ("wide alphabet", rnd, sl)
    eng.close()


def test_stripes_in_one_launch_at_size():
    """p stripes of one pair are jobs of ONE launch (flags between them): same score and alignment as p = 1 at a
    size where the stripes really overlap in time, and p > 8 (several launches of up to 8 stripes)."""
    a = psa_amd.synth_protein(0xA0000000 + 13, 1 << 13)
    b = psa_amd.synth_protein(0xB0000000 + 13, 1 << 13)
    ref = psa_amd.align(a, b, 1)
    assert ref.score == O.score(a, b, 8)
    for p in (2, 4, 8, 11):
        res = psa_amd.align(a, b, p)
        assert res.score == ref.score and np.array_equal(res.ai, ref.ai) and np.array_equal(res.bj, ref.bj), p
        assert psa_amd.score(a, b, p) == ref.score


def test_seqset_loader_records_and_directory(tmp_path):
    """Batch input: multi-record FASTA files and a directory of FASTA files -> sequences resident on
    the device in the layout the batch kernel takes -> all-pairs scores equal to the oracle's, pair by pair
    (including a target longer than the batch kernel's row buffer and an empty record)."""
    rng = np.random.default_rng(41)
    recs = [rng.integers(65, 65 + [3, 4, 20][k % 3], ln).astype(np.uint8).tobytes()
            for k, ln in enumerate([120, 1, 64, 300, 1500, 77, 512, 0, 255])]
    def fasta(items, crlf=False):
        out = b""
        for name, r in items:
            out += b">" + name + b"\n"
            for x in range(0, len(r), 60):
                out += r[x:x + 60] + b"\n"
        return out
    (tmp_path / "multi.fasta").write_bytes(fasta([(b"r%d" % k, recs[k]) for k in range(4)]))
    d = tmp_path / "dir"
    d.mkdir()
    (d / "b.fa").write_bytes(fasta([(b"r6", recs[6]), (b"r7 empty", recs[7]), (b"r8", recs[8])]))
    (d / "a.fa").write_bytes(fasta([(b"r4 long", recs[4]), (b"r5", recs[5])]))
    s = psa_amd.SeqSet([str(tmp_path / "multi.fasta"), str(d)])
    assert len(s) == 9 and [s.residues(k) for k in range(9)] == recs       # directory in name order: a.fa, b.fa
    assert s.name(0) == "r0" and s.name(4) == "r4 long" and s.name(7) == "r7 empty"
    got = s.score_block(0, 9, 0, 9)
    for q in range(9):
        for t in range(9):
            assert got[q, t] == O.score(recs[q], recs[t]), (q, t)
    sub = s.score_block(2, 5, 5, 9)
    assert np.array_equal(sub, got[2:5, 5:9])
    # the device buffers are what psa_score_batch_dev takes
    import torch
    pairs = torch.tensor([[0, 3], [8, 6]], dtype=torch.int32, device="cuda")
    sc = torch.zeros(2, dtype=torch.int32, device="cuda")
    ctr = torch.zeros(64, dtype=torch.int32, device="cuda")
    rc = psa_amd.lib().psa_score_batch_dev(s.seqs_dev(), s.offsets_dev(), pairs.data_ptr(), 2, 512, sc.data_ptr(), ctr.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert sc.tolist() == [int(got[0, 3]), int(got[8, 6])]
    s.close()


def test_chunks_in_flight_on_alternating_streams():
    """Row chunks of one stripe launched on alternating streams (chunk c+1 starts while chunk c runs: its first
    strip takes the granules and the corner triple the chunk above publishes), for a head stripe and for a stripe
    with a left boundary -- what distributed.py does on every rank."""
    import torch
    rng = np.random.default_rng(51)
    n, m, cut = 5000, 4100, 1700
    a = rng.integers(65, 69, n).astype(np.uint8)
    b = np.where(rng.random(m) < 0.7, np.resize(a, m), rng.integers(65, 69, m)).astype(np.uint8)
    sc, ai, bj = O.align(a, b)
    s0 = psa_amd.Stripe(a, b, 0, cut, traceback=True)
    s1 = psa_amd.Stripe(a, b, cut, m, traceback=True)
    streams = [torch.cuda.Stream() for _ in range(4)]
    for chunk in (512, 1024, 64 * 7):
        edges, evs = [], []
        for c, r0 in enumerate(range(0, n, chunk)):
            r1 = min(n, r0 + chunk)
            e = torch.zeros(3 * (r1 - r0), dtype=torch.int32, device="cuda")
            edges.append(e)
            with torch.cuda.stream(streams[c % 2]):
                s0.fill_rows(r0, r1, None, e.data_ptr(), streams[c % 2].cuda_stream)
                ev = torch.cuda.Event()
                ev.record(streams[c % 2])
                evs.append(ev)
        for c, r0 in enumerate(range(0, n, chunk)):
            r1 = min(n, r0 + chunk)
            sm = streams[2 + c % 2]
            sm.wait_event(evs[c])
            s1.fill_rows(r0, r1, edges[c].data_ptr(), None, sm.cuda_stream)
        torch.cuda.synchronize()
        t = s1.final()
        k, v = psa_amd.pick(*t)
        assert v == sc, chunk
        cap = 2 * (n + m) + 2
        xa, xb = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        i, j, k, ln = s1.traceback(n, m, k, xa, xb, 0)
        if i > 0 and j > 0:
            i, j, k, ln = s0.traceback(i, j, k, xa, xb, ln)
        fa, fb = psa_amd.finish_alignment(xa[:ln], xb[:ln], n, m)
        assert np.array_equal(fa, ai) and np.array_equal(fb, bj), chunk
    s0.close(); s1.close()


@pytest.mark.gpu
def test_flagged_fill_boundaries_by_flags():
    """psa_stripe_fill_flagged: each stripe is ONE launch over all rows; the right boundary of the head stripe leaves in
    chunks (psa_edge_take waits on the device for a chunk's last strip) and arrives at the next stripe in chunks
    (psa_edge_post raises the strips' flags) while both launches run -- the multi-GPU pipeline on one GPU.  The tail
    stripe is launched FIRST, so every one of its strips starts out waiting."""
    import torch
    rng = np.random.default_rng(77)
    n, m, cut = 6000, 5000, 2100
    a = rng.integers(65, 69, n).astype(np.uint8)
    b = np.where(rng.random(m) < 0.7, np.resize(a, m), rng.integers(65, 69, m)).astype(np.uint8)
    sc, ai, bj = O.align(a, b)
    s0 = psa_amd.Stripe(a, b, 0, cut, traceback=True)
    s1 = psa_amd.Stripe(a, b, cut, m, traceback=True)
    strips = (n + 63) // 64
    # (each fill on a stream of a priority class of its own, i.e. a hardware queue no other stream here shares: psa.h)
    f0, f1 = psa_amd.stream_create(+1), psa_amd.stream_create(-1)
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
    for chunk in (1024, 64 * 5, 8192):
        right = torch.empty(3 * n, dtype=torch.int32, device="cuda")
        left = torch.full((3 * n,), 0x3f3f3f3f, dtype=torch.int32, device="cuda")
        right_done = torch.zeros(strips, dtype=torch.int32, device="cuda")
        left_ready = torch.zeros(strips, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        s1.fill_flagged(left.data_ptr(), left_ready.data_ptr(), None, None, f1)
        s0.fill_flagged(None, None, right.data_ptr(), right_done.data_ptr(), f0)
        bufs = []
        for r0 in range(0, n, chunk):
            rows = min(n, r0 + chunk) - r0
            buf = torch.empty(3 * rows, dtype=torch.int32, device="cuda")
            bufs.append(buf)
            s0.edge_take(right.data_ptr(), right_done.data_ptr(), r0, rows, buf.data_ptr(), s_out.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(s_out)
            s_in.wait_event(ev)
            psa_amd.edge_post(buf.data_ptr(), r0, rows, n, left.data_ptr(), left_ready.data_ptr(), s_in.cuda_stream)
        torch.cuda.synchronize()
        assert int(right_done.min()) == 1 and int(left_ready.min()) == 1
        assert torch.equal(left, right), chunk
        k, v = psa_amd.pick(*s1.final())
        assert v == sc, chunk
        cap = 2 * (n + m) + 2
        xa, xb = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        i, j, k, ln = s1.traceback(n, m, k, xa, xb, 0)
        if i > 0 and j > 0:
            i, j, k, ln = s0.traceback(i, j, k, xa, xb, ln)
        fa, fb = psa_amd.finish_alignment(xa[:ln], xb[:ln], n, m)
        assert np.array_equal(fa, ai) and np.array_equal(fb, bj), chunk
    # flags without their buffer, a chunk off the 64-row grid
    with pytest.raises(psa_amd.PsaError):
        s0.fill_flagged(None, None, None, right_done.data_ptr(), f0)
    with pytest.raises(psa_amd.PsaError):
        psa_amd.edge_post(bufs[0].data_ptr(), 32, 64, n, left.data_ptr(), left_ready.data_ptr(), 0)
    s0.close(); s1.close()
    psa_amd.stream_destroy(f0); psa_amd.stream_destroy(f1)


def test_align_multi_stripes_as_separate_launches(golden_dir):
    """psa_align_multi / psa_score_multi (run() with its workers on several GPUs of one process, simple.cpp:261-277):
    every stripe is a launch of its own on its own stream and the boundary travels through fine-grained memory and
    per-strip flags that the neighbour's launch polls.  On the one GPU of the test box: ngpu = 2, 3 and 5 with every
    stripe on device 0 (the launches are resident together), against the oracle column for column; sizes that leave
    partial strips and narrow last stripes; the error paths."""
    rng = np.random.default_rng(91)
    for n, m, alpha in ((6000, 5000, 4), (700, 2211, 20), (130, 130, 2), (64, 7, 3), (5000, 9000, 20)):
        a = rng.integers(65, 65 + alpha, n).astype(np.uint8)
        b = np.where(rng.random(m) < 0.6, np.resize(a, m), rng.integers(65, 65 + alpha, m)).astype(np.uint8)
        sc, ai, bj = O.align(a, b)
        t1 = psa_amd.score(a, b, 1, triple=True)
        for ng in (2, 3, 5):
            res = psa_amd.align_multi(a, b, devices=[0] * ng)
            assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (n, m, ng)
            assert psa_amd.score_multi(a, b, devices=[0] * ng, triple=True) == t1, (n, m, ng)
        res = psa_amd.align_multi(a, b, ngpu=2)      # no device list: stripe g on GPU g mod the visible GPUs
        assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj), (n, m)
    # the reference's FASTA pairs and degenerate inputs
    a = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "Q16739_homo.fasta"))
    b = psa_amd.read_fasta(os.path.join(golden_dir, "sequences", "A0A2I3N2W7_baboon.fasta"))
    gold = [c for c in O.load_golden("ref_fasta_pairs.json") if c["a"] == "Q16739_homo.fasta" and c["b"] == "A0A2I3N2W7_baboon.fasta"][0]
    assert psa_amd.format_result(psa_amd.align_multi(a, b, devices=[0, 0, 0, 0]), a, b) == O.b64d(gold["stdout_b64"])
    for x, y in ((b"", b"ACD"), (b"ACD", b""), (b"A", b"C"), (b"ACDEF", b"AC")):
        res = psa_amd.align_multi(x, y, devices=[0, 0, 0])          # (ngpu > m: zero-width stripes own no cell)
        sc, ai, bj = O.align(x, y)
        assert (res.score, list(res.ai), list(res.bj)) == (sc, list(ai), list(bj)), (x, y)
    # a 60-symbol alphabet
    a = rng.integers(40, 100, 777).astype(np.uint8)
    b = rng.integers(40, 100, 555).astype(np.uint8)
    sc, ai, bj = O.align(a, b)
    res = psa_amd.align_multi(a, b, devices=[0, 0])
    assert res.score == sc and np.array_equal(res.ai, ai) and np.array_equal(res.bj, bj)
    # stripes that share a GPU and do not fit on it together; a device that does not exist; ngpu < 1
    big = psa_amd.synth_protein(1, 1 << 16)
    with pytest.raises(psa_amd.PsaError):
        psa_amd.align_multi(big, big, devices=[0, 0])
    with pytest.raises(psa_amd.PsaError):
        psa_amd.align_multi(a, b, devices=[0, 99])
    with pytest.raises(psa_amd.PsaError):
        psa_amd.align_multi(a, b, ngpu=0)


def test_cli_psa_gpus(golden_dir):
    """PSA_GPUS=N ./simple A B p: the front end's route to psa_align_multi, byte-exact stdout."""
    exe = os.path.join(ROOT, "simple")
    env = dict(os.environ, PSA_GPUS="3")
    for c in O.load_golden("ref_fasta_pairs.json")[:4]:
        r = subprocess.run([exe, c["a"], c["b"], str(c["p"])], cwd=golden_dir, capture_output=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout == O.b64d(c["stdout_b64"])
