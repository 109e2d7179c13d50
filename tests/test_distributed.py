"""CPU suite: the multi-rank column-stripe orchestration (parallel-sequence-alignment_amd/distributed.py)
with world_size 2 and 3 over gloo.  The compute engine is an oracle-backed stand-in injected by the
test (the product engine is the HIP stripe; it is exercised by the gpu suite), so what is checked
here is the partition, the boundary hand-off, the traceback chain and the gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleStripeEngine:
    """CPU stand-in with the interface of HipStripeEngine (test only)."""

    def __init__(self, a, b, c0, c1, traceback=True):
        self.a, self.b, self.c0, self.c1 = a, b, c0, c1
        self.n, self.m = len(a), len(b)
        self.w = c1 - c0
        self.top = np.array([[O.boundary(0, c0 + x)[k] for x in range(self.w + 1)] for k in range(3)], np.int32)
        self.corner = np.array(O.boundary(0, c0), np.int32)
        self.codes = np.zeros((self.n + 1, self.w + 1), np.uint8)
        # T1,T2,T3 of column c1 for every row: the walk's step INTO this stripe from the right is decided by them
        self.right_all = np.zeros((3, self.n + 1), np.int32)
        self.right_all[:, 0] = O.boundary(0, c1)
        self.final_triple = None

    def new_edge(self, rows):
        return torch.zeros(3 * rows, dtype=torch.int32)

    def fill_rows(self, r0, r1, left, right):
        rows = r1 - r0
        lf = np.zeros((3, rows + 1), np.int32)
        if left is None:
            for y in range(rows + 1):
                lf[:, y] = O.boundary(r0 + y, 0)
        else:
            lf[:, 0] = self.corner
            lf[:, 1:] = left.numpy().reshape(3, rows)
        bottom, rgt, codes = O.block_codes(self.a, self.b, r0, r1, self.c0, self.c1, self.top, lf)
        self.codes[r0 + 1:r1 + 1, 1:] = codes
        self.top = bottom
        self.corner = lf[:, rows].copy()
        self.right_all[:, r0 + 1:r1 + 1] = rgt[:, 1:]
        if right is not None:
            right.copy_(torch.from_numpy(rgt[:, 1:].reshape(-1).copy()))
        self.final_triple = tuple(int(x) for x in bottom[:, self.w])

    def final(self):
        return self.final_triple

    @staticmethod
    def winner_plain(t, kind):
        """Winner among a source cell's tables for aggregate kind (simple.cpp:162/179/198 tie-break)."""
        t1, t2, t3 = (int(x) for x in t)
        if kind == 2:
            t1, t2, t3 = t1 - 3, t2 - 1, t3 - 3
        elif kind == 3:
            t1, t2, t3 = t1 - 3, t2 - 3, t3 - 1
        if t1 >= t2 and t1 >= t3:
            return 1
        return 2 if t2 >= t3 else 3

    def traceback(self, i, j, k):
        """The hand-over convention of include/psa.h (psa_stripe_traceback): a state names a cell whose column
        is not emitted yet; the stripe that owns the PREDECESSOR's column takes the step."""
        ai, bj = [], []
        while i > 0 and j > 0:
            pi, pj = i - (k != 2), j - (k != 3)
            if self.c0 > 0 and pj <= self.c0:
                break                         # the predecessor belongs to the stripe on the left
            if j == self.c1 + 1:              # the cell lies in the stripe on the right, its predecessor in column c1
                w = self.winner_plain(self.right_all[:, pi], k)
            else:
                w = (int(self.codes[i, j - self.c0]) >> (2 * (k - 1))) & 3
            ai.append(-1 if w == 2 else i - 1)
            bj.append(-1 if w == 3 else j - 1)
            i, j, k = pi, pj, w
        return i, j, k, np.array(ai, np.int32), np.array(bj, np.int32)

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cases, chunk, out):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    D = importlib.import_module("parallel-sequence-alignment_amd.distributed")
    res = []
    for a, b, tb in cases:
        r = D.align_striped(a, b, chunk_rows=chunk, traceback=tb, engine_cls=OracleStripeEngine)
        res.append((r.score, r.ai.tolist(), r.bj.tolist()))
    if rank == world - 1:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _cases():
    rng = np.random.default_rng(3)
    cases = []
    for n, m, alpha in ((200, 190, 3), (130, 64, 4), (65, 300, 20), (1, 5, 2), (70, 2, 2), (257, 129, 3)):
        a = rng.integers(65, 65 + alpha, n).astype(np.uint8)
        b = rng.integers(65, 65 + alpha, m).astype(np.uint8)
        cases.append((a, b, True))
    cases.append((cases[0][0], cases[0][1], False))
    cases.append((np.zeros(0, np.uint8), cases[0][1], True))
    return cases


@pytest.mark.parametrize("world,chunk", [(2, 64), (3, 128)])
def test_striped_alignment_over_gloo(world, chunk):
    cases = _cases()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, chunk, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (a, b, tb), (score, ai, bj) in zip(cases, res):
        s, xa, xb = O.align(a, b)
        assert score == s
        if tb:
            assert ai == xa.tolist() and bj == xb.tolist()
        else:
            assert ai == [] and bj == []


def test_stripe_bounds_match_reference_partition():
    import importlib
    D = importlib.import_module("parallel-sequence-alignment_amd.distributed")
    # simple.cpp:90-99: block = size / p; last stripe takes size % p more
    for m, p in ((105, 4), (7, 8), (64, 1), (526, 16)):
        block = m // p
        for g in range(p):
            c0, c1 = D.stripe_bounds(m, p, g)
            assert c0 == g * block
            assert c1 == (m if g == p - 1 else c0 + block)
