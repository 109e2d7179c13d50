"""ctypes access to oracle/libpsa_oracle.so -- the CHECKER.  Only tests, smoke() and bench.py's
cpu_baseline leg may import this module; the product never does."""
import base64
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libpsa_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_lib = None
_i64, _i32 = C.c_int64, C.c_int32


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libpsa_oracle.so"])
        L = C.CDLL(SO)
        L.psa_oracle_format.restype = C.c_void_p
        L.psa_oracle_read_fasta.restype = C.c_void_p
        _lib = L
    return _lib


def seq(x):
    if isinstance(x, str):
        x = x.encode("latin-1")
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def score(a, b, p=0):
    a, b = seq(a), seq(b)
    s = _i32()
    t = (_i32 * 3)()
    if p:
        rc = lib().psa_oracle_score_mt(C.c_int(p), _p(a), _i64(a.size), _p(b), _i64(b.size), C.byref(s))
    else:
        rc = lib().psa_oracle_score(_p(a), _i64(a.size), _p(b), _i64(b.size), C.byref(s), t)
    assert rc == 0
    return s.value


def align(a, b, p=0):
    """Returns (score, ai, bj); p = 0 single-thread, p >= 1 the striped multi-thread variant."""
    a, b = seq(a), seq(b)
    cap = 2 * (a.size + b.size) + 2
    ai = np.zeros(cap, np.int32)
    bj = np.zeros(cap, np.int32)
    s, ln = _i32(), _i64()
    if p:
        rc = lib().psa_oracle_align_mt(C.c_int(p), _p(a), _i64(a.size), _p(b), _i64(b.size), C.byref(s), _p(ai),
                                       _p(bj), _i64(cap), C.byref(ln))
    else:
        rc = lib().psa_oracle_align(_p(a), _i64(a.size), _p(b), _i64(b.size), C.byref(s), _p(ai), _p(bj),
                                    _i64(cap), C.byref(ln))
    assert rc == 0, rc
    return s.value, ai[:ln.value].copy(), bj[:ln.value].copy()


def format_result(sc, a, b, ai, bj):
    a, b = seq(a), seq(b)
    ai = np.ascontiguousarray(ai, np.int32)
    bj = np.ascontiguousarray(bj, np.int32)
    n = _i64()
    p = lib().psa_oracle_format(_i32(sc), _p(a), _p(b), _p(ai), _p(bj), _i64(ai.size), C.byref(n))
    out = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return out


def stdout_for(a, b, p=0):
    sc, ai, bj = align(a, b, p)
    return format_result(sc, a, b, ai, bj)


def read_fasta(path):
    n = _i64()
    p = lib().psa_oracle_read_fasta(os.fsencode(path), C.byref(n))
    if not p:
        return b""
    out = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return out


def block(a, b, r0, r1, c0, c1, top, left):
    """top: int32[3][c1-c0+1], left: int32[3][r1-r0+1] -> (bottom, right) same shapes."""
    a, b = seq(a), seq(b)
    top = np.ascontiguousarray(top, np.int32)
    left = np.ascontiguousarray(left, np.int32)
    bottom = np.zeros_like(top)
    right = np.zeros_like(left)
    P3 = C.c_void_p * 3
    mk = lambda arr: P3(*[arr[k].ctypes.data for k in range(3)])
    rc = lib().psa_oracle_block(_p(a), _p(b), _i64(r0), _i64(r1), _i64(c0), _i64(c1), mk(top), mk(left), mk(bottom),
                                mk(right))
    assert rc == 0
    return bottom, right


def boundary(i, j):
    t = (_i32 * 3)()
    lib().psa_oracle_boundary(_i64(i), _i64(j), t)
    return tuple(t)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def b64d(s):
    return base64.b64decode(s)


def block_codes(a, b, r0, r1, c0, c1, top, left):
    """As block(), also returning the (r1-r0) x (c1-c0) matrix of destination-indexed winner codes."""
    a, b = seq(a), seq(b)
    top = np.ascontiguousarray(top, np.int32)
    left = np.ascontiguousarray(left, np.int32)
    bottom = np.zeros_like(top)
    right = np.zeros_like(left)
    codes = np.zeros((r1 - r0, c1 - c0), np.uint8)
    P3 = C.c_void_p * 3
    mk = lambda arr: P3(*[arr[k].ctypes.data for k in range(3)])
    rc = lib().psa_oracle_block_codes(_p(a), _p(b), _i64(r0), _i64(r1), _i64(c0), _i64(c1), mk(top), mk(left),
                                      mk(bottom), mk(right), _p(codes))
    assert rc == 0
    return bottom, right, codes
