"""ctypes binding of include/psa.h (the reference-side binding a maintainer would add)."""
import ctypes as C
import os
import sys
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.environ.get("PSA_LIB_PATH") or os.path.join(_HERE, "lib", "libpsa_hip.so")

PSA_F_TRACEBACK = 1

_lib = None


class PsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("psa error %d: %s" % (code, msg))
        self.code = code


#: mirror of the reference's ``result`` (simple.h:54-69): score + alignment columns
Result = namedtuple("Result", ["score", "ai", "bj"])

_i64, _i32, _u8p = C.c_int64, C.c_int32, C.POINTER(C.c_uint8)
_i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

_SIGNATURES = {
    "psa_last_error": (C.c_char_p, []),
    "psa_version": (C.c_char_p, []),
    "psa_device_count": (C.c_int, []),
    "psa_set_device": (C.c_int, [C.c_int]),
    "psa_free": (None, [C.c_void_p]),
    "psa_align": (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, C.c_int, _i32p, C.POINTER(C.c_void_p),
                            C.POINTER(C.c_void_p), _i64p]),
    "psa_align_banded": (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, _i64, _i32p, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), _i64p]),
    "psa_align_multi": (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, C.POINTER(C.c_int), C.c_int, _i32p,
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i64p]),
    "psa_score_multi": (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, C.POINTER(C.c_int), C.c_int, _i32p, _i32p]),
    "psa_multi_last_fill_ms": (C.c_double, []),
    "psa_score": (C.c_int, [C.c_void_p, _i64, C.c_void_p, _i64, C.c_int, _i32p, _i32p]),
    "psa_score_batch": (C.c_int, [C.c_void_p, C.c_void_p, _i64, C.c_void_p, _i64, C.c_void_p]),
    "psa_score_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i64, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "psa_score_batch_dev_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i64, C.c_int, C.c_void_p, C.c_void_p]),
    "psa_format_result": (C.c_void_p, [_i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64, _i64p]),
    "psa_finish_alignment": (C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64p]),
    "psa_stripe_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, _i64, C.c_void_p, _i64, _i64, _i64,
                                    C.c_uint32]),
    "psa_stripe_destroy": (None, [C.c_void_p]),
    "psa_stripe_set_sequences_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "psa_stripe_fill_rows": (C.c_int, [C.c_void_p, _i64, _i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "psa_stripe_fill_flagged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "psa_stream_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "psa_stream_destroy": (C.c_int, [C.c_void_p]),
    "psa_edge_post": (C.c_int, [C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "psa_edge_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i64, _i64, C.c_void_p, C.c_void_p]),
    "psa_stripe_final": (C.c_int, [C.c_void_p, C.c_void_p, _i32p]),
    "psa_stripe_traceback": (C.c_int, [C.c_void_p, _i64p, _i64p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                       _i64, _i64p, C.c_void_p]),
    "psa_stripe_last_fill_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "psa_stripe_last_traceback_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "psa_stripe_code_bytes": (_i64, [C.c_void_p]),
    "psa_read_fasta": (C.c_void_p, [C.c_char_p, _i64p]),
    "psa_read_fasta_records": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i64p]),
    "psa_synth_protein": (None, [C.c_uint64, _i64, C.c_void_p]),
    "psa_release_cache": (None, []),
    "psa_seqset_load": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_char_p), C.c_int]),
    "psa_seqset_destroy": (None, [C.c_void_p]),
    "psa_seqset_count": (_i64, [C.c_void_p]),
    "psa_seqset_length": (_i64, [C.c_void_p, _i64]),
    "psa_seqset_name": (C.c_char_p, [C.c_void_p, _i64]),
    "psa_seqset_residues": (C.c_void_p, [C.c_void_p, _i64]),
    "psa_seqset_seqs_dev": (C.c_void_p, [C.c_void_p]),
    "psa_seqset_offsets_dev": (C.c_void_p, [C.c_void_p]),
    "psa_seqset_score_block": (C.c_int, [C.c_void_p, _i64, _i64, _i64, _i64, C.c_void_p]),
    "psa_align_many": (C.c_int, [_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "psa_many_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _i64, _i64, C.c_uint32]),
    "psa_many_destroy": (None, [C.c_void_p]),
    "psa_many_set_pair": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "psa_many_set_pair_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _i64, C.c_void_p, _i64]),
    "psa_many_clear_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "psa_many_poll": (C.c_int, [C.c_void_p]),
    "psa_many_fill_ms_since": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "psa_many_launch": (C.c_int, [C.c_void_p]),
    "psa_many_launch_after": (C.c_int, [C.c_void_p, C.c_void_p]),
    "psa_many_wait": (C.c_int, [C.c_void_p]),
    "psa_many_result": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, C.c_void_p, C.c_void_p, _i64, _i64p]),
    "psa_many_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "psa_many_code_bytes": (_i64, [C.c_void_p, C.c_int]),
}

EXPORTS = tuple(sorted(_SIGNATURES))


def lib():
    """Load libpsa_hip.so (built in-tree by `make` / __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise PsaError(-2, "%s is missing: run `make` (there is no CPU fallback)" % lib_path)
        if "torch" in sys.modules:   # torch brings its own HIP runtime, which must see the GPU first
            try:
                if sys.modules["torch"].cuda.is_available():
                    sys.modules["torch"].cuda.init()
            except Exception:
                pass
        L = C.CDLL(lib_path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise PsaError(rc, lib().psa_last_error().decode("utf-8", "replace"))


def _seq(x):
    """bytes / str / uint8 array -> contiguous uint8 numpy array (identity byte encoding)."""
    if isinstance(x, str):
        x = x.encode("latin-1")
    if isinstance(x, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr.size else None


def device_count():
    return lib().psa_device_count()


def set_device(dev):
    _check(lib().psa_set_device(int(dev)))


def pick(t1, t2, t3):
    """The reference's three-way tie-break (simple.cpp:162/166/170): returns (table, value)."""
    if t1 >= t2 and t1 >= t3:
        return 1, t1
    if t2 >= t3:
        return 2, t2
    return 3, t3


def align(a, b, p=1, band_rows=None):
    """run(p, A, B) of the reference (simple.cpp:261): returns Result(score, ai, bj).  band_rows (a multiple
    of 64, or 0 for the library's choice) selects the linear-space variant psa_align_banded."""
    a, b = _seq(a), _seq(b)
    score = _i32()
    ai, bj, n = C.c_void_p(), C.c_void_p(), _i64()
    if band_rows is None:
        _check(lib().psa_align(_ptr(a), a.size, _ptr(b), b.size, int(p), C.byref(score), C.byref(ai), C.byref(bj),
                               C.byref(n)))
    else:
        _check(lib().psa_align_banded(_ptr(a), a.size, _ptr(b), b.size, int(band_rows), C.byref(score), C.byref(ai),
                                      C.byref(bj), C.byref(n)))
    try:
        if n.value:
            xa = np.ctypeslib.as_array(C.cast(ai, _i32p), shape=(n.value,)).copy()
            xb = np.ctypeslib.as_array(C.cast(bj, _i32p), shape=(n.value,)).copy()
        else:
            xa = np.zeros(0, np.int32)
            xb = np.zeros(0, np.int32)
    finally:
        lib().psa_free(ai)
        lib().psa_free(bj)
    return Result(score.value, xa, xb)


run = lambda p, a, b: align(a, b, p)  # noqa: E731  (argument order of the reference)


def _devices(devices, ngpu):
    if devices is None:
        return None, int(ngpu)
    arr = (C.c_int * len(devices))(*[int(d) for d in devices])
    return arr, len(devices)


def align_multi(a, b, ngpu=None, devices=None):
    """run(p, A, B) with its p = ngpu workers on several GPUs of this process (psa_align_multi): stripe g on
    devices[g] (default: GPUs 0 .. ngpu-1)."""
    a, b = _seq(a), _seq(b)
    dev, ng = _devices(devices, ngpu)
    score = _i32()
    ai, bj, n = C.c_void_p(), C.c_void_p(), _i64()
    _check(lib().psa_align_multi(_ptr(a), a.size, _ptr(b), b.size, dev, ng, C.byref(score), C.byref(ai), C.byref(bj), C.byref(n)))
    try:
        xa = np.ctypeslib.as_array(C.cast(ai, _i32p), shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
        xb = np.ctypeslib.as_array(C.cast(bj, _i32p), shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
    finally:
        lib().psa_free(ai)
        lib().psa_free(bj)
    return Result(score.value, xa, xb)


def multi_last_fill_ms():
    return lib().psa_multi_last_fill_ms()


def score_multi(a, b, ngpu=None, devices=None, triple=False):
    a, b = _seq(a), _seq(b)
    dev, ng = _devices(devices, ngpu)
    s = _i32()
    t = (_i32 * 3)()
    _check(lib().psa_score_multi(_ptr(a), a.size, _ptr(b), b.size, dev, ng, C.byref(s), t))
    return (s.value, tuple(t)) if triple else s.value


def score(a, b, p=1, triple=False):
    a, b = _seq(a), _seq(b)
    s = _i32()
    t = (_i32 * 3)()
    _check(lib().psa_score(_ptr(a), a.size, _ptr(b), b.size, int(p), C.byref(s), t))
    return (s.value, tuple(t)) if triple else s.value


def score_batch(seqs, pairs):
    """Scores of many independent pairs (psa_score_batch).  seqs: list of byte strings / uint8 arrays;
    pairs: iterable of (index of A, index of B).  Returns an int32 numpy array."""
    arrs = [_seq(x) for x in seqs]
    offsets = np.zeros(len(arrs) + 1, np.int64)
    offsets[1:] = np.cumsum([x.size for x in arrs])
    flat = np.concatenate(arrs) if arrs and offsets[-1] else np.zeros(1, np.uint8)
    pr = np.ascontiguousarray(np.asarray(list(pairs), dtype=np.int32).reshape(-1, 2))
    out = np.zeros(pr.shape[0], np.int32)
    _check(lib().psa_score_batch(_ptr(flat), _ptr(offsets), len(arrs), _ptr(pr), pr.shape[0], _ptr(out)))
    return out


def format_result(res, a, b):
    """The reference's stdout for a result (simple.cpp:312-313, 43-70), as bytes."""
    a, b = _seq(a), _seq(b)
    ai = np.ascontiguousarray(res.ai, np.int32)
    bj = np.ascontiguousarray(res.bj, np.int32)
    n = _i64()
    p = lib().psa_format_result(res.score, _ptr(a), _ptr(b), _ptr(ai), _ptr(bj), ai.size, C.byref(n))
    try:
        return C.string_at(p, n.value)
    finally:
        lib().psa_free(p)


def finish_alignment(ai_rev, bj_rev, n, m):
    """Prefix fix-up of simple.cpp:279-290 on a reversed traceback; returns forward (ai, bj)."""
    ln = len(ai_rev)
    cap = ln + n + m + 2
    ai = np.zeros(cap, np.int32)
    bj = np.zeros(cap, np.int32)
    ai[:ln] = ai_rev
    bj[:ln] = bj_rev
    out = _i64()
    _check(lib().psa_finish_alignment(_ptr(ai), _ptr(bj), ln, cap, C.byref(out)))
    return ai[:out.value].copy(), bj[:out.value].copy()


def read_fasta(path):
    n = _i64()
    p = lib().psa_read_fasta(os.fsencode(path), C.byref(n))
    if not p:
        return b""
    try:
        return C.string_at(p, n.value)
    finally:
        lib().psa_free(p)


def read_fasta_records(path):
    """Every '>' record of a FASTA file as its own byte string (psa_read_fasta_records): batch input."""
    seqs, offs, n = C.c_void_p(), C.c_void_p(), _i64()
    _check(lib().psa_read_fasta_records(os.fsencode(path), C.byref(seqs), C.byref(offs), C.byref(n)))
    try:
        off = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        flat = C.string_at(seqs, int(off[-1]))
        return [flat[off[r]:off[r + 1]] for r in range(n.value)]
    finally:
        lib().psa_free(seqs)
        lib().psa_free(offs)


def synth_protein(seed, n):
    out = np.zeros(int(n), np.uint8)
    lib().psa_synth_protein(C.c_uint64(seed), int(n), _ptr(out))
    return out


class SeqSet:
    """Sequences resident on the device, loaded from FASTA files / directories (psa_seqset_*)."""

    def __init__(self, paths):
        if isinstance(paths, (str, bytes, os.PathLike)):
            paths = [paths]
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        self._h = C.c_void_p()
        _check(lib().psa_seqset_load(C.byref(self._h), arr, len(paths)))

    def close(self):
        if self._h:
            lib().psa_seqset_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __len__(self):
        return lib().psa_seqset_count(self._h)

    def name(self, i):
        return lib().psa_seqset_name(self._h, i).decode("utf-8", "replace")

    def residues(self, i):
        return C.string_at(lib().psa_seqset_residues(self._h, i), lib().psa_seqset_length(self._h, i))

    def seqs_dev(self):
        return lib().psa_seqset_seqs_dev(self._h)

    def offsets_dev(self):
        return lib().psa_seqset_offsets_dev(self._h)

    def score_block(self, q0, q1, t0, t1):
        out = np.zeros((q1 - q0, t1 - t0), np.int32)
        _check(lib().psa_seqset_score_block(self._h, q0, q1, t0, t1, _ptr(out)))
        return out


def release_cache():
    lib().psa_release_cache()


def align_many(pairs, alignments=True):
    """run() for every (a, b) of `pairs` (psa_align_many): a list of Result; ai/bj are empty when alignments is False."""
    seqs = [(_seq(a), _seq(b)) for a, b in pairs]
    k = len(seqs)
    pa = (C.c_void_p * k)(*[x[0].ctypes.data if x[0].size else None for x in seqs])
    pb = (C.c_void_p * k)(*[x[1].ctypes.data if x[1].size else None for x in seqs])
    na = (_i64 * k)(*[x[0].size for x in seqs])
    nb = (_i64 * k)(*[x[1].size for x in seqs])
    scores = (_i32 * k)()
    if alignments:
        ai, bj, ln = (C.c_void_p * k)(), (C.c_void_p * k)(), (_i64 * k)()
        _check(lib().psa_align_many(k, pa, na, pb, nb, scores, ai, bj, ln))
    else:
        _check(lib().psa_align_many(k, pa, na, pb, nb, scores, None, None, None))
    out = []
    for i in range(k):
        if alignments and ln[i]:
            xa = np.ctypeslib.as_array(C.cast(ai[i], _i32p), shape=(ln[i],)).copy()
            xb = np.ctypeslib.as_array(C.cast(bj[i], _i32p), shape=(ln[i],)).copy()
        else:
            xa, xb = np.zeros(0, np.int32), np.zeros(0, np.int32)
        if alignments:
            lib().psa_free(ai[i])
            lib().psa_free(bj[i])
        out.append(Result(scores[i], xa, xb))
    return out


class Many:
    """psa_many engine: up to `slots` pairs filled by ONE launch and walked by ONE launch."""

    def __init__(self, slots, n_cap, m_cap, traceback=True):
        self._h = C.c_void_p()
        self.slots, self.n_cap, self.m_cap, self.traceback = slots, int(n_cap), int(m_cap), traceback
        _check(lib().psa_many_create(C.byref(self._h), slots, self.n_cap, self.m_cap, PSA_F_TRACEBACK if traceback else 0))
        self._keep = {}

    def close(self):
        if self._h:
            lib().psa_many_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_pair(self, slot, a, b):
        a, b = _seq(a), _seq(b)
        self._keep[slot] = (a, b)
        _check(lib().psa_many_set_pair(self._h, slot, _ptr(a), a.size, _ptr(b), b.size))

    def set_pair_dev(self, slot, a_ptr, n, b_ptr, m):
        _check(lib().psa_many_set_pair_dev(self._h, slot, a_ptr, int(n), b_ptr, int(m)))

    def clear_slot(self, slot):
        _check(lib().psa_many_clear_slot(self._h, slot))

    def launch(self, after=None):
        """after: another Many whose fill launch must have ended before this one's begins."""
        _check(lib().psa_many_launch_after(self._h, after._h if after is not None else None))

    def wait(self):
        _check(lib().psa_many_wait(self._h))

    def result(self, slot, n, m, want_alignment=True):
        """Result of a slot after launch(): (score, ai, bj), waiting for the launch if need be."""
        sc = _i32()
        if not (want_alignment and self.traceback):
            _check(lib().psa_many_result(self._h, slot, C.byref(sc), None, None, None, 0, None))
            return Result(sc.value, np.zeros(0, np.int32), np.zeros(0, np.int32))
        cap = 2 * (int(n) + int(m)) + 2
        ai, bj, ln = np.zeros(cap, np.int32), np.zeros(cap, np.int32), _i64()
        _check(lib().psa_many_result(self._h, slot, C.byref(sc), None, _ptr(ai), _ptr(bj), cap, C.byref(ln)))
        return Result(sc.value, ai[:ln.value].copy(), bj[:ln.value].copy())

    def last_ms(self):
        f, w = C.c_float(), C.c_float()
        _check(lib().psa_many_last_ms(self._h, C.byref(f), C.byref(w)))
        return f.value, w.value

    def poll(self):
        """True when the last launch's results are on the host; queues the walks of a fill that has ended (no blocking)."""
        rc = lib().psa_many_poll(self._h)
        if rc < 0:
            _check(rc)
        return rc == 1

    def fill_ms_since(self, ref_event):
        """(begin, end) of the last launch's fill in ms after `ref_event` (a hipEvent_t handle, e.g. torch.cuda.Event.cuda_event)."""
        b, e = C.c_float(), C.c_float()
        _check(lib().psa_many_fill_ms_since(self._h, C.c_void_p(ref_event), C.byref(b), C.byref(e)))
        return b.value, e.value

    def code_bytes(self, slot):
        return lib().psa_many_code_bytes(self._h, slot)


def stream_create(priority_class=-1, reserved_cus=0):
    """hipStream_t (as an int) of a priority class of its own (-1 lowest, 0 normal, +1 highest) -- a hardware queue no
    normal stream shares -- or, reserved_cus > 0, one that leaves that many CUs to other streams (psa.h)."""
    h = C.c_void_p()
    _check(lib().psa_stream_create(C.byref(h), int(priority_class), int(reserved_cus)))
    return h.value


def stream_destroy(stream):
    _check(lib().psa_stream_destroy(stream))


def edge_post(packed_ptr, r0, rows, n, left_ptr, left_ready_ptr, stream=0):
    """Queue on `stream`: packed int32[3][rows] -> rows r0+1..r0+rows of the left boundary int32[3][n], then their flags."""
    _check(lib().psa_edge_post(packed_ptr, r0, rows, n, left_ptr, left_ready_ptr, stream))


class Stripe:
    """One column stripe of B on the current GPU (a thread's share in simple.cpp:87-259)."""

    def __init__(self, a, b, c0, c1, traceback=True):
        self.a, self.b = _seq(a), _seq(b)
        self.n, self.m, self.c0, self.c1 = self.a.size, self.b.size, int(c0), int(c1)
        self._h = C.c_void_p()
        _check(lib().psa_stripe_create(C.byref(self._h), _ptr(self.a), self.n, _ptr(self.b), self.m, self.c0,
                                       self.c1, PSA_F_TRACEBACK if traceback else 0))

    def close(self):
        if self._h:
            lib().psa_stripe_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_sequences_dev(self, a_ptr, b_ptr, stream=0):
        _check(lib().psa_stripe_set_sequences_dev(self._h, a_ptr, b_ptr, stream))

    def fill_rows(self, r0, r1, left_ptr=None, right_ptr=None, stream=0):
        """left_ptr / right_ptr: device addresses (int) of int32[3][r1-r0] or None."""
        _check(lib().psa_stripe_fill_rows(self._h, r0, r1, left_ptr, right_ptr, stream))

    def fill_flagged(self, left_ptr=None, left_ready_ptr=None, right_ptr=None, right_done_ptr=None, stream=0):
        """One launch over all rows; left_ptr / right_ptr: device int32[3][n]; the flag arrays: device uint32[(n+63)//64],
        zeroed by the caller (psa.h, psa_stripe_fill_flagged)."""
        _check(lib().psa_stripe_fill_flagged(self._h, left_ptr, left_ready_ptr, right_ptr, right_done_ptr, stream))

    def edge_take(self, right_ptr, right_done_ptr, r0, rows, packed_ptr, stream=0):
        """Queue on `stream`: wait for rows r0+1..r0+rows of the right boundary, pack them as int32[3][rows]."""
        _check(lib().psa_edge_take(self._h, right_ptr, right_done_ptr, r0, rows, packed_ptr, stream))

    def final(self, stream=0):
        t = (_i32 * 3)()
        _check(lib().psa_stripe_final(self._h, stream, t))
        return tuple(t)

    def traceback(self, i, j, k, ai, bj, length, stream=0):
        """Append reversed columns to numpy int32 arrays ai/bj from `length`; returns (i, j, k, length)."""
        ci, cj, ck, cl = _i64(i), _i64(j), C.c_int(k), _i64(length)
        _check(lib().psa_stripe_traceback(self._h, C.byref(ci), C.byref(cj), C.byref(ck), _ptr(ai), _ptr(bj),
                                          ai.size, C.byref(cl), stream))
        return ci.value, cj.value, ck.value, cl.value

    def last_fill_ms(self):
        v = C.c_float()
        _check(lib().psa_stripe_last_fill_ms(self._h, C.byref(v)))
        return v.value

    def last_traceback_ms(self):
        v = C.c_float()
        _check(lib().psa_stripe_last_traceback_ms(self._h, C.byref(v)))
        return v.value

    def code_bytes(self):
        return lib().psa_stripe_code_bytes(self._h)
