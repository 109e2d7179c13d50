"""Column-stripe pipeline over several GPUs (one process per GPU, torch.distributed).

This is the reference's own decomposition (simple.cpp:90-101, 261-277) lifted from threads to
GPUs: rank g owns columns (c0, c1] of B with block = m // G and the last rank taking the
remainder; rows are processed in chunks; for every chunk rank g receives the T1,T2,T3 triple of
column c0 from rank g-1 (the mailbox of simple.cpp:136-154), fills its stripe, and sends the
triple of column c1 to rank g+1 (simple.cpp:215-235).  The exchange is a chain of point-to-point
messages (RCCL send/recv over xGMI with backend "nccl"), never a collective: 12 bytes per row
and boundary.  The traceback walks right to left across the ranks, handing (i, j, k) over.

The compute engine is pluggable only so that the orchestration can be exercised on CPU with
gloo in the test-suite (tests inject an oracle-backed engine); the default and only product
engine is the HIP stripe object of the C ABI.
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

from .binding import PsaError, Result, Stripe, finish_alignment, pick


class HipStripeEngine:
    """Stripe on the current GPU through the C ABI (include/psa.h psa_stripe_*)."""

    def __init__(self, a, b, c0, c1, traceback=True):
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.stripe = Stripe(a, b, c0, c1, traceback=traceback)
        self.n, self.m = self.stripe.n, self.stripe.m
        # consecutive row chunks go to alternating streams: chunk c+1 is launched while chunk c runs and its
        # first strip picks up the rows chunk c's last strip publishes (granules in HBM), so the stripe's
        # dependency chain stays alive across chunk boundaries instead of draining at each of them
        self.streams = [torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev)]
        self.fill_stream, self._fill_stream_h = None, None
        self.take_stream, self._take_stream_h = None, None

    def chunk_stream(self, c):
        return torch.cuda.stream(self.streams[c % 2])

    def sync_chunks(self):
        for s in self.streams:
            s.synchronize()
        if self.fill_stream is not None:
            self.fill_stream.synchronize()

    def new_edge(self, rows):
        return torch.empty(3 * rows, dtype=torch.int32, device=self.dev)

    # ---- one launch over all rows, boundary chunks arriving and leaving by flags (psa_stripe_fill_flagged) ----
    flagged = True

    def fill_flagged(self, left, left_ready, right, right_done):
        """Queue the whole stripe's fill on a stream of its own; left/right: int32[3*n] device tensors or None, the
        flag tensors int32[(n+63)//64], zeroed."""
        if self.fill_stream is None:
            # A stream of the lowest priority: the runtime keeps a pool of hardware queues per priority, so the fill
            # never sits in a hardware queue AHEAD of a kernel it waits for (the posts, the collective library's
            # receives: all on normal-priority streams), and those kernels win where they compete for a CU.
            from .binding import stream_create
            self._fill_stream_h = stream_create(-1, int(os.environ.get("PSA_FILL_RESERVED_CUS", "0")))
            self.fill_stream = torch.cuda.ExternalStream(self._fill_stream_h, device=self.dev)
        self.fill_stream.wait_stream(torch.cuda.current_stream())   # (the flag tensors were zeroed on the current one)
        ptr = lambda t: t.data_ptr() if t is not None else None
        self.stripe.fill_flagged(ptr(left), ptr(left_ready), ptr(right), ptr(right_done), self.fill_stream.cuda_stream)

    def out_stream(self):
        """Where the takes (kernels that wait for a chunk of the right boundary) and the sends behind them are queued:
        the highest priority, again a pool of hardware queues of its own -- a take must not sit ahead of the next
        chunk's receive and post in a queue they share."""
        if self.take_stream is None:
            from .binding import stream_create
            self._take_stream_h = stream_create(+1)
            self.take_stream = torch.cuda.ExternalStream(self._take_stream_h, device=self.dev)
        return self.take_stream

    def edge_post(self, packed, r0, rows, left, left_ready):
        from .binding import edge_post
        edge_post(packed.data_ptr(), r0, rows, self.n, left.data_ptr(), left_ready.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)

    def edge_take(self, right, right_done, r0, rows, packed):
        self.stripe.edge_take(right.data_ptr(), right_done.data_ptr(), r0, rows, packed.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)

    def fill_rows(self, r0, r1, left, right):
        stream = torch.cuda.current_stream().cuda_stream
        self.stripe.fill_rows(r0, r1, left.data_ptr() if left is not None else None,
                              right.data_ptr() if right is not None else None, stream)

    def final(self):
        return self.stripe.final(torch.cuda.current_stream().cuda_stream)

    def traceback(self, i, j, k):
        cap = self.n + (self.stripe.c1 - self.stripe.c0) + 2
        ai = np.zeros(cap, np.int32)
        bj = np.zeros(cap, np.int32)
        i, j, k, ln = self.stripe.traceback(i, j, k, ai, bj, 0, torch.cuda.current_stream().cuda_stream)
        return i, j, k, ai[:ln], bj[:ln]

    def close(self):
        from .binding import stream_destroy
        if self._fill_stream_h is not None:
            self.fill_stream.synchronize()
            stream_destroy(self._fill_stream_h)
            self.fill_stream, self._fill_stream_h = None, None
        if self._take_stream_h is not None:
            self.take_stream.synchronize()
            stream_destroy(self._take_stream_h)
            self.take_stream, self._take_stream_h = None, None
        self.stripe.close()


def stripe_bounds(m, world, rank):
    """simple.cpp:90-99: block = m / p, stripe id starts at id*block, the last one takes the remainder."""
    block = m // world
    c0 = rank * block
    c1 = m if rank == world - 1 else c0 + block
    return c0, c1


def _p2p_device(t):
    """gloo moves host memory; nccl (RCCL) moves device memory."""
    return t.cpu() if (dist.get_backend() == "gloo" and t.is_cuda) else t


def _send(t, dst):
    dist.send(_p2p_device(t).contiguous(), dst)


def _recv(t, src):
    if dist.get_backend() == "gloo" and t.is_cuda:
        # (pinned + asynchronous: the copy stays on the current stream and waits for nothing else -- a flagged fill
        # may be on the GPU waiting for this very chunk)
        tmp = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        dist.recv(tmp, src)
        t.copy_(tmp, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    else:
        dist.recv(t, src)


def auto_chunk_rows(width):
    """Rows per hand-off for a stripe `width` columns wide.  A chain that wide keeps about width / 450 workgroups
    (of 256 rows) busy at a time, and two chunks are in flight: chunks that small would leave the GPU idle on a
    wide stripe (4096 rows = 16 workgroups), chunks that large lengthen the pipeline's fill across the ranks.
    Two chunks in flight stay below the 256 workgroups one GPU holds at a time (every strip of chunk c+1
    waits for strips of chunk c, which must be able to get a CU)."""
    groups = max(16, min(96, (width // 450 + 1) // 2 + 1))
    return 256 * groups


_warmed_up = set()


def _warm_up(prev_owner, next_owner):
    """The first message between two ranks sets the connection up (RCCL allocates and may synchronise the device):
    get that over with before a launch is on the GPU that waits for those messages.  Every rank that owns a stripe
    does this, whichever way it fills; gloo needs none."""
    key = (prev_owner, next_owner)
    if dist.get_backend() != "nccl" or key in _warmed_up:
        return
    tiny = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    if prev_owner is not None:
        dist.recv(tiny, prev_owner)
    if next_owner is not None:
        dist.send(tiny, next_owner)
    torch.cuda.synchronize()
    _warmed_up.add(key)


def _fill_flagged(eng, n, chunk_rows, prev_owner, next_owner):
    """The stripe's fill as ONE launch; the left boundary arrives from prev_owner and the right boundary leaves for
    next_owner in chunks of chunk_rows rows while it runs.  Receives + posts are queued on one stream, takes + sends
    on another, the fill on a third: a chunk is handed on as soon as its last strip has ended, whatever the fill
    launch is doing elsewhere."""
    dev = eng.dev
    strips = (n + 63) // 64
    left = left_ready = right = right_done = None
    if prev_owner is not None:
        left = torch.empty(3 * n, dtype=torch.int32, device=dev)
        left_ready = torch.zeros(strips, dtype=torch.int32, device=dev)
    if next_owner is not None:
        right = torch.empty(3 * n, dtype=torch.int32, device=dev)
        right_done = torch.zeros(strips, dtype=torch.int32, device=dev)
    eng.fill_flagged(left, left_ready, right, right_done)
    rows_max = min(chunk_rows, n)
    s_in, s_out = torch.cuda.Stream(device=dev), eng.out_stream()
    s_in.wait_stream(torch.cuda.current_stream())
    s_out.wait_stream(torch.cuda.current_stream())
    inbuf = [torch.empty(3 * rows_max, dtype=torch.int32, device=dev) for _ in range(2)] if left is not None else None
    outbuf = [torch.empty(3 * rows_max, dtype=torch.int32, device=dev) for _ in range(2)] if right is not None else None
    for c, r0 in enumerate(range(0, n, chunk_rows)):
        rows = min(n, r0 + chunk_rows) - r0
        if left is not None:
            with torch.cuda.stream(s_in):
                buf = inbuf[c % 2][:3 * rows]
                _recv(buf, prev_owner)
                eng.edge_post(buf, r0, rows, left, left_ready)
        if right is not None:
            with torch.cuda.stream(s_out):
                buf = outbuf[c % 2][:3 * rows]
                eng.edge_take(right, right_done, r0, rows, buf)
                _send(buf, next_owner)
    s_in.synchronize()
    s_out.synchronize()
    eng.sync_chunks()
    for t in (left, left_ready, right, right_done):   # (used on streams other than the one they were made on)
        if t is not None:
            t.record_stream(eng.fill_stream)


def align_striped(a, b, chunk_rows=None, traceback=True, engine_cls=HipStripeEngine, keep_engine=None, flagged=None):
    """Align A against B with B cut into world_size column stripes, one per rank.

    Every rank calls this with the same a, b.  Returns Result(score, ai, bj) on every rank
    (ai, bj are empty arrays when traceback is False).  chunk_rows (multiple of 64; None or 0: chosen
    from the stripe width, auto_chunk_rows) is the hand-off granularity between neighbouring ranks.
    flagged=False: one fill launch per chunk (psa_stripe_fill_rows) instead of one launch over all rows whose
    boundaries arrive and leave by flags (psa_stripe_fill_flagged, the default where the engine has it); there two
    chunks are in flight at a time and must fit on the GPU together, so chunk_rows is capped at 24 576 (96 workgroups).
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    a = np.ascontiguousarray(np.frombuffer(a, np.uint8) if isinstance(a, (bytes, bytearray)) else a, np.uint8)
    b = np.ascontiguousarray(np.frombuffer(b, np.uint8) if isinstance(b, (bytes, bytearray)) else b, np.uint8)
    n, m = int(a.size), int(b.size)
    if not chunk_rows:
        chunk_rows = auto_chunk_rows(max(1, m // world))
    assert chunk_rows >= 64 and chunk_rows % 64 == 0
    # (launch-per-chunk path: two chunks are in flight, and every strip of chunk c+1 waits for strips of chunk c, which
    # must be able to get a CU -- one workgroup of 256 rows per CU, 256 CUs: chunks beyond 96 workgroups are cut down)
    max_chunk_rows = 256 * 96
    empty = np.zeros(0, np.int32)
    if n == 0 or m == 0:
        # no cell to fill: the final pick reads the boundary (simple.cpp:246 after an empty loop)
        t = (-1, 0, -(2 + m)) if n == 0 else (-1, -(2 + n), -1)
        return Result(pick(*t)[1], empty, empty)

    c0, c1 = stripe_bounds(m, world, rank)
    width = c1 - c0
    # ranks whose stripe is empty (m < world) only forward the boundary
    owners = [g for g in range(world) if stripe_bounds(m, world, g)[1] > stripe_bounds(m, world, g)[0]]
    prev_owner = max([g for g in owners if g < rank], default=None)
    next_owner = min([g for g in owners if g > rank], default=None)
    last_owner = owners[-1]

    eng = None
    if width > 0:
        eng = keep_engine if keep_engine is not None else engine_cls(a, b, c0, c1, traceback)
        _warm_up(prev_owner, next_owner)
        if flagged is None:
            # ranks that share one GPU (rehearsals) would hold each other's CUs with waiting strips
            flagged = int(os.environ.get("LOCAL_WORLD_SIZE", world)) <= max(1, torch.cuda.device_count())
        if getattr(eng, "flagged", False) and flagged:
            try:
                _fill_flagged(eng, n, chunk_rows, prev_owner, next_owner)
                chunk_rows = None
            except PsaError as e:
                if e.code != -6:    # PSA_E_STATE before anything was queued: this stripe needs the generic kernel
                    raise
    if eng is not None and chunk_rows is not None:
        # (engines without the flagged fill, or flagged=False: one launch per chunk)
        chunk_rows = min(chunk_rows, max_chunk_rows)
        # two chunks are in flight at a time (alternating streams): two boundary buffers each way
        left = [eng.new_edge(min(chunk_rows, n)) for _ in range(2)] if prev_owner is not None else None
        right = [eng.new_edge(min(chunk_rows, n)) for _ in range(2)] if next_owner is not None else None
        chunk_stream = getattr(eng, "chunk_stream", None)
        for c, r0 in enumerate(range(0, n, chunk_rows)):
            r1 = min(n, r0 + chunk_rows)
            rows = r1 - r0
            lv = left[c % 2][:3 * rows] if left is not None else None
            rv = right[c % 2][:3 * rows] if right is not None else None
            ctx = chunk_stream(c) if chunk_stream is not None else contextlib.nullcontext()
            with ctx:   # recv, fill and send of one chunk are ordered on its stream; the next chunk's run beside them
                if lv is not None:
                    _recv(lv, prev_owner)
                eng.fill_rows(r0, r1, lv, rv)
                if rv is not None:
                    _send(rv, next_owner)
        if hasattr(eng, "sync_chunks"):
            eng.sync_chunks()

    # ---- final pick on the last stripe, then the traceback chain right -> left ----
    if dist.get_backend() == "nccl":
        comm_dev = torch.device("cuda", torch.cuda.current_device())
    else:
        comm_dev = torch.device("cpu")
    # Every rank looks at its own fill (a wait that expired raises the stripe's abort flag: psa_stripe_final reports
    # it) and the ranks agree: either all go on or all raise, so that nobody is left alone in a collective.
    head = torch.zeros(5, dtype=torch.int64, device=comm_dev)  # score, i, j, k, ranks whose fill gave up
    failed, why, t = 0, "", None
    if eng is not None:
        try:
            t = eng.final()
        except PsaError as e:
            failed, why = 1, str(e)
    if rank == last_owner and not failed:
        kstar, score = pick(*t)
        head[0], head[1], head[2], head[3] = int(score), n, m, kstar
    head[4] = failed
    fails = torch.tensor([failed], dtype=torch.int64, device=comm_dev)
    dist.all_reduce(fails)
    if int(fails.item()):
        if eng is not None and keep_engine is None:
            eng.close()
        raise PsaError(-5, "align_striped: the fill gave up on %d rank(s)%s" % (int(fails.item()), (" (this one: %s)" % why) if failed else ""))
    dist.broadcast(head, last_owner)
    score = int(head[0].item())
    if not traceback:
        return Result(score, empty, empty)

    pieces_ai, pieces_bj = empty, empty
    state = torch.zeros(4, dtype=torch.int64, device=comm_dev)  # i, j, k, finished
    if width > 0:
        if rank == last_owner:
            i, j, k, fin = n, m, int(head[3].item()), 0
        else:
            dist.recv(state, next_owner)
            i, j, k, fin = (int(x) for x in state.tolist())
        if not fin:
            i, j, k, pieces_ai, pieces_bj = eng.traceback(i, j, k)
            fin = int(i == 0 or j == 0)
        if prev_owner is not None:
            state[0], state[1], state[2], state[3] = i, j, k, fin
            dist.send(state, prev_owner)
    # gather the reversed pieces on rank 0: walk order is last stripe first
    lens = torch.zeros(world, dtype=torch.int64, device=comm_dev)
    mine = torch.tensor([len(pieces_ai)], dtype=torch.int64, device=comm_dev)
    lens_list = [torch.zeros(1, dtype=torch.int64, device=comm_dev) for _ in range(world)]
    dist.all_gather(lens_list, mine)
    lens = [int(x.item()) for x in lens_list]
    total = sum(lens)
    packed = torch.from_numpy(np.concatenate([pieces_ai, pieces_bj]).astype(np.int32)).to(comm_dev)
    if rank == 0:
        ai_rev = np.zeros(total, np.int32)
        bj_rev = np.zeros(total, np.int32)
        pos = 0
        for g in range(world - 1, -1, -1):
            if lens[g] == 0:
                continue
            if g == 0:
                buf = packed
            else:
                buf = torch.empty(2 * lens[g], dtype=torch.int32, device=comm_dev)
                dist.recv(buf, g)
            h = buf.cpu().numpy()
            ai_rev[pos:pos + lens[g]] = h[:lens[g]]
            bj_rev[pos:pos + lens[g]] = h[lens[g]:]
            pos += lens[g]
        ai, bj = finish_alignment(ai_rev, bj_rev, n, m)
        out_len = torch.tensor([len(ai)], dtype=torch.int64, device=comm_dev)
    else:
        if lens[rank] > 0:
            dist.send(packed, 0)
        out_len = torch.zeros(1, dtype=torch.int64, device=comm_dev)
    dist.broadcast(out_len, 0)
    ln = int(out_len.item())
    res = torch.zeros(2 * ln, dtype=torch.int32, device=comm_dev)
    if rank == 0:
        res = torch.from_numpy(np.concatenate([ai, bj]).astype(np.int32)).to(comm_dev)
    dist.broadcast(res, 0)
    h = res.cpu().numpy()
    if eng is not None and keep_engine is None:
        eng.close()
    return Result(score, h[:ln].copy(), h[ln:].copy())
