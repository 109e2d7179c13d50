#!/usr/bin/env python3
"""bench.py -- GCUPS of the alignment hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete alignment of the workload pair: DP fill with packed winner codes
written to HBM, final pick, traceback walk, prefix fix-up (what run() does in simple.cpp:261-294).
Inputs are resident in HBM before the timed region.  N = 1: the synthetic 2^16 x 2^16 protein
pair of BASELINE config 4 on one GPU.  N > 1 (launched by torch.distributed.run, one rank per
GPU): every rank does the same with its own pairs, no data-path communication (weak scaling; the
"batched many-pair" sharding of the north star); --mode stripes instead cuts ONE pair into N
column stripes, boundary columns streamed g -> g+1 with RCCL send/recv (bench_multi.py; its
dependency chain is n + m steps long on any number of GPUs, so it buys capacity, not time, at
2^16).  One pair alone cannot fill the GPU (its anti-diagonal
dependency chain is n + m steps long whatever the hardware), so the job keeps --in-flight
(default 6) independent alignments going at a time, each on its own stream with its own code
buffer; "value" is the job's throughput, "one_pair_alone" the latency of a single alignment, and
the roofline object is the fill kernel measured alone.  Rank 0 prints ONE JSON line.

Extra keys beside the contract: roofline (int32 VALU), cpu_baseline (the oracle's
multi-threaded port timed on host cores, rank 0, N = 1 only), and secondary measurements
(score-only fill, titin pair) under "extra".
"""
import argparse
import json
import os
import sys
import time

# Every alignment in flight has a stream of its own; with HIP's default of 4 hardware queues per process
# streams share queues, and one stream's small kernels wait behind another stream's fill.
_HWQ = min(32, int(os.environ.get("PSA_BENCH_HW_QUEUES", "8")))
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < _HWQ:
    os.environ["GPU_MAX_HW_QUEUES"] = str(_HWQ)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VALU_PEAK_TOPS = 39.3216      # 256 CU x 64 int32 lanes/clk x 2.4 GHz (tools/valu_peak measures 38.8)
OPS_PER_CELL_CODES = 25       # algorithmic int32 VALU ops per cell with winner codes
OPS_PER_CELL_SCORE = 10       # score only
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--log2n", type=int, default=16, help="synthetic pair is 2^k x 2^k")
    ap.add_argument("--chunk-rows", type=int, default=4096, help="rows per inter-GPU hand-off (N > 1)")
    ap.add_argument("--in-flight", type=int, default=6,
                    help="independent alignments in flight per GPU (at most 8: one hardware queue each)")
    ap.add_argument("--mode", choices=["pairs", "stripes"], default="pairs",
                    help="N > 1: independent pairs per GPU (default) or one pair in N column stripes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    return ap.parse_args()


def cpu_baseline(a, b, log2s=16):
    """Oracle's striped multi-thread port (kind "port") on a bounded prefix sample of the workload."""
    import oracle_lib as O
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    s = min(len(a), len(b), 1 << log2s)
    t0 = time.perf_counter()
    sc, ai, bj = O.align(a[:s], b[:s], cores)
    dt = time.perf_counter() - t0
    return {"value": s * s / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "prefix %dx%d of the workload pair, fill + codes + traceback, %d threads (oracle/psa_oracle.c "
                      "psa_oracle_align_mt), %.2f s" % (s, s, cores, dt),
            "score": int(sc)}


def pmc_traffic(kernel_substr, grid):
    """HBM bytes per launch of the fill kernel from the committed rocprofv3 --pmc passes (WRITE_SIZE and
    FETCH_SIZE, one pass each, summarised by tools/rocpd_summary.py into profiles/).  FETCH_SIZE is doubled
    (gfx950 correction, MI355X_MICROARCH.md, HBM section).  None when the summaries are absent."""
    import csv
    tot, src = 0.0, []
    for name, scale in (("pmc_w_bench_2p16.csv", 1024.0), ("pmc_f_bench_2p16.csv", 2048.0)):
        path = os.path.join(ROOT, "profiles", "r1", name)
        if not os.path.exists(path):
            return None, None
        hit = False
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row["kernel"] and int(row["grid_size"]) == grid and row["counter"] in ("WRITE_SIZE", "FETCH_SIZE"):
                    tot += float(row["avg_value"]) * scale
                    hit = True
        if not hit:
            return None, None
        src.append("profiles/r1/" + name)
    return tot, src


def pmc_valu_per_cell(kernel_substr, grid, cells):
    """Executed VALU instructions per cell (SQ_INSTS_VALU x 64 lanes / cells) from the committed --pmc pass."""
    import csv
    path = os.path.join(ROOT, "profiles", "r1", "pmc_sq_bench_2p16.csv")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["kernel"] and int(row["grid_size"]) == grid and row["counter"] == "SQ_INSTS_VALU":
                return float(row["avg_value"]) * 64.0 / cells
    return None


def single_gpu(args):
    import threading
    import numpy as np
    import torch
    import psa_amd
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    backend = os.environ.get("PSA_BENCH_BACKEND", "nccl")   # gloo: rehearsal of the N > 1 logic on one GPU
    torch.cuda.set_device(local)
    psa_amd.set_device(local)
    dist = None
    if world > 1:
        # independent alignments per rank: no data-path communication, only the timing barrier
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)          # the first collective builds the communicator: fail here, not mid-run
                torch.cuda.synchronize()
            except Exception as e:              # the barrier is all this mode needs: fall back to gloo
                print("bench: nccl unavailable (%s), timing barrier over gloo" % e, file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
        if backend != "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world)
    k = args.log2n
    n = m = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n)
    b = psa_amd.synth_protein(0xB0000000 + k, m)
    a_dev = torch.from_numpy(a).cuda()
    b_dev = torch.from_numpy(b).cuda()
    # more streams than hardware queues makes the GPU's scheduler time-slice the queues, and a fill that
    # is switched out with its 96 KB of LDS per workgroup takes seconds (measured: 10 in flight = 1 s each)
    P = min(8, max(1, args.in_flight))
    stripes = [psa_amd.Stripe(a, b, 0, m, traceback=True) for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    cap = 2 * (n + m) + 2

    def one_alignment(st, stream, ai, bj):
        """run() of the reference for one pair whose residues are already in HBM."""
        st.set_sequences_dev(a_dev.data_ptr(), b_dev.data_ptr(), stream)
        st.fill_rows(0, n, None, None, stream)
        t = st.final(stream)
        kk, sc = psa_amd.pick(*t)
        _, _, _, ln = st.traceback(n, m, kk, ai, bj, 0, stream)
        fa, fb = psa_amd.finish_alignment(ai[:ln], bj[:ln], n, m)
        return sc, len(fa)

    def run_steps(count, stagger_s=0.0):
        """`count` independent alignments, up to P of them in flight: one host thread, stripe object and
        stream per slot; a slot takes the next alignment as soon as it has finished its last one.  Slot i
        starts i * stagger_s late, so that the slots do not march in step (all filling, then all walking)."""
        lock = threading.Lock()
        state = {"next": 0}
        res, fill_ms, tb_ms = [], [], []

        def worker(slot):
            torch.cuda.set_device(local)   # (the current device is per thread; the library also switches to the stripe's)
            psa_amd.set_device(local)
            st, stream = stripes[slot], streams[slot].cuda_stream
            ai = np.zeros(cap, np.int32)
            bj = np.zeros(cap, np.int32)
            if slot and stagger_s > 0:
                time.sleep(slot * stagger_s)
            while True:
                with lock:
                    if state["next"] >= count:
                        return
                    state["next"] += 1
                r = one_alignment(st, stream, ai, bj)
                with lock:
                    res.append(r); fill_ms.append(st.last_fill_ms()); tb_ms.append(st.last_traceback_ms())

        th = [threading.Thread(target=worker, args=(i,)) for i in range(min(P, max(count, 1)))]
        [x.start() for x in th]
        [x.join() for x in th]
        return res, fill_ms, tb_ms

    # first use of every slot's stream on its own: creating a hardware queue (and, under rocprofv3, its
    # interception) while fills of other slots are running was seen to freeze those fills for seconds
    _ai = np.zeros(cap, np.int32); _bj = np.zeros(cap, np.int32)
    for st_, sm_ in zip(stripes, streams):
        one_alignment(st_, sm_.cuda_stream, _ai, _bj)
    torch.cuda.synchronize()
    run_steps(args.warmup, 0.004)
    torch.cuda.synchronize()
    stagger = 0.004
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, fill_ovl, tb_ovl = run_steps(args.steps, stagger)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        dtt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(dtt, op=dist.ReduceOp.MAX)
        dt = float(dtt.item())
    assert len(res) == args.steps and len(set(res)) == 1, res
    sc, ln = res[0]
    cells = float(n) * m * world          # per step: one alignment on every rank
    ms_step = dt / args.steps * 1e3
    if rank != 0:
        for s_ in stripes:
            s_.close()
        dist.barrier()                    # rank 0 is still measuring; leave together
        dist.destroy_process_group()
        return None

    # the same alignment alone on the GPU: latency of one pair, and the fill kernel's own duration
    st = stripes[0]
    ai0 = np.zeros(cap, np.int32); bj0 = np.zeros(cap, np.int32)
    fill_ms, tb_ms = [], []
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    iso_reps = 3
    for _ in range(iso_reps):
        r = one_alignment(st, streams[0].cuda_stream, ai0, bj0)
        assert r == (sc, ln)
        fill_ms.append(st.last_fill_ms()); tb_ms.append(st.last_traceback_ms())
    torch.cuda.synchronize()
    lat_ms = (time.perf_counter() - t1) / iso_reps * 1e3
    fill = sum(fill_ms) / len(fill_ms)
    fill_o = sum(fill_ovl) / len(fill_ovl)
    cells1 = float(n) * m
    achieved = cells1 * OPS_PER_CELL_CODES / (fill * 1e-3) / 1e12
    aggregate = cells * OPS_PER_CELL_CODES * args.steps / dt / 1e12 / world   # per GPU
    code_bytes = st.code_bytes()
    traffic, traffic_src = pmc_traffic("fill_strips_v2<true>", 65536) if k == 16 else (None, None)
    out = {
        "metric": "GCUPS (cell updates/s), full alignment (fill + traceback) of 2^%d x 2^%d pairs" % (k, k),
        "value": cells / (dt / args.steps) / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "synthetic iid 20-letter protein pair 2^%d x 2^%d (mt19937_64 seeds 0xA/0xB0000000+%d), "
                               "fill with packed winner codes in HBM + traceback walk + prefix fix-up; one step = one "
                               "complete alignment per GPU, up to %d independent alignments in flight on each (a "
                               "stream and a code buffer each); no inter-GPU communication" % (k, k, k, P),
                   "n": n, "m": m, "score": int(sc), "alignment_len": int(ln), "stripes": 1, "in_flight": P,
                   "parallelism": "independent pairs per GPU" if world > 1 else "one GPU"},
        "roofline": {"bound": "valu", "kernel": "fill_strips_v2<true> (fill with winner codes)", "achieved": achieved,
                     "peak": VALU_PEAK_TOPS, "unit": "Tint32op/s", "frac": achieved / VALU_PEAK_TOPS,
                     "ops_per_cell": OPS_PER_CELL_CODES,
                     "valu_insts_per_cell_executed": pmc_valu_per_cell("fill_strips_v2<true>", 65536, cells1) if k == 16 else None,
                     "kernel_ms": fill,
                     "kernel_ms_note": "HIP events around the launch, alignment alone on the GPU (%d launches after the "
                                       "timed region); launches of the timed region overlap each other" % iso_reps,
                     "launch_ms_overlapped": fill_o, "fill_gcups": cells1 / (fill * 1e-3) / 1e9,
                     "aggregate_achieved": aggregate, "aggregate_frac": aggregate / VALU_PEAK_TOPS,
                     "structural_ceiling_one_pair": min(1.0, cells1 / ((n + m) * 65536.0)),
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (separate --pmc passes)",
                     "traffic_source": traffic_src,
                     "code_stream": {"bytes": code_bytes, "GBps": code_bytes / (fill * 1e-3) / 1e9,
                                     "frac_of_hbm_peak": code_bytes / (fill * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        "one_pair_alone": {"ms": lat_ms, "gcups": cells1 / (lat_ms * 1e-3) / 1e9, "fill_ms": fill,
                           "traceback_ms": sum(tb_ms) / len(tb_ms)},
        "traceback_ms": sum(tb_ms) / len(tb_ms),
    }
    for s_ in stripes[1:]:
        s_.close()
    stream = streams[0].cuda_stream
    st.close()
    extra = {}
    if not args.no_extra:
        # score-only fill of the same pair
        so = psa_amd.Stripe(a, b, 0, m, traceback=False)
        for _ in range(2):
            so.fill_rows(0, n, None, None, stream)
        ts = []
        for _ in range(3):
            so.fill_rows(0, n, None, None, stream)
            ts.append(so.last_fill_ms())
        t = so.final(stream)
        so.close()
        ms = sum(ts) / len(ts)
        extra["score_only_2p%d" % k] = {"gcups": cells1 / (ms * 1e-3) / 1e9, "kernel_ms": ms, "score": int(max(t)),
                                        "valu_frac": cells1 * OPS_PER_CELL_SCORE / (ms * 1e-3) / 1e12 / VALU_PEAK_TOPS}
        # titin pair (BASELINE config 3), full alignment
        seqdir = os.path.join(ROOT, "tests", "golden", "sequences")
        ta = psa_amd.read_fasta(os.path.join(seqdir, "titin_homo(50).fasta"))
        tb = psa_amd.read_fasta(os.path.join(seqdir, "titin_mus.fasta"))
        psa_amd.align(ta, tb, 1)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            res = psa_amd.align(ta, tb, 1)
        dtt = (time.perf_counter() - t0) / reps
        extra["titin_pair"] = {"gcups_end_to_end_host_buffers": len(ta) * len(tb) / dtt / 1e9, "ms": dtt * 1e3,
                               "score": int(res.score), "alignment_len": int(len(res.ai)),
                               "note": "psa_align from host buffers: includes allocation, H2D and D2H"}
        # BASELINE config 5 shape: a block of 512-residue proteins, all queries x all targets, score only
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_batch
        bb = bench_batch.measure(nq=2048, nt=256, L=512, reps=3, check=4)
        bb["valu_frac"] = bb["gcups"] * OPS_PER_CELL_SCORE / 1e3 / VALU_PEAK_TOPS
        extra["batch_512aa_score_only"] = bb
    out["extra"] = extra
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(a, b)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or world > 1) and args.mode == "stripes":
        from bench_multi import multi_gpu   # one pair cut into column stripes (strong scaling)
        out = multi_gpu(args)
    else:
        out = single_gpu(args)              # independent alignments on every GPU (weak scaling)
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
