#!/usr/bin/env python3
"""bench.py -- GCUPS of the alignment hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one BATCH of the synthetic input: --slots (20) DISTINCT pairs of the
workload size, each aligned completely -- DP fill with winner codes written to HBM, final pick, traceback walk,
prefix fix-up (what run() does in simple.cpp:261-294, once per pair) --, the batch being ONE fill launch and ONE launch
of the walks.  (Until round 3 a step was one alignment; with the driver's --steps 20 the timed region was then a single
launch, its ramp, drain and last walk in full view and no steady state: now it is 20 launches, 400 alignments.)  Inputs are
resident in HBM before the timed region.  N = 1: the synthetic 2^16 x 2^16 protein pair of BASELINE
config 4 on one GPU.  One pair alone cannot fill the GPU (its anti-diagonal dependency chain is
n + m steps long whatever the hardware), so the job is what psa_align_many does: ONE host thread
drives up to three psa_many engines that take turns; every batch of --slots pairs (20: what a launch's
argument block holds) is ONE fill launch (the pairs' strips interleaved) and ONE launch of their walks,
and the walks and copies of one batch run beside the fill of the next.  "value" is the job's throughput, "one_pair_alone" the latency of a
single alignment, and the roofline object is the dominant kernel -- the fill launch -- timed live
with HIP events on the engine's stream inside the timed region.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank does the same with its own
pairs, no data-path communication (weak scaling; the "batched many-pair" sharding of the north star);
the same line then also carries the strong-scaling figures of ONE pair cut into N column stripes
(boundary columns streamed g -> g+1 with RCCL send/recv) under "stripes".  Rank 0 prints ONE JSON line.

Extra keys beside the contract: roofline (int32 VALU), cpu_baseline (the oracle's
multi-threaded port timed on host cores, rank 0, N = 1 only), and secondary measurements
(score-only fill, titin pair, batch kernel) under "extra".
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VALU_PEAK_TOPS = 39.3216      # 256 CU x 64 int32 lanes/clk x 2.4 GHz (tools/valu_peak measures 38.8)
# Algorithmic int32 VALU operations per cell.  SURVEY 8d priced the recurrence at 10 (score only) and
# 25 (with winner bits: three compares and their packing per aggregate); with the one-byte winner code the
# recurrence is 8 operations (two lane shifts, substitution add, two gap adds, max3, two max) and the code 3.
# At the SURVEY's 25 the 8-pair launch would read 1.2 "of peak": the figures below use 11 and 8, and the
# SURVEY's are kept beside them (frac_at_survey_ops) for comparison with round 1.
OPS_PER_CELL_CODES = 11
OPS_PER_CELL_SCORE = 8
OPS_PER_CELL_CODES_PK = 13    # the packed fill (psa_fill_pk.hip.h): 13 instructions per step of TWO cells = 13 sixteen-bit lane operations
                              # per cell (2 DPP moves, 1 perm for the two substitution bytes, 3 adds, 4 max, 2 mad + 1 perm for the
                              # code bytes), priced against PK_PEAK_TOPS
OPS_PER_CELL_BATCH = 10       # batch kernel, int32 path: compares residues itself (v_cmp + v_cndmask) instead of a table byte
OPS_PER_CELL_BATCH_PK = 12    # packed path: 12 instructions per step of TWO cells = 12 sixteen-bit lane operations per cell,
MAX_SLOTS = 20   # pairs per fill launch: what the kernel's 4 KB argument block holds (psa_args.hip.h, kMaxJobs)
PK_PEAK_TOPS = 2 * VALU_PEAK_TOPS   # against a peak of two 16-bit lane operations per lane and clock (v_pk_* are full rate: tools/pk_rate.hip)
SURVEY_OPS_CODES, SURVEY_OPS_SCORE = 25, 10
HBM_PEAK_GBS = 8000.0
PROFILE_DIR = os.path.join("profiles", "r3")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12, help="batches (fill launches of --slots alignments each) in the timed region")
    ap.add_argument("--warmup", type=int, default=3, help="untimed batches before it (at least one per engine is run)")
    ap.add_argument("--log2n", type=int, default=16, help="synthetic pair is 2^k x 2^k")
    ap.add_argument("--chunk-rows", type=int, default=0, help="rows per inter-GPU hand-off (N > 1, stripes); 0 = from the stripe width")
    ap.add_argument("--slots", type=int, default=MAX_SLOTS, help="pairs per launch (1..%d); up to three engines take turns" % MAX_SLOTS)
    ap.add_argument("--engines", type=int, default=3, help="psa_many engines taking turns (2 or 3)")
    ap.add_argument("--one-process-leg", type=int, default=0, help=argparse.SUPPRESS)   # (internal: the child of the N > 1 run)
    ap.add_argument("--leg-gold-score", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--serial-fills", action="store_true",
                    help="each fill launch alone on the chip (the next engine's fill waits for this one's end); default: "
                         "the next launch is queued at once and fills the chip as this one drains")
    ap.add_argument("--mode", choices=["pairs", "stripes"], default="pairs",
                    help="N > 1: independent pairs per GPU plus the stripes figures (default), or only the stripes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-stripes", action="store_true", help="N > 1: skip the strong-scaling stripes figures")
    return ap.parse_args()


def cpu_threads():
    """Threads for the CPU baseline.  The north star asks for p = the host's core count, stated: os.cpu_count() is
    reported as host_cpu_count; what is USED is capped by the box's CPU share for one GPU (the harness sets
    OMP_NUM_THREADS to it, 16 on the one-GPU boxes of this pool, and kills runs that use more)."""
    host = os.cpu_count() or 1
    share = int(os.environ.get("PSA_BENCH_CPU_THREADS", os.environ.get("OMP_NUM_THREADS", "0")) or 0)
    return (min(host, share) if share > 0 else host), host


def cpu_baseline(a, b, log2s=16):
    """Oracle's striped multi-thread port (kind "port") on a bounded sample of the workload: the whole 2^16 pair when
    that is the workload (a few seconds), and the whole titin pair (BASELINE config 3)."""
    import numpy as np
    import oracle_lib as O
    import psa_amd
    cores, host = cpu_threads()
    s = min(len(a), len(b), 1 << log2s)
    t0 = time.perf_counter()
    sc, ai, bj = O.align(a[:s], b[:s], cores)
    dt = time.perf_counter() - t0
    out = {"value": s * s / dt / 1e9, "unit": "GCUPS", "cores": cores, "host_cpu_count": host, "kind": "port",
           "cores_note": "threads used = min(os.cpu_count(), the box's CPU share for one GPU: OMP_NUM_THREADS / "
                         "PSA_BENCH_CPU_THREADS); the reference itself cannot run this size (48 n m bytes, n^3 time: "
                         "BASELINE.md 3) -- the port keeps its thread decomposition (column stripes, per-row hand-off) "
                         "with one-byte codes instead of per-cell vectors",
           "sample": "prefix %dx%d of the workload pair, fill + codes + traceback, %d threads (oracle/psa_oracle.c "
                     "psa_oracle_align_mt), %.2f s" % (s, s, cores, dt),
           "score": int(sc)}
    out["reference_binary"] = reference_binary_baseline(cores)
    seqdir = os.path.join(ROOT, "tests", "golden", "sequences")
    ta = np.frombuffer(psa_amd.read_fasta(os.path.join(seqdir, "titin_homo(50).fasta")), np.uint8)
    tb = np.frombuffer(psa_amd.read_fasta(os.path.join(seqdir, "titin_mus.fasta")), np.uint8)
    t0 = time.perf_counter()
    tsc, tai, _ = O.align(ta, tb, cores)
    dtt = time.perf_counter() - t0
    out["titin_pair"] = {"value": float(len(ta)) * len(tb) / dtt / 1e9, "unit": "GCUPS", "cores": cores, "seconds": dtt,
                         "score": int(tsc), "alignment_len": int(len(tai)),
                         "sample": "the whole titin pair %d x %d, fill + codes + traceback" % (len(ta), len(tb))}
    return out


def reference_binary_baseline(cores):
    """The reference's own ./simple (oracle/_ref/simple: the reference's simple.cpp compiled where it lies by
    oracle/Makefile, g++ -O2 -pthread; the binary travels with the tree, its sources do not) with p = the cores used,
    on the pair its own timings.csv / report Table 1 were taken on (insulin-like, 1367 x 1373): the only size class it
    can run -- it keeps an alignment vector in every cell, about 48 n m bytes and n^3 time."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "simple")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/simple is not built (the reference checkout was absent at build time)"}
    cwd = os.path.join(ROOT, "tests", "golden")
    a, b = "homo_insulinlike.fasta", "mouse_insulinlike.fasta"
    try:
        t0 = time.perf_counter()
        r = subprocess.run([exe, a, b, str(cores)], cwd=cwd, capture_output=True, timeout=120)
        dt = time.perf_counter() - t0
        first = r.stdout.split(b"\n")[0].decode()
        n, m = 1367, 1373
        return {"kind": "reference", "pair": "%s x %s (%d x %d)" % (a, b, n, m), "p": cores, "seconds": dt,
                "value": n * m / dt / 1e9, "unit": "GCUPS", "mcups": n * m / dt / 1e6, "stdout_first_line": first,
                "score_ok": first == "Score: 2616", "returncode": r.returncode}
    except Exception as e:   # (a baseline leg: its failure is reported, the line stands)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def pmc_rows(name):
    import csv
    path = os.path.join(ROOT, PROFILE_DIR, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return list(csv.DictReader(f))


def pmc_traffic(kernel_substr, grid):
    """HBM bytes per launch of the fill kernel from the committed rocprofv3 --pmc passes (WRITE_SIZE and
    FETCH_SIZE, one pass each, summarised by tools/rocpd_summary.py into profiles/).  FETCH_SIZE is doubled
    (gfx950 correction, MI355X_MICROARCH.md, HBM section).  None when the summaries are absent."""
    tot, src = 0.0, []
    for name, scale in (("pmc_w_bench.csv", 1024.0), ("pmc_f_bench.csv", 2048.0)):
        rows = pmc_rows(name)
        if rows is None:
            return None, None
        hit = False
        for row in rows:
            if kernel_substr in row["kernel"] and int(row["grid_size"]) == grid and row["counter"] in ("WRITE_SIZE", "FETCH_SIZE"):
                tot += float(row["avg_value"]) * scale
                hit = True
        if not hit:
            return None, None
        src.append(PROFILE_DIR + "/" + name)
    return tot, src


def pmc_valu_per_cell(kernel_substr, grid, cells):
    """Executed VALU instructions per cell (SQ_INSTS_VALU x 64 lanes / cells) from the committed --pmc pass."""
    rows = pmc_rows("pmc_sq_bench.csv")
    if rows is None:
        return None
    for row in rows:
        if kernel_substr in row["kernel"] and int(row["grid_size"]) == grid and row["counter"] == "SQ_INSTS_VALU":
            return float(row["avg_value"]) * 64.0 / cells
    return None


def init_dist(local):
    """torch.distributed for N > 1: RCCL ("nccl") if it comes up, else gloo (rehearsals on one GPU)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("PSA_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            probe = torch.zeros(1, device="cuda")
            dist.all_reduce(probe)          # the first collective builds the communicator: fail here, not mid-run
            torch.cuda.synchronize()
        except Exception as e:              # the timing barrier is all the pairs mode needs: fall back to gloo
            print("bench: nccl unavailable (%s), falling back to gloo" % e, file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            backend = "gloo"
    if backend != "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, backend


def slot_pairs(psa_amd, torch, k, S):
    """The S DISTINCT pairs the slots of every launch hold: slot s = synthetic 2^k x 2^k pair of seeds
    0xA0000000+k+0x100*s / 0xB0000000+k+0x100*s (slot 0 is the pair BASELINE config 4 names), resident in HBM."""
    n = 1 << k
    out = []
    for sl in range(S):
        a = psa_amd.synth_protein(0xA0000000 + k + 0x100 * sl, n)
        b = psa_amd.synth_protein(0xB0000000 + k + 0x100 * sl, n)
        out.append((a, b, torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()))
    torch.cuda.synchronize()
    return out


def slot_goldens(k):
    """(score, alignment length) per slot from the committed restatement-only goldens (tests/golden/oracle_only.json,
    made by make_oracle_golden.py --bench-pairs): only the 2^16 pairs have them."""
    path = os.path.join(ROOT, "tests", "golden", "oracle_only.json")
    if k != 16 or not os.path.exists(path):
        return {}
    with open(path) as f:
        g = json.load(f)
    return {e["slot"]: (e["score"], e["alignment_len"]) for e in g.get("bench_pairs", []) if e["n"] == 1 << k and e["m"] == 1 << k}


def plan_batches(count, S):
    """`count` alignments as the fewest launches of at most S pairs, of equal size (a launch costs about 4 ms of ramp
    and drain beyond its pairs)."""
    if count <= 0:
        return []
    L = -(-count // S)
    base, rem = divmod(count, L)
    return [base + (1 if x < rem else 0) for x in range(L)]


def throughput_job(engines, pairs, n, m, batches, serial_fills, ref_event=None, fill_ends=None):
    """sum(batches) complete alignments through the engines, which take turns, ONE host thread; slot s of every launch
    holds pair s.  Returns (results [(slot, score, alignment length)], fill launch durations [(ms, pairs)], walk
    launch durations)."""
    S = engines[0].slots
    E = len(engines)
    res, fills, walks = [], [], []
    held = [0] * E

    def collect(x):
        eng = engines[x]
        eng.wait()
        for sl in range(held[x]):
            r = eng.result(sl, n, m)
            res.append((sl, r.score, len(r.ai)))
            for other in engines:   # (a fill that has ended meanwhile: its walks are queued now, not at its turn)
                if other is not eng:
                    other.poll()
        f, w = eng.last_ms()
        fills.append((f, held[x])); walks.append(w)
        if ref_event is not None:
            fill_ends.append(eng.fill_ms_since(ref_event) + (held[x],))
        held[x] = 0

    turn = 0
    for take in batches:
        eng = engines[turn]
        if held[turn]:
            collect(turn)
        for sl in range(S):
            if sl < take:
                eng.set_pair_dev(sl, pairs[sl][2].data_ptr(), n, pairs[sl][3].data_ptr(), m)
            else:
                eng.clear_slot(sl)
        eng.launch(engines[(turn - 1) % E] if serial_fills else None)
        held[turn] = take
        turn = (turn + 1) % E
    for x in range(E):
        if held[(turn + x) % E]:
            collect((turn + x) % E)
    return res, fills, walks


def single_gpu(args):
    import numpy as np
    import torch
    import psa_amd
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    psa_amd.set_device(local)
    dist, backend = (None, None)
    if world > 1:
        dist, backend = init_dist(local)
    k = args.log2n
    n = m = 1 << k
    S = min(MAX_SLOTS, max(1, args.slots))
    # (a slot holds n x m bytes of winner codes; slots before engines: 20 + 20 rather than 3 x 13; a twentieth of the
    # device memory stays free)
    free_b, _total_b = torch.cuda.mem_get_info()
    slot_b = float(n) * m * 1.02 + (64 << 20)
    fit = int(0.95 * free_b / slot_b)
    S = max(1, min(S, fit // 2 if args.steps > 1 else fit))
    batches = [S] * args.steps
    E = 2 if args.serial_fills else max(1, min(3, args.engines))
    E = max(1, min(E, len(batches), fit // S))
    pairs = slot_pairs(psa_amd, torch, k, S)
    a, b, a_dev, b_dev = pairs[0]
    gold = slot_goldens(k)
    # (an engine that does not fit after all -- another process on the GPU, a fragmented heap -- costs that engine, not the run)
    engines = []
    for _ in range(E):
        try:
            engines.append(psa_amd.Many(S, n, m, traceback=True))
        except Exception as exc:
            if not engines:
                raise
            print("bench: engine %d of %d not created (%s): going on with %d" % (len(engines) + 1, E, exc, len(engines)), file=sys.stderr)
            break
    E = len(engines)
    serial = args.serial_fills and E > 1

    # The timed region is `steps` batches = launches of S alignments each (--steps 20: 400 alignments; the default 12:
    # 240), three engines taking turns.  Warm-up: at least one launch on EACH engine (its ticket order, tables and code
    # pages are then in place), more if --warmup asks for more batches than that.
    per_launch = max(batches)
    warm_launches = max(E, args.warmup)
    warm_launches += (-warm_launches) % E
    throughput_job(engines, pairs, n, m, [per_launch] * warm_launches, serial)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
    ref.synchronize()
    fill_ends = []
    t0 = time.perf_counter()
    res, fills, walks = throughput_job(engines, pairs, n, m, batches, serial, ref.cuda_event, fill_ends)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        dtt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(dtt, op=dist.ReduceOp.MAX)
        dt = float(dtt.item())
    # every alignment of the timed region against the precomputed list (or, for sizes without goldens, against the
    # other launches' result for the same slot)
    aligned = args.steps * S
    assert len(res) == aligned, (len(res), aligned)
    by_slot = {}
    for sl, sc_, ln_ in res:
        by_slot.setdefault(sl, set()).add((sc_, ln_))
    assert all(len(v) == 1 for v in by_slot.values()), by_slot
    checked = 0
    for sl, v in by_slot.items():
        if sl in gold:
            assert next(iter(v)) == gold[sl], (sl, v, gold[sl])
            checked += 1
    sc, ln = next(iter(by_slot[0]))
    cells1 = float(n) * m
    cells = cells1 * world                # per alignment of the count below: one on every rank
    ms_step = dt / args.steps * 1e3
    ms_alignment = dt / aligned * 1e3
    code_bytes = engines[0].code_bytes(0)
    for e in engines:
        e.close()

    # ---- the dominant kernel: the fill launch over S pairs, HIP events on the engine's stream, timed region ----
    # One engine: the events in front of and behind the launch bracket it.  Several engines: the fills are queued together
    # and the chip takes workgroups from all of them (the event in front of a fill fires when it is QUEUED; the fills of
    # three engines end within a few ms of each other, fill_end_ms), so what a launch costs is the time from the first
    # launch's begin (queued on an idle chip) to the last launch's end, divided by the number of launches.
    full = [f for f, cnt in fills if cnt == per_launch]
    per_launch_pairs = per_launch
    ends = sorted(t1_ for _, t1_, _ in fill_ends)
    overlapped = E > 1 and not serial and len(ends) > 2 and all(cnt == per_launch for _, _, cnt in fill_ends)
    if overlapped:
        first_begin = min(t0_ for t0_, _, _ in fill_ends)
        fill_ms = (ends[-1] - first_begin) / len(ends)
        fill_ms_note = ("(end of the last fill launch - begin of the first) / %d launches of the timed region, HIP events on the "
                        "engines' fill streams: the launches of the %d engines are queued together and share the chip (their "
                        "start-to-end durations, fill_start_to_end_ms and rocprofv3's kernel trace, overlap and sum to more "
                        "than the time that passed); one launch alone on the chip: roofline.alone" % (len(ends), E))
    else:
        fill_ms = sum(full) / len(full)
        fill_ms_note = ("HIP events around the fill launch on the engine's stream, every launch of the timed region"
                        + ("; the other engine's walks (one wavefront per pair) run beside it" if E > 1 else ""))
    # Which fill kernel the launches took: couples of pairs through the packed 16-bit kernel (two pairs per lane) unless
    # PSA_FILL_PK=0 or one pair per launch; its roofline is the packed one (two 16-bit lane operations per lane and clock).
    packed = per_launch_pairs >= 2 and per_launch_pairs % 2 == 0 and os.environ.get("PSA_FILL_PK", "1") != "0"
    ops_cell = OPS_PER_CELL_CODES_PK if packed else OPS_PER_CELL_CODES
    peak = PK_PEAK_TOPS if packed else VALU_PEAK_TOPS
    achieved = per_launch_pairs * cells1 * ops_cell / (fill_ms * 1e-3) / 1e12
    aggregate = cells1 * ops_cell * aligned / dt / 1e12
    if packed:
        kname, kpat = "fill_strips_pk", "fill_strips_pk"
        grid = (per_launch_pairs // 2) * ((n + 255) // 256)
    else:
        kname, kpat = ("fill_strips_v2_dense<true>" if per_launch_pairs > 1 else "fill_strips_v2<true>"), "<true>(psa::FillBatch)"
        grid = per_launch_pairs * ((n + 255) // 256)
    traffic, traffic_src = pmc_traffic(kpat, grid * 256)
    out = {
        "metric": "GCUPS (cell updates/s), full alignment (fill + traceback) of 2^%d x 2^%d pairs" % (k, k),
        "value": cells * aligned / dt / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "ms_per_alignment": ms_alignment, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16" if packed else "int32", "data": "synthetic",
        "config": {"workload": "synthetic iid 20-letter protein pair 2^%d x 2^%d (mt19937_64 seeds 0xA/0xB0000000+%d), "
                               "fill with one-byte winner codes in HBM + traceback walk + prefix fix-up; one step = one "
                               "batch of %d complete alignments per GPU (one fill launch + one launch of the walks); ONE host thread, %d engine%s, %d DISTINCT pairs per "
                               "fill launch and per walk launch (slot s: seeds +0x100*s; psa_many / psa_align_many); no "
                               "inter-GPU communication" % (k, k, k, S, E, " (one launch holds the whole timed region)" if E == 1 else "s taking turns", per_launch),
                   "alignments_timed": aligned, "alignments_per_step": S,
                   "n": n, "m": m, "score": int(sc), "alignment_len": int(ln), "stripes": 1, "pairs_per_launch": per_launch,
                   "launch_sizes_timed": batches, "engine_slots": S, "engines": E,
                   "warmup_alignments_run": per_launch * warm_launches,
                   "warmup_note": "--warmup %d batches asked; run: %d launches of %d (one full launch on EACH engine at least)"
                                  % (args.warmup, warm_launches, per_launch),
                   "results_checked": "%d of %d slots against tests/golden/oracle_only.json (score, alignment length); every "
                                      "slot equal across launches" % (checked, len(by_slot)),
                   "fill_launches_overlap": not serial and E > 1, "host_threads": 1,
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                   "parallelism": "independent pairs per GPU" if world > 1 else "one GPU"},
        "roofline": {"bound": "valu",
                     "kernel": ("%s: ONE launch filling %d pairs (winner codes on), %s" % (
                         kname, per_launch_pairs, "two pairs per lane in packed 16-bit halves, three wavefronts per SIMD" if packed
                         else "two wavefronts per SIMD")) if per_launch_pairs > 1
                     else "fill_strips_v2<true>: one pair per launch (winner codes on), one wavefront per SIMD",
                     "achieved": achieved, "peak": peak, "unit": "T 16-bit lane operations / s" if packed else "Tint32op/s",
                     "frac": achieved / peak,
                     "ops_per_cell": ops_cell, "cells_per_launch": per_launch_pairs * cells1,
                     "frac_of_int32_peak_at_11_ops": per_launch_pairs * cells1 * OPS_PER_CELL_CODES / (fill_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                     "frac_at_survey_ops": per_launch_pairs * cells1 * SURVEY_OPS_CODES / (fill_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                     "ops_note": ("packed kernel: 13 instructions per step for the cells of TWO pairs (psa_fill_pk.hip.h), each a pair of "
                                  "16-bit lane operations, against a peak of 2 x 39.3 T such operations / s; "
                                  "frac_of_int32_peak_at_11_ops keeps the scale of the int32 kernel (11 = 8 for the recurrence + 3 for "
                                  "the one-byte winner code) and of earlier rounds, frac_at_survey_ops SURVEY 8d's 25") if packed else
                                 ("11 = 8 for the recurrence + 3 for the one-byte winner code; SURVEY 8d's 25 priced three "
                                  "compares per aggregate, which this kernel does not do (frac_at_survey_ops keeps that scale)"),
                     "valu_insts_per_cell_executed": pmc_valu_per_cell(kpat, grid * 256, per_launch_pairs * cells1),
                     "peak_note": ("the peak assumes 2.4 GHz; with its code stores (1 B per cell, 3.9 TB/s) the packed launch runs at "
                                   "about 1.8 GHz under the device's power management, VALU busy 0.92 of those cycles "
                                   "(SQ_BUSY_CYCLES over duration in timing-only builds, DESIGN 4.1b)") if packed else None,
                     "kernel_ms": fill_ms, "launches_timed": len(full),
                     "kernel_ms_note": fill_ms_note,
                     "fill_end_ms": [round(t, 3) for t in ends], "fill_start_to_end_ms": [round(f, 3) for f, _ in fills],
                     "fill_gcups": per_launch_pairs * cells1 / (fill_ms * 1e-3) / 1e9,
                     "aggregate_achieved": aggregate, "aggregate_frac": aggregate / peak,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (separate --pmc passes)",
                     "traffic_source": traffic_src,
                     "code_stream": {"bytes_per_launch": code_bytes * per_launch_pairs,
                                     "GBps": code_bytes * per_launch_pairs / (fill_ms * 1e-3) / 1e9,
                                     "frac_of_hbm_peak": code_bytes * per_launch_pairs / (fill_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        "traceback_ms": sum(walks) / len(walks),
    }
    stripes_info = None
    if dist and not args.no_stripes:
        # The stripes leg is the first place a run meets RCCL under a resident fill: it runs under a deadline, and a
        # leg that overruns it costs its own figures only (rank 0 prints the line measured above, all ranks leave).
        deadline = float(os.environ.get("PSA_BENCH_STRIPES_DEADLINE_S", "150"))

        def overrun():
            # (the replica line is still printed -- it is measured -- but the overrun is visible: a top-level flag in
            # the line and a non-zero exit status of every rank but 0, hence of the launcher, so that a hung fill
            # or a deadlock under it is investigated, not passed over)
            if rank == 0:
                out["stripes"] = {"error": "stripes leg not finished after %.0f s: abandoned, replica figures above stand" % deadline}
                out["stripes_failed"] = True
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            os._exit(0 if rank == 0 else 3)

        watchdog = threading.Timer(deadline + (5.0 if rank == 0 else 0.0), overrun)   # (rank 0 last: its line is the one that counts)
        watchdog.daemon = True
        watchdog.start()
        stripes_info = stripes_figures(args, dist, backend, local)
        watchdog.cancel()
    multi_leg = dist is not None and not args.no_stripes and torch.cuda.device_count() >= world and \
        os.environ.get("PSA_BENCH_ONE_PROCESS_STRIPES", "1") != "0"
    store = None
    if multi_leg:
        try:   # (the other ranks wait for rank 0's one-process leg on the rendezvous store: on the CPU, their GPUs idle)
            store = torch.distributed.distributed_c10d._get_default_store()
        except Exception:
            multi_leg = False
    if rank != 0:
        if multi_leg:
            try:
                store.wait(["psa_one_process_leg_done"], __import__("datetime").timedelta(seconds=240))
            except Exception as e:
                print("bench: rank %d stopped waiting for rank 0's one-process leg (%s)" % (rank, e), file=sys.stderr)
        try:
            dist.barrier()                # rank 0 is still measuring; leave together
            dist.destroy_process_group()
        except Exception as e:            # (a peer that left at the deadline above: this rank has nothing to print)
            print("bench: rank %d leaves without the closing barrier (%s)" % (rank, e), file=sys.stderr)
        if stripes_info is not None and "error" in stripes_info:
            sys.stdout.flush()
            os._exit(3)                   # (visible in the launcher's status; rank 0 prints the line with stripes_failed)
        return None

    # ---- the same alignment alone on the GPU: latency of one pair, and the one-pair fill kernel ----
    one = psa_amd.Many(1, n, m, traceback=True)
    lat, f1, w1 = [], [], []
    for rep in range(4):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one.set_pair_dev(0, a_dev.data_ptr(), n, b_dev.data_ptr(), m)
        one.launch()
        r = one.result(0, n, m)
        lat.append((time.perf_counter() - t1) * 1e3)
        assert (r.score, len(r.ai)) == (sc, ln)
        f, w = one.last_ms()
        f1.append(f); w1.append(w)
    one.close()
    lat_ms, fill1, walk1 = min(lat[1:]), sum(f1[1:]) / 3, sum(w1[1:]) / 3
    out["roofline"]["one_pair_alone"] = {"kernel_ms": fill1, "frac": cells1 * OPS_PER_CELL_CODES / (fill1 * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                                         "frac_at_survey_ops": cells1 * SURVEY_OPS_CODES / (fill1 * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                                         "structural_ceiling": min(1.0, cells1 / ((n + m) * 65536.0))}
    if overlapped:
        # the same launch alone on the chip: start-to-end duration, HIP events in front of and behind it (what rocprofv3's
        # kernel trace calls its duration)
        al = psa_amd.Many(per_launch, n, m, traceback=True)
        fa = []
        for rep in range(3):
            for sl in range(per_launch):
                al.set_pair_dev(sl, pairs[sl][2].data_ptr(), n, pairs[sl][3].data_ptr(), m)
            al.launch()
            al.wait()
            fa.append(al.last_ms()[0])
        al.close()
        ms_al = sum(fa[1:]) / len(fa[1:])
        out["roofline"]["alone"] = {"kernel_ms": ms_al, "pairs": per_launch,
                                    "frac": per_launch * cells1 * ops_cell / (ms_al * 1e-3) / 1e12 / peak,
                                    "note": "one launch with the chip to itself, start to end (ramp and drain included)"}
    out["one_pair_alone"] = {"ms": lat_ms, "gcups": cells1 / (lat_ms * 1e-3) / 1e9, "fill_ms": fill1, "traceback_ms": walk1}
    if stripes_info is not None:
        out["stripes"] = stripes_info
        if "error" in stripes_info:
            out["stripes_failed"] = True
    if multi_leg:
        # In a child process, under a deadline: the call has never driven more than one GPU (no multi-GPU box was
        # available), and whatever goes wrong there must cost that leg only -- the line measured above is still printed.
        try:
            import subprocess
            env = dict(os.environ)
            for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK"):
                env.pop(var, None)
            gs = gold.get(0, (None,))[0]
            cmd = [sys.executable, os.path.abspath(__file__), "--one-process-leg", str(world), "--log2n", str(k),
                   "--leg-gold-score", str(-1 if gs is None else gs)]
            limit = float(os.environ.get("PSA_BENCH_ONE_PROCESS_DEADLINE_S", "150"))
            try:
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit)
                lines = [ln for ln in cp.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
                out["stripes_one_process"] = json.loads(lines[-1]) if lines else {
                    "error": "child left with status %d and no result: %s" % (cp.returncode, cp.stderr.decode(errors="replace")[-300:])}
            except subprocess.TimeoutExpired:
                out["stripes_one_process"] = {"error": "not finished after %.0f s: abandoned" % limit}
            except Exception as e:
                out["stripes_one_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            store.set("psa_one_process_leg_done", "1")
    extra = {}
    if not args.no_extra:
        stream = torch.cuda.current_stream().cuda_stream
        # score-only fill of the same pair, alone
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
        extra["score_only_2p%d" % k]["structural_ceiling"] = min(1.0, cells1 / ((n + m) * 65536.0))
        # the same for S pairs in ONE launch (score only: what the batched fill reaches without the code stream)
        som = psa_amd.Many(S, n, m, traceback=False)
        fm = []
        for rep in range(3):
            for sl in range(S):
                som.set_pair_dev(sl, pairs[sl][2].data_ptr(), n, pairs[sl][3].data_ptr(), m)
            som.launch()
            som.wait()
            fm.append(som.last_ms()[0])
            got = [som.result(sl, n, m, want_alignment=False).score for sl in range(S)]
            assert all(gold[sl][0] == got[sl] for sl in range(S) if sl in gold), got
        som.close()
        ms = sum(fm[1:]) / len(fm[1:])
        so_pk = packed and S % 2 == 0   # (couples through the packed kernel: 10 instructions per step of two cells)
        extra["score_only_2p%d_x%d" % (k, S)] = {"gcups": S * cells1 / (ms * 1e-3) / 1e9, "kernel_ms": ms, "pairs": S,
                                                 "kernel": "fill_strips_pk<false>" if so_pk else "fill_strips_v2_dense<false>",
                                                 "valu_frac": (S * cells1 * 10 / (ms * 1e-3) / 1e12 / PK_PEAK_TOPS) if so_pk else
                                                              (S * cells1 * OPS_PER_CELL_SCORE / (ms * 1e-3) / 1e12 / VALU_PEAK_TOPS),
                                                 "valu_frac_of_int32_peak_at_8_ops": S * cells1 * OPS_PER_CELL_SCORE / (ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                                                 "structural_ceiling": min(1.0, S * cells1 / ((n + m) * 65536.0))}
        # BASELINE config 4's second size: ONE 2^20 x 2^20 pair, score only, one launch
        if os.environ.get("PSA_BENCH_2P20", "1") != "0":
            extra["score_only_2p20"] = huge_score_only(psa_amd, torch, 20)
        extra["titin_pair"] = titin_figures(psa_amd, torch, np)
        # BASELINE config 5 shape: a block of 512-residue proteins, all queries x all targets, score only
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_batch
        bench_batch.measure(nq=512, nt=256, L=512, reps=1, check=0)   # warm-up
        big = os.environ.get("PSA_BENCH_BATCH", "config5") == "config5"
        bb = bench_batch.measure(nq=100000 if big else 2048, nt=4096 if big else 256, L=512, reps=1 if big else 3, check=4, prewarm=not big)
        packed = os.environ.get("PSA_BATCH_PACKED", "1") != "0"
        bb["path"] = ("packed 16-bit, two pairs per lane (v_pk_add_u16 / v_pk_max_i16 / v_pk_min_u16): every couple of this "
                      "block has second sequences of one length") if packed else "int32, one pair per wavefront (PSA_BATCH_PACKED=0)"
        bb["ops_per_cell"] = OPS_PER_CELL_BATCH_PK if packed else OPS_PER_CELL_BATCH
        bb["peak_unit"] = "T 16-bit lane operations / s (2 x the int32 peak)" if packed else "T int32 operations / s"
        bb["valu_frac"] = (bb["gcups"] * OPS_PER_CELL_BATCH_PK / 1e3 / PK_PEAK_TOPS) if packed else (bb["gcups"] * OPS_PER_CELL_BATCH / 1e3 / VALU_PEAK_TOPS)
        bb["valu_frac_note"] = "= VALU instructions issued (6 per cell packed, 10 int32) x 64 lanes / 39.3 T lane-instructions per second"
        bb["note"] = ("BASELINE config 5 on one GPU: 100 000 synthetic 512-residue queries x a block of 4 096 targets, every "
                      "pair, score only, sequences and pair list resident in HBM (the full all-vs-all is 5e9 pairs: a "
                      "stated block of it, SURVEY 8d)") if big else "reduced block (PSA_BENCH_BATCH != config5)"
        extra["batch_512aa_score_only"] = bb
    out["extra"] = extra
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(a, b)
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:            # (peers that left at the stripes deadline: the line is measured, print it)
            print("bench: closing barrier failed (%s)" % e, file=sys.stderr)
    return out


def huge_score_only(psa_amd, torch, k):
    """One 2^k x 2^k pair, score only, one launch (1.1e12 cells at k = 20): the size at which one pair alone has
    strips enough for every SIMD (structural ceiling ~1)."""
    n = m = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n)
    b = psa_amd.synth_protein(0xB0000000 + k, m)
    stream = torch.cuda.current_stream().cuda_stream
    so = psa_amd.Stripe(a, b, 0, m, traceback=False)
    ts = []
    for rep in range(2):
        so.fill_rows(0, n, None, None, stream)
        ts.append(so.last_fill_ms())
    t = so.final(stream)
    so.close()
    ms, cells = ts[-1], float(n) * m
    want = None
    path = os.path.join(ROOT, "tests", "golden", "oracle_only.json")
    if os.path.exists(path):
        with open(path) as f:
            want = [e["score"] for e in json.load(f)["synthetic"] if e["log2"] == k]
    if want:
        assert int(max(t)) == want[0], (max(t), want)
    return {"gcups": cells / (ms * 1e-3) / 1e9, "kernel_ms": ms, "score": int(max(t)), "score_checked": bool(want),
            "valu_frac": cells * OPS_PER_CELL_SCORE / (ms * 1e-3) / 1e12 / VALU_PEAK_TOPS,
            "structural_ceiling": min(1.0, cells / ((n + m) * 65536.0)), "launches_timed": 1}


def titin_figures(psa_amd, torch, np):
    """The titin pair (BASELINE config 3): device-resident fill + traceback through the stripe API, and the
    end-to-end time of psa_align from host buffers (second call onwards: the thread's stripe objects are reused)."""
    seqdir = os.path.join(ROOT, "tests", "golden", "sequences")
    ta = np.frombuffer(psa_amd.read_fasta(os.path.join(seqdir, "titin_homo(50).fasta")), np.uint8)
    tb = np.frombuffer(psa_amd.read_fasta(os.path.join(seqdir, "titin_mus.fasta")), np.uint8)
    tn, tm = len(ta), len(tb)
    cells = float(tn) * tm
    stream = torch.cuda.current_stream().cuda_stream
    st = psa_amd.Stripe(ta, tb, 0, tm, traceback=True)
    ai = np.zeros(2 * (tn + tm) + 2, np.int32); bj = np.zeros(2 * (tn + tm) + 2, np.int32)
    fm, wm = [], []
    for rep in range(4):
        st.fill_rows(0, tn, None, None, stream)
        t = st.final(stream)
        kk, sc = psa_amd.pick(*t)
        _, _, _, ln = st.traceback(tn, tm, kk, ai, bj, 0, stream)
        fm.append(st.last_fill_ms()); wm.append(st.last_traceback_ms())
    st.close()
    fill, walk = sum(fm[1:]) / 3, sum(wm[1:]) / 3
    psa_amd.align(ta, tb, 1)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = psa_amd.align(ta, tb, 1)
    dtt = (time.perf_counter() - t0) / reps
    psa_amd.release_cache()
    return {"n": tn, "m": tm, "score": int(res.score), "alignment_len": int(len(res.ai)),
            "device_resident": {"fill_ms": fill, "traceback_ms": walk, "gcups_fill": cells / (fill * 1e-3) / 1e9,
                                "gcups_fill_plus_traceback": cells / ((fill + walk) * 1e-3) / 1e9,
                                "valu_frac_fill": cells * OPS_PER_CELL_CODES / (fill * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                                "structural_ceiling": min(1.0, cells / ((tn + tm) * 65536.0))},
            "gcups_end_to_end_host_buffers": cells / dtt / 1e9, "ms_end_to_end_host_buffers": dtt * 1e3,
            "note": "host-buffer figure: psa_align incl. H2D, table build, D2H and prefix fix-up (stripe objects reused)"}


def stripes_one_process(psa_amd, ngpu, log2n, gold_score):
    """Strong scaling of ONE pair through the C ABI's own multi-GPU call (psa_align_multi / psa_score_multi): ONE
    process -- this rank -- drives `ngpu` GPUs, stripe g on GPU g, boundaries written GPU to GPU through peer-mapped
    fine-grained memory (no collective library, no launcher).  2^k x 2^k full alignment and 2^20 x 2^20 score only."""
    info = {"gpus": ngpu, "route": "psa_align_multi / psa_score_multi from rank 0 (one process, peer access over xGMI)"}
    try:
        k = log2n
        n = m = 1 << k
        a = psa_amd.synth_protein(0xA0000000 + k, n)
        b = psa_amd.synth_protein(0xB0000000 + k, m)
        best, fills, res = None, [], None
        for rep in range(3):
            t0 = time.perf_counter()
            res = psa_amd.align_multi(a, b, ngpu=ngpu)
            dt = time.perf_counter() - t0
            fills.append(psa_amd.multi_last_fill_ms())
            best = dt if best is None else min(best, dt)
        assert gold_score is None or res.score == gold_score, (res.score, gold_score)
        fill = min(fills[1:])
        info["2p%d_full_alignment" % k] = {
            "fill_ms": fill, "gcups_fill": float(n) * m / (fill * 1e-3) / 1e9, "score": int(res.score), "alignment_len": int(len(res.ai)),
            "call_ms": best * 1e3, "call_note": "the whole call: buffers allocated, sequences uploaded, fill, walk across the GPUs, freed",
            "valu_frac_of_all_gpus": float(n) * m * OPS_PER_CELL_CODES / (fill * 1e-3) / 1e12 / (VALU_PEAK_TOPS * ngpu),
            "structural_ceiling": min(1.0, float(n) * m / ((n + m) * 65536.0 * ngpu)), "scaling": "strong"}
        k = int(os.environ.get("PSA_BENCH_STRIPES_LOG2N", "20"))
        n = m = 1 << k
        a = psa_amd.synth_protein(0xA0000000 + k, n)
        b = psa_amd.synth_protein(0xB0000000 + k, m)
        fills = []
        for rep in range(2):
            sc = psa_amd.score_multi(a, b, ngpu=ngpu)
            fills.append(psa_amd.multi_last_fill_ms())
        fill = min(fills)
        info["2p%d_score_only" % k] = {
            "fill_ms": fill, "gcups_fill": float(n) * m / (fill * 1e-3) / 1e9, "score": int(sc),
            "valu_frac_of_all_gpus": float(n) * m * OPS_PER_CELL_SCORE / (fill * 1e-3) / 1e12 / (VALU_PEAK_TOPS * ngpu),
            "structural_ceiling": min(1.0, float(n) * m / ((n + m) * 65536.0 * ngpu)), "scaling": "strong"}
    except Exception as e:   # (a secondary leg: its failure is reported, the line stands)
        info["error"] = "%s: %s" % (type(e).__name__, e)
    return info


def stripes_figures(args, dist, backend, local):
    """Strong scaling of ONE pair cut into world_size column stripes (BASELINE config 4): 2^16 x 2^16 full
    alignment and 2^20 x 2^20 score only, boundary columns streamed rank g -> g+1 (RCCL when the backend is nccl)."""
    import importlib
    import numpy as np
    import torch
    import psa_amd
    D = importlib.import_module("parallel-sequence-alignment_amd.distributed")
    world, rank = dist.get_world_size(), dist.get_rank()
    info = {"backend": backend, "chunk_rows": args.chunk_rows or "auto (distributed.auto_chunk_rows)"}
    try:
        ranks = [torch.zeros(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu") for _ in range(world)]
        dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64, device=ranks[0].device))
        info["rccl_ranks" if backend == "nccl" else "gloo_ranks"] = sorted(int(x.item()) for x in ranks)
        for k, tb, reps in ((args.log2n, True, 3), (int(os.environ.get("PSA_BENCH_STRIPES_LOG2N", "20")), False, 1)):
            n = m = 1 << k
            a = psa_amd.synth_protein(0xA0000000 + k, n)
            b = psa_amd.synth_protein(0xB0000000 + k, m)
            c0, c1 = D.stripe_bounds(m, world, rank)
            eng = D.HipStripeEngine(a, b, c0, c1, traceback=tb)
            # One launch per stripe with the boundaries travelling by flags where every rank has its own GPU
            # (align_striped's default); should that give up on any rank (all ranks raise together), a launch per chunk.
            flagged = {"0": False, "force": True}.get(os.environ.get("PSA_BENCH_STRIPES_FLAGGED", "1"), None)
            try:
                D.align_striped(a, b, chunk_rows=args.chunk_rows, traceback=tb, keep_engine=eng, flagged=flagged)   # warm-up
            except psa_amd.PsaError as e:
                if flagged is False:
                    raise
                info.setdefault("flagged_fill_gave_up", "%s" % e)
                flagged = False
                D.align_striped(a, b, chunk_rows=args.chunk_rows, traceback=tb, keep_engine=eng, flagged=flagged)
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                res = D.align_striped(a, b, chunk_rows=args.chunk_rows, traceback=tb, keep_engine=eng, flagged=flagged)
            dist.barrier(); torch.cuda.synchronize()
            dtt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64,
                               device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(dtt, op=dist.ReduceOp.MAX)
            eng.close()
            sec = float(dtt.item())
            ops = OPS_PER_CELL_CODES if tb else OPS_PER_CELL_SCORE
            info["2p%d_%s" % (k, "full_alignment" if tb else "score_only")] = {
                "ms": sec * 1e3, "gcups": float(n) * m / sec / 1e9, "score": int(res.score), "stripes": world, "scaling": "strong",
                "fill": "one launch per stripe, boundaries by flags" if (flagged or (flagged is None and int(os.environ.get("LOCAL_WORLD_SIZE", world)) <= torch.cuda.device_count())) else "one launch per row chunk",
                "valu_frac_of_all_gpus": float(n) * m * ops / sec / 1e12 / (VALU_PEAK_TOPS * world),
                "structural_ceiling": min(1.0, float(n) * m / ((n + m) * 65536.0 * world))}
    except Exception as e:   # the replica figure must not be lost to the stripes leg
        info["error"] = "%s: %s" % (type(e).__name__, e)
    return info


def main():
    args = parse()
    if args.one_process_leg > 0:   # (child of an N > 1 run: psa_align_multi / psa_score_multi over N GPUs, its own JSON line)
        import psa_amd
        print(json.dumps(stripes_one_process(psa_amd, args.one_process_leg, args.log2n,
                                             None if args.leg_gold_score < 0 else args.leg_gold_score)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or world > 1) and args.mode == "stripes":
        from bench_multi import multi_gpu   # one pair cut into column stripes (strong scaling) only
        out = multi_gpu(args)
    else:
        out = single_gpu(args)              # independent alignments on every GPU (weak scaling)
    if out is not None:
        print(json.dumps(out))
        sys.stdout.flush()
        if world > 1 and isinstance(out.get("stripes"), dict) and "error" in out["stripes"]:
            os._exit(0)                     # (a process group whose stripes leg failed may not shut down: the line is out, with stripes_failed; the other ranks leave with status 3)


if __name__ == "__main__":
    main()
