"""bench.py's N > 1 leg: one rank per GPU (torch.distributed.run), RCCL send/recv over xGMI.

The same 2^k x 2^k pair as the single-GPU bench, cut into N column stripes of B (the reference's
partition, simple.cpp:90-99), stripe g on GPU g.  Strong scaling: total work is fixed.  A step is
one complete alignment: pipelined fill of all stripes in row chunks, final pick, traceback chain
right to left, gather + prefix fix-up on rank 0.
"""
import os
import time

import numpy as np


def multi_gpu(args):
    import torch
    import torch.distributed as dist
    import importlib
    import psa_amd
    D = importlib.import_module("parallel-sequence-alignment_amd.distributed")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    psa_amd.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    k = args.log2n
    n = m = 1 << k
    a = psa_amd.synth_protein(0xA0000000 + k, n)
    b = psa_amd.synth_protein(0xB0000000 + k, m)
    c0, c1 = D.stripe_bounds(m, world, rank)
    eng = D.HipStripeEngine(a, b, c0, c1, traceback=True)

    def step():
        return D.align_striped(a, b, chunk_rows=args.chunk_rows, traceback=True, keep_engine=eng)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    cells = float(n) * m
    out = None
    if rank == 0:
        from bench import VALU_PEAK_TOPS, OPS_PER_CELL_CODES
        gcups = cells / (dt / args.steps) / 1e9
        out = {
            "metric": "GCUPS (cell updates/s), full alignment (fill + traceback) of a 2^%d x 2^%d pair" % (k, k),
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "synthetic iid 20-letter protein pair 2^%d x 2^%d, %d column stripes of B "
                                   "(one per GPU), boundary columns streamed rank g -> g+1 by RCCL send/recv in chunks "
                                   "of %d rows; fill with one-byte winner codes + traceback chain + prefix fix-up"
                                   % (k, k, world, args.chunk_rows or D.auto_chunk_rows(m // world)),
                       "n": n, "m": m, "score": int(res.score), "alignment_len": int(len(res.ai)), "stripes": world,
                       "chunk_rows": args.chunk_rows or D.auto_chunk_rows(m // world)},
            "roofline": {"bound": "valu", "achieved": gcups * OPS_PER_CELL_CODES / 1e3,
                         "peak": VALU_PEAK_TOPS * world, "unit": "Tint32op/s",
                         "frac": gcups * OPS_PER_CELL_CODES / 1e3 / (VALU_PEAK_TOPS * world),
                         "ops_per_cell": OPS_PER_CELL_CODES,
                         "structural_ceiling": min(1.0, cells / ((n + m) * 65536.0 * world)), "traffic": None,
                         "note": "whole-job figure; the dependent anti-diagonal chain (n + m steps) is the same on "
                                 "1 and on N GPUs, so at this size extra GPUs add capacity the problem cannot use"},
        }
    eng.close()
    dist.destroy_process_group()
    return out
