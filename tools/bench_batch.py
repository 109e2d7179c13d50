"""Measure the batch score kernel (BASELINE config 5 shape: synthetic 512-aa proteins, a block of
queries x targets, score only, everything resident in HBM).  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import psa_amd


def measure(nq=2048, nt=256, L=512, reps=3, check=8, prewarm=True):
    """nq queries x nt targets (the first nt of the same synthetic set), every pair, score only."""
    seqs = [psa_amd.synth_protein(0xC0000000 + i, L) for i in range(max(nq, nt))]
    flat = np.concatenate(seqs)
    offsets = np.arange(len(seqs) + 1, dtype=np.int64) * L
    d_flat = torch.from_numpy(flat).cuda(); d_off = torch.from_numpy(offsets).cuda()
    # the pair list is made on the device (100 000 x 4 096 pairs are 3.3 GB)
    d_pairs = torch.empty((nq * nt, 2), dtype=torch.int32, device="cuda")
    d_pairs[:, 0] = torch.arange(nq, dtype=torch.int32, device="cuda").repeat_interleave(nt)
    d_pairs[:, 1] = torch.arange(nt, dtype=torch.int32, device="cuda").repeat(nq)

    class _P:   # (shape and row access of the pair list, without a host copy)
        shape = (nq * nt, 2)

        def __getitem__(self, key):
            idx, col = key
            return int(idx) // nt if col == 0 else int(idx) % nt
    pairs = _P()
    d_scores = torch.zeros(pairs.shape[0], dtype=torch.int32, device="cuda")
    d_ctr = torch.zeros(64, dtype=torch.int32, device="cuda")
    lib = psa_amd.lib()
    stream = torch.cuda.current_stream().cuda_stream

    # launches of at most 2^22 pairs (about 0.3 s each): a kernel that runs for half a minute holds the
    # GPU's queue against everything else and reads as a stall in a kernel trace
    per_launch = 1 << 22

    def run():
        for p0 in range(0, pairs.shape[0], per_launch):
            cnt = min(per_launch, pairs.shape[0] - p0)
            rc = lib.psa_score_batch_dev(d_flat.data_ptr(), d_off.data_ptr(), d_pairs.data_ptr() + 8 * p0, cnt, L,
                                         d_scores.data_ptr() + 4 * p0, d_ctr.data_ptr(), stream)
            assert rc == 0, psa_amd.lib().psa_last_error()
    if prewarm:
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out = {"pairs": int(pairs.shape[0]), "residues": L, "pairs_per_launch": int(min(per_launch, pairs.shape[0])), "ms": dt * 1e3,
           "gcups": pairs.shape[0] * float(L) * L / dt / 1e9}
    if check:
        import oracle_lib as O
        rng = np.random.default_rng(5)
        for idx in rng.integers(0, pairs.shape[0], check):
            a, b = seqs[pairs[idx, 0]], seqs[pairs[idx, 1]]
            got = int(d_scores[int(idx)].item())
            assert got == O.score(a, b), (idx, got)
        out["checked_against_oracle"] = int(check)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config5":
        print(json.dumps(measure(nq=100000, nt=4096, L=512, reps=1, check=8, prewarm=False)))
    else:
        print(json.dumps(measure()))
