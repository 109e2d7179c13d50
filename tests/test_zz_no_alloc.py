"""The allocation-free entry point psa_score_batch_dev_noalloc (scratch = a static device array inside the
library) against psa_score_batch_dev (scratch = a workspace the caller hands over): same scores, twice in a
row, the second call with a smaller input.  Collected last (file name) so that a failure here hides nothing."""
import ctypes as C

import numpy as np
import pytest

import psa_amd

pytestmark = pytest.mark.gpu


def _batch(seed, nseq, max_len, npairs):
    rng = np.random.default_rng(seed)
    seqs = [rng.integers(65, 65 + [2, 4, 20][i % 3], int(rng.integers(1, max_len + 1))).astype(np.uint8) for i in range(nseq)]
    offsets = np.zeros(nseq + 1, np.int64)
    offsets[1:] = np.cumsum([s.size for s in seqs])
    pairs = rng.integers(0, nseq, (npairs, 2)).astype(np.int32)
    return seqs, np.concatenate(seqs), offsets, pairs


def test_noalloc_matches_workspace_variant_twice():
    import torch
    lib = psa_amd.lib()
    stream = torch.cuda.current_stream().cuda_stream
    for seed, nseq, max_len, npairs in ((1, 40, 700, 5000), (2, 12, 300, 900)):   # second call: smaller input
        seqs, flat, offsets, pairs = _batch(seed, nseq, max_len, npairs)
        d_flat, d_off, d_pairs = torch.from_numpy(flat).cuda(), torch.from_numpy(offsets).cuda(), torch.from_numpy(pairs).cuda()
        want = torch.full((npairs,), -7, dtype=torch.int32, device="cuda")
        got = torch.full((npairs,), -9, dtype=torch.int32, device="cuda")
        ctr = torch.zeros(64, dtype=torch.int32, device="cuda")
        max_b = int(max(s.size for s in seqs))
        rc = lib.psa_score_batch_dev(d_flat.data_ptr(), d_off.data_ptr(), d_pairs.data_ptr(), npairs, max_b,
                                     want.data_ptr(), ctr.data_ptr(), stream)
        assert rc == 0, lib.psa_last_error()
        torch.cuda.synchronize()
        rc = lib.psa_score_batch_dev_noalloc(d_flat.data_ptr(), d_off.data_ptr(), d_pairs.data_ptr(), npairs, max_b,
                                             got.data_ptr(), stream)
        assert rc == 0, lib.psa_last_error()
        torch.cuda.synchronize()
        assert torch.equal(want, got), (seed, int((want != got).sum()))
        # and both equal the host-buffer entry point (which also sends oversize pairs through psa_score)
        host = psa_amd.score_batch(seqs, [tuple(p) for p in pairs[:200]])
        assert np.array_equal(host, got[:200].cpu().numpy())


def test_noalloc_rejects_sizes_beyond_its_limits():
    import torch
    lib = psa_amd.lib()
    x = torch.zeros(64, dtype=torch.int64, device="cuda")
    rc = lib.psa_score_batch_dev_noalloc(x.data_ptr(), x.data_ptr(), x.data_ptr(), (1 << 30) + 1, 512, x.data_ptr(), None)
    assert rc == -1 and b"exceed the maximum" in lib.psa_last_error()
    rc = lib.psa_score_batch_dev_noalloc(x.data_ptr(), x.data_ptr(), x.data_ptr(), 10, 1025, x.data_ptr(), None)
    assert rc == -1 and b"max_len_b" in lib.psa_last_error()
