"""MI355X-native drop-in for the alignment hot path of agarfel/Parallel-Sequence-Alignment.

Host-side mirror of the reference's in-process interface (``result run(int p, str A, str B)``,
simple.cpp:261) over the C ABI of ``include/psa.h`` (``lib/libpsa_hip.so``, hand-written gfx950
kernels).  There is no CPU path in this package: without the built library or without a GPU
every compute call raises.

The directory name contains a hyphen, so import it through the ``psa_amd`` shim at the repo
root (``import psa_amd``) or with ``importlib.import_module("parallel-sequence-alignment_amd")``.
"""
from .binding import (  # noqa: F401
    Many,
    PsaError,
    Result,
    SeqSet,
    Stripe,
    align,
    align_many,
    align_multi,
    device_count,
    edge_post,
    finish_alignment,
    format_result,
    lib,
    lib_path,
    multi_last_fill_ms,
    pick,
    read_fasta,
    read_fasta_records,
    release_cache,
    run,
    score,
    score_batch,
    score_multi,
    stream_create,
    stream_destroy,
    set_device,
    synth_protein,
)
