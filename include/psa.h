/*
 * psa.h -- C ABI of the MI355X (gfx950) alignment library, libpsa_hip.so.
 *
 * This is the drop-in boundary for ONE path of agarfel/Parallel-Sequence-Alignment: the
 * three-table dynamic-programming fill and alignment reconstruction of `simple.cpp`.  The
 * reference has no FFI; its in-process boundary is `result run(int p, str A, str B)`
 * (simple.cpp:261) called once by `main` (simple.cpp:311).  Every entry point below cites the
 * reference code it replaces.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - return 0 on success, a negative PSA_E* code on failure; psa_last_error() gives the text
 *     (thread-local).  The library never falls back to a CPU path: with no usable GPU every
 *     compute entry point fails with PSA_E_NO_DEVICE.
 *   - residues are raw bytes; only byte equality is used (simple.cpp:13-16), so any injective
 *     byte->uint8 encoding gives identical results and the identity is used.
 *   - alignment columns are pairs (ai[k], bj[k]) of 0-based indices into A and B, -1 = gap
 *     (simple.h:24), already including the prefix fix-up of simple.cpp:279-290.
 *   - host output arrays are allocated by the library and released with psa_free().
 *   - one call at a time per stripe object; different stripe objects are independent.
 */
#ifndef PSA_H
#define PSA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSA_OK 0
#define PSA_E_ARG (-1)        /* bad argument */
#define PSA_E_NO_DEVICE (-2)  /* no usable HIP device: the product path has no CPU fallback */
#define PSA_E_HIP (-3)        /* a HIP runtime call failed */
#define PSA_E_NOMEM (-4)      /* host or device allocation failed */
#define PSA_E_KERNEL (-5)     /* a kernel reported a stalled hand-off (bounded spin expired) */
#define PSA_E_STATE (-6)      /* calls made in the wrong order */

#define PSA_F_TRACEBACK 1u    /* keep winner codes in HBM (one byte per cell) so an alignment can be rebuilt */
#define PSA_F_BANDED 2u       /* with PSA_F_TRACEBACK: codes for one band of rows at a time (psa_align_banded) */

const char* psa_last_error(void);
const char* psa_version(void);
int psa_device_count(void);
/* Select the HIP device used by subsequent calls of this thread (hipSetDevice). */
int psa_set_device(int device);
void psa_free(void* p);

/* ---- whole-pair entry points: replace run() (simple.cpp:261-294) ---------------------- */

/*
 * Align A (n bytes) against B (m bytes).  `p` is the reference's third argument: the number
 * of column stripes B is cut into (block = m/p, the last stripe takes the remainder,
 * simple.cpp:90-99).  Here the stripes run as a pipeline on the current GPU; as in the
 * reference the result does not depend on p.  p < 1 is an error (the reference divides by
 * zero, simple.cpp:92).
 * Outputs: *score_out = result.score; (*ai_out, *bj_out, *len_out) = result.al.data.
 * On one GPU all p stripes are jobs of ONE kernel launch: stripe g+1 takes the rows of its left
 * boundary as stripe g's strips finish them (the mailbox of simple.cpp:136-154 as flags in device
 * memory), so p costs a few per cent, not a factor.  The stripe objects of the calling thread's most
 * recent whole-pair call are kept and reused by the next call of the same shape (n, m, p):
 * psa_release_cache() frees them.
 */
int psa_align(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p,
              int32_t* score_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out);

/* Score only: the fill of simple.cpp:122-240 and the final pick of simple.cpp:244-256 with no
 * traceback storage.  triple_out (may be NULL) receives T1,T2,T3 at (n, m). */
int psa_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p,
              int32_t* score_out, int32_t triple_out[3]);

/* Free the device buffers psa_align / psa_score / psa_*_multi and the engines psa_align_many keep for the calling
 * thread (see psa_align, psa_align_many). */
void psa_release_cache(void);

/*
 * run() with its p workers on several GPUs of THIS process (simple.cpp:261-277 spawns its workers inside run(), so
 * the reference's caller gets the parallel run from the one call; this is that call for GPUs).  B is cut into ngpu
 * column stripes exactly as simple.cpp:90-99 cuts it for p = ngpu; stripe g is filled by ONE launch on GPU
 * devices[g] (NULL: GPU g modulo the number of visible GPUs).  The boundary between stripes g and g+1 -- T1,T2,T3 of stripe g's last
 * column per row and one "done" flag per 64 rows, the mailbox of simple.cpp:136-154 / 215-235 -- is fine-grained
 * memory of GPU g+1 that GPU g's fill writes directly (peer access over xGMI) and GPU g+1's fill polls: no
 * collective library, no launcher, no host loop while the fills run.  The walk goes right to left through the
 * stripes, each on its own GPU.  Result = psa_align's for any ngpu.  A device may be named more than once if the
 * stripes that share it fit on it together (ceil(n / 256) workgroups each, one per compute unit).  Needs hipDeviceCanAccessPeer
 * between neighbours.  psa_score_multi: the same without traceback storage (BASELINE config 4, 2^20 x 2^20).
 * The `./simple` front end calls psa_align_multi when PSA_GPUS=N (N > 1) is set in the environment.
 */
int psa_align_multi(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                    int32_t* score_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out);
int psa_score_multi(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                    int32_t* score_out, int32_t triple_out[3]);
/* Milliseconds (host clock) from the first stripe's fill being queued to every stripe's fill being complete, in the
 * calling thread's most recent psa_align_multi / psa_score_multi: the fill pipeline without allocation, upload and walk. */
double psa_multi_last_fill_ms(void);

/* ---- several pairs at once: run() called in a loop (simple.cpp:311), one launch per batch ------
 *
 * One 2^16 pair cannot fill the GPU (its dependency chain is n + m anti-diagonal steps long whatever
 * the hardware), so throughput comes from pairs in flight.  psa_align_many aligns npairs independent
 * pairs: a[k] (n[k] bytes) against b[k] (m[k] bytes); scores_out[k] = result.score, and -- when
 * ai_out, bj_out, len_out are not NULL -- (ai_out[k], bj_out[k], len_out[k]) = result.al.data of pair
 * k, each array released with psa_free().  Results are those of psa_align pair by pair.  Internally
 * the pairs go in as few batches of equal size as possible, a batch being up to PSA_MANY_MAX_SLOTS pairs: ONE fill
 * launch over all its pairs (the strips of the pairs interleaved, each pair advancing at the pace of its
 * own chain) and ONE launch of their traceback walks.  Up to three psa_many engines take turns (one when a
 * single batch holds every pair; two where memory is short): their fills are queued together and share the chip.
 * Pairs of the same size (taken side by side, whatever their order in the call) are filled two per lane in packed
 * 16-bit halves (psa_fill_pk.hip.h; PSA_FILL_PK=0: every pair through the int32 kernel).  One host thread.
 * The engines (up to 70 % of the free device memory when alignments are wanted) stay allocated for the calling thread's
 * next psa_align_many of the same kind (with or without alignments) whose pairs and batches are no larger; they are
 * freed by psa_release_cache(), by a call they do not fit, by a psa_align / psa_score / psa_align_banded call of this
 * thread whose buffers would take more than 70 % of the memory left beside them, and by its psa_*_multi calls.
 */
int psa_align_many(int64_t npairs, const uint8_t* const* a, const int64_t* n, const uint8_t* const* b,
                   const int64_t* m, int32_t* scores_out, int32_t** ai_out, int32_t** bj_out,
                   int64_t* len_out);

/* The engine behind it, for callers whose pairs are resident in device memory or who run their own
 * loop.  `slots` (1..PSA_MANY_MAX_SLOTS) pairs of up to n_cap x m_cap are in flight per launch; flags: PSA_F_TRACEBACK
 * for alignments, 0 for scores.  All device and pinned host buffers are allocated by create; the
 * other calls allocate nothing on the device (a changed set of pair sizes re-uploads a few KB of
 * ticket order).  An engine owns a stream; calls on one engine must not overlap, engines are independent.
 *   set_pair / set_pair_dev  bind a slot to a pair; clear_slot leaves it out.  Host residues are copied at once (the
 *                            caller's buffers are its own again on return).  Device residues are NOT read before the
 *                            next launch, whose binding kernels copy and encode the pairs of all slots together:
 *                            they must stay valid and unchanged until that launch's results have been waited for
 *   launch                   queue the binding kernels and ONE fill launch over all bound slots; returns at once
 *   wait                     block until those results are on the host.  The walks and the copies of the results are
 *                            queued here, once the fill has ended, on the engine's high-priority stream (queued at the
 *                            launch they would sit in that stream behind a wait for the fill, and a high-priority
 *                            queue with an unsatisfied wait was seen to keep fills of normal priority from being
 *                            dispatched for seconds): call wait when the results are wanted, not long after
 *   result                   score (and T1,T2,T3 at (n, m) when triple_out is not NULL) of a slot; with
 *                            ai/bj (capacity cap >= 2 (n + m) + 2 each) also the finished alignment,
 *                            prefix fix-up of simple.cpp:279-290 applied
 *   last_ms                  device time of the last launch's fill and of its walks (HIP events on
 *                            the engine's stream)
 */
#define PSA_MANY_MAX_SLOTS 20   /* what the 4 KB argument block of a fill launch holds */
typedef struct psa_many psa_many;
int psa_many_create(psa_many** out, int slots, int64_t n_cap, int64_t m_cap, uint32_t flags);
void psa_many_destroy(psa_many* e);
int psa_many_set_pair(psa_many* e, int slot, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m);
int psa_many_set_pair_dev(psa_many* e, int slot, const uint8_t* a_dev, int64_t n, const uint8_t* b_dev,
                          int64_t m);
int psa_many_clear_slot(psa_many* e, int slot);
int psa_many_launch(psa_many* e);
/* The same, but this engine's fill starts only when `after`'s fill launch has ended (NULL: at once): every fill launch
 * then has the chip to itself.  Since round 3 psa_align_many does NOT do this: a fill launch queued behind another one
 * fills the chip as that one drains (a launch of 2^16 x 2^16 pairs costs 4 ms of ramp and drain when it is alone:
 * 62 against 67 ms for two launches of 16 pairs), which needs the launch on the queue before the other drains -- so psa_many_launch
 * never waits on the host, not even for the alphabet of a pair bound from device memory (the table kernel takes any
 * byte alphabet). */
int psa_many_launch_after(psa_many* e, psa_many* after);
int psa_many_wait(psa_many* e);
/* Without blocking: 1 when the last launch's results are on the host (or nothing is launched), 0 when not yet, < 0 on
 * error.  Queues the engine's walks and result copies as soon as its fill has ended: a caller that is busy with the
 * results of one engine calls this on the others now and then, so that their walks do not wait for their turn. */
int psa_many_poll(psa_many* e);
int psa_many_result(psa_many* e, int slot, int32_t* score_out, int32_t triple_out[3], int32_t* ai,
                    int32_t* bj, int64_t cap, int64_t* len_out);
int psa_many_last_ms(psa_many* e, float* fill_ms_out, float* walk_ms_out);
/* Where the last launch's fill sits on the device's time line: milliseconds from `ref_event` (a hipEvent_t of the caller,
 * recorded with timing on this device and already reached) to the events the engine recorded in front of and behind its
 * fill launch.  With several engines in flight a fill is queued while the one before it runs, so the event in front of it
 * says when it was queued, not when it began: the spacing of the END times of consecutive launches is what a launch costs
 * then (bench.py).  No counterpart in the reference (measurement only). */
int psa_many_fill_ms_since(psa_many* e, void* ref_event, float* begin_ms_out, float* end_ms_out);
int64_t psa_many_code_bytes(const psa_many* e, int slot);

/*
 * Many independent pairs, score only, one GPU, no communication (BASELINE config 5).
 * seqs/offsets: sequence s is seqs[offsets[s] .. offsets[s+1]); pairs[2k], pairs[2k+1] are
 * the sequence numbers of pair k's A and B.  scores_out[k] = what run() would return as
 * result.score for that pair.  No counterpart in the reference beyond calling run() in a loop.
 */
int psa_score_batch(const uint8_t* seqs, const int64_t* offsets, int64_t nseqs,
                    const int32_t* pairs, int64_t npairs, int32_t* scores_out);

/* Same with everything RESIDENT in device memory (the form a serving loop and bench use):
 * seqs_dev/offsets_dev/pairs_dev as above, scores_dev receives npairs int32, counter_dev is a
 * 256-byte device scratch the library zeroes (its work counter).  max_len_b = the longest second
 * sequence among the pairs (sizes the per-wavefront LDS row buffer), at most 1024: a pair whose
 * second sequence is longer is NOT aligned, its score slot receives INT32_MIN, and the caller
 * sends it through psa_score (psa_score_batch does).  Asynchronous on `stream`. */
int psa_score_batch_dev(const uint8_t* seqs_dev, const int64_t* offsets_dev, const int32_t* pairs_dev,
                        int64_t npairs, int max_len_b, int32_t* scores_dev, unsigned* counter_dev,
                        void* stream);

/* psa_score_batch_dev for callers that cannot allocate device memory and have no workspace to hand
 * over: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  The kernel's only scratch -- its 256-byte work counter -- lives in a
 * static device array inside the library (256 bytes per GPU, held until the process ends); the call
 * performs no hipMalloc / hipFree and calls no allocating routine.  Limits: npairs <= 2^30 per call,
 * max_len_b <= 1024 (as psa_score_batch_dev); beyond them PSA_E_ARG with a message.  Every call on a
 * GPU shares that GPU's counter, which the call re-zeroes on `stream`: calls that overlap in time on
 * one GPU (different streams or threads) race and must be serialised by the caller. */
int psa_score_batch_dev_noalloc(const uint8_t* seqs_dev, const int64_t* offsets_dev, const int32_t* pairs_dev,
                                int64_t npairs, int max_len_b, int32_t* scores_dev, void* stream);

/* The five stdout lines of the reference (simple.cpp:312-313, output_alignement 43-70) for a
 * result; returned buffer is released with psa_free(), *out_len is its length in bytes. */
char* psa_format_result(int32_t score, const uint8_t* a, const uint8_t* b, const int32_t* ai,
                        const int32_t* bj, int64_t len, int64_t* out_len);

/* Host-side prefix fix-up of simple.cpp:279-290 + find_last (simple.cpp:24-41), applied in
 * place to a traceback given in REVERSE order (as psa_stripe_traceback emits it).  On return
 * the arrays hold the final alignment in forward order; cap must be >= len + n + m. */
int psa_finish_alignment(int32_t* ai, int32_t* bj, int64_t len, int64_t cap, int64_t* len_out);

/* ---- column-stripe objects: the reference's per-thread stripe, one per GPU -------------
 *
 * A psa_stripe is one thread's share in solve_subproblem_parallel (simple.cpp:87-259): all
 * rows of A against columns c0+1..c1 of B, resident on one GPU.  Rows are filled in chunks,
 * top to bottom.  For each chunk the caller supplies the left boundary -- T1,T2,T3 of column
 * c0 for the chunk's rows, what the reference passes through sharingT (simple.cpp:143-149)
 * -- and receives the right boundary (simple.cpp:223-229) to hand to the next stripe.
 * Between GPUs those buffers travel by RCCL send/recv; on one GPU they are passed directly.
 */
typedef struct psa_stripe psa_stripe;

/* a, b are HOST pointers to the full sequences; columns (c0, c1] of B are kept. 0<=c0<=c1<=m. */
int psa_stripe_create(psa_stripe** out, const uint8_t* a, int64_t n, const uint8_t* b,
                      int64_t m, int64_t c0, int64_t c1, uint32_t flags);
void psa_stripe_destroy(psa_stripe* s);

/* Same stripe, new sequences of the same lengths already RESIDENT in device memory
 * (a_dev: n bytes, b_dev: the stripe's c1-c0 bytes).  Device-to-device copy on `stream`. */
int psa_stripe_set_sequences_dev(psa_stripe* s, const uint8_t* a_dev, const uint8_t* b_dev,
                                 void* stream);

/*
 * Fill rows r0+1..r1.  Chunks must be given in order: the first call has r0 = 0, each later
 * call has r0 = the previous r1; r0 must be a multiple of 64 (one wavefront owns 64 rows).
 * left_dev : DEVICE pointer to 3*(r1-r0) int32, [k*(r1-r0) + y] = T_{k+1}(r0+1+y, c0), or
 *            NULL when c0 == 0 (the head stripe computes its own column 0,
 *            simple.cpp:124-133).
 * right_dev: DEVICE pointer to 3*(r1-r0) int32 receiving the same layout at column c1, or
 *            NULL when nobody consumes it.
 * stream   : hipStream_t (NULL = the default stream).  The call is asynchronous.
 */
int psa_stripe_fill_rows(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev,
                         int32_t* right_dev, void* stream);

/*
 * The same fill as ONE launch over all rows 1..n whose boundaries arrive and leave in row chunks WHILE it runs --
 * the per-row hand-off between neighbouring threads of simple.cpp:136-154 (wait for the left neighbour's row)
 * and :215-235 (publish the own last column), at the granularity of 64-row strips, for stripes that live on
 * different GPUs.  A launch per chunk (psa_stripe_fill_rows) leaves a wide stripe's chain with only as many
 * chunks in flight as there are streams; this one keeps the whole chain alive.
 * left_dev   : DEVICE int32[3][n], plane k row y = T_{k+1}(1+y, c0); NULL when c0 == 0.
 * left_ready : DEVICE uint32[(n+63)/64], zeroed by the caller before the call; strip g (rows 64g+1..64g+64) starts
 *              once left_ready[g] and left_ready[g-1] are non-zero.  psa_edge_post fills a chunk of left_dev from a
 *              packed [3][rows] buffer (what a neighbour sent) and raises its flags.  NULL: left_dev is complete.
 * right_dev  : DEVICE int32[3][n] receiving column c1, or NULL.
 * right_done : DEVICE uint32[(n+63)/64], zeroed by the caller; right_done[g] is raised when strip g's rows of
 *              right_dev are written.  psa_edge_take waits (on its stream, on the device) for a chunk's last strip
 *              and packs the chunk as [3][rows] for sending.  NULL: read right_dev after the launch.
 * The fill, the posts and the takes go to three different streams.  A strip's wait that lasts longer than 30 s (a
 * chunk that never arrives) ends the launch (psa_edge_take gives up after 60 s); psa_stripe_final then returns
 * PSA_E_KERNEL.  psa_edge_take waits for the flag of EVERY strip of its chunk.
 */
/* A stream (hipStream_t) for the flagged fill.  HIP multiplexes streams onto a few hardware queues (4 by default),
 * and a kernel does not start before the kernel queued ahead of it on the same hardware queue has ended: a flagged
 * fill whose stream shares a queue with the stream that delivers its boundary waits for ever.  The runtime keeps
 * one pool of hardware queues per stream PRIORITY, so the fill goes to a stream of a priority nothing else uses:
 * priority_class < 0 lowest (what distributed.py uses: communication kernels win), > 0 highest, 0 normal.
 * reserved_cus > 0: instead, a stream that leaves that many compute units to other streams
 * (hipExtStreamCreateWithCUMask: a queue of its own too, but a BLOCKING stream -- it synchronises with the null
 * stream); experimental. */
int psa_stream_create(void** stream_out, int priority_class, int reserved_cus);
int psa_stream_destroy(void* stream);
int psa_stripe_fill_flagged(psa_stripe* s, const int32_t* left_dev, const uint32_t* left_ready,
                            int32_t* right_dev, uint32_t* right_done, void* stream);
int psa_edge_post(const int32_t* packed_dev, int64_t r0, int64_t rows, int64_t n, int32_t* left_dev,
                  uint32_t* left_ready, void* stream);
int psa_edge_take(psa_stripe* s, const int32_t* right_dev, const uint32_t* right_done, int64_t r0,
                  int64_t rows, int32_t* packed_dev, void* stream);

/* T1,T2,T3 at (n, c1) after the last chunk (synchronises `stream`).  For the last stripe
 * pick() of these is the score (simple.cpp:246-256). */
int psa_stripe_final(psa_stripe* s, void* stream, int32_t triple_out[3]);

/*
 * Walk the alignment back through this stripe (needs PSA_F_TRACEBACK).  State (i, j, k):
 * cell (i, j) of table k; the first call starts at (n, m, k*) with k* the
 * winning table of the final pick.  Columns are APPENDED in walk order (i.e. reversed) to
 * ai/bj (HOST arrays of capacity cap) starting at *len_io.
 *
 * The hand-over between stripes, exactly (any engine that shares a walk with this one must
 * follow it; tests/test_distributed.py's CPU stand-in does):
 *   - a state names a cell whose alignment column has NOT been emitted yet.  One step of the
 *     walk: the predecessor cell is (pi, pj) = (i - (k != 2), j - (k != 3)); the winner w among
 *     the predecessor's three tables for aggregate k (simple.cpp:162/166/170 and siblings)
 *     decides the column -- push (i-1, j-1) / (-1, j-1) / (i-1, -1) for w = 1 / 2 / 3 -- and the
 *     state becomes (pi, pj, w).  The winner is a property of the PREDECESSOR cell, so a step
 *     is taken by the stripe that owns the predecessor's column.
 *   - a stripe with columns (c0, c1] therefore accepts states with c0 < j <= c1 + 1; j == c1 + 1
 *     only with k in {1, 2} (the cell lies in the stripe on the right, its predecessor in
 *     column c1 is this stripe's).
 *   - the walk ends with i == 0 or j == 0 (matrix boundary: finished), or -- when c0 > 0 --
 *     as soon as pj <= c0: the state returned then has j == c0 + 1 and k in {1, 2} (a vertical
 *     step never leaves a stripe) and is handed, unchanged, to the stripe on the left.
 */
int psa_stripe_traceback(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai,
                         int32_t* bj, int64_t cap, int64_t* len_io, void* stream);

/* Duration in milliseconds of the most recent psa_stripe_fill_rows / psa_stripe_traceback
 * kernel of this stripe, measured with HIP events on the launch stream (synchronises). */
int psa_stripe_last_fill_ms(psa_stripe* s, float* ms_out);
int psa_stripe_last_traceback_ms(psa_stripe* s, float* ms_out);
/* Bytes of packed winner codes this stripe writes per full fill (the traceback stream). */
int64_t psa_stripe_code_bytes(const psa_stripe* s);

/* run() in linear space (the idea complex.cpp:193-689 attempts): same result as
 * psa_align, but the winner codes of only `band_rows` rows (rounded up to a multiple of 256; 0 = chosen so that
 * they take about 8 GB, less when less is free) exist at any time.  Pass 1 fills score-only (one launch) and keeps the row
 * in front of every band (8 bytes per column and band); pass 2 goes through the bands bottom-up,
 * refills one band with codes -- only the columns left of where the walk enters it -- and walks it.
 * About twice the work of psa_align, for pairs whose codes (one byte per cell) do not fit in HBM.
 * psa_align switches to it by itself when they do not, or when PSA_BAND_ROWS is set. */
int psa_align_banded(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int64_t band_rows,
                     int32_t* score_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out);

/* ---- helpers shared with the command-line front end ------------------------------------ */

/* The reference's FASTA reader (read_fasta.cpp:8-23).  Returns a buffer to release with
 * psa_free() (never NULL on success; NULL + *len_out = 0 when the file cannot be opened,
 * after printing the reference's message on stderr). */
uint8_t* psa_read_fasta(const char* path, int64_t* len_out);

/* Batch input (no counterpart in the reference, which concatenates all records of a
 * file into one sequence, read_fasta.cpp:16-19): the same line handling as psa_read_fasta, but
 * every '>' header line starts a new record.  *seqs_out = all residues back to back,
 * (*offsets_out)[r] .. [r+1] = record r, *nrec_out records; the layout psa_score_batch takes.
 * Residues before the first header form record 0 (so that the concatenation of all records always
 * equals what psa_read_fasta returns).  Both buffers are released with psa_free().  Returns PSA_OK
 * or PSA_E_ARG (file cannot be opened; the reference's message goes to stderr). */
int psa_read_fasta_records(const char* path, uint8_t** seqs_out, int64_t** offsets_out, int64_t* nrec_out);

/* ---- batch input resident on the device (no counterpart in the reference) ----------
 * A psa_seqset holds many sequences in the layout psa_score_batch_dev takes -- all residues back to
 * back in device memory plus an offsets array -- loaded from multi-record FASTA files and/or
 * directories of FASTA files: every path is a file (each '>' record one sequence, line handling as
 * psa_read_fasta_records) or a directory (every regular file in it, in byte order of the names).
 * Staged through pinned host memory; the host copy is kept (names, lengths, fallback for long pairs).
 * psa_seqset_score_block scores records [q0, q1) against records [t0, t1) -- the pair list is made on
 * the device, targets longer than 1024 residues go through psa_score -- into
 * scores_out[(q - q0) * (t1 - t0) + (t - t0)].  The reference's own CLI still concatenates all
 * records of a file (read_fasta.cpp:16-19 = psa_read_fasta). */
typedef struct psa_seqset psa_seqset;
int psa_seqset_load(psa_seqset** out, const char* const* paths, int npaths);
void psa_seqset_destroy(psa_seqset* s);
int64_t psa_seqset_count(const psa_seqset* s);
int64_t psa_seqset_length(const psa_seqset* s, int64_t i);
const char* psa_seqset_name(const psa_seqset* s, int64_t i);
const uint8_t* psa_seqset_residues(const psa_seqset* s, int64_t i);      /* host copy of record i */
const uint8_t* psa_seqset_seqs_dev(const psa_seqset* s);                 /* device: all residues */
const int64_t* psa_seqset_offsets_dev(const psa_seqset* s);              /* device: count + 1 offsets */
int psa_seqset_score_block(const psa_seqset* s, int64_t q0, int64_t q1, int64_t t0, int64_t t1,
                           int32_t* scores_out);

/* Synthetic iid-uniform protein over ACDEFGHIKLMNPQRSTVWY: out[k] = alphabet[g() % 20] with
 * std::mt19937_64 g(seed).  Host-side. */
void psa_synth_protein(uint64_t seed, int64_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
