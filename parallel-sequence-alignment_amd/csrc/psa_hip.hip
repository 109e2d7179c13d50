// psa_hip.hip -- C ABI (include/psa.h) over the gfx950 kernels.  No CPU fallback: every
// compute entry point needs a HIP device and fails with PSA_E_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <mutex>
#include <vector>

#include "psa_host.h"
#include "psa_launch.h"

namespace psa_host {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int need_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt < 1)
        return fail(PSA_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PSA_OK;
}

}  // namespace psa_host

using psa_host::fail;
using psa_host::need_device;

namespace {

// simple.cpp:162/166/170: table 1 if t1>=t2 && t1>=t3, else 2 if t2>=t3, else 3.
int pick3(const int32_t t[3], int32_t* v) {
    if (t[0] >= t[1] && t[0] >= t[2]) { *v = t[0]; return 1; }
    if (t[1] >= t[2]) { *v = t[1]; return 2; }
    *v = t[2];
    return 3;
}

constexpr size_t kSeqPad = 64;
// Control blocks of a stripe: kCtrlRing blocks of 256 bytes, one per launch in flight (word 0 the ticket
// counter, word 32 the chain front, words 48.. statistics), and one block per PASS (word 0: the abort
// flag -- a hand-off wait that expired in ANY chunk of the pass stays visible until the pass is over).
constexpr int kCtrlRing = 4;
constexpr size_t kCtrlLaunchBytes = 256;
constexpr size_t kCtrlBytes = (kCtrlRing + 1) * kCtrlLaunchBytes;
constexpr int kAbortWord = kCtrlRing * 64;
constexpr int kCornerSlots = 4;                 // triples at (r0, c0) of the chunks in flight
constexpr int kFinalWord = 4 * kCornerSlots;    // the final triple lives behind them in `small`
using psa::kAlphaBytes;

// The current HIP device is a per-thread setting.  Every call that works on an object makes the object's
// device current for its own duration and puts the caller's back on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int rc = PSA_OK;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { rc = fail(PSA_E_HIP, "hipGetDevice failed"); return; }
        if (prev != dev) {
            if (hipSetDevice(dev) != hipSuccess) { rc = fail(PSA_E_HIP, "hipSetDevice(%d) failed", dev); return; }
            switched = true;
        }
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
};

thread_local int64_t g_band_rows_request = 0;   // set by psa_align_banded around the stripe's creation

}  // namespace

struct psa_stripe {
    int device = 0;              // the HIP device the stripe lives on (current when it was created)
    uint32_t flags = 0;
    // ---- capacity: what the device buffers were sized for ----
    int64_t n_cap = 0;
    int W_cap = 0, Wpad_cap = 0, pitch_cap = 0;
    size_t code_cap = 0;
    // ---- binding: the pair (and columns of it) the stripe currently holds ----
    int64_t n = 0, m = 0, c0 = 0, c1 = 0;
    int W = 0, Q = 0, Wpad = 0, tab_pitch = 0;
    size_t code_bytes = 0;
    uint8_t* a_dev = nullptr;
    uint8_t* b_dev = nullptr;    // = b_alloc + kSeqPad: the traceback's residue windows start up to 64 bytes in front
    uint8_t* b_alloc = nullptr;
    psa::u64* ring = nullptr;
    uint8_t* codes = nullptr;    // winner codes, one byte per cell in 4 KB tiles (psa_device.hip.h)
    int2* leftcol = nullptr;     // c0 > 0: (D, H') of column c0 per row, kept for the traceback
    int32_t* small = nullptr;    // kCornerSlots corner triples (pad 4), final[3] (pad 4)
    unsigned* ctrl = nullptr;    // kCtrlBytes, see above
    unsigned* done = nullptr;    // per strip: this stripe's right boundary rows are stored (stripes filled by one launch)
    int32_t* tb_ai = nullptr;
    int32_t* tb_bj = nullptr;
    int64_t tb_cap = 0;
    int64_t* tb_state = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pass_ev = nullptr;   // recorded behind the memsets that open a pass (r0 == 0); chunks on other streams wait for it
    hipStream_t pass_stream = nullptr;
    int64_t next_r0 = 0;
    int corner_cur = 0;
    unsigned launch_no = 0;
    // v2 kernel: symbol ids of A and the per-symbol substitution table of this stripe's B
    uint8_t* a_ids = nullptr;
    uint8_t* alpha = nullptr;    // present[256] (u32) | meta[4] (u32) | map[256] | inv[256]
    // engine slots (psa_many): a pair bound since the last launch is copied and encoded by the launch's binding kernels
    const uint8_t* src_a = nullptr;   // device residues to copy from (null: already in a_dev / b_dev)
    const uint8_t* src_b = nullptr;
    bool rebind = false;
    uint32_t* eqtab = nullptr;
    unsigned nsym = 0;
    bool use_v2 = false;
    // sequences bound from DEVICE memory: the number of distinct residues comes back asynchronously (pinned word + event)
    // and is looked at when the next fill is queued (settle_tables), not when the pair is bound
    unsigned* h_nsym = nullptr;
    hipEvent_t tab_ev = nullptr;
    bool nsym_pending = false;
    bool fill_timed = false, tb_timed = false;
    // banded (linear-space) mode: codes cover band_rows rows at a time, ckpt holds the hand-off ring as it
    // was before each band (see align_banded)
    int64_t band_rows = 0;
    psa::u64* ckpt = nullptr;
};

namespace {

int tab_pitch_for(int Q) { return psa::kTabPad + (64 * Q + 64) / 4 + 40; }   // a byte per column; the last superblock requests 32 words ahead
int steps_q(int W) { return (W + 63 + 63) / 64; }

// rows per band in banded mode: the request (argument or PSA_BAND_ROWS), else as many strips as 8 GiB --
// or, when less is free, a third of the free device memory -- of codes hold
int64_t band_rows_for(int64_t n, int Q) {
    int64_t want = g_band_rows_request;
    if (want <= 0) if (const char* e = getenv("PSA_BAND_ROWS")) want = atoll(e);
    if (want <= 0) {
        size_t free_b = 0, total_b = 0;
        int64_t budget = 8LL << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<int64_t>(budget, (int64_t)(free_b / 3));
        want = 64 * std::max<int64_t>(1, budget / ((int64_t)Q * psa::kTileBytes));
    }
    const int64_t wg_rows = 64 * psa::kNW;   // band boundaries coincide with workgroup boundaries
    want = std::max<int64_t>(wg_rows, ((want + wg_rows - 1) / wg_rows) * wg_rows);
    return std::min<int64_t>(want, ((n + wg_rows - 1) / wg_rows) * wg_rows);
}

void free_stripe(psa_stripe* s) {
    if (!s) return;
    DeviceGuard g(s->device);
    hipFree(s->a_dev); hipFree(s->b_alloc); hipFree(s->ring); hipFree(s->codes); hipFree(s->leftcol);
    hipFree(s->a_ids); hipFree(s->alpha); hipFree(s->eqtab);
    hipFree(s->ckpt); hipFree(s->done);
    hipFree(s->small); hipFree(s->ctrl); hipFree(s->tb_ai); hipFree(s->tb_bj); hipFree(s->tb_state);
    for (auto& e : s->ev) if (e) hipEventDestroy(e);
    if (s->pass_ev) hipEventDestroy(s->pass_ev);
    if (s->tab_ev) hipEventDestroy(s->tab_ev);
    hipHostFree(s->h_nsym);
    delete s;
}

// Device buffers for any stripe of up to n_cap rows and W_cap columns (banded: sized for exactly this pair).
int alloc_stripe(psa_stripe** out, int64_t n_cap, int W_cap, uint32_t flags, bool with_leftcol) {
    psa_stripe* s = new psa_stripe();
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    s->flags = flags;
    s->n_cap = n_cap; s->W_cap = W_cap;
    s->Wpad_cap = ((W_cap + 63) / 64) * 64;
    const int Q_cap = steps_q(W_cap);
    s->pitch_cap = tab_pitch_for(Q_cap);
    const int64_t strips = (n_cap + 63) / 64;
#define CREATE_TRY(expr)                                                          \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            free_stripe(s);                                                       \
            return fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));           \
        }                                                                         \
    } while (0)
    // (both padded: the traceback fetches whole 64-row / 128-column residue windows beside each code tile)
    CREATE_TRY(hipMalloc(&s->a_dev, (size_t)n_cap + 2 * kSeqPad));
    CREATE_TRY(hipMalloc(&s->b_alloc, (size_t)W_cap + 4 * kSeqPad));
    s->b_dev = s->b_alloc + kSeqPad;
    CREATE_TRY(hipMalloc(&s->ring, sizeof(psa::u64) * (2 * (size_t)s->Wpad_cap + 64)));  // + dump slots, see fill_strips_v2
    CREATE_TRY(hipMalloc(&s->small, 256));
    CREATE_TRY(hipMalloc(&s->ctrl, kCtrlBytes));
    CREATE_TRY(hipMalloc(&s->done, sizeof(unsigned) * (size_t)(strips + 64)));
    if (flags & PSA_F_TRACEBACK) {
        int64_t code_strips = strips;
        if (flags & PSA_F_BANDED) {
            s->band_rows = band_rows_for(n_cap, Q_cap);
            code_strips = s->band_rows / 64;
            const int64_t nbands = (n_cap + s->band_rows - 1) / s->band_rows;
            CREATE_TRY(hipMalloc(&s->ckpt, sizeof(psa::u64) * (size_t)(nbands + 1) * (size_t)s->Wpad_cap));
        }
        s->code_cap = (size_t)code_strips * (size_t)Q_cap * psa::kTileBytes;
        CREATE_TRY(hipMalloc(&s->codes, s->code_cap));
        if (with_leftcol) CREATE_TRY(hipMalloc(&s->leftcol, sizeof(int2) * (size_t)(n_cap + 1)));
        s->tb_cap = n_cap + W_cap + 2;
        CREATE_TRY(hipMalloc(&s->tb_ai, sizeof(int32_t) * (size_t)s->tb_cap));
        CREATE_TRY(hipMalloc(&s->tb_bj, sizeof(int32_t) * (size_t)s->tb_cap));
        CREATE_TRY(hipMalloc(&s->tb_state, sizeof(int64_t) * 16));
    }
    for (auto& e : s->ev) CREATE_TRY(hipEventCreate(&e));
    CREATE_TRY(hipEventCreateWithFlags(&s->pass_ev, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&s->tab_ev, hipEventDisableTiming));
    CREATE_TRY(hipHostMalloc(&s->h_nsym, 64));
    CREATE_TRY(hipMalloc(&s->a_ids, (size_t)n_cap + 64));
    CREATE_TRY(hipMalloc(&s->alpha, kAlphaBytes));
    CREATE_TRY(hipMalloc(&s->eqtab, sizeof(uint32_t) * (size_t)psa::kMaxSym * (size_t)s->pitch_cap));
    CREATE_TRY(hipMemset(s->ctrl, 0, kCtrlBytes));
    CREATE_TRY(hipMemset(s->alpha, 0, kAlphaBytes));
#undef CREATE_TRY
    *out = s;
    return PSA_OK;
}

// Point the stripe at columns (c0, c1] of an n x m pair (within its capacity).
int bind_stripe(psa_stripe* s, int64_t n, int64_t m, int64_t c0, int64_t c1) {
    const int W = (int)(c1 - c0);
    if (n < 1 || n > s->n_cap || W < 1 || W > s->W_cap || c0 < 0 || c1 > m)
        return fail(PSA_E_ARG, "stripe of %lld rows x %d columns does not fit its buffers (%lld x %d)", (long long)n, W,
                    (long long)s->n_cap, s->W_cap);
    if (c0 > 0 && (s->flags & PSA_F_TRACEBACK) && !s->leftcol) return fail(PSA_E_STATE, "stripe was allocated without a left boundary column");
    s->n = n; s->m = m; s->c0 = c0; s->c1 = c1;
    s->W = W; s->Q = steps_q(W); s->Wpad = ((W + 63) / 64) * 64; s->tab_pitch = tab_pitch_for(s->Q);
    if (s->flags & PSA_F_TRACEBACK) {
        const int64_t strips = (n + 63) / 64;
        s->code_bytes = (size_t)(s->band_rows ? s->band_rows / 64 : strips) * (size_t)s->Q * psa::kTileBytes;
    }
    s->next_r0 = 0; s->corner_cur = 0;
    s->fill_timed = s->tb_timed = false;
    return PSA_OK;
}

// FASTA-to-device encoding for the v2 kernel: symbol ids of A, substitution table of B.  nsym_known >= 0:
// the caller has counted A's distinct residues itself (host sequences) and no device round trip is needed.
int rebuild_tables(psa_stripe* s, hipStream_t st, int nsym_known) {
    unsigned* meta = psa::alpha_meta(s->alpha);
    HIP_TRY(psa::launch_tables(s->a_dev, s->n, s->b_dev, s->W, s->alpha, s->a_ids, s->eqtab, s->tab_pitch, st));
    if (nsym_known < 0) {
        // (no host round trip here: binding a pair must not wait for kernels that may be queued behind a fill which
        // holds every SIMD of the chip)
        HIP_TRY(hipMemcpyAsync(s->h_nsym, meta, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(s->tab_ev, st));
        s->nsym_pending = true;
        return PSA_OK;
    }
    s->nsym_pending = false;
    s->nsym = (unsigned)nsym_known;
    const char* force = getenv("PSA_FILL_KERNEL");
    s->use_v2 = s->nsym <= (unsigned)psa::kMaxSym && !(force && !strcmp(force, "v1"));
    return PSA_OK;
}

// Which fill kernel the stripe's current sequences take (the table kernel, unless A has more distinct residues than the
// table has rows -- kMaxSym = 256: never -- or PSA_FILL_KERNEL=v1 asks for the generic kernel).
int settle_tables(psa_stripe* s) {
    if (!s->nsym_pending) return PSA_OK;
    HIP_TRY(hipEventSynchronize(s->tab_ev));
    s->nsym_pending = false;
    s->nsym = *s->h_nsym;
    const char* force = getenv("PSA_FILL_KERNEL");
    s->use_v2 = s->nsym <= (unsigned)psa::kMaxSym && !(force && !strcmp(force, "v1"));
    return PSA_OK;
}

int count_symbols(const uint8_t* a, int64_t n) {
    bool seen[256] = {false};
    int k = 0;
    for (int64_t i = 0; i < n; ++i) if (!seen[a[i]]) { seen[a[i]] = true; ++k; }
    return k;
}

int upload_host(psa_stripe* s, const uint8_t* a, const uint8_t* b, hipStream_t st) {
    HIP_TRY(hipMemcpyAsync(s->a_dev, a, (size_t)s->n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->b_dev, b + s->c0, (size_t)s->W, hipMemcpyHostToDevice, st));
    return rebuild_tables(s, st, count_symbols(a, s->n));
}

// what one fill launch may override (banded mode)
struct FillOpts {
    int codes = -1;                  // -1: as the stripe was created; 0 / 1: without / with winner codes
    int W = 0;                       // > 0: only the stripe's first W columns
    uint8_t* codes_base = nullptr;   // where strip 0's codes would lie
    bool any_order = false;          // do not insist on r0 == end of the previous chunk
    const unsigned* left_done = nullptr;   // the stripe on the left is filled by the same launch (see FillArgs)
    unsigned* right_done = nullptr;
    psa::BindJob* clear_into = nullptr;    // not null: the caller zeroes what the fill must find zeroed (one kernel over all its
                                           // stripes: launch_bind) -- the regions are listed here instead of memsets on the stream
};

// Everything a fill of rows r0+1..r1 needs before its kernel can be launched (memsets on `st`), and the kernel
// arguments.  Does not launch.
int prepare_fill(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev, hipStream_t st,
                 const FillOpts& o, psa::FillArgs2* out, bool* with_codes_out) {
    if (r0 == 0) s->next_r0 = 0;  // a new pass over the same stripe
    if (o.any_order) s->next_r0 = r0;
    if (r0 != s->next_r0) return fail(PSA_E_STATE, "psa_stripe_fill_rows: chunks out of order (r0=%lld, expected %lld)",
                                      (long long)r0, (long long)s->next_r0);
    if ((r0 & 63) || r1 <= r0 || r1 > s->n)
        return fail(PSA_E_ARG, "psa_stripe_fill_rows: need r0 %% 64 == 0 and r0 < r1 <= n (r0=%lld r1=%lld n=%lld)",
                    (long long)r0, (long long)r1, (long long)s->n);
    if ((s->c0 == 0) != (left_dev == nullptr))
        return fail(PSA_E_ARG, "psa_stripe_fill_rows: left boundary must be given exactly when c0 > 0");
    if (r0 == 0 || o.any_order) {
        // A launch that does not continue the previous one must not find that one's granules: a granule
        // is recognised by its tag (strip number), and a strip that finds a left-over with the right tag
        // runs ahead of the strips it should follow and overwrites ring slots they have not read yet.
        // (A banded refill takes its input from the checkpoint row, never from the ring.)
        if (o.clear_into) {
            o.clear_into->clear[0] = s->ring; o.clear_into->clear_bytes[0] = (unsigned)(sizeof(psa::u64) * (2 * (size_t)s->Wpad + 64));
            o.clear_into->clear[1] = s->ctrl + kAbortWord; o.clear_into->clear_bytes[1] = (unsigned)kCtrlLaunchBytes;
            if (r0 == 0) { s->corner_cur = 0; o.clear_into->clear[2] = s->small; o.clear_into->clear_bytes[2] = 256; }
            s->pass_stream = st;   // (no later chunks on other streams in this mode)
        } else {
        HIP_TRY(hipMemsetAsync(s->ring, 0, sizeof(psa::u64) * (2 * (size_t)s->Wpad + 64), st));
        if (r0 == 0) {
            s->corner_cur = 0;
            HIP_TRY(hipMemsetAsync(s->small, 0, 256, st));   // corner slots are recognised by their tag (a row number): none may survive a pass
        }
        HIP_TRY(hipMemsetAsync(s->ctrl + kAbortWord, 0, kCtrlLaunchBytes, st));
        // Later chunks of this pass may be launched on OTHER streams (chunks in flight on alternating streams): they
        // must not start before these memsets have run -- a strip of chunk 1 that finds a granule or corner tag left by
        // the previous pass would run ahead of the strips it should follow, and a late memset would wipe what chunk 1
        // has published.
        HIP_TRY(hipEventRecord(s->pass_ev, st));
        s->pass_stream = st;
        }
    } else if (st != s->pass_stream) {
        HIP_TRY(hipStreamWaitEvent(st, s->pass_ev, 0));
    }
    // a control block of its own for every launch: the previous chunk's launch may still be running
    unsigned* launch_ctrl = s->ctrl + (s->launch_no++ % kCtrlRing) * (kCtrlLaunchBytes / sizeof(unsigned));
    if (o.clear_into) { o.clear_into->clear[3] = launch_ctrl; o.clear_into->clear_bytes[3] = (unsigned)kCtrlLaunchBytes; }
    else HIP_TRY(hipMemsetAsync(launch_ctrl, 0, kCtrlLaunchBytes, st));
    psa::FillArgs fa;
    fa.a = s->a_dev; fa.b = s->b_dev; fa.n = s->n; fa.r0 = r0; fa.r1 = r1; fa.c0 = s->c0;
    fa.W = o.W > 0 ? o.W : s->W; fa.Q = o.W > 0 ? steps_q(o.W) : s->Q; fa.Wpad = s->Wpad;
    const bool with_codes = o.codes < 0 ? s->codes != nullptr : o.codes != 0;
    fa.left = left_dev;
    fa.corner = s->small + 4 * s->corner_cur;
    fa.corner_next = s->small + 4 * ((s->corner_cur + 1) % kCornerSlots);
    fa.right = right_dev;
    fa.left_done = o.left_done; fa.right_done = o.right_done;
    fa.final_triple = s->small + kFinalWord;
    fa.ring = s->ring;
    fa.codes = with_codes ? (o.codes_base ? o.codes_base : s->codes) : nullptr;
    fa.leftcol = with_codes ? s->leftcol : nullptr;
    fa.ticket = launch_ctrl;
    fa.abort_flag = s->ctrl + kAbortWord;
    fa.prof = nullptr;
    fa.ckpt = s->ckpt; fa.band_strips = s->band_rows / 64;
    out->base = fa; out->a_ids = s->a_ids; out->eqtab = s->eqtab; out->base.tab_pitch = s->tab_pitch;
    *with_codes_out = with_codes;
    s->corner_cur = (s->corner_cur + 1) % kCornerSlots;
    s->next_r0 = r1;
    return PSA_OK;
}

unsigned groups_of(const psa::FillArgs2& j) {
    const int64_t strips = (j.base.r1 - j.base.r0 + 63) / 64;
    return (unsigned)((strips + psa::kNW - 1) / psa::kNW);
}

// One launch of fill_strips_v2 over up to kMaxJobs stripes.  order_dev: ticket -> job << 24 | group (null: one job).
int launch_fill_v2(const psa::FillArgs2* jobs, int njobs, const uint32_t* order_dev, unsigned total_groups,
                   unsigned* ticket_dev, bool with_codes, size_t keepout_bytes, hipStream_t st, bool dense = false) {
    bool edges = false;   // a right boundary somewhere: the kernel whose capturing strips take a second pass
    for (int j = 0; j < njobs; ++j) edges = edges || jobs[j].base.right != nullptr;
    psa::FillBatch fb;
    memset(&fb, 0, sizeof fb);
    for (int j = 0; j < njobs; ++j) fb.job[j] = jobs[j];
    fb.njobs = njobs; fb.order = order_dev; fb.ticket = ticket_dev;
    // independent pairs: two workgroups per CU (PSA_FILL_DENSE=0: the one-wave-per-SIMD kernel)
    const int kernel = dense ? psa::kFillDense : edges ? psa::kFillEdges : psa::kFillSparse;
    if (with_codes) HIP_TRY(psa::launch_fill_v2_codes(fb, total_groups, kernel, keepout_bytes, st));
    else HIP_TRY(psa::launch_fill_v2_score(fb, total_groups, kernel, keepout_bytes, st));
    return PSA_OK;
}

int fill_rows_impl(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev, void* stream,
                   const FillOpts& o) {
    if (!s) return fail(PSA_E_ARG, "psa_stripe_fill_rows: null stripe");
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    hipStream_t st = (hipStream_t)stream;
    psa::FillArgs2 fb;
    bool with_codes = false;
    if (int rc = settle_tables(s)) return rc;
    if (int rc = prepare_fill(s, r0, r1, left_dev, right_dev, st, o, &fb, &with_codes)) return rc;
    const char* prof_path = getenv("PSA_PROFILE_STRIPS");
    psa::u64* prof_dev = nullptr;
    const unsigned strips = (unsigned)((r1 - r0 + 63) / 64);
    if (prof_path && s->use_v2) {
        size_t prof_words = 4 * (size_t)strips;
        HIP_TRY(hipMalloc(&prof_dev, sizeof(psa::u64) * prof_words));
        HIP_TRY(hipMemsetAsync(prof_dev, 0, sizeof(psa::u64) * prof_words, st));
        fb.base.prof = prof_dev;
    }
    HIP_TRY(hipEventRecord(s->ev[0], st));
    if (s->use_v2) {
        // (a flagged fill keeps every CU occupied with strips that wait for their boundary: the smaller keep-out leaves
        // the kernels that bring it -- collective library, psa_edge_post -- 77 KB of every CU's LDS)
        const size_t keepout = (o.left_done || o.right_done) ? psa::kKeepoutManyBytes : psa::kKeepoutAloneBytes;
        // One pair alone gets one wavefront per SIMD because its strips are links of ONE chain -- as long as the chain
        // is what bounds it.  A pair with many more strips than SIMDs and rows long enough for all of them to be in
        // flight together (2^20 x 2^20: 16 384 strips, each 378 steps behind the one above and a million steps long)
        // is bound by instruction issue like a launch of independent pairs: two wavefronts per SIMD
        // (fill_strips_v2_dense).  Whole-stripe launches only: row chunks in flight on two streams must fit the chip
        // together (distributed.auto_chunk_rows).
        static const int64_t dense_min_w = [] { const char* e = getenv("PSA_FILL_DENSE_MIN_W"); return e ? atoll(e) : (1LL << 19); }();
        const unsigned groups = groups_of(fb);
        const bool dense = r0 == 0 && r1 == s->n && !o.left_done && !o.right_done && !right_dev && !prof_dev &&
                           groups >= 1024 && fb.base.W >= dense_min_w;
        if (int rc = launch_fill_v2(&fb, 1, nullptr, groups, fb.base.ticket, with_codes, keepout, st, dense)) return rc;
    } else if (with_codes) HIP_TRY(psa::launch_fill_v1_codes(fb.base, strips, st));
    else HIP_TRY(psa::launch_fill_v1_score(fb.base, strips, st));
    HIP_TRY(hipEventRecord(s->ev[1], st));
    s->fill_timed = true;
    if (prof_dev) {  // diagnostic only: dump per-strip start/end ticks and placement
        std::vector<psa::u64> h(4 * (size_t)strips);
        HIP_TRY(hipStreamSynchronize(st));   // (a stream created non-blocking is not ordered against the copy below)
        HIP_TRY(hipMemcpy(h.data(), prof_dev, sizeof(psa::u64) * h.size(), hipMemcpyDeviceToHost));
        std::string pp = prof_path;
        if (!pp.empty() && pp.back() == '+') {   // one file per stripe object
            char suffix[32];
            snprintf(suffix, sizeof suffix, "%p", (void*)s);
            pp.pop_back();
            pp += suffix;
        }
        if (FILE* f = fopen(pp.c_str(), "w")) {
            fprintf(f, "strip,start,end,xcc,hwid,wait_in,wait_room\n");
            for (unsigned k = 0; k < strips; ++k)
                fprintf(f, "%u,%lld,%lld,%llu,%llu,%llu,%llu\n", k, (long long)(h[4 * k] - h[0]),
                        (long long)(h[4 * k + 1] - h[0]), h[4 * k + 2] >> 32, h[4 * k + 2] & 0xffffffffu,
                        h[4 * k + 3] & 0xffffffffu, h[4 * k + 3] >> 32);
            fclose(f);
        }
        hipFree(prof_dev);
    }
    return PSA_OK;
}

int check_abort(psa_stripe* s, hipStream_t st) {
    unsigned ab = 0;
    HIP_TRY(hipMemcpyAsync(&ab, s->ctrl + kAbortWord, sizeof ab, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ab) return fail(PSA_E_KERNEL, "fill kernel: a strip's wait for the strip above (or for the stripe on its left) expired");
    return PSA_OK;
}

struct TraceOpts {
    const uint8_t* codes_base = nullptr;    // banded mode: where strip 0's codes would lie
    int W = 0;                              // > 0: the codes were written by a fill of the first W columns
    int64_t row_min = 0;                    // the codes cover rows > row_min only
};

int make_trace_args(psa_stripe* s, const TraceOpts& o, psa::TraceArgs* ta) {
    ta->codes = o.codes_base ? o.codes_base : s->codes; ta->c0 = s->c0; ta->cap = s->tb_cap;
    ta->a = s->a_dev; ta->b = s->b_dev; ta->leftcol = s->leftcol;
    ta->toprow = (o.row_min > 0 && s->band_rows > 0) ? s->ckpt + (size_t)(o.row_min / s->band_rows) * (size_t)s->Wpad : nullptr;
    if (o.row_min > 0 && !ta->toprow) return fail(PSA_E_STATE, "traceback: a row limit needs the checkpoint row of a banded stripe");
    ta->W = o.W > 0 ? o.W : s->W; ta->Q = o.W > 0 ? steps_q(o.W) : s->Q;
    ta->row_min = o.row_min;
    ta->out_ai = s->tb_ai; ta->out_bj = s->tb_bj; ta->state = s->tb_state;
    ta->n = s->n; ta->start_triple = nullptr;
    return PSA_OK;
}

// the fast walk keeps its state in 32-bit registers: coordinates and tile ids must fit
bool walk_fits32(const psa_stripe* s, int Q) {
    return s->n < 0x7fff0000LL && s->c0 + s->W < 0x7fff0000LL && ((s->n + 63) / 64) * (int64_t)Q < 0x7fff0000LL;
}

int traceback_impl(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai, int32_t* bj,
                   int64_t cap, int64_t* len_io, void* stream, const TraceOpts& o) {
    if (!s || !i_io || !j_io || !k_io || !ai || !bj || !len_io) return fail(PSA_E_ARG, "psa_stripe_traceback: null argument");
    if (!s->codes) return fail(PSA_E_STATE, "psa_stripe_traceback: stripe was created without PSA_F_TRACEBACK");
    if (*k_io < 1 || *k_io > 3 || *i_io < 0 || *i_io > s->n || *j_io < 0 || *j_io > s->c1 + 1)
        return fail(PSA_E_ARG, "psa_stripe_traceback: bad state (%lld,%lld,%d)", (long long)*i_io, (long long)*j_io, *k_io);
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    if (int rc = check_abort(s, st)) return rc;
    int64_t state[16] = {*i_io, *j_io, *k_io, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(s->tb_state, state, sizeof state, hipMemcpyHostToDevice, st));
    psa::TraceArgs ta;
    if (int rc = make_trace_args(s, o, &ta)) return rc;
    HIP_TRY(hipEventRecord(s->ev[2], st));
    {
        const char* force = getenv("PSA_TRACEBACK_KERNEL");
        const int kind = ((force && !strcmp(force, "v1")) || !walk_fits32(s, ta.Q)) ? psa::kWalkLane
                         : (force && !strcmp(force, "wave1")) ? psa::kWalkWave : psa::kWalkPair;
        HIP_TRY(psa::launch_traceback(kind, ta, st));
    }
    HIP_TRY(hipEventRecord(s->ev[3], st));
    s->tb_timed = true;
    HIP_TRY(hipMemcpyAsync(state, s->tb_state, sizeof state, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (getenv("PSA_TRACE_STATS")) {
        fprintf(stderr, "traceback: len %lld passes %lld tiles %lld single-cell steps %lld misses %lld\n", (long long)state[4],
                (long long)(state[6] >> 32), (long long)(state[6] & 0xffffffff), (long long)(state[7] >> 32), (long long)(state[7] & 0xffffffff));
        fprintf(stderr, "traceback: core cycles %lld, 100 MHz ticks %lld (the counts above need a -DPSA_TB_STATS build; fetcher: blocked %lld rounds %lld operations %lld)\n", (long long)state[8], (long long)state[9],
                (long long)(state[10] >> 32), (long long)(state[10] & 0xffffffff), (long long)state[11]);
        fprintf(stderr, "traceback: one-cell steps by what bounded the rectangle: row above %lld, column to the left %lld, the diagonal's own length %lld\n",
                (long long)state[12], (long long)state[13], (long long)state[14]);
    }
    if (state[5] == 2) return fail(PSA_E_KERNEL, "traceback: winner codes failed the decode check (values that cannot belong to one cell)");
    if (state[5] == 3) return fail(PSA_E_KERNEL, "traceback: the walk waited in vain for a code tile (fetching wavefront stalled)");
    if (state[5]) return fail(PSA_E_KERNEL, "traceback: device buffer overflow");
    const int64_t got = state[4];
    if (*len_io + got > cap) return fail(PSA_E_ARG, "psa_stripe_traceback: output capacity %lld too small", (long long)cap);
    if (got) {
        HIP_TRY(hipMemcpyAsync(ai + *len_io, s->tb_ai, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(bj + *len_io, s->tb_bj, sizeof(int32_t) * (size_t)got, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *len_io += got;
    *i_io = state[0]; *j_io = state[1]; *k_io = (int)state[2];
    if (!state[3] && s->c0 == 0 && o.row_min == 0)
        return fail(PSA_E_KERNEL, "traceback: head stripe stopped before the matrix boundary");
    return PSA_OK;
}

}  // namespace

extern "C" {

const char* psa_last_error(void) { return psa_host::g_err.c_str(); }
const char* psa_version(void) { return "psa-hip 0.2 (gfx950)"; }

int psa_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int psa_set_device(int device) {
    if (int rc = need_device()) return rc;
    HIP_TRY(hipSetDevice(device));
    return PSA_OK;
}

void psa_free(void* p) { free(p); }

void psa_stripe_destroy(psa_stripe* s) { free_stripe(s); }

int psa_stripe_create(psa_stripe** out, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                      int64_t c0, int64_t c1, uint32_t flags) {
    if (!out) return fail(PSA_E_ARG, "psa_stripe_create: out is null");
    *out = nullptr;
    if (n < 1 || m < 1 || c0 < 0 || c1 <= c0 || c1 > m || !a || !b)
        return fail(PSA_E_ARG, "psa_stripe_create: need n>=1, m>=1, 0<=c0<c1<=m (n=%lld m=%lld c0=%lld c1=%lld)",
                    (long long)n, (long long)m, (long long)c0, (long long)c1);
    if (n > (1LL << 27) || m > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
    if (int rc = need_device()) return rc;
    psa_stripe* s = nullptr;
    if (int rc = alloc_stripe(&s, n, (int)(c1 - c0), flags, c0 > 0)) return rc;
    int rc = bind_stripe(s, n, m, c0, c1);
    if (rc == PSA_OK) rc = upload_host(s, a, b, nullptr);
    if (rc == PSA_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(PSA_E_HIP, "psa_stripe_create: upload failed");
    if (rc != PSA_OK) { free_stripe(s); return rc; }
    *out = s;
    return PSA_OK;
}

int psa_stripe_set_sequences_dev(psa_stripe* s, const uint8_t* a_dev, const uint8_t* b_dev, void* stream) {
    if (!s || !a_dev || !b_dev) return fail(PSA_E_ARG, "psa_stripe_set_sequences_dev: null argument");
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(s->a_dev, a_dev, (size_t)s->n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->b_dev, b_dev, (size_t)s->W, hipMemcpyDeviceToDevice, st));
    return rebuild_tables(s, st, -1);
}

int psa_stripe_fill_rows(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev,
                         void* stream) {
    if (s && s->band_rows) return fail(PSA_E_STATE, "psa_stripe_fill_rows: a banded stripe is driven by psa_align_banded");
    return fill_rows_impl(s, r0, r1, left_dev, right_dev, stream, FillOpts());
}

// ---- one launch over all rows, boundaries arriving and leaving by flags (the multi-GPU pipeline) ----
int psa_stripe_fill_flagged(psa_stripe* s, const int32_t* left_dev, const uint32_t* left_ready, int32_t* right_dev,
                            uint32_t* right_done, void* stream) {
    if (!s) return fail(PSA_E_ARG, "psa_stripe_fill_flagged: null stripe");
    if (s->band_rows) return fail(PSA_E_STATE, "psa_stripe_fill_flagged: a banded stripe is driven by psa_align_banded");
    if (int rc = settle_tables(s)) return rc;
    if (!s->use_v2) return fail(PSA_E_STATE, "psa_stripe_fill_flagged: needs the table fill kernel (PSA_FILL_KERNEL=v1 is set)");
    if ((left_ready && !left_dev) || (right_done && !right_dev))
        return fail(PSA_E_ARG, "psa_stripe_fill_flagged: flags without their boundary buffer");
    FillOpts o;
    o.left_done = left_ready;
    o.right_done = right_done;
    return fill_rows_impl(s, 0, s->n, left_dev, right_dev, stream, o);
}

int psa_stream_create(void** stream_out, int priority_class, int reserved_cus) {
    if (!stream_out) return fail(PSA_E_ARG, "psa_stream_create: null argument");
    hipStream_t st = nullptr;
    int dev = 0;
    hipDeviceProp_t pr;
    if (reserved_cus > 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess &&
        pr.multiProcessorCount > reserved_cus) {
        const int cus = pr.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
        for (int c = 0; c < cus - reserved_cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            st = nullptr;
        }
    }
    if (!st) {
        int least = 0, greatest = 0;   // (numerically: least = the lowest priority)
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int prio = priority_class < 0 ? least : priority_class > 0 ? greatest : (least + greatest) / 2;
        HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
    }
    *stream_out = (void*)st;
    return PSA_OK;
}

int psa_stream_destroy(void* stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return PSA_OK;
}

int psa_edge_post(const int32_t* packed_dev, int64_t r0, int64_t rows, int64_t n, int32_t* left_dev, uint32_t* left_ready,
                  void* stream) {
    if (!packed_dev || !left_dev || !left_ready || rows < 1 || r0 < 0 || (r0 & 63) || r0 + rows > n)
        return fail(PSA_E_ARG, "psa_edge_post: bad arguments (r0=%lld rows=%lld n=%lld; r0 must be a multiple of 64)",
                    (long long)r0, (long long)rows, (long long)n);
    if (r0 + rows < n && (rows & 63)) return fail(PSA_E_ARG, "psa_edge_post: only the last chunk may have a row count off 64");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(psa::launch_edge_post(packed_dev, r0, rows, n, left_dev, left_ready, st));
    return PSA_OK;
}

int psa_edge_take(psa_stripe* s, const int32_t* right_dev, const uint32_t* right_done, int64_t r0, int64_t rows,
                  int32_t* packed_dev, void* stream) {
    if (!s || !right_dev || !right_done || !packed_dev || rows < 1 || r0 < 0 || (r0 & 63) || r0 + rows > s->n)
        return fail(PSA_E_ARG, "psa_edge_take: bad arguments (r0=%lld rows=%lld n=%lld; r0 must be a multiple of 64)",
                    (long long)r0, (long long)rows, s ? (long long)s->n : 0LL);
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(psa::launch_edge_take(right_dev, right_done, r0, rows, s->n, packed_dev, s->ctrl + kAbortWord, st));
    return PSA_OK;
}

int psa_stripe_final(psa_stripe* s, void* stream, int32_t triple_out[3]) {
    if (!s || !triple_out) return fail(PSA_E_ARG, "psa_stripe_final: null argument");
    if (s->next_r0 != s->n) return fail(PSA_E_STATE, "psa_stripe_final: rows filled up to %lld of %lld",
                                        (long long)s->next_r0, (long long)s->n);
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    if (int rc = check_abort(s, st)) return rc;
    HIP_TRY(hipMemcpyAsync(triple_out, s->small + kFinalWord, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));   // (on the stripe's own stream: the null stream would order it against other stripes' work)
    return PSA_OK;
}

int psa_stripe_traceback(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai, int32_t* bj,
                         int64_t cap, int64_t* len_io, void* stream) {
    if (s && s->band_rows) return fail(PSA_E_STATE, "psa_stripe_traceback: a banded stripe is driven by psa_align_banded");
    if (s && s->next_r0 != s->n) return fail(PSA_E_STATE, "psa_stripe_traceback: fill is not complete");
    return traceback_impl(s, i_io, j_io, k_io, ai, bj, cap, len_io, stream, TraceOpts());
}

int psa_stripe_last_fill_ms(psa_stripe* s, float* ms_out) {
    if (!s || !ms_out || !s->fill_timed) return fail(PSA_E_STATE, "no fill has been timed");
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    HIP_TRY(hipEventSynchronize(s->ev[1]));
    HIP_TRY(hipEventElapsedTime(ms_out, s->ev[0], s->ev[1]));
    return PSA_OK;
}

int psa_stripe_last_traceback_ms(psa_stripe* s, float* ms_out) {
    if (!s || !ms_out || !s->tb_timed) return fail(PSA_E_STATE, "no traceback has been timed");
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    HIP_TRY(hipEventSynchronize(s->ev[3]));
    HIP_TRY(hipEventElapsedTime(ms_out, s->ev[2], s->ev[3]));
    return PSA_OK;
}

int64_t psa_stripe_code_bytes(const psa_stripe* s) { return s ? (int64_t)s->code_bytes : 0; }

// simple.cpp:279-290 with find_last (simple.cpp:24-41).
int psa_finish_alignment(int32_t* ai, int32_t* bj, int64_t len, int64_t cap, int64_t* len_out) {
    if (!ai || !bj || !len_out || len < 0) return fail(PSA_E_ARG, "psa_finish_alignment: bad argument");
    int64_t fi = 0, fj = 0;  // std::pair value-initialises to 0 when nothing is found
    for (int64_t t = len - 1; t >= 0; --t) if (ai[t] != -1) { fi = ai[t]; break; }
    for (int64_t t = len - 1; t >= 0; --t) if (bj[t] != -1) { fj = bj[t]; break; }
    if (len + fi + fj > cap) return fail(PSA_E_ARG, "psa_finish_alignment: capacity too small");
    for (int64_t x = fi; x > 0; --x) { ai[len] = (int32_t)(x - 1); bj[len] = -1; ++len; }
    for (int64_t x = fj; x > 0; --x) { ai[len] = -1; bj[len] = (int32_t)(x - 1); ++len; }
    std::reverse(ai, ai + len);
    std::reverse(bj, bj + len);
    *len_out = len;
    return PSA_OK;
}

}  // extern "C"

namespace {

// ---- the column stripes of one pair on one GPU (simple.cpp:90-99, 261-277) ----
struct StripeSet {
    int device = -1;
    int64_t n = 0, m = 0;
    int p = 0;
    uint32_t flags = 0;
    std::vector<psa_stripe*> st;
    std::vector<int32_t*> edge;  // device boundary buffers between consecutive stripes: int32[3][n]
    uint32_t* order = nullptr;   // ticket order of the one-launch fill
    size_t order_cap = 0;
    void clear() {
        for (auto* s : st) free_stripe(s);
        for (auto* e : edge) hipFree(e);
        hipFree(order);
        st.clear(); edge.clear(); order = nullptr; order_cap = 0; p = 0;
    }
};

// The stripe objects of the most recent whole-pair call of this thread are kept and reused when the next
// call has the same shape (sizes, p, flags, device): no allocation on the second ./simple-style call.
// Never freed by a destructor (the HIP runtime may be gone by then); psa_release_cache() frees it.
thread_local StripeSet* g_cache = nullptr;
}  // namespace
void psa_drop_multi_cache_impl();   // (defined with the multi-GPU calls below)
void psa_drop_many_cache_impl();    // (defined with psa_align_many below)
// psa_align_many's engines go when an allocation of `need` bytes would take more than 70 % of what is free beside them
// (freeing and rebuilding them costs seconds, so not for a pair that fits anyway).
void psa_make_room(double need);
namespace {
inline void drop_multi_fwd() { psa_drop_multi_cache_impl(); }

int64_t chunk_rows_env() {
    if (const char* e = getenv("PSA_CHUNK_ROWS")) {
        long long v = atoll(e);
        if (v >= 64) return (v / 64) * 64;
    }
    return 0;
}

int get_stripes(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p, uint32_t flags, StripeSet** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!g_cache) g_cache = new StripeSet();
    StripeSet& set = *g_cache;
    const bool hit = set.device == dev && set.n == n && set.m == m && set.p == p && set.flags == flags && !set.st.empty();
    if (!hit) {
        set.clear();
        psa_make_room(((flags & PSA_F_TRACEBACK) ? 1.05 * (double)n * (double)m : 0.0) + 64.0 * (double)(n + m) * p + (double)(64 << 20));
        set.device = dev; set.n = n; set.m = m; set.flags = flags;
        const int64_t block = m / p;   // column stripes exactly as simple.cpp:90-99 cuts them; zero-width stripes (p > m) own no cell
        for (int id = 0; id < p; ++id) {
            const int64_t c0 = (int64_t)id * block;
            const int64_t c1 = (id == p - 1) ? m : c0 + block;
            if (c1 == c0) continue;
            psa_stripe* s = nullptr;
            int rc = alloc_stripe(&s, n, (int)(c1 - c0), flags, c0 > 0);
            if (rc == PSA_OK) rc = bind_stripe(s, n, m, c0, c1);
            if (rc != PSA_OK) { free_stripe(s); set.clear(); return rc; }
            set.st.push_back(s);
        }
        for (size_t g = 0; g + 1 < set.st.size(); ++g) {
            int32_t* e = nullptr;
            if (hipMalloc(&e, sizeof(int32_t) * 3 * (size_t)n) != hipSuccess) { set.clear(); return fail(PSA_E_NOMEM, "boundary buffer allocation failed"); }
            set.edge.push_back(e);
        }
        set.p = p;
    }
    for (auto* s : set.st) {
        if (int rc = bind_stripe(s, n, m, s->c0, s->c1)) return rc;
        if (int rc = upload_host(s, a, b, nullptr)) return rc;
    }
    *out = &set;
    return PSA_OK;
}

// Fill every stripe.  Default: ONE launch per group of up to kMaxJobs stripes -- the stripes are jobs of the
// same launch, stripe g+1 picks up the rows of its left boundary as stripe g's strips finish them (flags in
// device memory), and the ticket order lets it trail by about the time a strip takes.  With PSA_CHUNK_ROWS
// set (tests), or for alphabets the table kernel does not take: one launch per stripe and row chunk, in
// stream order, as a chain of GPUs would run them.
int run_stripes(StripeSet& set, int32_t triple[3]) {
    const size_t ns = set.st.size();
    const int64_t n = set.n;
    bool all_v2 = true;
    for (auto* s : set.st) all_v2 = all_v2 && s->use_v2;
    const int64_t chunk_env = chunk_rows_env();
    if (chunk_env > 0 || !all_v2) {
        const int64_t chunk = (ns > 1 && chunk_env > 0) ? std::min<int64_t>(chunk_env, ((n + 63) / 64) * 64) : n;
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t r1 = std::min(n, r0 + chunk);
            for (size_t g = 0; g < ns; ++g) {
                // (chunk buffers are cut out of the whole-column buffers: [3][rows] at the front)
                const int32_t* left = g ? set.edge[g - 1] : nullptr;
                int32_t* right = (g + 1 < ns) ? set.edge[g] : nullptr;
                if (int rc = fill_rows_impl(set.st[g], r0, r1, left, right, nullptr, FillOpts())) return rc;
            }
        }
        return psa_stripe_final(set.st.back(), nullptr, triple);
    }
    const bool with_codes = set.st[0]->codes != nullptr;
    for (size_t g0 = 0; g0 < ns; g0 += psa::kMaxJobs) {
        const int nj = (int)std::min<size_t>(psa::kMaxJobs, ns - g0);
        psa::FillArgs2 jobs[psa::kMaxJobs];
        unsigned groups[psa::kMaxJobs], delay[psa::kMaxJobs];
        unsigned total = 0;
        for (int j = 0; j < nj; ++j) {
            psa_stripe* s = set.st[g0 + j];
            const size_t g = g0 + j;
            FillOpts o;
            if (j > 0) o.left_done = set.st[g - 1]->done;          // inside this launch
            if (j + 1 < nj) o.right_done = s->done;
            if (o.right_done) HIP_TRY(hipMemsetAsync(s->done, 0, sizeof(unsigned) * (size_t)((n + 63) / 64), nullptr));
            bool wc = false;
            if (int rc = prepare_fill(s, 0, n, g ? set.edge[g - 1] : nullptr, (g + 1 < ns) ? set.edge[g] : nullptr, nullptr, o,
                                      &jobs[j], &wc)) return rc;
            groups[j] = groups_of(jobs[j]);
            total += groups[j];
            // stripe j's group r can start once stripe j-1's group r is through: about (W + 64) steps, while
            // groups of one stripe start some 400 steps apart; a stripe that draws its tickets a little early
            // (600 steps per ticket of delay, swept: PSA_STRIPE_SKEW_STEPS) waits less than one that draws them late
            static const int skew_steps = [] { const char* e = getenv("PSA_STRIPE_SKEW_STEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 600; }();
            delay[j] = j ? delay[j - 1] + (unsigned)((set.st[g - 1]->W + 64 + skew_steps - 1) / skew_steps) + 1 : 0;
        }
        const uint32_t* order_dev = nullptr;
        if (nj > 1) {
            std::vector<uint32_t> order;
            order.reserve(total);
            const unsigned last_slot = delay[nj - 1] + groups[nj - 1];
            for (unsigned slot = 0; slot < last_slot; ++slot)
                for (int j = 0; j < nj; ++j)
                    if (slot >= delay[j] && slot - delay[j] < groups[j]) order.push_back(((uint32_t)j << psa::kOrderJobShift) | (slot - delay[j]));
            if (set.order_cap < order.size()) {
                hipFree(set.order);
                set.order = nullptr; set.order_cap = 0;
                HIP_TRY(hipMalloc(&set.order, sizeof(uint32_t) * order.size()));
                set.order_cap = order.size();
            }
            HIP_TRY(hipMemcpy(set.order, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice));
            order_dev = set.order;
        }
        psa_stripe* s0 = set.st[g0];
        // diagnostic only (PSA_PROFILE_STRIPES=path): per-strip start/end ticks of every stripe of the launch
        const char* prof_path = getenv("PSA_PROFILE_STRIPES");
        const size_t strips = (size_t)((n + 63) / 64);
        psa::u64* prof_dev = nullptr;
        if (prof_path) {
            HIP_TRY(hipMalloc(&prof_dev, sizeof(psa::u64) * 4 * strips * nj));
            HIP_TRY(hipMemset(prof_dev, 0, sizeof(psa::u64) * 4 * strips * nj));
            for (int j = 0; j < nj; ++j) jobs[j].base.prof = prof_dev + 4 * strips * j;
        }
        HIP_TRY(hipEventRecord(s0->ev[0], nullptr));
        if (int rc = launch_fill_v2(jobs, nj, order_dev, total, jobs[0].base.ticket, with_codes,
                                    nj > 1 ? psa::kKeepoutManyBytes : psa::kKeepoutAloneBytes, nullptr)) return rc;
        HIP_TRY(hipEventRecord(s0->ev[1], nullptr));
        s0->fill_timed = true;
        if (prof_dev) {
            std::vector<psa::u64> h(4 * strips * nj);
            HIP_TRY(hipMemcpy(h.data(), prof_dev, sizeof(psa::u64) * h.size(), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(prof_path, "w")) {
                fprintf(f, "stripe,strip,start,end,xcc,hwid,wait_in,wait_room\n");
                for (int j = 0; j < nj; ++j)
                    for (size_t k = 0; k < strips; ++k) {
                        const psa::u64* q = &h[4 * (strips * j + k)];
                        fprintf(f, "%d,%zu,%lld,%lld,%llu,%llu,%llu,%llu\n", j, k, (long long)(q[0] - h[0]), (long long)(q[1] - h[0]),
                                q[2] >> 32, q[2] & 0xffffffffu, q[3] & 0xffffffffu, q[3] >> 32);
                    }
                fclose(f);
            }
            hipFree(prof_dev);
        }
    }
    for (size_t g = 0; g + 1 < ns; ++g)   // (an expired wait anywhere fails the call, not only in the last stripe)
        if (int rc = check_abort(set.st[g], nullptr)) return rc;
    return psa_stripe_final(set.st.back(), nullptr, triple);
}

}  // namespace

extern "C" {

namespace {
void release_pair_caches() {   // the objects psa_align / psa_score / the multi-GPU calls keep for their next call
    if (g_cache) { g_cache->clear(); delete g_cache; g_cache = nullptr; }
    drop_multi_fwd();
}
}  // namespace

void psa_release_cache(void) {
    release_pair_caches();
    psa_drop_many_cache_impl();   // (psa_align_many's engines)
}

int psa_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p, int32_t* score_out,
              int32_t triple_out[3]) {
    if (!score_out || n < 0 || m < 0 || (n && !a) || (m && !b)) return fail(PSA_E_ARG, "psa_score: bad argument");
    if (p < 1) return fail(PSA_E_ARG, "p must be >= 1 (the reference divides by p, simple.cpp:92)");
    if (int rc = need_device()) return rc;
    int32_t t[3];
    if (n == 0) { t[0] = -1; t[1] = 0; t[2] = (int32_t)-(2 + m); }       // loop of simple.cpp:122 never runs
    else if (m == 0) { t[0] = -1; t[1] = (int32_t)-(2 + n); t[2] = -1; } // only column 0 exists (simple.cpp:124-133)
    else {
        if (n > (1LL << 27) || m > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
        StripeSet* set = nullptr;
        if (int rc = get_stripes(a, n, b, m, p, 0, &set)) return rc;
        if (int rc = run_stripes(*set, t)) return rc;
    }
    pick3(t, score_out);
    if (triple_out) memcpy(triple_out, t, sizeof t);
    return PSA_OK;
}

int psa_align(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p, int32_t* score_out,
              int32_t** ai_out, int32_t** bj_out, int64_t* len_out) {
    if (!score_out || !ai_out || !bj_out || !len_out || n < 0 || m < 0 || (n && !a) || (m && !b))
        return fail(PSA_E_ARG, "psa_align: bad argument");
    if (p < 1) return fail(PSA_E_ARG, "p must be >= 1 (the reference divides by p, simple.cpp:92)");
    if (int rc = need_device()) return rc;
    *ai_out = *bj_out = nullptr;
    *len_out = 0;
    const int64_t cap = 2 * (n + m) + 2;
    if (n > 0 && m > 0) {
        if (n > (1LL << 27) || m > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
        // codes of the whole matrix (1 B per cell) against what the device has free (plus what this thread's
        // cached stripes of the same shape already hold): beyond 70 % of it -- or when PSA_BAND_ROWS asks
        // for it -- align in linear space
        size_t free_b = 0, total_b = 0;
        const double code_bytes = (double)n * (double)m;
        const bool forced = getenv("PSA_BAND_ROWS") != nullptr;
        const bool cached = g_cache && g_cache->n == n && g_cache->m == m && g_cache->p == p && g_cache->flags == PSA_F_TRACEBACK;
        if (!cached) psa_make_room(1.05 * code_bytes + (double)(64 << 20));   // (before the memory is measured)
        if (forced || (!cached && hipMemGetInfo(&free_b, &total_b) == hipSuccess && code_bytes > 0.7 * (double)free_b)) {
            psa_release_cache();
            return psa_align_banded(a, n, b, m, 0, score_out, ai_out, bj_out, len_out);
        }
    }
    int32_t* ai = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* bj = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    if (!ai || !bj) { free(ai); free(bj); return fail(PSA_E_NOMEM, "psa_align: host allocation failed"); }
    int32_t t[3];
    int64_t len = 0;
    int rc = PSA_OK;
    if (n == 0) { t[0] = -1; t[1] = 0; t[2] = (int32_t)-(2 + m); }
    else if (m == 0) { t[0] = -1; t[1] = (int32_t)-(2 + n); t[2] = -1; }
    else {
        StripeSet* set = nullptr;
        rc = get_stripes(a, n, b, m, p, PSA_F_TRACEBACK, &set);
        if (rc == PSA_OK) rc = run_stripes(*set, t);
        if (rc == PSA_OK) {
            int32_t v;
            int64_t i = n, j = m;
            int k = pick3(t, &v);
            for (size_t g = set->st.size(); g-- > 0 && rc == PSA_OK;) {
                rc = psa_stripe_traceback(set->st[g], &i, &j, &k, ai, bj, cap, &len, nullptr);
                if (i == 0 || j == 0) break;
            }
        }
    }
    if (rc == PSA_OK) rc = psa_finish_alignment(ai, bj, len, cap, &len);
    if (rc != PSA_OK) { free(ai); free(bj); return rc; }
    pick3(t, score_out);
    *ai_out = ai; *bj_out = bj; *len_out = len;
    return PSA_OK;
}

// Linear-space alignment: see include/psa.h.  One stripe (results do not depend on p).
int psa_align_banded(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int64_t band_rows,
                     int32_t* score_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out) {
    if (!score_out || !ai_out || !bj_out || !len_out || n < 1 || m < 1 || !a || !b)
        return fail(PSA_E_ARG, "psa_align_banded: bad argument (needs n, m >= 1)");
    if (band_rows < 0 || (band_rows & 63)) return fail(PSA_E_ARG, "psa_align_banded: band_rows must be a multiple of 64");
    if (int rc = need_device()) return rc;
    *ai_out = *bj_out = nullptr;
    *len_out = 0;
    psa_make_room((double)(24LL << 30));   // (a band takes 8 GiB when three times that are free, band_rows_for)
    psa_stripe* s = nullptr;
    g_band_rows_request = band_rows;
    int rc = psa_stripe_create(&s, a, n, b, m, 0, m, PSA_F_TRACEBACK | PSA_F_BANDED);
    g_band_rows_request = 0;
    if (rc) return rc;
    struct Owner { psa_stripe* s; ~Owner() { free_stripe(s); } } owner{s};
    const int64_t B = s->band_rows;
    // pass 1: score only, one launch; the last strip of every band leaves its bottom row in ckpt
    {
        // (granules are recognised by their tag = strip number: rows left over from another pair must go)
        const int64_t nbands = (n + B - 1) / B;
        HIP_TRY(hipMemsetAsync(s->ckpt, 0, sizeof(psa::u64) * (size_t)(nbands + 1) * (size_t)s->Wpad, nullptr));
        FillOpts fo;
        fo.codes = 0;
        if ((rc = fill_rows_impl(s, 0, n, nullptr, nullptr, nullptr, fo))) return rc;
    }
    int32_t t[3];
    if ((rc = psa_stripe_final(s, nullptr, t))) return rc;
    const int64_t cap = 2 * (n + m) + 2;
    int32_t* ai = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* bj = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    if (!ai || !bj) { free(ai); free(bj); return fail(PSA_E_NOMEM, "psa_align_banded: host allocation failed"); }
    int32_t v;
    int64_t i = n, j = m, len = 0;
    int k = pick3(t, &v);
    // pass 2: bottom band first; refill with codes the columns the walk can still reach, walk, hand on
    for (int64_t bnd = (i - 1) / B; bnd >= 0 && i > 0 && j > 0 && rc == PSA_OK; --bnd) {
        const int64_t r0 = bnd * B, r1 = std::min(n, r0 + B);
        if (r0 > 0 && i - (k != 2) <= r0) continue;            // the walk left this band without a step in it
        const int Wr = (int)std::min<int64_t>(s->W, j);
        const int Qr = steps_q(Wr);
        uint8_t* base = (uint8_t*)((uintptr_t)s->codes - (uintptr_t)psa::code_tile_offset(r0 / 64, Qr, 0));
        if (getenv("PSA_TRACE_STATS")) fprintf(stderr, "banded: band %lld rows %lld..%lld columns %d, walk at (%lld,%lld,%d)\n", (long long)bnd, (long long)r0, (long long)r1, Wr, (long long)i, (long long)j, k);
        FillOpts fo;
        fo.codes = 1; fo.W = Wr; fo.codes_base = base; fo.any_order = true;
        if ((rc = fill_rows_impl(s, r0, r1, nullptr, nullptr, nullptr, fo))) break;
        TraceOpts to;
        to.codes_base = base; to.W = Wr; to.row_min = r0;
        rc = traceback_impl(s, &i, &j, &k, ai, bj, cap, &len, nullptr, to);
    }
    if (rc == PSA_OK) rc = psa_finish_alignment(ai, bj, len, cap, &len);
    if (rc != PSA_OK) { free(ai); free(bj); return rc; }
    *score_out = v;
    *ai_out = ai; *bj_out = bj; *len_out = len;
    return PSA_OK;
}

}  // extern "C"


// ---- run() with its p workers on several GPUs of ONE process (psa_align_multi / psa_score_multi) ----
// simple.cpp:261-277 spawns its workers inside run(): the caller of the drop-in boundary gets the parallel run without
// a launcher, a process group or a collective library.  Here: stripe g of B (simple.cpp:90-99) lives on GPU
// devices[g] and is ONE flagged fill launch on a stream of that GPU; the boundary between stripes g and g+1 --
// int32[3][n] and one flag per strip, the mailbox of simple.cpp:136-154 / 215-235 -- is fine-grained memory of GPU
// g+1 that GPU g's fill writes directly over xGMI (peer access; system-scope write-through stores, s_waitcnt
// vmcnt(0), then the flag) and GPU g+1's fill polls locally.  No post / take kernels, no host loop over chunks, no
// dependence on how streams map to hardware queues: every launch only ever waits for a launch on ANOTHER GPU that
// was started before it (or, for stripes that share a GPU, for workgroups that are resident beside its own).
namespace {

thread_local double g_multi_fill_ms = 0.0;   // host clock: first fill queued -> every stripe's fill complete

struct MultiRun {
    std::vector<psa_stripe*> st;
    std::vector<int> dev;
    std::vector<hipStream_t> stream;
    std::vector<int32_t*> edge;     // boundary g -> g+1 on dev[g+1]
    std::vector<uint32_t*> flags;
    int caller_dev = 0;
    int64_t n = 0, m = 0;           // what the buffers were made for (the thread's cache: g_multi)
    uint32_t mode = 0;
    std::vector<int> dev_key;
    ~MultiRun() {
        if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = 0;
        for (size_t g = 0; g < stream.size(); ++g) {
            if (!stream[g]) continue;
            hipSetDevice(dev[g]);
            hipStreamSynchronize(stream[g]);
            hipStreamDestroy(stream[g]);
        }
        for (auto* s : st) free_stripe(s);
        for (size_t g = 0; g < edge.size(); ++g) {
            hipSetDevice(dev[g + 1]);
            hipFree(edge[g]);
            if (g < flags.size()) hipFree(flags[g]);
        }
        hipSetDevice(caller_dev);
    }
};

// Fills every stripe; t = T1,T2,T3 at (n, m).  PSA_E_STATE (nothing launched) when a stripe needs the generic kernel.
// The stripes, streams and boundary buffers of the calling thread's most recent multi-GPU call are kept and reused by the
// next call of the same shape (sizes, devices, with or without codes), as psa_align does: psa_release_cache() frees them.
thread_local MultiRun* g_multi = nullptr;

}  // namespace
void psa_drop_multi_cache_impl() { delete g_multi; g_multi = nullptr; }
namespace {
void drop_multi() { psa_drop_multi_cache_impl(); }

int multi_launch(MultiRun& R, int64_t n, int32_t t[3]);

int multi_fill_on(MultiRun& R, bool reuse, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                  uint32_t flags, int32_t t[3]);

int multi_fill(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu, uint32_t flags,
               MultiRun** out, int32_t t[3]) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    std::vector<int> key;
    for (int id = 0; id < ngpu; ++id) key.push_back(devices ? devices[id] : (ndev > 0 ? id % ndev : 0));
    const bool reuse = g_multi && g_multi->n == n && g_multi->m == m && g_multi->mode == flags && g_multi->dev_key == key;
    if (!reuse) {
        drop_multi();
        psa_drop_many_cache_impl();
        g_multi = new MultiRun();
    }
    int rc = multi_fill_on(*g_multi, reuse, a, n, b, m, devices, ngpu, flags, t);
    if (rc != PSA_OK) { drop_multi(); return rc; }
    g_multi->n = n; g_multi->m = m; g_multi->mode = flags; g_multi->dev_key = key;
    *out = g_multi;
    return PSA_OK;
}

int multi_fill_on(MultiRun& R, bool reuse, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                  uint32_t flags, int32_t t[3]) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    HIP_TRY(hipGetDevice(&R.caller_dev));
    struct Back { int d; ~Back() { hipSetDevice(d); } } back{R.caller_dev};   // (the caller's device is current again on return)
    if (reuse) {
        const size_t G = R.st.size();
        const size_t strips = (size_t)((n + 63) / 64);
        for (size_t g = 0; g < G; ++g) {
            HIP_TRY(hipSetDevice(R.dev[g]));
            psa_stripe* s = R.st[g];
            if (int rc = bind_stripe(s, n, m, s->c0, s->c1)) return rc;
            if (int rc = upload_host(s, a, b, R.stream[g])) return rc;
            if (!s->use_v2) return fail(PSA_E_STATE, "psa_align_multi: the generic kernel is forced (PSA_FILL_KERNEL=v1; %d table rows)", psa::kMaxSym);
            if (g > 0) HIP_TRY(hipMemsetAsync(R.flags[g - 1], 0, sizeof(uint32_t) * (strips + 64), R.stream[g]));   // (its consumer's stream and device)
        }
        return multi_launch(R, n, t);
    }
    R.dev.clear();
    std::vector<std::pair<int64_t, int64_t>> cols;
    const int64_t block = m / ngpu;   // simple.cpp:90-99; zero-width stripes (ngpu > m) own no cell
    for (int id = 0; id < ngpu; ++id) {
        const int64_t c0 = (int64_t)id * block, c1 = (id == ngpu - 1) ? m : c0 + block;
        if (c1 == c0) continue;
        const int d = devices ? devices[id] : id % ndev;   // (no list: round the visible GPUs; sharing is checked below)
        if (d < 0 || d >= ndev) return fail(PSA_E_ARG, "psa_align_multi: stripe %d names device %d, %d device(s) visible", id, d, ndev);
        cols.push_back({c0, c1});
        R.dev.push_back(d);
    }
    const size_t G = cols.size();
    // Stripes that share a GPU are separate launches whose waiting strips hold their CUs: all of them must be
    // resident together (one workgroup of 256 rows per CU).  One GPU for everything is psa_align(p)'s case: there the
    // stripes are jobs of ONE launch and any size goes.
    const int64_t groups = ((n + 63) / 64 + psa::kNW - 1) / psa::kNW;
    for (size_t g = 0; g < G; ++g) {
        int64_t sharing = 0;
        for (size_t h = 0; h < G; ++h) sharing += R.dev[h] == R.dev[g];
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, R.dev[g]));
        if (sharing > 1 && sharing * groups > cus)
            return fail(PSA_E_ARG, "psa_align_multi: %lld stripes of %lld rows share device %d and do not fit on it together "
                        "(%lld workgroups each, %d compute units): give every stripe a GPU of its own, or use psa_align(p)",
                        (long long)sharing, (long long)n, R.dev[g], (long long)groups, cus);
    }
    const size_t strips = (size_t)((n + 63) / 64);
    for (size_t g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(R.dev[g]));
        hipStream_t st = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        R.stream.push_back(st);
        psa_stripe* s = nullptr;
        int rc = alloc_stripe(&s, n, (int)(cols[g].second - cols[g].first), flags, cols[g].first > 0);
        if (rc == PSA_OK) R.st.push_back(s);
        if (rc == PSA_OK) rc = bind_stripe(s, n, m, cols[g].first, cols[g].second);
        if (rc == PSA_OK) rc = upload_host(s, a, b, st);
        if (rc != PSA_OK) return rc;
        if (!s->use_v2) return fail(PSA_E_STATE, "psa_align_multi: the generic kernel is forced (PSA_FILL_KERNEL=v1; %d table rows)", psa::kMaxSym);
        if (g + 1 < G) {
            if (R.dev[g] != R.dev[g + 1]) {
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, R.dev[g], R.dev[g + 1]));
                if (!can) return fail(PSA_E_HIP, "psa_align_multi: device %d cannot access the memory of device %d", R.dev[g], R.dev[g + 1]);
                const hipError_t e = hipDeviceEnablePeerAccess(R.dev[g + 1], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(PSA_E_HIP, "hipDeviceEnablePeerAccess(%d) from device %d failed: %s", R.dev[g + 1], R.dev[g], hipGetErrorString(e));
                (void)hipGetLastError();
            }
            HIP_TRY(hipSetDevice(R.dev[g + 1]));   // the boundary lives with its consumer: polled locally, written remotely
            int32_t* e = nullptr;
            uint32_t* f = nullptr;
            HIP_TRY(hipExtMallocWithFlags((void**)&e, sizeof(int32_t) * 3 * (size_t)n, hipDeviceMallocFinegrained));
            R.edge.push_back(e);
            HIP_TRY(hipExtMallocWithFlags((void**)&f, sizeof(uint32_t) * (strips + 64), hipDeviceMallocFinegrained));
            R.flags.push_back(f);
            HIP_TRY(hipMemset(f, 0, sizeof(uint32_t) * (strips + 64)));
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    return multi_launch(R, n, t);
}

int multi_launch(MultiRun& R, int64_t n, int32_t t[3]) {
    const size_t G = R.st.size();
    for (size_t g = 0; g < G; ++g) {   // sequences, tables and cleared flags in place before any fill is queued
        HIP_TRY(hipSetDevice(R.dev[g]));
        HIP_TRY(hipStreamSynchronize(R.stream[g]));
    }
    const auto t_begin = std::chrono::steady_clock::now();
    for (size_t g = 0; g < G; ++g) {   // head stripe first: a launch only ever waits for launches queued before it
        FillOpts o;
        o.left_done = g ? R.flags[g - 1] : nullptr;
        o.right_done = (g + 1 < G) ? R.flags[g] : nullptr;
        if (int rc = fill_rows_impl(R.st[g], 0, n, g ? R.edge[g - 1] : nullptr, (g + 1 < G) ? R.edge[g] : nullptr, R.stream[g], o)) return rc;
    }
    for (size_t g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(R.dev[g]));
        HIP_TRY(hipStreamSynchronize(R.stream[g]));
    }
    g_multi_fill_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    for (size_t g = 0; g + 1 < G; ++g) {
        DeviceGuard dg(R.dev[g]);
        if (dg.rc) return dg.rc;
        if (int rc = check_abort(R.st[g], R.stream[g])) return rc;
    }
    return psa_stripe_final(R.st.back(), R.stream.back(), t);
}

int multi_args_ok(const char* who, int64_t n, const uint8_t* a, int64_t m, const uint8_t* b, int ngpu) {
    if (n < 0 || m < 0 || (n && !a) || (m && !b)) return fail(PSA_E_ARG, "%s: bad argument", who);
    if (ngpu < 1) return fail(PSA_E_ARG, "%s: ngpu must be >= 1 (the reference divides by p, simple.cpp:92)", who);
    if (n > (1LL << 27) || m > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
    return need_device();
}

}  // namespace

extern "C" {

double psa_multi_last_fill_ms(void) { return g_multi_fill_ms; }

int psa_score_multi(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                    int32_t* score_out, int32_t triple_out[3]) {
    if (!score_out) return fail(PSA_E_ARG, "psa_score_multi: bad argument");
    if (int rc = multi_args_ok("psa_score_multi", n, a, m, b, ngpu)) return rc;
    if (n == 0 || m == 0) return psa_score(a, n, b, m, 1, score_out, triple_out);
    int32_t t[3];
    MultiRun* R = nullptr;
    const int rc = multi_fill(a, n, b, m, devices, ngpu, 0, &R, t);
    if (rc == PSA_E_STATE) return psa_score(a, n, b, m, ngpu, score_out, triple_out);   // generic kernel: this GPU only
    if (rc != PSA_OK) return rc;
    pick3(t, score_out);
    if (triple_out) memcpy(triple_out, t, sizeof t);
    return PSA_OK;
}

int psa_align_multi(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, const int* devices, int ngpu,
                    int32_t* score_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out) {
    if (!score_out || !ai_out || !bj_out || !len_out) return fail(PSA_E_ARG, "psa_align_multi: bad argument");
    if (int rc = multi_args_ok("psa_align_multi", n, a, m, b, ngpu)) return rc;
    if (n == 0 || m == 0) return psa_align(a, n, b, m, 1, score_out, ai_out, bj_out, len_out);
    *ai_out = *bj_out = nullptr;
    *len_out = 0;
    const int64_t cap = 2 * (n + m) + 2;
    int32_t* ai = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* bj = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    if (!ai || !bj) { free(ai); free(bj); return fail(PSA_E_NOMEM, "psa_align_multi: host allocation failed"); }
    int32_t t[3], v = 0;
    int64_t len = 0;
    MultiRun* R = nullptr;
    int rc = multi_fill(a, n, b, m, devices, ngpu, PSA_F_TRACEBACK, &R, t);
    if (rc == PSA_OK) {
        // the walk: right to left through the stripes, each on its own GPU, (i, j, k) handed on (psa.h)
        int64_t i = n, j = m;
        int k = pick3(t, &v);
        for (size_t g = R->st.size(); g-- > 0 && rc == PSA_OK;) {
            rc = psa_stripe_traceback(R->st[g], &i, &j, &k, ai, bj, cap, &len, R->stream[g]);
            if (i == 0 || j == 0) break;
        }
        if (rc != PSA_OK) drop_multi();
    }
    if (rc == PSA_E_STATE) { free(ai); free(bj); return psa_align(a, n, b, m, ngpu, score_out, ai_out, bj_out, len_out); }
    if (rc == PSA_OK) rc = psa_finish_alignment(ai, bj, len, cap, &len);
    if (rc != PSA_OK) { free(ai); free(bj); return rc; }
    *score_out = v;
    *ai_out = ai; *bj_out = bj; *len_out = len;
    return PSA_OK;
}

}  // extern "C"


// ---- several pairs in ONE launch (psa_many_*, psa_align_many): see include/psa.h ----
struct psa_many {
    int device = 0;
    int slots = 0;
    uint32_t flags = 0;
    int64_t n_cap = 0, m_cap = 0;
    std::vector<psa_stripe*> st;
    std::vector<char> bound;        // 0: free, 1: holds a pair with cells, 2: holds a pair without cells (n or m is 0)
    std::vector<int64_t> n, m;
    hipStream_t stream = nullptr;        // pair binding and the fill launch
    hipStream_t walk_stream = nullptr;   // the walks and the copies of the results: highest priority (see psa_many_create)
    unsigned* ticket = nullptr;     // 256 bytes
    uint32_t* order = nullptr;
    size_t order_cap = 0;
    std::vector<unsigned> order_sig;   // group counts the order on the device was built for
    int32_t* d_res = nullptr;       // slots x kResultWords: walk state, final triple, abort word of every slot (collect_results)
    int32_t* h_res = nullptr;       // pinned copy
    std::vector<int32_t*> h_ai, h_bj;   // pinned: reversed columns as the walk emitted them
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // fill start / end, walks end, results on the host, walks start
    bool launched = false;
    bool walks_queued = false;      // the launch's walks and result copies are on the walk stream (see queue_walks)
};

namespace {

// Stream pairs of the engines (fill stream, normal priority; walk stream, highest).  HIP multiplexes the streams of a
// priority onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); with five engines -- five streams per priority
// -- a launch was seen never to complete (a walk stream's wait for its fill ahead of that fill in a shared queue).  So a
// device has at most kEnginePairs pairs: further engines share the pairs of the first ones, their work takes turns in
// stream order, and nothing can wait for something queued behind it.
constexpr int kEnginePairs = 3;
struct EnginePair { int device = -1; hipStream_t fill = nullptr, walk = nullptr; int users = 0; };
std::mutex g_pair_mutex;
std::vector<EnginePair> g_pairs;

hipError_t acquire_engine_pair(int device, hipStream_t* fill, hipStream_t* walk) {
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    EnginePair* best = nullptr;
    int on_device = 0;
    for (auto& p : g_pairs)
        if (p.device == device) { ++on_device; if (!best || p.users < best->users) best = &p; }
    if (!best || (best->users > 0 && on_device < kEnginePairs)) {
        int least = 0, greatest = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e != hipSuccess) return e;
        EnginePair np;
        np.device = device;
        if ((e = hipStreamCreateWithPriority(&np.fill, hipStreamNonBlocking, (least + greatest) / 2)) != hipSuccess) return e;
        const char* wp = getenv("PSA_WALK_PRIORITY");   // (experiment switch: "normal" puts the walk streams in the fills' class)
        const int walk_prio = (wp && !strcmp(wp, "normal")) ? (least + greatest) / 2 : greatest;
        if ((e = hipStreamCreateWithPriority(&np.walk, hipStreamNonBlocking, walk_prio)) != hipSuccess) { hipStreamDestroy(np.fill); return e; }
        g_pairs.push_back(np);
        best = &g_pairs.back();
    }
    ++best->users;
    *fill = best->fill; *walk = best->walk;
    return hipSuccess;
}
void release_engine_pair(hipStream_t fill) {
    std::lock_guard<std::mutex> lock(g_pair_mutex);
    for (size_t k = 0; k < g_pairs.size(); ++k)
        if (g_pairs[k].fill == fill) {
            if (--g_pairs[k].users == 0) {
                hipStreamDestroy(g_pairs[k].fill); hipStreamDestroy(g_pairs[k].walk);
                g_pairs.erase(g_pairs.begin() + (long)k);
            }
            return;
        }
}

void free_many(psa_many* e) {
    if (!e) return;
    DeviceGuard g(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->walk_stream) hipStreamSynchronize(e->walk_stream);
    for (auto* s : e->st) free_stripe(s);
    for (auto* p : e->h_ai) hipHostFree(p);
    for (auto* p : e->h_bj) hipHostFree(p);
    hipFree(e->ticket); hipFree(e->order); hipFree(e->d_res);
    hipHostFree(e->h_res);
    for (auto& x : e->ev) if (x) hipEventDestroy(x);
    if (e->stream) release_engine_pair(e->stream);
    delete e;
}

int many_bind(psa_many* e, int slot, int64_t n, int64_t m) {
    if (!e || slot < 0 || slot >= e->slots) return fail(PSA_E_ARG, "psa_many: no such slot");
    if (n < 0 || m < 0 || n > e->n_cap || m > e->m_cap)
        return fail(PSA_E_ARG, "psa_many: pair of %lld x %lld does not fit the engine (%lld x %lld)", (long long)n, (long long)m,
                    (long long)e->n_cap, (long long)e->m_cap);
    e->n[slot] = n; e->m[slot] = m;
    e->bound[slot] = (n == 0 || m == 0) ? 2 : 1;
    if (e->bound[slot] == 1) return bind_stripe(e->st[slot], n, m, 0, m);
    return PSA_OK;
}

}  // namespace

extern "C" {

void psa_many_destroy(psa_many* e) { free_many(e); }

int psa_many_create(psa_many** out, int slots, int64_t n_cap, int64_t m_cap, uint32_t flags) {
    if (!out) return fail(PSA_E_ARG, "psa_many_create: out is null");
    *out = nullptr;
    if (slots < 1 || slots > psa::kMaxJobs) return fail(PSA_E_ARG, "psa_many_create: 1 to %d slots", psa::kMaxJobs);
    if (n_cap < 1 || m_cap < 1 || n_cap > (1LL << 27) || m_cap > (1LL << 27)) return fail(PSA_E_ARG, "psa_many_create: bad capacity");
    if (flags & PSA_F_BANDED) return fail(PSA_E_ARG, "psa_many_create: banded alignment goes through psa_align_banded");
    if (int rc = need_device()) return rc;
    psa_many* e = new psa_many();
    if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
    e->slots = slots; e->flags = flags; e->n_cap = n_cap; e->m_cap = m_cap;
    e->bound.assign(slots, 0); e->n.assign(slots, 0); e->m.assign(slots, 0);
#define MANY_TRY(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            free_many(e);                                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                      \
    } while (0)
    {
        // Two streams per engine.  The fill launch of a batch wants every SIMD of the chip for 30 ms, and the next
        // engine's fill, queued behind it, takes over as it drains: a walk (16 workgroups, 2 ms) or a result copy that
        // waited its turn behind them would wait for a whole fill (seen: 29 ms), and with it the host that wants the
        // results before it re-binds the engine.  So walks and copies go to a stream of the HIGHEST priority: they get
        // slots that fill workgroups free ahead of the fill's own next workgroups (not reliably: a walk still waits
        // 16-29 ms behind a full fill where it takes 2.3 ms on an idle chip, which is why psa_align_many lets three
        // engines take turns).  Every fill stream has the SAME (normal) priority: with one engine's fill on the lowest
        // priority the high-priority stream's wait for that fill never ended (two engines in flight: the queue
        // scheduler starves the lowest class while a higher one has work, even work that waits for it).
        MANY_TRY(acquire_engine_pair(e->device, &e->stream, &e->walk_stream));
    }
    for (auto& x : e->ev) MANY_TRY(hipEventCreate(&x));
    MANY_TRY(hipMalloc(&e->ticket, 256));
    MANY_TRY(hipMalloc(&e->d_res, sizeof(int32_t) * psa::kResultWords * slots));
    MANY_TRY(hipHostMalloc(&e->h_res, sizeof(int32_t) * psa::kResultWords * slots));
    for (int k = 0; k < slots; ++k) {
        psa_stripe* s = nullptr;
        if (int rc = alloc_stripe(&s, n_cap, (int)m_cap, flags, false)) { free_many(e); return rc; }
        e->st.push_back(s);
        if (flags & PSA_F_TRACEBACK) {
            int32_t *pa = nullptr, *pb = nullptr;
            MANY_TRY(hipHostMalloc(&pa, sizeof(int32_t) * (size_t)(n_cap + m_cap + 2)));
            e->h_ai.push_back(pa);
            MANY_TRY(hipHostMalloc(&pb, sizeof(int32_t) * (size_t)(n_cap + m_cap + 2)));
            e->h_bj.push_back(pb);
        }
    }
    // first use of the engine's stream now, not while fills of other engines are running (creating a
    // hardware queue under resident fills was seen to stall them for seconds under a profiler)
    MANY_TRY(hipMemsetAsync(e->ticket, 0, 256, e->stream));
    MANY_TRY(hipStreamSynchronize(e->stream));
    MANY_TRY(hipMemsetAsync(e->ticket, 0, 256, e->walk_stream));
    MANY_TRY(hipStreamSynchronize(e->walk_stream));
#undef MANY_TRY
    *out = e;
    return PSA_OK;
}

int psa_many_set_pair(psa_many* e, int slot, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m) {
    if (e && e->launched) return fail(PSA_E_STATE, "psa_many_set_pair: results of the last launch have not been waited for");
    if ((n > 0 && !a) || (m > 0 && !b)) return fail(PSA_E_ARG, "psa_many_set_pair: null sequence");
    if (int rc = many_bind(e, slot, n, m)) return rc;
    if (e->bound[slot] != 1) return PSA_OK;
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    // (the residues go up now -- the caller's buffers are its own again when this returns --; alphabet, ids and table are
    // built by the launch's binding kernels, for all slots at once)
    psa_stripe* s = e->st[slot];
    HIP_TRY(hipMemcpyAsync(s->a_dev, a, (size_t)n, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(s->b_dev, b, (size_t)m, hipMemcpyHostToDevice, e->stream));
    s->src_a = s->src_b = nullptr;
    s->rebind = true;
    return PSA_OK;
}

int psa_many_set_pair_dev(psa_many* e, int slot, const uint8_t* a_dev, int64_t n, const uint8_t* b_dev, int64_t m) {
    if (e && e->launched) return fail(PSA_E_STATE, "psa_many_set_pair_dev: results of the last launch have not been waited for");
    if ((n > 0 && !a_dev) || (m > 0 && !b_dev)) return fail(PSA_E_ARG, "psa_many_set_pair_dev: null sequence");
    if (int rc = many_bind(e, slot, n, m)) return rc;
    if (e->bound[slot] != 1) return PSA_OK;
    // nothing is queued here: the next launch's binding kernels copy the residues into the slot's buffers and encode
    // them, all slots in three launches (a dozen stream operations per slot before: 2.7 ms for 20 slots)
    psa_stripe* s = e->st[slot];
    s->src_a = a_dev; s->src_b = b_dev;
    s->rebind = true;
    return PSA_OK;
}

int psa_many_clear_slot(psa_many* e, int slot) {
    if (!e || slot < 0 || slot >= e->slots) return fail(PSA_E_ARG, "psa_many_clear_slot: no such slot");
    if (e->launched) return fail(PSA_E_STATE, "psa_many_clear_slot: results of the last launch have not been waited for");
    e->bound[slot] = 0;
    return PSA_OK;
}

// ONE fill launch over all bound slots, ONE launch of their walks, results copied to pinned host memory:
// everything asynchronous on the engine's stream.
int psa_many_launch(psa_many* e) { return psa_many_launch_after(e, nullptr); }

int psa_many_launch_after(psa_many* e, psa_many* after) {
    if (!e) return fail(PSA_E_ARG, "psa_many_launch: null engine");
    if (e->launched) return fail(PSA_E_STATE, "psa_many_launch: results of the last launch have not been waited for");
    if (after && after->device != e->device) return fail(PSA_E_ARG, "psa_many_launch_after: engines on different devices");
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    hipStream_t st = e->stream;
    // this engine's fill begins when the other engine's fill has ended: its walks and copies then run beside
    // this fill, and every fill launch has the chip to itself (but for those walks)
    if (after && after != e && after->launched) HIP_TRY(hipStreamWaitEvent(st, after->ev[1], 0));
    const bool with_codes = (e->flags & PSA_F_TRACEBACK) != 0;
    psa::FillArgs2 jobs[psa::kMaxJobs];
    unsigned groups[psa::kMaxJobs];
    int nj = 0;
    unsigned total = 0, rounds = 0;
    // Binding: everything the slots need before the fill -- residues of re-bound pairs into the slots' own buffers, their
    // alphabet, ids and table, and the zeroing of rings and control blocks -- in three launches over all slots.
    psa::BindBatch bb;
    memset(&bb, 0, sizeof bb);
    const char* force_fill = getenv("PSA_FILL_KERNEL");
    const bool generic = force_fill && !strcmp(force_fill, "v1");
    for (int k = 0; k < e->slots; ++k) {
        if (e->bound[k] != 1) continue;
        psa_stripe* s = e->st[k];
        psa::BindJob& bj = bb.job[bb.njobs++];
        bj.a = s->a_dev; bj.b = s->b_dev; bj.n = s->n; bj.W = s->W; bj.pitch = s->tab_pitch;
        bj.alpha = (unsigned*)s->alpha; bj.ids = s->a_ids; bj.eqtab = s->eqtab;
        if (s->rebind) { bj.src_a = s->src_a; bj.src_b = s->src_b; bj.tables = 1; s->rebind = false; }
        s->nsym_pending = false;
        s->use_v2 = !generic;   // (the table kernel takes any byte alphabet: a table row per distinct residue, kMaxSym = 256)
        if (generic) continue;  // the generic kernel (PSA_FILL_KERNEL=v1): launches of its own below, with their own memsets
        bool wc = false;
        FillOpts fo;
        fo.clear_into = &bj;
        if (int rc = prepare_fill(s, 0, s->n, nullptr, nullptr, st, fo, &jobs[nj], &wc)) return rc;
        groups[nj] = groups_of(jobs[nj]);
        total += groups[nj];
        rounds = std::max(rounds, groups[nj]);
        ++nj;
    }
    HIP_TRY(psa::launch_bind(bb, st));
    if (generic)
        for (int k = 0; k < e->slots; ++k)
            if (e->bound[k] == 1)
                if (int rc = fill_rows_impl(e->st[k], 0, e->st[k]->n, nullptr, nullptr, st, FillOpts())) return rc;
    // Couples for the packed kernel (two pairs per lane, psa_fill_pk.hip.h): pairs of the same size, two by two; whatever
    // is left over goes through the one-pair kernel in launches of its own.  PSA_FILL_PK=0: everything as one launch of
    // the int32 kernel (the cross-check).
    int pk_of[psa::kMaxJobs], npk = 0, rest_of[psa::kMaxJobs], nrest = 0;
    {
        const char* pkv = getenv("PSA_FILL_PK");   // (read at every launch: the tests switch it)
        const bool pk_ok = !(pkv && pkv[0] == '0');
        if (pk_ok && nj >= 2) {
            int idx[psa::kMaxJobs];
            for (int j = 0; j < nj; ++j) idx[j] = j;
            std::stable_sort(idx, idx + nj, [&](int x, int y) {
                return jobs[x].base.n != jobs[y].base.n ? jobs[x].base.n < jobs[y].base.n : jobs[x].base.W < jobs[y].base.W;
            });
            for (int j = 0; j < nj;) {
                const psa::FillArgs &fx = jobs[idx[j]].base;
                if (j + 1 < nj && fx.n == jobs[idx[j + 1]].base.n && fx.W == jobs[idx[j + 1]].base.W && fx.W >= 256 && fx.n >= 64 &&
                    fx.W < (1 << 24)) {
                    pk_of[npk++] = idx[j]; pk_of[npk++] = idx[j + 1];
                    j += 2;
                } else rest_of[nrest++] = idx[j++];
            }
        }
    }
    if (npk > 0) {
        const int nc = npk / 2;
        psa::FillBatch fb;
        memset(&fb, 0, sizeof fb);
        unsigned cgroups[psa::kMaxJobs], ctotal = 0, crounds = 0;
        for (int x = 0; x < npk; ++x) fb.job[x] = jobs[pk_of[x]];
        for (int cidx = 0; cidx < nc; ++cidx) {
            const psa::FillArgs& fx = jobs[pk_of[2 * cidx]].base;   // (workgroups of kPkNW strips there)
            cgroups[cidx] = (unsigned)(((fx.r1 - fx.r0 + 63) / 64 + psa::kPkNW - 1) / psa::kPkNW);
            ctotal += cgroups[cidx];
            crounds = std::max(crounds, cgroups[cidx]);
        }
        std::vector<unsigned> sig(cgroups, cgroups + nc);
        sig.push_back(0x504b0000u);   // (an order over couples, not over jobs)
        if (sig != e->order_sig) {
            std::vector<uint32_t> order;
            order.reserve(ctotal);
            for (unsigned r = 0; r < crounds; ++r)
                for (int cidx = 0; cidx < nc; ++cidx)
                    if (r < cgroups[cidx]) order.push_back(((uint32_t)cidx << psa::kOrderJobShift) | r);
            if (e->order_cap < order.size()) {
                hipFree(e->order);
                e->order = nullptr; e->order_cap = 0;
                HIP_TRY(hipMalloc(&e->order, sizeof(uint32_t) * order.size()));
                e->order_cap = order.size();
            }
            HIP_TRY(hipMemcpyAsync(e->order, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            e->order_sig = sig;
        }
        fb.njobs = npk; fb.order = e->order; fb.ticket = e->ticket;
        HIP_TRY(hipMemsetAsync(e->ticket, 0, 256, st));
        HIP_TRY(hipEventRecord(e->ev[0], st));
        HIP_TRY(psa::launch_fill_pk(fb, ctotal, with_codes, st));
        for (int x = 0; x < nrest; ++x) {
            const psa::FillArgs2& fj = jobs[rest_of[x]];
            if (int rc = launch_fill_v2(&fj, 1, nullptr, groups[rest_of[x]], fj.base.ticket, with_codes, psa::kKeepoutAloneBytes, st, false)) return rc;
        }
    } else if (nj > 0) {
        const uint32_t* order_dev = nullptr;
        if (nj > 1) {   // round-robin over the jobs that still have groups left
            const std::vector<unsigned> sig(groups, groups + nj);
            if (sig != e->order_sig) {   // (the same sizes as last time: the order on the device still holds)
                std::vector<uint32_t> order;
                order.reserve(total);
                for (unsigned r = 0; r < rounds; ++r)
                    for (int j = 0; j < nj; ++j)
                        if (r < groups[j]) order.push_back(((uint32_t)j << psa::kOrderJobShift) | r);
                if (e->order_cap < order.size()) {
                    hipFree(e->order);
                    e->order = nullptr; e->order_cap = 0;
                    HIP_TRY(hipMalloc(&e->order, sizeof(uint32_t) * order.size()));
                    e->order_cap = order.size();
                }
                HIP_TRY(hipMemcpyAsync(e->order, order.data(), sizeof(uint32_t) * order.size(), hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));   // (the vector goes out of scope; the copy is a few KB)
                e->order_sig = sig;
            }
            order_dev = e->order;
        }
        HIP_TRY(hipMemsetAsync(e->ticket, 0, 256, st));
        HIP_TRY(hipEventRecord(e->ev[0], st));   // (the events bracket the fill launch itself)
        static const bool dense_ok = [] { const char* v = getenv("PSA_FILL_DENSE"); return !(v && v[0] == '0'); }();
        if (int rc = launch_fill_v2(jobs, nj, order_dev, total, e->ticket, with_codes,
                                    nj > 1 ? psa::kKeepoutManyBytes : psa::kKeepoutAloneBytes, st, nj > 1 && dense_ok)) return rc;
    } else HIP_TRY(hipEventRecord(e->ev[0], st));
    HIP_TRY(hipEventRecord(e->ev[1], st));
    e->launched = true;
    e->walks_queued = false;
    return PSA_OK;
}

namespace {
// The walks of a launch -- one workgroup per pair, all in one launch, each starting from its pair's final pick -- and the
// copies of the results, on the engine's high-priority stream.  Queued by psa_many_wait once the fill HAS ENDED, not at
// the launch behind a wait for the fill's event: a high-priority queue that holds an unsatisfied wait keeps the queue
// scheduler from the normal-priority queues -- with short launches on two or more engines the fills it waited for were
// not dispatched for seconds (launches of 4 096 x 4 096 pairs: waits of 1-58 s; nothing of the kind with the walk streams
// at normal priority, which costs the 2^16 bench 13 %).  A caller in the usual loop (launch the other engines, wait for
// this one) loses nothing: it is in psa_many_wait when the fill ends.
int queue_walks(psa_many* e) {
    const bool with_codes = (e->flags & PSA_F_TRACEBACK) != 0;
    HIP_TRY(hipEventSynchronize(e->ev[1]));
    hipStream_t st = e->walk_stream;
    HIP_TRY(hipEventRecord(e->ev[4], st));
    if (with_codes) {
        psa::TraceBatch tb;
        memset(&tb, 0, sizeof tb);
        int nt = 0;
        for (int k = 0; k < e->slots; ++k) {
            if (e->bound[k] != 1) continue;
            psa_stripe* s = e->st[k];
            if (!walk_fits32(s, s->Q)) return fail(PSA_E_ARG, "psa_many: pair too large for the batched walk");
            if (int rc = make_trace_args(s, TraceOpts(), &tb.job[nt])) return rc;
            tb.job[nt].start_triple = s->small + kFinalWord;
            ++nt;
        }
        tb.njobs = nt;
        if (nt > 0) {
            const char* force = getenv("PSA_TRACEBACK_KERNEL");
            HIP_TRY(psa::launch_traceback_many((force && !strcmp(force, "wave1")) ? psa::kWalkWave : psa::kWalkPair, tb, nt, st));
        }
    }
    HIP_TRY(hipEventRecord(e->ev[2], st));
    psa::CollectBatch cb;
    memset(&cb, 0, sizeof cb);
    cb.out = e->d_res;
    for (int k = 0; k < e->slots; ++k) {
        if (e->bound[k] != 1) continue;
        psa_stripe* s = e->st[k];
        psa::CollectJob& cj = cb.job[cb.njobs++];
        cj.state = with_codes ? s->tb_state : nullptr; cj.triple = s->small + kFinalWord; cj.abort_flag = s->ctrl + kAbortWord; cj.slot = k;
    }
    if (cb.njobs > 0) {
        HIP_TRY(psa::launch_collect(cb, st));
        HIP_TRY(hipMemcpyAsync(e->h_res, e->d_res, sizeof(int32_t) * psa::kResultWords * (size_t)e->slots, hipMemcpyDeviceToHost, st));
    }
    for (int k = 0; k < e->slots && with_codes; ++k) {
        if (e->bound[k] != 1) continue;
        psa_stripe* s = e->st[k];
        const size_t cnt = (size_t)(s->n + s->W + 2);   // (the walk's length is not known on the host yet: all it can be)
        HIP_TRY(hipMemcpyAsync(e->h_ai[k], s->tb_ai, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(e->h_bj[k], s->tb_bj, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(e->ev[3], st));
    e->walks_queued = true;
    return PSA_OK;
}
}  // namespace


int psa_many_poll(psa_many* e) {
    if (!e) return fail(PSA_E_ARG, "psa_many_poll: null engine");
    if (!e->launched) return 1;
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    if (!e->walks_queued) {
        const hipError_t q = hipEventQuery(e->ev[1]);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return fail(PSA_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        if (int rc = queue_walks(e)) return rc;
    }
    const hipError_t q = hipEventQuery(e->ev[3]);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return fail(PSA_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
    return 1;
}

int psa_many_wait(psa_many* e) {
    if (!e) return fail(PSA_E_ARG, "psa_many_wait: null engine");
    if (!e->launched) return PSA_OK;
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    if (!e->walks_queued) if (int rc = queue_walks(e)) return rc;
    HIP_TRY(hipEventSynchronize(e->ev[3]));
    e->launched = false;
    return PSA_OK;
}

int psa_many_result(psa_many* e, int slot, int32_t* score_out, int32_t triple_out[3], int32_t* ai, int32_t* bj, int64_t cap,
                    int64_t* len_out) {
    if (!e || slot < 0 || slot >= e->slots || !score_out) return fail(PSA_E_ARG, "psa_many_result: bad argument");
    if (e->launched) if (int rc = psa_many_wait(e)) return rc;
    if (e->bound[slot] == 0) return fail(PSA_E_STATE, "psa_many_result: slot %d holds no pair", slot);
    const int64_t n = e->n[slot], m = e->m[slot];
    int32_t t[3];
    if (n == 0) { t[0] = -1; t[1] = 0; t[2] = (int32_t)-(2 + m); }
    else if (m == 0) { t[0] = -1; t[1] = (int32_t)-(2 + n); t[2] = -1; }
    else {
        const int32_t* r = e->h_res + (size_t)psa::kResultWords * slot;
        if (r[27]) return fail(PSA_E_KERNEL, "fill kernel: a strip's wait for the strip above expired (slot %d)", slot);
        memcpy(t, r + 24, sizeof t);
    }
    pick3(t, score_out);
    if (triple_out) memcpy(triple_out, t, sizeof t);
    if (!(e->flags & PSA_F_TRACEBACK) || !ai || !bj) { if (len_out) *len_out = 0; return PSA_OK; }
    if (!len_out) return fail(PSA_E_ARG, "psa_many_result: len_out is null");
    int64_t len = 0;
    if (n > 0 && m > 0) {
        int64_t stt[12];
        memcpy(stt, e->h_res + (size_t)psa::kResultWords * slot, sizeof stt);
        if (stt[5] == 2) return fail(PSA_E_KERNEL, "traceback: winner codes failed the decode check (slot %d)", slot);
        if (stt[5] == 3) return fail(PSA_E_KERNEL, "traceback: the walk waited in vain for a code tile (slot %d)", slot);
        if (stt[5]) return fail(PSA_E_KERNEL, "traceback: device buffer overflow (slot %d)", slot);
        if (!stt[3]) return fail(PSA_E_KERNEL, "traceback: the walk stopped before the matrix boundary (slot %d)", slot);
        len = stt[4];
        if (len + n + m > cap) return fail(PSA_E_ARG, "psa_many_result: output capacity %lld too small", (long long)cap);
        memcpy(ai, e->h_ai[slot], sizeof(int32_t) * (size_t)len);
        memcpy(bj, e->h_bj[slot], sizeof(int32_t) * (size_t)len);
    }
    return psa_finish_alignment(ai, bj, len, cap, len_out);
}

int psa_many_last_ms(psa_many* e, float* fill_ms_out, float* walk_ms_out) {
    if (!e) return fail(PSA_E_ARG, "psa_many_last_ms: null engine");
    if (e->launched) if (int rc = psa_many_wait(e)) return rc;
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    if (fill_ms_out) HIP_TRY(hipEventElapsedTime(fill_ms_out, e->ev[0], e->ev[1]));
    if (walk_ms_out) HIP_TRY(hipEventElapsedTime(walk_ms_out, e->ev[4], e->ev[2]));
    return PSA_OK;
}

int psa_many_fill_ms_since(psa_many* e, void* ref_event, float* begin_ms_out, float* end_ms_out) {
    if (!e || !ref_event) return fail(PSA_E_ARG, "psa_many_fill_ms_since: null argument");
    if (e->launched) if (int rc = psa_many_wait(e)) return rc;
    DeviceGuard g(e->device);
    if (g.rc) return g.rc;
    if (begin_ms_out) HIP_TRY(hipEventElapsedTime(begin_ms_out, (hipEvent_t)ref_event, e->ev[0]));
    if (end_ms_out) HIP_TRY(hipEventElapsedTime(end_ms_out, (hipEvent_t)ref_event, e->ev[1]));
    return PSA_OK;
}

int64_t psa_many_code_bytes(const psa_many* e, int slot) {
    return (e && slot >= 0 && slot < e->slots && e->bound[slot] == 1) ? (int64_t)e->st[slot]->code_bytes : 0;
}

// The engines of the thread's last psa_align_many call, kept for its next one: allocating and freeing a quarter of a
// terabyte of code buffers per call took 0.4 s the first and 3.8 s the second time for 60 pairs of 2^16 x 2^16 that need
// 0.08 s of GPU time.  psa_release_cache() frees them.
struct ManyCache {
    psa_many* eng[3] = {nullptr, nullptr, nullptr};
    int E = 0, slots = 0, device = -1;
    int64_t n_cap = 0, m_cap = 0;
    uint32_t flags = 0;
};
thread_local ManyCache* g_many = nullptr;
}  // extern "C" (closed for the helper below)
void psa_make_room(double need) {
    if (!g_many) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > 0.7 * (double)free_b) psa_drop_many_cache_impl();
}
void psa_drop_many_cache_impl() {
    if (!g_many) return;
    for (auto*& e : g_many->eng) { free_many(e); e = nullptr; }
    delete g_many;
    g_many = nullptr;
}
extern "C" {

// run() in a loop (simple.cpp:311 called once per pair): the pairs go through two engines that take
// turns, so that the walks and copies of one batch run beside the fill of the next.
int psa_align_many(int64_t npairs, const uint8_t* const* a, const int64_t* n, const uint8_t* const* b, const int64_t* m,
                   int32_t* scores_out, int32_t** ai_out, int32_t** bj_out, int64_t* len_out) {
    if (npairs < 0 || (npairs && (!a || !n || !b || !m || !scores_out))) return fail(PSA_E_ARG, "psa_align_many: bad argument");
    const bool want_al = ai_out && bj_out && len_out;
    if (npairs == 0) return PSA_OK;
    if (int rc = need_device()) return rc;
    int64_t n_cap = 1, m_cap = 1;
    for (int64_t k = 0; k < npairs; ++k) {
        if (n[k] < 0 || m[k] < 0 || (n[k] && !a[k]) || (m[k] && !b[k])) return fail(PSA_E_ARG, "psa_align_many: pair %lld is malformed", (long long)k);
        if (n[k] > (1LL << 27) || m[k] > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
        n_cap = std::max(n_cap, n[k]); m_cap = std::max(m_cap, m[k]);
        if (want_al) { ai_out[k] = bj_out[k] = nullptr; len_out[k] = 0; }
    }
    // Engines and slots.  A fill launch costs about 4 ms beyond 1.75 ms per 2^16 pair (its ramp and drain), so as few
    // launches as the argument block allows (kMaxJobs pairs each), of equal size.  Three engines taking turns where
    // there are that many launches and the device memory holds them -- each fill is then queued while the one before it
    // runs and takes over the chip as that one drains, and nobody waits for an engine's walks, which get their turn at
    // the next drain --, else two, each fill after the other's end (the walks and copies of one beside the fill of the
    // other); slots before engines (20 + 20 rather than 3 x 13).  70 % of the free memory at most.
    release_pair_caches();
    const uint32_t eflags = want_al ? PSA_F_TRACEBACK : 0;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    auto slot_bytes = [](bool codes, int64_t nc, int64_t mc) {
        return (codes ? (double)nc * (double)mc * 1.05 : 0.0) + 64.0 * (double)(nc + mc) + (1 << 20);
    };
    // (engines kept from the last call can serve this one if they are of its kind and large enough; their memory counts
    // as free for the plan, and they go if the plan wants more than they are)
    bool compatible = g_many && g_many->device == device && g_many->flags == eflags && g_many->n_cap >= n_cap && g_many->m_cap >= m_cap;
    if (g_many && !compatible) psa_drop_many_cache_impl();
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const double per_slot = compatible ? slot_bytes(want_al, g_many->n_cap, g_many->m_cap) : slot_bytes(want_al, n_cap, m_cap);
    const double held_b = compatible ? per_slot * g_many->E * g_many->slots : 0.0;
    const int64_t fit = (int64_t)std::min<double>(1e6, 0.7 * ((double)free_b + held_b) / per_slot);   // slots the memory holds
    int slots = (int)std::min<int64_t>(std::min<int64_t>(psa::kMaxJobs, fit), npairs);
    int E = 1;
    if (slots >= 1) {
        int64_t launches = (npairs + slots - 1) / slots;
        if (launches > 1 && fit < 2 * (int64_t)slots) {   // more than one launch: two engines at least, so that walks overlap fills
            slots = (int)std::max<int64_t>(1, fit / 2);
            launches = (npairs + slots - 1) / slots;
        }
        E = (int)std::min<int64_t>(std::min<int64_t>(3, launches), std::max<int64_t>(1, fit / slots));
        slots = (int)((npairs + launches - 1) / launches);
    }
    if (slots < 1) {
        // not even one pair per engine fits with its codes: one at a time through psa_align (linear space if need be)
        for (int64_t k = 0; k < npairs; ++k) {
            if (want_al) { if (int rc = psa_align(a[k], n[k], b[k], m[k], 1, &scores_out[k], &ai_out[k], &bj_out[k], &len_out[k])) return rc; }
            else if (int rc = psa_score(a[k], n[k], b[k], m[k], 1, &scores_out[k], nullptr)) return rc;
        }
        return PSA_OK;
    }
    psa_many* eng[3] = {nullptr, nullptr, nullptr};
    std::vector<int64_t> held[3];
    int rc = PSA_OK;
    auto collect = [&](int x) -> int {
        if (int r = psa_many_wait(eng[x])) return r;
        for (size_t sl = 0; sl < held[x].size(); ++sl) {
            const int64_t k = held[x][sl];
            if (want_al) {
                const int64_t cap = 2 * (n[k] + m[k]) + 2;
                ai_out[k] = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
                bj_out[k] = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
                if (!ai_out[k] || !bj_out[k]) return fail(PSA_E_NOMEM, "psa_align_many: host allocation failed");
                if (int r = psa_many_result(eng[x], (int)sl, &scores_out[k], nullptr, ai_out[k], bj_out[k], cap, &len_out[k])) return r;
            } else if (int r = psa_many_result(eng[x], (int)sl, &scores_out[k], nullptr, nullptr, nullptr, 0, nullptr)) return r;
            // (an engine whose fill has ended meanwhile gets its walks queued now, not when its turn to be collected comes)
            for (int y = 0; y < 3; ++y) if (y != x && eng[y]) { const int pr = psa_many_poll(eng[y]); if (pr < 0) return pr; }
        }
        held[x].clear();
        return PSA_OK;
    };
    if (compatible && (g_many->E < E || g_many->slots < slots)) { psa_drop_many_cache_impl(); compatible = false; }
    if (compatible) {
        for (int x = 0; x < E; ++x) eng[x] = g_many->eng[x];
    } else {
        g_many = new ManyCache();
        g_many->device = device; g_many->flags = eflags; g_many->n_cap = n_cap; g_many->m_cap = m_cap;
        for (int x = 0; x < E && rc == PSA_OK; ++x) {
            rc = psa_many_create(&eng[x], slots, n_cap, m_cap, eflags);
            g_many->eng[x] = eng[x];
        }
        g_many->E = E; g_many->slots = slots;
    }
    // pairs of the same size side by side: a launch fills them two per lane (fill_strips_pk)
    std::vector<int64_t> by_size((size_t)npairs);
    for (int64_t k = 0; k < npairs; ++k) by_size[(size_t)k] = k;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int64_t x, int64_t y) { return n[x] != n[y] ? n[x] < n[y] : m[x] < m[y]; });
    int turn = 0;
    for (int64_t k0 = 0; k0 < npairs && rc == PSA_OK; k0 += slots, turn = (turn + 1) % E) {
        if (!held[turn].empty()) rc = collect(turn);
        for (int sl = 0; sl < g_many->slots && rc == PSA_OK; ++sl) {   // (kept engines may have more slots than this call uses)
            if (sl < slots && k0 + sl < npairs) {
                const int64_t k = by_size[(size_t)(k0 + sl)];
                rc = psa_many_set_pair(eng[turn], sl, a[k], n[k], b[k], m[k]);
                held[turn].push_back(k);
            } else rc = psa_many_clear_slot(eng[turn], sl);
        }
        if (rc == PSA_OK) rc = (E == 2) ? psa_many_launch_after(eng[turn], eng[turn ^ 1]) : psa_many_launch(eng[turn]);
    }
    for (int x = 0; x < E && rc == PSA_OK; ++x) if (!held[(turn + x) % E].empty()) rc = collect((turn + x) % E);
    if (rc != PSA_OK) {
        psa_drop_many_cache_impl();   // (engines in an unknown state: gone)
        if (want_al)
            for (int64_t k = 0; k < npairs; ++k) { free(ai_out[k]); free(bj_out[k]); ai_out[k] = bj_out[k] = nullptr; }
    }
    return rc;
}

int psa_score_batch(const uint8_t* seqs, const int64_t* offsets, int64_t nseqs, const int32_t* pairs,
                    int64_t npairs, int32_t* scores_out) {
    if (npairs < 0 || nseqs < 0 || (npairs && (!seqs || !offsets || !pairs || !scores_out)))
        return fail(PSA_E_ARG, "psa_score_batch: bad argument");
    if (npairs == 0) return PSA_OK;
    if (int rc = need_device()) return rc;
    // pairs whose B exceeds the LDS row buffer go through the single-pair path
    std::vector<int64_t> big;
    for (int64_t k = 0; k < npairs; ++k) {
        const int32_t sa = pairs[2 * k], sb = pairs[2 * k + 1];
        if (sa < 0 || sb < 0 || sa >= nseqs || sb >= nseqs) return fail(PSA_E_ARG, "psa_score_batch: pair %lld names a missing sequence", (long long)k);
        if (offsets[sb + 1] - offsets[sb] > psa::kBatchMaxLen) big.push_back(k);
    }
    const size_t total = (size_t)offsets[nseqs];
    uint8_t* d_seqs = nullptr; int64_t* d_off = nullptr; int32_t* d_pairs = nullptr; int32_t* d_scores = nullptr;
    unsigned* d_ctr = nullptr;
    int rc = PSA_OK;
    auto cleanup = [&]() { hipFree(d_seqs); hipFree(d_off); hipFree(d_pairs); hipFree(d_scores); hipFree(d_ctr); };
#define B_TRY(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) { cleanup(); return fail(PSA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
    B_TRY(hipMalloc(&d_seqs, total + 16));
    B_TRY(hipMalloc(&d_off, sizeof(int64_t) * (size_t)(nseqs + 1)));
    B_TRY(hipMalloc(&d_pairs, sizeof(int32_t) * 2 * (size_t)npairs));
    B_TRY(hipMalloc(&d_scores, sizeof(int32_t) * (size_t)npairs));
    B_TRY(hipMalloc(&d_ctr, 256));
    B_TRY(hipMemcpy(d_seqs, seqs, total, hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(d_off, offsets, sizeof(int64_t) * (size_t)(nseqs + 1), hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)npairs, hipMemcpyHostToDevice));
    int max_b = 0;
    for (int64_t k = 0; k < npairs; ++k) {
        const int64_t lb = offsets[pairs[2 * k + 1] + 1] - offsets[pairs[2 * k + 1]];
        if (lb <= psa::kBatchMaxLen) max_b = std::max(max_b, (int)lb);
    }
    rc = psa_score_batch_dev(d_seqs, d_off, d_pairs, npairs, max_b, d_scores, d_ctr, nullptr);
    if (rc == PSA_OK) B_TRY(hipMemcpy(scores_out, d_scores, sizeof(int32_t) * (size_t)npairs, hipMemcpyDeviceToHost));
#undef B_TRY
    cleanup();
    if (rc != PSA_OK) return rc;
    for (int64_t k : big) {
        const int32_t sa = pairs[2 * k], sb = pairs[2 * k + 1];
        rc = psa_score(seqs + offsets[sa], offsets[sa + 1] - offsets[sa], seqs + offsets[sb], offsets[sb + 1] - offsets[sb], 1,
                       &scores_out[k], nullptr);
        if (rc != PSA_OK) return rc;
    }
    return PSA_OK;
}

// simple.cpp:312 + output_alignement (simple.cpp:43-70); both Length lines print al_A's size.
char* psa_format_result(int32_t score, const uint8_t* a, const uint8_t* b, const int32_t* ai, const int32_t* bj,
                        int64_t len, int64_t* out_len) {
    std::string out = "Score: " + std::to_string(score) + "\n";
    out += "Length Sequence A : " + std::to_string(len) + "\n";
    out += "Length Sequence B : " + std::to_string(len) + "\n";
    std::string ra((size_t)len, '-'), rb((size_t)len, '-');
    for (int64_t k = 0; k < len; ++k) {
        if (ai[k] != -1) ra[(size_t)k] = (char)a[ai[k]];
        if (bj[k] != -1) rb[(size_t)k] = (char)b[bj[k]];
    }
    out += "Sequence A : " + ra + "\nSequence B : " + rb + "\n";
    char* buf = (char*)malloc(out.size() + 1);
    if (!buf) return nullptr;
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    if (out_len) *out_len = (int64_t)out.size();
    return buf;
}

// read_fasta.cpp:8-23: keep every non-empty line that does not start with '>', verbatim.
uint8_t* psa_read_fasta(const char* path, int64_t* len_out) {
    if (len_out) *len_out = 0;
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) {
        fprintf(stderr, "Error opening file: %s\n", path ? path : "(null)");
        return nullptr;
    }
    std::string seq, line;
    int ch;
    bool any = false;
    auto flush = [&]() {
        if (!line.empty() && line[0] != '>') seq += line;
        line.clear();
    };
    while ((ch = fgetc(f)) != EOF) {
        any = true;
        if (ch == '\n') flush(); else line.push_back((char)ch);
    }
    if (any) flush();
    fclose(f);
    uint8_t* buf = (uint8_t*)malloc(seq.size() + 1);
    if (!buf) return nullptr;
    memcpy(buf, seq.data(), seq.size());
    if (len_out) *len_out = (int64_t)seq.size();
    return buf;
}

int psa_read_fasta_records(const char* path, uint8_t** seqs_out, int64_t** offsets_out, int64_t* nrec_out) {
    if (!seqs_out || !offsets_out || !nrec_out) return fail(PSA_E_ARG, "psa_read_fasta_records: null argument");
    *seqs_out = nullptr; *offsets_out = nullptr; *nrec_out = 0;
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) {
        fprintf(stderr, "Error opening file: %s\n", path ? path : "(null)");
        return fail(PSA_E_ARG, "psa_read_fasta_records: cannot open %s", path ? path : "(null)");
    }
    std::string seq, line;
    std::vector<int64_t> starts;   // offset at which each record begins
    int ch;
    bool any = false;
    auto flush = [&]() {
        if (!line.empty() && line[0] == '>') starts.push_back((int64_t)seq.size());
        else if (!line.empty()) {
            if (starts.empty()) starts.push_back(0);   // residues before any header: record 0
            seq += line;
        }
        line.clear();
    };
    while ((ch = fgetc(f)) != EOF) {
        any = true;
        if (ch == '\n') flush(); else line.push_back((char)ch);
    }
    if (any) flush();
    fclose(f);
    uint8_t* buf = (uint8_t*)malloc(seq.size() + 1);
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (starts.size() + 1));
    if (!buf || !off) { free(buf); free(off); return fail(PSA_E_NOMEM, "psa_read_fasta_records: out of memory"); }
    memcpy(buf, seq.data(), seq.size());
    for (size_t r = 0; r < starts.size(); ++r) off[r] = starts[r];
    off[starts.size()] = (int64_t)seq.size();
    *seqs_out = buf; *offsets_out = off; *nrec_out = (int64_t)starts.size();
    return PSA_OK;
}

}  // extern "C"

// ---- batch input resident on the device: FASTA records / directories -> seqs + offsets in HBM ----
struct psa_seqset {
    int device = 0;
    int64_t count = 0, total = 0;
    int max_len = 0;
    uint8_t* h_seqs = nullptr;     // pinned host staging (kept: the fallback for over-long pairs reads it)
    int64_t* h_off = nullptr;      // pinned
    uint8_t* d_seqs = nullptr;
    int64_t* d_off = nullptr;
    std::vector<std::string> names;   // header line of every record (without '>'), or the file name
};

namespace {

// one file: every '>' record its own sequence, line handling of read_fasta.cpp:16-19 (see psa_read_fasta_records)
int append_fasta_file(const std::string& path, std::string& seq, std::vector<int64_t>& starts, std::vector<std::string>& names) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "Error opening file: %s\n", path.c_str());
        return fail(PSA_E_ARG, "psa_seqset_load: cannot open %s", path.c_str());
    }
    std::string line;
    int ch;
    bool any = false, have_record = false;
    auto flush = [&]() {
        if (!line.empty() && line[0] == '>') { starts.push_back((int64_t)seq.size()); names.push_back(line.substr(1)); have_record = true; }
        else if (!line.empty()) {
            if (!have_record) { starts.push_back((int64_t)seq.size()); names.push_back(path); have_record = true; }   // residues before any header
            seq += line;
        }
        line.clear();
    };
    while ((ch = fgetc(f)) != EOF) {
        any = true;
        if (ch == '\n') flush(); else line.push_back((char)ch);
    }
    if (any) flush();
    fclose(f);
    return PSA_OK;
}

__global__ void block_pairs_kernel(int32_t* pairs, int q0, int nq, int t0, int nt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (int64_t)nq * nt) return;
    pairs[2 * k] = q0 + (int32_t)(k / nt);
    pairs[2 * k + 1] = t0 + (int32_t)(k % nt);
}

}  // namespace

#include <dirent.h>
#include <sys/stat.h>

extern "C" {

void psa_seqset_destroy(psa_seqset* s) {
    if (!s) return;
    DeviceGuard g(s->device);
    hipFree(s->d_seqs); hipFree(s->d_off);
    hipHostFree(s->h_seqs); hipHostFree(s->h_off);
    delete s;
}

int psa_seqset_load(psa_seqset** out, const char* const* paths, int npaths) {
    if (!out || npaths < 0 || (npaths && !paths)) return fail(PSA_E_ARG, "psa_seqset_load: bad argument");
    *out = nullptr;
    if (int rc = need_device()) return rc;
    std::string seq;
    std::vector<int64_t> starts;
    std::vector<std::string> names;
    for (int k = 0; k < npaths; ++k) {
        if (!paths[k]) return fail(PSA_E_ARG, "psa_seqset_load: null path");
        struct stat sb;
        if (stat(paths[k], &sb) == 0 && S_ISDIR(sb.st_mode)) {
            // a directory: every regular file in it, in byte order of the names
            std::vector<std::string> files;
            if (DIR* d = opendir(paths[k])) {
                while (struct dirent* de = readdir(d)) {
                    const std::string full = std::string(paths[k]) + "/" + de->d_name;
                    struct stat fs;
                    if (stat(full.c_str(), &fs) == 0 && S_ISREG(fs.st_mode)) files.push_back(full);
                }
                closedir(d);
            } else return fail(PSA_E_ARG, "psa_seqset_load: cannot read directory %s", paths[k]);
            std::sort(files.begin(), files.end());
            for (auto& f : files) if (int rc = append_fasta_file(f, seq, starts, names)) return rc;
        } else if (int rc = append_fasta_file(paths[k], seq, starts, names)) return rc;
    }
    psa_seqset* s = new psa_seqset();
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    s->count = (int64_t)starts.size(); s->total = (int64_t)seq.size(); s->names = names;
#define SET_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) { psa_seqset_destroy(s); return fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
    SET_TRY(hipHostMalloc(&s->h_seqs, seq.size() + 64));
    SET_TRY(hipHostMalloc(&s->h_off, sizeof(int64_t) * (starts.size() + 1)));
    memcpy(s->h_seqs, seq.data(), seq.size());
    for (size_t r = 0; r < starts.size(); ++r) s->h_off[r] = starts[r];
    s->h_off[starts.size()] = (int64_t)seq.size();
    for (int64_t r = 0; r < s->count; ++r) s->max_len = (int)std::max<int64_t>(s->max_len, s->h_off[r + 1] - s->h_off[r]);
    SET_TRY(hipMalloc(&s->d_seqs, seq.size() + 64));
    SET_TRY(hipMalloc(&s->d_off, sizeof(int64_t) * (starts.size() + 1)));
    SET_TRY(hipMemcpyAsync(s->d_seqs, s->h_seqs, seq.size(), hipMemcpyHostToDevice, nullptr));
    SET_TRY(hipMemcpyAsync(s->d_off, s->h_off, sizeof(int64_t) * (starts.size() + 1), hipMemcpyHostToDevice, nullptr));
    SET_TRY(hipStreamSynchronize(nullptr));
#undef SET_TRY
    *out = s;
    return PSA_OK;
}

int64_t psa_seqset_count(const psa_seqset* s) { return s ? s->count : 0; }
int64_t psa_seqset_length(const psa_seqset* s, int64_t i) { return (s && i >= 0 && i < s->count) ? s->h_off[i + 1] - s->h_off[i] : -1; }
const char* psa_seqset_name(const psa_seqset* s, int64_t i) { return (s && i >= 0 && i < s->count) ? s->names[(size_t)i].c_str() : nullptr; }
const uint8_t* psa_seqset_seqs_dev(const psa_seqset* s) { return s ? s->d_seqs : nullptr; }
const int64_t* psa_seqset_offsets_dev(const psa_seqset* s) { return s ? s->d_off : nullptr; }
const uint8_t* psa_seqset_residues(const psa_seqset* s, int64_t i) { return (s && i >= 0 && i < s->count) ? s->h_seqs + s->h_off[i] : nullptr; }

// scores_out[(q - q0) * (t1 - t0) + (t - t0)] = result.score of run(p, record q, record t)
int psa_seqset_score_block(const psa_seqset* s, int64_t q0, int64_t q1, int64_t t0, int64_t t1, int32_t* scores_out) {
    if (!s || !scores_out || q0 < 0 || q1 < q0 || q1 > s->count || t0 < 0 || t1 < t0 || t1 > s->count)
        return fail(PSA_E_ARG, "psa_seqset_score_block: bad argument");
    const int64_t nq = q1 - q0, nt = t1 - t0, np = nq * nt;
    if (np == 0) return PSA_OK;
    if (np > (1LL << 30)) return fail(PSA_E_ARG, "psa_seqset_score_block: at most 2^30 pairs per call");
    DeviceGuard g(s->device);
    if (g.rc) return g.rc;
    int32_t *d_pairs = nullptr, *d_scores = nullptr;
    unsigned* d_ctr = nullptr;
    auto cleanup = [&]() { hipFree(d_pairs); hipFree(d_scores); hipFree(d_ctr); };
#define BLK_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) { cleanup(); return fail(PSA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
    BLK_TRY(hipMalloc(&d_pairs, sizeof(int32_t) * 2 * (size_t)np));
    BLK_TRY(hipMalloc(&d_scores, sizeof(int32_t) * (size_t)np));
    BLK_TRY(hipMalloc(&d_ctr, 256));
    hipLaunchKernelGGL(block_pairs_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, nullptr, d_pairs, (int)q0, (int)nq, (int)t0, (int)nt);
    int max_b = 0;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t lb = s->h_off[t + 1] - s->h_off[t];
        if (lb <= psa::kBatchMaxLen) max_b = std::max(max_b, (int)lb);
    }
    int rc = psa_score_batch_dev(s->d_seqs, s->d_off, d_pairs, np, max_b, d_scores, d_ctr, nullptr);
    if (rc == PSA_OK) BLK_TRY(hipMemcpy(scores_out, d_scores, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost));
#undef BLK_TRY
    cleanup();
    if (rc != PSA_OK) return rc;
    // targets longer than the batch kernel's row buffer: through the single-pair path
    for (int64_t t = t0; t < t1; ++t) {
        if (s->h_off[t + 1] - s->h_off[t] <= psa::kBatchMaxLen) continue;
        for (int64_t q = q0; q < q1; ++q) {
            rc = psa_score(s->h_seqs + s->h_off[q], s->h_off[q + 1] - s->h_off[q], s->h_seqs + s->h_off[t], s->h_off[t + 1] - s->h_off[t], 1,
                           &scores_out[(q - q0) * nt + (t - t0)], nullptr);
            if (rc != PSA_OK) return rc;
        }
    }
    return PSA_OK;
}

void psa_synth_protein(uint64_t seed, int64_t n, uint8_t* out) {
    static const char alphabet[] = "ACDEFGHIKLMNPQRSTVWY";
    std::mt19937_64 g(seed);
    for (int64_t k = 0; k < n; ++k) out[k] = (uint8_t)alphabet[g() % 20];
}

}  // extern "C"
