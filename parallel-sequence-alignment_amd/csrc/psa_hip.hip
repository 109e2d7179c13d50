// psa_hip.hip -- C ABI (include/psa.h) over the gfx950 kernels.  No CPU fallback: every
// compute entry point needs a HIP device and fails with PSA_E_NO_DEVICE otherwise.
#include "../../include/psa.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "psa_fill_v2.hip.h"
#include "psa_batch.hip.h"

namespace {

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

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                       \
    } while (0)

int need_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt < 1)
        return fail(PSA_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PSA_OK;
}

// simple.cpp:162/166/170: table 1 if t1>=t2 && t1>=t3, else 2 if t2>=t3, else 3.
int pick3(const int32_t t[3], int32_t* v) {
    if (t[0] >= t[1] && t[0] >= t[2]) { *v = t[0]; return 1; }
    if (t[1] >= t[2]) { *v = t[1]; return 2; }
    *v = t[2];
    return 3;
}

// control block: bytes 0..255 are per launch (ticket, front, statistics), bytes 256..511 per pass (the abort
// flag: a hand-off wait that expired in ANY chunk of the pass stays visible until the pass is over)
constexpr size_t kSeqPad = 64;
constexpr size_t kCtrlLaunchBytes = 256;
constexpr size_t kCtrlBytes = 512;
constexpr int kAbortWord = 64;

// rows per band in banded mode: PSA_BAND_ROWS, else as many strips as 8 GiB of codes hold
thread_local int64_t g_band_rows_request = 0;   // set by psa_align_banded around the stripe's creation
int64_t band_rows_for(int64_t n, int Q) {
    int64_t want = g_band_rows_request;
    if (want <= 0) if (const char* e = getenv("PSA_BAND_ROWS")) want = atoll(e);
    if (want <= 0) want = 64 * std::max<int64_t>(1, (8LL << 30) / ((int64_t)Q * psa::kTileBytes));
    const int64_t wg_rows = 64 * psa::kNW;   // band boundaries coincide with workgroup boundaries
    want = std::max<int64_t>(wg_rows, ((want + wg_rows - 1) / wg_rows) * wg_rows);
    return std::min<int64_t>(want, ((n + wg_rows - 1) / wg_rows) * wg_rows);
}

}  // namespace

struct psa_stripe {
    int64_t n = 0, m = 0, c0 = 0, c1 = 0;
    int W = 0, Q = 0, Wpad = 0;
    uint32_t flags = 0;
    uint8_t* a_dev = nullptr;
    uint8_t* b_dev = nullptr;    // = b_alloc + kSeqPad: the traceback's residue windows start up to 64 bytes in front
    uint8_t* b_alloc = nullptr;
    psa::u64* ring = nullptr;
    uint8_t* codes = nullptr;    // winner codes, one byte per cell in 4 KB tiles (psa_device.hip.h)
    int2* leftcol = nullptr;     // c0 > 0: (D, H') of column c0 per row, kept for the traceback
    size_t code_bytes = 0;
    int32_t* small = nullptr;    // corner A[3] (pad 4), corner B[3] (pad 4), final[3] (pad 4)
    unsigned* ctrl = nullptr;    // [0] ticket, [32] front, [64] abort flag (kCtrlBytes)
    int32_t* tb_ai = nullptr;
    int32_t* tb_bj = nullptr;
    int64_t tb_cap = 0;
    int64_t* tb_state = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t next_r0 = 0;
    int corner_cur = 0;
    // v2 kernel: symbol ids of A and the per-symbol substitution table of this stripe's B
    uint8_t* a_ids = nullptr;
    uint8_t* alpha = nullptr;    // present[256] (u32) | meta[4] (u32) | map[256] | inv[32]
    uint32_t* eqtab = nullptr;
    int tab_pitch = 0;
    unsigned nsym = 0;
    bool use_v2 = false;
    bool fill_timed = false, tb_timed = false;
    // banded (linear-space) mode: codes cover band_rows rows at a time, ckpt holds the hand-off ring as it
    // was before each band (see align_banded)
    int64_t band_rows = 0;
    psa::u64* ckpt = nullptr;
    int device = 0;              // the HIP device the stripe lives on (current when it was created)
};

namespace {
// The current HIP device is a per-thread setting and a new host thread starts on device 0: every call
// that works on a stripe first makes the stripe's own device current (callers keep one thread per
// alignment in flight, bench.py for one).
int enter_device(const psa_stripe* s) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != s->device) HIP_TRY(hipSetDevice(s->device));
    return PSA_OK;
}
// what one fill launch may override (banded mode)
struct FillOpts {
    int codes = -1;                  // -1: as the stripe was created; 0 / 1: without / with winner codes
    int W = 0;                       // > 0: only the stripe's first W columns
    uint8_t* codes_base = nullptr;   // where strip 0's codes would lie
    bool any_order = false;          // do not insist on r0 == end of the previous chunk
};
int fill_rows_impl(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev, void* stream,
                   const FillOpts& o);
}  // namespace

namespace {

constexpr size_t kAlphaBytes = 256 * 4 + 16 + 256 + 32;

// FASTA-to-device encoding for the v2 kernel: symbol ids of A, substitution table of B.
int rebuild_tables(psa_stripe* s, hipStream_t st) {
    unsigned* present = (unsigned*)s->alpha;
    unsigned* meta = present + 256;
    uint8_t* map = (uint8_t*)(meta + 4);
    uint8_t* inv = map + 256;
    HIP_TRY(hipMemsetAsync(s->alpha, 0, kAlphaBytes, st));
    const int blocks = (int)std::min<int64_t>(1024, (s->n + 255) / 256);
    hipLaunchKernelGGL(psa::alphabet_presence, dim3(blocks), dim3(256), 0, st, s->a_dev, s->n, present);
    hipLaunchKernelGGL(psa::alphabet_compact, dim3(1), dim3(64), 0, st, present, meta, map, inv);
    hipLaunchKernelGGL(psa::encode_rows, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, st, s->a_dev, s->n, map,
                       s->a_ids);
    hipLaunchKernelGGL(psa::build_eqtab, dim3((unsigned)((s->tab_pitch + 255) / 256), psa::kMaxSym), dim3(256), 0, st,
                       s->b_dev, s->W, meta, inv, s->eqtab, s->tab_pitch);
    HIP_TRY(hipGetLastError());
    unsigned k = 0;
    HIP_TRY(hipMemcpyAsync(&k, meta, sizeof k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    s->nsym = k;
    const char* force = getenv("PSA_FILL_KERNEL");
    s->use_v2 = k <= (unsigned)psa::kMaxSym && !(force && !strcmp(force, "v1"));
    return PSA_OK;
}

}  // namespace

extern "C" {

const char* psa_last_error(void) { return g_err.c_str(); }
const char* psa_version(void) { return "psa-hip 0.1 (gfx950)"; }

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

void psa_stripe_destroy(psa_stripe* s) {
    if (!s) return;
    enter_device(s);
    hipFree(s->a_dev); hipFree(s->b_alloc); hipFree(s->ring); hipFree(s->codes); hipFree(s->leftcol);
    hipFree(s->a_ids); hipFree(s->alpha); hipFree(s->eqtab);
    hipFree(s->ckpt);
    hipFree(s->small); hipFree(s->ctrl); hipFree(s->tb_ai); hipFree(s->tb_bj); hipFree(s->tb_state);
    for (auto& e : s->ev) if (e) hipEventDestroy(e);
    delete s;
}

int psa_stripe_create(psa_stripe** out, const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                      int64_t c0, int64_t c1, uint32_t flags) {
    if (!out) return fail(PSA_E_ARG, "psa_stripe_create: out is null");
    *out = nullptr;
    if (n < 1 || m < 1 || c0 < 0 || c1 <= c0 || c1 > m || !a || !b)
        return fail(PSA_E_ARG, "psa_stripe_create: need n>=1, m>=1, 0<=c0<c1<=m (n=%lld m=%lld c0=%lld c1=%lld)",
                    (long long)n, (long long)m, (long long)c0, (long long)c1);
    if (n > (1LL << 27) || m > (1LL << 27)) return fail(PSA_E_ARG, "sequence longer than 2^27 is not supported");
    if (int rc = need_device()) return rc;
    psa_stripe* s = new psa_stripe();
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
    s->n = n; s->m = m; s->c0 = c0; s->c1 = c1; s->flags = flags;
    s->W = (int)(c1 - c0);
    s->Q = (s->W + 63 + 63) / 64;
    s->Wpad = ((s->W + 63) / 64) * 64;
    const int64_t strips = (n + 63) / 64;
#define CREATE_TRY(expr)                                                          \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            psa_stripe_destroy(s);                                                \
            return fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));           \
        }                                                                         \
    } while (0)
    // (both padded: the traceback fetches whole 64-row / 128-column residue windows beside each code tile)
    CREATE_TRY(hipMalloc(&s->a_dev, (size_t)n + 2 * kSeqPad));
    CREATE_TRY(hipMalloc(&s->b_alloc, (size_t)s->W + 4 * kSeqPad));
    s->b_dev = s->b_alloc + kSeqPad;
    CREATE_TRY(hipMemcpy(s->a_dev, a, (size_t)n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(s->b_dev, b + c0, (size_t)s->W, hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&s->ring, sizeof(psa::u64) * (2 * (size_t)s->Wpad + 64)));  // + dump slots, see fill_strips_v2
    CREATE_TRY(hipMalloc(&s->small, 64));
    CREATE_TRY(hipMalloc(&s->ctrl, kCtrlBytes));
    if (flags & PSA_F_TRACEBACK) {
        int64_t code_strips = strips;
        if (flags & PSA_F_BANDED) {
            s->band_rows = band_rows_for(n, s->Q);
            code_strips = s->band_rows / 64;
            const int64_t nbands = (n + s->band_rows - 1) / s->band_rows;
            CREATE_TRY(hipMalloc(&s->ckpt, sizeof(psa::u64) * (size_t)(nbands + 1) * (size_t)s->Wpad));
        }
        s->code_bytes = (size_t)code_strips * (size_t)s->Q * psa::kTileBytes;
        CREATE_TRY(hipMalloc(&s->codes, s->code_bytes));
        if (c0 > 0) CREATE_TRY(hipMalloc(&s->leftcol, sizeof(int2) * (size_t)(n + 1)));
        s->tb_cap = n + s->W + 2;
        CREATE_TRY(hipMalloc(&s->tb_ai, sizeof(int32_t) * (size_t)s->tb_cap));
        CREATE_TRY(hipMalloc(&s->tb_bj, sizeof(int32_t) * (size_t)s->tb_cap));
        CREATE_TRY(hipMalloc(&s->tb_state, sizeof(int64_t) * 12));
    }
    for (auto& e : s->ev) CREATE_TRY(hipEventCreate(&e));
    s->tab_pitch = psa::kTabPad + (64 * s->Q + 64) / 4 + 40;   // a byte per column; the last superblock requests 32 words ahead
    CREATE_TRY(hipMalloc(&s->a_ids, (size_t)n));
    CREATE_TRY(hipMalloc(&s->alpha, kAlphaBytes));
    CREATE_TRY(hipMalloc(&s->eqtab, sizeof(uint32_t) * (size_t)psa::kMaxSym * (size_t)s->tab_pitch));
#undef CREATE_TRY
    if (int rc = rebuild_tables(s, nullptr)) { psa_stripe_destroy(s); return rc; }
    *out = s;
    return PSA_OK;
}

int psa_stripe_set_sequences_dev(psa_stripe* s, const uint8_t* a_dev, const uint8_t* b_dev, void* stream) {
    if (!s || !a_dev || !b_dev) return fail(PSA_E_ARG, "psa_stripe_set_sequences_dev: null argument");
    if (int rc = enter_device(s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(s->a_dev, a_dev, (size_t)s->n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->b_dev, b_dev, (size_t)s->W, hipMemcpyDeviceToDevice, st));
    return rebuild_tables(s, st);
}

int psa_stripe_fill_rows(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev,
                         void* stream) {
    if (s && s->band_rows) return fail(PSA_E_STATE, "psa_stripe_fill_rows: a banded stripe is driven by psa_align_banded");
    return fill_rows_impl(s, r0, r1, left_dev, right_dev, stream, FillOpts());
}

}  // extern "C"

namespace {
int fill_rows_impl(psa_stripe* s, int64_t r0, int64_t r1, const int32_t* left_dev, int32_t* right_dev, void* stream,
                   const FillOpts& o) {
    if (!s) return fail(PSA_E_ARG, "psa_stripe_fill_rows: null stripe");
    if (int rc = enter_device(s)) return rc;
    if (r0 == 0) s->next_r0 = 0;  // a new pass over the same stripe
    if (o.any_order) s->next_r0 = r0;
    if (r0 != s->next_r0) return fail(PSA_E_STATE, "psa_stripe_fill_rows: chunks out of order (r0=%lld, expected %lld)",
                                      (long long)r0, (long long)s->next_r0);
    if ((r0 & 63) || r1 <= r0 || r1 > s->n)
        return fail(PSA_E_ARG, "psa_stripe_fill_rows: need r0 %% 64 == 0 and r0 < r1 <= n (r0=%lld r1=%lld n=%lld)",
                    (long long)r0, (long long)r1, (long long)s->n);
    if ((s->c0 == 0) != (left_dev == nullptr))
        return fail(PSA_E_ARG, "psa_stripe_fill_rows: left boundary must be given exactly when c0 > 0");
    hipStream_t st = (hipStream_t)stream;
    if (r0 == 0 || o.any_order) {
        // A launch that does not continue the previous one must not find that one's granules: a granule
        // is recognised by its tag (strip number), and a strip that finds a left-over with the right tag
        // runs ahead of the strips it should follow and overwrites ring slots they have not read yet.
        // (A banded refill takes its input from the checkpoint row, never from the ring.)
        HIP_TRY(hipMemsetAsync(s->ring, 0, sizeof(psa::u64) * (2 * (size_t)s->Wpad + 64), st));
        if (r0 == 0) s->corner_cur = 0;
        HIP_TRY(hipMemsetAsync(s->ctrl + kAbortWord, 0, kCtrlBytes - kCtrlLaunchBytes, st));
    }
    HIP_TRY(hipMemsetAsync(s->ctrl, 0, kCtrlLaunchBytes, st));
    psa::FillArgs fa;
    fa.a = s->a_dev; fa.b = s->b_dev; fa.n = s->n; fa.r0 = r0; fa.r1 = r1; fa.c0 = s->c0;
    fa.W = o.W > 0 ? o.W : s->W; fa.Q = o.W > 0 ? (o.W + 63 + 63) / 64 : s->Q; fa.Wpad = s->Wpad;
    const bool with_codes = o.codes < 0 ? s->codes != nullptr : o.codes != 0;
    fa.left = left_dev;
    fa.corner = s->small + 4 * s->corner_cur;
    fa.corner_next = s->small + 4 * (s->corner_cur ^ 1);
    fa.right = right_dev;
    fa.final_triple = s->small + 8;
    fa.ring = s->ring;
    fa.codes = with_codes ? (o.codes_base ? o.codes_base : s->codes) : nullptr;
    fa.leftcol = with_codes ? s->leftcol : nullptr;
    fa.ticket = s->ctrl;
    fa.abort_flag = s->ctrl + kAbortWord;
    fa.prof = nullptr;
    fa.ckpt = s->ckpt; fa.band_strips = s->band_rows / 64;
    const char* prof_path = getenv("PSA_PROFILE_STRIPS");
    psa::u64* prof_dev = nullptr;
    const unsigned nstrips_prof = (unsigned)((r1 - r0 + 63) / 64);
    if (prof_path && s->use_v2) {
        size_t prof_words = 4 * (size_t)nstrips_prof;
        HIP_TRY(hipMalloc(&prof_dev, sizeof(psa::u64) * prof_words));
        HIP_TRY(hipMemsetAsync(prof_dev, 0, sizeof(psa::u64) * prof_words, st));
        fa.prof = prof_dev;
    }
    const unsigned strips = (unsigned)((r1 - r0 + 63) / 64);
    HIP_TRY(hipEventRecord(s->ev[0], st));
    if (s->use_v2) {
        psa::FillArgs2 fb;
        fb.base = fa; fb.a_ids = s->a_ids; fb.eqtab = s->eqtab; fb.tab_pitch = s->tab_pitch;
        const unsigned wgs = (strips + psa::kNW - 1) / psa::kNW;
        if (with_codes) hipLaunchKernelGGL(psa::fill_strips_v2<true>, dim3(wgs), dim3(psa::kNW * 64), 0, st, fb);
        else hipLaunchKernelGGL(psa::fill_strips_v2<false>, dim3(wgs), dim3(psa::kNW * 64), 0, st, fb);
    } else if (with_codes) hipLaunchKernelGGL(psa::fill_strips_v1<true>, dim3(strips), dim3(64), 0, st, fa);
    else hipLaunchKernelGGL(psa::fill_strips_v1<false>, dim3(strips), dim3(64), 0, st, fa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev[1], st));
    s->fill_timed = true;
    if (prof_dev) {  // diagnostic only: dump per-strip start/end ticks and placement
        std::vector<psa::u64> h(4 * (size_t)nstrips_prof);
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
            for (unsigned k = 0; k < nstrips_prof; ++k)
                fprintf(f, "%u,%lld,%lld,%llu,%llu,%llu,%llu\n", k, (long long)(h[4 * k] - h[0]),
                        (long long)(h[4 * k + 1] - h[0]), h[4 * k + 2] >> 32, h[4 * k + 2] & 0xffffffffu,
                        h[4 * k + 3] & 0xffffffffu, h[4 * k + 3] >> 32);
            fclose(f);
        }
        hipFree(prof_dev);
        prof_dev = nullptr;
    }
    s->corner_cur ^= 1;
    s->next_r0 = r1;
    return PSA_OK;
}
}  // namespace

extern "C" {

static int check_abort(psa_stripe* s, hipStream_t st) {
    unsigned ctrl[kAbortWord + 1] = {0};
    HIP_TRY(hipMemcpyAsync(ctrl, s->ctrl, sizeof ctrl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
#ifdef PSA_FILL_STATS
    {
        unsigned long long stats[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(stats, s->ctrl + 48, sizeof stats, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (getenv("PSA_FILL_STATS") && ctrl[0])
            fprintf(stderr, "fillstats: groups %u wait_before_start_us %.1f wave0_work_us %.1f wg_resident_us %.1f (means per group)\n",
                    ctrl[0], stats[0] / 100.0 / ctrl[0], stats[1] / 100.0 / ctrl[0], stats[2] / 100.0 / ctrl[0]);
    }
#endif
    if (ctrl[kAbortWord])
        return fail(PSA_E_KERNEL, "fill kernel: a strip's wait for the strip above expired (workgroups started %u, "
                    "chain front at workgroup %u)", ctrl[0], ctrl[psa::kFrontWord]);
    return PSA_OK;
}

int psa_stripe_final(psa_stripe* s, void* stream, int32_t triple_out[3]) {
    if (!s || !triple_out) return fail(PSA_E_ARG, "psa_stripe_final: null argument");
    if (s->next_r0 != s->n) return fail(PSA_E_STATE, "psa_stripe_final: rows filled up to %lld of %lld",
                                        (long long)s->next_r0, (long long)s->n);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = enter_device(s)) return rc;
    if (int rc = check_abort(s, st)) return rc;
    HIP_TRY(hipMemcpyAsync(triple_out, s->small + 8, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));   // (on the stripe's own stream: the null stream would order it against other stripes' work)
    return PSA_OK;
}

}  // extern "C"

namespace {
struct TraceOpts {
    const uint8_t* codes_base = nullptr;    // banded mode: where strip 0's codes would lie
    int W = 0;                              // > 0: the codes were written by a fill of the first W columns
    int64_t row_min = 0;                    // the codes cover rows > row_min only
};
int traceback_impl(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai, int32_t* bj,
                   int64_t cap, int64_t* len_io, void* stream, const TraceOpts& o);
}  // namespace

extern "C" {

int psa_stripe_traceback(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai, int32_t* bj,
                         int64_t cap, int64_t* len_io, void* stream) {
    if (s && s->band_rows) return fail(PSA_E_STATE, "psa_stripe_traceback: a banded stripe is driven by psa_align_banded");
    if (s && s->next_r0 != s->n) return fail(PSA_E_STATE, "psa_stripe_traceback: fill is not complete");
    return traceback_impl(s, i_io, j_io, k_io, ai, bj, cap, len_io, stream, TraceOpts());
}

}  // extern "C"

namespace {
int traceback_impl(psa_stripe* s, int64_t* i_io, int64_t* j_io, int* k_io, int32_t* ai, int32_t* bj,
                   int64_t cap, int64_t* len_io, void* stream, const TraceOpts& o) {
    if (!s || !i_io || !j_io || !k_io || !ai || !bj || !len_io) return fail(PSA_E_ARG, "psa_stripe_traceback: null argument");
    if (!s->codes) return fail(PSA_E_STATE, "psa_stripe_traceback: stripe was created without PSA_F_TRACEBACK");
    if (*k_io < 1 || *k_io > 3 || *i_io < 0 || *i_io > s->n || *j_io < 0 || *j_io > s->c1 + 1)
        return fail(PSA_E_ARG, "psa_stripe_traceback: bad state (%lld,%lld,%d)", (long long)*i_io, (long long)*j_io, *k_io);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = enter_device(s)) return rc;
    if (int rc = check_abort(s, st)) return rc;
    int64_t state[12] = {*i_io, *j_io, *k_io, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(s->tb_state, state, sizeof state, hipMemcpyHostToDevice, st));
    psa::TraceArgs ta;
    ta.codes = o.codes_base ? o.codes_base : s->codes; ta.c0 = s->c0; ta.cap = s->tb_cap;
    ta.a = s->a_dev; ta.b = s->b_dev; ta.leftcol = s->leftcol;
    ta.toprow = (o.row_min > 0 && s->band_rows > 0) ? s->ckpt + (size_t)(o.row_min / s->band_rows) * (size_t)s->Wpad : nullptr;
    if (o.row_min > 0 && !ta.toprow) return fail(PSA_E_STATE, "traceback: a row limit needs the checkpoint row of a banded stripe");
    ta.W = o.W > 0 ? o.W : s->W; ta.Q = o.W > 0 ? (o.W + 63 + 63) / 64 : s->Q;
    ta.row_min = o.row_min;
    ta.out_ai = s->tb_ai; ta.out_bj = s->tb_bj; ta.state = s->tb_state;
    HIP_TRY(hipEventRecord(s->ev[2], st));
    {
        const char* force = getenv("PSA_TRACEBACK_KERNEL");
        // the fast kernel keeps the walk in 32-bit registers: coordinates and tile ids must fit
        const bool fits32 = s->n < 0x7fff0000LL && s->c0 + s->W < 0x7fff0000LL &&
                            ((s->n + 63) / 64) * (int64_t)ta.Q < 0x7fff0000LL;
        if ((force && !strcmp(force, "v1")) || !fits32) hipLaunchKernelGGL(psa::traceback_v1, dim3(1), dim3(64), 0, st, ta);
        else hipLaunchKernelGGL(psa::traceback_wave, dim3(1), dim3(64), 0, st, ta);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev[3], st));
    s->tb_timed = true;
    HIP_TRY(hipMemcpyAsync(state, s->tb_state, sizeof state, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (getenv("PSA_TRACE_STATS")) fprintf(stderr, "traceback: len %lld iterations %lld runs/tiles %lld misses %lld\n", (long long)state[4], (long long)(state[6] >> 32), (long long)(state[6] & 0xffffffff), (long long)state[7]);
    if (getenv("PSA_TRACE_STATS")) fprintf(stderr, "traceback: core cycles %lld, 100 MHz ticks %lld; sections %lld %lld %lld %lld\n", (long long)state[8], (long long)state[9], (long long)(state[10] >> 32), (long long)(state[10] & 0xffffffff), (long long)(state[11] >> 32), (long long)(state[11] & 0xffffffff));
    if (state[5] == 2) return fail(PSA_E_KERNEL, "traceback: winner codes failed the decode check (values that cannot belong to one cell)");
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

int psa_stripe_last_fill_ms(psa_stripe* s, float* ms_out) {
    if (!s || !ms_out || !s->fill_timed) return fail(PSA_E_STATE, "no fill has been timed");
    if (int rc = enter_device(s)) return rc;
    HIP_TRY(hipEventSynchronize(s->ev[1]));
    HIP_TRY(hipEventElapsedTime(ms_out, s->ev[0], s->ev[1]));
    return PSA_OK;
}

int psa_stripe_last_traceback_ms(psa_stripe* s, float* ms_out) {
    if (!s || !ms_out || !s->tb_timed) return fail(PSA_E_STATE, "no traceback has been timed");
    if (int rc = enter_device(s)) return rc;
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

struct StripeSet {
    std::vector<psa_stripe*> st;
    std::vector<int32_t*> edge;  // device boundary buffers between consecutive stripes
    ~StripeSet() {
        for (auto* s : st) psa_stripe_destroy(s);
        for (auto* e : edge) hipFree(e);
    }
};

int64_t chunk_rows_default() {
    if (const char* e = getenv("PSA_CHUNK_ROWS")) {
        long long v = atoll(e);
        if (v >= 64) return (v / 64) * 64;
    }
    return 8192;
}

// Column stripes exactly as simple.cpp:90-99 cuts them; zero-width stripes (p > m) own no cell.
int run_stripes(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p, uint32_t flags,
                StripeSet& set, int32_t triple[3]) {
    const int64_t block = m / p;
    for (int id = 0; id < p; ++id) {
        const int64_t c0 = (int64_t)id * block;
        const int64_t c1 = (id == p - 1) ? m : c0 + block;
        if (c1 == c0) continue;
        psa_stripe* s = nullptr;
        if (int rc = psa_stripe_create(&s, a, n, b, m, c0, c1, flags)) return rc;
        set.st.push_back(s);
    }
    const size_t ns = set.st.size();
    const int64_t chunk = ns > 1 ? std::min<int64_t>(chunk_rows_default(), ((n + 63) / 64) * 64) : n;
    for (size_t g = 0; g + 1 < ns; ++g) {
        int32_t* e = nullptr;
        HIP_TRY(hipMalloc(&e, sizeof(int32_t) * 3 * (size_t)chunk));
        set.edge.push_back(e);
    }
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t r1 = std::min(n, r0 + chunk);
        for (size_t g = 0; g < ns; ++g) {
            const int32_t* left = g ? set.edge[g - 1] : nullptr;
            int32_t* right = (g + 1 < ns) ? set.edge[g] : nullptr;
            if (int rc = psa_stripe_fill_rows(set.st[g], r0, r1, left, right, nullptr)) return rc;
        }
    }
    return psa_stripe_final(set.st.back(), nullptr, triple);
}

}  // namespace

extern "C" {

int psa_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m, int p, int32_t* score_out,
              int32_t triple_out[3]) {
    if (!score_out || n < 0 || m < 0 || (n && !a) || (m && !b)) return fail(PSA_E_ARG, "psa_score: bad argument");
    if (p < 1) return fail(PSA_E_ARG, "p must be >= 1 (the reference divides by p, simple.cpp:92)");
    if (int rc = need_device()) return rc;
    int32_t t[3];
    if (n == 0) { t[0] = -1; t[1] = 0; t[2] = (int32_t)-(2 + m); }       // loop of simple.cpp:122 never runs
    else if (m == 0) { t[0] = -1; t[1] = (int32_t)-(2 + n); t[2] = -1; } // only column 0 exists (simple.cpp:124-133)
    else {
        StripeSet set;
        if (int rc = run_stripes(a, n, b, m, p, 0, set, t)) return rc;
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
        // codes of the whole matrix (1 B per cell) against what the device has free: beyond 70 %
        // of it -- or when PSA_BAND_ROWS asks for it -- align in linear space
        size_t free_b = 0, total_b = 0;
        const double code_bytes = (double)n * (double)m;
        const bool forced = getenv("PSA_BAND_ROWS") != nullptr;
        if (forced || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && code_bytes > 0.7 * (double)free_b))
            return psa_align_banded(a, n, b, m, 0, score_out, ai_out, bj_out, len_out);
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
        StripeSet set;
        rc = run_stripes(a, n, b, m, p, PSA_F_TRACEBACK, set, t);
        if (rc == PSA_OK) {
            int32_t v;
            int64_t i = n, j = m;
            int k = pick3(t, &v);
            for (size_t g = set.st.size(); g-- > 0 && rc == PSA_OK;) {
                rc = psa_stripe_traceback(set.st[g], &i, &j, &k, ai, bj, cap, &len, nullptr);
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
    psa_stripe* s = nullptr;
    g_band_rows_request = band_rows;
    int rc = psa_stripe_create(&s, a, n, b, m, 0, m, PSA_F_TRACEBACK | PSA_F_BANDED);
    g_band_rows_request = 0;
    if (rc) return rc;
    StripeSet owner;
    owner.st.push_back(s);
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
        const int Qr = (Wr + 63 + 63) / 64;
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

int psa_score_batch_dev(const uint8_t* seqs_dev, const int64_t* offsets_dev, const int32_t* pairs_dev, int64_t npairs,
                        int max_len_b, int32_t* scores_dev, unsigned* counter_dev, void* stream) {
    if (npairs <= 0) return PSA_OK;
    if (max_len_b < 0 || max_len_b > psa::kBatchMaxLen)
        return fail(PSA_E_ARG, "psa_score_batch_dev: max_len_b must be in [0, %d]", psa::kBatchMaxLen);
    if (!seqs_dev || !offsets_dev || !pairs_dev || !scores_dev || !counter_dev)
        return fail(PSA_E_ARG, "psa_score_batch_dev: null argument");
    if (int rc = need_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(counter_dev, 0, 256, st));
    psa::BatchArgs ba;
    ba.seqs = seqs_dev; ba.offsets = offsets_dev; ba.pairs = pairs_dev; ba.npairs = npairs; ba.scores = scores_dev;
    ba.next_pair = counter_dev;
    ba.max_b = std::max(64, ((max_len_b + 63) / 64) * 64);
    const size_t lds_bytes = (size_t)psa::kBatchWaves * psa::batch_wave_bytes(ba.max_b);
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t want = (npairs + psa::kBatchWaves - 1) / psa::kBatchWaves;
    const unsigned grid = (unsigned)std::min<int64_t>(want, (int64_t)cus * 8);
    hipLaunchKernelGGL(psa::batch_score_kernel, dim3(grid), dim3(psa::kBatchWaves * 64), lds_bytes, st, ba);
    HIP_TRY(hipGetLastError());
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

void psa_synth_protein(uint64_t seed, int64_t n, uint8_t* out) {
    static const char alphabet[] = "ACDEFGHIKLMNPQRSTVWY";
    std::mt19937_64 g(seed);
    for (int64_t k = 0; k < n; ++k) out[k] = (uint8_t)alphabet[g() % 20];
}

}  // extern "C"
