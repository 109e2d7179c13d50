// psa_fill_v2.hip.h -- the production fill kernel for gfx950 (any byte alphabet: a table row per distinct residue of A).
//
// Same mathematics and the same outputs as fill_strips_v1 (psa_fill_v1.hip.h), organised for a short
// dependent instruction stream per anti-diagonal step: at 2^16 x 2^16 one pair is bound by the length of
// its dependency chain (N + M steps), a launch of many pairs by instruction issue.
//   * one workgroup = kNW wavefronts = kNW consecutive strips of 64 rows; strip w hands its last row to
//     strip w+1 through a ring in LDS (slot per column, flow-controlled by two LDS counters), only the last
//     strip of a workgroup publishes through HBM granules (as v1 does for every strip);
//   * substitution scores come from a per-symbol table of BYTES (6 or 8, the biased mismatch / match score;
//     four columns per word) built once per stripe (psa_fill_aux.hip.h): the step adds the byte to the
//     diagonal input with an SDWA source select (v_add_u32_sdwa ... src1_sel:BYTE_k) -- no extract, no
//     compare, no B residues travelling through the lanes; table words are requested 64 steps ahead;
//   * lane 0's inputs (row above the strip) are pre-loaded, 16 steps at a time, into the destination
//     registers of the DPP moves: a DPP wave_shr:1 leaves lane 0's destination untouched, so the per-step
//     code has no special case for the strip's first row;
//   * the winner code of a cell is ONE byte, (11 D + 4 H' + V') & 0xff (psa_device.hip.h): v_lshl_add_u32,
//     v_mad_u32_u24 and a v_alignbit_b32 that shifts it into the word being collected; four steps fill a
//     word, 64 steps four words per lane, which leave as 16-byte-per-lane coalesced stores (layout as v1);
//   * lane 63's row leaves as one 16-byte LDS write per two steps (step16);
//   * steps run in blocks of 16 inside superblocks of 64; blocks in which every lane is inside the stripe
//     take the unpredicated path compiled per strip role, the 64-step head takes a predicated one, and the
//     values at the stripe's last column are captured either in line (run_tail) or in a second pass off
//     the dependency chain (run_capture_pass).
#pragma once
#include "psa_fill_v1.hip.h"

namespace psa {

#ifndef PSA_WAVES_PER_EU
#define PSA_WAVES_PER_EU 1
#endif
#ifndef PSA_GDIST
#define PSA_GDIST 2
#endif
#ifndef PSA_GRAN_COLS
#define PSA_GRAN_COLS 32
#endif
// Columns of the row above that wave 0 of a workgroup takes over from HBM at a time in the steady
// state: 16 (every block, as head and tail do), 32 or 64.  Fetch, tag check, unpack, lane shift and
// the write into the wave's own ring cost ~27 instructions a time; per block that made the
// HBM-fed strip 8 % slower than the LDS-fed ones, and the slowest strip sets the pace of its chain.
constexpr int kGranCols = PSA_GRAN_COLS;
constexpr int kGDist = PSA_GDIST;  // granule requests run this many blocks ahead (1, 2 or 4)
constexpr int kRing = 256;     // LDS ring slots per strip boundary (8 bytes each)
constexpr int kBlk = 16;       // steps per block

// ---- per-workgroup LDS ----
struct __attribute__((aligned(128))) FillLds {
    int2 ring[kNW][kRing];       // ring[w]: input of wave w; slot s = (D of column s-65, V of column s-64)
    int2 stage[kNW][32];         // staging of a wave's last row on its way to HBM granules
    int dummy[kNW][320];         // where lanes other than 63 drop their publish writes (16 bytes a lane; see out_base)
    int dummy2[kNW][64];         // where lanes that own no counter drop their counter posts
    int prod[kNW];               // steps completed by the producer of ring[w]
    int cons[kNW];               // ring[w] slots below this number have been read
    int snap[kNW][6][64];        // a strip's state at the superblock that holds column W-1 (run_capture_pass)
};

// Winner code of one cell (layout: psa_device.hip.h, "winner codes"): the byte (11*D + 4*H' + V') & 0xff
// enters the word being collected from the top; four steps fill a word.
__device__ __forceinline__ void push_code(unsigned& acc, int D, int Hn, int Vn) {
    unsigned t;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(t) : "v"(Hn), "v"(Vn));
    asm("v_mad_u32_u24 %0, %1, 11, %2" : "=v"(t) : "v"(D), "v"(t));
    acc = __builtin_amdgcn_alignbit(t, acc, 8);
}

struct StripState {
    int D0, D1;   // D of this row after the last even / odd step
    int H;        // H' of this row (x2 of the next column)
    int V;        // V' of this row at its current column (x3 of the row below)
    unsigned acc;          // winner codes of the last four steps (push_code)
    int r1, r2, r3;  // x1, x2, x3 of this row at the stripe's last column (captured in the tail)
};

// How a block treats lanes that are outside the stripe:
//   kFast    every lane is inside (64 <= t0, t0 + 16 <= W - 1): no predication at all
//   kHead    t < 64 and W >= 64: lane l enters at step l, commits are predicated on lane <= t
//   kTail    lanes leave on the right: nothing they compute is read again, so no predication;
//            each lane captures its values at column W-1 (right boundary / final triple)
//   kGeneric narrow stripes: full predicate and capture
enum : int { kFast = 0, kHead = 1, kTail = 2, kGeneric = 3 };

// One anti-diagonal step.  k = step inside the block (compile-time), pre = lane 0's inputs for
// steps 2j, 2j+1 as (D of column-1, V of column) pairs, pa = where this lane drops the block's slots.
// a + byte `sel` of w in one instruction (SDWA source select): the substitution score needs no extract
__device__ __forceinline__ int add_byte(int a, unsigned w, int sel) {
    int r;
    switch (sel) {
        case 0: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(w)); break;
        case 1: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(w)); break;
        case 2: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(w)); break;
        default: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(w)); break;
    }
    return r;
}

template <bool CODES, int MODE>
__device__ __forceinline__ void step16(const int k, StripState& S, const int4& pre, const unsigned mword,
                                       int* pa, const int t, const int lane, const int W,
                                       const bool row_ok) {
    const int preD = (k & 1) ? pre.z : pre.x;
    const int preV = (k & 1) ? pre.w : pre.y;
    const int Dm2 = S.D0, Vm1 = S.V;   // (even k) committed D of step k-2, V of step k-1: gone after this step
    const int x1 = add_byte(shift_down1(preD, (k & 1) ? S.D1 : S.D0), mword, k & 3);
    const int x3 = shift_down1(preV, S.V);
    const int x2 = S.H;
    const int x2p = x2 + 2, x3p = x3 + 2;
    const int D = max3i(x1, x2, x3);
    const int Hn = max(D, x2p);
    const int Vn = max(D, x3p);
    if (CODES) push_code(S.acc, D, Hn, Vn);
    if (MODE == kTail || MODE == kGeneric) {
        if (lane == t - (W - 1)) { S.r1 = x1; S.r2 = x2; S.r3 = x3; }
    }
    bool act = true;
    if (MODE == kHead) act = lane <= t;
    if (MODE == kGeneric) act = row_ok && lane <= t && t - lane < W;
    if (act) {
        if (k & 1) S.D1 = D; else S.D0 = D;
        S.H = Hn;
        S.V = Vn;
    }
    // publish (committed values).  Slot s of the block = (D of step s-2, V of step s-1); slots k and k+1 leave together
    // after every even step as ONE 16-byte write.  An LDS instruction keeps the wavefront that issues it from issuing
    // anything else for 16 (4 bytes a lane), 20 (8) or 30 (16) cycles -- a VALU instruction for 5 -- so one write per two
    // steps instead of one per step is worth 5 cycles a step to a wavefront that has its SIMD to itself
    // (tools/lds_issue.hip, tools/step_mix.hip).  Slot 16 (and D of the last step, in slot 17) is written by the next block's first
    // step, or by run_block after the strip's last block.
    if ((k & 1) == 0) *(int4*)(pa + 2 * k) = make_int4(Dm2, Vm1, S.D1, S.V);
}

// LDS counters: the LDS executes one wavefront's instructions in order, so data written before
// a counter is visible before it and reads issued before a counter store are done before it.
// Only the COMPILER must keep the order; release/acquire atomics would also drain the vector
// memory counter (table prefetch, code stores, granules) once per block -- a microsecond each.
#define PSA_COMPILER_BARRIER() asm volatile("" ::: "memory")
__device__ __forceinline__ int lds_peek(int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_post(int* p, int v) {
    PSA_COMPILER_BARRIER();
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    PSA_COMPILER_BARRIER();
}

// Scope of the loads and stores that carry a stripe's boundary (values and "done" flags) to the stripe beside it.
// The neighbour may be filled by another launch on ANOTHER GPU of the same process (psa_align_multi: the buffers
// are fine-grained memory of the consumer's GPU, written over xGMI), so these few accesses per strip are
// system-scope: write-through stores, s_waitcnt vmcnt(0), then the flag; the consumer polls the flag and reads
// past its caches.  On one GPU the cost is nil (a dozen accesses per strip of 64 rows).
constexpr int kEdgeScope = __HIP_MEMORY_SCOPE_SYSTEM;

// Bounded waiting.  The bound is wall-clock time (s_memrealtime, 100 MHz), not an iteration count:
// with several kernels or processes sharing the GPU a wait can be long in iterations without
// anything being wrong.  Looked at once per 1024 polls, together with the abort flag (rarely:
// hundreds of waiting waves share that one address).
struct Spin {
    unsigned n = 0;
    unsigned long long t0 = 0;
};
__device__ __forceinline__ bool spin_expired(Spin& s, unsigned* abort_flag) {
    __builtin_amdgcn_s_sleep(1);
    if ((++s.n & 1023u) == 0) {
        const unsigned long long now = wall_clock64();
        if (s.t0 == 0) s.t0 = now;
        const unsigned ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ab || now - s.t0 > kSpinTicks) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

// Everything one strip carries from block to block.  Members marked (u) are wave-uniform and
// live in SGPRs (they derive from readfirstlane'd values and kernel arguments).
template <bool CODES>
struct StripCtx {
    StripState S;
    int lane;
    int wv;                 // (u) wave inside the workgroup
    int wg;                 // (u) workgroup's ticket = its place in the chain of this launch
    int W, nblocks;         // (u)
    bool from_lds, to_lds, to_hbm, top_analytic;  // (u)
    bool row_ok;
    int64_t sg, c0, i, R, y_row;                  // sg, c0, R (u)
    // substitution table window
    const uint32_t* trow;
    int msh, mword_idx;
    unsigned mw[17];        // table words covering this superblock's 64 columns (+ alignment spill)
    int prod_seen;          // (u) last value read from the producer's counter
    int cons_seen;          // (u) last value looked at of the consumer's counter (producer side)
    int cons_pending;       // the consumer's counter as requested one superblock ago
    // hand-off
    int2* my_ring;
    int* out_base;
    int out_wrap;
    int* prod_post;         // lane 63 of an LDS producer: the consumer's prod counter; else a dump word
    int* cons_post;         // lane 0 of an LDS consumer: its cons counter; else a dump word
    const u64* gin;
    u64* gout;
    u64* gdump;             // 64 granules behind the ring that nobody reads (one per lane)
    u64 tag_in, tag_out;
    int carryD;
    u64 gbatch;             // steady state, kGranCols > 16: the next kGranCols granules, one per lane
    u64 gq[kGDist];         // granules requested kGDist blocks ahead (wave 0 path), by block index mod kGDist
    unsigned gq_valid;      // (u) bit j: gq[j] holds a request
    unsigned qd[16];        // winner codes of the current superblock (see run_superblock)
    unsigned wait_in, wait_room;  // (u) spin iterations, for the profile only
    int snap_sb;            // (u) REPLAY kernels: the superblock at whose start the state is kept (-1: none)
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Where a strip's row-above comes from / where its last row goes.  The steady-state loop is
// compiled once per role so that a role's code holds only its own memory operations (a wait on
// the vector-memory counter placed for one role's loads would stall every other role too);
// kAny = decided at run time from the context flags (head and tail of a strip).
// kInReplay = the strip's second pass over its last superblocks (see fill_strips_body): lane 0's inputs are in the ring
// already, nothing is waited for, fetched or posted.
enum : int { kAny = -1, kInLds = 0, kInHbm = 1, kInTop = 2, kInReplay = 3, kOutLds = 0, kOutHbm = 1, kOutNone = 2 };

// One block of 16 steps: wait for lane 0's inputs, load them, run the steps, post progress.
template <bool CODES, bool DENSE, int MODE, int IN, int OUT, int JB>
__device__ __forceinline__ bool run_block(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, const int t0,
                                          const unsigned (&mq)[4]) {
    const int lane = c.lane;
    const int W = c.W;
    const bool from_lds = (IN == kAny) ? c.from_lds : (IN == kInLds);
    const bool top_analytic = (IN == kAny) ? c.top_analytic : (IN == kInTop);
    const bool to_hbm = (OUT == kAny) ? c.to_hbm : (OUT == kOutHbm);
    // ---- 1. lane 0's inputs for columns t0..t0+15: ring slots t0+64..t0+79 ----
    const int slot0 = (t0 + 64) & (kRing - 1);
    // (only lane 0 uses what it reads.  The other lanes read the 16 bytes that follow their neighbour's -- whatever lies
    // there, inside FillLds -- instead of the same 16: sixty-four lanes on one address cost the LDS twice the cycles)
    const int4* rp = (const int4*)(c.my_ring + slot0) + lane;
    int4 pre[8];
    bool pre_loaded = false;
    if (IN == kInReplay) {
        // (the ring holds this block's slots since the first pass)
    } else if (from_lds) {
        if (MODE == kFast || MODE == kHead || t0 < W) {   // (no run-time condition in the steady state)
            // slots t0+64..t0+79 are complete once the producer has finished step t0+78; it stops at nblocks*16
            const int need = min(t0 + 80, c.nblocks * kBlk);
            if (c.prod_seen < need) {
                // A strip runs right behind the one above, so it has to look at the progress counter in
                // nearly every block.  The look and the reads of the slots go out together: LDS performs
                // them in order, so if the counter (read first) says "complete", the slot values read
                // after it are the complete ones -- one LDS round trip per block instead of two.
                int pk = lds_peek(&lds.prod[c.wv]);
                PSA_COMPILER_BARRIER();
#pragma unroll
                for (int j = 0; j < 8; ++j) pre[j] = rp[j];
                asm volatile("" : "+v"(pk) : : "memory");   // (keeps the wait for the counter behind the issue of the reads)
                c.prod_seen = __builtin_amdgcn_readfirstlane(pk);
                pre_loaded = true;
                if (c.prod_seen < need) {
                    Spin spins;
                    while ((c.prod_seen = __builtin_amdgcn_readfirstlane(lds_peek(&lds.prod[c.wv]))) < need) {
                        if (spin_expired(spins, p.abort_flag)) return false;
                    }
                    c.wait_in += spins.n;
                    PSA_COMPILER_BARRIER();
#pragma unroll
                    for (int j = 0; j < 8; ++j) pre[j] = rp[j];
                }
            }
        }
    } else if (MODE == kFast && IN == kInHbm && kGranCols > kBlk) {
        // steady state, wave 0: kGranCols columns at a time, every lane one column (columns past the end
        // are clamped to the last one: a real granule, whose values nobody uses).  Requested one batch
        // ago; all loads unconditional, so the waits on the vector-memory counter stay counted ones.
        if (JB % (kGranCols / kBlk) == 0) {
            const int lcol = lane & (kGranCols - 1);
            u64 g = c.gbatch;
            if (!__all((g >> 32) == c.tag_in)) {
                Spin spins;
                do {
                    if (spin_expired(spins, p.abort_flag)) return false;
                    g = load_granule(c.gin + min(t0 + lcol, W - 1));
                } while (!__all((g >> 32) == c.tag_in));
                c.wait_in += spins.n;
            }
            const int gD = (int)((unsigned)g >> 2);
            const int gV = gD + (int)((unsigned)g & 3u);
            c.gbatch = load_granule(c.gin + min(t0 + kGranCols + lcol, W - 1));
            const int Dm1 = shift_down1(c.carryD, gD);
            c.carryD = __builtin_amdgcn_readlane(gD, kGranCols - 1);
            if (kGranCols == 64 || lane < kGranCols) c.my_ring[slot0 + lane] = make_int2(Dm1, gV);
        }
    } else if (MODE == kFast || MODE == kHead || t0 < W) {
        // (unconditional in the steady state, also in its blocks past the last column, where col_ok is
        // false for every lane: a request that may be skipped would spoil the counted waits)
        // wave 0: fetch (D, V) of the row above for 16 columns into lanes 0..15, write the ring itself
        int gD, gV;
        const int col = t0 + lane;
        const bool col_ok = lane < kBlk && col < W;
        if (top_analytic) {
            int t1, t2, t3;
            const int64_t j = c.c0 + 1 + col;
            boundary_triple(0, j, t1, t2, t3);
            gD = max3i(t1, t2, t3) + (int)(3 * j);
            gV = max3i(t1, t2, t3 + 2) + (int)(3 * j);
        } else {
            // granules of this block were requested one block ago (a request any further ahead only
            // makes the strip settle that much further behind the one above: it runs ahead until its
            // requests start to miss); if the strip above had not published them yet, poll until it has
            Spin spins;
            u64 g = c.gq[JB % kGDist];
            if (MODE == kFast && IN == kInHbm) {
                // steady state: the request is always there.  No load on the straight path, so the wait
                // for g is a counted one; a load that MAY have been issued just before would turn it
                // into a wait for everything in flight (table words, code stores).
                bool ok = !col_ok || (g >> 32) == c.tag_in;
                if (!__all(ok)) {
                    do {
                        if (spin_expired(spins, p.abort_flag)) return false;
                        g = col_ok ? load_granule(c.gin + col) : (c.tag_in << 32);
                        ok = !col_ok || (g >> 32) == c.tag_in;
                    } while (!__all(ok));
                }
            } else {
                bool have = (c.gq_valid >> (JB % kGDist)) & 1;
                for (;;) {
                    if (!have) g = col_ok ? load_granule(c.gin + col) : (c.tag_in << 32);
                    have = false;
                    const bool ok = !col_ok || (g >> 32) == c.tag_in;
                    if (__all(ok)) break;
                    if (spin_expired(spins, p.abort_flag)) return false;
                }
            }
            c.wait_in += spins.n;
            gD = (int)((unsigned)g >> 2);
            gV = gD + (int)((unsigned)g & 3u);
            // request the block kGDist blocks ahead
            if (MODE == kFast && IN == kInHbm) {
                // unconditional and unmasked (lanes 16..63 repeat lanes 0..15, columns past the end are
                // clamped and never looked at): a load that may be skipped would force every later wait
                // on the vector-memory counter to be a wait for zero
                c.gq[JB % kGDist] = load_granule(c.gin + min(t0 + kGDist * kBlk + (lane & (kBlk - 1)), W - 1));
                c.gq_valid |= 1u << (JB % kGDist);
            } else {
                const int ncol = col + kGDist * kBlk;
                const bool nvalid = (t0 + kGDist * kBlk < W);
                if (nvalid) c.gq[JB % kGDist] = (lane < kBlk && ncol < W) ? load_granule(c.gin + ncol) : (c.tag_in << 32);
                c.gq_valid = (c.gq_valid & ~(1u << (JB % kGDist))) | ((unsigned)nvalid << (JB % kGDist));
            }
        }
        // slot(col) = (D of col-1, V of col)
        const int Dm1 = shift_down1(c.carryD, gD);
        c.carryD = __builtin_amdgcn_readlane(gD, kBlk - 1);
        if (lane < kBlk) c.my_ring[slot0 + lane] = make_int2(Dm1, gV);
    }
    if (MODE == kHead && JB == 0 && IN != kInLds) {
        // first block of wave 0 (t0 = 0), its input has arrived: move the front (see the kernel's prologue)
        if (lane == 0) __hip_atomic_fetch_max(p.ticket + kFrontWord, (unsigned)c.wg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!pre_loaded) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = rp[j];
    }
    // One pair alone: once per superblock is enough -- the strip above looks once per superblock, a superblock ahead
    // of its need, and this strip trails by 80 of the ring's 256 slots; a post is an LDS instruction, 16 issue cycles
    // (7.03 -> 6.95 ms).  Independent pairs sharing the SIMDs: every block (the sparser post cost that launch 5 %).
    if ((IN == kAny || IN == kInLds) && (DENSE || JB == 3)) lds_post(c.cons_post, t0 + 80);

    int* pa = c.out_base + (c.out_wrap ? ((2 * t0) & (c.out_wrap - 1)) : 0);

    // ---- 2. the sixteen steps ----
    step16<CODES, MODE>(0, c.S, pre[0], mq[0], pa, t0, lane, W, c.row_ok);

    // A strip that publishes through HBM reads back what lane 63 dropped into the staging slots
    // during the PREVIOUS block and in the step just done (columns t0-80 .. t0-65: slots t0-16 .. t0, the
    // last of which left with step 0's write; the LDS runs a wavefront's instructions in order) and stores the
    // granules a few steps into this block: the LDS round trip hides behind those steps.
    constexpr int kFlushAt = 4;
    constexpr bool flush_always = (MODE == kFast && OUT == kOutHbm);   // no branch, no mask (see the request above)
    // (steady state: lanes 16..63 repeat lanes 0..15 -- same address, same data -- instead of being masked off)
    const int fcol = t0 - 80 + (flush_always ? (lane & (kBlk - 1)) : lane);
    const bool flush = to_hbm && t0 >= 80 && t0 - 80 < W;
    int fv = 0, fd = 0;
    PSA_COMPILER_BARRIER();
    if (flush_always || (flush && lane < kBlk && fcol < W)) {
        // staging slot of column c is c + 64 (numbered like the ring): V in .y of the slot, D in .x of the next
        fv = lds.stage[c.wv][(fcol + 64) & 31].y;
        fd = lds.stage[c.wv][(fcol + 65) & 31].x;
    }
    PSA_COMPILER_BARRIER();
#pragma unroll
    for (int k = 1; k < kFlushAt; ++k) {
        step16<CODES, MODE>(k, c.S, pre[k >> 1], mq[k >> 2], pa, t0 + k, lane, W, c.row_ok);
        if (CODES && (k & 3) == 3) c.qd[4 * JB + (k >> 2)] = c.S.acc;
    }
    if (OUT == kAny || OUT == kOutHbm) {
        const u64 gran = (c.tag_out << 32) | ((unsigned)fd << 2) | (unsigned)(fv - fd);
        if (flush_always) {
            // every lane stores; in the few blocks per strip whose columns would be negative or past
            // the end the stores go to dump slots.  Not more often than that: a few
            // lines written by every strip of the device at every block are a hot spot that slows
            // every other access to their memory channel.
            store_granule((fcol >= 0 && fcol < W) ? c.gout + fcol : c.gdump, gran);
            PSA_COMPILER_BARRIER();
        } else if (flush) {
            if (lane < kBlk && fcol < W) store_granule(c.gout + fcol, gran);
            PSA_COMPILER_BARRIER();
        }
    }
#pragma unroll
    for (int k = kFlushAt; k < kBlk; ++k) {
        step16<CODES, MODE>(k, c.S, pre[k >> 1], mq[k >> 2], pa, t0 + k, lane, W, c.row_ok);
        if (CODES && (k & 3) == 3) c.qd[4 * JB + (k >> 2)] = c.S.acc;   // four steps fill a word
    }
    // the strip's last block: nobody comes to write slot 16 and the last step's D (slot 17)
    if (t0 + kBlk == c.nblocks * kBlk) {
        int* pb = c.out_base + (c.out_wrap ? ((2 * (t0 + kBlk)) & (c.out_wrap - 1)) : 0);
        *(int4*)pb = make_int4(c.S.D0, c.S.V, c.S.D1, 0);
    }

    // ---- 3. publish progress ----
    if (OUT == kAny || OUT == kOutLds) lds_post(c.prod_post, t0 + kBlk);
    return true;
}

// Four blocks = 64 steps.  Everything that touches vector memory on a fixed schedule happens
// here, once per 64 steps, so that no wait on the vector-memory counter sits in the 16-step
// path: the substitution-table words of the NEXT superblock are requested at the start of
// this one, the winner bits of this one are stored at its end.
template <bool CODES, bool DENSE, int MODE, int IN, int OUT>
__device__ __forceinline__ bool run_superblock(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, const int sb) {
    const int t0 = sb * 64;
    // table words one superblock ahead (16 new words; word 16 of the current set is word 0 of the next)
    const u32x4_a4 n0 = *(const u32x4_a4*)(c.trow + c.mword_idx + 17);
    const u32x4_a4 n1 = *(const u32x4_a4*)(c.trow + c.mword_idx + 21);
    const u32x4_a4 n2 = *(const u32x4_a4*)(c.trow + c.mword_idx + 25);
    const u32x4_a4 n3 = *(const u32x4_a4*)(c.trow + c.mword_idx + 29);
    // room in the consumer's ring for the slots this superblock writes (up to t0 + 65)
    // The consumer trails by ~80 of the ring's 256 slots, so there is room practically always: the
    // counter is read one superblock ahead of its use (the value is in a register long before it is looked
    // at) -- read and used on the spot, the LDS round trip, taken right behind the table requests, cost
    // ~5 cycles of every step.
    if ((OUT == kAny) ? c.to_lds : (OUT == kOutLds)) {
        const int need = t0 + 66 - kRing;
        if (c.cons_seen < need) {
            c.cons_seen = __builtin_amdgcn_readfirstlane(c.cons_pending);
            if (c.cons_seen < need) {
                Spin spins;
                while ((c.cons_seen = __builtin_amdgcn_readfirstlane(lds_peek(&lds.cons[c.wv + 1]))) < need) {
                    if (spin_expired(spins, p.abort_flag)) return false;
                }
                c.wait_room += spins.n;
            }
        }
        c.cons_pending = lds_peek(&lds.cons[c.wv + 1]);
    }
#define PSA_BLOCK(j)                                                                                   \
    {                                                                                                  \
        const unsigned mq[4] = {__builtin_amdgcn_alignbit(c.mw[4 * j + 1], c.mw[4 * j], c.msh),        \
                                __builtin_amdgcn_alignbit(c.mw[4 * j + 2], c.mw[4 * j + 1], c.msh),    \
                                __builtin_amdgcn_alignbit(c.mw[4 * j + 3], c.mw[4 * j + 2], c.msh),    \
                                __builtin_amdgcn_alignbit(c.mw[4 * j + 4], c.mw[4 * j + 3], c.msh)};   \
        if (!run_block<CODES, DENSE, MODE, IN, OUT, j>(c, lds, p, t0 + j * kBlk, mq)) return false;           \
    }
    PSA_BLOCK(0)
    PSA_BLOCK(1)
    PSA_BLOCK(2)
    PSA_BLOCK(3)
#undef PSA_BLOCK
    if (CODES) {
        // (nontemporal: nobody reads the codes before the walk, and plain stores are slow to be acknowledged, which every
        // later wait on the vector-memory counter then pays for -- see psa_fill_pk.hip.h)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* dst = (v4u*)(p.codes + code_tile_offset(c.sg, p.Q, sb)) + c.lane;
        const v4u q0 = {c.qd[0], c.qd[1], c.qd[2], c.qd[3]}, q1 = {c.qd[4], c.qd[5], c.qd[6], c.qd[7]};
        const v4u q2 = {c.qd[8], c.qd[9], c.qd[10], c.qd[11]}, q3 = {c.qd[12], c.qd[13], c.qd[14], c.qd[15]};
        __builtin_nontemporal_store(q0, dst);
        __builtin_nontemporal_store(q1, dst + 64);
        __builtin_nontemporal_store(q2, dst + 128);
        __builtin_nontemporal_store(q3, dst + 192);
    }
    c.mw[0] = c.mw[16];
    c.mw[1] = n0.x; c.mw[2] = n0.y; c.mw[3] = n0.z; c.mw[4] = n0.w;
    c.mw[5] = n1.x; c.mw[6] = n1.y; c.mw[7] = n1.z; c.mw[8] = n1.w;
    c.mw[9] = n2.x; c.mw[10] = n2.y; c.mw[11] = n2.z; c.mw[12] = n2.w;
    c.mw[13] = n3.x; c.mw[14] = n3.y; c.mw[15] = n3.z; c.mw[16] = n3.w;
    c.mword_idx += 16;
    return true;
}

template <bool CODES, bool DENSE, bool SNAP, int IN, int OUT>
__device__ __forceinline__ bool run_steady(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, int lo, int hi) {
    // superblock 0 is the head (lanes entering), then the unpredicated steady state
    if (lo == 0 && hi > 0) {
        if (!run_superblock<CODES, DENSE, kHead, IN, OUT>(c, lds, p, 0)) return false;
        lo = 1;
    }
    if (IN == kInHbm && kGranCols > kBlk)   // first batch of the steady state
        c.gbatch = load_granule(c.gin + min(64 * lo + (c.lane & (kGranCols - 1)), c.W - 1));
    for (int sb = lo; sb < hi; ++sb) {
        if (SNAP && sb == c.snap_sb) {   // (kept in LDS, not in registers held across the loop)
            int (*sn)[64] = lds.snap[c.wv];
            sn[0][c.lane] = c.S.D0; sn[1][c.lane] = c.S.D1; sn[2][c.lane] = c.S.H; sn[3][c.lane] = c.S.V;
            sn[4][c.lane] = (int)c.S.acc; sn[5][c.lane] = c.mword_idx;
        }
        if (!run_superblock<CODES, DENSE, kFast, IN, OUT>(c, lds, p, sb)) return false;
    }
    if (IN == kInHbm && kGranCols > kBlk) c.gq_valid = 0;   // the tail fetches for itself
    return true;
}

template <bool CODES, bool DENSE, bool SNAP, int IN>
__device__ __forceinline__ bool run_steady_out(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, int lo, int hi) {
    if (c.to_lds) return run_steady<CODES, DENSE, SNAP, IN, kOutLds>(c, lds, p, lo, hi);
    if (c.to_hbm) return run_steady<CODES, DENSE, SNAP, IN, kOutHbm>(c, lds, p, lo, hi);
    return run_steady<CODES, DENSE, SNAP, IN, kOutNone>(c, lds, p, lo, hi);
}

// The capturing pass.  Lanes leave the stripe on the right one per step, and a strip that owes somebody its values at
// column W-1 (a right boundary; the last row's final triple) has to keep them as each lane passes: a compare and three
// selects on top of a step's 8 instructions.  A strip ends a fixed number of STEPS after the strip above (85 inside a
// workgroup, 142 across two), so done in line those steps cost every strip of such a stripe 1.7 us, 1 023 times over
// (a lone 16 384-column stripe: 4.11 ms without a right boundary, 5.81 ms with one; profiles/r2/strip_lags_by_boundary_role.txt).
// So the strip runs to its end in the steady-state code like any other, hands its bottom row on, and THEN goes over
// its last one or two superblocks a second time from the state it kept (five registers and the table index, in LDS): inputs
// still in the ring, nothing waited for, nothing handed on (lane 63 rewrites the slots it wrote with the same values),
// capturing.  The second pass is off the chain: it costs the strip's own wavefront 7 us, on a GPU with CUs to spare.
template <bool CODES, bool DENSE>
__device__ __forceinline__ bool run_capture_pass(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, int lo, int hi) {
    int (*sn)[64] = lds.snap[c.wv];
    c.S.D0 = sn[0][c.lane]; c.S.D1 = sn[1][c.lane]; c.S.H = sn[2][c.lane]; c.S.V = sn[3][c.lane];
    c.S.acc = (unsigned)sn[4][c.lane]; c.mword_idx = sn[5][c.lane];
#pragma unroll
    for (int j = 0; j < 17; ++j) c.mw[j] = c.trow[c.mword_idx + j];
    for (int sb = lo; sb < hi; ++sb)
        if (!run_superblock<CODES, DENSE, kTail, kInReplay, kOutNone>(c, lds, p, sb)) return false;
    return true;
}

// The capturing tail in line (kernels without the second pass: one pair per launch, independent pairs -- only the strip
// with the last row captures there), compiled per role for the roles nearly every strip has.
template <bool CODES, bool DENSE, int IN, int OUT>
__device__ __forceinline__ bool run_tail(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, int lo, int hi) {
    for (int sb = lo; sb < hi; ++sb)
        if (!run_superblock<CODES, DENSE, kTail, IN, OUT>(c, lds, p, sb)) return false;
    return true;
}
template <bool CODES, bool DENSE>
__device__ __forceinline__ bool run_tail_roles(StripCtx<CODES>& c, FillLds& lds, const FillArgs& p, int lo, int hi) {
    if (c.from_lds && c.to_lds) return run_tail<CODES, DENSE, kInLds, kOutLds>(c, lds, p, lo, hi);
    if (c.from_lds && c.to_hbm) return run_tail<CODES, DENSE, kInLds, kOutHbm>(c, lds, p, lo, hi);
    if (!c.from_lds && !c.top_analytic && c.to_lds) return run_tail<CODES, DENSE, kInHbm, kOutLds>(c, lds, p, lo, hi);
    return run_tail<CODES, DENSE, kAny, kAny>(c, lds, p, lo, hi);
}

// The body of the fill kernels (the entry points below differ in their occupancy attribute and in REPLAY).
template <bool CODES, bool DENSE, bool REPLAY>
__device__ __forceinline__ void fill_strips_body(const FillBatch& B) {
    __shared__ FillLds lds;
    __shared__ unsigned wg_ticket;
    // Dynamic LDS nobody uses (its size is chosen at the launch): with it a workgroup takes more than half
    // of a CU's 160 KB, so that no second fill workgroup can be placed beside it.  Every strip is a link
    // of one dependency chain: a wave that shares its SIMD's issue slots holds up the whole chain.
    extern __shared__ unsigned lds_keepout[];
    StripCtx<CODES> c;
    const int lane = c.lane = threadIdx.x & 63;
    const int wv = c.wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) wg_ticket = atomicAdd(B.ticket, 1u);
    if (threadIdx.x < kNW) { lds.prod[threadIdx.x] = 0; lds.cons[threadIdx.x] = 0; }
    __syncthreads();
    // ticket -> (job, group)
    const unsigned tk = __builtin_amdgcn_readfirstlane(wg_ticket);
    const unsigned og = B.order ? __builtin_amdgcn_readfirstlane(B.order[tk]) : tk;
    const unsigned group = og & ((1u << kOrderJobShift) - 1);
    const FillArgs2 q = B.job[og >> kOrderJobShift];   // (a copy: loaded once, kept in SGPRs)
    const FillArgs& p = q.base;
    if (p.W < 0) lds_keepout[threadIdx.x] = group;   // (never true; keeps the dynamic LDS referenced)
    const int64_t R = c.R = p.r1 - p.r0;
    const int64_t strips = (R + 63) >> 6;
    const int64_t s = (int64_t)group * kNW + wv;  // strip inside the chunk
    if (s >= strips) return;
#ifdef PSA_FILL_STATS   // experiment: time a workgroup sits on its CU before / while it works (control block words 48..)
    const unsigned long long st_entry = wall_clock64();
#endif
    const int64_t row_base = p.r0 + 64 * s;
    const int64_t sg = c.sg = row_base >> 6;
    const int64_t i = c.i = row_base + lane + 1;
    const bool row_ok = c.row_ok = i <= p.r1;
    const int W = c.W = p.W;
    const int64_t c0 = c.c0 = p.c0;
    const bool last_in_chunk = (s == strips - 1);
    c.to_lds = (wv + 1 < kNW) && !last_in_chunk;              // strip s+1 is wave wv+1 of this workgroup
    c.to_hbm = !c.to_lds && !(last_in_chunk && p.r1 == p.n);  // someone (maybe a later launch) reads it
    c.from_lds = wv > 0;
    c.top_analytic = (sg == 0);

    // ---- left boundary (as v1) ----
    int o1, o2, o3, u1, u2, u3;
    if (p.left) {
        const int64_t y = i - p.r0 - 1;
        const int64_t yo = row_ok ? y : (R - 1);
        if (p.left_done) {
            // the stripe on the left is filled by this same launch: wait until it has stored this strip's rows
            // (and the row above them), then read them past this CU's cache (agent scope), as it stored them
            Spin spins;
            for (;;) {
                const unsigned f1 = __hip_atomic_load(p.left_done + sg, __ATOMIC_RELAXED, kEdgeScope);
                const unsigned f0 = sg > 0 ? __hip_atomic_load(p.left_done + sg - 1, __ATOMIC_RELAXED, kEdgeScope) : 1u;
                if (__builtin_amdgcn_readfirstlane(f1 & f0)) break;
                if (spin_expired(spins, p.abort_flag)) return;
                __builtin_amdgcn_s_sleep(32);
            }
            o1 = __hip_atomic_load(p.left + yo, __ATOMIC_RELAXED, kEdgeScope);
            o2 = __hip_atomic_load(p.left + R + yo, __ATOMIC_RELAXED, kEdgeScope);
            o3 = __hip_atomic_load(p.left + 2 * R + yo, __ATOMIC_RELAXED, kEdgeScope);
            if (yo >= 1) {
                u1 = __hip_atomic_load(p.left + yo - 1, __ATOMIC_RELAXED, kEdgeScope);
                u2 = __hip_atomic_load(p.left + R + yo - 1, __ATOMIC_RELAXED, kEdgeScope);
                u3 = __hip_atomic_load(p.left + 2 * R + yo - 1, __ATOMIC_RELAXED, kEdgeScope);
            } else boundary_triple(0, c0, u1, u2, u3);   // (one launch covers all rows: r0 == 0)
        } else {
            o1 = p.left[yo]; o2 = p.left[R + yo]; o3 = p.left[2 * R + yo];
            if (yo >= 1) { u1 = p.left[yo - 1]; u2 = p.left[R + yo - 1]; u3 = p.left[2 * R + yo - 1]; }
            else if (p.r0 > 0) { if (!take_corner(p, u1, u2, u3)) u1 = u2 = u3 = 0; }   // (gave up: the abort flag is set)
            else boundary_triple(0, c0, u1, u2, u3);
        }
    } else {
        boundary_triple(i, 0, o1, o2, o3);
        boundary_triple(i - 1, 0, u1, u2, u3);
    }
    if (p.left && !p.left_done && s == 0 && lane == 0 && p.r1 < p.n) publish_corner(p, R);
    const int bias_own = (int)(3 * (i + c0));
    const int D_own = max3i(o1, o2, o3) + bias_own;
    const int D_up = max3i(u1, u2, u3) + bias_own - 3;
    c.S.D0 = c.S.D1 = D_own;
    c.S.H = max3i(o1, o2 + 2, o3) + bias_own;
    if (CODES && p.leftcol && row_ok) p.leftcol[i] = make_int2(D_own, c.S.H);
    c.S.V = 0;
    c.S.acc = 0;
    c.S.r1 = c.S.r2 = c.S.r3 = 0;
    c.y_row = i - p.r0 - 1;

    // ---- substitution table row of this lane; window of 16 columns starting at t0 - lane ----
    const int sym = row_ok ? (int)q.a_ids[i - 1] : 0;
    c.trow = q.eqtab + (size_t)sym * q.base.tab_pitch + kTabPad;
    c.msh = 8 * ((64 - lane) & 3);                      // (t0 - lane) mod 4 bytes, t0 % 16 == 0
    c.mword_idx = (-lane - ((64 - lane) & 3)) / 4;      // floor((0 - lane) / 4), exact division
#pragma unroll
    for (int j = 0; j < 17; ++j) c.mw[j] = c.trow[c.mword_idx + j];
    c.prod_seen = 0;
    c.cons_seen = 0;
    c.cons_pending = 0;

    // ---- hand-off plumbing ----
    c.my_ring = lds.ring[wv];
    if (lane == 63 && c.to_lds) { c.out_base = (int*)lds.ring[wv + 1]; c.out_wrap = kRing * 2; }
    else if (lane == 63 && c.to_hbm) { c.out_base = (int*)lds.stage[wv]; c.out_wrap = 32 * 2; }
    // (lane l's dump slots start 16 l bytes into the row: the lanes' 16-byte writes of a step are consecutive)
    else { c.out_base = lds.dummy[wv] + 4 * lane; c.out_wrap = 0; }
    c.prod_post = (lane == 63 && c.to_lds) ? &lds.prod[wv + 1] : &lds.dummy2[wv][lane];
    c.cons_post = (lane == 0 && c.from_lds) ? &lds.cons[wv] : &lds.dummy2[wv][lane];
    c.tag_in = (u64)sg;
    c.tag_out = (u64)(sg + 1);
    c.gin = p.ring + (size_t)(((sg - 1) / kNW) & 1) * p.Wpad;  // bottom row of strip sg-1
    c.gout = p.ring + (size_t)((sg / kNW) & 1) * p.Wpad;
    c.gdump = p.ring + 2 * (size_t)p.Wpad + lane;
    if (const u64* ci = ckpt_row_in(p, sg)) c.gin = ci;       // banded mode: band boundaries go through
    if (u64* co = ckpt_row_out(p, sg)) c.gout = co;           // the checkpoint rows, not the ring
    c.carryD = D_up;          // D of column t0-1 of the row above (wave 0 path)
    for (int j = 0; j < kGDist; ++j) c.gq[j] = 0;
    c.gbatch = 0;
    c.gq_valid = 0;
    c.wait_in = c.wait_room = 0;

    const int nblocks = c.nblocks = p.Q * (64 / kBlk);
    // A workgroup far down the chain waits for a long time before its first granule arrives.
    // Hundreds of them polling -- even slowly, if it is the same few cache lines -- slow the loads of
    // the strips that do run.  So a waiting workgroup first sleeps on the chain's "front": the number
    // of workgroups that have received their first input (kFrontWord of the control block, raised
    // by wave 0 of each workgroup in its very first block).  The sleep between two looks grows with
    // the distance to the front, so a workgroup looks a handful of times in all.  Once the
    // workgroup two above has begun, this one's input is some tens of microseconds away and it
    // goes on to poll its own granules finely.
    c.wg = (int)group;
    if (wv == 0 && c.wg >= 2 && W >= 128) {
        Spin spins;
        for (;;) {
            const unsigned f = __builtin_amdgcn_readfirstlane(
                __hip_atomic_load(p.ticket + kFrontWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int d = (int)c.wg - 1 - (int)f;   // workgroup wg-2 has begun once f >= (wg-2)+1
            if (d <= 0) break;
            const int naps = min(2 * d, 256);       // 3.4 us each; a workgroup step is 15-30 us
            for (int j = 0; j < naps; ++j) __builtin_amdgcn_s_sleep(127);
            const unsigned ab = __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long now = wall_clock64();
            if (spins.t0 == 0) spins.t0 = now;
            if (ab || now - spins.t0 > 4 * kSpinTicks) {
                __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
#ifdef PSA_FILL_STATS
    const unsigned long long st_start = wall_clock64();
#endif
    if (p.prof && lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.prof[4 * s + 0] = wall_clock64();
        p.prof[4 * s + 2] = ((u64)xcc << 32) | hwid;
    }

    // Wide stripes: head + steady state compiled per role over ALL superblocks (in the last ones the lanes leave
    // on the right: nothing they compute there is read again, so no predication).  Strips that owe their values at
    // column W-1 -- a right boundary was asked for, or this strip holds the last row -- keep the state at the
    // superblock that holds column W-1 and come back for it below (run_capture_pass).
    // Narrow stripes (W < 128): one generic, fully predicated loop that captures as it goes.
    const bool capture = W >= 128 && (p.right != nullptr || (row_base < p.n && p.n <= row_base + 64));
    const int capture_lo = (W - 1) / 64;
    c.snap_sb = (REPLAY && capture) ? capture_lo : -1;
    if (W >= 128) {
        // REPLAY: every strip goes to its end in the steady-state code.  Else: capturing strips (here: the one with the
        // last row, and any strip of a stripe with a right boundary that came to this kernel) leave it at the
        // superblock that holds column W-1 and capture in line.
        const int fast_hi = (capture && !REPLAY) ? capture_lo : p.Q;
        bool ok;
        if (c.from_lds) ok = run_steady_out<CODES, DENSE, REPLAY, kInLds>(c, lds, p, 0, fast_hi);
        else if (c.top_analytic) ok = run_steady_out<CODES, DENSE, REPLAY, kInTop>(c, lds, p, 0, fast_hi);
        else ok = run_steady_out<CODES, DENSE, REPLAY, kInHbm>(c, lds, p, 0, fast_hi);
        if (!ok) return;
        if (!REPLAY && !run_tail_roles<CODES, DENSE>(c, lds, p, fast_hi, p.Q)) return;
    } else {
        for (int sb = 0; sb < p.Q; ++sb)
            if (!run_superblock<CODES, DENSE, kGeneric, kAny, kAny>(c, lds, p, sb)) return;
    }
    if (!REPLAY) {   // (captured in line: out with it at once, as this kernel always did)
        if (row_ok) {
            const int bias = (int)(3 * (i + c0 + W));
            if (p.right && p.right_done) {   // read by the next stripe inside this launch: write through
                __hip_atomic_store(p.right + c.y_row, c.S.r1 - bias, __ATOMIC_RELAXED, kEdgeScope);
                __hip_atomic_store(p.right + R + c.y_row, c.S.r2 - bias, __ATOMIC_RELAXED, kEdgeScope);
                __hip_atomic_store(p.right + 2 * R + c.y_row, c.S.r3 - bias, __ATOMIC_RELAXED, kEdgeScope);
            } else if (p.right) {
                p.right[c.y_row] = c.S.r1 - bias; p.right[R + c.y_row] = c.S.r2 - bias; p.right[2 * R + c.y_row] = c.S.r3 - bias;
            }
            if (i == p.n) {
                p.final_triple[0] = c.S.r1 - bias; p.final_triple[1] = c.S.r2 - bias; p.final_triple[2] = c.S.r3 - bias;
            }
        }
        if (p.right_done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left before the flag does
            if (lane == 0) __hip_atomic_store(p.right_done + sg, 1u, __ATOMIC_RELAXED, kEdgeScope);
        }
    }
#ifdef PSA_FILL_STATS
    if (lane == 0) {
        const unsigned long long st_end = wall_clock64();
        unsigned long long* stats = (unsigned long long*)(p.ticket + 48);
        if (wv == 0) { atomicAdd(stats + 0, st_start - st_entry); atomicAdd(stats + 1, st_end - st_start); }
        if (wv == kNW - 1 || s == strips - 1) atomicAdd(stats + 2, st_end - st_entry);
    }
#endif
    if (p.prof && lane == 0) {   // (the end of what the strips below wait for)
        p.prof[4 * s + 1] = wall_clock64();
        p.prof[4 * s + 3] = ((u64)c.wait_room << 32) | c.wait_in;
    }
    // ---- trailing granule group (columns whose flush block lies beyond the last block) ----
    if (c.to_hbm) {
        for (int cb = nblocks * kBlk - 80; cb < W; cb += kBlk) {
            if (cb < 0) continue;
            PSA_COMPILER_BARRIER();
            const int col = cb + lane;
            if (lane < kBlk && col < W) {
                const int vs = lds.stage[wv][(col + 64) & 31].y;
                const int ds = lds.stage[wv][(col + 65) & 31].x;
                store_granule(c.gout + col, (c.tag_out << 32) | ((unsigned)ds << 2) | (unsigned)(vs - ds));
            }
        }
    }
    // ---- right boundary / final triple: second pass over the superblocks in which the lanes pass column W-1 ----
    if (REPLAY && capture && !run_capture_pass<CODES, DENSE>(c, lds, p, capture_lo, p.Q)) return;
    if (REPLAY) {
        if (row_ok) {
            const int bias = (int)(3 * (i + c0 + W));
            if (p.right && p.right_done) {   // read by the next stripe inside this launch: write through
                __hip_atomic_store(p.right + c.y_row, c.S.r1 - bias, __ATOMIC_RELAXED, kEdgeScope);
                __hip_atomic_store(p.right + R + c.y_row, c.S.r2 - bias, __ATOMIC_RELAXED, kEdgeScope);
                __hip_atomic_store(p.right + 2 * R + c.y_row, c.S.r3 - bias, __ATOMIC_RELAXED, kEdgeScope);
            } else if (p.right) {
                p.right[c.y_row] = c.S.r1 - bias; p.right[R + c.y_row] = c.S.r2 - bias; p.right[2 * R + c.y_row] = c.S.r3 - bias;
            }
            if (i == p.n) {
                p.final_triple[0] = c.S.r1 - bias; p.final_triple[1] = c.S.r2 - bias; p.final_triple[2] = c.S.r3 - bias;
            }
        }
        if (p.right_done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left before the flag does
            if (lane == 0) __hip_atomic_store(p.right_done + sg, 1u, __ATOMIC_RELAXED, kEdgeScope);
        }
    }
}

// One pair (or the stripes of one pair) per launch: one workgroup per CU, one wavefront per SIMD.  Every strip is
// a link of one dependency chain, so a second wavefront on a SIMD would hold up every strip behind the one it
// shares issue slots with.  The occupancy attribute makes the register budget 512 per wave; the launch's
// dynamic-LDS keep-out keeps a second workgroup off the CU.
template <bool CODES>
__global__ __launch_bounds__(kNW * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void fill_strips_v2(FillBatch B) {
    fill_strips_body<CODES, false, false>(B);
}
// The same for launches in which stripes have a right boundary: capturing strips take the second pass (run_capture_pass).
// A kernel of its own because the steady loop that looks for the superblock to keep its state at compiles to code
// that costs a lone pair 1 % (7.07 against 6.99 ms), which the launches without right boundaries need not pay.
template <bool CODES>
__global__ __launch_bounds__(kNW * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void fill_strips_v2_edges(FillBatch B) {
    fill_strips_body<CODES, false, true>(B);
}
// Independent pairs in one launch: the chains of different pairs do not wait for each other, and a lone
// wavefront uses less than half of its SIMD's issue slots (one instruction per 4.7 cycles, 12.5 of 21 of them
// VALU).  Two wavefronts per SIMD -- two workgroups per CU, 256 registers each -- interleave: the same code
// fills 8 pairs in 18.0 ms instead of 23.4 (the register cap costs spills in the run-time-role paths and 2 % on
// a lone pair, which is why the sparse kernel stays).
template <bool CODES>
__global__ __launch_bounds__(kNW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fill_strips_v2_dense(FillBatch B) {
    fill_strips_body<CODES, true, false>(B);
}

}  // namespace psa
