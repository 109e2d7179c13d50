// psa_fill_pk.hip.h -- the fill for launches of many independent pairs, TWO pairs per lane in packed 16-bit halves.
//
// Same decomposition as fill_strips_v2 (psa_fill_v2.hip.h: strips of 64 rows, lane = row, anti-diagonal wavefront with
// DPP wave_shr:1, four strips per workgroup handing their last row on through LDS rings, workgroups through HBM
// granules, tickets), same winner codes, same granules -- but a lane carries the cell of pair X in the low and the cell
// of pair Y in the high half of every register (the two pairs of a COUPLE have the same n and m), and the step is made
// of packed instructions: per step and couple
//     2 DPP moves (the 32-bit move shifts both halves), v_perm_b32 (the two substitution bytes side by side),
//     3 v_pk_add_u16, 4 v_pk_max_i16, 2 v_pk_mad_u16 (the two code bytes), 1 v_perm_b32 (code packing, 4 per 4 steps)
// = 13 instructions for two cells (+ 0.5 for aligning the table words) where the int32 kernel spends 11.4 on one.
//
// Frames.  A biased value grows by ~3 per row and column and does not fit 16 bits, but the values a wavefront holds at
// any one time lie within a few hundred of each other (64 neighbouring cells of an anti-diagonal).  So every strip
// keeps its registers relative to a FRAME, a pair of 32-bit bases (one per pair, multiples of 16 so that the low four
// bits of D -- the part of D that goes into the code byte -- are those of the absolute value), and moves the frame at
// superblock (64-step) boundaries by subtracting a wave-uniform packed delta from its four state registers.  Slots in
// an LDS ring are in the frame the producer had when it wrote them: everything strip s writes during its superblock
// k + 1 is what strip s + 1 reads during its superblock k (the ring's slot numbering lags 64 steps), so the consumer
// simply ADOPTS the producer's frame of superblock k + 1 at the start of its own superblock k (bases published per
// superblock beside the ring) and never converts a value.  The first strip of a workgroup picks its own frame (its lane
// 0's D) and converts at the HBM interface only: granules carry absolute values, 64 columns at a time, one per lane.
//
// Memory.  The winner codes leave per block of 16 steps as nontemporal 16-byte stores; the table words are loaded into
// the registers whose words are dead, two blocks before their first use; the steady loop exists once per input role
// (row above from LDS / from HBM granules) so that neither carries the other's vector-memory operations, spills nothing
// and waits on the vector-memory counter with counted waits only; the hand-over between strips is written by lane 63
// and read by lane 0 alone.  Why each of these: DESIGN 4.1b (the launch runs into the device's power management).
#pragma once
#include "psa_fill_v2.hip.h"

namespace psa {

constexpr int kPkStage = 128;   // staging slots of a strip that publishes through HBM (flushed 64 columns at a time)
constexpr int kPkFrames = 8;    // published frames per ring (a producer is at most three superblocks ahead)

struct __attribute__((aligned(128))) PkLds {
    int2 ring[kPkNW][kRing];          // ring[w]: input of wave w; slot s = (D of column s-65, V of column s-64), both pairs packed
    int2 stage[kPkNW][kPkStage];      // a wave's last row on its way to HBM granules (same slot numbering)
    int dummy[kPkNW][320];            // where lanes other than 63 drop their publish writes (PSA_PK_MASKED bit 0 off only)
    int dummy2[kPkNW][64];            // where lanes that own no counter drop their counter posts
    int prod[kPkNW];                  // steps completed by the producer of ring[w]
    int cons[kPkNW];                  // ring[w] slots below this number have been read
    int frame[kPkNW][kPkFrames][2];   // frame[w][k & 7]: bases (pair X, pair Y) of ring[w]'s producer during its superblock k
};

typedef short pkv __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pkv pk_v(int x) { return __builtin_bit_cast(pkv, x); }
__device__ __forceinline__ int pk_i(pkv x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk_add(int a, int b) { return pk_i(pk_v(a) + pk_v(b)); }
__device__ __forceinline__ int pk_sub(int a, int b) { return pk_i(pk_v(a) - pk_v(b)); }
__device__ __forceinline__ int pk_max(int a, int b) { return pk_i(__builtin_elementwise_max(pk_v(a), pk_v(b))); }
__device__ __forceinline__ int pk_pair(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int pk_lo(int x) { return (int)(short)(x & 0xffff); }   // sign-extended halves
__device__ __forceinline__ int pk_hi(int x) { return x >> 16; }
// a * b + c per half (v_pk_mad_u16); b is a wave-uniform packed constant
__device__ __forceinline__ int pk_mad(int a, int b, int c) {
    int r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

struct PkCtx {
    int lane, wv;
    int W, Q, nblocks;
    bool from_lds, to_lds, to_hbm, top;   // wave-uniform
    int baseX, baseY;                     // this strip's frame (wave-uniform)
    int prod_seen, cons_seen;
    int2* my_ring;
    int* out_base;
    int out_wrap;
    int* prod_post;
    int* cons_post;
    const u64* ginX; const u64* ginY;
    u64* goutX; u64* goutY;
    u64 tag_in, tag_out;
    u64 gbX, gbY;                         // granules of the next superblock's 64 columns, one column per lane
    int carryX, carryY;                   // absolute D of the row above at the column before the next batch (lane-uniform use)
    const uint32_t* trowX; const uint32_t* trowY;
    int msh, widx;                        // widx: word of the table row that holds this lane's column of step 0
    unsigned mwX[17], mwY[17];            // table words of this superblock's 64 columns (+ alignment spill)
    uint8_t* codesX; uint8_t* codesY;     // (wave-uniform) code tiles of this strip: block 0 of superblock 0
    unsigned code_lane;                   // this lane's 16 bytes inside a block of a tile
    int cap1, cap2, cap3;                 // x1, x2, x3 at column W-1 (relative), per lane, and the frame they were taken in
    int capBX, capBY;
    unsigned* abortX; unsigned* abortY;
    bool dead;                            // (wave-uniform) a wait expired: the abort flags are up, the strip runs on without waiting
};

// Bounded waiting without a vector-memory operation in the loop (spin_expired also looks at the abort flag: a load in a
// loop beside the straight path makes the compiler wait for EVERYTHING in flight -- the block's code stores -- where the
// paths join again; seen here as an s_waitcnt vmcnt(0) per block).  Wall clock only; every waiter gives up by itself.
// No early exits either: a strip whose wait expired runs to its end on whatever it finds (exit edges out of the unrolled
// steady loop cost the same full waits where they join).
__device__ __forceinline__ bool pk_expired(Spin& s, PkCtx& c) {
    __builtin_amdgcn_s_sleep(1);
    if ((++s.n & 1023u) == 0) {
        const unsigned long long now = wall_clock64();
        if (s.t0 == 0) s.t0 = now;
        if (now - s.t0 > kSpinTicks) {   // (the abort words are raised at the strip's end: no store on a path that rejoins the steady loop)
            c.dead = true;
            return true;
        }
    }
    return false;
}

enum : int { kPkFast = 0, kPkHead = 1, kPkTail = 2 };
// Where a strip's row above comes from: the steady loop is compiled once for the strips fed through LDS and once for the
// first strip of a workgroup (granules from HBM), head and tail decide at run time (kPkInAny).  In one loop for both,
// the registers that hold the next batch of granules are carried around the loop, and on the path of the strips that
// never load them the compiler still has to assume a load in flight: it put s_waitcnt vmcnt(0) there, at the start of
// every superblock of EVERY strip, right behind the requests for the next superblock's table words -- a memory round
// trip and the completion of the superblock's code stores waited out 1 025 times per strip.
enum : int { kPkInAny = 0, kPkInLds = 1, kPkInHbm = 2 };
template <int IN>
__device__ __forceinline__ bool pk_from_lds(const bool dyn) { return IN == kPkInAny ? dyn : (IN == kPkInLds); }

// Experiment switch (tools/x_sweep.sh): 1 = every hand-over is taken to have arrived -- no strip waits for the strip above,
// for room in a ring or for granules.  The results are wrong; the launch's duration is what the instruction stream
// costs without the dependency chain.
#ifndef PSA_PK_NOSYNC
#define PSA_PK_NOSYNC 0
#endif
#if PSA_PK_NOSYNC
#warning "PSA_PK_NOSYNC: timing-only build of fill_strips_pk, its results are WRONG"
#endif
#ifndef PSA_PK_X   // further timing-only switches: 1 no publish writes, 2 no ring reads, 4 no code stores, 8 no table loads (the first words stay)
#define PSA_PK_X 0
#endif
#if PSA_PK_X
#warning "PSA_PK_X: timing-only build of fill_strips_pk, its results are WRONG"
#endif

// Lane 63's 16 bytes into the next strip's ring.  PSA_PK_MASKED bit 0: under an execution mask of lane 63 alone (two
// scalar moves around the instruction) instead of every lane writing its 16 bytes somewhere.  The LDS takes as long for
// the instruction either way (tools/lds_tp.hip: 14 cycles), but the launch is held back by the device's power
// management, not by issue slots -- the same instruction stream ran at 2.15 GHz without the code stores and at 1.77 GHz
// with them (SQ_BUSY_CYCLES over the launch's duration, timing-only builds) -- and 63 lanes' worth of data that does not
// move is power: 3 630 -> 3 690-3 750 GCUPS.  Bit 1, the same for the ring reads, lane 0 alone, as a divergent region
// of the compiler's own making (3 770; with the mask set by hand inside asm the compiler does not see the LDS
// operations and the hand-placed waits cost what the lanes saved: 3 690).
#ifndef PSA_PK_MASKED
#define PSA_PK_MASKED 3
#endif
template <int OFF>
__device__ __forceinline__ void pk_publish(int* pa, const int4 v) {
#if PSA_PK_MASKED
    // only lane 63 writes (the execution mask around the instruction: two scalar moves; the LDS takes as long, but 63
    // lanes' worth of data does not move)
    typedef int v4i __attribute__((ext_vector_type(4)));
    const unsigned a = (unsigned)(size_t)pa;
    const v4i w = {v.x, v.y, v.z, v.w};
    asm volatile("s_mov_b64 exec, %2\n\tds_write_b128 %0, %1 offset:%3\n\ts_mov_b64 exec, -1" ::"v"(a), "v"(w), "s"(1ull << 63), "i"(OFF) : "memory");
#else
    *(int4*)(pa + OFF / 4) = v;
#endif
}

// One block of 16 steps.  D0/D1/H/V: the strip's state (named registers: see psa_batch.hip.h on why not a struct).
template <bool CODES, int MODE, int IN>
__device__ __forceinline__ void pk_block(PkCtx& c, PkLds& lds, int& D0, int& D1, int& H, int& V, const int t0, const int jb, const int sb) {
    const int lane = c.lane;
    const bool from_lds = pk_from_lds<IN>(c.from_lds);
    // ---- lane 0's inputs for columns t0..t0+15: ring slots t0+64..t0+79 ----
    const int slot0 = (t0 + 64) & (kRing - 1);
    const int4* rp = (const int4*)(c.my_ring + slot0) + lane;   // (only lane 0 uses what it reads, see fill_strips_v2)
    if (from_lds) {
        const int need = PSA_PK_NOSYNC ? 0 : min(t0 + 80, c.nblocks * kBlk);
        if (c.prod_seen < need) {
            Spin spins;
            while (!c.dead && (c.prod_seen = __builtin_amdgcn_readfirstlane(lds_peek(&lds.prod[c.wv]))) < need) pk_expired(spins, c);
        }
    }
    PSA_COMPILER_BARRIER();
    int4 pre[8];
#if PSA_PK_MASKED & 2
    // only lane 0 reads (a divergent region the compiler builds itself, so that its waits stay counted ones; the other
    // lanes' registers are never looked at: the DPP moves overwrite them)
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = rp[j];
    }
#define PK_PRE_READY(j)
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = (PSA_PK_X & 2) ? make_int4(D0, V, D1, H) : rp[j];
#define PK_PRE_READY(j)
#endif
    if (from_lds) lds_post(c.cons_post, t0 + 80);
    int* pa = c.out_base + (c.out_wrap ? ((2 * t0) & (c.out_wrap - 1)) : 0);
    // table words of the block, aligned to this lane's column phase
    unsigned ax[4], ay[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ax[q] = __builtin_amdgcn_alignbit(c.mwX[4 * jb + q + 1], c.mwX[4 * jb + q], c.msh);
        ay[q] = __builtin_amdgcn_alignbit(c.mwY[4 * jb + q + 1], c.mwY[4 * jb + q], c.msh);
    }
    // Table words, into the registers whose words are dead (no second set, no copies): words 9..16 of THIS superblock at
    // its block 0, words 1..8 of the NEXT one at block 2 -- each two blocks before its first use.  (A whole superblock
    // ahead into a second set of 32 registers, as fill_strips_v2 does, did not fit beside two pairs' state: spills.)
    if (!(PSA_PK_X & 8) && (jb == 0 || jb == 2)) {
        const int at = c.widx + (jb == 0 ? 9 : 17), to = (jb == 0 ? 9 : 1);
        const u32x4_a4 x0 = *(const u32x4_a4*)(c.trowX + at), x1 = *(const u32x4_a4*)(c.trowX + at + 4);
        const u32x4_a4 y0 = *(const u32x4_a4*)(c.trowY + at), y1 = *(const u32x4_a4*)(c.trowY + at + 4);
        c.mwX[to] = x0.x; c.mwX[to + 1] = x0.y; c.mwX[to + 2] = x0.z; c.mwX[to + 3] = x0.w;
        c.mwX[to + 4] = x1.x; c.mwX[to + 5] = x1.y; c.mwX[to + 6] = x1.z; c.mwX[to + 7] = x1.w;
        c.mwY[to] = y0.x; c.mwY[to + 1] = y0.y; c.mwY[to + 2] = y0.z; c.mwY[to + 3] = y0.w;
        c.mwY[to + 4] = y1.x; c.mwY[to + 5] = y1.y; c.mwY[to + 6] = y1.z; c.mwY[to + 7] = y1.w;
    }
    const int two = 0x00020002, four = 0x00040004, eleven = 0x000b000b;
    int cd[4];
    unsigned qx[4], qy[4];   // winner codes of the block: four steps a word
#define PSA_PK_STEP(k, DREG)                                                                                         \
    {                                                                                                                \
        const int t = t0 + (k);                                                                                      \
        if (((k) & 1) == 0) { PK_PRE_READY((k) >> 1); }                                                              \
        const int preD = ((k) & 1) ? pre[(k) >> 1].z : pre[(k) >> 1].x;                                              \
        const int preV = ((k) & 1) ? pre[(k) >> 1].w : pre[(k) >> 1].y;                                              \
        const int Dm2 = D0, Vm1 = V;                                                                                 \
        /* byte k&3 of the pair's table words side by side: (score X, score Y) as 16-bit halves */                   \
        const int sc = (int)__builtin_amdgcn_perm(ay[(k) >> 2], ax[(k) >> 2], 0x0c040c00u + 0x00010001u * ((k) & 3)); \
        const int x1 = pk_add(shift_down1(preD, DREG), sc);                                                          \
        const int x3 = shift_down1(preV, V);                                                                         \
        const int x2 = H;                                                                                            \
        const int M = pk_max(x1, x2);                                                                                \
        const int D = pk_max(M, x3);                                                                                 \
        const int Hn = pk_max(D, pk_add(x2, two));                                                                   \
        const int Vn = pk_max(D, pk_add(x3, two));                                                                   \
        if (CODES) cd[(k) & 3] = pk_mad(D, eleven, pk_mad(Hn, four, Vn));                                            \
        if (MODE == kPkTail) {                                                                                       \
            if (lane == t - (c.W - 1)) { c.cap1 = x1; c.cap2 = x2; c.cap3 = x3; }                                    \
        }                                                                                                            \
        if (MODE != kPkHead || lane <= t) { DREG = D; H = Hn; V = Vn; }                                              \
        if (!(PSA_PK_X & 1) && ((k) & 1) == 0) pk_publish<8 * (k)>(pa, make_int4(Dm2, Vm1, D1, V));                  \
        if (CODES && ((k) & 3) == 3) {                                                                               \
            const unsigned p01 = __builtin_amdgcn_perm((unsigned)cd[1], (unsigned)cd[0], 0x06020400u);               \
            const unsigned p23 = __builtin_amdgcn_perm((unsigned)cd[3], (unsigned)cd[2], 0x06020400u);               \
            qx[(k) >> 2] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);                                             \
            qy[(k) >> 2] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);                                             \
        }                                                                                                            \
    }
    PSA_PK_STEP(0, D0) PSA_PK_STEP(1, D1) PSA_PK_STEP(2, D0) PSA_PK_STEP(3, D1)
    PSA_PK_STEP(4, D0) PSA_PK_STEP(5, D1) PSA_PK_STEP(6, D0) PSA_PK_STEP(7, D1)
    PSA_PK_STEP(8, D0) PSA_PK_STEP(9, D1) PSA_PK_STEP(10, D0) PSA_PK_STEP(11, D1)
    PSA_PK_STEP(12, D0) PSA_PK_STEP(13, D1) PSA_PK_STEP(14, D0) PSA_PK_STEP(15, D1)
#undef PSA_PK_STEP
#undef PK_PRE_READY
    // The block's codes leave here, 16 bytes a lane and pair (a KB per instruction), not at the superblock's end: sixteen
    // registers a pair held across the superblock made the steady loop spill, and a spill's reload is a vector-memory
    // operation whose wait also waits for every code store in flight.  (Uniform base + 32-bit lane offset: one
    // address register.)
    if (CODES && (PSA_PK_X & 4)) asm volatile("" ::"v"(qx[0]), "v"(qx[1]), "v"(qx[2]), "v"(qx[3]), "v"(qy[0]), "v"(qy[1]), "v"(qy[2]), "v"(qy[3]));
    if (CODES && !(PSA_PK_X & 4)) {
        const size_t at = (size_t)sb * kTileBytes + (size_t)(1024 * jb);
        // Nontemporal: 86 GB of codes per launch of 20 pairs that nobody reads before the walk.  As plain stores they
        // took the launch from 19 to 28-31 ms (timing-only builds without them, tools/x_sweep.sh): every later wait on the
        // vector-memory counter -- the table words -- is also a wait for the stores before it, and those were slow to
        // be acknowledged; streaming past the L2 they are acknowledged early (24 ms; the bench 3 200 -> 3 580 GCUPS).
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u vx = {qx[0], qx[1], qx[2], qx[3]}, vy = {qy[0], qy[1], qy[2], qy[3]};
        __builtin_nontemporal_store(vx, (v4u*)(c.codesX + at + c.code_lane));
        __builtin_nontemporal_store(vy, (v4u*)(c.codesY + at + c.code_lane));
    }
    // the strip's last block: nobody comes to write slot 16 and the last step's D (slot 17)
    if (t0 + kBlk == c.nblocks * kBlk) {
        int* pb = c.out_base + (c.out_wrap ? ((2 * (t0 + kBlk)) & (c.out_wrap - 1)) : 0);
        *(int4*)pb = make_int4(D0, V, D1, 0);
    }
    lds_post(c.prod_post, t0 + kBlk);
}

// absolute (D, V) of 64 columns of the row above, one column per lane -> relative, packed, into this wave's own ring
__device__ __forceinline__ void pk_feed_ring(PkCtx& c, const int t0, const int dX, const int vX, const int dY, const int vY) {
    // slot of column col = (D of col-1, V of col)
    const int dmX = shift_down1(c.carryX, dX), dmY = shift_down1(c.carryY, dY);
    c.carryX = __builtin_amdgcn_readlane(dX, 63);
    c.carryY = __builtin_amdgcn_readlane(dY, 63);
    c.my_ring[(t0 + 64 + c.lane) & (kRing - 1)] = make_int2(pk_pair(dmX - c.baseX, dmY - c.baseY), pk_pair(vX - c.baseX, vY - c.baseY));
}

// HBM-out strips: column block cb (64 columns, one per lane) from the staging slots to granules, in the CURRENT frame.
// last_from_reg: the D of the block's last column is not in a slot yet (it leaves with the next step's write) but in
// lane 63's D0.
__device__ __forceinline__ void pk_flush(PkCtx& c, PkLds& lds, const int cb, const int D0, const bool last_from_reg) {
    const int col = 64 * cb + c.lane;
    PSA_COMPILER_BARRIER();
    const int v = lds.stage[c.wv][(col + 64) & (kPkStage - 1)].y;
    int d = lds.stage[c.wv][(col + 65) & (kPkStage - 1)].x;
    PSA_COMPILER_BARRIER();
    const int dlast = __builtin_amdgcn_readlane(D0, 63);
    if (last_from_reg && c.lane == 63) d = dlast;
    if (col < c.W) {
        const int dx = pk_lo(d) + c.baseX, dy = pk_hi(d) + c.baseY;
        const unsigned ex = (unsigned)(pk_lo(v) - pk_lo(d)) & 3u, ey = (unsigned)(pk_hi(v) - pk_hi(d)) & 3u;
        store_granule(c.goutX + col, (c.tag_out << 32) | ((unsigned)dx << 2) | ex);
        store_granule(c.goutY + col, (c.tag_out << 32) | ((unsigned)dy << 2) | ey);
    }
}

template <bool CODES, int MODE, int IN>
__device__ __forceinline__ void pk_superblock(PkCtx& c, PkLds& lds, const FillArgs& px, const FillArgs& py, int& D0, int& D1, int& H,
                                              int& V, const int sb, const int64_t sg) {
    const int t0 = sb * 64;
    const int lane = c.lane;
    const bool from_lds = pk_from_lds<IN>(c.from_lds);
    // ---- HBM out: what this strip wrote during the superblock before, still in the frame it was written in ----
    if (c.to_hbm && sb >= 2) pk_flush(c, lds, sb - 2, D0, true);
    // ---- the frame of this superblock ----
    int nbX, nbY;
    if (from_lds) {
        // (the producer is past its step t0 + 78, hence inside its superblock sb + 1 or beyond: that frame is published)
        const int need = PSA_PK_NOSYNC ? 0 : min(t0 + 80, c.nblocks * kBlk);
        if (c.prod_seen < need) {
            Spin spins;
            while (!c.dead && (c.prod_seen = __builtin_amdgcn_readfirstlane(lds_peek(&lds.prod[c.wv]))) < need) pk_expired(spins, c);
        }
        PSA_COMPILER_BARRIER();
        nbX = __builtin_amdgcn_readfirstlane(lds_peek(&lds.frame[c.wv][(sb + 1) & (kPkFrames - 1)][0]));
        nbY = __builtin_amdgcn_readfirstlane(lds_peek(&lds.frame[c.wv][(sb + 1) & (kPkFrames - 1)][1]));
    } else {
        // first strip of a workgroup: lane 0's D becomes (nearly) zero; multiples of 16 only
        const int d = __builtin_amdgcn_readfirstlane(D0);
        nbX = c.baseX + (pk_lo(d) & ~15);
        nbY = c.baseY + (pk_hi(d) & ~15);
    }
    {
        const int delta = pk_pair(nbX - c.baseX, nbY - c.baseY);
        D0 = pk_sub(D0, delta); D1 = pk_sub(D1, delta); H = pk_sub(H, delta); V = pk_sub(V, delta);
        c.baseX = nbX; c.baseY = nbY;
    }
    if (c.to_lds) {
        // room in the consumer's ring for the slots this superblock writes (up to t0 + 65)
        const int need = PSA_PK_NOSYNC ? -(1 << 30) : t0 + 66 - kRing;
        if (c.cons_seen < need) {
            Spin spins;
            while (!c.dead && (c.cons_seen = __builtin_amdgcn_readfirstlane(lds_peek(&lds.cons[c.wv + 1]))) < need) pk_expired(spins, c);
        }
        // the frame goes out before anything written in it (the LDS takes a wavefront's instructions in order)
        // (and, in the last superblock, the frame of the slots written after it: there is no rebase behind it)
        lds_post(&lds.frame[c.wv + 1][sb & (kPkFrames - 1)][0], nbX);
        lds_post(&lds.frame[c.wv + 1][sb & (kPkFrames - 1)][1], nbY);
        if (sb == c.Q - 1) {
            lds_post(&lds.frame[c.wv + 1][(sb + 1) & (kPkFrames - 1)][0], nbX);
            lds_post(&lds.frame[c.wv + 1][(sb + 1) & (kPkFrames - 1)][1], nbY);
        }
    }
    // ---- first strip of a workgroup: the row above, 64 columns, from HBM granules (or the matrix's row 0) ----
    if (!from_lds) {
        int dX, vX, dY, vY;
        const int col = t0 + lane;
        if (c.top) {
            int t1, t2, t3;
            const int64_t j = 1 + col;
            boundary_triple(0, j, t1, t2, t3);
            dX = dY = max3i(t1, t2, t3) + (int)(3 * j);
            vX = vY = max3i(t1, t2, t3 + 2) + (int)(3 * j);
        } else {
            u64 gx = c.gbX, gy = c.gbY;
            const int cc = min(col, c.W - 1);
            if (!PSA_PK_NOSYNC && !__all((gx >> 32) == c.tag_in && (gy >> 32) == c.tag_in)) {
                Spin spins;
                do {
                    pk_expired(spins, c);
                    gx = load_granule(c.ginX + cc);
                    gy = load_granule(c.ginY + cc);
                } while (!__all((gx >> 32) == c.tag_in && (gy >> 32) == c.tag_in) && !c.dead);   // (the tags first: no load in flight on any way out)
            }
            const int cn = min(col + 64, c.W - 1);
            if (!(PSA_PK_X & 32)) {   // (32: timing only, the first strip of a workgroup requests no granules)
                c.gbX = load_granule(c.ginX + cn);
                c.gbY = load_granule(c.ginY + cn);
            }
            dX = (int)((unsigned)gx >> 2); vX = dX + (int)((unsigned)gx & 3u);
            dY = (int)((unsigned)gy >> 2); vY = dY + (int)((unsigned)gy & 3u);
        }
        pk_feed_ring(c, t0, dX, vX, dY, vY);
        if (sb == 0 && lane == 0)   // its input has arrived: move the front (see fill_strips_v2's prologue)
            __hip_atomic_fetch_max(px.ticket + kFrontWord, (unsigned)(sg / kPkNW) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pk_block<CODES, MODE, IN>(c, lds, D0, D1, H, V, t0, 0, sb);
    pk_block<CODES, MODE, IN>(c, lds, D0, D1, H, V, t0 + 16, 1, sb);
    pk_block<CODES, MODE, IN>(c, lds, D0, D1, H, V, t0 + 32, 2, sb);
    pk_block<CODES, MODE, IN>(c, lds, D0, D1, H, V, t0 + 48, 3, sb);
    if (MODE == kPkTail) {
        // lanes that passed column W-1 in this superblock: their capture is in this superblock's frame
        const int ts = c.W - 1 + lane;
        if (ts >= t0 && ts < t0 + 64) { c.capBX = c.baseX; c.capBY = c.baseY; }
    }
    c.mwX[0] = c.mwX[16];
    c.mwY[0] = c.mwY[16];
    c.widx += 16;
}

// Jobs 2c and 2c + 1 of the batch are the two pairs of couple c (same n, same W, whole matrix, winner codes on);
// order[ticket] = couple << 24 | group of four strips.
template <bool CODES>
__device__ __forceinline__ void fill_pk_body(const FillBatch& B) {
    __shared__ PkLds lds;
    __shared__ unsigned wg_ticket;
    extern __shared__ unsigned lds_keepout[];
    PkCtx c;
    const int lane = c.lane = threadIdx.x & 63;
    const int hw_wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) wg_ticket = atomicAdd(B.ticket, 1u);
    if (threadIdx.x < kPkNW) { lds.prod[threadIdx.x] = 0; lds.cons[threadIdx.x] = 0; }
    __syncthreads();
    const unsigned tk = __builtin_amdgcn_readfirstlane(wg_ticket);
    const unsigned og = B.order ? __builtin_amdgcn_readfirstlane(B.order[tk]) : tk;
    const unsigned group = og & ((1u << kOrderJobShift) - 1);
    const unsigned couple = og >> kOrderJobShift;
    // (strip = wavefront number.  Taking turns -- strip (wavefront + workgroup) mod 4, so that the strips that do more
    // than the others, the first and the last of a workgroup, are not always the same wavefront of a CU -- was measured:
    // 3 139 against 3 160-3 210 GCUPS.)
    const int wv = c.wv = hw_wv;
    const FillArgs2& qx = B.job[2 * couple];
    const FillArgs2& qy = B.job[2 * couple + 1];
    const FillArgs& px = qx.base;
    const FillArgs& py = qy.base;
    if (px.W < 0) lds_keepout[threadIdx.x] = group;   // (never true; keeps the dynamic LDS referenced)
    const int64_t n = px.n;
    const int64_t strips = (n + 63) >> 6;
    const int64_t sg = (int64_t)group * kPkNW + wv;
    if (sg >= strips) return;
    const int64_t i = 64 * sg + lane + 1;
    const bool row_ok = i <= n;
    const int W = c.W = px.W;
    c.Q = px.Q;
    c.nblocks = px.Q * (64 / kBlk);
    const bool last = (sg == strips - 1);
    c.to_lds = (wv + 1 < kPkNW) && !last;
    c.to_hbm = !c.to_lds && !last;
    c.from_lds = wv > 0;
    c.top = (sg == 0);
    c.abortX = px.abort_flag; c.abortY = py.abort_flag;
    c.dead = false;

    // ---- column 0 (simple.cpp:116-133; the same for both pairs), relative to the strip's first frame ----
    int o1, o2, o3, u1, u2, u3;
    boundary_triple(i, 0, o1, o2, o3);
    boundary_triple(i - 1, 0, u1, u2, u3);
    const int bias_own = (int)(3 * i);
    const int D_own = max3i(o1, o2, o3) + bias_own;
    const int H_own = max3i(o1, o2 + 2, o3) + bias_own;
    const int D_up = max3i(u1, u2, u3) + bias_own - 3;
    c.baseX = c.baseY = __builtin_amdgcn_readfirstlane(D_own) & ~15;
    int D0 = pk_pair(D_own - c.baseX, D_own - c.baseY), D1 = D0;
    int H = pk_pair(H_own - c.baseX, H_own - c.baseY);
    int V = 0;
    c.carryX = c.carryY = D_up;   // (lane 0's: D of the row above at column 0)

    // ---- substitution table rows of this lane in both pairs ----
    const int symX = (row_ok && !(PSA_PK_X & 64)) ? (int)qx.a_ids[i - 1] : 0;   // (64: timing only, every lane on one table row)
    const int symY = (row_ok && !(PSA_PK_X & 64)) ? (int)qy.a_ids[i - 1] : 0;
    c.trowX = qx.eqtab + (size_t)symX * px.tab_pitch + kTabPad;
    c.trowY = qy.eqtab + (size_t)symY * py.tab_pitch + kTabPad;
    c.msh = 8 * ((64 - lane) & 3);
    c.widx = (-lane - ((64 - lane) & 3)) / 4;
#pragma unroll
    for (int j = 0; j < 17; ++j) { c.mwX[j] = c.trowX[c.widx + j]; c.mwY[j] = c.trowY[c.widx + j]; }
    c.codesX = CODES ? px.codes + code_tile_offset(sg, px.Q, 0) : nullptr;
    c.codesY = CODES ? py.codes + code_tile_offset(sg, px.Q, 0) : nullptr;
    c.code_lane = 16u * (unsigned)lane;
    c.prod_seen = 0;
    c.cons_seen = 0;

    // ---- hand-off plumbing ----
    c.my_ring = lds.ring[wv];
    if (lane == 63 && c.to_lds) { c.out_base = (int*)lds.ring[wv + 1]; c.out_wrap = kRing * 2; }
    else if (lane == 63 && c.to_hbm) { c.out_base = (int*)lds.stage[wv]; c.out_wrap = kPkStage * 2; }
    else { c.out_base = lds.dummy[wv] + 4 * lane; c.out_wrap = 0; }
    c.prod_post = (lane == 63 && c.to_lds) ? &lds.prod[wv + 1] : &lds.dummy2[wv][lane];
    c.cons_post = (lane == 0 && c.from_lds) ? &lds.cons[wv] : &lds.dummy2[wv][lane];
    c.tag_in = (u64)sg;
    c.tag_out = (u64)(sg + 1);
    c.ginX = px.ring + (size_t)(((sg - 1) / kPkNW) & 1) * px.Wpad;
    c.ginY = py.ring + (size_t)(((sg - 1) / kPkNW) & 1) * py.Wpad;
    c.goutX = px.ring + (size_t)((sg / kPkNW) & 1) * px.Wpad;
    c.goutY = py.ring + (size_t)((sg / kPkNW) & 1) * py.Wpad;
    c.gbX = c.gbY = 0;

    // a workgroup far down the chain sleeps on the chain's front first (see fill_strips_v2)
    if (!PSA_PK_NOSYNC && wv == 0 && group >= 2) {
        Spin spins;
        for (;;) {
            const unsigned f = __builtin_amdgcn_readfirstlane(
                __hip_atomic_load(px.ticket + kFrontWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int d = (int)group - 1 - (int)f;
            if (d <= 0) break;
            const int naps = min(2 * d, 256);
            for (int j = 0; j < naps; ++j) __builtin_amdgcn_s_sleep(127);
            const unsigned ab = __hip_atomic_load(px.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long now = wall_clock64();
            if (spins.t0 == 0) spins.t0 = now;
            if (ab || now - spins.t0 > 4 * kSpinTicks) {
                __hip_atomic_store(px.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(py.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
    if (!c.from_lds && !c.top) {   // the first batch of granules
        const int cc = min(lane, W - 1);
        c.gbX = load_granule(c.ginX + cc);
        c.gbY = load_granule(c.ginY + cc);
    }

    // head, steady state, and -- for the strip that holds the last row -- the superblocks in which the lanes pass
    // column W-1, capturing
    const int Q = c.Q;
    const int capture_lo = last ? (W - 1) / 64 : Q;
    pk_superblock<CODES, kPkHead, kPkInAny>(c, lds, px, py, D0, D1, H, V, 0, sg);
    if (c.from_lds) {
        for (int sb = 1; sb < capture_lo; ++sb)
            pk_superblock<CODES, kPkFast, kPkInLds>(c, lds, px, py, D0, D1, H, V, sb, sg);
    } else {
        for (int sb = 1; sb < capture_lo; ++sb)
            pk_superblock<CODES, kPkFast, kPkInHbm>(c, lds, px, py, D0, D1, H, V, sb, sg);
    }
    c.cap1 = c.cap2 = c.cap3 = 0;   // (set here, not before the steady loop: five registers it does not have to carry)
    c.capBX = c.capBY = 0;
    for (int sb = capture_lo < 1 ? 1 : capture_lo; sb < Q; ++sb)
        pk_superblock<CODES, kPkTail, kPkInAny>(c, lds, px, py, D0, D1, H, V, sb, sg);

    if (c.to_hbm) {
        if (Q >= 2) pk_flush(c, lds, Q - 2, D0, false);
        pk_flush(c, lds, Q - 1, D0, false);
    }
    if (c.dead) {
        __hip_atomic_store(c.abortX, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c.abortY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last && i == n) {
        const int bias = (int)(3 * (i + W));
        px.final_triple[0] = pk_lo(c.cap1) + c.capBX - bias; px.final_triple[1] = pk_lo(c.cap2) + c.capBX - bias;
        px.final_triple[2] = pk_lo(c.cap3) + c.capBX - bias;
        py.final_triple[0] = pk_hi(c.cap1) + c.capBY - bias; py.final_triple[1] = pk_hi(c.cap2) + c.capBY - bias;
        py.final_triple[2] = pk_hi(c.cap3) + c.capBY - bias;
    }
}

// Wavefronts per SIMD (Makefile: PKWAVES).  Three: the steady loop fits 168 registers without spills and the launches of
// three engines sharing the chip run 6 % faster than with two (27.0 against 28.6 ms per launch of 20 pairs), although a
// launch alone is slower (32.3 against 31.1 ms: more wave slots to fill and to drain).  Four (128 registers) spill in
// the steady loop: 54 ms.
#ifndef PSA_PK_WAVES
#define PSA_PK_WAVES 3
#endif
template <bool CODES>
__global__ __launch_bounds__(kPkNW * 64) __attribute__((amdgpu_waves_per_eu(PSA_PK_WAVES, PSA_PK_WAVES))) void fill_strips_pk(FillBatch B) {
    fill_pk_body<CODES>(B);
}

}  // namespace psa
