// psa_batch.hip.h -- many independent pairs, score only (BASELINE config 5).
//
// One wavefront aligns one pair at a time (pairs are drawn from a counter); a pair is swept
// strip by strip (64 rows each) with the same anti-diagonal step as the single-pair kernels.
// The last row of a strip reaches the next strip through a per-wave row buffer in LDS that is
// updated in place: lane 63 writes column t-63 while lane 0 reads column t, so the writer
// trails the reader.  No hand-off between wavefronts, no HBM traffic beyond the sequences and
// one score per pair: inter-sequence parallelism fills the chip (up to 32 waves per CU).
#pragma once
#include "psa_args.hip.h"

namespace psa {

// How a block of 16 steps treats lanes that are outside the pair's columns (as fill_strips_v2):
//   kBSteady  every lane is inside (t0 >= 64, t0 + 16 <= nB - 1): no predication at all
//   kBHead    lanes still entering (t < 64): commits are predicated on lane <= t
//   kBTail    lanes leaving on the right: nothing they compute is read again, so no predication
//   kBTailCap the same in the strip that holds the last row: the lane of row nA captures the final triple
//   kBGeneric short second sequences (nB < 96): full predicate and capture
// Rows below the last row (i > nA) need no predicate either: values only ever flow downwards.
enum : int { kBSteady = 0, kBHead = 1, kBTail = 2, kBTailCap = 3, kBGeneric = 4 };

struct BatchStrip {
    int D0, D1, H, V;
    int fin1, fin2, fin3;
};

template <int MODE>
__device__ __forceinline__ void batch_block(BatchStrip& S, const int2* row, const uint32_t* bw, int& bword, uint32_t& w0,
                                            const int bsh, const int a_res, int* pa, const int t0, const int lane,
                                            const int nB, const bool last_row) {
    const int4* rp = (const int4*)(row + t0 + 64);
    int4 pre[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) pre[x] = rp[x];
    const uint32_t w1 = bw[bword + 1], w2 = bw[bword + 2], w3 = bw[bword + 3], w4 = bw[bword + 4];
    uint32_t bq[4];
    bq[0] = __builtin_amdgcn_alignbit(w1, w0, bsh);
    bq[1] = __builtin_amdgcn_alignbit(w2, w1, bsh);
    bq[2] = __builtin_amdgcn_alignbit(w3, w2, bsh);
    bq[3] = __builtin_amdgcn_alignbit(w4, w3, bsh);
    w0 = w4;
    bword += 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = t0 + k;
        const int bres = (int)__builtin_amdgcn_ubfe(bq[k >> 2], 8 * (k & 3), 8);
        const int f = (a_res == bres) ? kSubMatch : kSubMismatch;
        const int preD = (k & 1) ? pre[k >> 1].z : pre[k >> 1].x;
        const int preV = (k & 1) ? pre[k >> 1].w : pre[k >> 1].y;
        const int x1 = shift_down1(preD, (k & 1) ? S.D1 : S.D0) + f;
        const int x3 = shift_down1(preV, S.V);
        const int x2 = S.H;
        const int D = max3i(x1, x2, x3);
        const int Hn = max(D, x2 + 2);
        const int Vn = max(D, x3 + 2);
        bool act = true;
        if (MODE == kBHead) act = lane <= t;
        if (MODE == kBGeneric) act = lane <= t && t - lane < nB;
        if (MODE == kBTailCap || MODE == kBGeneric) {
            if (last_row && t - lane == nB - 1) { S.fin1 = x1; S.fin2 = x2; S.fin3 = x3; }
        }
        if (act) {
            if (k & 1) S.D1 = D; else S.D0 = D;
            S.H = Hn;
            S.V = Vn;
        }
        const int Dc = (k & 1) ? S.D1 : S.D0;
        pa[2 * (k + 1) + 1] = S.V;   // V of column t-63 -> slot t+1
        pa[2 * (k + 2)] = Dc;        // D of column t-63 -> slot t+2
    }
}

// ---- one pair per wavefront, int32 (any lengths up to max_b columns) ----
__device__ __forceinline__ void batch_one_pair(const BatchArgs& p, const int64_t pair, const int lane, int2* row, uint8_t* bb,
                                               int* dummy) {
    const int sa = p.pairs[2 * pair], sb = p.pairs[2 * pair + 1];
    const uint8_t* A = p.seqs + p.offsets[sa];
    const uint8_t* B = p.seqs + p.offsets[sb];
    const int nA = (int)(p.offsets[sa + 1] - p.offsets[sa]);
    const int nB = (int)(p.offsets[sb + 1] - p.offsets[sb]);
    if (nB > p.max_b || nA == 0 || nB == 0) {
        // nB > max_b: the per-wave row and B buffers are sized from max_b, a longer pair would write into the next
        // wave's buffers.  Not aligned here: the score is marked missing and the caller goes through psa_score
        // (psa_score_batch does; psa_score_batch_dev documents the sentinel).
        // No cell: the final pick reads the boundary (simple.cpp:246 after an empty loop / column 0 only).
        if (lane == 0) p.scores[pair] = (nB > p.max_b) ? kBatchSkipped : (nA == 0) ? 0 : -1;
    } else {
    // ---- stage B and the top boundary row ----
    for (int c = lane; c < nB + 64; c += 64) bb[c] = (c < nB) ? B[c] : (uint8_t)0;
    for (int c = lane - 128; c < 0; c += 64) bb[c] = 0;
    for (int jj = lane; jj < nB + 32; jj += 64) {
        int t1, t2, t3, v1, v2, v3;
        boundary_triple(0, jj, t1, t2, t3);          // D of global column jj
        boundary_triple(0, jj + 1, v1, v2, v3);      // V' of global column jj + 1
        row[64 + jj] = make_int2(max3i(t1, t2, t3) + 3 * jj, max3i(v1, v2, v3 + 2) + 3 * (jj + 1));
    }
    BatchStrip S;
    S.fin1 = S.fin2 = S.fin3 = 0;
    const int nstrips = (nA + 63) >> 6;
    const int nblocks = (nB + 63 + 15) >> 4;
    // blocks [0, 4) are the head, [4, steady_hi) the steady state (t0 + 16 <= nB - 1: the step at which lane 0
    // passes the last column belongs to the tail, which captures), the rest the tail
    const bool wide = nB >= 96;
    const int steady_hi = wide ? (nB - 1) >> 4 : 0;
    for (int s = 0; s < nstrips; ++s) {
        const int i = 64 * s + lane + 1;
        const bool row_ok = i <= nA;
        const int a_res = row_ok ? (int)A[i - 1] : 0x100;
        int o1, o2, o3;
        boundary_triple(i, 0, o1, o2, o3);
        const int bias_own = 3 * i;
        S.D0 = S.D1 = max3i(o1, o2, o3) + bias_own;
        S.H = max3i(o1, o2 + 2, o3) + bias_own;
        S.V = 0;
        const bool full = (64 * s + 64 <= nA);   // only full strips feed a next strip
        int* out_base = (lane == 63 && full) ? (int*)row : dummy + lane;
        const int out_step = (lane == 63 && full) ? 32 : 0;   // ints per block of 16 steps
        const bool last_row = (i == nA);
        // window of B bytes for this lane: columns t0 - lane .. t0 - lane + 15
        const int bsh = ((64 - lane) & 3) * 8;
        int bword = (-lane - ((64 - lane) & 3)) / 4;   // dword index of column -lane (exact division)
        const uint32_t* bw = (const uint32_t*)bb;
        uint32_t w0 = bw[bword];
        int* pa = out_base;
        if (wide) {
            int blk = 0;
            for (; blk < 4; ++blk, pa += out_step)
                batch_block<kBHead>(S, row, bw, bword, w0, bsh, a_res, pa, blk * 16, lane, nB, last_row);
            for (; blk < steady_hi; ++blk, pa += out_step)
                batch_block<kBSteady>(S, row, bw, bword, w0, bsh, a_res, pa, blk * 16, lane, nB, last_row);
            if (s == nstrips - 1) {
                for (; blk < nblocks; ++blk, pa += out_step)
                    batch_block<kBTailCap>(S, row, bw, bword, w0, bsh, a_res, pa, blk * 16, lane, nB, last_row);
            } else {
                for (; blk < nblocks; ++blk, pa += out_step)
                    batch_block<kBTail>(S, row, bw, bword, w0, bsh, a_res, pa, blk * 16, lane, nB, last_row);
            }
        } else {
            for (int blk = 0; blk < nblocks; ++blk, pa += out_step)
                batch_block<kBGeneric>(S, row, bw, bword, w0, bsh, a_res, pa, blk * 16, lane, nB, last_row);
        }
    }
    // final pick (simple.cpp:246-256) on the lane that holds row nA
    const int owner = (nA - 1) & 63;
    const int bias = 3 * (nA + nB);
    const int t1 = __shfl(S.fin1, owner) - bias;
    const int t2 = __shfl(S.fin2, owner) - bias;
    const int t3 = __shfl(S.fin3, owner) - bias;
    int score = t3;
    if (t1 >= t2 && t1 >= t3) score = t1; else if (t2 >= t3) score = t2;
    if (lane == 0) p.scores[pair] = score;
    }
}

// ---- TWO pairs per wavefront in packed 16-bit halves (v_pk_add_u16 / v_pk_max_i16 / v_pk_min_u16) ----
// A couple of pairs whose second sequences have the same length share every column-dependent decision (head,
// steady state, tail, the capture step), so one lane can carry pair X in the low and pair Y in the high half of
// each register: the same 64-row strip sweep, one instruction stream, two cells per lane and step.  Every biased
// value is a non-negative number below 3 (nA + nB) + 2 min(nA, nB) + 8, which has to stay below 2^15 (kPkMaxSum
// bounds nA + nB).  Per step: 12 VALU instructions for two cells -- residue difference, min with 2 (residues are
// staged times two, so different residues differ by at least 2), 8 - that = the substitution score 8 | 6, two DPP
// lane shifts (the 32-bit move carries both halves), three packed adds, four packed max -- against 10 for one cell
// in the int32 kernel.  The rows of the two pairs may differ in number (rows below a pair's last row compute
// values nobody reads): each half's final triple is picked up from the strip and lane of its own last row.
typedef short pk16 __attribute__((ext_vector_type(2)));
constexpr int kPkMaxSum = 6000;     // nA + nB of a couple's longer pair: 3 * 6000 + 2 * 3000 + 8 < 2^15
constexpr int kPkBFront = 128;      // entries of padding in front of the packed B array

__device__ __forceinline__ pk16 pk_of(int v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ int pk_bits(pk16 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk16 pk_both(int v) { return pk_of((v & 0xffff) | (v << 16)); }

struct BatchStripPk {   // (plain ints, two 16-bit halves each: a struct of short vectors does not leave the stack)
    int D0, D1, H, V;
    int fin1, fin2, fin3;
};

// One block of 16 steps for a couple.  row: packed slots (D of column-1, V of column) of the row above; bp: this
// lane's window of the packed, doubled B residues (column t0 - lane first); a2: this lane's two residues, doubled.
template <int MODE>
__device__ __forceinline__ void batch_block_pk(BatchStripPk& S, const int2* row, const int* bp, const pk16 a2, int* pa,
                                               const bool publish, const int t0, const int lane, const int nB) {
    const int4* rp = (const int4*)(row + t0 + 64);
    int4 pre[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) pre[x] = rp[x];
    int bq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bq[k] = bp[k];
    int hd[16], hv[16];   // lane 63's row as it goes: (D one step ago, V) per step
    const pk16 two = pk_both(2), eight = pk_both(8);
    // (the slot address in a VGPR for the whole block -- as a uniform value it is moved there at every write --, kept
    // a 32-bit LDS address: the barrier on the generic pointer would turn the writes into flat stores)
    typedef __attribute__((address_space(3))) int lds_int;
    unsigned pa_off = (unsigned)(uintptr_t)(lds_int*)pa;
    asm volatile("" : "+v"(pa_off));
    lds_int* pl = (lds_int*)(uintptr_t)pa_off;
    // (the state in locals, the two D registers named explicitly per even / odd step: selecting a member of S by the step's
    // parity keeps the whole struct on the stack -- a scratch store and load per step)
    int D0 = S.D0, D1 = S.D1, H = S.H, V = S.V, f1 = S.fin1, f2 = S.fin2, f3 = S.fin3;
    typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
#define PSA_PK_STEP(k, DDIAG, DOUT, DPREV)                                                                           \
    {                                                                                                                \
        const int t = t0 + (k);                                                                                      \
        const upk16 d = __builtin_bit_cast(upk16, bq[k]) - __builtin_bit_cast(upk16, a2);                            \
        const upk16 z = __builtin_elementwise_min(d, __builtin_bit_cast(upk16, two)); /* 0: match, 2: mismatch */     \
        const pk16 f = eight - __builtin_bit_cast(pk16, z);                                                          \
        const int preD = ((k) & 1) ? pre[(k) >> 1].z : pre[(k) >> 1].x;                                              \
        const int preV = ((k) & 1) ? pre[(k) >> 1].w : pre[(k) >> 1].y;                                              \
        const pk16 x1 = pk_of(shift_down1(preD, DDIAG)) + f;                                                         \
        const pk16 x3 = pk_of(shift_down1(preV, V));                                                                 \
        const pk16 x2 = pk_of(H);                                                                                    \
        const pk16 M = pk_max(x1, x2);                                                                               \
        const pk16 D = pk_max(M, x3);                                                                                \
        const pk16 Vn = pk_max(M, x3 + two); /* = max(D, x3 + 2) */                                                  \
        const pk16 Hn = pk_max(D, x2 + two);                                                                         \
        bool act = true;                                                                                             \
        if (MODE == kBHead) act = lane <= t;                                                                         \
        if (MODE == kBGeneric) act = lane <= t && t - lane < nB;                                                     \
        if (MODE == kBTailCap || MODE == kBGeneric) {                                                                \
            if (t - lane == nB - 1) { f1 = pk_bits(x1); f2 = pk_bits(x2); f3 = pk_bits(x3); } /* every lane: the owner is chosen afterwards */ \
        }                                                                                                            \
        hd[k] = DPREV; /* D of this row one step (column) ago */                                                     \
        if (act) { DOUT = pk_bits(D); H = pk_bits(Hn); V = pk_bits(Vn); }                                            \
        hv[k] = V;                                                                                                   \
    }
    // lane 63's row leaves twice per block, eight slots at a time: slot of its column c = t - 63 is t + 1 = (D of column
    // c - 1, V of column c).  Only lane 63 writes (the other lanes have nobody to hand their row to), and eight writes
    // share ONE change of the execution mask (tools/pk_step.hip: per block 62, per step 63, unmasked writes of all lanes
    // 67 cycles a step at four wavefronts per SIMD, where the arithmetic alone takes 55-60; eight steps of history fit
    // the 128 registers of four wavefronts per SIMD, sixteen do not: 5 234 against 4 950 GCUPS).
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i lds_v2i;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2 += 2) {
        PSA_PK_STEP(k2, D0, D0, D1)
        PSA_PK_STEP(k2 + 1, D1, D1, D0)
        if (k2 == 6 && publish) {
#pragma unroll
            for (int k = 0; k < 8; ++k) *(lds_v2i*)(pl + 2 * (k + 1)) = v2i{hd[k], hv[k]};
        }
    }
#undef PSA_PK_STEP
    S.D0 = D0; S.D1 = D1; S.H = H; S.V = V; S.fin1 = f1; S.fin2 = f2; S.fin3 = f3;
    // (second half of the block's hand-off, see above)
    if (publish) {
#pragma unroll
        for (int k = 8; k < 16; ++k) *(lds_v2i*)(pl + 2 * (k + 1)) = v2i{hd[k], hv[k]};
    }
}

// pairs x (low halves) and y (high halves); same nB (checked by the caller), nA within kPkMaxSum - nB
__device__ __forceinline__ void batch_couple_pk(const BatchArgs& p, const int64_t px, const int64_t py, const int lane, int2* row,
                                                int* bpk, const int nB) {
    const int sax = p.pairs[2 * px], say = p.pairs[2 * py];
    const uint8_t* AX = p.seqs + p.offsets[sax];
    const uint8_t* AY = p.seqs + p.offsets[say];
    const uint8_t* BX = p.seqs + p.offsets[p.pairs[2 * px + 1]];
    const uint8_t* BY = p.seqs + p.offsets[p.pairs[2 * py + 1]];
    const int nAX = (int)(p.offsets[sax + 1] - p.offsets[sax]);
    const int nAY = (int)(p.offsets[say + 1] - p.offsets[say]);
    // ---- stage the packed B (doubled residues; 0x7ffe never equals a doubled residue) and the top boundary row ----
    for (int c = lane - kPkBFront; c < nB + 96; c += 64) {
        const bool in = c >= 0 && c < nB;
        bpk[c] = in ? (((int)BX[c] << 1) | ((int)BY[c] << 17)) : 0x7ffe7ffe;
    }
    for (int jj = lane; jj < nB + 32; jj += 64) {
        int t1, t2, t3, v1, v2, v3;
        boundary_triple(0, jj, t1, t2, t3);          // D of global column jj
        boundary_triple(0, jj + 1, v1, v2, v3);      // V' of global column jj + 1
        row[64 + jj] = make_int2(pk_bits(pk_both(max3i(t1, t2, t3) + 3 * jj)), pk_bits(pk_both(max3i(v1, v2, v3 + 2) + 3 * (jj + 1))));
    }
    BatchStripPk S;
    S.fin1 = S.fin2 = S.fin3 = 0;
    const int nAmax = max(nAX, nAY);
    const int nstrips = (nAmax + 63) >> 6;
    const int nblocks = (nB + 63 + 15) >> 4;
    const bool wide = nB >= 96;
    const int steady_hi = wide ? (nB - 1) >> 4 : 0;
    const int sx = (nAX - 1) >> 6, sy = (nAY - 1) >> 6;   // the strips that hold the last rows
    int fx1 = 0, fx2 = 0, fx3 = 0, fy1 = 0, fy2 = 0, fy3 = 0;
    for (int s = 0; s < nstrips; ++s) {
        const int i = 64 * s + lane + 1;
        const int ax = (i <= nAX) ? (int)AX[i - 1] : 0x100;   // (rows below a pair's last row: a residue nothing equals)
        const int ay = (i <= nAY) ? (int)AY[i - 1] : 0x100;
        const pk16 a2 = pk_of((ax << 1) | (ay << 17));
        int o1, o2, o3;
        boundary_triple(i, 0, o1, o2, o3);
        const int bias_own = 3 * i;
        S.D0 = S.D1 = pk_bits(pk_both(max3i(o1, o2, o3) + bias_own));
        S.H = pk_bits(pk_both(max3i(o1, o2 + 2, o3) + bias_own));
        S.V = 0;
        const bool publish = lane == 63 && s + 1 < nstrips;   // the last strip feeds nobody
        const bool cap = (s == sx) || (s == sy);
        const int* bp = bpk - lane;
        int* pa = (int*)row;
        if (wide) {
            int blk = 0;
            for (; blk < 4; ++blk, pa += 32, bp += 16) batch_block_pk<kBHead>(S, row, bp, a2, pa, publish, blk * 16, lane, nB);
            for (; blk < steady_hi; ++blk, pa += 32, bp += 16) batch_block_pk<kBSteady>(S, row, bp, a2, pa, publish, blk * 16, lane, nB);
            if (cap) {
                for (; blk < nblocks; ++blk, pa += 32, bp += 16) batch_block_pk<kBTailCap>(S, row, bp, a2, pa, publish, blk * 16, lane, nB);
            } else {
                for (; blk < nblocks; ++blk, pa += 32, bp += 16) batch_block_pk<kBTail>(S, row, bp, a2, pa, publish, blk * 16, lane, nB);
            }
        } else {
            for (int blk = 0; blk < nblocks; ++blk, pa += 32, bp += 16)
                batch_block_pk<kBGeneric>(S, row, bp, a2, pa, publish, blk * 16, lane, nB);
        }
        if (s == sx) {
            const int l = (nAX - 1) & 63;
            fx1 = (short)__builtin_amdgcn_readlane(S.fin1, l); fx2 = (short)__builtin_amdgcn_readlane(S.fin2, l);
            fx3 = (short)__builtin_amdgcn_readlane(S.fin3, l);
        }
        if (s == sy) {
            const int l = (nAY - 1) & 63;
            fy1 = __builtin_amdgcn_readlane(S.fin1, l) >> 16; fy2 = __builtin_amdgcn_readlane(S.fin2, l) >> 16;
            fy3 = __builtin_amdgcn_readlane(S.fin3, l) >> 16;
        }
    }
    // final picks (simple.cpp:246-256)
    {
        const int bias = 3 * (nAX + nB);
        const int t1 = fx1 - bias, t2 = fx2 - bias, t3 = fx3 - bias;
        int score = t3;
        if (t1 >= t2 && t1 >= t3) score = t1; else if (t2 >= t3) score = t2;
        if (lane == 0) p.scores[px] = score;
    }
    {
        const int bias = 3 * (nAY + nB);
        const int t1 = fy1 - bias, t2 = fy2 - bias, t3 = fy3 - bias;
        int score = t3;
        if (t1 >= t2 && t1 >= t3) score = t1; else if (t2 >= t3) score = t2;
        if (lane == 0) p.scores[py] = score;
    }
}

// Work is drawn in COUPLES of consecutive pairs (2c, 2c + 1): a couple whose second sequences have the same
// length (and whose sizes keep every value below 2^15) goes through the packed path, any other couple through
// the int32 path one pair after the other.  Callers that want the packed rate list pairs that share a target (or
// have targets of one length) next to each other -- an all-vs-all block ordered by target does (BASELINE config 5).
#ifndef PSA_BATCH_MIN_WAVES
#define PSA_BATCH_MIN_WAVES 4
#endif
__global__ __launch_bounds__(kBatchWaves * 64) __attribute__((amdgpu_waves_per_eu(PSA_BATCH_MIN_WAVES, 8))) void batch_score_kernel(BatchArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* mine = smem + (size_t)wv * batch_wave_bytes(p.max_b);
    int2* row = (int2*)mine;
    char* bregion = mine + batch_row_slots(p.max_b) * 8;                 // packed: int[max_b + 256]; int32 path: bytes
    uint8_t* bb = (uint8_t*)bregion + 128;                               // bb[c] = B[c]
    int* bpk = (int*)bregion + kPkBFront;                                // bpk[c] = 2 B_x[c] | 2 B_y[c] << 16
    int* dummy = (int*)(bregion + 4 * (p.max_b + 256));
    const int64_t ncouples = (p.npairs + 1) >> 1;
    for (;;) {
        unsigned pk = 0;
        if (lane == 0) pk = atomicAdd(p.next_pair, 1u);
        const int64_t couple = __builtin_amdgcn_readfirstlane(pk);
        if (couple >= ncouples) return;
        const int64_t px = 2 * couple, py = px + 1;
        int packed = 0, nB = 0;
        if (py < p.npairs && !p.no_packed) {
            const int sax = p.pairs[2 * px], sbx = p.pairs[2 * px + 1], say = p.pairs[2 * py], sby = p.pairs[2 * py + 1];
            const int nAX = (int)(p.offsets[sax + 1] - p.offsets[sax]), nAY = (int)(p.offsets[say + 1] - p.offsets[say]);
            const int nBX = (int)(p.offsets[sbx + 1] - p.offsets[sbx]), nBY = (int)(p.offsets[sby + 1] - p.offsets[sby]);
            nB = nBX;
            packed = nBX == nBY && nBX >= 1 && nBX <= p.max_b && nAX >= 1 && nAY >= 1 && max(nAX, nAY) + nBX <= kPkMaxSum;
        }
        // (wave-uniform by construction -- everything above derives from the couple number --, said so to the compiler:
        // the paths below shift values across lanes and must be entered by the whole wavefront)
        packed = __builtin_amdgcn_readfirstlane(packed);
        nB = __builtin_amdgcn_readfirstlane(nB);
        if (packed) {
            batch_couple_pk(p, px, py, lane, row, bpk, nB);
        } else {
            const int cnt = (py < p.npairs) ? 2 : 1;
            for (int h = 0; h < cnt; ++h) batch_one_pair(p, px + h, lane, row, bb, dummy);
        }
    }
}

}  // namespace psa
