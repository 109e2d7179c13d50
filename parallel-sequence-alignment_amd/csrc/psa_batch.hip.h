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

__global__ __launch_bounds__(kBatchWaves * 64) void batch_score_kernel(BatchArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* mine = smem + (size_t)wv * batch_wave_bytes(p.max_b);
    int2* row = (int2*)mine;
    uint8_t* bb = (uint8_t*)(mine + batch_row_slots(p.max_b) * 8) + 128;   // bb[c] = B[c]
    int* dummy = (int*)(mine + batch_row_slots(p.max_b) * 8 + p.max_b + 256);
    for (;;) {
        unsigned pk = 0;
        if (lane == 0) pk = atomicAdd(p.next_pair, 1u);
        const int64_t pair = __builtin_amdgcn_readfirstlane(pk);
        if (pair >= p.npairs) return;
        const int sa = p.pairs[2 * pair], sb = p.pairs[2 * pair + 1];
        const uint8_t* A = p.seqs + p.offsets[sa];
        const uint8_t* B = p.seqs + p.offsets[sb];
        const int nA = (int)(p.offsets[sa + 1] - p.offsets[sa]);
        const int nB = (int)(p.offsets[sb + 1] - p.offsets[sb]);
        if (nB > p.max_b) {
            // the per-wave row and B buffers are sized from max_b: a longer pair would write into the next
            // wave's buffers.  Not aligned here: the score is marked missing and the caller goes through
            // psa_score (psa_score_batch does; psa_score_batch_dev documents the sentinel).
            if (lane == 0) p.scores[pair] = kBatchSkipped;
            continue;
        }
        if (nA == 0 || nB == 0) {
            // no cell: the final pick reads the boundary (simple.cpp:246 after an empty loop / column 0 only)
            if (lane == 0) p.scores[pair] = (nA == 0) ? 0 : -1;
            continue;
        }
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

}  // namespace psa
