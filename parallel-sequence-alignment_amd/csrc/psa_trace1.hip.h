// psa_trace1.hip.h -- gfx950 kernels: the traceback walk, single-lane (traceback_v1) and single-wave
// (traceback_wave) variants; the production two-wave walk is psa_trace2.hip.h.  Argument blocks: psa_args.hip.h.
#pragma once
#include "psa_args.hip.h"

namespace psa {

// The values the reference compared at a cell, modulo 16 (psa_device.hip.h, "winner codes"), for any cell
// the walk can meet as a NEIGHBOUR of a predecessor cell: matrix boundary (analytic), the stripe's left
// boundary column (leftcol), the row above a band (toprow), or a stored code byte.
__device__ __forceinline__ unsigned tb_code(const TraceArgs& p, int64_t i, int64_t j) {
    const int64_t strip = (i - 1) >> 6;
    const int lane = (int)((i - 1) & 63);
    const int64_t t = (j - p.c0 - 1) + lane;
    return p.codes[code_tile_offset(strip, p.Q, t >> 6) + code_byte_offset(lane, (int)(t & 63))];
}
__device__ __forceinline__ int tb_D(const TraceArgs& p, int64_t i, int64_t j) {
    if (i == 0 || j == 0) return boundary_D(i, j);
    if (j == p.c0) return p.leftcol[i].x;
    if (i == p.row_min) return (int)((unsigned)p.toprow[j - p.c0 - 1] >> 2);
    return (int)(tb_code(p, i, j) >> 4);
}
__device__ __forceinline__ int tb_Hn(const TraceArgs& p, int64_t i, int64_t j) {
    if (j == 0) return boundary_Hn(i, 0);
    if (j == p.c0) return p.leftcol[i].y;
    const unsigned c = tb_code(p, i, j);
    return (int)((c >> 4) + ((c >> 2) & 3u));
}
__device__ __forceinline__ int tb_Vn(const TraceArgs& p, int64_t i, int64_t j) {
    if (i == 0) return boundary_Vn(0, j);
    if (i == p.row_min) {
        const unsigned g = (unsigned)p.toprow[j - p.c0 - 1];
        return (int)((g >> 2) + (g & 3u));
    }
    const unsigned c = tb_code(p, i, j);
    return (int)((c >> 4) + (c & 3u));
}
// winner at the interior cell (pi, pj) for aggregate k
__device__ __forceinline__ int tb_winner(const TraceArgs& p, int64_t pi, int64_t pj, int k, int* bad) {
    const int dp = (int)(tb_code(p, pi, pj) >> 4);
    const int sub = (p.a[pi - 1] == p.b[pj - p.c0 - 1]) ? kSubMatch : kSubMismatch;
    const int x1 = tb_D(p, pi - 1, pj - 1) + sub;
    const int x2 = tb_Hn(p, pi, pj - 1);
    const int x3 = tb_Vn(p, pi - 1, pj);
    return winner_mod16(dp, x1 & 15, x2 & 15, x3 & 15, k, bad);
}

__global__ __launch_bounds__(64) void traceback_v1(TraceArgs p) {
    if (threadIdx.x != 0) return;
    int64_t i = p.state[0], j = p.state[1];
    int k = (int)p.state[2];
    int64_t len = 0, done = 0, err = 0;
    const int64_t c0 = p.c0;
    int bad = 0;
    for (;;) {
        if (i == 0 || j == 0) { done = 1; break; }
        const int64_t pi = i - (k != 2), pj = j - (k != 3);
        if (c0 > 0 && pj <= c0) break;  // predecessor belongs to the stripe on the left
        if (p.row_min > 0 && pi <= p.row_min) break;  // ... or to the band of rows above
        int w;
        if (pi == 0 || pj == 0) {
            int t1, t2, t3;
            boundary_triple(pi, pj, t1, t2, t3);
            w = winner_plain(t1, t2, t3, k);
        } else {
            w = tb_winner(p, pi, pj, k, &bad);
        }
        if (len >= p.cap) { err = 1; break; }
        p.out_ai[len] = (w == 2) ? -1 : (int32_t)(i - 1);
        p.out_bj[len] = (w == 3) ? -1 : (int32_t)(j - 1);
        ++len;
        i = pi; j = pj; k = w;
    }
    if (bad) err = 2;
    p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
}

// ---- traceback, wave-cooperative (production): one wavefront walks ----
// A lone wavefront issues one instruction every four to five cycles whatever the instruction, so the
// walk is written for a low instruction count per tile entered and per pass:
//   * cells are addressed by (row, u), u = column-in-stripe + row: a global version of the "wavefront step"
//     the fill kernel stores by.  In these coordinates the neighbours of a cell are (row, u-1) left,
//     (row-1, u-1) up and (row-1, u-2) diagonal, across strip and block boundaries alike, and the code tile
//     of strip s, block q holds rows 64 s .. 64 s + 63 x steps u in [64 g, 64 g + 64), g = s + q;
//   * LDS holds a direct-mapped window of 16 tiles: tile (s, g) lives in plane g & 3, rows (64 s) & 255 of a
//     [4 planes][256 rows][64 steps] image, so a cell's LDS address is a bit merge of its row and step (no
//     directory look-up per cell); the fetch is four LDS-DMA instructions that re-lay the tile [row][step]
//     (each lane names its own 16 source bytes).  Residues sit in two 1 KB rings (A by row, B by column),
//     fetched 256 at a time two blocks ahead;
//   * the directory (tile id per slot and the number of the vector-memory operation that completed its
//     fetch) lives in lanes 0..15 of two VGPRs: a look-up is a v_readlane + compare.  Using a tile waits
//     only for that tile's fetch: "all but the N most recent vector-memory operations are complete";
//   * entering a tile looks up the tiles to its left and above; those present widen the rectangle of
//     (row, u) in which every lane of a pass may read, those absent narrow it (they are fetched when the
//     walk needs them, normally ahead of time: fetches run kTbAhead tiles ahead along the walk's diagonal);
//   * one pass = one step of any kind followed by the diagonal run behind it: lane d decodes the cell d
//     rows up the diagonal through the predecessor cell -- its own code byte and those of its left, upper
//     and upper-left neighbours plus the two residues --, the compares' lane masks give the length of the
//     run of "table 1 wins", the lanes emit the run's columns in parallel (buffer stores whose range check
//     drops the lanes past the run), and the non-diagonal step that ends the run is taken in the same pass;
//   * cells whose neighbours lie on the matrix boundary, on the stripe's left boundary column, in the row
//     above a band or in a tile that is not in LDS are decoded one at a time (rare);
//   * the walk state (row, u, table, output length) is 32-bit and wave-uniform (SGPRs); the host checks
//     that the stripe's coordinates and tile count fit (else traceback_v1 runs).
constexpr int kTbPlaneBytes = 16384;              // [256 rows][64 steps]
constexpr int kTbCodeBytes = 4 * kTbPlaneBytes;   // planes g & 3
constexpr int kTbRingA = 0;                       // A[row & 1023]
constexpr int kTbRingB = 1024;                    // B[column & 1023]
constexpr int kTbCodes = 2048;                    // (the rings come first: every address fits the 16-bit offset field)
constexpr int kTbLdsBytes = kTbCodes + kTbCodeBytes;
#ifndef PSA_TB_CORNER
#define PSA_TB_CORNER 4
#endif
constexpr int kTbCorner = PSA_TB_CORNER;                      // a diagonal that leaves a tile within this many rows of its corner: fetch the tiles beside it too
constexpr int kTbAheadRows = 128;                 // the fetches run this many rows (about six tiles) ahead; the window is four strips tall
constexpr int kTbFetchOps = 4;                    // vector-memory operations per tile fetch
constexpr int kTbPassLanes = 63;                  // lanes 0..62 decode (lane masks are built as (1 << n) - 1, n <= 63)

__device__ __forceinline__ void tb_wait_vm(int n_after) {
    // wait until at most n_after vector-memory operations are outstanding (rounded down to a value that
    // exists as an immediate here): everything issued before the n_after most recent ones is complete.
    // (one block of assembly: the compiler's version of this decision tree ran to some 25 instructions)
    asm volatile(
        "s_cmp_lt_u32 %0, 12\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_cmp_lt_u32 %0, 24\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_lt_u32 %0, 40\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_waitcnt vmcnt(40)\n\t"
        "s_branch 9f\n"
        "3:\n\t"
        "s_waitcnt vmcnt(24)\n\t"
        "s_branch 9f\n"
        "2:\n\t"
        "s_waitcnt vmcnt(12)\n\t"
        "s_branch 9f\n"
        "1:\n\t"
        "s_cmp_lt_u32 %0, 6\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_waitcnt vmcnt(6)\n\t"
        "s_branch 9f\n"
        "4:\n\t"
        "s_cmp_lt_u32 %0, 2\n\t"
        "s_cbranch_scc1 5f\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "s_branch 9f\n"
        "5:\n\t"
        "s_waitcnt vmcnt(0)\n"
        "9:\n"
        :: "s"(n_after) : "scc", "memory");
}

#define PSA_LIKELY(x) __builtin_expect(!!(x), 1)
#define PSA_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define PSA_SGPR(x) __builtin_amdgcn_readfirstlane(x)
// (mask & a) | (~mask & b), and a scalar minimum the compiler does not move to the vector unit
__device__ __forceinline__ unsigned tb_bfi(unsigned mask, unsigned a, unsigned b) {
    unsigned d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned tb_uminu(unsigned a, unsigned b) {
    unsigned d;
    asm("s_min_u32 %0, %1, %2" : "=s"(d) : "s"(a), "s"(b) : "scc");
    return d;
}
__device__ __forceinline__ int tb_smax(int a, int b) {
    int d;
    asm("s_max_i32 %0, %1, %2" : "=s"(d) : "s"(a), "s"(b) : "scc");
    return d;
}
__device__ __forceinline__ int tb_smin(int a, int b) {
    int d;
    asm("s_min_i32 %0, %1, %2" : "=s"(d) : "s"(a), "s"(b) : "scc");
    return d;
}

__device__ __forceinline__ void tb_walk(const TraceArgs& p) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[kTbLdsBytes];
    const int lane = threadIdx.x;
    int i = PSA_SGPR((int)p.state[0]);
    int j = PSA_SGPR((int)p.state[1]);
    int k = PSA_SGPR((int)p.state[2]);
    const int c0 = (int)p.c0;
    if (p.start_triple) {
        const int t1 = PSA_SGPR(p.start_triple[0]), t2 = PSA_SGPR(p.start_triple[1]), t3 = PSA_SGPR(p.start_triple[2]);
        i = (int)p.n; j = c0 + p.W;
        k = (t1 >= t2 && t1 >= t3) ? 1 : (t2 >= t3 ? 2 : 3);
    }
    const int Q = p.Q, W = p.W, nrows = (int)p.n;
    const int rmin = (int)p.row_min;
    const int cap = (int)(p.cap < 0x1fffff00 ? p.cap : 0x1fffff00);   // 4 * cap fits the buffer descriptor
    int len = 0, done = 0, err = 0, bad = 0;
    u64 badm = 0;                          // lanes whose decode check failed in a pass that used them
    int vtag = -1, vops = 0;               // lanes 0..15: directory of the window (tile id, op count after its fetch)
    int ops = 0, ops_done = 0;             // ops: vector-memory instructions issued so far
    int rlo = 0x7fffffff, ulo = 0x7fffffff;   // its first row and step (the passes stay at or above both); maximum: none entered
    int rowfloor1 = 0, ufloor = 0;         // a lane may decode a cell of row >= rowfloor1 and step >= ufloor
    int cr = -1;                           // fetch cursor: a row ahead of the walk (its step: on the walk's diagonal)
    int curA = 0x7fffffff, curB = 0x7fffffff, a_need = 0, a_new = 0, b_need = 0, b_new = 0;   // residue rings
#ifdef PSA_TB_STATS
    int n_iter = 0, n_miss = 0, n_tiles = 0, n_single = 0;   // (counters cost scalar registers the walk is short of)
#define TB_STAT(x) x
#else
#define TB_STAT(x)
#endif
    const long long tc0 = (long long)__builtin_readcyclecounter(), tw0 = (long long)wall_clock64();
    // per-lane constants
    const unsigned lane2h = (unsigned)(2 * lane) << 24;
    const unsigned lane6 = (unsigned)lane << 6;
    const int lane4 = lane * 4;
    const int lane0m = lane == 0 ? -1 : 0;
    // source offset of this lane's 16 bytes in the fetch that fills rows 16 f .. 16 f + 15 of a tile's LDS image
    // (f = 0..3 adds 256 f): the tile is stored [part][row][16]
    const int swz = (lane & 3) * 1024 + (lane >> 2) * 16;

    auto slot_of = [&](int s, int g) -> int { return ((g & 3) << 2) | (s & 3); };
    auto tile_id = [&](int s, int g) -> int { return s * Q + (g - s); };
    auto lds_of = [&](int row, int uu) -> int { return ((uu & 63) | ((row & 255) << 6) | (((uu >> 6) & 3) << 14)); };
    // fetch tile (s, g) into its slot of the window
    auto issue = [&](int s, int g, int slot, int T) {
        uint8_t* dst = lds + kTbCodes + (slot >> 2) * kTbPlaneBytes + (slot & 3) * 4096;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(p.codes + (long long)T * kTileBytes), 0, kTileBytes, 0x00020000);
        // (the 256 f goes into the scalar offset: an immediate offset would also move the LDS address)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst), 16, swz, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 1024), 16, swz, 256, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 2048), 16, swz, 512, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 3072), 16, swz, 768, 0, 0);
        ops += kTbFetchOps;
        // (value and lane select are both scalar: the lane select goes through m0)
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                     : "+v"(vtag), "+v"(vops) : "s"(T), "s"(ops), "s"(slot) : "m0");
    };
    // 256 residues of A (rows 256 blk ..) / B (stripe columns 256 blk ..) into their ring; lanes past the end of
    // the sequence are out of the descriptor's range and bring zeros (the range is rounded up to whole dwords -- a
    // dword that straddles the end would be dropped whole --: the sequence buffers carry kSeqPad bytes of padding)
    auto issue_a = [&](int blk) {
        const int left = nrows - 256 * blk;
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + 256 * (long long)blk), 0, left < 256 ? ((left + 3) & ~3) : 256, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (__attribute__((address_space(3))) void*)(lds + kTbRingA + 256 * (blk & 3)), 4, lane4, 0, 0, 0);
        ++ops;
    };
    auto issue_b = [&](int blk) {
        const int left = W - 256 * blk;
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + 256 * (long long)blk), 0, left < 256 ? ((left + 3) & ~3) : 256, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (__attribute__((address_space(3))) void*)(lds + kTbRingB + 256 * (blk & 3)), 4, lane4, 0, 0, 0);
        ++ops;
    };
    // every fetch up to operation number sq has landed
    auto landed_upto = [&](int sq) {
        if (sq > ops_done) {
            tb_wait_vm(ops - sq);
            ops_done = sq;
            __builtin_amdgcn_s_barrier();   // single wave: orders the LDS reads that follow behind the wait
        }
    };
    const __amdgpu_buffer_rsrc_t rs_ai = __builtin_amdgcn_make_buffer_rsrc(p.out_ai, 0, cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bj = __builtin_amdgcn_make_buffer_rsrc(p.out_bj, 0, cap * 4, 0x00020000);
    auto put1 = [&](int va, int vb) {   // one step, from lane 0
        const int off = lane == 0 ? 4 * len : -1;
        __builtin_amdgcn_raw_buffer_store_b32(va, rs_ai, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(vb, rs_bj, off, 0, 0);
        ops += 2;
    };
    // ---- one cell at a time (wave-uniform), any neighbour source: used next to boundaries only ----
    auto code_at = [&](int ci, int cj) -> unsigned {
        const int row = ci - 1, uu = (cj - c0 - 1) + row;
        const int s = row >> 6, g = uu >> 6, slot = slot_of(s, g), T = tile_id(s, g);
        if (__builtin_amdgcn_readlane(vtag, slot) != T) { TB_STAT(++n_miss); issue(s, g, slot, T); }
        landed_upto(__builtin_amdgcn_readlane(vops, slot));
        return (unsigned)PSA_SGPR((int)lds[kTbCodes + lds_of(row, uu)]);
    };
    auto D_at = [&](int ci, int cj) -> int {
        if (ci == 0 || cj == 0) return boundary_D(ci, cj);
        if (cj == c0) return PSA_SGPR(p.leftcol[ci].x);
        if (ci == rmin) return PSA_SGPR((int)((unsigned)p.toprow[cj - c0 - 1] >> 2));
        return (int)(code_at(ci, cj) >> 4);
    };
    auto Hn_at = [&](int ci, int cj) -> int {
        if (cj == 0) return boundary_Hn(ci, 0);
        if (cj == c0) return PSA_SGPR(p.leftcol[ci].y);
        const unsigned c = code_at(ci, cj);
        return (int)((c >> 4) + ((c >> 2) & 3u));
    };
    auto Vn_at = [&](int ci, int cj) -> int {
        if (ci == 0) return boundary_Vn(0, cj);
        if (ci == rmin) {
            const unsigned g = (unsigned)PSA_SGPR((int)(unsigned)p.toprow[cj - c0 - 1]);
            return (int)((g >> 2) + (g & 3u));
        }
        const unsigned c = code_at(ci, cj);
        return (int)((c >> 4) + (c & 3u));
    };

    // fetch tile (s, g) ahead of the walk unless it is in the window; false: its slot holds a tile the passes of
    // the current tile T0 may read (T0 itself or T0 - 1 left; T0 - Q + 1 above or T0 - Q above left) -- later
    auto fetch_ahead = [&](int s, int g, int T0) -> bool {
        const int slot = slot_of(s, g), T = tile_id(s, g);
        const int held = __builtin_amdgcn_readlane(vtag, slot);
        if (held == T) return true;
        if (tb_uminu((unsigned)(T0 - held), (unsigned)(T0 - Q + 1 - held)) <= 1u) return false;
        issue(s, g, slot, T);
        return true;
    };
    // the tiles beside tile (cs, cg) of the stripe that exist: left, above, above left
    auto fetch_beside = [&](int cs, int cg, int T0) {
        const int cq = cg - cs;
        const int cup = ((0 - cs) & (rmin - (cs << 6))) >> 31;     // -1: the rows above strip cs have codes
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            // c = 0: left (block cq - 1 >= 0); 1: above (block cq + 1 < Q); 2: above left
            const int ok = c == 0 ? (0 - cq) >> 31 : (c == 1 ? cup & ((cq + 1 - Q) >> 31) : cup);
            if (ok) fetch_ahead(cs - (c != 0), cg - (c != 1), T0);
        }
    };

    for (;;) {
        // ---- cold loop: the walk stands on cell (i, j) in table k ----
        if (i == 0 || j == 0) { done = 1; break; }
        const int pi = i - (k != 2), pj = j - (k != 3);   // the predecessor cell
        if (pj <= c0 || pi <= rmin) {
            if (c0 > 0 && pj <= c0) break;  // it belongs to the stripe on the left
            if (rmin > 0) break;            // ... or to the band of rows above
            // on the matrix boundary (pi == 0 or pj == 0)
            int t1, t2, t3;
            boundary_triple(pi, pj, t1, t2, t3);
            const int w = winner_plain(t1, t2, t3, k);
            if (len >= cap) { err = 1; break; }
            put1((w == 2) ? -1 : (i - 1), (w == 3) ? -1 : (j - 1));
            ++len; i = pi; j = pj; k = w;
            continue;
        }
        int r = pi - 1;                       // 0-based row of the predecessor cell
        int u = (pj - c0 - 1) + r;            // its step: 0-based column inside the stripe + row
        // (scalar state in plain integer arithmetic -- masks and sign bits, no booleans: the compiler turns uniform
        // booleans into 64-bit lane masks at several instructions apiece.  nH / nV = -1: the walk stands in table 2 / 3)
        int nH = -(int)(k == 2), nV = -(int)(k == 3);
        // ---- hot loop: tile entries and passes, until a cell the passes cannot decode ----
        for (;;) {
            if (((r - rlo) | (u - ulo)) < 0) {
                // ---- enter the tile of the predecessor cell ----
                TB_STAT(++n_tiles);
                const int S0 = r >> 6, G0 = u >> 6;
                rlo = S0 << 6; ulo = G0 << 6;
                const int q0 = G0 - S0, T0 = S0 * Q + q0;
                const int gs0 = (G0 & 3) << 2, gs1 = ((G0 - 1) & 3) << 2, ss0 = S0 & 3, ss1 = (S0 - 1) & 3;
                const int slot0 = gs0 | ss0, slotL = gs1 | ss0, slotU0 = gs0 | ss1, slotU1 = gs1 | ss1;
                if (PSA_UNLIKELY(__builtin_amdgcn_readlane(vtag, slot0) != T0)) {
                    TB_STAT(++n_miss);
                    issue(S0, G0, slot0, T0);
                    cr = r;
                }
                cr = tb_smin(cr, r);                 // (-1: the cursor has reached the end of the codes)
                int cu = 2 * cr + (u - 2 * r);       // the cursor follows the diagonal the walk is on now (gaps move the walk off the old one)
                // residue rings: blocks of 256 rows / columns, the block in use, the one before it and one more
                const int ablk = r >> 8, bblk = (u - r) >> 8;
                if (PSA_UNLIKELY(ablk != curA)) {
                    if (curA == 0x7fffffff) {
                        issue_a(ablk);
                        if (ablk >= 1) issue_a(ablk - 1);
                        a_new = ops;
                    }
                    if (ablk >= 2) issue_a(ablk - 2);
                    a_need = a_new; a_new = ops; curA = ablk;
                }
                if (PSA_UNLIKELY(bblk != curB)) {
                    if (curB == 0x7fffffff) {
                        issue_b(bblk);
                        if (bblk >= 1) issue_b(bblk - 1);
                        b_new = ops;
                    }
                    if (bblk >= 2) issue_b(bblk - 2);
                    b_need = b_new; b_new = ops; curB = bblk;
                }
                // keep the fetches kTbAheadRows rows (about six tiles) ahead of the walk, along its current diagonal (2
                // steps a row).  Where the diagonal leaves a tile within kTbCorner rows of its upper left corner the walk
                // reads the tiles beside the corner as well (and gaps may move it into them): those are fetched too.
#pragma unroll 1
                for (int rep = 0; rep < 2; ++rep) {
                    if ((cr | (kTbAheadRows - 1 - r + cr)) < 0) break;
                    const int s1 = (cr & 63) + 1, s2 = ((cu & 63) >> 1) + 1;
                    const int st = tb_smin(s1, s2);            // rows to the first cell of the next tile on the diagonal
                    const int nr = cr - st, nu = cu - 2 * st;
                    if (((nr - rmin) | (nu - nr)) < 0) { cr = -1; break; }   // (rows < rmin: no codes at hand; columns < 0: none)
                    if (!fetch_ahead(nr >> 6, nu >> 6, T0)) break;   // (later)
                    if (PSA_UNLIKELY((unsigned)(s1 - s2 + kTbCorner) <= 2u * kTbCorner)) fetch_beside(cr >> 6, cu >> 6, T0);
                    cr = nr; cu = nu;
                }
                // The tiles the passes may read beside this one: the block in front of it, and the strip above at this
                // block and the one in front.  Those in the window (they lie on or next to the diagonal) widen the
                // rectangle the lanes of a pass may touch; a tile that does not exist (in front of the stripe's first
                // block, behind its last) holds no cell a valid lane can name and counts as present.  A walk that enters
                // near the tile's upper left corner without them (gaps moved it off the diagonal the fetches followed)
                // asks for them now.  xL, xU0, xU1 = 0: the tile can be used.
                const int up = ((0 - S0) & (rmin - rlo)) >> 31;          // -1: the rows above the strip have codes
                const int mexL = (0 - q0) >> 31, mexU0 = up & ((q0 + 1 - Q) >> 31);   // -1: the tile exists
                int xL = (__builtin_amdgcn_readlane(vtag, slotL) ^ (T0 - 1)) & mexL;
                int xU0 = ((__builtin_amdgcn_readlane(vtag, slotU0) ^ (T0 - Q + 1)) & mexU0) | ~up;
                int xU1 = (__builtin_amdgcn_readlane(vtag, slotU1) ^ (T0 - Q)) | ~up;
                if (PSA_UNLIKELY((xL | ((xU0 | xU1) & up)) != 0)) {
                    const int tr = (r & 63) + 1, tl = ((u & 63) >> 1) + 1;
                    if ((unsigned)(tr - tl + kTbCorner) <= 2u * kTbCorner) {
                        fetch_beside(S0, G0, 0x3ffffff0);
                        xL = 0; xU0 = xU1 = ~up;
                    }
                }
                const int oL = __builtin_amdgcn_readlane(vops, slotL) & mexL;
                const int oU0 = __builtin_amdgcn_readlane(vops, slotU0) & mexU0;
                const int oU1 = __builtin_amdgcn_readlane(vops, slotU1) & up;
                // left: if xL == 0; above: if xU0 == 0, and with left only if xU1 == 0 as well
                const int xU = xU0 | (xL == 0 ? xU1 : 0);
                int sq = __builtin_amdgcn_readlane(vops, slot0);
                ufloor = xL == 0 ? ulo - 62 : ulo + 2;
                sq = tb_smax(sq, xL == 0 ? oL : 0);
                const int rowfloor = xU == 0 ? rlo - 64 : rlo;
                sq = tb_smax(sq, xU == 0 ? oU0 : 0);
                sq = tb_smax(sq, (xL | xU) == 0 ? oU1 : 0);
                rowfloor1 = tb_smax(rowfloor, rmin) + 1;
                landed_upto(tb_smax(sq, tb_smax(a_need, b_need)));
            }
            // ---- one pass: the predecessor cell is (r, u) ----
            // lane d decodes the cell d rows up the diagonal: row r - d, step u - 2 d, column (u - r) - d.  Its left
            // and upper neighbours must be cells of this stripe with codes in the window: column >= 1, and the
            // rectangle set when the tile was entered.
            const int dmax = tb_smin(tb_smin(kTbPassLanes - 1, u - r - 1), tb_smin(r - rowfloor1, (u - ufloor) >> 1));
            if (PSA_UNLIKELY(dmax < 0)) break;
            TB_STAT(++n_iter);
            const unsigned y0 = ((unsigned)u << 24) - lane2h;      // step of the lane's cell, modulo 256, in the top byte
            const unsigned y1 = y0 - (1u << 24), y2 = y0 - (2u << 24);
            const unsigned m0 = __builtin_amdgcn_perm(y0, y0, 0x0c0c0303u);   // step | step << 8
            const unsigned m1 = __builtin_amdgcn_perm(y1, y1, 0x0c0c0303u);
            const unsigned m2 = __builtin_amdgcn_perm(y2, y2, 0x0c0c0303u);
            const unsigned w0 = ((unsigned)r << 6) - lane6, w1 = w0 - 64u;    // row << 6 (the merge keeps bits 6..13 only)
            const unsigned cb = lds[kTbCodes + tb_bfi(0x3fc0u, w0, m0)];
            const unsigned cl = lds[kTbCodes + tb_bfi(0x3fc0u, w0, m1)];
            const unsigned cu_ = lds[kTbCodes + tb_bfi(0x3fc0u, w1, m1)];
            const unsigned cd = lds[kTbCodes + tb_bfi(0x3fc0u, w1, m2)];
            const unsigned ra = lds[kTbRingA + ((r - lane) & 1023)];
            const unsigned rb = lds[kTbRingB + ((u - r - lane) & 1023)];
            const int x1 = (int)(cd >> 4) + (ra == rb ? kSubMatch : kSubMismatch);
            const int x2 = (int)((cl >> 4) + ((cl >> 2) & 3u));
            const int x3 = (int)((cu_ >> 4) + (cu_ & 3u));
            const int dp = (int)(cb >> 4);
            const int e1 = (dp - x1) & 15;
            int e2 = (dp - x2) & 15, e3 = (dp - x3) & 15;
            const u64 mb = __builtin_amdgcn_ballot_w64(min(min(e1, e2), e3) != 0);
            const int nk2 = lane0m & nH, nk3 = lane0m & nV;   // the first step leaves table k, the others table 1
            e2 += 2 * nk2;
            e3 += 2 * nk3;
            const u64 c12 = __builtin_amdgcn_ballot_w64(e1 <= e2);
            const u64 c13 = __builtin_amdgcn_ballot_w64(e1 <= e3);
            const u64 c23 = __builtin_amdgcn_ballot_w64(e2 <= e3);
            const u64 stop = ~(c12 & c13) | (1ull << (dmax + 1));       // cells at which table 1 does not win; the lane past the last
            const int run = (int)__builtin_ctzll(stop);                 // diagonal moves in this pass
            const int has_g = (int)((unsigned)(run - dmax - 1) >> 31);  // the run ends in a decoded non-diagonal step ...
            const int gH = has_g & (int)((unsigned)(c23 >> run) & 1u);  // ... into table 2
            const int gV = has_g - gH;                                  // ... into table 3
            const int cnt = run + has_g;                                // (>= 1)
            badm |= mb & ((1ull << cnt) - 1);
            if (PSA_UNLIKELY(len + cnt > cap)) { err = 1; break; }
            {
                // step d leaves cell (row r - d + 1, ...) of table 1 -- step 0: cell (i, j) of table k; the lanes past
                // the pass aim behind the end of the buffer and are dropped
                const int gl2 = run | (gH - 1), gl3 = run | (gV - 1);   // lane of the gap step, or -1
                const int va = (r + 1 - lane) + nk2, vb = (c0 + 1 + u - r - lane) + nk3;
                const int off = lane < cnt ? 4 * len + lane4 : -1;
                __builtin_amdgcn_raw_buffer_store_b32(lane == gl2 ? -1 : va, rs_ai, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(lane == gl3 ? -1 : vb, rs_bj, off, 0, 0);
                ops += 2;
            }
            len += cnt;
            // the walk now stands on the cell its last step reached, in the table that step named; the new
            // predecessor cell is inside the rectangle (its row and column were a valid lane's, or one less)
            nH = -gH; nV = -gV;
            r -= run + gV;
            u -= 2 * run + has_g;
        }
        if (err) break;
        k = 1 - nH - 2 * nV;
        const int qi = r + 1, qj = c0 + 1 + (u - r);   // the predecessor cell now
        i = qi + (k != 2); j = qj + (k != 3);
        {
            // the passes stopped in front of a cell they cannot decode: a neighbour lies on a boundary or in a
            // tile that is not in the window.  One cell, wave-uniform.
            TB_STAT(++n_single);
            const unsigned cb = (unsigned)PSA_SGPR((int)lds[kTbCodes + lds_of(r, u)]);
            const unsigned ras = (unsigned)PSA_SGPR((int)lds[kTbRingA + (r & 1023)]);
            const unsigned rbs = (unsigned)PSA_SGPR((int)lds[kTbRingB + ((u - r) & 1023)]);
            const int x1 = D_at(qi - 1, qj - 1) + (ras == rbs ? kSubMatch : kSubMismatch);
            const int x2 = Hn_at(qi, qj - 1);
            const int x3 = Vn_at(qi - 1, qj);
            const int w = winner_mod16((int)(cb >> 4), x1 & 15, x2 & 15, x3 & 15, k, &bad);
            if (len >= cap) { err = 1; break; }
            put1((w == 2) ? -1 : (i - 1), (w == 3) ? -1 : (j - 1));
            ++len;
            i = qi; j = qj; k = w;
            rlo = 0x7fffffff;   // (the neighbours' tiles may have joined the window: enter the tile again)
        }
    }
    if ((bad || badm) && !err) err = 2;
    if (lane == 0) {
        p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
#ifdef PSA_TB_STATS
        p.state[6] = ((long long)n_iter << 32) | (unsigned)n_tiles; p.state[7] = ((long long)n_single << 32) | (unsigned)n_miss;
#endif
        p.state[8] = (long long)__builtin_readcyclecounter() - tc0; p.state[9] = (long long)wall_clock64() - tw0;
    }
}

__global__ __launch_bounds__(64) void traceback_wave(TraceArgs p) { tb_walk(p); }

// several walks in one launch: workgroup w walks job w
__global__ __launch_bounds__(64) void traceback_wave_many(TraceBatch B) {
    const TraceArgs p = B.job[blockIdx.x];
    tb_walk(p);
}

}  // namespace psa
