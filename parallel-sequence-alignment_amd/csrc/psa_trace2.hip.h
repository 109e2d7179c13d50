// psa_trace2.hip.h -- the traceback walk with a second wavefront that does the fetching.
//
// Same walk, same LDS window and same passes as tb_walk (psa_trace1.hip.h, which stays as the single-wave
// variant, PSA_TRACEBACK_KERNEL=wave1).  A lone wavefront pays 6-7 cycles for every instruction it issues, and
// in tb_walk two thirds of the instructions of a tile entry belong to fetching: the cursor along the diagonal,
// the LDS-DMA issue, the directory update, the residue rings, the counted waits.  Here wave 1 of the workgroup
// (another SIMD of the same CU) does all of that and the walker's entry shrinks to: post (row, step), read six
// words of LDS, choose the rectangle.
//
//   walker (wave 0)  -> post[0..1] = (row, step) of the predecessor cell, post[2] = sequence number, at every
//                       tile entry; post[3] = 1 when the walk is over
//   fetcher (wave 1) -> tags[slot] = id of the tile that has LANDED in the slot; ready[0..1] = lowest block of
//                       256 residues of A / B that has landed in its ring (the two above it are there too)
//
// Rules that make this safe without any handshake per tile:
//   * the walker reads a tile only after it has seen its id in tags[] (written after the fetcher's wait for
//     its own DMA: vector-memory counters are per wavefront), and only tiles of the rectangle it chose from
//     the tags at its latest entry;
//   * the fetcher overwrites a slot only if the tile in it is DEAD for the position it last read -- its strip
//     or its block lies behind the walk, which only ever moves to lower rows and steps, so dead stays dead
//     however far the walker has moved on meanwhile -- or if the walker is waiting for the slot's own tile
//     (it then reads nothing);
//   * a tile the walker would like beside its own and does not find is simply not part of the rectangle; the
//     cell that needs it is decoded one at a time from global memory (tb_winner, the path of traceback_v1);
//   * every wait is bounded by wall-clock time: the walker gives up with an error, the fetcher just ends.
#pragma once
#include "psa_trace1.hip.h"

namespace psa {

constexpr int kT2Tags = kTbLdsBytes;          // int tags[16]
constexpr int kT2Post = kT2Tags + 64;         // int post[4]
constexpr int kT2Ready = kT2Post + 16;        // int ready[2] (+ 2 unused)
constexpr int kT2LdsBytes = kT2Ready + 16;
constexpr unsigned long long kT2WalkerTicks = 5ull * 100000000ull;    // 5 s at 100 MHz: a fetch is microseconds
constexpr unsigned long long kT2FetcherTicks = 10ull * 100000000ull;

__device__ __forceinline__ int t2_peek(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void t2_post(int* p, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// ---- wave 1: keeps the window, the rings and the tags ahead of the walker ----
__device__ __forceinline__ void t2_fetcher(const TraceArgs& p, uint8_t* lds) {
    const int lane = threadIdx.x & 63;
    const int Q = p.Q, W = p.W, nrows = (int)p.n;
    const int rmin = (int)p.row_min;
    int* tags = (int*)(lds + kT2Tags);
    int* post = (int*)(lds + kT2Post);
    int* ready = (int*)(lds + kT2Ready);
    int vtag = -1, vs = 0, vg = 0;       // lanes 0..15: tile id, strip and block (g) of the tile in each slot
    int cr = -1;                         // fetch cursor (row), as in tb_walk
    int a_lo = 0x7fffffff, b_lo = 0x7fffffff;   // lowest residue block issued so far
#ifdef PSA_TB_STATS
    int n_blocked = 0, n_rounds = 0, n_fetch = 0;
#endif
    int seen = 0, issued = 0;          // issued: vector-memory operations of the current round
    unsigned idle = 0;
    unsigned long long t_last = wall_clock64();
    const int lane4 = lane * 4;
    const int swz = (lane & 3) * 1024 + (lane >> 2) * 16;

    auto slot_of = [&](int s, int g) -> int { return ((g & 3) << 2) | (s & 3); };
    auto tile_id = [&](int s, int g) -> int { return s * Q + (g - s); };
    auto issue = [&](int s, int g, int slot, int T) {
        uint8_t* dst = lds + kTbCodes + (slot >> 2) * kTbPlaneBytes + (slot & 3) * 4096;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(p.codes + (long long)T * kTileBytes), 0, kTileBytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst), 16, swz, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 1024), 16, swz, 256, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 2048), 16, swz, 512, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 3072), 16, swz, 768, 0, 0);
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                     : "+v"(vtag), "+v"(vs), "+v"(vg) : "s"(T), "s"(s), "s"(g), "s"(slot) : "m0");
        issued += 4;
    };
    auto issue_a = [&](int blk) {
        const int left = nrows - 256 * blk;
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + 256 * (long long)blk), 0, left < 256 ? ((left + 3) & ~3) : 256, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (__attribute__((address_space(3))) void*)(lds + kTbRingA + 256 * (blk & 3)), 4, lane4, 0, 0, 0);
        issued += 1;
    };
    auto issue_b = [&](int blk) {
        const int left = W - 256 * blk;
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + 256 * (long long)blk), 0, left < 256 ? ((left + 3) & ~3) : 256, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (__attribute__((address_space(3))) void*)(lds + kTbRingB + 256 * (blk & 3)), 4, lane4, 0, 0, 0);
        issued += 1;
    };
    // fetch tile (s, g) unless it is there; only over an empty slot or a tile that is dead for (S0, G0)
    auto fetch_ahead = [&](int s, int g, int S0, int G0) -> bool {
        const int slot = slot_of(s, g), T = tile_id(s, g);
        const int held = __builtin_amdgcn_readlane(vtag, slot);
        if (held == T) return true;
        if (held >= 0) {
            const int hs = __builtin_amdgcn_readlane(vs, slot), hg = __builtin_amdgcn_readlane(vg, slot);
            if (((S0 - hs) | (G0 - hg)) >= 0) {                 // strip <= S0 and block <= G0: the walk may still come to it
#ifdef PSA_TB_STATS
                ++n_blocked;
#endif
                return false;
            }
        }
        issue(s, g, slot, T);
        return true;
    };
    auto fetch_beside = [&](int cs, int cg, int S0, int G0) {
        const int cq = cg - cs;
        const int cup = ((0 - cs) & (rmin - (cs << 6))) >> 31;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const int ok = c == 0 ? (0 - cq) >> 31 : (c == 1 ? cup & ((cq + 1 - Q) >> 31) : cup);
            if (ok) fetch_ahead(cs - (c != 0), cg - (c != 1), S0, G0);
        }
    };

    // A round issues its fetches, waits for them and publishes the directory.  (Publishing a round late, so that
    // the wave never sits out a memory latency with nothing in flight, was tried: the later tags cost more
    // one-cell steps than the extra fetch rate saved.)
    auto publish = [&](int tagv, int a, int b) {
        if (lane < 16) tags[lane] = tagv;
        if (lane == 0) { ready[0] = a; ready[1] = b; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (;;) {
        if (PSA_SGPR(t2_peek(post + 3))) break;
        const int seq = PSA_SGPR(t2_peek(post + 2));
        if (seq == seen) {
            __builtin_amdgcn_s_sleep(2);
            if ((++idle & 0x3fffu) == 0 && wall_clock64() - t_last > kT2FetcherTicks) break;
            continue;
        }
        seen = seq;
#ifdef PSA_TB_STATS
        ++n_rounds;
#endif
        idle = 0;
        t_last = wall_clock64();
        issued = 0;
        const int r = PSA_SGPR(t2_peek(post)), u = PSA_SGPR(t2_peek(post + 1));   // (written before the sequence number)
        const int S0 = r >> 6, G0 = u >> 6;
        const int T0 = tile_id(S0, G0), slot0 = slot_of(S0, G0);
        // the walker's own tile: it waits for it and reads nothing meanwhile, so whatever is in the slot may go
        if (__builtin_amdgcn_readlane(vtag, slot0) != T0) {
            issue(S0, G0, slot0, T0);
            cr = r;
        }
        cr = tb_smin(cr, r);
        int cu = 2 * cr + (u - 2 * r);
        // residue rings: the block in use, the one before it and one more
        {
            const int ablk = r >> 8, bblk = (u - r) >> 8;
            const int wa = ablk >= 2 ? ablk - 2 : 0, wb = bblk >= 2 ? bblk - 2 : 0;
            if (a_lo == 0x7fffffff) a_lo = ablk + 1;
            if (b_lo == 0x7fffffff) b_lo = bblk + 1;
#pragma unroll 1
            while (a_lo > wa) issue_a(--a_lo);
#pragma unroll 1
            while (b_lo > wb) issue_b(--b_lo);
        }
        // the tiles beside the walker's when it stands near its tile's upper left corner
        {
            const int tr = (r & 63) + 1, tl = ((u & 63) >> 1) + 1;
            if ((unsigned)(tr - tl + kTbCorner) <= 2u * kTbCorner) fetch_beside(S0, G0, S0, G0);
        }
        // the tiles ahead along the walk's current diagonal, and those beside the corners it passes
#pragma unroll 1
        for (int rep = 0; rep < 3; ++rep) {
            if ((cr | (kTbAheadRows - 1 - r + cr)) < 0) break;
            const int s1 = (cr & 63) + 1, s2 = ((cu & 63) >> 1) + 1;
            const int st = tb_smin(s1, s2);
            const int nr = cr - st, nu = cu - 2 * st;
            if (((nr - rmin) | (nu - nr)) < 0) { cr = -1; break; }
            if (!fetch_ahead(nr >> 6, nu >> 6, S0, G0)) break;
            if ((unsigned)(s1 - s2 + kTbCorner) <= 2u * kTbCorner) fetch_beside(cr >> 6, cu >> 6, S0, G0);
            cr = nr; cu = nu;
        }
        if (issued) {
#ifdef PSA_TB_STATS
            n_fetch += issued;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's fetches have landed in LDS
            publish(vtag, a_lo, b_lo);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PSA_TB_STATS
    if (lane == 0) { p.state[10] = ((long long)n_blocked << 32) | (unsigned)n_rounds; p.state[11] = n_fetch; }
#endif
}

// ---- wave 0: the walk ----
__device__ __forceinline__ void t2_walker(const TraceArgs& p, uint8_t* lds) {
    const int lane = threadIdx.x & 63;
    int i = PSA_SGPR((int)p.state[0]);
    int j = PSA_SGPR((int)p.state[1]);
    int k = PSA_SGPR((int)p.state[2]);
    const int c0 = (int)p.c0;
    if (p.start_triple) {
        const int t1 = PSA_SGPR(p.start_triple[0]), t2 = PSA_SGPR(p.start_triple[1]), t3 = PSA_SGPR(p.start_triple[2]);
        i = (int)p.n; j = c0 + p.W;
        k = (t1 >= t2 && t1 >= t3) ? 1 : (t2 >= t3 ? 2 : 3);
    }
    const int Q = p.Q;
    const int rmin = (int)p.row_min;
    const int cap = (int)(p.cap < 0x1fffff00 ? p.cap : 0x1fffff00);
    int* tags = (int*)(lds + kT2Tags);
    int* post = (int*)(lds + kT2Post);
    int* ready = (int*)(lds + kT2Ready);
    int len = 0, done = 0, err = 0, bad = 0, seq = 0;
    u64 badm = 0;
    int rlo = 0x7fffffff, ulo = 0x7fffffff;
    int rowfloor1 = 0, ufloor = 0;
#ifdef PSA_TB_STATS
    int n_iter = 0, n_miss = 0, n_tiles = 0, n_single = 0, n_c_row = 0, n_c_left = 0, n_c_len = 0;
#endif
    const long long tc0 = (long long)__builtin_readcyclecounter(), tw0 = (long long)wall_clock64();
    const unsigned lane2h = (unsigned)(2 * lane) << 24;
    const unsigned lane6 = (unsigned)lane << 6;
    const int lane4 = lane * 4;
    const int lane0m = lane == 0 ? -1 : 0;
    const __amdgpu_buffer_rsrc_t rs_ai = __builtin_amdgcn_make_buffer_rsrc(p.out_ai, 0, cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bj = __builtin_amdgcn_make_buffer_rsrc(p.out_bj, 0, cap * 4, 0x00020000);
    auto put1 = [&](int va, int vb) {
        const int off = lane == 0 ? 4 * len : -1;
        __builtin_amdgcn_raw_buffer_store_b32(va, rs_ai, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(vb, rs_bj, off, 0, 0);
    };

    for (;;) {
        // ---- cold loop: the walk stands on cell (i, j) in table k ----
        if (i == 0 || j == 0) { done = 1; break; }
        const int pi = i - (k != 2), pj = j - (k != 3);
        if (pj <= c0 || pi <= rmin) {
            if (c0 > 0 && pj <= c0) break;
            if (rmin > 0) break;
            int t1, t2, t3;
            boundary_triple(pi, pj, t1, t2, t3);
            const int w = winner_plain(t1, t2, t3, k);
            if (len >= cap) { err = 1; break; }
            put1((w == 2) ? -1 : (i - 1), (w == 3) ? -1 : (j - 1));
            ++len; i = pi; j = pj; k = w;
            continue;
        }
        int r = pi - 1;
        int u = (pj - c0 - 1) + r;
        int nH = -(int)(k == 2), nV = -(int)(k == 3);
        int fresh = 0;     // 1: the tags were read for this very cell
        for (;;) {
            if (((r - rlo) | (u - ulo)) < 0) {
                fresh = 1;
                // ---- enter the tile of the predecessor cell: tell the fetcher, read what has landed ----
                TB_STAT(++n_tiles);
                ++seq;
                t2_post(post, r);
                t2_post(post + 1, u);
                t2_post(post + 2, seq);
                const int S0 = r >> 6, G0 = u >> 6;
                rlo = S0 << 6; ulo = G0 << 6;
                const int q0 = G0 - S0, T0 = S0 * Q + q0;
                const int gs0 = (G0 & 3) << 2, gs1 = ((G0 - 1) & 3) << 2, ss0 = S0 & 3, ss1 = (S0 - 1) & 3;
                const int ablk = r >> 8, bblk = (u - r) >> 8;
                const int need_a = ablk >= 1 ? ablk - 1 : 0, need_b = bblk >= 1 ? bblk - 1 : 0;
                int t0, tL, tU0, tU1, ra_, rb_;
                unsigned spins = 0;
                unsigned long long t_first = 0;
                for (;;) {
                    t0 = t2_peek(tags + (gs0 | ss0)); tL = t2_peek(tags + (gs1 | ss0));
                    tU0 = t2_peek(tags + (gs0 | ss1)); tU1 = t2_peek(tags + (gs1 | ss1));
                    ra_ = t2_peek(ready); rb_ = t2_peek(ready + 1);
                    t0 = PSA_SGPR(t0); ra_ = PSA_SGPR(ra_); rb_ = PSA_SGPR(rb_);
                    if (PSA_LIKELY(t0 == T0 && ra_ <= need_a && rb_ <= need_b)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 0xfffu) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (t_first == 0) t_first = now;
                        if (now - t_first > kT2WalkerTicks) { err = 3; break; }
                    }
                }
                if (PSA_UNLIKELY(err)) break;
                TB_STAT(n_miss += spins != 0);
                tL = PSA_SGPR(tL); tU0 = PSA_SGPR(tU0); tU1 = PSA_SGPR(tU1);
                const int up = ((0 - S0) & (rmin - rlo)) >> 31;
                const int mexL = (0 - q0) >> 31, mexU0 = up & ((q0 + 1 - Q) >> 31);
                const int xL = (tL ^ (T0 - 1)) & mexL;
                const int xU0 = ((tU0 ^ (T0 - Q + 1)) & mexU0) | ~up;
                const int xU1 = (tU1 ^ (T0 - Q)) | ~up;
                const int xU = xU0 | (xL == 0 ? xU1 : 0);
                ufloor = xL == 0 ? ulo - 62 : ulo + 2;
                const int rowfloor = xU == 0 ? rlo - 64 : rlo;
                rowfloor1 = tb_smax(rowfloor, rmin) + 1;
            }
            // ---- one pass (as in tb_walk) ----
            const int dmax = tb_smin(tb_smin(kTbPassLanes - 1, u - r - 1), tb_smin(r - rowfloor1, (u - ufloor) >> 1));
            if (PSA_UNLIKELY(dmax < 0)) {
#ifdef PSA_TB_STATS
                if (fresh) { if (r - rowfloor1 < 0) ++n_c_row; else if (((u - ufloor) >> 1) < 0) ++n_c_left; else ++n_c_len; }
#endif
                if (fresh) break;
                rlo = 0x7fffffff;       // what this cell needs may have landed since the tile was entered: look again
                continue;
            }
            fresh = 0;
            TB_STAT(++n_iter);
            const unsigned y0 = ((unsigned)u << 24) - lane2h;
            const unsigned y1 = y0 - (1u << 24), y2 = y0 - (2u << 24);
            const unsigned m0 = __builtin_amdgcn_perm(y0, y0, 0x0c0c0303u);
            const unsigned m1 = __builtin_amdgcn_perm(y1, y1, 0x0c0c0303u);
            const unsigned m2 = __builtin_amdgcn_perm(y2, y2, 0x0c0c0303u);
            const unsigned w0 = ((unsigned)r << 6) - lane6, w1 = w0 - 64u;
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
            const int nk2 = lane0m & nH, nk3 = lane0m & nV;
            e2 += 2 * nk2;
            e3 += 2 * nk3;
            const u64 c12 = __builtin_amdgcn_ballot_w64(e1 <= e2);
            const u64 c13 = __builtin_amdgcn_ballot_w64(e1 <= e3);
            const u64 c23 = __builtin_amdgcn_ballot_w64(e2 <= e3);
            const u64 stop = ~(c12 & c13) | (1ull << (dmax + 1));
            const int run = (int)__builtin_ctzll(stop);
            const int has_g = (int)((unsigned)(run - dmax - 1) >> 31);
            const int gH = has_g & (int)((unsigned)(c23 >> run) & 1u);
            const int gV = has_g - gH;
            const int cnt = run + has_g;
            badm |= mb & ((1ull << cnt) - 1);
            if (PSA_UNLIKELY(len + cnt > cap)) { err = 1; break; }
            {
                const int gl2 = run | (gH - 1), gl3 = run | (gV - 1);
                const int va = (r + 1 - lane) + nk2, vb = (c0 + 1 + u - r - lane) + nk3;
                const int off = lane < cnt ? 4 * len + lane4 : -1;
                __builtin_amdgcn_raw_buffer_store_b32(lane == gl2 ? -1 : va, rs_ai, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(lane == gl3 ? -1 : vb, rs_bj, off, 0, 0);
            }
            len += cnt;
            nH = -gH; nV = -gV;
            r -= run + gV;
            u -= 2 * run + has_g;
        }
        if (err) break;
        k = 1 - nH - 2 * nV;
        const int qi = r + 1, qj = c0 + 1 + (u - r);
        i = qi + (k != 2); j = qj + (k != 3);
        {
            // a cell the passes cannot decode: one cell, from global memory (every neighbour source: boundary,
            // left boundary column, row above a band, code byte)
            TB_STAT(++n_single);
            int w;
            if (PSA_LIKELY(qi - 1 > rmin && qi - 1 >= 1 && qj - 1 > c0)) {
                // all three neighbours are cells of this stripe with codes: four independent loads, one wait
                const unsigned cP = tb_code(p, qi, qj), cD = tb_code(p, qi - 1, qj - 1), cL = tb_code(p, qi, qj - 1), cU = tb_code(p, qi - 1, qj);
                const unsigned ras = lds[kTbRingA + (r & 1023)], rbs = lds[kTbRingB + ((u - r) & 1023)];
                const int x1 = (int)(cD >> 4) + (ras == rbs ? kSubMatch : kSubMismatch);
                const int x2 = (int)((cL >> 4) + ((cL >> 2) & 3u));
                const int x3 = (int)((cU >> 4) + (cU & 3u));
                w = PSA_SGPR(winner_mod16((int)(cP >> 4), x1 & 15, x2 & 15, x3 & 15, k, &bad));
            } else {
                w = PSA_SGPR(tb_winner(p, qi, qj, k, &bad));
            }
            if (len >= cap) { err = 1; break; }
            put1((w == 2) ? -1 : (i - 1), (w == 3) ? -1 : (j - 1));
            ++len;
            i = qi; j = qj; k = w;
            rlo = 0x7fffffff;   // (the fetcher may have brought the neighbours meanwhile: enter the tile again)
        }
    }
    t2_post(post + 3, 1);
    if ((bad || badm) && !err) err = 2;
    if (lane == 0) {
        p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
#ifdef PSA_TB_STATS
        p.state[6] = ((long long)n_iter << 32) | (unsigned)n_tiles; p.state[7] = ((long long)n_single << 32) | (unsigned)n_miss;
        p.state[12] = n_c_row; p.state[13] = n_c_left; p.state[14] = n_c_len;
#endif
        p.state[8] = (long long)__builtin_readcyclecounter() - tc0; p.state[9] = (long long)wall_clock64() - tw0;
    }
}

__device__ __forceinline__ void t2_run(const TraceArgs& p) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[kT2LdsBytes];
    // directory and mailbox start empty (LDS is not cleared between workgroups)
    if (threadIdx.x < 16) ((int*)(lds + kT2Tags))[threadIdx.x] = -1;
    if (threadIdx.x < 4) ((int*)(lds + kT2Post))[threadIdx.x] = 0;
    if (threadIdx.x < 4) ((int*)(lds + kT2Ready))[threadIdx.x] = 0x7fffffff;
    __syncthreads();
    if (threadIdx.x < 64) t2_walker(p, lds);
    else t2_fetcher(p, lds);
}

__global__ __launch_bounds__(128) void traceback_pair(TraceArgs p) { t2_run(p); }

// several walks in one launch: workgroup w walks job w
__global__ __launch_bounds__(128) void traceback_pair_many(TraceBatch B) {
    const TraceArgs p = B.job[blockIdx.x];
    t2_run(p);
}

}  // namespace psa
