// psa_kernels.hip.h -- gfx950 kernels: strip-wavefront fill and traceback walk.
//
// Fill (replaces the row/column loops of simple.cpp:122-240): one wavefront owns a strip of 64
// rows (lane l = row), sweeps it left to right with lane l one column behind lane l-1
// (anti-diagonal wavefront), the up / up-left dependencies cross lanes with DPP wave_shr:1,
// the left dependency stays in the lane.  Strips are chained top to bottom: a strip's last
// row is published column by column as 8-byte {tag, value} granules (write-through stores,
// polled by the next strip), so all strips of a stripe run as one pipelined launch.  Strip
// numbers are drawn from a ticket counter, so a strip only ever waits for one that already
// runs: no assumption on dispatch order or residency.
#pragma once
#include "psa_device.hip.h"

namespace psa {

struct FillArgs {
    const uint8_t* a;        // all of A (n bytes)
    const uint8_t* b;        // this stripe's columns of B (W bytes)
    int64_t n;               // rows of the whole matrix
    int64_t r0, r1;          // this chunk fills rows r0+1..r1
    int64_t c0;              // global column of the left boundary; stripe = columns c0+1..c0+W
    int W;                   // stripe width (>= 1)
    int Q;                   // blocks of 64 steps per strip = ceil((W+63)/64)
    int Wpad;                // ring pitch (granules per parity)
    const int32_t* left;     // [3][R] plain T1,T2,T3 at column c0 for rows r0+1..r1; null = column 0
    const int32_t* corner;   // [3] plain triple at (r0, c0) when left != null and r0 > 0
    int32_t* corner_next;    // [3] receives the triple at (r1, c0)
    int32_t* right;          // [3][R] plain triple at column c0+W, or null
    int32_t* final_triple;   // [3] plain triple at (n, c0+W), written by the strip holding row n
    u64* ring;               // [2][Wpad] granules: bottom row of strip s lives in parity s&1
    uint32_t* codes;         // packed winner bits or null
    unsigned* ticket;        // strip ticket counter (zeroed per launch)
    unsigned* abort_flag;    // set when a bounded spin expires
    u64* prof;               // optional [strips][4]: start, end (100 MHz ticks), placement; null = off
    u64* ckpt;               // banded mode: [bands][Wpad] granules; the last strip of band b-1 publishes its
    int64_t band_strips;     // bottom row to ckpt[b] instead of the ring and band b's first strip reads it
                             // there (0 = off).  A multiple of the strips per workgroup.
};

// where strip sg's bottom row goes / where strip sg's input row comes from, when it is not the ring
__device__ __forceinline__ u64* ckpt_row_out(const FillArgs& p, int64_t sg) {
    return (p.band_strips > 0 && (sg + 1) % p.band_strips == 0) ? p.ckpt + (size_t)((sg + 1) / p.band_strips) * p.Wpad : nullptr;
}
__device__ __forceinline__ const u64* ckpt_row_in(const FillArgs& p, int64_t sg) {
    return (p.band_strips > 0 && sg > 0 && sg % p.band_strips == 0) ? p.ckpt + (size_t)(sg / p.band_strips) * p.Wpad : nullptr;
}

__device__ __forceinline__ u64 load_granule(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_granule(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool CODES>
__global__ __launch_bounds__(64) void fill_strips_v1(FillArgs p) {
    __shared__ unsigned stage[64];
    const int lane = threadIdx.x;
    unsigned s_ = 0;
    if (lane == 0) s_ = atomicAdd(p.ticket, 1u);
    const int64_t s = __builtin_amdgcn_readfirstlane(s_);
    const int64_t row_base = p.r0 + 64 * s;   // global index of the row above this strip
    const int64_t sg = row_base >> 6;         // global strip number
    const int64_t R = p.r1 - p.r0;
    const int64_t i = row_base + lane + 1;    // this lane's row (1-based)
    const bool row_ok = i <= p.r1;
    const int rows_here = (int)min((int64_t)64, p.r1 - row_base);
    const int publane = rows_here - 1;
    const int W = p.W;
    const int64_t c0 = p.c0;

    // ---- left boundary: own row and the row above (plain -> biased aggregates) ----
    int o1, o2, o3, u1, u2, u3;
    if (p.left) {
        const int64_t y = i - p.r0 - 1;  // 0-based row in the chunk
        const int64_t yo = row_ok ? y : (R - 1);
        o1 = p.left[yo]; o2 = p.left[R + yo]; o3 = p.left[2 * R + yo];
        if (yo >= 1) { u1 = p.left[yo - 1]; u2 = p.left[R + yo - 1]; u3 = p.left[2 * R + yo - 1]; }
        else if (p.r0 > 0) { u1 = p.corner[0]; u2 = p.corner[1]; u3 = p.corner[2]; }
        else boundary_triple(0, c0, u1, u2, u3);
    } else {
        boundary_triple(i, 0, o1, o2, o3);
        boundary_triple(i - 1, 0, u1, u2, u3);
    }
    if (i == p.r1) { p.corner_next[0] = o1; p.corner_next[1] = o2; p.corner_next[2] = o3; }
    const int bias_own = (int)(3 * (i + c0));
    const int bias_up = bias_own - 3;
    const int D_own = max3i(o1, o2, o3) + bias_own;
    const int H_own = max3i(o1, o2 + 2, o3) + bias_own;
    const int D_up = max3i(u1, u2, u3) + bias_up;

    int Dreg0 = D_own, Dreg1 = D_own;  // D of this row one / two steps ago (alternating)
    int H = H_own, V = 0;
    int Dtop_prev = __builtin_amdgcn_readfirstlane(D_up);  // lane 0's up-left input
    const int a_res = row_ok ? (int)p.a[i - 1] : 0x100;
    int bsh = 0x200;                   // B residues shift down the lanes with the wavefront
    unsigned accQ = 0, accH = 0, accV = 0;
    int topD = 0, topV = 0, bch = 0;
    const u64* ring_in = p.ring + (size_t)((sg + 1) & 1) * p.Wpad;  // parity of strip sg-1
    u64* ring_out = p.ring + (size_t)(sg & 1) * p.Wpad;
    if (const u64* ci = ckpt_row_in(p, sg)) ring_in = ci;
    if (u64* co = ckpt_row_out(p, sg)) ring_out = co;
    const u64 tag_in = (u64)sg, tag_out = (u64)(sg + 1);

    for (int q = 0; q < p.Q; ++q) {
        // ---- stage the next 64 columns: B residues and the row above (lane k = column) ----
        const int jj0 = q * 64;
        if (jj0 < W) {
            const int col = jj0 + lane;
            const bool col_ok = col < W;
            bch = col_ok ? (int)p.b[col] : 0x200;
            if (sg == 0) {
                int t1, t2, t3;
                const int64_t j = c0 + 1 + col;
                boundary_triple(0, j, t1, t2, t3);
                const int bias = (int)(3 * j);
                topD = max3i(t1, t2, t3) + bias;
                topV = max3i(t1, t2, t3 + 2) + bias;
            } else {
                unsigned spins = 0;
                for (;;) {
                    u64 g = col_ok ? load_granule(ring_in + col) : (tag_in << 32);
                    const bool ok = (g >> 32) == tag_in;
                    if (__all(ok)) {
                        topD = (int)((unsigned)g >> 2);
                        topV = topD + (int)((unsigned)g & 3u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0) {
                        const unsigned ab = __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ab || spins > (1u << 28)) {
                            if (lane == 0) __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = d * 16 + u;
                const int t = jj0 + k;
                const int jj = t - lane;
                const bool act = row_ok && jj >= 0 && jj < W;
                // cross-lane inputs
                int upV = shift_down1(V, V);
                int upD = shift_down1(Dreg0, (k & 1) ? Dreg1 : Dreg0);
                bsh = shift_down1(bsh, bsh);
                const int tV = __builtin_amdgcn_readlane(topV, k);
                const int tD = __builtin_amdgcn_readlane(topD, k);
                const int tb = __builtin_amdgcn_readlane(bch, k);
                if (lane == 0) { upV = tV; upD = Dtop_prev; bsh = tb; }
                Dtop_prev = tD;
                const int sub = (a_res == bsh) ? kSubMatch : kSubMismatch;
                const int x1 = upD + sub, x2 = H, x3 = upV;
                const int D = max3i(x1, x2, x3);
                const int Hn = max(D, x2 + 2);
                const int Vn = max(D, x3 + 2);
                if (CODES) {
                    // layout: psa_device.hip.h, "packed winner codes" (accQ sums signed nibbles, like fill_strips_v2)
                    const int qc = min(max(x2 - x3, -3), 2);
                    accQ = (accQ << 4) + (unsigned)(2 * qc + (int)(x1 == D));
                    accH = (accH << 1) | (unsigned)(x1 == Hn);
                    accV = (accV << 1) | (unsigned)(x1 == Vn);
                    if ((k & 7) == 7) p.codes[code_word_index(sg, p.Q, q, k >> 5, lane, (k >> 3) & 3)] = accQ;
                    if ((k & 31) == 31) {
                        p.codes[code_word_index(sg, p.Q, q, 2, lane, k >> 5)] = accH;
                        p.codes[code_word_index(sg, p.Q, q, 2, lane, 2 + (k >> 5))] = accV;
                    }
                }
                if (act) {
                    if (k & 1) Dreg1 = D; else Dreg0 = D;
                    H = Hn;
                    V = Vn;
                    if (jj == W - 1) {
                        const int bias = (int)(3 * (i + c0 + W));
                        if (p.right) {
                            const int64_t y = i - p.r0 - 1;
                            p.right[y] = x1 - bias; p.right[R + y] = x2 - bias; p.right[2 * R + y] = x3 - bias;
                        }
                        if (i == p.n) {
                            p.final_triple[0] = x1 - bias; p.final_triple[1] = x2 - bias; p.final_triple[2] = x3 - bias;
                        }
                    }
                }
                // publish the strip's last row
                const int jjp = t - publane;
                if (jjp >= 0 && jjp < W) {
                    if (lane == publane) stage[jjp & 63] = ((unsigned)D << 2) | (unsigned)(Vn - D);
                    if ((jjp & 63) == 63 || jjp == W - 1) {
                        __syncthreads();
                        const int base = jjp & ~63;
                        if (lane <= (jjp & 63)) store_granule(ring_out + base + lane, (tag_out << 32) | stage[lane]);
                        __syncthreads();
                    }
                }
            }
        }
    }
}

// ---- traceback walk (replaces the per-cell alignment vectors of simple.cpp:164-211) ----
struct TraceArgs {
    const uint32_t* codes;
    int64_t c0;          // stripe = columns c0+1..c0+W
    int W, Q;
    int64_t cap;         // capacity of out_ai / out_bj
    int32_t* out_ai;     // columns in walk (reverse) order
    int32_t* out_bj;
    int64_t* state;      // in: i, j, k ; out: i, j, k, done (1 = reached the matrix boundary), len, error
    int64_t row_min;     // banded mode: the codes at hand cover rows > row_min only (0: whole matrix);
                         // `codes` then points row_min/64 strips before the band's first tile
};

__global__ __launch_bounds__(64) void traceback_v1(TraceArgs p) {
    if (threadIdx.x != 0) return;
    int64_t i = p.state[0], j = p.state[1];
    int k = (int)p.state[2];
    int64_t len = 0, done = 0, err = 0;
    const int64_t c0 = p.c0;
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
            const int64_t strip = (pi - 1) >> 6;
            const int lane = (int)((pi - 1) & 63);
            const int64_t t = (pj - c0 - 1) + lane;
            const uint32_t* tw = p.codes + code_word_index(strip, p.Q, t >> 6, 0, 0, 0);
            const int u = (int)(t & 63);
            w = code_winner(tw[code_q_offset(lane, u)], k == 1 ? 0u : tw[code_hv_offset(lane, u, k)], u, k);
        }
        if (len >= p.cap) { err = 1; break; }
        p.out_ai[len] = (w == 2) ? -1 : (int32_t)(i - 1);
        p.out_bj[len] = (w == 3) ? -1 : (int32_t)(j - 1);
        ++len;
        i = pi; j = pj; k = w;
    }
    p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
}


// The packed winner codes are stored in tiles of 64 rows x 64 wavefront steps (3 KB: three parts
// x 64 lanes x 16 bytes), the unit the fill kernel writes; the wave-cooperative walk caches tiles in LDS.
constexpr int kTbSlots = 16;

// ---- traceback, wave-cooperative (production): one wavefront walks ----
//   * while the walk is in table 1 and the predecessor's "table 1 wins" bit is set it moves on a
//     diagonal: all 64 lanes test their own row's bit at once, a ballot gives the run length
//     inside the tile and the lanes emit the run's columns in parallel;
//   * the walk state (i, j, table, output length, tile ids) is 32-bit and wave-uniform (SGPRs);
//     the host checks that the stripe's coordinates and tile count fit (else traceback_v1 runs);
//   * the tile directory (which tile sits in which LDS slot, and the issue number of its fetch)
//     lives in lanes 0..15 of two VGPRs: a lookup is one compare + ballot, no LDS round trip;
//   * tiles are fetched ahead of the walk along its current diagonal (LDS-DMA, asynchronous);
//     entering a tile waits only for that tile's fetch: "all but the N most recent vector-memory
//     operations are complete", N = the operations issued after it.  To make N known exactly the
//     output columns go out through buffer stores that are issued unconditionally, two per step
//     (lanes that have nothing to write aim past the end of the buffer and are dropped);
//   * in table 1 one ballot pass handles a whole diagonal run AND the non-diagonal step that
//     ends it (its two bits sit in the word the lane already holds).
constexpr int kTb3Ahead = 6;

__device__ __forceinline__ void tb_wait_vm(int n_after) {
    // wait until at most n_after vector-memory operations are outstanding (rounded down to a value that
    // exists as an immediate here): everything issued before the n_after most recent ones is complete
    if (n_after >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (n_after >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n_after >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n_after >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n_after >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n_after >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n_after >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n_after >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n_after >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n_after >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n_after >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(64) void traceback_wave(TraceArgs p) {
    __shared__ uint4 tile[kTbSlots][3][64];
    const int lane = threadIdx.x;
    int i = __builtin_amdgcn_readfirstlane((int)p.state[0]);
    int j = __builtin_amdgcn_readfirstlane((int)p.state[1]);
    int k = __builtin_amdgcn_readfirstlane((int)p.state[2]);
    const int c0 = (int)p.c0;
    const int Q = p.Q;
    const int rmin = (int)p.row_min;
    const int cap = (int)(p.cap < 0x1fffff00 ? p.cap : 0x1fffff00);   // 4 * cap fits the buffer descriptor
    int len = 0, done = 0, err = 0;
    int my_tile = -1, my_ops = 0;          // lanes 0..15: directory of the LDS slots (tile id, op count after its fetch)
    int victim = 0, ops = 0, ops_done = 0; // ops: vector-memory instructions issued so far
    const __amdgpu_buffer_rsrc_t rs_ai = __builtin_amdgcn_make_buffer_rsrc(p.out_ai, 0, cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bj = __builtin_amdgcn_make_buffer_rsrc(p.out_bj, 0, cap * 4, 0x00020000);
    int cur_T = -1, cur_slot = 0;
    int cpi = 0, ahead = 0;                // prefetch cursor: a row on the walk's diagonal, tiles ahead
    int n_iter = 0, n_miss = 0, n_tiles = 0;
    const uint4* codes4 = (const uint4*)p.codes;
    const long long tc0 = (long long)__builtin_readcyclecounter(), tw0 = (long long)wall_clock64();
#ifdef PSA_TB_TIMING
    long long tsec[4] = {0, 0, 0, 0}, tmark = tc0;
#define TB_MARK(x) { const long long now_ = (long long)__builtin_readcyclecounter(); tsec[x] += now_ - tmark; tmark = now_; }
#else
#define TB_MARK(x)
#endif

    auto lookup = [&](int T) -> int {
        const unsigned long long m = __ballot(my_tile == T);
        return m ? (int)__builtin_ctzll(m) : -1;
    };
    auto issue = [&](int T) {
#pragma unroll
        for (int kind = 0; kind < 3; ++kind)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(codes4 + ((long long)T * 3 + kind) * 64 + lane),
                (__attribute__((address_space(3))) void*)&tile[victim][kind][0], 16, 0, 0);
        ops += 3;
        if (lane == victim) { my_tile = T; my_ops = ops; }
        victim = (victim + 1) & (kTbSlots - 1);
    };

    for (;;) {
        if (i == 0 || j == 0) { done = 1; break; }
        ++n_iter;
        const int pi = i - (k != 2), pj = j - (k != 3);
        if (c0 > 0 && pj <= c0) break;  // predecessor belongs to the stripe on the left
        if (pi <= rmin && rmin > 0) break;   // ... or to the band of rows above
        if (pi == 0 || pj == 0) {
            int t1, t2, t3;
            boundary_triple(pi, pj, t1, t2, t3);
            const int w = winner_plain(t1, t2, t3, k);
            if (len >= cap) { err = 1; break; }
            {
                const int off = lane == 0 ? 4 * len : -1;
                __builtin_amdgcn_raw_buffer_store_b32((w == 2) ? -1 : (i - 1), rs_ai, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((w == 3) ? -1 : (j - 1), rs_bj, off, 0, 0);
                ops += 2;
            }
            ++len; i = pi; j = pj; k = w;
            continue;
        }
        const int strip = (pi - 1) >> 6;
        const int lp = (pi - 1) & 63;
        const int t = (pj - c0 - 1) + lp;
        const int q = t >> 6;
        const int T = strip * Q + q;
        if (T != cur_T) {
            ++n_tiles;
            int slot = lookup(T);
            if (slot < 0) {
                ++n_miss;
                slot = victim;
                issue(T);
                cpi = pi; ahead = 0;
            } else {
                ahead = ahead > 0 ? ahead - 1 : 0;
                if (cpi > pi) { cpi = pi; ahead = 0; }
            }
            // keep the fetches kTb3Ahead tiles ahead of the walk, along its current diagonal
            const int diag = pi - pj;
#pragma unroll 1
            for (int rep = 0; rep < 2 && ahead < kTb3Ahead && cpi >= 1; ++rep) {
                const int cj = cpi - diag;
                if (cj < c0 + 1) { cpi = 0; break; }
                const int bl = (cpi - 1) & 63;
                const int bt = (cj - c0 - 1) + bl;
                const int s1 = ((bt & 63) >> 1) + 1, s2 = bl + 1;
                cpi -= s1 < s2 ? s1 : s2;          // first cell of the next tile on the diagonal
                const int nj = cpi - diag;
                if (cpi <= rmin || nj < c0 + 1) { cpi = 0; break; }   // (rows <= rmin: no codes at hand)
                const int nl = (cpi - 1) & 63;
                const int Tn = ((cpi - 1) >> 6) * Q + (((nj - c0 - 1) + nl) >> 6);
                if (lookup(Tn) < 0) issue(Tn);
                ++ahead;
            }
            // the tile being entered must have landed
            const int sq = __builtin_amdgcn_readlane(my_ops, slot);
            if (sq > ops_done) { tb_wait_vm(ops - sq); ops_done = sq; }
            __builtin_amdgcn_s_barrier();   // single wave: orders the LDS reads below behind the wait
            cur_T = T; cur_slot = slot;
        }
        // ---- inside the tile: the walk goes on in tile-local coordinates (lane ll, step tl of the
        // predecessor cell) until it leaves the tile, the stripe or the matrix ----
        const uint32_t* tile_w = (const uint32_t*)&tile[cur_slot][0][0];   // [kind][lane][4] dwords
        int ll = lp, tl = t & 63;
        int gi = pi, gj = pj;   // the predecessor cell; the walk stands on (gi + (k != 2), gj + (k != 3)) in table k
        --n_iter;
        TB_MARK(0)
        do {
            ++n_iter;
            if (k == 1) {
                // lane l' looks at cell (gi - d, gj - d), d = ll - l': a run of "table 1 wins" bits from
                // lane ll downwards is a diagonal run; the first cell without the bit (if it is still
                // inside this tile and stripe) gives the step that follows the run
                const int dlim = min(tl >> 1, gj - c0 - 1);
                const int d = ll - lane;
                const int tt = tl - 2 * d;
                const bool valid = (unsigned)d <= (unsigned)dlim;
                unsigned pair = 0;
                if (valid) {
                    // aggregate D of step tt: nibble 2*(q'+3) + b1; pair = (b1, q' >= 0)
                    const unsigned nib = ((tile_w[code_q_offset(lane, tt)] + kCodeBias) >> (4 * (7 - (tt & 7)))) & 15u;
                    pair = ((nib & 1u) << 1) | (nib >= 6u ? 1u : 0u);
                }
                const unsigned long long mv = __ballot(valid);
                const unsigned long long m1 = __ballot((pair & 2u) != 0);
                const unsigned long long m23 = __ballot((pair & 1u) != 0);
                TB_MARK(1)
                const unsigned long long upto = (2ull << ll) - 1;           // lanes 0..ll (ll = 63: wraps to all ones)
                const unsigned long long inv = ~m1 & upto;
                const int lr = inv ? 63 - (int)__builtin_clzll(inv) : -1;   // lane of the first cell that is not a diagonal move
                const int r = ll - lr;                                       // run length
                const bool has_g = lr >= 0 && ((mv >> lr) & 1ull);
                const int w = ((m23 >> (lr & 63)) & 1ull) ? 2 : 3;
                const int cnt = r + (has_g ? 1 : 0);
                if (len + cnt > cap) { err = 1; break; }
                {
                    const bool g = (d == r);
                    const int off = (unsigned)d < (unsigned)cnt ? 4 * (len + d) : -1;   // -1: past the end, dropped
                    __builtin_amdgcn_raw_buffer_store_b32((g && w == 2) ? -1 : (gi - d), rs_ai, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32((g && w == 3) ? -1 : (gj - d), rs_bj, off, 0, 0);
                    ops += 2;
                }
                len += cnt;
                if (has_g) {
                    k = w;
                    gi -= r + (w != 2); gj -= r + (w != 3);
                    ll = lr - (w != 2); tl -= 2 * r + 1;
                } else {
                    gi -= r; gj -= r; ll = lr; tl -= 2 * r;
                }
                TB_MARK(2)
            } else {
                // tables 2 and 3 (inside a gap): one step
                const uint32_t qw = __builtin_amdgcn_readfirstlane(tile_w[code_q_offset(ll, tl)]);
                const uint32_t hw = __builtin_amdgcn_readfirstlane(tile_w[code_hv_offset(ll, tl, k)]);
                const int w = code_winner(qw, hw, tl, k);
                if (len >= cap) { err = 1; break; }
                {
                    const int off = lane == 0 ? 4 * len : -1;
                    __builtin_amdgcn_raw_buffer_store_b32((w == 2) ? -1 : (gi - (k == 2)), rs_ai, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32((w == 3) ? -1 : (gj - (k == 3)), rs_bj, off, 0, 0);
                    ops += 2;
                }
                ++len;
                k = w;
                ll -= (w != 2); tl -= (w != 2) + (w != 3);
                gi -= (w != 2); gj -= (w != 3);
                TB_MARK(3)
            }
        } while (ll >= 0 && tl >= 0 && gj > c0 && gi > rmin);
        if (err) break;
        i = gi + (k != 2); j = gj + (k != 3);
    }
    if (lane == 0) {
        p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
        p.state[6] = ((long long)n_iter << 32) | (unsigned)n_tiles; p.state[7] = n_miss;
        p.state[8] = (long long)__builtin_readcyclecounter() - tc0; p.state[9] = (long long)wall_clock64() - tw0;
#ifdef PSA_TB_TIMING
        p.state[10] = (tsec[0] << 32) | (tsec[1] & 0xffffffffLL); p.state[11] = (tsec[2] << 32) | (tsec[3] & 0xffffffffLL);
#endif
    }
}


}  // namespace psa
