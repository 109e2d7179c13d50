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
    const int32_t* corner;   // [4] plain triple at (r0, c0) and the tag r0, when left != null and r0 > 0: published by
    int32_t* corner_next;    // the chunk above (its first strip stores {left triple of row r1, tag r1} here), which may
                             // still be running -- chunks of one stripe are launched on alternating streams
    int32_t* right;          // [3][R] plain triple at column c0+W, or null
    const unsigned* left_done;  // stripes of one pair filled by ONE launch: left_done[strip] != 0 once the stripe on the
    unsigned* right_done;       // left has stored that strip's rows of `left`; this stripe raises right_done[strip]
                                // behind its stores to `right` (both null: boundaries cross a kernel boundary)
    int32_t* final_triple;   // [3] plain triple at (n, c0+W), written by the strip holding row n
    u64* ring;               // [2][Wpad] granules: bottom row of strip s lives in parity s&1
    uint8_t* codes;          // winner codes (one byte per cell, psa_device.hip.h) or null
    int2* leftcol;           // c0 > 0 and codes: receives (D, H') (biased) of column c0 for every row, index = global row
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

// The triple at (r1, c0) for the chunk below: the last row of this chunk's left boundary.  Published by the FIRST
// strip of a chunk (the chunk below may be launched while this one runs and needs it at once), tag last.
__device__ __forceinline__ void publish_corner(const FillArgs& p, int64_t R) {
    __hip_atomic_store(p.corner_next + 0, p.left[R - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.corner_next + 1, p.left[2 * R - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.corner_next + 2, p.left[3 * R - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p.corner_next + 3, (int32_t)p.r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and its reader: waits (bounded) for the tag of this chunk's first row; false = gave up
__device__ __forceinline__ bool take_corner(const FillArgs& p, int& u1, int& u2, int& u3) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p.corner + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)p.r0) break;
        __builtin_amdgcn_s_sleep(8);
        if ((spins & 1023u) == 1023u) {
            const unsigned ab = __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab || spins > (1u << 26)) {
                __hip_atomic_store(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    u1 = __hip_atomic_load(p.corner + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u2 = __hip_atomic_load(p.corner + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u3 = __hip_atomic_load(p.corner + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
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
        else if (p.r0 > 0) { if (!take_corner(p, u1, u2, u3)) u1 = u2 = u3 = 0; }   // (gave up: the abort flag is set)
        else boundary_triple(0, c0, u1, u2, u3);
        if (s == 0 && lane == 0 && p.r1 < p.n) publish_corner(p, R);
    } else {
        boundary_triple(i, 0, o1, o2, o3);
        boundary_triple(i - 1, 0, u1, u2, u3);
    }
    const int bias_own = (int)(3 * (i + c0));
    const int bias_up = bias_own - 3;
    const int D_own = max3i(o1, o2, o3) + bias_own;
    const int H_own = max3i(o1, o2 + 2, o3) + bias_own;
    const int D_up = max3i(u1, u2, u3) + bias_up;
    if (CODES && p.leftcol && row_ok) p.leftcol[i] = make_int2(D_own, H_own);

    int Dreg0 = D_own, Dreg1 = D_own;  // D of this row one / two steps ago (alternating)
    int H = H_own, V = 0;
    int Dtop_prev = __builtin_amdgcn_readfirstlane(D_up);  // lane 0's up-left input
    const int a_res = row_ok ? (int)p.a[i - 1] : 0x100;
    int bsh = 0x200;                   // B residues shift down the lanes with the wavefront
    unsigned acc = 0;
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
                    // layout: psa_device.hip.h, "winner codes": four steps fill a dword
                    acc = __builtin_amdgcn_alignbit(code_byte(D, Hn, Vn), acc, 8);
                    if ((k & 3) == 3)
                        ((uint32_t*)(p.codes + code_tile_offset(sg, p.Q, q)))[((k >> 4) * 64 + lane) * 4 + ((k & 15) >> 2)] = acc;
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
    const uint8_t* codes;    // tile of strip 0, block 0 (banded mode: a virtual base, see row_min)
    const uint8_t* a;        // all of A
    const uint8_t* b;        // this stripe's columns of B: b[c] = residue of column c0 + 1 + c
    const int2* leftcol;     // c0 > 0: (D, H') (biased) of column c0, index = global row; null when c0 == 0
    const u64* toprow;       // row_min > 0: granules (D << 2 | V' - D) of row row_min, index = column - c0 - 1
    int64_t c0;              // stripe = columns c0+1..c0+W
    int W, Q;
    int64_t cap;             // capacity of out_ai / out_bj
    int32_t* out_ai;         // columns in walk (reverse) order
    int32_t* out_bj;
    int64_t* state;          // in: i, j, k ; out: i, j, k, done (1 = reached the matrix boundary), len, error
    int64_t row_min;         // banded mode: the codes at hand cover rows > row_min only (0: whole matrix);
                             // `codes` then points row_min/64 strips before the band's first tile
    int64_t n;               // rows of the whole matrix
    const int32_t* start_triple;   // not null: the walk starts at (n, c0 + W) in the table the final pick of this
                                   // triple names (simple.cpp:246-256) instead of at state[0..2]
};
constexpr int kMaxWalks = 8;
struct TraceBatch {          // one wavefront (workgroup) per walk
    TraceArgs job[kMaxWalks];
    int njobs;
};

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
//   * code tiles (4 KB: 64 rows x 64 wavefront steps, the unit the fill kernel writes) are cached in
//     LDS, re-laid out as [row][64 steps] by the fetch itself (each lane of the LDS-DMA names its own
//     source address), together with the residues they span (64 bytes of A, 128 bytes of B); tiles
//     are fetched ahead of the walk along its current diagonal, and using a tile waits only for that
//     tile's fetch: "all but the N most recent vector-memory operations are complete", N = the
//     operations issued after it.  To make N known exactly the output columns go out through buffer
//     stores that are issued unconditionally, two per pass (lanes with nothing to write aim past the end);
//   * the tile directory (which tile sits in which LDS slot, and the issue number of its fetch)
//     lives in lanes 0..15 of two VGPRs: a lookup is one compare + ballot, no LDS round trip;
//   * one pass = one step of any kind followed by the diagonal run behind it: lane l decodes the cell
//     of the diagonal through the predecessor cell that lies in row l of the strip -- its own code
//     byte, those of its left, upper and upper-left neighbours and the two residues, taken from the
//     tile itself, the tile to its left or, for row 0, the tile of the strip above --, the compares'
//     lane masks give the length of the run of "table 1 wins", the lanes emit the run's columns in
//     parallel, and the non-diagonal step that ends the run is taken in the same pass;
//   * cells whose neighbours lie on the matrix boundary, on the stripe's left boundary column or in
//     the row above a band are decoded one at a time (rare);
//   * the walk state (i, j, table, output length, tile ids) is 32-bit and wave-uniform (SGPRs);
//     the host checks that the stripe's coordinates and tile count fit (else traceback_v1 runs).
constexpr int kTbSlots = 16;
constexpr int kTbA = kTileBytes;          // slot layout: tile [64 rows][64 steps], A[64 rows] at kTbA,
constexpr int kTbB = kTileBytes + 64;     // B[128 columns from 64 q - 64] at kTbB, 192 bytes the fetch spills into
constexpr int kTbSlotBytes = kTileBytes + 384;
constexpr int kTbAhead = 6;
constexpr int kTbFetchOps = 6;            // vector-memory operations per tile fetch

__device__ __forceinline__ void tb_wait_vm(int n_after) {
    // wait until at most n_after vector-memory operations are outstanding (rounded down to a value that
    // exists as an immediate here): everything issued before the n_after most recent ones is complete
    if (n_after >= 12) {
        if (n_after >= 32) {
            if (n_after >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        } else if (n_after >= 18) {
            if (n_after >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (n_after >= 6) {
        if (n_after >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (n_after >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define PSA_LIKELY(x) __builtin_expect(!!(x), 1)
#define PSA_UNLIKELY(x) __builtin_expect(!!(x), 0)

__device__ __forceinline__ void tb_walk(const TraceArgs& p) {
    __shared__ __attribute__((aligned(16))) uint8_t tiles[kTbSlots * kTbSlotBytes];
    const int lane = threadIdx.x;
    int i = __builtin_amdgcn_readfirstlane((int)p.state[0]);
    int j = __builtin_amdgcn_readfirstlane((int)p.state[1]);
    int k = __builtin_amdgcn_readfirstlane((int)p.state[2]);
    const int c0 = (int)p.c0;
    if (p.start_triple) {
        const int t1 = __builtin_amdgcn_readfirstlane(p.start_triple[0]), t2 = __builtin_amdgcn_readfirstlane(p.start_triple[1]),
                  t3 = __builtin_amdgcn_readfirstlane(p.start_triple[2]);
        i = (int)p.n; j = c0 + p.W;
        k = (t1 >= t2 && t1 >= t3) ? 1 : (t2 >= t3 ? 2 : 3);
    }
    const int Q = p.Q;
    const int rmin = (int)p.row_min;
    const int cap = (int)(p.cap < 0x1fffff00 ? p.cap : 0x1fffff00);   // 4 * cap fits the buffer descriptor
    int len = 0, done = 0, err = 0, bad = 0;
    int my_tile = -1, my_ops = 0;          // lanes 0..15: directory of the LDS slots (tile id, op count after its fetch)
    int victim = 0, ops = 0, ops_done = 0; // ops: vector-memory instructions issued so far
    const __amdgpu_buffer_rsrc_t rs_ai = __builtin_amdgcn_make_buffer_rsrc(p.out_ai, 0, cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bj = __builtin_amdgcn_make_buffer_rsrc(p.out_bj, 0, cap * 4, 0x00020000);
    int cur_T = -1, cur_slot = -1;         // the tile of the predecessor cell ...
    int slotL = -1, slotU0 = -1, slotU1 = -1;   // ... and the tiles beside it that its passes read (all pinned: never a victim)
    int cpi = 0, ahead = 0;                // prefetch cursor: a row on the walk's diagonal, tiles ahead
    int n_iter = 0, n_miss = 0, n_tiles = 0, n_single = 0;
    const long long tc0 = (long long)__builtin_readcyclecounter(), tw0 = (long long)wall_clock64();
#ifdef PSA_TB_TIMING
    long long tsec[4] = {0, 0, 0, 0}, tmark = tc0;
#define TB_MARK(x) { const long long now_ = (long long)__builtin_readcyclecounter(); tsec[x] += now_ - tmark; tmark = now_; }
#else
#define TB_MARK(x)
#endif
    // source offset of this lane's 16 bytes in the fetch that fills rows 16 f .. 16 f + 15 of the LDS image
    // (f = 0..3 adds 256 f): the tile is stored [part][row][16]
    const int swz = (lane & 3) * 1024 + (lane >> 2) * 16;

    auto lookup = [&](int T) -> int {
        const unsigned long long m = __ballot(my_tile == T);
        return m ? (int)__builtin_ctzll(m) : -1;
    };
    // fetch tile (strip, q) and its residues into the next slot that is not pinned; returns the slot
    auto issue = [&](int strip, int q) -> int {
        while (victim == cur_slot || victim == slotL || victim == slotU0 || victim == slotU1) victim = (victim + 1) & (kTbSlots - 1);
        const int slot = victim;
        const int T = strip * Q + q;
        uint8_t* dst = tiles + slot * kTbSlotBytes;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(p.codes + (long long)T * kTileBytes), 0, kTileBytes, 0x00020000);
        // (the 256 f goes into the scalar offset: an immediate offset would also move the LDS address)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst), 16, swz, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 1024), 16, swz, 256, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 2048), 16, swz, 512, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(dst + 3072), 16, swz, 768, 0, 0);
        // residues (lanes past the 64 / 128 bytes are out of the descriptor's range: nothing is read for them)
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + 64 * (long long)strip), 0, 64, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + (64 * (long long)q - 64)), 0, 128, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (__attribute__((address_space(3))) void*)(dst + kTbA), 4, 4 * lane, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (__attribute__((address_space(3))) void*)(dst + kTbB), 4, 4 * lane, 0, 0, 0);
        ops += kTbFetchOps;
        if (lane == slot) { my_tile = T; my_ops = ops; }
        victim = (victim + 1) & (kTbSlots - 1);
        return slot;
    };
    // every fetch up to operation number sq has landed
    auto landed_upto = [&](int sq) {
        if (sq > ops_done) {
            tb_wait_vm(ops - sq);
            ops_done = sq;
            __builtin_amdgcn_s_barrier();   // single wave: orders the LDS reads that follow behind the wait
        }
    };
    // slot of tile (strip, q), fetched if need be, and landed
    auto ensure = [&](int strip, int q) -> int {
        int slot = lookup(strip * Q + q);
        if (slot < 0) { ++n_miss; slot = issue(strip, q); }
        landed_upto(__builtin_amdgcn_readlane(my_ops, slot));
        return slot;
    };
    // ---- one cell at a time (wave-uniform), any neighbour source: used next to boundaries only ----
    auto code_at = [&](int ci, int cj) -> unsigned {
        const int st = (ci - 1) >> 6, ln = (ci - 1) & 63, ta = (cj - c0 - 1) + ln;
        const int sl = ensure(st, ta >> 6);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)tiles[sl * kTbSlotBytes + ln * 64 + (ta & 63)]);
    };
    auto D_at = [&](int ci, int cj) -> int {
        if (ci == 0 || cj == 0) return boundary_D(ci, cj);
        if (cj == c0) return __builtin_amdgcn_readfirstlane(p.leftcol[ci].x);
        if (ci == rmin) return __builtin_amdgcn_readfirstlane((int)((unsigned)p.toprow[cj - c0 - 1] >> 2));
        return (int)(code_at(ci, cj) >> 4);
    };
    auto Hn_at = [&](int ci, int cj) -> int {
        if (cj == 0) return boundary_Hn(ci, 0);
        if (cj == c0) return __builtin_amdgcn_readfirstlane(p.leftcol[ci].y);
        const unsigned c = code_at(ci, cj);
        return (int)((c >> 4) + ((c >> 2) & 3u));
    };
    auto Vn_at = [&](int ci, int cj) -> int {
        if (ci == 0) return boundary_Vn(0, cj);
        if (ci == rmin) {
            const unsigned g = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p.toprow[cj - c0 - 1]);
            return (int)((g >> 2) + (g & 3u));
        }
        const unsigned c = code_at(ci, cj);
        return (int)((c >> 4) + (c & 3u));
    };

    for (;;) {
        if (i == 0 || j == 0) { done = 1; break; }
        int pi = i - (k != 2), pj = j - (k != 3);   // the predecessor cell
        if (PSA_UNLIKELY(pj <= c0 || pi <= rmin)) {
            if (c0 > 0 && pj <= c0) break;  // it belongs to the stripe on the left
            if (rmin > 0) break;            // ... or to the band of rows above
            // on the matrix boundary (pi == 0 or pj == 0)
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
        int ll = (pi - 1) & 63;                // row of the predecessor cell inside the strip
        int jl = pj - c0 - 1;                  // its 0-based column inside the stripe
        const int q = (jl + ll) >> 6;          // block of 64 wavefront steps
        int tl = (jl + ll) & 63;               // its step inside the block
        const int T = strip * Q + q;
        const bool up_strip = strip >= 1 && 64 * strip > rmin;     // the row above the strip has codes
        TB_MARK(0)
        if (T != cur_T) {
            ++n_tiles;
            cur_slot = slotL = slotU0 = slotU1 = -1;   // (nothing is pinned while the walk is between tiles)
            int slot = lookup(T);
            if (PSA_UNLIKELY(slot < 0)) {
                ++n_miss;
                slot = issue(strip, q);
                cpi = pi; ahead = 0;
            } else {
                ahead = ahead > 0 ? ahead - 1 : 0;
                if (cpi > pi) { cpi = pi; ahead = 0; }
            }
            cur_T = T; cur_slot = slot;
            // keep the fetches kTbAhead tiles ahead of the walk, along its current diagonal
            const int diag = pi - pj;
#pragma unroll 1
            for (int rep = 0; rep < 2 && ahead < kTbAhead && cpi >= 1; ++rep) {
                const int cj = cpi - diag;
                if (cj < c0 + 1) { cpi = 0; break; }
                const int bl = (cpi - 1) & 63;
                const int bt = (cj - c0 - 1) + bl;
                const int s1 = ((bt & 63) >> 1) + 1, s2 = bl + 1;
                cpi -= s1 < s2 ? s1 : s2;          // first cell of the next tile on the diagonal
                const int nj = cpi - diag;
                if (cpi <= rmin || cpi < 1 || nj < c0 + 1) { cpi = 0; break; }   // (rows <= rmin: no codes at hand)
                const int nl = (cpi - 1) & 63;
                const int ns = (cpi - 1) >> 6, nq = ((nj - c0 - 1) + nl) >> 6;
                if (lookup(ns * Q + nq) < 0) issue(ns, nq);
                ++ahead;
            }
            // The tiles the passes may read beside this one -- the block in front of it, and the strip above at
            // this block and the next -- if they are in the directory (they lie on or next to the diagonal): one
            // wait covers all of them.
            slotL = q >= 1 ? lookup(T - 1) : -1;
            slotU0 = up_strip ? lookup(T - Q) : -1;
            slotU1 = up_strip && q + 1 < Q ? lookup(T - Q + 1) : -1;
            int sq = __builtin_amdgcn_readlane(my_ops, slot);
            sq = max(sq, slotL >= 0 ? __builtin_amdgcn_readlane(my_ops, slotL & 15) : 0);
            sq = max(sq, slotU0 >= 0 ? __builtin_amdgcn_readlane(my_ops, slotU0 & 15) : 0);
            sq = max(sq, slotU1 >= 0 ? __builtin_amdgcn_readlane(my_ops, slotU1 & 15) : 0);
            landed_upto(sq);
        }
        TB_MARK(1)
        const int baseC = cur_slot * kTbSlotBytes;
        const unsigned ra = tiles[baseC + kTbA + lane];              // residue of this lane's row
        // ---- passes inside the tile: the predecessor cell is (row ll, step tl) of it ----
        for (;;) {
            // Lane ll's cell can be decoded from tiles when its left / upper-left neighbours are cells of this
            // stripe (jl >= 1) and the row above has codes: inside the strip, or the last row of the strip
            // above with both neighbours in one tile of it.  Lanes whose neighbours lie in a tile that is not in
            // LDS are left out; if lane ll is one of them, the slow path below brings the tile in.
            const int t0 = tl - 2 * ll;                               // step of the diagonal's cell in row 0
            const int slotU = (t0 + 62 >= 64) ? slotU1 : slotU0;
            const bool u_ok = slotU >= 0 && t0 >= -62 && jl - ll >= 1 && ((t0 + 63) & 63) != 0;
            if (PSA_UNLIKELY(!(jl >= 1 && (ll >= 1 || u_ok) && (slotL >= 0 || tl >= 2)))) break;
            TB_MARK(2)
            ++n_iter;
            const int bL = slotL * kTbSlotBytes + 64;
            const int aU1 = slotU * kTbSlotBytes + 63 * 64 + ((t0 + 63) & 63);   // row 0's upper neighbours: row 63 of the strip above
            const int aU2 = slotU * kTbSlotBytes + 63 * 64 + ((t0 + 62) & 63);
            // lane l' looks at the cell of row l' on the diagonal through the predecessor: d = ll - l' moves on
            const int d = ll - lane;
            const int tt = tl - 2 * d;             // its step inside the block (negative: the block in front)
            const bool valid = d >= 0 && jl - d >= 1 && tt >= (slotL >= 0 ? -62 : 2) && (lane >= 1 || u_ok);
            const int tv = valid ? tt : tl, lv = valid ? lane : ll;   // (the others repeat lane ll's cell)
            const int s0 = tv >= 0 ? baseC : bL, s1 = tv >= 1 ? baseC : bL, s2 = tv >= 2 ? baseC : bL;
            const int rt = lv * 64 + tv;
            const int au = lv == 0 ? aU1 : s1 + rt - 65;
            const int ad = lv == 0 ? aU2 : s2 + rt - 66;
            const unsigned cb = tiles[s0 + rt];
            const unsigned cl = tiles[s1 + rt - 1];
            const unsigned cu = tiles[au];
            const unsigned cd = tiles[ad];
            const unsigned rb = tiles[s0 + kTbB + 64 + (jl - ll - 64 * q) + lv];   // column jl - d of the window from 64 q - 64
            const int x1 = (int)(cd >> 4) + (ra == rb ? kSubMatch : kSubMismatch);
            const int x2 = (int)((cl >> 4) + ((cl >> 2) & 3u));
            const int x3 = (int)((cu >> 4) + (cu & 3u));
            const int dp = (int)(cb >> 4);
            const int e1 = (dp - x1) & 15;
            int e2 = (dp - x2) & 15, e3 = (dp - x3) & 15;
            const bool lbad = min(min(e1, e2), e3) != 0;
            e2 -= (d == 0 && k == 2) ? 2 : 0;      // the first step leaves table k, the others table 1
            e3 -= (d == 0 && k == 3) ? 2 : 0;
            const unsigned long long mv = __ballot(valid);
            const unsigned long long m1 = __ballot(valid && e1 <= e2 && e1 <= e3);
            const unsigned long long m2 = __ballot(e2 <= e3);
            const unsigned long long mb = __ballot(lbad);
            const unsigned long long upto = (2ull << ll) - 1;           // lanes 0..ll (ll = 63: wraps to all ones)
            const unsigned long long inv = ~m1 & upto;
            const int lr = inv ? 63 - (int)__builtin_clzll(inv) : -1;   // lane of the first cell that is not a diagonal move
            const int r = ll - lr;                                       // diagonal moves in this pass
            const int has_g = lr >= 0 ? (int)((mv >> (lr & 63)) & 1ull) : 0;   // ... followed by a decoded non-diagonal step
            const int w = ((m2 >> (lr & 63)) & 1ull) ? 2 : 3;
            const int cnt = r + has_g;                                   // (>= 1: lane ll is valid)
            bad |= (mb & upto & ~((1ull << (ll + 1 - cnt)) - 1)) ? 1 : 0;   // lanes ll - cnt + 1 .. ll were used
            if (PSA_UNLIKELY(len + cnt > cap)) { err = 1; break; }
            {
                // step d leaves cell (pi - d + 1, pj - d + 1) of table 1 -- step 0: cell (i, j) of table k
                const int wd = d < r ? 1 : w;
                const int off = (unsigned)d < (unsigned)cnt ? 4 * (len + d) : -1;   // -1: past the end, dropped
                const int va = pi - d - ((d == 0 && k == 2) ? 1 : 0), vb = pj - d - ((d == 0 && k == 3) ? 1 : 0);
                __builtin_amdgcn_raw_buffer_store_b32((wd == 2) ? -1 : va, rs_ai, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((wd == 3) ? -1 : vb, rs_bj, off, 0, 0);
                ops += 2;
            }
            len += cnt;
            // the walk now stands on the cell its last step reached, in the table that step named
            const int di = r + (has_g & (w != 2)), dj = r + (has_g & (w != 3));
            k = has_g ? w : 1;
            pi -= di; pj -= dj; ll -= di; jl -= dj; tl -= 2 * r + has_g;
            TB_MARK(3)
            if (!(ll >= 0 && tl >= 0 && jl >= 0)) break;
        }
        if (PSA_UNLIKELY(err)) break;
        i = pi + (k != 2); j = pj + (k != 3);
        if (PSA_UNLIKELY(ll >= 0 && tl >= 0 && jl >= 0)) {
            // the pass loop stopped in front of a cell it cannot decode: a neighbour lies on a boundary, the
            // two upper ones lie in different tiles, or a tile is not in LDS.  One cell, wave-uniform.
            ++n_single;
            const unsigned cb = (unsigned)__builtin_amdgcn_readfirstlane((int)tiles[baseC + ll * 64 + tl]);
            const unsigned ras = (unsigned)__builtin_amdgcn_readlane((int)ra, ll);
            const unsigned rb = (unsigned)__builtin_amdgcn_readfirstlane((int)tiles[baseC + kTbB + 64 + jl - 64 * q]);
            const int x1 = D_at(pi - 1, pj - 1) + (ras == rb ? kSubMatch : kSubMismatch);
            const int x2 = Hn_at(pi, pj - 1);
            const int x3 = Vn_at(pi - 1, pj);
            const int w = winner_mod16((int)(cb >> 4), x1 & 15, x2 & 15, x3 & 15, k, &bad);
            if (len >= cap) { err = 1; break; }
            {
                const int off = lane == 0 ? 4 * len : -1;
                __builtin_amdgcn_raw_buffer_store_b32((w == 2) ? -1 : (i - 1), rs_ai, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((w == 3) ? -1 : (j - 1), rs_bj, off, 0, 0);
                ops += 2;
            }
            ++len;
            i = pi; j = pj; k = w;
            cur_T = -1;   // (the slow path may have replaced the tiles beside this one: look them up again)
        }
    }
    if (bad && !err) err = 2;
    if (lane == 0) {
        p.state[0] = i; p.state[1] = j; p.state[2] = k; p.state[3] = done; p.state[4] = len; p.state[5] = err;
        p.state[6] = ((long long)n_iter << 32) | (unsigned)n_tiles; p.state[7] = ((long long)n_single << 32) | (unsigned)n_miss;
        p.state[8] = (long long)__builtin_readcyclecounter() - tc0; p.state[9] = (long long)wall_clock64() - tw0;
#ifdef PSA_TB_TIMING
        p.state[10] = (tsec[0] << 32) | (tsec[1] & 0xffffffffLL); p.state[11] = (tsec[2] << 32) | (tsec[3] & 0xffffffffLL);
#endif
    }
}

__global__ __launch_bounds__(64) void traceback_wave(TraceArgs p) { tb_walk(p); }

// several walks in one launch: workgroup w walks job w
__global__ __launch_bounds__(64) void traceback_wave_many(TraceBatch B) {
    const TraceArgs p = B.job[blockIdx.x];
    tb_walk(p);
}

}  // namespace psa
