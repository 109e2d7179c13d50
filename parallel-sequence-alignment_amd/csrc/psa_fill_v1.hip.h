// psa_fill_v1.hip.h -- gfx950 kernels: the generic strip-wavefront fill (argument blocks: psa_args.hip.h).
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
#include "psa_args.hip.h"

namespace psa {

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

}  // namespace psa
