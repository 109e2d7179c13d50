// psa_fill_v2.hip.h -- the production fill kernel for gfx950 (alphabets of <= 32 symbols).
//
// Same mathematics and the same outputs as fill_strips_v1 (psa_kernels.hip.h), organised for
// a short dependent instruction stream per anti-diagonal step, because at 2^16 x 2^16 the
// fill is bound by the length of the dependency chain (N + M steps), not by throughput:
//   * one workgroup = kNW wavefronts = kNW consecutive strips of 64 rows; strip w hands its
//     last row to strip w+1 through a ring in LDS (slot per column, flow-controlled), only the
//     last strip of a workgroup publishes through HBM granules (as v1 does for every strip);
//   * substitution scores come from a per-symbol table of 4-bit entries (6 or 8, the biased
//     mismatch / match score) built once per stripe: one v_bfe_u32 per step instead of
//     shifting B through the lanes and comparing;
//   * lane 0's inputs (row above the strip) are pre-loaded, 16 steps at a time, into the
//     destination registers of the DPP moves: a DPP wave_shr:1 leaves lane 0's destination
//     untouched, so the per-step code has no special case for the strip's first row;
//   * winner bits are shifted into per-lane accumulators with v_cmp + v_addc_co_u32
//     (acc = 2*acc + bit) and leave as 16-byte-per-lane coalesced stores, layout as v1;
//   * steps run in blocks of 16; blocks in which every lane is inside the stripe take the
//     unpredicated path, the 64-step head and tail of a strip take a predicated one.
#pragma once
#include "psa_kernels.hip.h"

namespace psa {

constexpr int kNW = 4;         // strips (wavefronts) per workgroup
constexpr int kRing = 256;     // LDS ring slots per strip boundary (8 bytes each)
constexpr int kBlk = 16;       // steps per block
constexpr int kMaxSym = 32;    // rows of the substitution table
constexpr int kTabPad = 16;    // table words (8 columns each) of padding in front of a row

struct FillArgs2 {
    FillArgs base;
    const uint8_t* a_ids;     // symbol id (< kMaxSym) of every row of A
    const uint32_t* eqtab;    // [kMaxSym][tab_pitch] words, nibble c%8 of word kTabPad + c/8 = score(sym, B[c])
    int tab_pitch;
};

// ---- alphabet + table construction (FASTA-to-device encoding) ----
__global__ void alphabet_presence(const uint8_t* a, int64_t n, unsigned* present /*[256]*/) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) present[a[i]] = 1u;
}
// meta[0] = number of distinct symbols; map[byte] = id (in byte order), inv[id] = byte
__global__ void alphabet_compact(const unsigned* present, unsigned* meta, uint8_t* map, uint8_t* inv) {
    if (threadIdx.x != 0) return;
    unsigned k = 0;
    for (int b = 0; b < 256; ++b) {
        if (present[b]) {
            map[b] = (uint8_t)(k < (unsigned)kMaxSym ? k : 0);
            if (k < (unsigned)kMaxSym) inv[k] = (uint8_t)b;
            ++k;
        }
    }
    meta[0] = k;
}
__global__ void encode_rows(const uint8_t* a, int64_t n, const uint8_t* map, uint8_t* ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = map[a[i]];
}
__global__ void build_eqtab(const uint8_t* b, int W, const unsigned* meta, const uint8_t* inv, uint32_t* tab,
                            int pitch) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;  // word index inside the row
    const unsigned sym = blockIdx.y;
    if (w >= pitch) return;
    uint32_t word = 0x66666666u;
    if (sym < meta[0] && sym < (unsigned)kMaxSym) {
        const int c0 = (w - kTabPad) * 8;
        const unsigned ch = inv[sym];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            if (c >= 0 && c < W && b[c] == ch) word += 2u << (4 * k);
        }
    }
    tab[(size_t)sym * pitch + w] = word;
}

// ---- per-workgroup LDS ----
struct __attribute__((aligned(16))) FillLds {
    int2 ring[kNW][kRing];       // ring[w]: input of wave w; slot s = (D of column s-65, V of column s-64)
    int2 stage[kNW][32];         // staging of a wave's last row on its way to HBM granules
    int dummy[kNW][64 + 48];     // where lanes other than 63 drop their publish writes
    int prod[kNW];               // steps completed by the producer of ring[w]
    int cons[kNW];               // ring[w] slots below this number have been read
};

__device__ __forceinline__ void push_bits6(unsigned& aD, unsigned& aH, unsigned& aV, int x1, int x2, int x3,
                                           int x2p, int x3p, int D, int Hn, int Vn) {
    // acc = 2*acc + bit, first the "table 1 wins" bit then the "table 2 before table 3" bit.
    // Six compares first, six add-with-carry after: every SGPR written by a v_cmp is read at
    // least five instructions later (gfx950 needs two wait states between them).
    unsigned long long s0, s1, s2, s3, s4, s5;
    asm volatile(
        "v_cmp_eq_i32_e64 %[s0], %[x1], %[D]\n\t"
        "v_cmp_ge_i32_e64 %[s1], %[x2], %[x3]\n\t"
        "v_cmp_eq_i32_e64 %[s2], %[x1], %[Hn]\n\t"
        "v_cmp_ge_i32_e64 %[s3], %[x2p], %[x3]\n\t"
        "v_cmp_eq_i32_e64 %[s4], %[x1], %[Vn]\n\t"
        "v_cmp_ge_i32_e64 %[s5], %[x2], %[x3p]\n\t"
        "v_addc_co_u32_e64 %[aD], %[s0], %[aD], %[aD], %[s0]\n\t"
        "v_addc_co_u32_e64 %[aH], %[s2], %[aH], %[aH], %[s2]\n\t"
        "v_addc_co_u32_e64 %[aV], %[s4], %[aV], %[aV], %[s4]\n\t"
        "v_addc_co_u32_e64 %[aD], %[s1], %[aD], %[aD], %[s1]\n\t"
        "v_addc_co_u32_e64 %[aH], %[s3], %[aH], %[aH], %[s3]\n\t"
        "v_addc_co_u32_e64 %[aV], %[s5], %[aV], %[aV], %[s5]"
        : [aD] "+v"(aD), [aH] "+v"(aH), [aV] "+v"(aV), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2),
          [s3] "=&s"(s3), [s4] "=&s"(s4), [s5] "=&s"(s5)
        : [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x2p] "v"(x2p), [x3p] "v"(x3p), [D] "v"(D), [Hn] "v"(Hn),
          [Vn] "v"(Vn));
}

struct StripState {
    int D0, D1;   // D of this row after the last even / odd step
    int H;        // H' of this row (x2 of the next column)
    int V;        // V' of this row at its current column (x3 of the row below)
    unsigned aD, aH, aV;
};

// One anti-diagonal step.  k = step inside the block (compile-time), pre = lane 0's inputs for
// steps 2j, 2j+1 as (D of column-1, V of column) pairs, pa/pb = where this lane drops (D, V).
template <bool CODES, bool PRED>
__device__ __forceinline__ void step16(const int k, StripState& S, const int4& pre, const unsigned mword,
                                       int* pa, int* pb, const bool act, int& x1o,
                                       int& x2o, int& x3o) {
    const int f = (int)__builtin_amdgcn_ubfe(mword, 4 * (k & 7), 4);
    const int preD = (k & 1) ? pre.z : pre.x;
    const int preV = (k & 1) ? pre.w : pre.y;
    const int x1 = shift_down1(preD, (k & 1) ? S.D1 : S.D0) + f;
    const int x3 = shift_down1(preV, S.V);
    const int x2 = S.H;
    const int x2p = x2 + 2, x3p = x3 + 2;
    const int D = max3i(x1, x2, x3);
    const int Hn = max(D, x2p);
    const int Vn = max(D, x3p);
    if (CODES) push_bits6(S.aD, S.aH, S.aV, x1, x2, x3, x2p, x3p, D, Hn, Vn);
    if (PRED) {
        if (act) {
            if (k & 1) S.D1 = D; else S.D0 = D;
            S.H = Hn;
            S.V = Vn;
        }
        x1o = x1; x2o = x2; x3o = x3;
    } else {
        if (k & 1) S.D1 = D; else S.D0 = D;
        S.H = Hn;
        S.V = Vn;
    }
    // publish (committed values): V of this step's column -> slot k+1, D -> slot k+2
    const int Dc = (k & 1) ? S.D1 : S.D0;
    if (k + 1 < kBlk) pa[2 * (k + 1) + 1] = S.V; else pb[2 * (k + 1 - kBlk) + 1] = S.V;
    if (k + 2 < kBlk) pa[2 * (k + 2)] = Dc; else pb[2 * (k + 2 - kBlk)] = Dc;
}

__device__ __forceinline__ bool spin_expired(unsigned& spins, unsigned* abort_flag, unsigned limit = 1u << 25) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) == 0) {
        const unsigned ab = __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ab || spins > limit) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

template <bool CODES>
__global__ __launch_bounds__(kNW * 64) void fill_strips_v2(FillArgs2 q) {
    __shared__ FillLds lds;
    __shared__ unsigned wg_ticket;
    const FillArgs& p = q.base;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) wg_ticket = atomicAdd(p.ticket, 1u);
    if (threadIdx.x < kNW) { lds.prod[threadIdx.x] = 0; lds.cons[threadIdx.x] = 0; }
    __syncthreads();
    const int64_t R = p.r1 - p.r0;
    const int64_t strips = (R + 63) >> 6;
    const int64_t s = (int64_t)wg_ticket * kNW + wv;  // strip inside the chunk
    if (s >= strips) return;
    const int64_t row_base = p.r0 + 64 * s;
    const int64_t sg = row_base >> 6;
    const int64_t i = row_base + lane + 1;
    const bool row_ok = i <= p.r1;
    const int W = p.W;
    const int64_t c0 = p.c0;
    const bool last_in_chunk = (s == strips - 1);
    const bool to_lds = (wv + 1 < kNW) && !last_in_chunk;          // strip s+1 is wave wv+1 of this workgroup
    const bool to_hbm = !to_lds && !(last_in_chunk && p.r1 == p.n);  // someone (maybe a later launch) reads it
    const bool from_lds = wv > 0;
    const bool top_analytic = (sg == 0);

    // ---- left boundary (as v1) ----
    int o1, o2, o3, u1, u2, u3;
    if (p.left) {
        const int64_t y = i - p.r0 - 1;
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
    const int D_own = max3i(o1, o2, o3) + bias_own;
    const int D_up = max3i(u1, u2, u3) + bias_own - 3;
    StripState S;
    S.D0 = S.D1 = D_own;
    S.H = max3i(o1, o2 + 2, o3) + bias_own;
    S.V = 0;
    S.aD = S.aH = S.aV = 0;

    // ---- substitution table row of this lane; window of 16 columns starting at t0 - lane ----
    const int sym = row_ok ? (int)q.a_ids[i - 1] : 0;
    const uint32_t* trow = q.eqtab + (size_t)sym * q.tab_pitch + kTabPad;
    const int msh = 4 * ((64 - lane) & 7);                      // (t0 - lane) mod 8 nibbles, t0 % 16 == 0
    int mword_idx = (-lane - ((64 - lane) & 7)) / 8;            // floor((0 - lane) / 8), exact division
    unsigned mw0 = trow[mword_idx];

    // ---- hand-off plumbing ----
    int2* my_ring = lds.ring[wv];
    int* out_base;      // lane 63: where its (D, V) go; other lanes: a private dump
    int out_wrap;       // ints after which out_base wraps (ring size), 0 = no wrapping (dump)
    if (lane == 63 && to_lds) { out_base = (int*)lds.ring[wv + 1]; out_wrap = kRing * 2; }
    else if (lane == 63 && to_hbm) { out_base = (int*)lds.stage[wv]; out_wrap = 32 * 2; }
    else { out_base = lds.dummy[wv] + lane; out_wrap = 0; }
    const u64 tag_in = (u64)sg, tag_out = (u64)(sg + 1);
    const u64* gin = p.ring + (size_t)(((sg - 1) / kNW) & 1) * p.Wpad;  // bottom row of strip sg-1
    u64* gout = p.ring + (size_t)((sg / kNW) & 1) * p.Wpad;
    int carryD = D_up;          // D of column t0-1 of the row above (wave 0 path)
    u64 gpre = 0;               // prefetched granule of the next block (wave 0 path, lanes 0..15)
    bool gpre_valid = false;

    const int nblocks = p.Q * (64 / kBlk);
    if (p.prof && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.prof[4 * s + 0] = wall_clock64();
        p.prof[4 * s + 2] = ((u64)xcc << 32) | hwid;
    }
    uint4 cD = {0, 0, 0, 0}, cH = {0, 0, 0, 0}, cV = {0, 0, 0, 0};
    const int64_t y_row = i - p.r0 - 1;

    for (int blk = 0; blk < nblocks; ++blk) {
        const int t0 = blk * kBlk;
        // ---- 1. lane 0's inputs for columns t0..t0+15: ring slots t0+64..t0+79 ----
        const int slot0 = (t0 + 64) & (kRing - 1);
        if (from_lds) {
            if (t0 < W) {
                unsigned spins = 0;
                // slots t0+64..t0+79 are complete once the producer has finished step t0+78; it stops at nblocks*16
                const int need = min(t0 + 80, nblocks * kBlk);
                while (__hip_atomic_load(&lds.prod[wv], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                    if (spin_expired(spins, p.abort_flag)) return;
                }
            }
        } else {
            // wave 0: fetch (D, V) of the row above for 16 columns into lanes 0..15, write the ring itself
            int gD = 0, gV = 0;
            if (t0 < W) {
                const int col = t0 + lane;
                const bool col_ok = lane < kBlk && col < W;
                if (top_analytic) {
                    int t1, t2, t3;
                    const int64_t j = c0 + 1 + col;
                    boundary_triple(0, j, t1, t2, t3);
                    gD = max3i(t1, t2, t3) + (int)(3 * j);
                    gV = max3i(t1, t2, t3 + 2) + (int)(3 * j);
                } else {
                    unsigned spins = 0;
                    u64 g = gpre;
                    bool have = gpre_valid;
                    for (;;) {
                        if (!have) g = col_ok ? load_granule(gin + col) : (tag_in << 32);
                        have = false;
                        const bool ok = !col_ok || (g >> 32) == tag_in;
                        if (__all(ok)) break;
                        if (spin_expired(spins, p.abort_flag, 1u << 22)) return;
                    }
                    gD = (int)((unsigned)g >> 2);
                    gV = gD + (int)((unsigned)g & 3u);
                    // prefetch the next block's granules; validated when that block starts
                    const int ncol = col + kBlk;
                    gpre_valid = (t0 + kBlk < W);
                    if (gpre_valid) gpre = (lane < kBlk && ncol < W) ? load_granule(gin + ncol) : (tag_in << 32);
                }
                // slot(col) = (D of col-1, V of col)
                const int Dm1 = shift_down1(carryD, gD);
                carryD = __builtin_amdgcn_readlane(gD, kBlk - 1);
                if (lane < kBlk) my_ring[slot0 + lane] = make_int2(Dm1, gV);
            }
        }
        const int4* rp = (const int4*)(my_ring + slot0);
        int4 pre[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = rp[j];
        if (from_lds && lane == 0)
            __hip_atomic_store(&lds.cons[wv], t0 + 80, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

        // ---- 2. substitution nibbles for this lane's 16 columns ----
        const unsigned mw1 = trow[mword_idx + 1], mw2 = trow[mword_idx + 2];
        const unsigned mlo = __builtin_amdgcn_alignbit(mw1, mw0, msh);
        const unsigned mhi = __builtin_amdgcn_alignbit(mw2, mw1, msh);
        mw0 = mw2;
        mword_idx += 2;

        // ---- 3. room in the consumer's ring (slots up to t0+17 are written by this block) ----
        if (to_lds) {
            unsigned spins = 0;
            while (__hip_atomic_load(&lds.cons[wv + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <
                   t0 + 18 - kRing) {
                if (spin_expired(spins, p.abort_flag)) return;
            }
        }
        int* pa = out_base + (out_wrap ? ((2 * t0) & (out_wrap - 1)) : 0);
        int* pb = out_base + (out_wrap ? ((2 * (t0 + kBlk)) & (out_wrap - 1)) : 0);

        // ---- 4. the sixteen steps ----
        const bool fast = (t0 >= 64) && (t0 + kBlk <= W - 1);
        if (fast) {
            int d0, d1, d2;
#pragma unroll
            for (int k = 0; k < kBlk; ++k)
                step16<CODES, false>(k, S, pre[k >> 1], (k < 8) ? mlo : mhi, pa, pb, true, d0, d1, d2);
        } else {
#pragma unroll
            for (int k = 0; k < kBlk; ++k) {
                const int jj = t0 + k - lane;
                const bool act = row_ok && jj >= 0 && jj < W;
                int x1, x2, x3;
                step16<CODES, true>(k, S, pre[k >> 1], (k < 8) ? mlo : mhi, pa, pb, act, x1, x2, x3);
                if (act && jj == W - 1) {
                    const int bias = (int)(3 * (i + c0 + W));
                    if (p.right) {
                        p.right[y_row] = x1 - bias; p.right[R + y_row] = x2 - bias; p.right[2 * R + y_row] = x3 - bias;
                    }
                    if (i == p.n) {
                        p.final_triple[0] = x1 - bias; p.final_triple[1] = x2 - bias; p.final_triple[2] = x3 - bias;
                    }
                }
            }
        }

        // ---- 5. publish progress / flush granules ----
        if (to_lds) {
            if (lane == 63) __hip_atomic_store(&lds.prod[wv + 1], t0 + kBlk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (to_hbm) {
            // columns t0-64 .. t0-49 were completed by the end of this block (lane 63, step k = column + 63)
            const int cb = t0 - 64;
            if (cb >= 0 && cb < W) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const int col = cb + lane;
                if (lane < kBlk && col < W) {
                    // staging slot of column c is c + 64 (same numbering as the ring): V in .y of slot, D in .x of slot+1
                    const int vs = lds.stage[wv][(col + 64) & 31].y;
                    const int ds = lds.stage[wv][(col + 65) & 31].x;
                    store_granule(gout + col, (tag_out << 32) | ((unsigned)ds << 2) | (unsigned)(vs - ds));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }

        // ---- 6. winner bits: one dword per 16 steps per aggregate, stored 16 bytes per lane every 64 steps ----
        if (CODES) {
            const int d = blk & 3;
            if (d == 0) { cD.x = S.aD; cH.x = S.aH; cV.x = S.aV; }
            else if (d == 1) { cD.y = S.aD; cH.y = S.aH; cV.y = S.aV; }
            else if (d == 2) { cD.z = S.aD; cH.z = S.aH; cV.z = S.aV; }
            else {
                cD.w = S.aD; cH.w = S.aH; cV.w = S.aV;
                uint4* dst = (uint4*)(p.codes + code_word_index(sg, p.Q, blk >> 2, 0, lane, 0));
                dst[0] = cD; dst[64] = cH; dst[128] = cV;
            }
        }
    }
    if (p.prof && lane == 0) p.prof[4 * s + 1] = wall_clock64();
    // ---- trailing granule group (columns whose flush block lies beyond the last block) ----
    if (to_hbm) {
        for (int cb = nblocks * kBlk - 64; cb < W; cb += kBlk) {
            if (cb < 0) continue;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int col = cb + lane;
            if (lane < kBlk && col < W) {
                const int vs = lds.stage[wv][(col + 64) & 31].y;
                const int ds = lds.stage[wv][(col + 65) & 31].x;
                store_granule(gout + col, (tag_out << 32) | ((unsigned)ds << 2) | (unsigned)(vs - ds));
            }
        }
    }
}

}  // namespace psa
