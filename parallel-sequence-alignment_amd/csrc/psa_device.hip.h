// psa_device.hip.h -- device-side building blocks shared by the gfx950 kernels.
//
// Formulation: the reference's three tables T1 (ends in a pair), T2 (gap in
// A, horizontal) and T3 (gap in B, vertical) of simple.cpp:157-213 are kept in a biased
// domain x_k(i,j) = T_k(i,j) + 3*(i+j).  For every cell three aggregates feed its neighbours:
//     D  = max3(x1, x2,   x3  )   -> T1 of (i+1, j+1):  x1' = D + (A==B ? 8 : 6)
//     H' = max3(x1, x2+2, x3  )   -> T2 of (i,   j+1):  x2' = H'   ( = max(D, x2+2) )
//     V' = max3(x1, x2,   x3+2)   -> T3 of (i+1, j  ):  x3' = V'   ( = max(D, x3+2) )
// and the reference's tie-break "T1 if t1>=t2 && t1>=t3, else T2 if t2>=t3, else T3"
// (simple.cpp:162/166/170 ...) is a property of the SOURCE cell: for each aggregate two bits
//     b1  = (x1 == aggregate)            table 1 wins
//     b23 = (x2(+2) >= x3(+2))           else table 2 before table 3
// are recorded with the source cell (the three b23 as one clamped difference, see "packed winner
// codes" below); the traceback of a cell in table k reads the bits of its predecessor cell for
// aggregate k.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psa {

typedef unsigned long long u64;

constexpr int kWave = 64;
constexpr int kSubMismatch = 6;  // 0 + 2*3 (a diagonal step raises the bias by 6)
constexpr int kSubMatch = 8;     // 2 + 2*3

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __host__ __forceinline__ int max3i_hd(int a, int b, int c) { const int m = a > b ? a : b; return m > c ? m : c; }

// DPP wave_shr:1 -- lane l receives src of lane l-1; lane 0 keeps `old`.
__device__ __forceinline__ int shift_down1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);
}

// Boundary values of the reference, plain domain (simple.cpp:116-133, simple.h:50).
__device__ __host__ __forceinline__ void boundary_triple(int64_t i, int64_t j, int& t1, int& t2, int& t3) {
    if (i == 0 && j == 0) { t1 = 0; t2 = -2; t3 = -1; }
    else if (i == 0) { t1 = -1; t2 = 0; t3 = (int)-(2 + j); }
    else { t1 = -1; t2 = (int)-(2 + i); t3 = -1; }
}

// Winner among a source cell's three values for aggregate kind (1 = D, 2 = H', 3 = V'),
// plain-domain inputs; used for boundary cells whose bits are not stored.
__device__ __host__ __forceinline__ int winner_plain(int t1, int t2, int t3, int kind) {
    if (kind == 2) { t1 -= 3; t2 -= 1; t3 -= 3; }
    else if (kind == 3) { t1 -= 3; t2 -= 3; t3 -= 1; }
    if (t1 >= t2 && t1 >= t3) return 1;
    return t2 >= t3 ? 2 : 3;
}

// Winner codes, one byte per cell (the "traceback stream", 1 B per cell):
//     byte = ((D & 15) << 4) | ((H' - D) << 2) | (V' - D)          H' - D and V' - D are in {0, 1, 2}
// which is (11*D + 4*H' + V') & 0xff: two instructions (v_lshl_add_u32, v_mad_u32_u24) and one
// v_alignbit_b32 to shift the byte into the word being collected.  The traceback of a cell p needs
// the three values the reference compared there (simple.cpp:162/166/170),
//     x1(p) = D(diag) + (6 | 8),   x2(p) = H'(left),   x3(p) = V'(up)      (biased domain)
// only relative to each other.  All three lie within 8 below D(p) (D - x2 <= (D(p) - D(left)) + 3,
// horizontal and vertical neighbours of D differ by -3..+5 in the plain domain, diagonal ones by
// 0..2: the bounds of the Gotoh recurrence, checked exhaustively in tests/test_oracle.py), so the
// low four bits of D of p's three neighbours and of p itself decide every comparison:
//     e_k = (D(p) - x_k(p)) & 15     in 0..8, and min(e1, e2, e3) == 0.
// Storage: per strip (64 rows), per block of 64 wavefront steps one 4 KB tile, [part 0..3][lane][16 bytes]:
// step u of lane l sits at byte ((u >> 4) * 64 + l) * 16 + (u & 15); a wavefront stores 4 x 1 KB per
// 64 steps, 16 bytes per lane, fully coalesced although the lanes sit on different columns.
constexpr int kTileBytes = 4096;

__device__ __host__ __forceinline__ int64_t code_tile_offset(int64_t strip, int64_t qblocks, int64_t q) {
    return (strip * qblocks + q) * (int64_t)kTileBytes;
}
__device__ __host__ __forceinline__ int code_byte_offset(int lane, int u) {
    return (((u >> 4) * 64 + lane) << 4) + (u & 15);
}
__device__ __host__ __forceinline__ unsigned code_byte(int D, int Hn, int Vn) {
    return (unsigned)(11 * D + 4 * Hn + Vn) & 0xffu;
}
// Winner (1, 2 or 3) for aggregate `kind` (1 = D, 2 = H', 3 = V') at a cell whose own D and whose three
// inputs are known modulo 16.  *bad is set when the values cannot belong to one cell (decode check).
__device__ __host__ __forceinline__ int winner_mod16(int dp, int x1, int x2, int x3, int kind, int* bad) {
    int e1 = (dp - x1) & 15, e2 = (dp - x2) & 15, e3 = (dp - x3) & 15;
    if (bad && (e1 > 2 || e2 > 8 || e3 > 8 || (e1 && e2 && e3))) *bad = 1;
    if (kind == 2) e2 -= 2; else if (kind == 3) e3 -= 2;
    if (e1 <= e2 && e1 <= e3) return 1;   // x1 >= x2' && x1 >= x3'
    return e2 <= e3 ? 2 : 3;
}
// Biased boundary values of the reference (row 0 / column 0; psa_device boundary_triple + 3*(i+j)):
// D, H' of column 0 and D, V' of row 0.
__device__ __host__ __forceinline__ int boundary_D(int64_t i, int64_t j) {
    int t1, t2, t3;
    boundary_triple(i, j, t1, t2, t3);
    return max3i_hd(t1, t2, t3) + (int)(3 * (i + j));
}
__device__ __host__ __forceinline__ int boundary_Hn(int64_t i, int64_t j) {
    int t1, t2, t3;
    boundary_triple(i, j, t1, t2, t3);
    return max3i_hd(t1, t2 + 2, t3) + (int)(3 * (i + j));
}
__device__ __host__ __forceinline__ int boundary_Vn(int64_t i, int64_t j) {
    int t1, t2, t3;
    boundary_triple(i, j, t1, t2, t3);
    return max3i_hd(t1, t2, t3 + 2) + (int)(3 * (i + j));
}

}  // namespace psa
