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

// Packed winner codes: per strip (64 rows), per block of 64 wavefront steps ("tile"), three parts,
// per lane, four dwords (48 bytes per lane and tile = 0.75 byte per cell):
//   part 0, 1: eight words of eight 4-bit steps; word u>>3 holds step u (0..63) in bits 4*(7-(u&7))..+3
//              as nibble 2*(q'+3) + b1_D, q' = clamp(x2 - x3, -3, 2), b1_D = (x1 == D), AFTER
//              kCodeBias has been added to the stored word (the writer accumulates signed q')
//   part 2:    dwords 0,1: b1_H = (x1 == H') of steps 0..31 / 32..63, step u in bit 31-(u&31);
//              dwords 2,3: b1_V = (x1 == V') likewise
// The three "table 2 before table 3" bits follow from q' = x2 - x3 clamped:
//   D: x2 >= x3  <=> q' >= 0;   H': x2 + 2 >= x3  <=> q' >= -2;   V': x2 >= x3 + 2  <=> q' >= 2.
constexpr uint32_t kCodeBias = 0x66666666u;

__device__ __host__ __forceinline__ int64_t code_word_index(int64_t strip, int64_t qblocks, int64_t q,
                                                            int part, int lane, int d) {
    return ((((strip * qblocks + q) * 3 + part) * 64 + lane) * 4) + d;
}
// word offsets inside one tile ([part][lane][4] dwords) of step u's nibble word and b1_H / b1_V word
__device__ __host__ __forceinline__ int code_q_offset(int lane, int u) { return ((u >> 5) * 64 + lane) * 4 + ((u >> 3) & 3); }
__device__ __host__ __forceinline__ int code_hv_offset(int lane, int u, int kind) {
    return (2 * 64 + lane) * 4 + (kind == 3 ? 2 : 0) + (u >> 5);
}
// winner (1, 2 or 3) for aggregate `kind` (1 = D, 2 = H', 3 = V') from step u's words
__device__ __host__ __forceinline__ int code_winner(uint32_t qword, uint32_t hvword, int u, int kind) {
    const unsigned nib = ((qword + kCodeBias) >> (4 * (7 - (u & 7)))) & 15u;
    const unsigned q3 = nib >> 1;   // q' + 3, 0..5
    bool b1, b23;
    if (kind == 1) { b1 = nib & 1u; b23 = q3 >= 3; }
    else {
        b1 = (hvword >> (31 - (u & 31))) & 1u;
        b23 = (kind == 2) ? (q3 >= 1) : (q3 >= 5);
    }
    return b1 ? 1 : (b23 ? 2 : 3);
}

}  // namespace psa
