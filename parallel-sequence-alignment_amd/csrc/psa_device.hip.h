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
// are recorded with the source cell; the traceback of a cell in table k reads the bits of its
// predecessor cell for aggregate k.
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

// Packed winner bits: per strip (64 rows), per block of 64 wavefront steps, per aggregate,
// per lane, four dwords; dword d holds steps 16d..16d+15, step u in bits 31-2u (b1), 30-2u (b23).
__device__ __host__ __forceinline__ int64_t code_word_index(int64_t strip, int64_t qblocks, int64_t q,
                                                            int kind0, int lane, int d) {
    return ((((strip * qblocks + q) * 3 + kind0) * 64 + lane) * 4) + d;
}

}  // namespace psa
