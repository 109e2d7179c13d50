// store_geom.hip -- the packed fill's store pattern under a plain VALU stream: 10 240 wavefronts (3 per SIMD resident), each
// writing the 1 025 tiles of "its strip" of two pairs (2 x 4 KB per 64 "steps" of 15 VALU instructions), one KB per
// store, nontemporal or plain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT, bool STORES, int KEEP = 63>
__global__ __launch_bounds__(256) void k(uint8_t* bufx, uint8_t* bufy, int Q, int seed) {
    const int lane = threadIdx.x & 63;
    const long long strip = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pair = (int)(strip / 1024), sg = (int)(strip % 1024);
    uint8_t* px = bufx + ((size_t)pair * 1024 + sg) * (size_t)Q * 4096 + 16 * lane;
    uint8_t* py = bufy + ((size_t)pair * 1024 + sg) * (size_t)Q * 4096 + 16 * lane;
    unsigned a = seed + lane, b = a * 3, c = a * 5, d = a * 7;
    for (int sb = 0; sb < Q; ++sb) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
            for (int r = 0; r < 60; ++r)   // 240 VALU per block of 16 steps
                asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_max_i16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (STORES) {
                const v4u v = {a, b, c, d}, w = {b, c, d, a};
                v4u* qx = (v4u*)(px + (size_t)sb * 4096 + jb * 1024);
                v4u* qy = (v4u*)(py + (size_t)sb * 4096 + jb * 1024);
                if (NT) { __builtin_nontemporal_store(v, qx); __builtin_nontemporal_store(w, qy); } else { *qx = v; *qy = w; }
                // (KEEP < 63: wait until at most KEEP stores are in flight, as a wait for a load issued KEEP stores ago would)
                if (KEEP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if (KEEP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) bufx[0] = 1;
}
template <bool NT, bool STORES, int KEEP = 63>
void run(uint8_t* bx, uint8_t* by, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int Q = 1025;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NT, STORES, KEEP>), dim3(2560), dim3(256), 0, 0, bx, by, Q, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s %7.2f ms  %.0f GB/s\n", what, ms, STORES ? 10240.0 * Q * 8192 / ms * 1e-6 : 0.0);
    fflush(stdout);
}
int main() {
    const size_t bytes = (size_t)10 * 1024 * 1025 * 4096 + (1 << 20);
    uint8_t *bx, *by;
    if (hipMalloc(&bx, bytes) != hipSuccess || hipMalloc(&by, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<false, false>(bx, by, "no stores");
    run<false, true>(bx, by, "plain stores");
    run<true, true>(bx, by, "nontemporal stores");
    run<false, false>(bx, by, "no stores");
    run<true, true, 2>(bx, by, "nt, at most 2 in flight");
    run<true, true, 4>(bx, by, "nt, at most 4 in flight");
    run<true, true, 8>(bx, by, "nt, at most 8 in flight");
    run<true, true, 16>(bx, by, "nt, at most 16 in flight");
    run<false, true, 4>(bx, by, "plain, at most 4 in flight");
    run<false, true, 8>(bx, by, "plain, at most 8 in flight");
    run<false, true, 16>(bx, by, "plain, at most 16 in flight");
    return 0;
}
