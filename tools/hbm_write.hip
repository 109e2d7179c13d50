// hbm_write.hip -- what the device takes in plain 16-byte-per-lane stores (1 KB per wavefront instruction, consecutive
// tiles of 4 KB per wavefront as the fill kernels write their winner codes), W wavefronts per SIMD, with and without
// VALU work between the stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int VALU>
__global__ __launch_bounds__(256) void k(uint4* out, long long tiles_per_wave, int seed) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4* p = out + wave * tiles_per_wave * 256 + lane;
    unsigned a = seed + lane, b = a * 3, c = a * 5, d = a * 7;
    for (long long t = 0; t < tiles_per_wave; ++t) {
#pragma unroll
        for (int r = 0; r < VALU; ++r) {
            asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_max_i16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
        p[0] = make_uint4(a, b, c, d);
        p[64] = make_uint4(b, c, d, a);
        p[128] = make_uint4(c, d, a, b);
        p[192] = make_uint4(d, a, b, c);
        p += 256;
    }
}
template <int VALU>
void run(uint4* buf, size_t bytes, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4, 8}) {
        const long long waves = (long long)cus * 4 * wps;
        const long long tiles = (long long)(bytes / 4096) / waves;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<VALU>, dim3(cus * wps), dim3(256), 0, 0, buf, tiles, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  valu %3d per tile  %d waves/SIMD: %.2f ms  %.0f GB/s\n", 4 * VALU, wps, ms, (double)waves * tiles * 4096 / ms * 1e-6);
        fflush(stdout);
    }
}
int main() {
    const size_t bytes = 32ull << 30;
    uint4* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipMemsetAsync(buf, 1, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("hipMemsetAsync 32 GiB: %.2f ms  %.0f GB/s\n", ms, bytes / ms * 1e-6);
    run<0>(buf, bytes, cus);
    run<64>(buf, bytes, cus);    // 256 VALU per 4 KB tile: a quarter of the packed kernel's work per byte
    run<256>(buf, bytes, cus);   // 1024 VALU per 4 KB: the packed kernel's ratio (2 x 4 KB per 64 steps of ~16)
    return 0;
}
