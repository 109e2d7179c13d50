// store_cost.hip -- what 16-byte-per-lane stores cost a VALU-bound wavefront: 1 024 packed VALU instructions per round
// and S stores (1 KB each) spread over the round, 3 wavefronts per SIMD on every CU; plain or nontemporal; the stored
// registers rewritten right after the store or left alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int S, bool NT, bool REWRITE>
__global__ __launch_bounds__(256) void k(v4u* out, long long rounds, int seed) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    v4u* p = out + wave * rounds * 64 * (S ? S : 1) + lane;
    unsigned a = seed + lane, b = a * 3, c = a * 5, d = a * 7;
    unsigned e = a ^ 1, f = b ^ 2, g = c ^ 3, h = d ^ 4;   // the stored registers when REWRITE is false
    for (long long t = 0; t < rounds; ++t) {
#pragma unroll
        for (int s = 0; s < (S ? S : 1); ++s) {
#pragma unroll
            for (int r = 0; r < 256 / (S ? S : 1); ++r)
                asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_max_i16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (S) {
                const v4u v = REWRITE ? (v4u){a, b, c, d} : (v4u){e, f, g, h};
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                p += 64;
            }
        }
    }
    if ((a ^ b ^ c ^ d) == 0x12345678u) out[0] = (v4u){e, f, g, h};
}
template <int S, bool NT, bool REWRITE>
void run(v4u* buf, int cus, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wps = 3;
    const long long rounds = 2048;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<S, NT, REWRITE>), dim3(cus * wps), dim3(256), 0, 0, buf, rounds, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double valu_ms = (double)rounds * 1024 * 4 * wps / 2.4e9 * 1e3;
    printf("%-44s %7.2f ms  (VALU alone %.2f)  %.0f GB/s\n", what, ms, valu_ms, (double)cus * 4 * wps * rounds * S * 1024 / ms * 1e-6);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    v4u* buf;
    const size_t bytes = (size_t)cus * 4 * 3 * 2048 * 16 * 1024 + (1 << 20);
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0, false, true>(buf, cus, "no stores");
    run<4, false, true>(buf, cus, "4 stores / 1024 VALU, plain, rewritten");
    run<8, false, true>(buf, cus, "8 stores / 1024 VALU, plain, rewritten");
    run<8, true, true>(buf, cus, "8 stores / 1024 VALU, nontemporal, rewritten");
    run<8, false, false>(buf, cus, "8 stores / 1024 VALU, plain, registers kept");
    run<8, true, false>(buf, cus, "8 stores / 1024 VALU, nontemporal, kept");
    run<16, true, true>(buf, cus, "16 stores / 1024 VALU, nontemporal, rewritten");
    return 0;
}
