// lds_tp.hip -- LDS throughput per CU of each write/read kind with nearly no VALU beside it (2 VALU per DS instruction),
// 1..4 waves per SIMD: cycles of CU time per DS instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define KINDS(X) \
  X(1, "ds_write_b32 %2, %0", "ds_write_b32") \
  X(3, "ds_write_b64 %2, %3", "ds_write_b64") \
  X(4, "ds_write_b128 %2, %4", "ds_write_b128") \
  X(6, "ds_read_b32 %5, %2", "ds_read_b32") \
  X(7, "ds_read_b64 %6, %2", "ds_read_b64") \
  X(8, "ds_read_b128 %7, %2", "ds_read_b128")
template <int KIND, int MODE>   // MODE 0: every lane its own address; 1: exec = lane 63 only; 2: every lane the same address
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    __shared__ int dump[4][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v0 = seed + lane, v1 = v0 + 1, w = seed + 7;
    constexpr int stride = (KIND == 3 || KIND == 7) ? 2 : (KIND == 4 || KIND == 8) ? 4 : 1;
    int addr = (int)((size_t)(dump[wv] + stride * (MODE == 2 ? 0 : lane)) & 0xffff);
    long long p64 = seed, r64 = 0;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i p128 = {seed, 1, 2, 3}, r128 = {0, 0, 0, 0};
    int r32 = 0;
    const unsigned long long lane63 = 1ull << 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define ONE(id, txt, name)                                                                                     \
    if (KIND == id && MODE == 1)                                                                               \
        asm volatile("v_add_u32_e32 %0, %1, %0\n s_mov_b64 exec, %8\n" txt "\n s_mov_b64 exec, -1\n"         \
                     : "+v"(v0) : "v"(w), "v"(addr), "v"(p64), "v"(p128), "v"(r32), "v"(r64), "v"(r128), "s"(lane63) : "memory"); \
    if (KIND == id && MODE != 1)                                                                               \
        asm volatile("v_add_u32_e32 %0, %1, %0\n" txt "\n"                                                    \
                     : "+v"(v0) : "v"(w), "v"(addr), "v"(p64), "v"(p128), "v"(r32), "v"(r64), "v"(r128) : "memory");
            KINDS(ONE)
#undef ONE
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if ((v0 ^ v1) == 0x7fffffff) out[0] = r32 + r128.x + (int)r64;
}
template <int KIND, int MODE>
void run(const char* name, int cus, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-16s mode %d", name, MODE);
    for (int wps : {1, 2, 3, 4}) {
        dim3 grid(cus * wps), block(256);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<KIND, MODE>), grid, block, 0, 0, out, iters, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // CU cycles per DS instruction: the CU's 4 x wps waves issue 4 x wps x iters x 8 of them
        printf("  %dw %.2f", wps, ms * 1e-3 * 2.4e9 / ((double)4 * wps * iters * 8));
    }
    printf("   (CU cycles per instruction)\n");
}
int main() {
    int* out;
    hipMalloc(&out, 1 << 20);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    run<1, 0>("warm-up", cus, out);
#define RUN(id, txt, name) run<id, 0>(name, cus, out); run<id, 1>(name, cus, out); run<id, 2>(name, cus, out);
    KINDS(RUN)
    return 0;
}
