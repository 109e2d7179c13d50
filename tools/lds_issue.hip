// lds_issue.hip -- what an LDS instruction costs the wave that issues it (issue slots, not latency):
// a loop of 6 independent VALU instructions + one DS instruction of each kind, 1..2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define KINDS(X) \
  X(0, "", "no LDS instruction") \
  X(1, "ds_write_b32 %8, %0", "ds_write_b32") \
  X(2, "ds_write2_b32 %8, %0, %1 offset0:3 offset1:4", "ds_write2_b32") \
  X(3, "ds_write_b64 %8, %10", "ds_write_b64") \
  X(4, "ds_write_b128 %8, %11", "ds_write_b128") \
  X(5, "ds_write2_b64 %8, %10, %12 offset0:1 offset1:2", "ds_write2_b64") \
  X(6, "ds_read_b32 %9, %8", "ds_read_b32 (never waited for)") \
  X(7, "ds_read_b128 %13, %8", "ds_read_b128 (never waited for)") \
  X(8, "ds_write_b8 %8, %0", "ds_write_b8") \
  X(9, "ds_write_b16 %8, %0", "ds_write_b16")
template <int KIND, bool ONE_LANE, bool MASKED = false>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    __shared__ int dump[4][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v0 = seed + lane, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, w = seed + 7;
    int addr = (int)(size_t)(dump[wv] + 4 * lane);   // LDS byte address: low 32 bits of the generic pointer's offset
    constexpr int stride = (KIND == 3) ? 2 : (KIND == 4 || KIND == 7) ? 4 : (KIND == 5) ? 2 : 1;   // dwords per lane: conflict-free
    addr = (int)((size_t)(dump[wv] + stride * (ONE_LANE ? 0 : lane)) & 0xffff);
    long long p64 = seed, q64 = seed + 1;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i p128 = {seed, 1, 2, 3}, r128 = {0, 0, 0, 0};
    int r32 = 0;
    const unsigned long long lane63 = 1ull << 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define ONE(id, txt, name)                                                                                                    \
    if (KIND == id && MASKED)                                                                                                 \
        asm volatile("v_add_u32_e32 %0, %7, %0\n v_max_i32_e32 %1, %7, %1\n v_add_u32_e32 %2, %7, %2\n s_mov_b64 exec, %14\n" txt \
                     "\n s_mov_b64 exec, -1\n v_max_i32_e32 %3, %7, %3\n v_add_u32_e32 %4, %7, %4\n v_max_i32_e32 %5, %7, %5" \
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(r32)                                  \
                     : "v"(w), "v"(addr), "v"(r32), "v"(p64), "v"(p128), "v"(q64), "v"(r128), "s"(lane63)                     \
                     : "memory");                                                                                             \
    if (KIND == id && !MASKED)                                                                                                \
        asm volatile("v_add_u32_e32 %0, %7, %0\n v_max_i32_e32 %1, %7, %1\n v_add_u32_e32 %2, %7, %2\n" txt                  \
                     "\n v_max_i32_e32 %3, %7, %3\n v_add_u32_e32 %4, %7, %4\n v_max_i32_e32 %5, %7, %5"                      \
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(r32)                                  \
                     : "v"(w), "v"(addr), "v"(r32), "v"(p64), "v"(p128), "v"(q64), "v"(r128)                                  \
                     : "memory");
            KINDS(ONE)
#undef ONE
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5) == 0x7fffffff) out[0] = r32 + r128.x;
}
template <int KIND, bool ONE_LANE = false, bool MASKED = false>
void run(const char* name, int cus, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-34s", name);
    for (int wps : {1, 2, 3}) {
        dim3 grid(cus * wps), block(256);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<KIND, ONE_LANE, MASKED>), grid, block, 0, 0, out, iters, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  %dw %.1f cyc/group", wps, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8));
    }
    printf("\n");
}
int main() {
    int* out;
    hipMalloc(&out, 1 << 20);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
#define RUN(id, txt, name) run<id>(name, cus, out);
    run<0>("warm-up", cus, out);
    KINDS(RUN)
    printf("all lanes on one address:\n");
#define RUN1(id, txt, name) run<id, true>(name, cus, out);
    KINDS(RUN1)
    printf("exec = lane 63 only around the LDS instruction (two s_mov_b64 more per group):\n");
#define RUNM(id, txt, name) run<id, false, true>(name, cus, out);
    KINDS(RUNM)
    return 0;
}
