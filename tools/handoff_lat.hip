// handoff_lat.hip -- diagnostic: visibility latency of an 8-byte write-through (sc1) granule
// between two workgroups, polled with sc1 loads.  Producer stamps wall_clock64 into the value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;

__global__ void k(u64* slots, u64* out, int rounds, int prod_block, int cons_block, int prepoll) {
    if (threadIdx.x != 0) return;
    if ((int)blockIdx.x == prod_block) {
        for (int r = 0; r < rounds; ++r) {
            u64 t = wall_clock64();
            while (wall_clock64() - t < 300) {}  // 3 us between publications
            u64 now = wall_clock64();
            __hip_atomic_store(&slots[r * 16], (now << 8) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if ((int)blockIdx.x == cons_block) {
        for (int r = 0; r < rounds; ++r) {
            u64 v;
            int polls = 0;
            do { v = __hip_atomic_load(&slots[r * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ++polls;
                 if (!prepoll && !(v & 1)) __builtin_amdgcn_s_sleep(2); } while (!(v & 1) && polls < 10000000);
            u64 now = wall_clock64();
            out[2 * r] = now - (v >> 8);
            out[2 * r + 1] = polls;
        }
    }
}

int main() {
    const int rounds = 200;
    u64 *slots, *out;
    hipMalloc(&slots, rounds * 16 * 8);
    hipMalloc(&out, rounds * 16);
    for (int cons : {1, 2, 8, 9, 17, 100}) {
        hipMemset(slots, 0, rounds * 16 * 8);
        hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, slots, out, rounds, 0, cons, 1);
        hipDeviceSynchronize();
        std::vector<u64> h(2 * rounds);
        hipMemcpy(h.data(), out, rounds * 16, hipMemcpyDeviceToHost);
        std::vector<u64> lat, polls;
        for (int r = 10; r < rounds; ++r) { lat.push_back(h[2 * r]); polls.push_back(h[2 * r + 1]); }
        std::sort(lat.begin(), lat.end()); std::sort(polls.begin(), polls.end());
        printf("consumer block %3d: latency ticks(10ns) min %llu med %llu max %llu ; polls med %llu\n", cons, lat[0],
               lat[lat.size() / 2], lat.back(), polls[polls.size() / 2]);
    }
    return 0;
}
