// step_mix.hip -- the fill step's own instruction mix (step16 of psa_fill_v2.hip.h), no hand-off, no memory:
// how fast can a SIMD run it with 1, 2, 3, 4 waves?
#include "../parallel-sequence-alignment_amd/csrc/psa_fill_v2.hip.h"
#include <cstdio>
using namespace psa;
template <bool CODES, int PUBLISH, bool KV = false>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    __shared__ __attribute__((aligned(16))) int dump[4][512];
    StripState S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    S.D0 = seed + lane; S.D1 = seed - lane; S.H = seed + 1; S.V = seed + 2; S.acc = 0; S.r1 = S.r2 = S.r3 = 0;
    int4 pre[8];
    for (int j = 0; j < 8; ++j) pre[j] = make_int4(seed + j, seed - j, seed + 2 * j, seed - 2 * j);
    unsigned mq[4] = {0x06080606u + seed, 0x08060608u, 0x06060806u, 0x08080606u};
    int* pa = dump[wv] + 2 * lane;
    int sink[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned q = 0;
    int Dm2 = seed, Dm1 = seed + 1, Vm1 = seed + 2;
    int* pq = dump[wv] + 4 * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (PUBLISH == 1) step16<CODES, kFast>(k, S, pre[k >> 1], mq[k >> 2], pq, 64 + k, lane, 1 << 16, true);
            else {
                // the same arithmetic without the LDS writes
                const int preD = (k & 1) ? pre[k >> 1].z : pre[k >> 1].x;
                const int preV = (k & 1) ? pre[k >> 1].w : pre[k >> 1].y;
                const int x1 = add_byte(shift_down1(preD, (k & 1) ? S.D1 : S.D0), mq[k >> 2], k & 3);
                const int x3 = shift_down1(preV, S.V);
                const int x2 = S.H;
                const int D = max3i(x1, x2, x3);
                const int Hn = max(D, x2 + 2), Vn = max(D, x3 + 2);
                if (CODES) push_code(S.acc, D, Hn, Vn);
                if (k & 1) S.D1 = D; else S.D0 = D;
                S.H = Hn; S.V = Vn;
                if (PUBLISH == 2) {   // what the kernel did before: V of this step and D of this step, 8 bytes a step
                    pa[2 * (k + 1) + 1] = Vn;
                    pa[2 * (k + 2)] = D;
                }
                if (PUBLISH == 3) {   // as two 8-byte halves of one instruction
                    if ((k & 1) == 0) {
                        long long lo = ((long long)(unsigned)Vm1 << 32) | (unsigned)Dm2, hi = ((long long)(unsigned)Vn << 32) | (unsigned)Dm1;
                        asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"((int)(size_t)pq & 0xffff), "v"(lo), "v"(hi), "n"(k), "n"(k + 1) : "memory");
                    }
                    Dm2 = Dm1; Dm1 = D; Vm1 = Vn;
                }
            }
            if (CODES && (k & 3) == 3) q ^= S.acc;
        }
        // (keep the pre registers alive and changing, as the loads from the ring do)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(pre[j].x), "+v"(pre[j].y), "+v"(pre[j].z), "+v"(pre[j].w));
    }
    if ((S.D0 ^ S.D1 ^ S.H ^ S.V ^ (int)q) == 0x7fffffff) out[0] = 1;
}
template <bool CODES, int PUBLISH, bool KV = false>
void run(const char* name, int cus, int* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-40s", name);
    for (int wps : {1, 2, 3, 4}) {
        dim3 grid(cus * wps), block(256);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<CODES, PUBLISH, KV>), grid, block, 0, 0, out, iters, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 16);   // SIMD cycles per wave-step
        printf("  %dw %.1f cyc/step", wps, cyc);
    }
    printf("\n");
}
int main() {
    int* out;
    hipMalloc(&out, 1 << 20);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    run<true, 0>("warm-up", cus, out);
    run<true, 0>("codes, arithmetic only", cus, out);
    run<true, 1>("codes, step16 (16 B every 2nd step)", cus, out);
    run<true, 2>("codes, 8 bytes every step", cus, out);
    run<true, 3>("codes, ds_write2_b64 every 2nd step", cus, out);
    run<false, 0>("score only, arithmetic only", cus, out);
    run<false, 1>("score only, step16 (16 B every 2nd)", cus, out);
    run<false, 2>("score only, 8 bytes every step", cus, out);
    run<false, 3>("score only, ds_write2_b64 every 2nd", cus, out);
    return 0;
}
