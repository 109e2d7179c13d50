// valu_peak.hip -- calibrates the int32 VALU roofline on the box:
// independent v_max3_i32 / v_add_u32 chains, W waves per SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ void k_peak(int* out, int iters, int seed) {
    int v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x + c + seed;
    int w = seed + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                v[c] = max(max(v[c], w), v[(c + 1) % CHAINS]);  // v_max3_i32
                v[c] = v[c] + 1;                                 // v_add_u32
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s ^= v[c];
    if (s == 0x7fffffff) out[0] = s;
}

// dependent DPP wave_shr chain, for the cross-lane latency
__global__ void k_dpp(int* out, int iters, int seed) {
    int v = threadIdx.x + seed, a = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int s = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
            v = max(s + 1, a);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    int* out;
    hipMalloc(&out, 1 << 24);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 4, 8}) {
        const int waves_per_cu = 4 * wps;
        dim3 grid(pr.multiProcessorCount * waves_per_cu / 4), block(256);  // 4 waves per block
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_peak<8>, grid, block, 0, 0, out, iters, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)grid.x * 256 * iters * 8.0 * 8 * 2;
            if (rep) printf("waves/SIMD %d: %.3f ms  %.2f Tops/s int32 (max3+add)\n", wps, ms, ops / ms / 1e9);
        }
    }
    {
        dim3 grid(pr.multiProcessorCount), block(256);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<1>, grid, block, 0, 0, out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("dependent chain, 1 wave/SIMD: %.2f ns per instruction\n", ms * 1e6 / (iters * 8.0 * 2));
    }
    {
        dim3 grid(pr.multiProcessorCount), block(256);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dpp, grid, block, 0, 0, out, iters, 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("dependent dpp(wave_shr)+add+max chain, 1 wave/SIMD: %.2f ns per step of 3 instr\n", ms * 1e6 / (iters * 16.0));
        int h[64];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        printf("dpp lanes: %d %d %d ... %d %d (expect lane l -> l-dependent values, shift crosses rows 15->16: %d %d)\n", h[0], h[1], h[2], h[62], h[63], h[15], h[16]);
    }
    return 0;
}
