// pk_rate.hip -- issue rate of the packed 16-bit integer VALU instructions (v_pk_add_u16, v_pk_max_i16, v_pk_min_u16, v_pk_mad_u16; v_perm_b32)
// against their 32-bit counterparts on the box: independent chains, 4 waves per SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(int* out, int iters, int seed) {
    int v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = threadIdx.x + c + seed;
    const int w = seed * 0x10001 + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) { v[c] = max(v[c], w); v[c] = v[c] + 1; }
                if (MODE == 1) { v[c] = __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(pk16, v[c]), __builtin_bit_cast(pk16, w)));
                                 v[c] = __builtin_bit_cast(int, __builtin_bit_cast(pk16, v[c]) + __builtin_bit_cast(pk16, 0x00010001)); }
                if (MODE == 2) { v[c] = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(upk16, v[c]), __builtin_bit_cast(upk16, w)));
                                 v[c] = __builtin_bit_cast(int, __builtin_bit_cast(upk16, v[c]) - __builtin_bit_cast(upk16, 0x00010001)); }
                if (MODE == 3) { v[c] = __builtin_amdgcn_update_dpp(v[c], v[(c + 1) & 7], 0x138, 0xf, 0xf, false);
                                 v[c] = __builtin_bit_cast(int, __builtin_bit_cast(pk16, v[c]) + __builtin_bit_cast(pk16, 0x00010001)); }
                if (MODE == 5) { int r_; asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r_) : "v"(v[c]), "s"(w), "v"(v[(c + 1) & 7])); v[c] = r_;
                                 asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r_) : "v"(v[c]), "s"(w), "v"(v[(c + 3) & 7])); v[c] = r_; }
                if (MODE == 6) { v[c] = (int)__builtin_amdgcn_perm((unsigned)v[c], (unsigned)v[(c + 1) & 7], 0x06020400u);
                                 v[c] = (int)__builtin_amdgcn_perm((unsigned)v[c], (unsigned)v[(c + 3) & 7], 0x05040100u); }
                if (MODE == 7) { v[c] = (int)__builtin_amdgcn_alignbit((unsigned)v[c], (unsigned)v[(c + 1) & 7], w & 24);
                                 int r_; asm volatile("v_mad_u32_u24 %0, %1, 11, %2" : "=v"(r_) : "v"(v[c]), "v"(v[(c + 3) & 7])); v[c] = r_; }
                if (MODE == 4) { v[c] = __builtin_amdgcn_update_dpp(v[c], v[(c + 1) & 7], 0x138, 0xf, 0xf, false);
                                 v[c] = v[c] + 1; }
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s ^= v[c];
    if (s == 0x7fffffff) out[0] = s;
}
int main() {
    int* out; hipMalloc(&out, 1 << 20);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"v_max_i32 + v_add_u32", "v_pk_max_i16 + v_pk_add_u16", "v_pk_min_u16 + v_pk_sub_u16", "v_mov_dpp + v_pk_add_u16", "v_mov_dpp + v_add_u32",
                           "v_pk_mad_u16 x 2", "v_perm_b32 x 2", "v_alignbit_b32 + v_mad_u32_u24"};
    dim3 grid(pr.multiProcessorCount * 4), block(256);
    for (int mode = 0; mode < 8; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, rep);
            if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, rep);
            if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, rep);
            if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters, rep);
            if (mode == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, iters, rep);
            if (mode == 5) hipLaunchKernelGGL(k<5>, grid, block, 0, 0, out, iters, rep);
            if (mode == 6) hipLaunchKernelGGL(k<6>, grid, block, 0, 0, out, iters, rep);
            if (mode == 7) hipLaunchKernelGGL(k<7>, grid, block, 0, 0, out, iters, rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)grid.x * 256 * iters * 8.0 * 8 * 2;
            if (rep) printf("%-32s %.3f ms  %.2f T lane-instructions/s (int32 peak 39.3)\n", names[mode], ms, insts / ms / 1e9);
        }
    return 0;
}
