// pk_step.hip -- what bounds the packed 16-bit step of the batch kernel (psa_batch.hip.h, batch_block_pk): the same
// sixteen-step block in a loop, W waves per SIMD on every CU, with the row hand-off done in different ways.
//   mode 0: no LDS write (values folded into a checksum so that nothing is dead)     mode 1: masked write per step
//   mode 2: masked, once per block (16 x 8 bytes)   mode 3: unmasked 8-byte write per step (other lanes to a dump row)
//   mode 4: as 0 without the LDS reads either (pure VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_of(int v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ int pk_bits(pk16 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ int shift_down1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }

template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int blocks, int seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* mine = smem + wv * 9216;
    int2* row = (int2*)mine;                  // 672 slots
    int* bpk = (int*)(mine + 5376) + 128;     // 768 ints
    int* dump = (int*)(mine + 5376 + 3072);   // 768 bytes
    for (int c = lane; c < 672; c += 64) row[c] = make_int2(c * 0x10001, c * 0x20002);
    for (int c = lane - 128; c < 640; c += 64) bpk[c] = (c * 7 + seed) & 0x00fe00fe;
    pk16 D0 = pk_of(lane), D1 = pk_of(lane + 1), H = pk_of(3), V = pk_of(5);
    const pk16 a2 = pk_of((lane * 2) | (lane << 17));
    const pk16 two = pk_of(0x00020002), eight = pk_of(0x00080008);
    int check = 0;
    for (int b = 0; b < blocks; ++b) {
        const int t0 = (b & 31) * 16;
        int4 pre[8];
        int bq[16];
        if (MODE != 4) {
            const int4* rp = (const int4*)(row + t0 + 64);
#pragma unroll
            for (int x = 0; x < 8; ++x) pre[x] = rp[x];
            const int* bp = bpk - lane + t0;
#pragma unroll
            for (int x = 0; x < 16; ++x) bq[x] = bp[x];
        } else {
#pragma unroll
            for (int x = 0; x < 8; ++x) pre[x] = make_int4(b + x, b, x, seed);
#pragma unroll
            for (int x = 0; x < 16; ++x) bq[x] = b * x + seed;
        }
        typedef __attribute__((address_space(3))) int lds_int;
        typedef int v2i __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) v2i lds_v2i;
        unsigned pa_off = (unsigned)(uintptr_t)(lds_int*)((int*)row + 2 * t0);
        asm volatile("" : "+v"(pa_off));
        lds_int* pl = (lds_int*)(uintptr_t)pa_off;
        unsigned du_off = (unsigned)(uintptr_t)(lds_int*)(dump + 2 * lane);
        lds_int* dl = (lds_int*)(uintptr_t)du_off;
        int hd[16], hv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const upk16 d = __builtin_bit_cast(upk16, bq[kk]) - __builtin_bit_cast(upk16, a2);
            const upk16 z = __builtin_elementwise_min(d, __builtin_bit_cast(upk16, two));
            const pk16 f = eight - __builtin_bit_cast(pk16, z);
            const int preD = (kk & 1) ? pre[kk >> 1].z : pre[kk >> 1].x;
            const int preV = (kk & 1) ? pre[kk >> 1].w : pre[kk >> 1].y;
            const pk16 x1 = pk_of(shift_down1(preD, pk_bits((kk & 1) ? D1 : D0))) + f;
            const pk16 x3 = pk_of(shift_down1(preV, pk_bits(V)));
            const pk16 x2 = H;
            const pk16 M = pk_max(x1, x2);
            const pk16 D = pk_max(M, x3);
            const pk16 Vn = pk_max(M, x3 + two);
            const pk16 Hn = pk_max(D, x2 + two);
            const pk16 Dprev = (kk & 1) ? D0 : D1;
            if (kk & 1) D1 = D; else D0 = D;
            H = Hn; V = Vn;
            if (MODE == 1) { if (lane == 63) *(lds_v2i*)(pl + 2 * (kk + 1)) = v2i{pk_bits(Dprev), pk_bits(V)}; }
            if (MODE == 3) *(lds_v2i*)((lane == 63 ? pl + 2 * (kk + 1) : dl)) = v2i{pk_bits(Dprev), pk_bits(V)};
            hd[kk] = pk_bits(Dprev); hv[kk] = pk_bits(V);
        }
        if (MODE == 2) {
            if (lane == 63) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) *(lds_v2i*)(pl + 2 * (kk + 1)) = v2i{hd[kk], hv[kk]};
            }
        }
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int kk = 0; kk < 16; kk += 15) check ^= hd[kk] + hv[kk];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = check ^ pk_bits(D0) ^ pk_bits(D1) ^ pk_bits(H) ^ pk_bits(V);
}
int main() {
    int* out; hipMalloc(&out, 1 << 24);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 20000;
    const char* names[] = {"no write (checksum)", "masked write per step", "masked, once per block", "unmasked 8 B per step", "pure VALU"};
    for (int wps : {1, 2, 4})
        for (int mode = 0; mode < 5; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                dim3 grid(pr.multiProcessorCount * wps), block(256);
                const size_t lds = 4 * 9216;
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, lds, 0, out, blocks, rep);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, lds, 0, out, blocks, rep);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, lds, 0, out, blocks, rep);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, lds, 0, out, blocks, rep);
                if (mode == 4) hipLaunchKernelGGL(k<4>, grid, block, lds, 0, out, blocks, rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // cycles of SIMD time per wave-step at 2.4 GHz
                if (rep) printf("waves/SIMD %d  %-24s %8.3f ms  %6.1f SIMD cycles per step\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / (blocks * 16.0 * wps));
            }
    return 0;
}
