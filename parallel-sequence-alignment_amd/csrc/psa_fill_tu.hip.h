// psa_fill_tu.hip.h -- body of the two fill translation units (with / without winner codes): the launchers of
// fill_strips_v2 (three entry points) and fill_strips_v1 for one value of CODES.
#pragma once
#include "psa_fill_v2.hip.h"
#include "psa_launch.h"

namespace psa {

template <bool CODES>
hipError_t launch_fill_v2_t(const FillBatch& fb, unsigned total_groups, int kernel, size_t keepout_bytes, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {   // dynamic LDS beyond 64 KB has to be allowed once per kernel
        hipError_t e = hipFuncSetAttribute((const void*)fill_strips_v2<CODES>, hipFuncAttributeMaxDynamicSharedMemorySize, kKeepoutAloneBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)fill_strips_v2_edges<CODES>, hipFuncAttributeMaxDynamicSharedMemorySize, kKeepoutAloneBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 grid(total_groups), block(kNW * 64);
    if (kernel == kFillDense) hipLaunchKernelGGL(fill_strips_v2_dense<CODES>, grid, block, kKeepoutDenseBytes, st, fb);
    else if (kernel == kFillEdges) hipLaunchKernelGGL(fill_strips_v2_edges<CODES>, grid, block, keepout_bytes, st, fb);
    else hipLaunchKernelGGL(fill_strips_v2<CODES>, grid, block, keepout_bytes, st, fb);
    return hipGetLastError();
}

template <bool CODES>
hipError_t launch_fill_v1_t(const FillArgs& fa, unsigned strips, hipStream_t st) {
    hipLaunchKernelGGL(fill_strips_v1<CODES>, dim3(strips), dim3(64), 0, st, fa);
    return hipGetLastError();
}

}  // namespace psa
