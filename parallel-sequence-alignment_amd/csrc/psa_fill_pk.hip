// psa_fill_pk.hip -- translation unit: the packed fill of many independent pairs, two pairs per lane (psa_fill_pk.hip.h).
#include "psa_fill_pk.hip.h"
#include "psa_launch.h"

namespace psa {
hipError_t launch_fill_pk(const FillBatch& fb, unsigned total_groups, bool with_codes, hipStream_t st) {
    if (with_codes) hipLaunchKernelGGL(fill_strips_pk<true>, dim3(total_groups), dim3(kPkNW * 64), kKeepoutDenseBytes, st, fb);
    else hipLaunchKernelGGL(fill_strips_pk<false>, dim3(total_groups), dim3(kPkNW * 64), kKeepoutDenseBytes, st, fb);
    return hipGetLastError();
}
}  // namespace psa
