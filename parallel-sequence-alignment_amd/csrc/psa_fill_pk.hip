// psa_fill_pk.hip -- translation unit: the packed fill of many independent pairs, two pairs per lane (psa_fill_pk.hip.h).
#include "psa_fill_pk.hip.h"
#include "psa_launch.h"

namespace psa {
hipError_t launch_fill_pk(const FillBatch& fb, unsigned total_groups, hipStream_t st) {
    hipLaunchKernelGGL(fill_strips_pk, dim3(total_groups), dim3(kNW * 64), kKeepoutDenseBytes, st, fb);
    return hipGetLastError();
}
}  // namespace psa
