// psa_trace.hip -- translation unit: the traceback walks (two-wave default, single-wave, single-lane).
#include "psa_trace2.hip.h"
#include "psa_launch.h"

namespace psa {
hipError_t launch_traceback(int kind, const TraceArgs& ta, hipStream_t st) {
    if (kind == kWalkLane) hipLaunchKernelGGL(traceback_v1, dim3(1), dim3(64), 0, st, ta);
    else if (kind == kWalkWave) hipLaunchKernelGGL(traceback_wave, dim3(1), dim3(64), 0, st, ta);
    else hipLaunchKernelGGL(traceback_pair, dim3(1), dim3(128), 0, st, ta);
    return hipGetLastError();
}
hipError_t launch_traceback_many(int kind, const TraceBatch& tb, int nwalks, hipStream_t st) {
    if (kind == kWalkWave) hipLaunchKernelGGL(traceback_wave_many, dim3(nwalks), dim3(64), 0, st, tb);
    else hipLaunchKernelGGL(traceback_pair_many, dim3(nwalks), dim3(128), 0, st, tb);
    return hipGetLastError();
}
}  // namespace psa
