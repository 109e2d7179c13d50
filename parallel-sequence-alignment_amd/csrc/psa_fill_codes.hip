// psa_fill_codes.hip -- translation unit: the fill kernels WITH winner codes (one byte per cell to HBM).
#include "psa_fill_tu.hip.h"

namespace psa {
hipError_t launch_fill_v2_codes(const FillBatch& fb, unsigned total_groups, int kernel, size_t keepout_bytes, hipStream_t st) {
    return launch_fill_v2_t<true>(fb, total_groups, kernel, keepout_bytes, st);
}
hipError_t launch_fill_v1_codes(const FillArgs& fa, unsigned strips, hipStream_t st) { return launch_fill_v1_t<true>(fa, strips, st); }
}  // namespace psa
