// psa_fill_score.hip -- translation unit: the score-only fill kernels and the small kernels around the fill
// (alphabet / substitution table of a stripe, boundary chunks of a flagged fill).
#include <algorithm>

#include "psa_fill_tu.hip.h"
#include "psa_fill_aux.hip.h"

namespace psa {
hipError_t launch_fill_v2_score(const FillBatch& fb, unsigned total_groups, int kernel, size_t keepout_bytes, hipStream_t st) {
    return launch_fill_v2_t<false>(fb, total_groups, kernel, keepout_bytes, st);
}
hipError_t launch_fill_v1_score(const FillArgs& fa, unsigned strips, hipStream_t st) { return launch_fill_v1_t<false>(fa, strips, st); }

hipError_t launch_tables(const uint8_t* a_dev, int64_t n, const uint8_t* b_dev, int W, uint8_t* alpha, uint8_t* a_ids,
                         uint32_t* eqtab, int tab_pitch, hipStream_t st) {
    unsigned* present = (unsigned*)alpha;
    unsigned* meta = alpha_meta(alpha);
    uint8_t* map = (uint8_t*)(meta + 4);
    uint8_t* inv = map + 256;
    hipError_t e = hipMemsetAsync(alpha, 0, kAlphaBytes, st);
    if (e != hipSuccess) return e;
    const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(alphabet_presence, dim3(blocks), dim3(256), 0, st, a_dev, n, present);
    hipLaunchKernelGGL(alphabet_compact, dim3(1), dim3(64), 0, st, present, meta, map, inv);
    hipLaunchKernelGGL(encode_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a_dev, n, map, a_ids);
    hipLaunchKernelGGL(build_eqtab, dim3((unsigned)((tab_pitch + 255) / 256), kMaxSym), dim3(256), 0, st, b_dev, W, meta, inv,
                       eqtab, tab_pitch);
    return hipGetLastError();
}

hipError_t launch_bind(const BindBatch& bb, hipStream_t st) {
    if (bb.njobs < 1) return hipSuccess;
    int64_t most = 0;
    bool tables = false;
    for (int j = 0; j < bb.njobs; ++j) {
        const BindJob& q = bb.job[j];
        for (int r = 0; r < kBindRegions; ++r) most = std::max<int64_t>(most, q.clear_bytes[r] / 16);
        if (q.tables) { tables = true; most = std::max<int64_t>(most, std::max<int64_t>(q.n, (int64_t)kMaxSym * q.pitch / 8)); }
    }
    const unsigned blocks = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, (most + 1023) / 1024));
    hipLaunchKernelGGL(bind_stage, dim3(blocks, bb.njobs), dim3(256), 0, st, bb);
    if (tables) {
        hipLaunchKernelGGL(bind_alphabet, dim3(bb.njobs), dim3(256), 0, st, bb);
        hipLaunchKernelGGL(bind_tables, dim3(blocks, bb.njobs), dim3(256), 0, st, bb);
    }
    return hipGetLastError();
}

hipError_t launch_collect(const CollectBatch& cb, hipStream_t st) {
    if (cb.njobs < 1) return hipSuccess;
    hipLaunchKernelGGL(collect_results, dim3(cb.njobs), dim3(kResultWords), 0, st, cb);
    return hipGetLastError();
}

hipError_t launch_edge_post(const int32_t* packed, int64_t r0, int64_t rows, int64_t n, int32_t* plane, uint32_t* flags, hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<int64_t>((3 * rows + 255) / 256, 1024);
    hipLaunchKernelGGL(edge_scatter, dim3(blocks), dim3(256), 0, st, packed, r0, rows, n, plane);
    // (a kernel of its own: everything the scatter stored is visible device-wide when this one starts)
    const unsigned nflags = (unsigned)((rows + 63) / 64);
    hipLaunchKernelGGL(edge_flags, dim3((nflags + 255) / 256), dim3(256), 0, st, flags + r0 / 64, nflags);
    return hipGetLastError();
}

hipError_t launch_edge_take(const int32_t* plane, const uint32_t* done, int64_t r0, int64_t rows, int64_t n, int32_t* packed,
                            unsigned* abort_flag, hipStream_t st) {
    const unsigned blocks = (unsigned)std::min<int64_t>((3 * rows + 255) / 256, 64);
    hipLaunchKernelGGL(edge_gather, dim3(blocks), dim3(256), 0, st, plane, done, r0, rows, n, packed, abort_flag);
    return hipGetLastError();
}
}  // namespace psa
