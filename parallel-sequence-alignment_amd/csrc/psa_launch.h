// psa_launch.h -- host-side launchers of the kernels, one per kernel translation unit.  The library is built from
// several translation units so that a change to one kernel family recompiles only that family (Makefile):
//   psa_fill_codes.hip  fill kernels with winner codes      psa_fill_score.hip  score-only fill kernels + the small
//   psa_trace.hip       traceback walks                                          kernels around the fill
//   psa_batch.hip       many-pairs kernel + its C ABI       psa_hip.hip         everything else of the C ABI (host)
//   psa_fill_pk.hip     packed fill of couples of pairs (multi-pair launches with winner codes)
#pragma once
#include <hip/hip_runtime.h>

#include "psa_args.hip.h"

namespace psa {

// which entry point of the v2 fill (psa_fill_v2.hip.h): one wavefront per SIMD, the same with the second capturing
// pass (launches in which a stripe has a right boundary), two wavefronts per SIMD
enum : int { kFillSparse = 0, kFillEdges = 1, kFillDense = 2 };

hipError_t launch_fill_v2_codes(const FillBatch& fb, unsigned total_groups, int kernel, size_t keepout_bytes, hipStream_t st);
hipError_t launch_fill_v2_score(const FillBatch& fb, unsigned total_groups, int kernel, size_t keepout_bytes, hipStream_t st);
// couples of pairs, two pairs per lane in packed 16-bit halves (psa_fill_pk.hip.h): jobs 2c, 2c + 1 = couple c
hipError_t launch_fill_pk(const FillBatch& fb, unsigned total_groups, bool with_codes, hipStream_t st);
hipError_t launch_fill_v1_codes(const FillArgs& fa, unsigned strips, hipStream_t st);
hipError_t launch_fill_v1_score(const FillArgs& fa, unsigned strips, hipStream_t st);

// FASTA-to-device encoding of one stripe (psa_fill_aux.hip.h): alphabet of A -> ids of A's rows -> table of B
hipError_t launch_tables(const uint8_t* a_dev, int64_t n, const uint8_t* b_dev, int W, uint8_t* alpha, uint8_t* a_ids,
                         uint32_t* eqtab, int tab_pitch, hipStream_t st);
constexpr size_t kAlphaBytes = 256 * 4 + 16 + 256 + 256;   // present[256] (u32) | meta[4] (u32) | map[256] | inv[256]
inline unsigned* alpha_meta(uint8_t* alpha) { return (unsigned*)alpha + 256; }

// the same for all slots of a multi-pair launch in three launches, with the zeroing the fill needs (BindBatch), and the
// slots' results gathered for one copy to the host (CollectBatch)
hipError_t launch_bind(const BindBatch& bb, hipStream_t st);
hipError_t launch_collect(const CollectBatch& cb, hipStream_t st);

hipError_t launch_edge_post(const int32_t* packed, int64_t r0, int64_t rows, int64_t n, int32_t* plane, uint32_t* flags, hipStream_t st);
hipError_t launch_edge_take(const int32_t* plane, const uint32_t* done, int64_t r0, int64_t rows, int64_t n, int32_t* packed,
                            unsigned* abort_flag, hipStream_t st);

// traceback walks (psa_trace.hip)
enum : int { kWalkPair = 0, kWalkWave = 1, kWalkLane = 2 };   // two wavefronts (default), one wavefront, one lane
hipError_t launch_traceback(int kind, const TraceArgs& ta, hipStream_t st);
hipError_t launch_traceback_many(int kind, const TraceBatch& tb, int nwalks, hipStream_t st);

}  // namespace psa
