// psa_batch.hip -- translation unit: the many-pairs score kernel (BASELINE config 5) and its C ABI entry points
// (psa_score_batch_dev, psa_score_batch_dev_noalloc; include/psa.h).
#include <algorithm>
#include <cstdlib>

#include "psa_batch.hip.h"
#include "psa_host.h"

using psa_host::fail;
using psa_host::need_device;

extern "C" {

int psa_score_batch_dev(const uint8_t* seqs_dev, const int64_t* offsets_dev, const int32_t* pairs_dev, int64_t npairs,
                        int max_len_b, int32_t* scores_dev, unsigned* counter_dev, void* stream) {
    if (npairs <= 0) return PSA_OK;
    if (max_len_b < 0 || max_len_b > psa::kBatchMaxLen)
        return fail(PSA_E_ARG, "psa_score_batch_dev: max_len_b must be in [0, %d]", psa::kBatchMaxLen);
    if (!seqs_dev || !offsets_dev || !pairs_dev || !scores_dev || !counter_dev)
        return fail(PSA_E_ARG, "psa_score_batch_dev: null argument");
    if (int rc = need_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(counter_dev, 0, 256, st));
    psa::BatchArgs ba;
    ba.seqs = seqs_dev; ba.offsets = offsets_dev; ba.pairs = pairs_dev; ba.npairs = npairs; ba.scores = scores_dev;
    ba.next_pair = counter_dev;
    static const int no_packed = [] { const char* v = getenv("PSA_BATCH_PACKED"); return (v && v[0] == '0') ? 1 : 0; }();
    ba.no_packed = no_packed;
    ba.max_b = std::max(64, ((max_len_b + 63) / 64) * 64);
    const size_t lds_bytes = (size_t)psa::kBatchWaves * psa::batch_wave_bytes(ba.max_b);
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t want = ((npairs + 1) / 2 + psa::kBatchWaves - 1) / psa::kBatchWaves;   // (a wavefront draws couples of pairs)
    const unsigned grid = (unsigned)std::min<int64_t>(want, (int64_t)cus * 8);
    hipLaunchKernelGGL(psa::batch_score_kernel, dim3(grid), dim3(psa::kBatchWaves * 64), lds_bytes, st, ba);
    HIP_TRY(hipGetLastError());
    return PSA_OK;
}


}  // extern "C"

// ---- psa_score_batch_dev without any allocation and without a workspace argument ----
// The batch kernel's scratch -- its work counter, the only device memory it uses beyond inputs and outputs --
// lives in this file-scope device array (256 bytes on every GPU the library runs on, until the process ends).
// In host code the array's NAME is only a stand-in: the address kernels and memsets get is the one
// hipGetSymbolAddress returns for the current device, looked up once per device and kept.
static __device__ __align__(256) unsigned scratch[64];

namespace {
constexpr int kMaxScratchDevices = 64;
constexpr int64_t kNoAllocMaxPairs = 1LL << 30;   // the counter is 32 bits wide; the kernel adds to it once per pair and wave
unsigned* g_scratch_ptr[kMaxScratchDevices] = {nullptr};

int scratch_for_current_device(unsigned** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxScratchDevices) return fail(PSA_E_ARG, "device %d: only devices 0..%d have a scratch slot", dev, kMaxScratchDevices - 1);
    unsigned* ptr = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!ptr) {   // first call on this device (it is current): ask the runtime where this GPU's copy of the array lies
        HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(PSA_E_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
        __atomic_store_n(&g_scratch_ptr[dev], ptr, __ATOMIC_RELEASE);   // (two threads racing here store the same address)
    }
    *out = ptr;
    return PSA_OK;
}
}  // namespace

extern "C" {

int psa_score_batch_dev_noalloc(const uint8_t* seqs_dev, const int64_t* offsets_dev, const int32_t* pairs_dev, int64_t npairs,
                                int max_len_b, int32_t* scores_dev, void* stream) {
    if (npairs <= 0) return PSA_OK;
    if (npairs > kNoAllocMaxPairs)
        return fail(PSA_E_ARG, "psa_score_batch_dev_noalloc: %lld pairs exceed the maximum of %lld per call", (long long)npairs,
                    (long long)kNoAllocMaxPairs);
    if (int rc = need_device()) return rc;
    unsigned* ptr = nullptr;
    if (int rc = scratch_for_current_device(&ptr)) return rc;
    // (psa_score_batch_dev zeroes the counter it is handed -- here this GPU's scratch, which still holds the
    // last call's count -- and launches the kernel; it allocates nothing)
    return psa_score_batch_dev(seqs_dev, offsets_dev, pairs_dev, npairs, max_len_b, scores_dev, ptr, stream);
}

}  // extern "C"
