// psa_host.h -- host-side helpers shared by the translation units that implement the C ABI (psa_hip.hip, psa_batch.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/psa.h"

namespace psa_host {
// records the message psa_last_error() returns (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// PSA_OK, or PSA_E_NO_DEVICE with a message: the library has no CPU path
int need_device();
}  // namespace psa_host

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return psa_host::fail(e_ == hipErrorOutOfMemory ? PSA_E_NOMEM : PSA_E_HIP, "%s failed: %s", \
                                  #expr, hipGetErrorString(e_));                                         \
    } while (0)
