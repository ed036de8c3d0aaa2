// Shared host-side helpers of the gfx950 library (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ganet_hip.h"

#define GANET_RETURN_IF_HIP(expr)              \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return GANET_ERR_HIP; \
  } while (0)

static inline size_t ganet_align_up(size_t v, size_t a) {
  return (v + a - 1) / a * a;
}

static inline int ganet_check_launch() {
  return hipGetLastError() == hipSuccess ? GANET_OK : GANET_ERR_HIP;
}
