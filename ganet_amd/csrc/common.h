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

// Ranges inside which the host arithmetic of the deform-conv plans (products of these fields
// in int, long long and size_t) cannot overflow; called after the positivity checks.  The
// reference has no such limits on paper, its int-indexed kernels have tighter ones in practice.
static inline int ganet_dcn_ranges_ok(const ganet_dcn_params *p) {
  const int lim_k = 1 << 12, lim_c = 1 << 20, lim_p = 1 << 20;
  if (p->kH > lim_k || p->kW > lim_k || p->sH > lim_k || p->sW > lim_k || p->dH > lim_k ||
      p->dW > lim_k)
    return 0;
  if (p->pH > lim_p || p->pW > lim_p || p->Cin > lim_c || p->Cout > lim_c) return 0;
  if ((long long)p->kH * p->kW > (1 << 16)) return 0;
  const long long bh = (long long)p->B * p->H;          // both below 2^31
  if (bh >= (1ll << 31) || bh * p->W >= (1ll << 31)) return 0;
  return 1;
}
