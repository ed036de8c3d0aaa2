// Status strings and ABI version of the gfx950 library.
#include "common.h"

extern "C" const char *ganet_status_string(int status) {
  switch (status) {
    case GANET_OK: return "ok";
    case GANET_ERR_NULL: return "a required pointer is NULL";
    case GANET_ERR_BAD_SHAPE: return "invalid shape or parameter";
    case GANET_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case GANET_ERR_WORKSPACE: return "workspace missing or too small";
    case GANET_ERR_HIP: return "HIP runtime / kernel launch error";
    case GANET_ERR_TOO_SMALL: return "convolution input is too small";
    case GANET_ERR_NOT_FINITE: return "a coordinate is not a finite 32-bit integer";
    default: return "unknown status";
  }
}

extern "C" int ganet_abi_version(void) { return GANET_HIP_ABI_VERSION; }
