// acez_common.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/acez.h"

namespace acez {
void set_error(const char* fmt, ...);
}

#define ACEZ_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      acez::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ACEZ_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define ACEZ_REQUIRE(cond, msg)                          \
  do {                                                   \
    if (!(cond)) {                                       \
      acez::set_error("invalid argument: %s", msg);      \
      return ACEZ_ERR_INVALID;                           \
    }                                                    \
  } while (0)
