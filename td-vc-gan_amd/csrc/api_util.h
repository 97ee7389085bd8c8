// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tdvc.h"

// Records the message for tdvc_last_error() (thread-local) and returns `code`.
int tdvc_fail(int code, const char* msg);

#define TDVC_CHECK_LAUNCH()                                                      \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e__)); \
  } while (0)

static inline int tdvc_grid(long n, int block, int cap) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
