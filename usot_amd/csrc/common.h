// Shared helpers for the libusot_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#define USOT_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH; \
    } while (0)

static inline int usot_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
