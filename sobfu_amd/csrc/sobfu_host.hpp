// Host-side helpers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "sobfu_hip.h"

#define SOBFU_CHECK_ARGS(cond)                \
    do {                                      \
        if (!(cond)) return SOBFU_E_BADARG;   \
    } while (0)

#define SOBFU_HIP_TRY(expr)                   \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int) _e; \
    } while (0)

#define SOBFU_TRY(expr)                       \
    do {                                      \
        int _rc = (expr);                     \
        if (_rc != 0) return _rc;             \
    } while (0)
