// Small host-side helpers shared by the translation units of libnsff_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/nsff_render.h"

inline thread_local hipError_t g_nsff_last_err = hipSuccess;
inline int nsff_hip_fail(hipError_t e) { g_nsff_last_err = e; return NSFF_ERR_HIP; }
inline int nsff_launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? NSFF_OK : nsff_hip_fail(e);
}
