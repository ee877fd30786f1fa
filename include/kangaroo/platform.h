// platform.h -- export / host-device decoration macros for the roo:: headers.
// Counterpart of the reference's include/kangaroo/platform.h:5-9 (KANGAROO_EXPORT is empty off MSVC).
#pragma once

#include <kangaroo/config.h>

#define KANGAROO_EXPORT

// The containers are plain structs usable on the host and inside HIP kernels.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KANGAROO_HD __host__ __device__
#else
#include <hip/hip_vector_types.h>
#define KANGAROO_HD
#endif
