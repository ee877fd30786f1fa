// config.h -- build configuration of the MI355X implementation.  The reference generates this
// file from config.h.in with CMake (HAVE_EIGEN / HAVE_THRUST / HAVE_NPP / HAVE_OPENCV switches);
// none of those optional libraries is used here, so it is static.
#pragma once

#define _UNIX_
#define _LINUX_
#define KANGAROO_HIP 1          // device back end: HIP / gfx950 through libkfx (include/kfx.h)

#if (__cplusplus > 199711L)
#define CALLEE_HAS_CPP11
#define CALLEE_HAS_RVALREF
#endif
