// launch_utils.h -- error convention of the operator wrappers.
// The reference checks cudaGetLastError after SdfFuse / RaycastSdf / SdfSphere / SdfReset launches and,
// on failure, prints "cudaCheckError() failed at file:line : msg" and calls exit(-1)
// (include/kangaroo/launch_utils.h:29-47); BilateralFilter, DepthToVbo and NormalsFromVbo are not
// checked at all (quirk Q9).  The C ABI returns a status instead; these helpers map it back.
#pragma once

#include <cstdio>
#include <cstdlib>

#include <kfx.h>

#define GpuCheckStatus(st) roo::__StatusOrDie((st), __FILE__, __LINE__)
#define GpuNoteStatus(st) roo::__StatusOrWarn((st), __FILE__, __LINE__)

namespace roo
{

inline void __StatusOrDie(int st, const char* file, const int line)
{
    if (st != 0) {
        fprintf(stderr, "hipCheckError() failed at %s:%i : %s\n", file, line, kfx_last_error_string());
        exit(-1);
    }
}

// unchecked ops of the reference: keep going, but leave a trace on stderr
inline void __StatusOrWarn(int st, const char* file, const int line)
{
    if (st != 0) fprintf(stderr, "warning: launch failed at %s:%i : %s\n", file, line, kfx_last_error_string());
}

}

inline int GetLevelFromMaxPixels(size_t w, size_t h, unsigned long maxpixels)
{
    int level = 0;
    while ((w >> level) * (h >> level) > maxpixels) ++level;
    return level;
}
