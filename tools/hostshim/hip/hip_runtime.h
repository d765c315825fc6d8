// host-only stand-in so frr_exact.h (HIP __host__ __device__ helpers) compiles with plain g++ for
// tools/atan2f_check.cpp.  Not used by the product build.
#pragma once
#define __host__
#define __device__
#ifndef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#endif
