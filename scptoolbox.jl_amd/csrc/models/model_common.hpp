// Shared helpers for the compiled device models that replace the reference's
// user closures traj.f/A/B/F (src/parser/problem.jl:432-450).
#pragma once
#include <hip/hip_runtime.h>

#define SCP_DEV __host__ __device__ __forceinline__

namespace scp {

template <int N>
SCP_DEV void zero(double (&a)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = 0.0;
}

}  // namespace scp
