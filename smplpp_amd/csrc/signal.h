// Cross-stream hand-over through device flags (the waiting stream uses hipStreamWaitValue32): shared by the IK kernels and the
// VPoser decoder kernels.  See the comment in front of st_agent's first use in ik.hip for the measurements behind the scheme.
#pragma once
#include <hip/hip_runtime.h>

namespace smplpp_hip
{
template<class T>
__device__ inline void st_agent(T * p, T v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void wg_signal(unsigned * __restrict__ flag, unsigned * __restrict__ counter, unsigned tick)
{
  if(!flag) return;
  // every thread's stores have been acknowledged before the workgroup counts itself in: the wait is explicit (the barrier
  // alone orders LDS and, outside threadgroup-split mode, is not defined to drain the vector-memory counter)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if(threadIdx.x == 0)
  {
    if(__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
    {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag, tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

} // namespace smplpp_hip
