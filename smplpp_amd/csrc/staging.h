// Host/device argument staging for the non-hot entry points (stage-level ops, mesh queries, setters/getters).
#pragma once
#include "common.h"

namespace smplpp_hip
{
// An input that must be readable on the device: either the caller's device pointer or a temporary upload.
template<class T>
struct In
{
  const T * d = nullptr;
  T * tmp = nullptr;
  hipError_t init(const T * p, size_t count, int space, hipStream_t st)
  {
    if(!p || count == 0) return hipSuccess;
    if(space == SMPLPP_DEVICE)
    {
      d = p;
      return hipSuccess;
    }
    hipError_t e = hipMalloc((void **)&tmp, sizeof(T) * count);
    if(e != hipSuccess) return e;
    d = tmp;
    return hipMemcpyAsync(tmp, p, sizeof(T) * count, hipMemcpyHostToDevice, st);
  }
  ~In()
  {
    if(tmp) (void)hipFree(tmp);
  }
};

// An output: the caller's device pointer, or a temporary that is copied back by finish().
template<class T>
struct Out
{
  T * d = nullptr;
  T * tmp = nullptr;
  T * host = nullptr;
  size_t count = 0;
  hipError_t init(T * p, size_t cnt, int space)
  {
    if(!p || cnt == 0) return hipSuccess;
    count = cnt;
    if(space == SMPLPP_DEVICE)
    {
      d = p;
      return hipSuccess;
    }
    host = p;
    hipError_t e = hipMalloc((void **)&tmp, sizeof(T) * cnt);
    if(e == hipSuccess) d = tmp;
    return e;
  }
  hipError_t finish(hipStream_t st)
  {
    if(!tmp) return hipSuccess;
    return hipMemcpyAsync(host, tmp, sizeof(T) * count, hipMemcpyDeviceToHost, st);
  }
  ~Out()
  {
    if(tmp) (void)hipFree(tmp);
  }
};

inline int check_space(int space, const char * fn)
{
  if(space != SMPLPP_HOST && space != SMPLPP_DEVICE) return fail(SMPLPP_ERR_INVALID, std::string(fn) + ": bad memory space");
  return SMPLPP_OK;
}
} // namespace smplpp_hip
