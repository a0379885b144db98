// Micro-benchmark: cross-stream dependency through a device flag (hipStreamWaitValue32) against an event.
//   S1: B (tb us) whose last thread writes flag = rep ; S0: A (ta us) -> wait(flag >= rep) -> C.  Gap = C start - max(A end, B end).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/waitvalue_cost.hip -o /tmp/wv ; run: /tmp/wv
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

__global__ void spin(long long ticks, unsigned long long * out, unsigned int * flag, unsigned int val, unsigned int * counter)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
  while((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  if(out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0, out[1] = __builtin_amdgcn_s_memrealtime();
  if(flag && threadIdx.x == 0)
  {
    __threadfence();
    if(atomicAdd(counter, 1u) == gridDim.x - 1)
    {
      *counter = 0;
      __threadfence();
      __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if(out) out[4] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

int main()
{
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned long long * d;
  CK(hipMalloc(&d, 128));
  unsigned int * flag = nullptr, * counter = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
  if(e != hipSuccess) CK(hipMalloc((void **)&flag, 64));
  CK(hipMalloc((void **)&counter, 64));
  CK(hipMemset(flag, 0, 64));
  CK(hipMemset(counter, 0, 64));
  unsigned long long h[16];
  unsigned int rep = 0;
  const int ta = 60;
  {
    // reference point: A -> C back to back on one stream, no wait in between
    std::vector<double> gaps;
    for(int r = 0; r < 40; r++)
    {
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, (long long)ta * 100, d, (unsigned int *)nullptr, 0u, (unsigned int *)nullptr);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, 500LL, d + 2, (unsigned int *)nullptr, 0u, (unsigned int *)nullptr);
      CK(hipStreamSynchronize(s0));
      CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
      if(r >= 5) gaps.push_back(((double)h[2] - (double)h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("no wait: C starts %.2f us (median; min %.2f) after A's end\n", gaps[gaps.size() / 2], gaps[0]);
  }
  for(int tb : {5, 30, 50, 58, 62, 70})
  {
    std::vector<double> gaps;
    for(int r = 0; r < 40; r++)
    {
      rep++;
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, (long long)ta * 100, d, (unsigned int *)nullptr, 0u, (unsigned int *)nullptr);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, (long long)tb * 100, d + 8, flag, rep, counter);
      hipError_t w = hipStreamWaitValue32(s0, flag, rep, hipStreamWaitValueGte, 0xffffffffu);
      if(w != hipSuccess) { printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(w)); return 1; }
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, 500LL, d + 2, (unsigned int *)nullptr, 0u, (unsigned int *)nullptr);
      CK(hipStreamSynchronize(s0));
      CK(hipStreamSynchronize(s1));
      CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
      const double a_end = (double)h[1], b_flag = (double)h[12], c_start = (double)h[2];
      if(r >= 5) gaps.push_back((c_start - std::max(a_end, b_flag)) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("wait-value: A %d us on s0, B %d us on s1 (flag by its last workgroup) -> C starts %.2f us (median; min %.2f) after max(A end, flag write)\n", ta, tb,
           gaps[gaps.size() / 2], gaps[0]);
  }
  return 0;
}
