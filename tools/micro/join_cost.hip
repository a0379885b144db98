// Micro-benchmark: what a cross-stream join costs the waiting stream, as a function of how long before the waiter arrives the
// event was signalled.  Stream S0: A(ta us) -> wait(ev) -> C ; stream S1: B(tb us) records ev on completion.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/join_cost.hip -o /tmp/join_cost ; run: /tmp/join_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void spin(long long ticks, unsigned long long * out)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
  while((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  if(out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0, out[1] = __builtin_amdgcn_s_memrealtime();
}

int main()
{
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t ev, fork;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventCreateWithFlags(&fork, hipEventDisableTiming);
  unsigned long long * d;
  hipMalloc(&d, 64);
  unsigned long long h[6];
  const int ta = 60; // us of A
  for(int mode = 0; mode < 2; mode++)
    for(int tb : {5, 30, 50, 58, 62, 70, 90})
    {
      std::vector<double> gaps;
      for(int rep = 0; rep < 40; rep++)
      {
        // F: a tiny kernel on s0 whose completion forks s1 (as the IK loop does), then A on s0 and B on s1
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, nullptr, fork, 0, 100LL, (unsigned long long *)nullptr);
        hipStreamWaitEvent(s1, fork, 0);
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, (long long)ta * 100, d);
        if(mode == 0)
          hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, nullptr, ev, 0, (long long)tb * 100, (unsigned long long *)nullptr);
        else
        {
          hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, (long long)tb * 100, (unsigned long long *)nullptr);
          hipEventRecord(ev, s1);
        }
        hipStreamWaitEvent(s0, ev, 0);
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, 500LL, d + 2);
        hipStreamSynchronize(s0);
        hipStreamSynchronize(s1);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        if(rep >= 5) gaps.push_back(((double)h[2] - (double)h[1]) / 100.0);
      }
      std::sort(gaps.begin(), gaps.end());
      printf("%s: A %d us on s0, B %d us on s1 -> gap A end to C start: median %.2f us (min %.2f)\n", mode == 0 ? "stopEvent " : "eventRecord", ta, tb,
             gaps[gaps.size() / 2], gaps[0]);
    }
  // baseline: no wait at all
  {
    std::vector<double> gaps;
    for(int rep = 0; rep < 40; rep++)
    {
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, (long long)ta * 100, d);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, 500LL, d + 2);
      hipStreamSynchronize(s0);
      hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      if(rep >= 5) gaps.push_back(((double)h[2] - (double)h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("same stream, no wait: gap median %.2f us\n", gaps[gaps.size() / 2]);
  }
  return 0;
}
