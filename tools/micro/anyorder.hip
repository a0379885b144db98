// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this runtime? (dev probe)
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder.hip -o /tmp/anyorder ; run: /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin_kernel(unsigned long long ticks, int * out)
{
  const unsigned long long t0 = wall_clock64();
  while(wall_clock64() - t0 < ticks) {}
  if(out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main()
{
  int * d;
  hipMalloc(&d, 4);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned long long ticks = 100ull * 100; // s_memtime runs at 100 MHz: 100 us
  for(int mode = 0; mode < 3; mode++)
  {
    for(int rep = 0; rep < 3; rep++)
    {
      hipStreamSynchronize(st);
      hipEventRecord(a, st);
      hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, st, nullptr, nullptr, 0, ticks, d);
      hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, st, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0, ticks, d);
      if(mode == 2) hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, st, nullptr, nullptr, 0, ticks, d);
      hipEventRecord(b, st);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "two ordered kernels" : (mode == 1 ? "second any-order" : "ordered, any-order, ordered"), ms * 1e3);
    }
  }
  return 0;
}
