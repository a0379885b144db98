// Micro-probe: the largest dynamic LDS size hipFuncSetAttribute accepts for a kernel with a given static LDS footprint.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_limit.hip -o /tmp/lds_limit ; run: /tmp/lds_limit
#include <hip/hip_runtime.h>
#include <cstdio>
template<int S>
__global__ void k(float * out)
{
  __shared__ float st[S / 4];
  extern __shared__ float dyn[];
  st[threadIdx.x] = threadIdx.x;
  dyn[threadIdx.x] = 1.0f;
  __syncthreads();
  out[threadIdx.x] = st[(threadIdx.x + 1) % 64] + dyn[(threadIdx.x + 2) % 64];
}
template<int S>
void probe()
{
  int best = -1;
  for(int d = 100 * 1024; d <= 164 * 1024; d += 16)
    if(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<S>), hipFuncAttributeMaxDynamicSharedMemorySize, d) == hipSuccess) best = d;
  (void)hipGetLastError();
  float * o;
  hipMalloc(&o, 1024);
  k<S><<<1, 64, best>>>(o);
  hipError_t e = hipDeviceSynchronize();
  printf("static %6d: largest dynamic accepted %6d (sum %6d), launch at that size: %s\n", S, best, S + best, hipGetErrorString(e));
}
int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu, sharedMemPerBlockOptin %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.maxSharedMemoryPerMultiProcessor);
  probe<1024>();
  probe<30880>();
  probe<36464>();
  return 0;
}
