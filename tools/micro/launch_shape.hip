// Micro-benchmark (dev probe): what a kernel of the fused FK kernel's launch shape costs outside its own instructions.
//   (a) empty body, 256 workgroups x 256 threads, with and without the 157 KiB dynamic LDS allocation
//   (b) a body that only writes the step's 85 MB of output (one 12-byte store per lane and row, like the fused kernel), for
//       each cache policy of the stores: in-kernel time (first start .. last end, 100 MHz counter) against the stream time
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/launch_shape.hip -o /tmp/launch_shape ; run: /tmp/launch_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef unsigned v3u __attribute__((ext_vector_type(3)));

__global__ __launch_bounds__(256, 1) void empty_kernel(unsigned long long * t)
{
  extern __shared__ unsigned char lds[];
  if(t && threadIdx.x == 0)
  {
    t[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

template<int AUX>
__global__ __launch_bounds__(256, 1) void store_kernel(float * out, long long nrow, int rowBytes, unsigned long long * t)
{
  // workgroup b writes rows b, b + grid, ...; a row = rowBytes bytes = 12-byte elements, 256 lanes x 12 B per pass
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffff00, 0x00020000);
  const v3u val = {threadIdx.x, blockIdx.x, 7u};
  for(long long r = blockIdx.x; r < nrow; r += gridDim.x)
    for(int o = threadIdx.x * 12; o + 12 <= rowBytes; o += 256 * 12)
      __builtin_amdgcn_raw_buffer_store_b96(val, rs, o, (int)(r * rowBytes), AUX);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if(threadIdx.x == 0)
  {
    t[2 * blockIdx.x] = t0;
    t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

template<class L>
static double stream_us(L launch, int reps)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for(int i = 0; i < 20; i++) launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for(int i = 0; i < reps; i++) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / reps;
}

int main()
{
  unsigned long long * t;
  hipMalloc(&t, 2 * 256 * 8);
  const int LDSB = 160768;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  printf("empty, no LDS      : %.2f us per launch back to back\n", stream_us([&] { empty_kernel<<<256, 256, 0, 0>>>(t); }, 300));
  printf("empty, 157 KiB LDS : %.2f us per launch back to back\n", stream_us([&] { empty_kernel<<<256, 256, LDSB, 0>>>(t); }, 300));
  {
    std::vector<unsigned long long> h(512);
    empty_kernel<<<256, 256, LDSB, 0>>>(t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, 512 * 8, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0;
    for(int b = 0; b < 256; b++) mn = std::min(mn, h[2 * b]), mx = std::max(mx, h[2 * b]);
    printf("empty, 157 KiB LDS : workgroup starts spread over %.2f us\n", (mx - mn) / 100.0);
  }
  const long long nrow = 1024; // frames
  const int rowBytes = 6890 * 12;
  float * out;
  hipMalloc(&out, nrow * rowBytes);
  auto run = [&](auto tag, const char * name) {
    constexpr int AUX = decltype(tag)::value;
    const double us = stream_us([&] { store_kernel<AUX><<<256, 256, 0, 0>>>(out, nrow, rowBytes, t); }, 200);
    std::vector<unsigned long long> h(512);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, 512 * 8, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0;
    for(int b = 0; b < 256; b++) mn = std::min(mn, h[2 * b]), mx = std::max(mx, h[2 * b + 1]);
    printf("store 85 MB, aux %2d (%s): %.2f us per launch back to back; first start .. last end inside %.2f us -> %.2f TB/s\n", AUX, name, us,
           (mx - mn) / 100.0, nrow * rowBytes / us / 1e6);
  };
  run(std::integral_constant<int, 0>{}, "default");
  run(std::integral_constant<int, 1>{}, "sc0");
  run(std::integral_constant<int, 2>{}, "nt");
  run(std::integral_constant<int, 3>{}, "sc0 nt");
  run(std::integral_constant<int, 16>{}, "sc1");
  run(std::integral_constant<int, 17>{}, "sc0 sc1");
  run(std::integral_constant<int, 19>{}, "sc0 sc1 nt");
  return 0;
}
