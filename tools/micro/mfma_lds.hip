// Microbenchmark (development aid): one wavefront per SIMD issuing v_mfma_f32_32x32x16_bf16 whose operands come from
// ds_read_b128 (12 reads per 18 MFMAs, as in skin_b.hip).  Prints cycles per MFMA (s_memtime) and wall time.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_lds tools/micro/mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template<int MODE> // 0: MFMA on fixed registers; 1: reads issued, MFMA on fixed registers; 2: MFMA on the read data (next k-step)
__global__ __launch_bounds__(256, 1) void k(const float * in, float * out, long long * cyc, int iters)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  for(int i = tid; i < 24576 * 2 / 4; i += 256) reinterpret_cast<float *>(lds)[i] = in[i % 4096];
  __syncthreads();
  const unsigned char * base = lds + (tid & 63) * 16 + (tid >> 6) * 1024;
  f32x16 acc[3] = {};
  v4f a[2][3], b[2][3][3];
  for(int p = 0; p < 2; p++)
    for(int s = 0; s < 3; s++)
    {
      a[p][s] = *reinterpret_cast<const v4f *>(base + s * 1024);
      for(int x = 0; x < 3; x++) b[p][x][s] = *reinterpret_cast<const v4f *>(base + (3 + 3 * x + s) * 1024);
    }
  long long t0 = __builtin_readcyclecounter();
  for(int it = 0; it < iters; it++)
  {
#pragma unroll
    for(int ks = 0; ks < 2; ks++)
    {
#pragma unroll
      for(int m = 0; m < 18; m++)
      {
        const int x = m / 6, q = m % 6;
        const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        const int P = (MODE == 2) ? ks : 0;
        acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[P][pa[q]]), __builtin_bit_cast(bf16x8, b[P][x][pb[q]]), acc[x], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if(MODE >= 1 && m >= 6 && m < 15)
        {
          const int np = (MODE == 2) ? (ks ^ 1) : 1, xx = (m - 6) / 3, sp = (m - 6) % 3;
          const unsigned char * img = base + ((it + ks) & 1) * 24576;
          if(xx == 0) a[np][sp] = *reinterpret_cast<const v4f *>(img + sp * 1024);
          b[np][xx][sp] = *reinterpret_cast<const v4f *>(img + (3 + 3 * xx + sp) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for(int x = 0; x < 3; x++)
    for(int r = 0; r < 16; r++) s += acc[x][r];
  if(MODE == 1) s += a[1][0].x + b[1][0][0].x + b[1][1][1].y + b[1][2][2].z + a[1][1].x + a[1][2].x;
  out[blockIdx.x * 256 + tid] = s;
  if(tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int MODE>
void run(const char * name, float * in, float * out, long long * cyc, int iters)
{
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<256, 256, 65536>>>(in, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 256, 65536>>>(in, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for(auto c : h) avg += c;
  avg /= 256;
  const double n = 36.0 * iters;
  printf("%-44s %8.1f us   %6.1f memtime-ticks/MFMA   %6.1f ns/MFMA\n", name, ms * 1e3, avg / n, ms * 1e6 / n);
}

int main()
{
  float *in, *out;
  long long * cyc;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 8);
  std::vector<float> h(4096);
  for(int i = 0; i < 4096; i++) h[i] = (float)((i * 2654435761u) >> 8) * 1e-9f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 400;
  run<0>("MFMA, fixed operand registers", in, out, cyc, iters);
  run<1>("MFMA fixed + 9 ds_read_b128 per 18 (unused)", in, out, cyc, iters);
  run<2>("MFMA on the data read one k-step earlier", in, out, cyc, iters);
  return 0;
}
