// Microbenchmark (development aid, round 5): v_mfma_i32_32x32x32_i8 — (1) the operand lane map, checked with exact integer data
// against a host product (assumed: lane l = 32 h + r holds A[row r][k = 16 h + j] / B[k = 16 h + j][col r] in byte j of its 16-byte
// operand, C/D as the f16 forms); (2) cycles per instruction beside v_mfma_f32_32x32x16_f16 in the order of skin_kernel_h's GEMM
// slots: per k-step and coordinate one f16 MFMA and one i8 MFMA (two accumulators) against three f16 MFMAs (one accumulator).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_i8_probe tools/micro/mfma_i8_probe.hip && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void layout_kernel(const signed char * A /*[32][32]*/, const signed char * B /*[32][32] k-major*/, int * D /*[32][32]*/)
{
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { signed char b[16]; i32x4 v; } a, b;
  for(int j = 0; j < 16; j++)
  {
    a.b[j] = A[r * 32 + 16 * h + j];
    b.b[j] = B[(16 * h + j) * 32 + r];
  }
  i32x16 c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
  for(int reg = 0; reg < 16; reg++) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg]; // row = (reg&3) + 8 (reg>>2) + 4 h, col = lane & 31
}

template<int MODE>
__global__ __launch_bounds__(256, 1) void rate_kernel(const int * in, float * out, unsigned long long * cyc, int iters)
{
  const int tid = threadIdx.x;
  i32x4 a[4], b[4];
  for(int i = 0; i < 4; i++)
    for(int j = 0; j < 4; j++)
    {
      a[i][j] = in[(tid * 61 + i * 4 + j) & 4095];
      b[i][j] = in[(tid * 37 + i * 4 + j + 977) & 4095];
    }
  f32x16 facc[3] = {};
  i32x16 iacc[3] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for(int it = 0; it < iters; it++)
  {
    if constexpr(MODE == 0) // today's slot: three f16 MFMAs per coordinate into one accumulator
    {
#pragma unroll
      for(int m = 0; m < 36; m++)
        facc[(m / 3) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), facc[(m / 3) % 3], 0, 0, 0);
    }
    else if constexpr(MODE == 1) // one f16 + one i8 per coordinate
    {
#pragma unroll
      for(int m = 0; m < 24; m++)
      {
        if(m & 1)
          iacc[(m / 2) % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m & 3], b[(m >> 2) & 3], iacc[(m / 2) % 3], 0, 0, 0);
        else
          facc[(m / 2) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), facc[(m / 2) % 3], 0, 0, 0);
      }
    }
    else // i8 alone, 36 per turn
    {
#pragma unroll
      for(int m = 0; m < 36; m++) iacc[m % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m & 3], b[(m >> 2) & 3], iacc[m % 3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for(int i = 0; i < 3; i++)
    for(int j = 0; j < 16; j++) s += facc[i][j] + (float)iacc[i][j];
  out[blockIdx.x * 256 + tid] = s;
  if(tid == 0)
  {
    cyc[blockIdx.x * 2] = t1 - t0;
    cyc[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

int main()
{
  // ---- layout
  std::vector<signed char> A(32 * 32), B(32 * 32);
  for(int i = 0; i < 32; i++)
    for(int k = 0; k < 32; k++)
    {
      A[i * 32 + k] = (signed char)(((i * 7 + k * 3) % 23) - 11);
      B[k * 32 + i] = (signed char)(((k * 5 + i * 11 + 3) % 19) - 9); // asymmetric
    }
  std::vector<int> ref(32 * 32, 0), got(32 * 32, -1);
  for(int i = 0; i < 32; i++)
    for(int j = 0; j < 32; j++)
      for(int k = 0; k < 32; k++) ref[i * 32 + j] += (int)A[i * 32 + k] * (int)B[k * 32 + j];
  signed char *dA, *dB;
  int * dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(got.data(), dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for(int i = 0; i < 1024; i++) bad += got[i] != ref[i];
  printf("v_mfma_i32_32x32x32_i8 operand map (lane 32 h + r: k = 16 h + j in byte j; C/D as the f16 forms): %s (%d of 1024 differ)\n", bad ? "WRONG" : "confirmed", bad);
  // ---- rate
  const int nb = 256 * 1, iters = 2000;
  std::vector<int> hin(4096);
  for(auto & x : hin) x = rand();
  int * din; float * dout; unsigned long long * dc;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dout, nb * 256 * 4); hipMalloc(&dc, nb * 16);
  hipMemcpy(din, hin.data(), 4096 * 4, hipMemcpyHostToDevice);
  auto run = [&](auto kern, const char * name, int per_turn) {
    for(int rep = 0; rep < 3; rep++)
    {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      kern<<<nb, 256>>>(din, dout, dc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hc(nb * 2);
      hipMemcpy(hc.data(), dc, nb * 16, hipMemcpyDeviceToHost);
      double cy = 0, rt = 0;
      for(int i = 0; i < nb; i++) { cy += hc[2 * i]; rt += hc[2 * i + 1]; }
      cy /= nb; rt /= nb;
      if(rep == 2)
        printf("%-44s %.1f cycles per MFMA, clock %.2f GHz, %.3f ms for %d turns of %d MFMAs\n", name, cy / ((double)iters * per_turn), cy / (rt * 10.0) , ms, iters, per_turn);
    }
  };
  run(rate_kernel<0>, "3 x f16 per (k-step, coordinate)", 36);
  run(rate_kernel<1>, "1 x f16 + 1 x i8 per (k-step, coordinate)", 24);
  run(rate_kernel<2>, "i8 alone", 36);
  return bad ? 1 : 0;
}
