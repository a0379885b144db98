// Microbenchmark (development aid): bare f16 MFMA loops on random operands in registers, one wavefront per SIMD on every
// CU: cycles per instruction (s_memtime), in-kernel clock (s_memtime / s_memrealtime) and wall time per unit of FLOPs for
//   0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_16x16x32_f16   2: v_mfma_f32_32x32x8_f16 (the K = 8 form)
//   3: 32x32x16 issued as DEPENDENT TRIPLES (acc0 x3, acc1 x3, acc2 x3: the order of skin_kernel_h's GEMM slots)
//   4: 32x32x16 as one dependent chain (a single accumulator)   5: dependent pairs
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shapes tools/micro/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template<int MODE>
__global__ __launch_bounds__(256, 1) void k(const _Float16 * in, float * out, unsigned long long * cyc, int iters)
{
  const int tid = threadIdx.x;
  f16x8 a[4], b[4];
  for(int i = 0; i < 4; i++)
    for(int j = 0; j < 8; j++)
    {
      a[i][j] = in[(tid * 61 + i * 8 + j) & 4095];
      b[i][j] = in[(tid * 37 + i * 8 + j + 977) & 4095];
    }
  f32x16 acc32[3] = {};
  f32x4 acc16[12] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for(int it = 0; it < iters; it++)
  {
    if constexpr(MODE == 0)
    {
#pragma unroll
      for(int m = 0; m < 36; m++) acc32[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc32[m % 3], 0, 0, 0);
    }
    else if constexpr(MODE == 3)
    {
#pragma unroll
      for(int m = 0; m < 36; m++) acc32[(m / 3) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc32[(m / 3) % 3], 0, 0, 0);
    }
    else if constexpr(MODE == 4)
    {
#pragma unroll
      for(int m = 0; m < 36; m++) acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc32[0], 0, 0, 0);
    }
    else if constexpr(MODE == 5)
    {
#pragma unroll
      for(int m = 0; m < 36; m++) acc32[(m / 2) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc32[(m / 2) % 3], 0, 0, 0);
    }
    else if constexpr(MODE == 1)
    {
#pragma unroll
      for(int m = 0; m < 72; m++) acc16[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 3], b[(m >> 2) & 3], acc16[m % 12], 0, 0, 0);
    }
    else
    {
#pragma unroll
      for(int m = 0; m < 36; m++)
      {
        f16x4 a4 = {a[m & 3][0], a[m & 3][1], a[m & 3][2], a[m & 3][3]}, b4 = {b[(m >> 2) & 3][0], b[(m >> 2) & 3][1], b[(m >> 2) & 3][2], b[(m >> 2) & 3][3]};
        acc32[m % 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc32[m % 3], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for(int x = 0; x < 3; x++)
    for(int r = 0; r < 16; r++) s += acc32[x][r];
  for(int x = 0; x < 12; x++)
    for(int r = 0; r < 4; r++) s += acc16[x][r];
  out[blockIdx.x * 256 + tid] = s;
  if(tid == 0)
  {
    cyc[blockIdx.x * 2] = t1 - t0;
    cyc[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template<int MODE>
static void run(const char * name, int per_iter, double flop_per_instr, const _Float16 * din, float * dout, unsigned long long * dcyc)
{
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for(int rep = 0; rep < 3; rep++) k<MODE><<<blocks, 256>>>(din, dout, dcyc, iters);
  hipEventRecord(e0);
  for(int rep = 0; rep < 5; rep++) k<MODE><<<blocks, 256>>>(din, dout, dcyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  std::vector<unsigned long long> h(blocks * 2);
  hipMemcpy(h.data(), dcyc, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for(int i = 0; i < blocks; i++)
  {
    c += (double)h[2 * i];
    r += (double)h[2 * i + 1];
  }
  const double instr = (double)iters * per_iter;
  std::printf("%-26s %6.2f cycles/instr, clock %4.0f MHz, %7.3f ms, %7.1f TFLOP/s\n", name, c / blocks / instr, c / r * 100.0, ms,
              instr * flop_per_instr * blocks * 4 / (ms * 1e-3) / 1e12);
}

int main()
{
  std::vector<_Float16> hin(4096);
  unsigned s = 12345;
  for(auto & v : hin)
  {
    s = s * 1664525u + 1013904223u;
    v = (_Float16)(((float)(s >> 8) / 8388608.0f - 1.0f) * 4.0f);
  }
  _Float16 * din;
  float * dout;
  unsigned long long * dcyc;
  hipMalloc(&din, 4096 * 2);
  hipMalloc(&dout, 256 * 256 * 4);
  hipMalloc(&dcyc, 256 * 16);
  hipMemcpy(din, hin.data(), 4096 * 2, hipMemcpyHostToDevice);
  for(int round = 0; round < 2; round++)
  {
    run<0>("v_mfma_f32_32x32x16_f16", 36, 2.0 * 32 * 32 * 16, din, dout, dcyc);
    run<1>("v_mfma_f32_16x16x32_f16", 72, 2.0 * 16 * 16 * 32, din, dout, dcyc);
    run<2>("v_mfma_f32_32x32x8_f16", 36, 2.0 * 32 * 32 * 8, din, dout, dcyc);
    run<3>("32x32x16 dependent triples", 36, 2.0 * 32 * 32 * 16, din, dout, dcyc);
    run<5>("32x32x16 dependent pairs", 36, 2.0 * 32 * 32 * 16, din, dout, dcyc);
    run<4>("32x32x16 one chain", 36, 2.0 * 32 * 32 * 16, din, dout, dcyc);
  }
  return 0;
}
