// probe (round 4): how long ONE workgroup takes to pull the live block of a capture solve's Jacobian (123 rows x 75 columns of
// doubles out of a [164][D] array) into LDS by LDS-DMA, 16 bytes per lane, for D = 157 (rows only 8-byte aligned) and D = 158
// (16-byte aligned rows); cold (array last written by another kernel) and warm (second pass).  8 workgroups, one per "chain".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((address_space(3))) void * lds_ptr_t;
__global__ void fill(double * p, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) p[i] = (double)(i % 977) * 0.001;
}
template<int BYTES>
__global__ __launch_bounds__(256) void pull(const double * J, int D, int rows, int W, unsigned long long * out, double * chk)
{
  extern __shared__ __attribute__((aligned(16))) double Jc[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double * src = J + (size_t)blockIdx.x * 164 * D;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, 164 * D * 8, 0x00020000);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  constexpr int PER = BYTES / 4;             // dwords per lane
  const int SPR = (2 * W + PER - 1) / PER;    // slots per row
  const int cnt = rows * SPR;
  for(int base = wave * 64; base < cnt; base += 256)
  {
    const int d = base + lane;
    const int row = d / SPR, within = d - row * SPR;
    const int r = row + row / 3; // three of every four rows are live
    const int voff = d < cnt ? (r * D * 2 + within * PER) * 4 : 0x7ffffff0;
    if constexpr(BYTES == 16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(reinterpret_cast<unsigned char *>(Jc) + (size_t)base * 16), 16, voff, 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(reinterpret_cast<unsigned char *>(Jc) + (size_t)base * 4), 4, voff, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if(tid == 0) out[blockIdx.x] = t1 - t0;
  if(tid == 0) out[8 + blockIdx.x] = r1 - r0;
  if(tid == 0) chk[blockIdx.x] = Jc[5] + Jc[SPR * PER / 2 * 7 + 3];
}
__global__ __launch_bounds__(256) void pull_regs(const double * J, int D, int rows, int W, unsigned long long * out, double * chk)
{
  extern __shared__ __attribute__((aligned(16))) double Jc[];
  const int tid = threadIdx.x;
  const double * src = J + (size_t)blockIdx.x * 164 * D;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int cnt = rows * W;
  for(int q0 = 0; q0 < cnt; q0 += 256 * 16)
  {
    double t[16];
#pragma unroll
    for(int u = 0; u < 16; u++)
    {
      const int q = q0 + u * 256 + tid, qq = q < cnt ? q : cnt - 1;
      const int rr = qq / W;
      t[u] = src[(size_t)(rr + rr / 3) * D + (qq - rr * W)];
    }
#pragma unroll
    for(int u = 0; u < 16; u++)
    {
      const int q = q0 + u * 256 + tid;
      if(q < cnt) Jc[q] = t[u];
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if(tid == 0) out[blockIdx.x] = t1 - t0;
  if(tid == 0) chk[blockIdx.x] = Jc[5];
}
// the way ik_eval_kernel leaves J: 8-byte stores, a thread per (row, column) in row-strided order from 256-thread workgroups
__global__ void fill_scattered(double * p, int D, int nb)
{
  const int b = blockIdx.x / 32, part = blockIdx.x % 32;
  for(int r = part; r < 164; r += 32)
    for(int c = threadIdx.x; c < D; c += 256) p[((size_t)b * 164 + r) * D + c] = (double)((r * 31 + c) % 977) * 0.001;
}
int main()
{
  const int NB = 8;
  unsigned long long * out; double * chk; hipMalloc(&out, 8 * NB * 2); hipMemset(out, 0, 16 * NB); hipMalloc(&chk, 8 * NB);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&pull<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&pull<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&pull_regs), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for(int D : {157, 158})
  {
    const size_t n = (size_t)NB * 164 * D;
    double * J; hipMalloc(&J, n * 8);
    for(int mode = 0; mode < 3; mode++)
      for(int pass = 0; pass < 2; pass++)
      {
        if(pass == 0 && getenv("SCATTER")) fill_scattered<<<NB * 32, 256>>>(J, D, NB); // (no synchronisation: same stream, like eval -> solve)
        else if(pass == 0) { fill<<<(unsigned)((n + 255) / 256), 256>>>(J, n); hipDeviceSynchronize(); }
        if(mode == 0) pull<16><<<NB, 256, 100 * 1024>>>(J, D, 123, 75, out, chk);
        if(mode == 1) pull<4><<<NB, 256, 100 * 1024>>>(J, D, 123, 75, out, chk);
        if(mode == 2) pull_regs<<<NB, 256, 100 * 1024>>>(J, D, 123, 75, out, chk);
        hipDeviceSynchronize();
        unsigned long long h[2 * NB]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mn = ~0ull; for(int i = 0; i < NB; i++) { auto v = h[i]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        printf("D %d %s %s: %llu .. %llu shader cycles; workgroup 0: %llu ticks of 10 ns = %.2f us -> %.2f GHz\n", D, mode == 0 ? "dma16" : (mode == 1 ? "dma4 " : "regs "), pass ? "warm" : "cold", mn, mx, h[8], h[8] * 0.01, mode < 2 ? h[0] / (h[8] * 10.0) : 0.0);
      }
    hipFree(J);
  }
}
