// Microbenchmark (development aid, round 6): can a SECOND wavefront per SIMD carry the skinning beside a matrix wavefront?
// skin_kernel_e runs one wavefront per SIMD that issues an item's 252 MFMAs AND the previous item's ~1100 vector instructions and
// ~200 LDS reads in order: 11.6-11.9 k cycles per item against 8.5 k for the MFMAs alone.  Here, per "k-step" (one s_barrier):
//   matrix role:   18 v_mfma_f32_32x32x16_bf16 (three accumulators x 6, dependent), 9 ds_read_b128 (operand fragments), [5 LDS-DMA pieces]
//   skinning role: 14 ds_read_b128 at per-lane addresses (matrix rows), 4 FMAs per read + 16 chained, one 12-byte store per two k-steps
// MODE 0: 256 threads, both roles in every wavefront (today's shape)      MODE 1: 512 threads, wavefronts 0-3 matrix, 4-7 skinning
// MODE 2: 512 threads, wavefronts 4-7 only join the barriers (matrix alone) MODE 3: as 1 with s_setprio 2 on the matrix wavefronts
// One workgroup per CU on every CU (the clock the chip holds is part of the answer).
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -o /tmp/mfma_beside_valu tools/micro/mfma_beside_valu.hip && /tmp/mfma_beside_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
#define ESB() __builtin_amdgcn_sched_barrier(0)

constexpr int LDS_G = 72 * 1024, LDS_RING = 60 * 1024, LDS_TOTAL = LDS_G + LDS_RING;

__device__ __forceinline__ void kbarrier()
{
  asm volatile("s_barrier" ::: "memory"); // (the compiler waits for each LDS read at its first use; nothing is written to LDS by lanes)
}

template<int I>
__device__ __forceinline__ void dma_piece(const __amdgpu_buffer_rsrc_t rs, unsigned char * dst, int voff, int soff)
{
  typedef __attribute__((address_space(3))) void * lds_ptr_t;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, soff, I * 1024, 0);
}
__device__ __forceinline__ void dma_piece_n(int i, const __amdgpu_buffer_rsrc_t rs, unsigned char * dst, int voff, int soff)
{
  switch(i)
  {
  case 0: dma_piece<0>(rs, dst, voff, soff); break;
  case 1: dma_piece<1>(rs, dst, voff, soff); break;
  case 2: dma_piece<2>(rs, dst, voff, soff); break;
  case 3: dma_piece<3>(rs, dst, voff, soff); break;
  default: dma_piece<4>(rs, dst, voff, soff); break;
  }
}

template<int MODE, bool DMA>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, 1) void probe(const float * __restrict__ in, const unsigned char * __restrict__ stream,
                                                                   float * __restrict__ out, unsigned long long * __restrict__ cyc, int ksteps)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for(int i = tid; i < LDS_TOTAL / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = in[(i * 7 + blockIdx.x) & 65535];
  __syncthreads();
  const int mw = wave & 3, jlane = lane >> 3;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(stream), 0, 1 << 26, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(out, 0, 64 * 4194304 + (1 << 23), 0x00020000);
  float result = 0.f;
  unsigned long long t0 = 0, r0 = 0;
  // ---- the two roles are two LOOPS (a wavefront runs one of them): their registers overlap
  // (addresses as the kernel forms them: a per-lane base per joint, the row and the matrix row in the instruction's offset)
  const unsigned char * gj[4];
  for(int j = 0; j < 4; j++) gj[j] = lds + (lane >> 5) * 1152 + ((jlane * 5 + j * 7) % 24) * 48;
  auto g_row = [&](int q, int k) { (void)k; return *reinterpret_cast<const v4f *>(gj[q & 3] + 2 * q * 1152 + (q % 3) * 16); };
  if(MODE == 0)
  {
    // today's shape: every slot one MFMA, about four vector instructions and 0.8 LDS reads of the skinning
    v4f areg[6], bfr[9];
    for(int i = 0; i < 6; i++)
      for(int j = 0; j < 4; j++) areg[i][j] = in[(tid * 61 + i * 4 + j) & 65535];
    for(int i = 0; i < 9; i++) bfr[i] = *reinterpret_cast<const v4f *>(lds + LDS_G + (i * 64 + lane) * 16);
    f32x16 acc[3] = {};
    float w[4] = {in[tid & 65535], in[(tid + 1) & 65535], in[(tid + 2) & 65535], in[(tid + 3) & 65535]};
    v4f m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0;
    float rx = in[(tid + 9) & 65535], ry = in[(tid + 10) & 65535], rz = in[(tid + 11) & 65535], ox = 0.f, oy = 0.f, oz = 0.f;
    v4f gq[6];
    for(int q = 0; q < 6; q++) gq[q] = g_row(q, 0);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for(int ks = 0; ks < ksteps; ks++)
    {
      const int img = LDS_G + (ks % 3) * (20 * 1024);
#pragma unroll
      for(int s = 0; s < 18; s++)
      {
        acc[s / 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[s % 6]), __builtin_bit_cast(bf16x8, bfr[(s / 6) * 3 + s % 3]), acc[s / 6], 0, 0, 0);
        ESB();
        if(s >= 6 && s < 15) bfr[s - 6] = *reinterpret_cast<const v4f *>(lds + img + ((s - 6) * 64 + lane) * 16 + (mw >> 1) * 9216);
        if(DMA && s >= 7 && s < 12)
          dma_piece_n(s - 7, rs, lds + LDS_G + ((ks + 1) % 3) * (20 * 1024) + mw * 5120, lane * 16 + mw * 5120, (((ks % 196) * 20480 + (blockIdx.x & 7) * 196 * 20480) & ((1 << 25) - 1)));
        if(s < 14)
        {
          // the row used six slots from now is requested here (a ring of six register sets)
          const v4f g = gq[s % 6];
          gq[s % 6] = g_row((s + 6) % 14, ks + (s + 6) / 14);
          v4f & mm = (s % 3 == 0 ? m0 : (s % 3 == 1 ? m1 : m2));
          const float ww = w[s & 3];
          mm.x = __builtin_fmaf(ww, g.x, mm.x);
          mm.y = __builtin_fmaf(ww, g.y, mm.y);
          mm.z = __builtin_fmaf(ww, g.z, mm.z);
          mm.w = __builtin_fmaf(ww, g.w, mm.w);
          ox = __builtin_fmaf(mm.x, rx, ox);
        }
        ESB();
      }
      oy = __builtin_fmaf(m1.y, ry, oy);
      oz = __builtin_fmaf(m2.z, rz, oz);
      if(ks & 1)
      {
        v3f ov = {ox, oy, oz}; // (a buffer store as in the kernel: its data is read at issue, no wait for its completion follows)
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rso, (blockIdx.x * 512 + tid) * 16, (ks & 63) * 4194304, 0);
        ESB();
        asm volatile("s_nop 1");
        ESB();
      }
      if(DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      kbarrier();
    }
    result = ox + oy + oz + m0.x + m1.y + m2.z;
    for(int i = 0; i < 3; i++)
      for(int j = 0; j < 16; j++) result += acc[i][j];
  }
  else if(wave < 4)
  {
    // matrix wavefront: 18 MFMAs, the next k-step's nine fragments behind the sixth, five DMA pieces
    v4f areg[6], bfr[9];
    for(int i = 0; i < 6; i++)
      for(int j = 0; j < 4; j++) areg[i][j] = in[(tid * 61 + i * 4 + j) & 65535];
    for(int i = 0; i < 9; i++) bfr[i] = *reinterpret_cast<const v4f *>(lds + LDS_G + (i * 64 + lane) * 16);
    f32x16 acc[3] = {};
    if(MODE == 3) __builtin_amdgcn_s_setprio(2);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for(int ks = 0; ks < ksteps; ks++)
    {
      const int img = LDS_G + (ks % 3) * (20 * 1024);
      if(MODE != 4)
      {
#pragma unroll
        for(int s = 0; s < 18; s++)
        {
          acc[s / 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[s % 6]), __builtin_bit_cast(bf16x8, bfr[(s / 6) * 3 + s % 3]), acc[s / 6], 0, 0, 0);
          ESB();
          if(s >= 6 && s < 15) bfr[s - 6] = *reinterpret_cast<const v4f *>(lds + img + ((s - 6) * 64 + lane) * 16 + (mw >> 1) * 9216);
          if(DMA && s >= 7 && s < 12)
            dma_piece_n(s - 7, rs, lds + LDS_G + ((ks + 1) % 3) * (20 * 1024) + mw * 5120, lane * 16 + mw * 5120, (((ks % 196) * 20480 + (blockIdx.x & 7) * 196 * 20480) & ((1 << 25) - 1)));
          ESB();
        }
        if(DMA) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      }
      kbarrier();
    }
    for(int i = 0; i < 3; i++)
      for(int j = 0; j < 16; j++) result += acc[i][j];
  }
  else
  {
    // skinning wavefront: per k-step 14 reads (each six rows ahead of its use), 56 + 16 FMAs, a store every second k-step
    float w[4] = {in[tid & 65535], in[(tid + 1) & 65535], in[(tid + 2) & 65535], in[(tid + 3) & 65535]};
    v4f m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0;
    float rx = in[(tid + 9) & 65535], ry = in[(tid + 10) & 65535], rz = in[(tid + 11) & 65535], ox = 0.f, oy = 0.f, oz = 0.f;
    v4f gq[6];
    for(int q = 0; q < 6; q++) gq[q] = g_row(q, 0);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for(int ks = 0; ks < ksteps; ks++)
    {
      if(MODE != 2)
      {
#pragma unroll
        for(int s = 0; s < 14; s++)
        {
          const v4f g = gq[s % 6];
          gq[s % 6] = g_row((s + 6) % 14, ks + (s + 6) / 14);
          v4f & mm = (s % 3 == 0 ? m0 : (s % 3 == 1 ? m1 : m2));
          const float ww = w[s & 3];
          mm.x = __builtin_fmaf(ww, g.x, mm.x);
          mm.y = __builtin_fmaf(ww, g.y, mm.y);
          mm.z = __builtin_fmaf(ww, g.z, mm.z);
          mm.w = __builtin_fmaf(ww, g.w, mm.w);
          ox = __builtin_fmaf(mm.x, rx, ox);
          ESB();
        }
        oy = __builtin_fmaf(m1.y, ry, oy);
        oz = __builtin_fmaf(m2.z, rz, oz);
        if(ks & 1)
        {
          v3f ov = {ox, oy, oz}; // (a buffer store as in the kernel: its data is read at issue, no wait for its completion follows)
          __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3u, ov), rso, (blockIdx.x * 512 + tid) * 16, (ks & 63) * 4194304, 0);
          ESB();
          asm volatile("s_nop 1");
          ESB();
        }
      }
      kbarrier();
    }
    result = ox + oy + oz + m0.x + m1.y + m2.z;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * 512 + tid] = result;
  if(tid == 0)
  {
    cyc[blockIdx.x * 2] = t1 - t0;
    cyc[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

template<int MODE, bool DMA>
static int run(const float * in, const unsigned char * stream, float * out, unsigned long long * cyc, const char * what)
{
  const int ksteps = 14 * 7 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
  for(int rep = 0; rep < 3; rep++)
  {
    probe<MODE, DMA><<<256, MODE == 0 ? 256 : 512, LDS_TOTAL>>>(in, stream, out, cyc, ksteps);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(512);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost));
  double c = 0, r = 0;
  for(int i = 0; i < 256; i++) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
  c /= 256; r /= 256;
  printf("%-62s %7.1f cycles per k-step (%.1f per MFMA), %6.3f us per k-step, clock %.0f MHz\n", what, c / ksteps, c / ksteps / 18, r / 100.0 / ksteps, c / r * 100.0);
  return 0;
}

int main()
{
  float * in; unsigned char * stream; float * out; unsigned long long * cyc;
  CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&stream, 1 << 26)); CK(hipMalloc(&out, (size_t)64 * 1048576 * 4 + (1 << 24))); CK(hipMalloc(&cyc, 512 * 8));
  std::vector<float> h(65536);
  srand(1);
  for(auto & x : h) x = (float)(rand() & 65535) / 65536.0f - 0.5f;
  CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  CK(hipMemset(stream, 0x3c, 1 << 26));
  for(int pass = 0; pass < 2; pass++)
  {
    if(run<2, false>(in, stream, out, cyc, "matrix wavefronts alone (no DMA)")) return 1;
    if(run<2, true>(in, stream, out, cyc, "matrix wavefronts alone + ring DMA")) return 1;
    if(run<4, false>(in, stream, out, cyc, "skinning wavefronts alone (second wavefront of each SIMD)")) return 1;
    if(run<0, false>(in, stream, out, cyc, "one wavefront per SIMD, both roles (today's shape), no DMA")) return 1;
    if(run<0, true>(in, stream, out, cyc, "one wavefront per SIMD, both roles (today's shape) + DMA")) return 1;
    if(run<1, false>(in, stream, out, cyc, "two wavefronts per SIMD, roles split, no DMA")) return 1;
    if(run<1, true>(in, stream, out, cyc, "two wavefronts per SIMD, roles split + DMA")) return 1;
    if(run<3, true>(in, stream, out, cyc, "two wavefronts per SIMD, roles split, matrix at prio 2 + DMA")) return 1;
  }
  return 0;
}
