// Probe (development aid): operand / result lane maps of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double * out)
{
  const int l = threadIdx.x;
  // hypothesis: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]
  // run 1: A[i][k] = i (if k == 0), B[k][j] = 1 (if k == 0)  -> D[i][j] = i
  // run 2: A[i][k] = 1 (k == 0),    B[k][j] = j (k == 0)      -> D[i][j] = j
  // run 3: A[i][k] = k + 1,          B[k][j] = 10^k           -> D = sum_k (k+1) 10^k = 4321 if the k maps agree
  d4 c = {0, 0, 0, 0};
  d4 r1 = __builtin_amdgcn_mfma_f64_16x16x4f64((l / 16 == 0) ? (double)(l % 16) : 0.0, (l / 16 == 0) ? 1.0 : 0.0, c, 0, 0, 0);
  d4 r2 = __builtin_amdgcn_mfma_f64_16x16x4f64((l / 16 == 0) ? 1.0 : 0.0, (l / 16 == 0) ? (double)(l % 16) : 0.0, c, 0, 0, 0);
  const double p10[4] = {1, 10, 100, 1000};
  d4 r3 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(l / 16 + 1), p10[l / 16], c, 0, 0, 0);
  for(int i = 0; i < 4; i++)
  {
    out[(0 * 64 + l) * 4 + i] = r1[i];
    out[(1 * 64 + l) * 4 + i] = r2[i];
    out[(2 * 64 + l) * 4 + i] = r3[i];
  }
}
int main()
{
  double * d;
  hipMalloc(&d, 3 * 64 * 4 * 8);
  k<<<1, 64>>>(d);
  double h[3 * 64 * 4];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for(int l : {0, 1, 15, 16, 17, 32, 48, 63})
    printf("lane %2d: row(i) regs = %g %g %g %g | col(j) regs = %g %g %g %g | ksum = %g\n", l, h[(0 * 64 + l) * 4], h[(0 * 64 + l) * 4 + 1],
           h[(0 * 64 + l) * 4 + 2], h[(0 * 64 + l) * 4 + 3], h[(64 + l) * 4], h[(64 + l) * 4 + 1], h[(64 + l) * 4 + 2], h[(64 + l) * 4 + 3],
           h[(128 + l) * 4]);
  return 0;
}
