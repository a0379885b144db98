// probe: butterfly sum of doubles over a wavefront by __shfl_xor (used for |e|^2 in ik_solve_kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double * e, int rows, double * out)
{
  __shared__ double ebuf[256];
  const int tid = threadIdx.x;
  if(tid < rows) ebuf[tid] = e[tid];
  __syncthreads();
  if(tid < 64)
  {
    double v[3];
#pragma unroll
    for(int a = 0; a < 3; a++) v[a] = (tid + 64 * a < rows) ? ebuf[tid + 64 * a] : 0.0;
    double s = v[0] * v[0];
    s = fma(v[1], v[1], s);
    s = fma(v[2], v[2], s);
#pragma unroll
    for(int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if(tid == 0) out[0] = s;
    if(tid == 5) out[1] = s;
  }
}
int main()
{
  double h[256], *d, *o, r[2];
  for(int i = 0; i < 256; i++) h[i] = 0.1 * (i + 1);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for(int rows : {24, 64, 100, 164})
  {
    k<<<1, 256>>>(d, rows, o);
    hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    double ref = 0; for(int i = 0; i < rows; i++) ref += h[i] * h[i];
    printf("rows %d: %.10g %.10g ref %.10g\n", rows, r[0], r[1], ref);
  }
}
