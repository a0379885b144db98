// Development aid: semantics of v_dot2c_f32_f16 (__builtin_amdgcn_fdot2) at large fp16 magnitudes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float * out)
{
  h2 a = {(_Float16)12000.0f, (_Float16)3.0f}, b = {(_Float16)9000.0f, (_Float16)-5.0f};
  out[0] = __builtin_amdgcn_fdot2(a, b, 1.0f, false);                 // 1.08e8 - 15 + 1
  h2 c = {(_Float16)0.5f, (_Float16)2.0e-5f}, d = {(_Float16)3.0e-5f, (_Float16)4.0f}; // subnormal-range halves
  out[1] = __builtin_amdgcn_fdot2(c, d, 0.0f, false);                 // 1.5e-5 + 8e-5
  float acc = 0.f;
  for(int i = 0; i < 512; i++)
  {
    h2 x = {(_Float16)(100.0f + i), (_Float16)(-50.0f + i)}, y = {(_Float16)(7.0f - i), (_Float16)(3.0f + i)};
    acc = __builtin_amdgcn_fdot2(x, y, acc, false);
  }
  out[2] = acc;
}
int main()
{
  float * d, h[3];
  (void)hipMalloc(&d, 12);
  k<<<1, 1>>>(d);
  (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  double ref = 0;
  for(int i = 0; i < 512; i++) ref += (100.0 + i) * (7.0 - i) + (-50.0 + i) * (3.0 + i);
  std::printf("big %.9g (expect 107999986)  small %.9g (expect ~9.5e-05)  chain %.9g (expect %.9g)\n", h[0], h[1], h[2], ref);
  return 0;
}
